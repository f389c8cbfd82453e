"""-m gpu: whole VVIntegrator steps (fused path) against the oracle's step driver on the same seeded inputs,
through the reference-shaped Python surface (VVIntegrator(...).step(n)).  Covers both schemes, all three
precision modes, TGNH + hard wall + cos acceleration (bulk), Langevin + E-field + images (EDL), plain NH (water).
Tolerance: 1e-5 relative on positions and velocities (BASELINE.json north_star), checked after 20 steps in mixed and
double precision (measured: ~1e-15) and after 2 steps in single precision (measured: ~3e-6; per step ~1e-7).  Single
precision drifts faster only because float reductions depend on summation order: the serial float sums of the
one-thread oracle are themselves off in the 6th digit, so longer single-precision runs compare rounding noise."""
import importlib

import numpy as np
import pytest

from oracle import oracle as O

pkg = importlib.import_module("openmm-velocityverlet_amd")
H, I, systems = pkg.vvhip, pkg.integrator, pkg.systems
pytestmark = pytest.mark.gpu


NSTEPS = {"single": 2, "mixed": 20, "double": 20}


def _pair(spec, prec, middle, nsteps, cos=0.0, maxd=0.02, T=333.0, efield=0.0, mirror=0.0, seed_random=1, dt=0.001, oracle_prec=None):
    p = O.Params(temperature=T, drude_temperature=1.0, step_size=dt, max_drude_distance=maxd, cos_acceleration=cos,
                 use_middle_scheme=middle, electric_field=efield, mirror_location=mirror)
    rnd = np.random.default_rng(seed_random).standard_normal((4096, 4)).astype(np.float32)
    osys = O.OracleSystem(spec, p, oracle_prec or prec, random=rnd, force_mode=1)
    it = I.VVIntegrator(T, 10.0, 1.0, 40.0, dt)
    it.setMaxDrudeDistance(maxd)
    it.setCosAcceleration(cos)
    it.setUseMiddleScheme(middle)
    it.setElectricField(efield)
    it.setMirrorLocation(mirror)
    ctx = I.Context(spec, it, precision=prec, force_provider="tether", random=rnd)
    osys.step(nsteps)
    it.step(nsteps)
    return osys, ctx, it


def _check(osys, ctx, prec, tol=1e-5, label=""):
    x_o, x_g = osys.positions(), ctx.getPositions()
    v_o, v_g = osys.velm[:, :3].astype(np.float64), ctx.getVelocities()
    massive = osys.velm[:, 3] != 0
    ex = np.abs(x_g - x_o).max() / np.abs(x_o).max()
    ev = np.abs(v_g[massive] - v_o[massive]).max() / np.abs(v_o[massive]).max()
    assert np.isfinite(x_g).all() and np.isfinite(v_g[massive]).all(), label
    assert ex < tol and ev < tol, f"{label}: rel err pos {ex:.2e} vel {ev:.2e}"
    st = ctx.getNHState()
    ke_o, ke_g = osys.ke2(), np.array(list(st.ke2))
    ntg = osys.s.num_tg
    assert np.allclose(ke_g[:ntg], ke_o[:ntg], rtol=20 * tol), f"{label}: 2KE {ke_g} vs {ke_o}"
    vs_o, vs_g = osys.vscale(), np.array(list(st.vscale))
    assert np.allclose(vs_g, vs_o, rtol=0, atol=20 * tol), f"{label}: vscale {vs_g} vs {vs_o}"
    return ex, ev


@pytest.mark.parametrize("prec", O.PRECISIONS)
@pytest.mark.parametrize("middle", [True, False])
@pytest.mark.parametrize("cos", [0.0, 0.02])
def test_bulk_drude_il(prec, middle, cos):
    spec = systems.drude_il(cells=(1, 1, 1), pairs_per_cell=40, seed=7)
    osys, ctx, it = _pair(spec, prec, middle, nsteps=NSTEPS[prec], cos=cos)
    try:
        ex, ev = _check(osys, ctx, prec, label=f"bulk/{prec}/middle={middle}/cos={cos}")
        # chain state (device, fp64) against the host-double chain of the oracle
        st, ch = ctx.getNHState(), osys.chain_state()
        # eta_dot[0] ~ (2KE - target)/Q is a difference of nearly equal numbers: a float KE sum (single mode, ~1e-6
        # relative, order dependent) is amplified ~1000x; in mixed/double the device chain tracks the host one to ~1e-12
        rtol = 2e-2 if prec == "single" else 1e-8
        for g in range(3):
            assert np.allclose(list(st.eta_dot[g])[:3], ch["eta_dot"][g][:3], rtol=rtol, atol=1e-12 if prec != "single" else 1e-7)
        if cos != 0:
            v_g, inv_g = it.getViscosity()
            v_o, inv_o = osys.viscosity()
            assert v_g == pytest.approx(v_o, rel=1e-4, abs=1e-9) and inv_g == pytest.approx(inv_o, rel=1e-4, abs=1e-9)
        print(f"bulk/{prec}/middle={middle}/cos={cos}: rel err pos {ex:.2e} vel {ev:.2e}")
    finally:
        ctx.close()


@pytest.mark.parametrize("prec", O.PRECISIONS)
@pytest.mark.parametrize("middle", [True, False])
def test_edl_langevin_efield_images(prec, middle):
    spec = systems.edl_slab(num_ion_pairs=20, num_electrode=60, seed=9)
    lz = float(spec.box[2])
    osys, ctx, it = _pair(spec, prec, middle, nsteps=NSTEPS[prec], mirror=lz / 2, efield=2.0 / lz * 2 * 1.602176634e-22)
    try:
        _check(osys, ctx, prec, label=f"edl/{prec}/middle={middle}")
        # image particles: x,y bit copies of the parent, z mirrored (north_star: bit-exact index mirroring)
        posq = ctx.getPosq()
        ip = np.array(spec.image_pairs)
        assert np.array_equal(posq[ip[:, 0], :2].view(np.uint8), posq[ip[:, 1], :2].view(np.uint8))
        # z: the mirror arithmetic of K/imageCharge.cu:10-26 redone on the GPU's own parent positions must give the stored image
        # bits exactly (no tolerance: this does not depend on how far the trajectory has drifted from the oracle's); w: the image
        # keeps its own charge
        mirror = lz / 2
        if prec == "mixed":
            corr = ctx.getPosqCorrection()
            zz = mirror * 2 - (posq[ip[:, 1], 2].astype(np.float64) + corr[ip[:, 1], 2].astype(np.float64))
            assert np.array_equal(posq[ip[:, 0], 2].view(np.uint8), zz.astype(np.float32).view(np.uint8))
            assert np.array_equal(corr[ip[:, 0], 2].view(np.uint8), (zz - zz.astype(np.float32)).astype(np.float32).view(np.uint8))
            assert np.array_equal(corr[ip[:, 0], :2].view(np.uint8), corr[ip[:, 1], :2].view(np.uint8))
        else:
            M = np.float32 if prec == "single" else np.float64
            zz = (2 * M(mirror) - posq[ip[:, 1], 2].astype(M)).astype(posq.dtype)
            assert np.array_equal(posq[ip[:, 0], 2].view(np.uint8), zz.view(np.uint8))
        assert np.array_equal(posq[ip[:, 0], 3], spec.charges[ip[:, 0]].astype(posq.dtype))
        # ... and the oracle's images agree to the trajectory tolerance
        assert np.allclose(posq[ip[:, 0]], osys.posq[ip[:, 0]], rtol=1e-5)
    finally:
        ctx.close()


@pytest.mark.parametrize("prec", O.PRECISIONS)
@pytest.mark.parametrize("middle", [True, False])
def test_water_plain_nh(prec, middle):
    spec = systems.spce_water(300, seed=5)
    osys, ctx, it = _pair(spec, prec, middle, nsteps=NSTEPS[prec], maxd=0.0, T=300.0, dt=0.002)
    try:
        assert ctx.info.num_temp_groups == 1 and not ctx.info.use_com_temp_group
        _check(osys, ctx, prec, label=f"water/{prec}/middle={middle}")
    finally:
        ctx.close()


def test_split_entry_points_equal_fused_step():
    """The per-KernelImpl entry points called in VVIntegrator::stepMiddle's order (what the OpenMM adapter does)
    must land on the same state as the fused vvhip_step_middle."""
    spec = systems.drude_il(cells=(1, 1, 1), pairs_per_cell=30, seed=3)
    outs = []
    for split in (False, True):
        it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
        it.setMaxDrudeDistance(0.02)
        it.setCosAcceleration(0.02)
        ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
        L, p = H.lib, ctx.plan
        for _ in range(5):
            ctx.calcForces()
            if not split:
                H.check(L.vvhip_step_middle(p, 0), p)
            else:                                                     # openmmapi/src/VVIntegrator.cpp:238-268
                for fn in (L.vvhip_reset_extra_force, L.vvhip_apply_cosine_force, L.vvhip_middle_kick, L.vvhip_middle_half_drift1,
                           L.vvhip_calc_velocity_bias, L.vvhip_remove_velocity_bias, L.vvhip_scale_velocity,
                           L.vvhip_restore_velocity_bias, L.vvhip_middle_half_drift2, L.vvhip_middle_finish):
                    H.check(fn(p), p)
        outs.append((ctx.getPositions(), ctx.getVelocities()))
        ctx.close()
    assert np.allclose(outs[0][0], outs[1][0], rtol=0, atol=1e-12) and np.allclose(outs[0][1], outs[1][1], rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("middle", [True, False])
def test_graph_replay_equals_eager_and_is_bit_reproducible(middle):
    spec = systems.drude_il(cells=(1, 1, 1), pairs_per_cell=60, seed=4)
    res = []
    for mode in ("eager", "graph", "graph"):
        it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
        it.setMaxDrudeDistance(0.02)
        it.setUseMiddleScheme(middle)
        ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
        if mode == "eager":
            it.step(24)
        else:
            ctx.run_graph(24, steps_per_graph=8)
        res.append((ctx.getPosq(), ctx.getVelm()))
        ctx.close()
    for a, b in ((0, 1), (1, 2)):       # fixed-point accumulators => identical bits run to run and eager vs graph
        assert np.array_equal(res[a][0].view(np.uint8), res[b][0].view(np.uint8))
        assert np.array_equal(res[a][1].view(np.uint8), res[b][1].view(np.uint8))


def test_full_size_c3_properties():
    """BASELINE.json C3 at full size (111 000 particles): size-independent properties instead of the oracle."""
    spec = systems.make_config("C3")
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
    it.setMaxDrudeDistance(0.02)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    try:
        v0 = ctx.getVelm()
        it.step(1)
        st = ctx.getNHState()
        # 2KE of the three groups recomputed in numpy from the pre-step state is not available after the kick, so check
        # (a) equipartition targets are the right order of magnitude, (b) scale factors are near 1, (c) nothing blew up
        dof, nkbt = np.array(list(ctx.info.dof)), np.array(list(ctx.info.nkbt))
        ke2 = np.array(list(st.ke2))
        assert (ke2 > 0).all() and np.all(np.abs(np.array(list(st.vscale)) - 1) < 5e-2)
        T = ke2 / dof / O.BOLTZ
        assert 250 < T[0] < 420 and 250 < T[1] < 420 and 0.2 < T[2] < 200, T
        it.step(50)
        x, v = ctx.getPositions(), ctx.getVelm()
        assert np.isfinite(x).all() and np.isfinite(v).all()
        assert np.array_equal(v[:, 3], v0[:, 3])            # inverse masses untouched
        d = spec.drude_pairs
        r = np.linalg.norm(x[d[:, 0]] - x[d[:, 1]], axis=1)
        assert r.max() < 0.02 * 1.5, r.max()                 # hard wall keeps Drudes near their parents
    finally:
        ctx.close()


@pytest.mark.parametrize("prec", ["mixed", "double"])      # single: see the module docstring -- at 1e5 particles the reference's serial float
@pytest.mark.parametrize("cfg,cos,hbonds", [("C3", 0.0, False), ("C3", 0.02, False), ("C3", 0.0, True), ("C5", 0.0, False), ("C5", 0.0, True),   # sums carry ~1e-4 noise
                                            ("C3-classic", 0.0, False), ("C3-classic", 0.02, True), ("C2", 0.0, True), ("C1", 0.0, False)])
def test_full_size_configs_against_the_oracle(cfg, cos, hbonds, prec):
    """BASELINE.json's configurations at their FULL size against the oracle (the C restatement needs ~0.1 s for these 10 steps at
    111 000 particles): positions and velocities within 1e-5 relative (measured ~1e-15), the group sums within 1e-10."""
    middle = "classic" not in cfg
    cfg = cfg.split("-")[0]
    spec = systems.make_config(cfg, hbonds=hbonds)          # C3 / C5: the reference's example models (tests/golden/topo_*.npz)
    kw = {}
    if cfg in ("C1", "C2"):
        kw = dict(maxd=0.0, T=300.0 if cfg == "C2" else 333.0, dt=0.002 if cfg == "C2" else 0.001)
    if cfg == "C5":
        lz = float(spec.box[2])
        kw = dict(mirror=lz / 2, efield=2.0 / lz * 2 * 1.602176634e-22)
    osys, ctx, it = _pair(spec, prec, middle, nsteps=10, cos=cos, **kw)
    try:
        ex, ev = _check(osys, ctx, prec, label=f"full {cfg} cos={cos} hbonds={hbonds} {prec}")
        print(f"full-size {cfg} cos={cos} hbonds={hbonds} {prec}: {spec.num_atoms} particles, rel err pos {ex:.2e} vel {ev:.2e}")
        st = ctx.getNHState()
        ntg = osys.s.num_tg
        assert np.allclose(np.array(list(st.ke2))[:ntg], osys.ke2()[:ntg], rtol=1e-10)
    finally:
        ctx.close()


def test_full_size_c3_single_precision():
    """Single precision at the FULL C3 size, GPU against the one-thread oracle in the same mode.  The element-wise stages are
    bit-identical; what differs is the order of the float sums behind the thermostat (111 000 terms: serial on the CPU, waves and
    blocks on the GPU), i.e. scale factors that differ in their last float bits.  That is enough to flip the rounding of a float
    POSITION now and then -- one ulp at x = 18 nm is 1.9e-6 nm -- and the Drude spring (209 200 kJ/mol/nm^2 on a 0.4 u particle)
    turns a one-ulp position difference into dv = k dx dt / m ~ 1e-3 nm/ps on that Drude particle within a step: 1e-4 of the
    largest velocity (measured: 1.1e-4 after 4 steps), in the oracle-vs-oracle comparison of two summation orders just the same.
    That noise is INTERNAL to a Drude pair (the spring force cancels in the pair's momentum), so the check is: positions to
    1e-6, velocities of all particles outside Drude pairs and the centre-of-mass velocity of every pair to 1e-5 (north_star), the
    group sums to 1e-4.  It is also why the reference's examples run mixed precision (examples/run-bulk.py:78)."""
    spec = systems.make_config("C3")
    nsteps = 4
    o_single, ctx, it = _pair(spec, "single", True, nsteps=nsteps)
    try:
        v_o, v_g = o_single.velm[:, :3].astype(np.float64), ctx.getVelocities()
        x_o, x_g = o_single.positions(), ctx.getPositions()
        scale = np.abs(v_o).max()
        ex = np.abs(x_g - x_o).max() / np.abs(x_o).max()
        d, p = spec.drude_pairs[:, 0], spec.drude_pairs[:, 1]
        free = np.ones(spec.num_atoms, dtype=bool)
        free[d] = False
        free[p] = False
        ev_free = np.abs(v_g[free] - v_o[free]).max() / scale
        m = spec.masses
        com = lambda v: (m[d, None] * v[d] + m[p, None] * v[p]) / (m[d] + m[p])[:, None]
        ev_com = np.abs(com(v_g) - com(v_o)).max() / scale
        ev_raw = np.abs(v_g - v_o).max() / scale
        ke_o, ke_g = o_single.ke2(), np.array(list(ctx.getNHState().ke2))
        print(f"single precision, full C3, {nsteps} steps: GPU vs oracle rel err pos {ex:.2e}; vel outside pairs {ev_free:.2e}, pair COM {ev_com:.2e}, "
              f"raw incl. Drude relative motion {ev_raw:.2e}; 2KE {ke_g} vs {ke_o}")
        assert ex < 1e-6 and ev_free < 1e-5 and ev_com < 1e-5
        assert ev_raw < 1e-3
        assert np.allclose(ke_g, ke_o, rtol=1e-4)
    finally:
        ctx.close()


def test_mixed_precision_stays_on_the_oracle_for_200_steps():
    """Drift check: 200 steps in mixed precision (fp64 velocities and sums) stay on the oracle's trajectory far inside the 1e-5 of
    north_star -- element-wise stages are bit-identical, only summation order and the last bits of exp / cos differ."""
    spec = systems.make_config("C3", 0.01)              # 30 ion pairs of bulk_Im21: 1 110 particles
    for cos in (0.0, 0.02):
        osys, ctx, it = _pair(spec, "mixed", True, nsteps=200, cos=cos)
        try:
            ex, ev = _check(osys, ctx, "mixed", tol=1e-9, label=f"200 steps cos={cos}")
            print(f"200 steps mixed cos={cos}: rel err pos {ex:.2e} vel {ev:.2e}")
        finally:
            ctx.close()
