"""-m gpu: constraint topologies BEYOND the examples' defaults on the fused path (round-3 review "missing" item 5):
constraints=AllBonds / HAngles (examples/ommhelper/oplspsffile.py:948-951) give rings, chains and triangles of constraints, which OpenMM
relaxes with CCMA between the reference's launches (CudaVVKernels.cpp:151,176,351,427).  Here every connected component of the constraint
graph sits in one wave with its molecule and the wave relaxes its constraints by coloured Gauss-Seidel sweeps inside kernel A
(velocities) and kernel B (positions): stage bits A_GCONS / B_GCONS, vv_device.inc: general_velocities / general_positions.
Checked against the oracle's statement of the same sweeps (oracle/vv_oracle.c: vvo_general_*) and through the constraint invariants,
which do not depend on the oracle.  Parity with OpenMM's own solver is unpinned, as for every constraint on this path."""
import importlib

import numpy as np
import pytest

from oracle import oracle as O

pkg = importlib.import_module("openmm-velocityverlet_amd")
H, I, systems = pkg.vvhip, pkg.integrator, pkg.systems
pytestmark = pytest.mark.gpu
TOL = 1e-5


def _system(kind, pairs=12):
    spec = systems.bulk_Im21(cells=(1, 1, 1), pairs_per_cell=pairs)
    return systems.constrain_all_bonds(spec, hangles=(kind == "hangles"))


def _run(spec, prec, middle, nsteps, cos=0.0, graph=False):
    p = O.Params(temperature=333.0, drude_temperature=1.0, max_drude_distance=0.02, use_middle_scheme=middle, cos_acceleration=cos)
    osys = O.OracleSystem(spec, p, prec, force_mode=1)
    assert osys.general is not None and osys.clusters is None
    it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
    it.setMaxDrudeDistance(0.02)
    it.setUseMiddleScheme(middle)
    it.setCosAcceleration(cos)
    ctx = I.Context(spec, it, precision=prec, force_provider="tether")
    osys.step(nsteps)
    if graph:
        ctx.run_graph(nsteps, nsteps)
    else:
        it.step(nsteps)
    return osys, ctx


def _check(spec, osys, ctx, prec, middle, label):
    x_o, x_g = osys.positions(), ctx.getPositions()
    v_o, v_g = osys.velm[:, :3].astype(np.float64), ctx.getVelocities()
    assert np.isfinite(x_g).all() and np.isfinite(v_g).all(), label
    ex = np.abs(x_g - x_o).max() / np.abs(x_o).max()
    ev = np.abs(v_g - v_o).max() / np.abs(v_o).max()
    # same sweeps in the same order on both sides; a convergence test that falls the other way within rounding costs one update of tolerance size
    assert ex < 1e-5 and ev < (1e-4 if prec == "single" else 1e-5), f"{label}: rel err pos {ex:.2e} vel {ev:.2e}"
    c, d = np.asarray(spec.constraints), np.asarray(spec.constraint_distances)
    r = x_g[c[:, 0]] - x_g[c[:, 1]]
    r2 = (r * r).sum(1)
    slack = 2.0 * TOL + (4e-5 if prec == "single" else 1e-9)
    assert np.abs(r2 - d * d).max() < slack * (d * d).max(), f"{label}: |r^2 - d^2| / d^2 = {np.abs(r2 - d * d).max() / (d * d).max():.2e}"
    if not middle:
        # The classic scheme's step ends: half kick -> velocity constraints -> thermostat (API:327-336).  The thermostat scales a Drude pair's
        # centre-of-mass and relative motion with different factors, so a ring bond between a polarisable atom and its neighbour picks up
        # ~3e-3 nm/ps of bond-parallel relative velocity again (1e-2 in single precision) -- the reference's order of operations, and the
        # oracle's, held by the parity assertion above; thermal velocities are ~1 nm/ps.
        rel = ((v_g[c[:, 0]] - v_g[c[:, 1]]) * r).sum(1) / np.sqrt(r2)
        assert np.abs(rel).max() < 5e-2, f"{label}: bond-parallel relative velocity {np.abs(rel).max():.2e} nm/ps"
    print(f"{label}: rel err pos {ex:.2e} vel {ev:.2e}, constraints within {np.abs(r2 - d * d).max() / (d * d).max():.1e}")


@pytest.mark.parametrize("prec", O.PRECISIONS)
@pytest.mark.parametrize("middle", [True, False])
@pytest.mark.parametrize("kind", ["allbonds", "hangles"])
def test_ionic_liquid_with_all_bonds_constrained(kind, middle, prec):
    spec = _system(kind)
    osys, ctx = _run(spec, prec, middle, 2 if prec == "single" else 20)
    try:
        info = ctx.info
        assert info.constraints_fused == 1 and info.num_shake_clusters == 0 and info.num_settle_clusters == 0
        assert info.num_general_constraints == len(spec.constraints) > 0
        assert list(info.dof)[0] == list(osys.t["dof"])[0]                      # constraints leave the atom group (HOST:505-509), whatever solves them
        _check(spec, osys, ctx, prec, middle, f"{kind}/{prec}/middle={middle}")
        assert tuple(ctx.generic_launches()[0]) == (0, 0)                        # stage sets outside the compiled list are compiled at run time
    finally:
        ctx.close()


def test_rings_hold_over_a_long_run_and_under_graph_replay():
    """2 000 steps from replayed graphs: every ring and chain constraint within tolerance at the end, Drude separations sane, eager and
    replayed runs bit-identical (the sweeps are deterministic: LDS operations of a wave execute in order)."""
    spec = _system("hangles", pairs=24)
    it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
    it.setMaxDrudeDistance(0.02)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    it2 = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
    it2.setMaxDrudeDistance(0.02)
    ctx2 = I.Context(spec, it2, precision="mixed", force_provider="tether")
    try:
        ctx.run_graph(40, 20)
        it2.step(40)
        assert np.array_equal(ctx.getPosq().view(np.uint8), ctx2.getPosq().view(np.uint8)) and np.array_equal(ctx.getVelm().view(np.uint8), ctx2.getVelm().view(np.uint8))
        ctx.run_graph(1960, 40)
        x = ctx.getPositions()
        c, d = np.asarray(spec.constraints), np.asarray(spec.constraint_distances)
        r = np.linalg.norm(x[c[:, 0]] - x[c[:, 1]], axis=1)
        assert np.isfinite(x).all() and np.abs(r - d).max() / d.max() < 2e-5
        pr = np.asarray(spec.drude_pairs)
        assert np.linalg.norm(x[pr[:, 0]] - x[pr[:, 1]], axis=1).max() < 0.02
        assert ctx.status() == (False, False)
    finally:
        ctx.close(); ctx2.close()


def test_general_constraints_with_the_cos_perturbation_and_without_com_group():
    """The other stage combinations a general-constraint System can ask for: cos acceleration (moment form), and a plain (non-COM) thermostat,
    where the wave layout is built from the closure of Drude pairs and constraint partners instead of whole molecules."""
    spec = _system("allbonds")
    osys, ctx = _run(spec, "mixed", True, 12, cos=0.02)
    try:
        _check(spec, osys, ctx, "mixed", True, "allbonds + cos")
    finally:
        ctx.close()
    spec = _system("allbonds")
    p = O.Params(temperature=333.0, drude_temperature=1.0, max_drude_distance=0.02, use_com_temp_group=False, auto_set_com_temp_group=False)
    osys = O.OracleSystem(spec, p, "mixed", force_mode=1)
    it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
    it.setMaxDrudeDistance(0.02)
    it.setUseCOMTempGroup(False)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    try:
        assert ctx.info.num_general_constraints == len(spec.constraints)
        osys.step(12); it.step(12)
        _check(spec, osys, ctx, "mixed", True, "allbonds, no COM group")
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", range(4000, 4024))
def test_random_constraint_graphs(seed):
    """Random forests and rings of constraints inside the molecules (systems.add_random_constraints: chains, stars up to degree 6, rings,
    triangles, lengths of every size the molecule offers), on water, on the polarisable liquid and on the electrode slab, both schemes, with
    and without the molecular temperature group: the in-wave sweeps against the oracle's, and the constraints themselves."""
    rng = np.random.default_rng(seed)
    flavour, mirror = seed % 3, 0.0
    if flavour == 0: base = systems.spce_water(int(rng.integers(5, 60)), seed=seed)
    elif flavour == 1: base = systems.drude_il(cells=(1, 1, 1), pairs_per_cell=int(rng.integers(3, 24)), seed=seed)
    else:
        base = systems.edl_slab(num_ion_pairs=int(rng.integers(3, 14)), num_electrode=int(rng.integers(4, 30)), seed=seed)
        mirror = float(base.box[2]) / 2
    spec = systems.add_random_constraints(base, rng)
    middle = bool(rng.integers(0, 2))
    com = [None, True, False][int(rng.integers(0, 3))]
    maxd = 0.02 if len(spec.drude_pairs) else 0.0
    p = O.Params(temperature=300.0, drude_temperature=1.0, max_drude_distance=maxd, use_middle_scheme=middle, mirror_location=mirror)
    if com is not None:
        p.use_com_temp_group, p.auto_set_com_temp_group = com, False
    rnd = np.random.default_rng(seed + 1).standard_normal((4096, 4)).astype(np.float32)
    osys = O.OracleSystem(spec, p, "mixed", random=rnd, force_mode=1)
    it = I.VVIntegrator(300.0, 10.0, 1.0, 40.0, 0.001)
    it.setMaxDrudeDistance(maxd)
    it.setUseMiddleScheme(middle)
    it.setMirrorLocation(mirror)
    if com is not None:
        it.setUseCOMTempGroup(com)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether", random=rnd)
    try:
        info = ctx.info
        assert info.constraints_fused == 1
        if osys.general is not None:
            assert info.num_general_constraints == len(spec.constraints) > 0
        osys.step(8)
        # (a random topology now and then holds a cluster that the sweeps leave at their cap of 150 rounds -- on the GPU and in the oracle alike,
        # seed 4007 of this range -- still inside the tolerance asserted below: since round 5 the kernels SAY so, tests/test_gpu_status.py)
        try:
            it.step(8)
            ctx.synchronize()
        except H.VVHipError as e:
            assert e.code == H.ERR_CONSTRAINT, e
            assert ctx.status_words()[3] == 1
            ctx.status_clear()
        x_o, x_g = osys.positions(), ctx.getPositions()
        v_o, v_g = osys.velm[:, :3].astype(np.float64), ctx.getVelocities()
        massive = np.asarray(spec.masses) != 0
        ex = np.abs(x_g - x_o).max() / np.abs(x_o).max()
        ev = np.abs(v_g[massive] - v_o[massive]).max() / np.abs(v_o[massive]).max()
        assert ex < 1e-5 and ev < 1e-4, (seed, ex, ev)
        c, d = np.asarray(spec.constraints), np.asarray(spec.constraint_distances)
        r = x_g[c[:, 0]] - x_g[c[:, 1]]
        assert np.abs((r * r).sum(1) - d * d).max() < 2.5e-5 * (d * d).max(), seed
    finally:
        ctx.close()
