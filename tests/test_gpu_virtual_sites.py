"""-m gpu: virtual sites on the fused path (integration.computeVirtualSites() follows every position update of the
reference, CudaVVKernels.cpp:214, 374; the round-3 review lists them with the constraint topologies as work the fused step left to
OpenMM's own launches).  A site described to the plan (vvhip_system_desc.virtual_sites) is tied to the wave of its parents and
kernel B places it after the hard wall and before the image mirror (stage bit B_VSITE, vv_device.inc: place_virtual_site) -- from the
lane of one of its parents as a rule, so that sites cost no lanes; from a lane of its own where no parent is free or the site has an image.
Checked against the oracle's statement (oracle/vv_oracle.c: vvo_compute_virtual_sites) and, independently of it, against the documented
definitions of OpenMM's four site classes evaluated in float64 on the final parent positions.  Parity with OpenMM's own kernel is
unpinned (its source is not under /root/reference), as for every OpenMM service on this path."""
import importlib

import numpy as np
import pytest

from oracle import oracle as O

pkg = importlib.import_module("openmm-velocityverlet_amd")
H, I, systems = pkg.vvhip, pkg.integrator, pkg.systems
pytestmark = pytest.mark.gpu


def _run(spec, prec, middle, nsteps, graph=False, com=None, cos=0.0):
    p = O.Params(temperature=300.0, drude_temperature=1.0, max_drude_distance=0.02, use_middle_scheme=middle, cos_acceleration=cos)
    if com is not None:
        p.use_com_temp_group, p.auto_set_com_temp_group = com, False
    osys = O.OracleSystem(spec, p, prec, force_mode=1)
    it = I.VVIntegrator(300.0, 10.0, 1.0, 40.0, 0.001)
    it.setMaxDrudeDistance(0.02)
    it.setUseMiddleScheme(middle)
    it.setCosAcceleration(cos)
    if com is not None:
        it.setUseCOMTempGroup(com)
    ctx = I.Context(spec, it, precision=prec, force_provider="tether")
    osys.step(nsteps)
    if graph:
        ctx.run_graph(nsteps, nsteps)
    else:
        it.step(nsteps)
    return osys, ctx


def _check(spec, osys, ctx, prec, label, tol=None):
    x_o, x_g = osys.positions(), ctx.getPositions()
    v_o, v_g = osys.velm[:, :3].astype(np.float64), ctx.getVelocities()
    assert np.isfinite(x_g).all() and np.isfinite(v_g).all(), label
    tol = tol if tol is not None else (1e-5 if prec == "single" else 1e-12)
    ex = np.abs(x_g - x_o).max() / np.abs(x_o).max()
    massive = np.asarray(spec.masses) != 0
    ev = np.abs(v_g[massive] - v_o[massive]).max() / np.abs(v_o[massive]).max()
    assert ex < tol and ev < tol, f"{label}: rel err pos {ex:.2e} vel {ev:.2e}"
    # the definitions themselves, float64, on the parents the GPU stored
    worst = 0.0
    for site, kind, parents, prm in spec.virtual_sites:
        want = systems.virtual_site_position(kind, np.asarray(prm, dtype=np.float32 if prec != "double" else np.float64).astype(np.float64),
                                             *[x_g[i] for i in parents])
        worst = max(worst, np.abs(x_g[site] - want).max())
    scale = np.abs(x_g).max()
    assert worst < (2e-6 if prec == "single" else 1e-12) * scale, f"{label}: site off its definition by {worst:.2e} nm"
    print(f"{label}: rel err pos {ex:.2e} vel {ev:.2e}, sites within {worst:.1e} nm of their definition")


@pytest.mark.parametrize("prec", O.PRECISIONS)
@pytest.mark.parametrize("middle", [True, False])
@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_water_with_one_site_per_molecule(kind, middle, prec):
    """O H H M: every kind of site on an unconstrained three-site molecule, both schemes, three precisions."""
    spec = systems.add_virtual_sites(systems.spce_water(60, seed=5), kinds=(kind,))
    osys, ctx = _run(spec, prec, middle, 10)
    try:
        assert ctx.info.num_virtual_sites == 60 == len(spec.virtual_sites)
        assert list(ctx.info.dof)[0] == list(osys.t["dof"])[0] == 3 * 180 - 3          # a massless site adds no degree of freedom (HOST:496-503)
        _check(spec, osys, ctx, prec, f"water/kind {kind}/{prec}/middle={middle}")
        assert tuple(ctx.generic_launches()[0]) == (0, 0)                # the stage set with B_VSITE is compiled at run time (csrc/vv_rtc.cpp)
    finally:
        ctx.close()


@pytest.mark.parametrize("prec", O.PRECISIONS)
@pytest.mark.parametrize("middle", [True, False])
def test_more_sites_than_parents(middle, prec):
    """Four sites on a three-site molecule: three are placed from their parents' lanes (a site costs no lane as a rule), the fourth has no
    parent left that places nothing yet and gets a lane of its own -- both ways of placing in one wave."""
    spec = systems.add_virtual_sites(systems.spce_water(60, seed=5), kinds=(1, 3, 0, 2))
    osys, ctx = _run(spec, prec, middle, 10)
    try:
        assert ctx.info.num_virtual_sites == 240 and ctx.info.num_slots_used == 180 + 60
        _check(spec, osys, ctx, prec, f"crowded/{prec}/middle={middle}")
    finally:
        ctx.close()


@pytest.mark.parametrize("com", [True, False])
@pytest.mark.parametrize("middle", [True, False])
def test_drude_liquid_with_lone_pairs_appended(middle, com):
    """A polarisable liquid with a lone pair (local coordinates, examples/ommhelper/oplspsffile.py:982-991) and a two-particle average per
    molecule, stored behind the last real particle: Drude pairs, hard wall and the molecular temperature group share the wave with them."""
    spec = systems.add_virtual_sites(systems.drude_il(cells=(1, 1, 1), pairs_per_cell=16, seed=4), kinds=(3, 0), interleaved=False)
    osys, ctx = _run(spec, "mixed", middle, 12, com=com)
    try:
        assert ctx.info.num_virtual_sites == len(spec.virtual_sites) == 64
        _check(spec, osys, ctx, "mixed", f"lone pairs/com={com}/middle={middle}")
    finally:
        ctx.close()


@pytest.mark.parametrize("prec", ["mixed", "double"])
@pytest.mark.parametrize("base", ["settle", "hbonds", "allbonds"])
def test_sites_on_constrained_molecules(base, prec):
    """TIP4P's shape -- a rigid triangle (SETTLE) with an averaged site -- and sites on molecules with hydrogen-type and general clusters:
    the site hangs on the CONSTRAINED positions (HOST:176 before :214)."""
    if base == "settle":
        spec = systems.add_virtual_sites(systems.rigid_water(systems.spce_water(60, seed=6)), kinds=(1,))
    elif base == "hbonds":
        spec = systems.add_virtual_sites(systems.bulk_Im21(cells=(1, 1, 1), pairs_per_cell=10, hbonds=True), kinds=(2,))
    else:
        spec = systems.add_virtual_sites(systems.constrain_all_bonds(systems.bulk_Im21(cells=(1, 1, 1), pairs_per_cell=10)), kinds=(3,))
    osys, ctx = _run(spec, prec, True, 10)
    try:
        assert ctx.info.constraints_fused == 1 and ctx.info.num_virtual_sites == len(spec.virtual_sites) > 0
        _check(spec, osys, ctx, prec, f"{base}/{prec}", tol=1e-5 if base == "allbonds" else 1e-9)
        c, d = np.asarray(spec.constraints), np.asarray(spec.constraint_distances)
        r = ctx.getPositions()[c[:, 0]] - ctx.getPositions()[c[:, 1]]
        assert np.abs((r * r).sum(1) - d * d).max() < 3e-5 * (d * d).max()
    finally:
        ctx.close()


def test_graph_replay_places_the_sites_like_eager_steps():
    spec = systems.add_virtual_sites(systems.drude_il(cells=(1, 1, 1), pairs_per_cell=16, seed=4), kinds=(3,))
    out = []
    for graph in (False, True):
        osys, ctx = _run(spec, "mixed", True, 16, graph=graph, cos=0.02)
        try:
            _check(spec, osys, ctx, "mixed", f"graph={graph}")
            out.append((ctx.getPosq().copy(), ctx.getVelm().copy()))
        finally:
            ctx.close()
    assert np.array_equal(out[0][0].view(np.uint8), out[1][0].view(np.uint8)) and np.array_equal(out[0][1].view(np.uint8), out[1][1].view(np.uint8))


def test_images_mirror_the_sites_new_position():
    """run-edl.py's machinery with a site among the parents of the image particles: the reference places the site inside the integrate
    kernel (HOST:214) and mirrors afterwards (API:266-268), so the image of a site is the mirror of where the site has just been put."""
    base = systems.edl_slab(num_ion_pairs=12, num_electrode=24)
    spec = systems.add_virtual_sites(base, kinds=(1,), interleaved=False)
    # give the first few sites an image each: massless, in the site's molecule
    n0 = spec.num_atoms
    extra = [s[0] for s in spec.virtual_sites[:6]]
    k = len(extra)
    spec.masses = np.concatenate([spec.masses, np.zeros(k)]); spec.charges = np.concatenate([spec.charges, np.full(k, 0.3)])
    pos = spec.positions[extra].copy(); pos[:, 2] = float(spec.box[2]) - pos[:, 2]
    spec.positions = np.concatenate([spec.positions, pos]); spec.velocities = np.concatenate([spec.velocities, np.zeros((k, 3))])
    spec.mol_id = np.concatenate([spec.mol_id, spec.mol_id[extra]]).astype(np.int32)
    spec.image_pairs = list(spec.image_pairs) + [(n0 + j, extra[j]) for j in range(k)]
    lz = float(spec.box[2])
    p = O.Params(temperature=333.0, drude_temperature=1.0, max_drude_distance=0.02, mirror_location=lz / 2)
    rnd = np.random.default_rng(3).standard_normal((4096, 4)).astype(np.float32)
    osys = O.OracleSystem(spec, p, "mixed", random=rnd, force_mode=1)
    it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
    it.setMaxDrudeDistance(0.02)
    it.setMirrorLocation(lz / 2)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether", random=rnd)
    try:
        assert ctx.info.num_virtual_sites == len(spec.virtual_sites)
        osys.step(6)
        it.step(6)
        _check(spec, osys, ctx, "mixed", "edl + sites + images of sites")
        x = ctx.getPositions()
        for j in range(k):
            assert abs((lz - x[extra[j], 2]) - x[n0 + j, 2]) < 1e-6 and np.allclose(x[extra[j], :2], x[n0 + j, :2], atol=1e-7)
    finally:
        ctx.close()


def test_full_size_c3_with_a_lone_pair_on_every_molecule():
    """BASELINE.json C3 (111 000 particles) + one local-coordinates site per molecule (6 000 sites), size-independent properties only: every
    site sits where its definition puts it on the positions the GPU stored, the real particles move as they do without the sites
    (a massless site takes part in nothing: HOST:496-503, K/middle.cu:11), and the step still takes two launches."""
    base = systems.make_config("C3")
    spec = systems.add_virtual_sites(base, kinds=(3,), interleaved=False)
    out = []
    for s in (spec, base):
        it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
        it.setMaxDrudeDistance(0.02)
        ctx = I.Context(s, it, precision="mixed", force_provider="tether")
        try:
            if s is spec:
                assert ctx.info.num_virtual_sites == 6000 == len(spec.virtual_sites)
            ctx.run_graph(64, steps_per_graph=8)
            out.append((ctx.getPositions(), ctx.getVelm().copy(), ctx.getPosq().copy()))
        finally:
            ctx.close()
    n = base.num_atoms
    # (to rounding: the sites' lanes shift the real particles into other waves and blocks, and a block's kinetic-energy sum is formed in
    # double before it enters the exact fixed-point accumulators -- another order of the same additions)
    dv = np.abs(out[0][1][:n, :3] - out[1][1][:, :3]).max() / np.abs(out[1][1][:, :3]).max()
    dx = np.abs(out[0][0][:n] - out[1][0]).max() / np.abs(out[0][0]).max()
    assert dv < 1e-11 and dx < 1e-11, (dv, dx)
    x = out[0][0]
    sites = np.array([s[0] for s in spec.virtual_sites])
    par = np.array([s[2] for s in spec.virtual_sites])
    w = np.asarray(spec.virtual_sites[0][3], dtype=np.float32).astype(np.float64)
    p1, p2, p3 = x[par[:, 0]], x[par[:, 1]], x[par[:, 2]]
    o = p1 * w[0] + p2 * w[1] + p3 * w[2]
    xd = p1 * w[3] + p2 * w[4] + p3 * w[5]
    yd = p1 * w[6] + p2 * w[7] + p3 * w[8]
    zd = np.cross(xd, yd)
    xd /= np.linalg.norm(xd, axis=1)[:, None]
    zd /= np.linalg.norm(zd, axis=1)[:, None]
    yd = np.cross(zd, xd)
    want = o + xd * w[9] + yd * w[10] + zd * w[11]
    assert np.abs(x[sites] - want).max() < 1e-12 * np.abs(x).max()


@pytest.mark.parametrize("seed", range(3000, 3024))
def test_random_sites_on_random_small_systems(seed):
    """tools/probes/fuzz_sites.py in small: any massive particles of a molecule as parents, several sites per molecule and per parent (placed
    from a parent's lane or from a lane of their own), random kinds and weights, both schemes, with / without the molecular temperature
    group, on plain, hydrogen-constrained, rigid and all-bonds-constrained molecules and on the electrode slab."""
    rng = np.random.default_rng(seed)
    flavour, mirror = seed % 6, 0.0
    if flavour == 0: base = systems.spce_water(int(rng.integers(5, 60)), seed=seed)
    elif flavour == 1: base = systems.rigid_water(systems.spce_water(int(rng.integers(5, 60)), seed=seed))
    elif flavour == 2: base = systems.drude_il(cells=(1, 1, 1), pairs_per_cell=int(rng.integers(3, 30)), seed=seed)
    elif flavour == 3: base = systems.constrain_hydrogens(systems.drude_il(cells=(1, 1, 1), pairs_per_cell=int(rng.integers(3, 30)), seed=seed))
    elif flavour == 4: base = systems.constrain_all_bonds(systems.bulk_Im21(cells=(1, 1, 1), pairs_per_cell=int(rng.integers(3, 16))))
    else:
        base = systems.edl_slab(num_ion_pairs=int(rng.integers(3, 14)), num_electrode=int(rng.integers(4, 30)), seed=seed)
        mirror = float(base.box[2]) / 2
    spec = systems.add_random_virtual_sites(base, rng)
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
        assert ctx.info.num_virtual_sites == len(spec.virtual_sites) > 0
        osys.step(6)
        it.step(6)
        _check(spec, osys, ctx, "mixed", f"random sites seed {seed} flavour {flavour}", tol=1e-5 if flavour == 4 else 1e-10)
    finally:
        ctx.close()


def test_sites_that_cannot_be_placed_are_refused_by_this_host():
    """A site hanging on another site is left to the caller's own kernel (vvhip_plan_info.num_virtual_sites == 0); the OpenMM adapters
    then keep calling computeVirtualSites, the stand-alone Python host has nothing of the kind and says so instead of leaving sites stale."""
    spec = systems.add_virtual_sites(systems.spce_water(8), kinds=(1,))
    spec.virtual_sites[1] = (spec.virtual_sites[1][0], 0, (spec.virtual_sites[0][0], 4), (0.5, 0.5))
    it = I.VVIntegrator(300.0, 10.0, 1.0, 40.0, 0.001)
    with pytest.raises(H.VVHipError, match="virtual sites cannot be placed"):
        I.Context(spec, it, precision="mixed", force_provider="tether")


def test_sites_follow_the_position_update_of_the_split_entry_points_too():
    """The per-KernelImpl entry points (vvhip_middle_finish behind OpenMM's constraint solver; run_eager_unfused drives them in the reference's
    order) place the plan's sites with the same stage as the fused step (round-4 advisor: they used to leave the sites where they were)."""
    spec = systems.add_virtual_sites(systems.spce_water(40, seed=3), kinds=(1,))
    outs = []
    for split in (False, True):
        it = I.VVIntegrator(300.0, 10.0, 1.0, 40.0, 0.001)
        ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
        try:
            assert ctx.info.num_virtual_sites == len(spec.virtual_sites) > 0
            (ctx.run_eager_unfused if split else ctx.run_eager)(5)
            ctx.synchronize()
            outs.append(ctx.getPositions())
        finally:
            ctx.close()
    sites = [s[0] for s in spec.virtual_sites]
    assert np.abs(outs[0] - outs[1]).max() < 1e-12                      # the two paths agree (sums in another order: rounding level)
    x = outs[1]
    for site, kind, parents, prm in spec.virtual_sites[:8]:            # ... and the sites sit where their definition puts them
        want = systems.virtual_site_position(kind, prm, *[x[q] for q in parents])
        assert np.abs(x[site] - want).max() < 1e-6
