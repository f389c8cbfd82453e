"""Random virtual sites on random small systems, GPU against the oracle and against the sites' definitions: any massive particles of a molecule as
parents (Drude particles included), several sites per molecule and per parent (so that some are placed from a parent's lane and some from a lane of
their own), all four kinds with random weights, both schemes, with and without the molecular temperature group, on plain / hydrogen-constrained /
rigid / all-bonds-constrained / randomly constrained molecules and on the electrode slab (Langevin wall, images)."""
import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import oracle as O
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems


with_random_sites = S.add_random_virtual_sites


bad = 0
worst = 0.0
hosted = own = 0
for seed in range(3000, 3000 + int(sys.argv[1]) if len(sys.argv) > 1 else 3060):
    rng = np.random.default_rng(seed)
    flavour = seed % 7
    mirror = 0.0
    if flavour == 0: base = S.spce_water(int(rng.integers(5, 60)), seed=seed)
    elif flavour == 1: base = S.rigid_water(S.spce_water(int(rng.integers(5, 60)), seed=seed))
    elif flavour == 2: base = S.drude_il(cells=(1, 1, 1), pairs_per_cell=int(rng.integers(3, 30)), seed=seed)
    elif flavour == 3: base = S.constrain_hydrogens(S.drude_il(cells=(1, 1, 1), pairs_per_cell=int(rng.integers(3, 30)), seed=seed))
    elif flavour == 4: base = S.constrain_all_bonds(S.bulk_Im21(cells=(1, 1, 1), pairs_per_cell=int(rng.integers(3, 16))))
    elif flavour == 6: base = S.add_random_constraints(S.drude_il(cells=(1, 1, 1), pairs_per_cell=int(rng.integers(3, 20)), seed=seed), rng)
    else:
        base = S.edl_slab(num_ion_pairs=int(rng.integers(3, 14)), num_electrode=int(rng.integers(4, 30)), seed=seed); mirror = float(base.box[2]) / 2
    spec = with_random_sites(base, rng)
    if not spec.virtual_sites:
        continue
    middle = bool(rng.integers(0, 2))
    com = [None, True, False][int(rng.integers(0, 3))]
    maxd = 0.02 if len(spec.drude_pairs) else 0.0
    p = O.Params(temperature=300.0, drude_temperature=1.0, max_drude_distance=maxd, use_middle_scheme=middle, mirror_location=mirror)
    if com is not None: p.use_com_temp_group, p.auto_set_com_temp_group = com, False
    rnd = np.random.default_rng(seed + 1).standard_normal((4096, 4)).astype(np.float32)
    osys = O.OracleSystem(spec, p, "mixed", random=rnd, force_mode=1)
    it = I.VVIntegrator(300.0, 10.0, 1.0, 40.0, 0.001)
    it.setMaxDrudeDistance(maxd); it.setUseMiddleScheme(middle); it.setMirrorLocation(mirror)
    if com is not None: it.setUseCOMTempGroup(com)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether", random=rnd)
    nsites = ctx.info.num_virtual_sites
    lanes0 = I.plan_layout(base, it)[0].num_slots_used
    own += ctx.info.num_slots_used - lanes0; hosted += nsites - (ctx.info.num_slots_used - lanes0)
    osys.step(6)
    try:
        it.step(6); ctx.synchronize()
    except pkg.vvhip.VVHipError as e:      # (since round 5 a cluster that stops at the iteration cap raises the sticky word [3]: the run is reported, not silently kept --
        if e.code != pkg.vvhip.ERR_CONSTRAINT: raise      # here the state is compared all the same: the oracle stops at the same cap)
        reported = globals().get("reported", 0) + 1
        ctx.status_clear()
    x_o, x_g = osys.positions(), ctx.getPositions()
    ctx.close()
    if nsites != len(spec.virtual_sites):
        print("NOT PLACED IN-KERNEL", seed, flavour, nsites, len(spec.virtual_sites)); bad += 1; continue
    tol = 1e-5 if flavour in (4, 6) else 1e-10
    ex = np.abs(x_g - x_o).max() / np.abs(x_o).max()
    dev = 0.0
    for site, kind, par, prm in spec.virtual_sites:
        want = S.virtual_site_position(kind, np.asarray(prm, np.float32).astype(np.float64), *[x_g[i] for i in par])
        dev = max(dev, np.abs(x_g[site] - want).max())
    worst = max(worst, dev)
    if not (ex < tol and dev < 1e-11 * max(1.0, np.abs(x_g).max())):
        print("MISMATCH seed", seed, "flavour", flavour, "middle", middle, "com", com, "rel err", ex, "site off definition", dev); bad += 1
print("fuzz done: sites placed from a parent's lane", hosted, ", from a lane of their own", own, "; mismatches", bad, "; worst distance from the definition %.1e nm" % worst)
