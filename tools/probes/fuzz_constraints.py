"""Random constraint graphs (systems.add_random_constraints) on random small systems, GPU against the oracle and against the constraints themselves:
the loop of tests/test_gpu_general_constraints.py::test_random_constraint_graphs over many more seeds."""
import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import oracle as O
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0; capped = 0; worst_c = 0.0; kinds = {"general": 0, "clusters": 0}
for seed in range(7000, 7000 + n):
    rng = np.random.default_rng(seed)
    flavour, mirror = seed % 3, 0.0
    if flavour == 0: base = S.spce_water(int(rng.integers(5, 60)), seed=seed)
    elif flavour == 1: base = S.drude_il(cells=(1, 1, 1), pairs_per_cell=int(rng.integers(3, 24)), seed=seed)
    else:
        base = S.edl_slab(num_ion_pairs=int(rng.integers(3, 14)), num_electrode=int(rng.integers(4, 30)), seed=seed); mirror = float(base.box[2]) / 2
    spec = S.add_random_constraints(base, rng, max_degree=int(rng.integers(2, 9)))
    if len(spec.constraints) == 0: continue
    middle = bool(rng.integers(0, 2)); com = [None, True, False][int(rng.integers(0, 3))]
    prec = ["mixed", "double", "mixed"][int(rng.integers(0, 3))]
    maxd = 0.02 if len(spec.drude_pairs) else 0.0
    p = O.Params(temperature=300.0, drude_temperature=1.0, max_drude_distance=maxd, use_middle_scheme=middle, mirror_location=mirror)
    if com is not None: p.use_com_temp_group, p.auto_set_com_temp_group = com, False
    rnd = np.random.default_rng(seed + 1).standard_normal((4096, 4)).astype(np.float32)
    osys = O.OracleSystem(spec, p, prec, random=rnd, force_mode=1)
    kinds["general" if osys.general is not None else "clusters"] += 1
    it = I.VVIntegrator(300.0, 10.0, 1.0, 40.0, 0.001); it.setMaxDrudeDistance(maxd); it.setUseMiddleScheme(middle); it.setMirrorLocation(mirror)
    if com is not None: it.setUseCOMTempGroup(com)
    ctx = I.Context(spec, it, precision=prec, force_provider="tether", random=rnd)
    fused = ctx.info.constraints_fused
    if not fused:
        ctx.close(); print("not fused", seed, len(spec.constraints)); continue
    osys.step(8)
    try:
        it.step(8); ctx.synchronize()
    except pkg.vvhip.VVHipError as e:      # (since round 5 a cluster that stops at the iteration cap raises the sticky word [3]: the run is reported, not silently kept --
        if e.code != pkg.vvhip.ERR_CONSTRAINT: raise      # here the state is compared all the same: the oracle stops at the same cap)
        reported = globals().get("reported", 0) + 1
        ctx.status_clear()
    x_o, x_g = osys.positions(), ctx.getPositions()
    v_o, v_g = osys.velm[:, :3].astype(np.float64), ctx.getVelocities()
    ctx.close()
    massive = np.asarray(spec.masses) != 0
    ex = np.abs(x_g - x_o).max() / np.abs(x_o).max(); ev = np.abs(v_g[massive] - v_o[massive]).max() / np.abs(v_o[massive]).max()
    c, d = np.asarray(spec.constraints), np.asarray(spec.constraint_distances)
    r = x_g[c[:, 0]] - x_g[c[:, 1]]
    viol = np.abs((r * r).sum(1) - d * d).max() / (d * d).max()
    worst_c = max(worst_c, viol)
    if ex < 1e-9 and ev < 1e-9 and viol >= 2.5e-5:
        # GPU and oracle agree to rounding and both stopped at the cap of 150 sweeps: an ill-conditioned cluster (seen: rigid triangles with an
        # angle of 1-2 degrees, which random positions produce now and then) -- the method's limit, OpenMM's sweeps stop at theirs likewise
        print("unconverged at the sweep cap, GPU = oracle: seed", seed, "constraint", viol); capped += 1
    elif not (ex < 1e-5 and ev < 1e-4 and viol < 2.5e-5 and np.isfinite(x_g).all()):
        print("MISMATCH seed", seed, "flavour", flavour, prec, "middle", middle, "com", com, "pos", ex, "vel", ev, "constraint", viol); bad += 1
print("runs that reported an unconverged cluster (VVHIP_ERR_CONSTRAINT):", globals().get("reported", 0)); print("fuzz done:", n, "systems", kinds, "; mismatches", bad, "; stopped at the sweep cap (GPU = oracle)", capped, "; worst constraint violation %.1e" % worst_c)
