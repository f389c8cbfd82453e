"""Rigid three-site molecules of RANDOM shape and masses (apex-partner distance, partner-partner distance, light or heavy apex), GPU against the oracle's
independently written SETTLE and against the constraints themselves (bond lengths, momentum): what tests/test_gpu_constraints.py::test_rigid_water_settle
checks for SPC/E water, over many geometries -- written for round 6's rewrite of the rigid-triangle arithmetic.   python tools/probes/fuzz_settle.py [cases]"""
import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import oracle as O
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = dict(pos=0.0, vel=0.0, bond=0.0, mom=0.0); bad = 0
for seed in range(9000, 9000 + n):
    rng = np.random.default_rng(seed)
    spec = S.spce_water(int(rng.integers(20, 200)), seed=seed)
    m_apex, m_part = float(rng.uniform(1.0, 40.0)), float(rng.uniform(1.0, 40.0))
    spec.masses = np.tile([m_apex, m_part, m_part], spec.num_atoms // 3)
    d_ab = float(rng.uniform(0.08, 0.16))
    d_bb = float(rng.uniform(0.35, 1.9)) * d_ab          # apex angle from ~20 to ~145 degrees
    spec = S.rigid_water(spec, d_oh=d_ab, d_hh=d_bb)
    middle = bool(rng.integers(0, 2)); prec = ["mixed", "double"][int(rng.integers(0, 2))]
    dt = float(rng.choice([0.001, 0.002, 0.004]))
    p = O.Params(temperature=300.0, drude_temperature=1.0, step_size=dt, max_drude_distance=0.0, use_middle_scheme=middle)
    osys = O.OracleSystem(spec, p, prec, force_mode=1)
    it = I.VVIntegrator(300.0, 10.0, 1.0, 40.0, dt); it.setMaxDrudeDistance(0.0); it.setUseMiddleScheme(middle)
    ctx = I.Context(spec, it, precision=prec, force_provider="tether")
    assert ctx.info.constraints_fused and ctx.info.num_settle_clusters == spec.num_atoms // 3
    m = np.asarray(spec.masses)
    osys.step(10); it.step(10)
    x_o, x_g = osys.positions(), ctx.getPositions()
    v_o, v_g = osys.velm[:, :3].astype(np.float64), ctx.getVelocities()
    ctx.close()
    ex = np.abs(x_g - x_o).max() / np.abs(x_o).max(); ev = np.abs(v_g - v_o).max() / np.abs(v_o).max()
    c, d = np.asarray(spec.constraints), np.asarray(spec.constraint_distances).astype(np.float32).astype(np.float64)
    eb = np.abs(np.linalg.norm(x_g[c[:, 0]] - x_g[c[:, 1]], axis=1) - d).max()
    worst["pos"] = max(worst["pos"], ex); worst["vel"] = max(worst["vel"], ev); worst["bond"] = max(worst["bond"], eb)
    ok = ex < 1e-5 and ev < 1e-5 and eb < 1e-12 and np.isfinite(x_g).all()
    if not ok:
        bad += 1
        print(f"seed {seed}: m {m_apex:.2f}/{m_part:.2f} d {d_ab:.3f}/{d_bb:.3f} {prec} middle={middle} dt={dt}: pos {ex:.2e} vel {ev:.2e} bond {eb:.2e}", flush=True)
print(f"{n} cases, {bad} outside (1e-5 against the oracle, 1e-12 nm on the bonds); worst: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items() if k != 'mom'))
