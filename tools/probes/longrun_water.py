"""Long-run sanity of the rigid SPC/E box (SETTLE in the one-launch step, round 6's arithmetic): temperature, bond lengths and total momentum over 400 000 steps."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
spec = S.make_config("C2", hbonds=True)
it = I.VVIntegrator(300.0, 10, 1.0, 40, 0.002); it.setMaxDrudeDistance(0.0)
ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
m = np.asarray(spec.masses)
c, dist = np.asarray(spec.constraints), np.asarray(spec.constraint_distances).astype(np.float32).astype(np.float64)
t0 = time.perf_counter()
for k in range(8):
    ctx.run_graph(50000, 100); ctx.synchronize()
    T = ctx.getGroupTemperatures(); x = ctx.getPositions(); v = ctx.getVelocities()
    rr = np.linalg.norm(x[c[:, 0]] - x[c[:, 1]], axis=1)
    rel = np.abs(((v[c[:, 0]] - v[c[:, 1]]) * (x[c[:, 0]] - x[c[:, 1]])).sum(1)).max()
    print(f"  step {(k + 1) * 50000:7d}: T {T[0]:7.2f} K   max |r - d| {np.abs(rr - dist).max():.1e} nm   max |v_ij . r_ij| {rel:.1e}   one launch {ctx.fused_status()[0]}   finite {np.isfinite(x).all()}", flush=True)
print(f"  {400000 / (time.perf_counter() - t0):.0f} steps/s sustained, status words {ctx.status_words()}")
ctx.close()
