"""Fixed cost of a timed region as bench.py brackets it: region(K) for one graph of K steps, K = 1 .. 200 (C3, mixed).  The intercept is what a
20-step region pays once; the slope is the in-run step."""
import importlib, os, sys, time, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
spec = S.make_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
rows = []
for K in (1, 2, 5, 10, 20, 40, 100, 200):
    ctx.run_graph(2 * K, K); ctx.graph_prepare(K); ctx.synchronize(); torch.cuda.synchronize()
    tl, tr = [], []
    for _ in range(100):
        ctx.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter(); ctx.run_graph(K, K); t1 = time.perf_counter(); ctx.synchronize(); torch.cuda.synchronize(); t2 = time.perf_counter()
        tl.append((t1 - t0) * 1e6); tr.append((t2 - t0) * 1e6)
    rows.append((K, statistics.median(tl), statistics.median(tr)))
    print("K %4d: host call %7.1f us, region %8.1f us  (%.2f us/step)" % (K, rows[-1][1], rows[-1][2], rows[-1][2] / K))
(k1, _, r1), (k2, _, r2) = rows[-2], rows[-1]
slope = (r2 - r1) / (k2 - k1)
print("slope (100 -> 200 steps) %.2f us/step; intercepts: " % slope + ", ".join("K=%d: %.1f" % (k, r - slope * k) for k, _, r in rows))
ctx.close()
