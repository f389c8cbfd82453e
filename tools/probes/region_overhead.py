"""Where the fixed cost of a 20-step timed region goes (bench.py under the driver's flags): host time of the run_graph call, of the two synchronisations
on an idle GPU, and the region as bench.py times it."""
import importlib, os, sys, time, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
spec = S.make_config("C3")
it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
ctx.run_graph(200, 20); ctx.graph_prepare(20); ctx.synchronize(); torch.cuda.synchronize()
def med(f, n=200):
    v = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); v.append((time.perf_counter() - t0) * 1e6)
    return statistics.median(v)
print("sync pair on an idle GPU      %.1f us" % med(lambda: (ctx.synchronize(), torch.cuda.synchronize())))
def launch_only():
    ctx.run_graph(20, 20)
t_launch = []
t_region = []
for _ in range(200):
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter(); ctx.run_graph(20, 20); t1 = time.perf_counter(); ctx.synchronize(); torch.cuda.synchronize(); t2 = time.perf_counter()
    t_launch.append((t1 - t0) * 1e6); t_region.append((t2 - t0) * 1e6)
print("run_graph(20) host call       %.1f us" % statistics.median(t_launch))
print("region (launch + both syncs)  %.1f us  -> %.0f steps/s" % (statistics.median(t_region), 20 / statistics.median(t_region) * 1e6))
ctx.run_graph(2000, 100); ctx.synchronize()
t0 = time.perf_counter(); ctx.run_graph(20000, 100); ctx.synchronize(); t = time.perf_counter() - t0
print("20 steps inside a long run    %.1f us" % (t / 1000 * 1e6))
ctx.close()
