"""What a K-step timed region costs beyond its K steps (the driver's --steps 20 --warmup 5: ONE region of 20 steps between two fences): the
region's time for K = 20 replayed as graphs of 20 / 10 / 4 / 2 steps (a small first graph lets the GPU start while the host still enqueues the
rest), the host time until the enqueue returns, and the slope / intercept over K.  (Round 5: ~22 us of a 230 us region are fixed -- one graph
launch + one synchronisation; smaller graphs cost ~8 us each on the GPU side; asking the stream in a loop instead of hipStreamSynchronize
changes nothing, the runtime spins already: profiles/r05i_region_overhead.txt.)
usage: python tools/probes/region_overhead.py [config]"""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
spec = S.make_config(cfg)
it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
it.setMaxDrudeDistance(0.02)
ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
ctx.run_graph(400, 100); ctx.synchronize()


def region(k, spg, reps=41):
    ctx.graph_prepare(spg)
    ts, enq = [], []
    for _ in range(reps):
        ctx.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.run_graph(k, spg)
        t1 = time.perf_counter()
        ctx.synchronize()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append(t2 - t0); enq.append(t1 - t0)
    return statistics.median(ts) * 1e6, statistics.median(enq) * 1e6, min(ts) * 1e6


print(f"{cfg}: one-launch step active: {ctx.fused_status()[0]}")
for k, spg in [(20, 20), (20, 10), (20, 4), (20, 2), (40, 40), (40, 20), (40, 4), (100, 100), (100, 20), (200, 100), (200, 20), (2000, 100)]:
    med, enq, lo = region(k, spg, 41 if k <= 200 else 9)
    print(f"  K = {k:5d} as graphs of {spg:3d}: region {med:8.1f} us (min {lo:8.1f}), enqueue returned after {enq:7.1f} us -> {k / med * 1e6:9.0f} steps/s, {med / k:6.2f} us/step")
ctx.close()
