"""How to launch a SHORT timed region (the driver's --steps 20 --warmup 5: ONE region of 20 steps between two fences)?  One replay of a 20-step graph
carries ~22 us of its own (round 5, region_overhead.py: ~8 us on the GPU side per graph launch + the host's launch and synchronisation latency).  With
ONE integrator launch per step the host can keep ahead of the GPU with plain launches (two launches of ~3.5 us per 9.6 us step): the region's time for
K = 20 / 40 / 100 / 2000 as a graph replay, as plain launches from the C loop (vvhip_run_eager), and as a few plain steps in front of a graph.
usage: python tools/probes/region_modes.py [config]"""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
spec = S.make_config("C3" if cfg == "C4" else cfg)
it = I.VVIntegrator(300.0 if cfg == "C2" else 333.0, 10, 1.0, 40, 0.002 if cfg == "C2" else 0.001)
if cfg not in ("C1", "C2"): it.setMaxDrudeDistance(0.02)
if cfg == "C4": it.setCosAcceleration(0.02)
ctx = I.Context(spec, it, precision="mixed", force_provider=os.environ.get("PROVIDER", "tether"))
ctx.run_graph(400, 100); ctx.synchronize()


def region(fn, reps=41):
    ts, enq = [], []
    for _ in range(reps):
        ctx.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        ctx.synchronize()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append(t2 - t0); enq.append(t1 - t0)
    return statistics.median(ts) * 1e6, statistics.median(enq) * 1e6, min(ts) * 1e6


print(f"{cfg}: one-launch step active: {ctx.fused_status()[0]}")
for k in (20, 40, 100, 2000):
    g = min(k, 100)
    ctx.graph_prepare(g)
    rows = [("graph replay", lambda: ctx.run_graph(k, g)), ("plain launches", lambda: ctx.run_eager(k))]
    if k >= 20:
        lead = 2
        gl = min(k - lead, 100); gl -= gl % 2
        ctx.run_graph(gl, gl); ctx.synchronize()
        rows.append((f"{lead} plain steps + graphs of {gl}", lambda: (ctx.run_eager(lead), ctx.run_graph(k - lead, gl))))
    if k >= 20:
        for lead in (2, 4):
            gl = k - lead
            if gl > 100 or gl % 2: continue
            ctx.graph_prepare(lead); ctx.graph_prepare(gl)
            rows.append((f"graph of {lead} + graph of {gl}", (lambda lead=lead, gl=gl: (ctx.run_graph(lead, lead), ctx.run_graph(gl, gl)))))
    for name, fn in rows:
        fn(); ctx.synchronize()
        med, enq, lo = region(fn, 41 if k <= 200 else 9)
        print(f"  K = {k:5d}, {name:32s}: region {med:8.1f} us (min {lo:8.1f}), enqueue returned after {enq:7.1f} us -> {k / med * 1e6:9.0f} steps/s, {med / k:6.2f} us/step", flush=True)
ctx.close()
