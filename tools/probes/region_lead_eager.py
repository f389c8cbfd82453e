"""The driver's 20-step timed region as ONE graph replay vs k eager steps (plain launches, inside vvhip_run_graph) in front of a graph of the
remaining 20 - k: plain launches reach the GPU sooner than a graph launch does, and the graph launch then travels while the GPU is busy.
(Scratch build with the test hook key `experiment` = k.)"""
import importlib, os, sys, time, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
for cfg in sys.argv[1:] or ["C3", "C4"]:
    spec = S.make_config(cfg)
    for k in (0, 2, 4, 0, 2, 4):
        it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
        if cfg == "C4": it.setCosAcceleration(0.02)
        ctx = I.Context(spec, it, precision="mixed", force_provider="tether", tune={"experiment": k})
        for _ in range(5): ctx.run_graph(20, 20 - k)
        ctx.synchronize(); torch.cuda.synchronize()
        tr = []
        for _ in range(200):
            ctx.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.run_graph(20, 20 - k)
            ctx.synchronize(); torch.cuda.synchronize()
            tr.append((time.perf_counter() - t0) * 1e6)
        print("%s: %d eager steps + graph of %2d: region %.1f us (min %.1f)  -> %.0f steps/s" % (cfg, k, 20 - k, statistics.median(tr), min(tr), 20 / statistics.median(tr) * 1e6))
        ctx.close()
