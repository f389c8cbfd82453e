"""Steps/s of the three ways to run the same steps on one GPU: hipGraph replay of fused steps, fused steps enqueued from C,
and the un-fused per-KernelImpl sequence enqueued from C (what the OpenMM adapter issues around a host constraint solver)."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
for cfg, cos in (("C3", 0.0), ("C3", 0.02), ("C2", 0.0), ("C3-classic", 0.0)):
    res = {}
    ref = None
    for mode in (("graph", "eager") if "classic" in cfg else ("graph", "eager", "unfused")):
        spec = S.make_config(cfg.split("-")[0])
        it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.0 if cfg == "C2" else 0.02); it.setCosAcceleration(cos); it.setUseMiddleScheme("classic" not in cfg)
        ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
        run = {"graph": lambda n: ctx.run_graph(n, 100), "eager": ctx.run_eager, "unfused": ctx.run_eager_unfused}[mode]
        run(20); ctx.synchronize()
        x20 = ctx.getPositions()
        if mode == 'graph': ref20 = x20
        else: print(f'   {mode}: max |x - x_graph| after 20 steps = {np.abs(x20 - ref20).max():.2e}')
        run(180); ctx.synchronize()
        t0 = time.perf_counter(); run(3000); ctx.synchronize(); dt = time.perf_counter() - t0
        res[mode] = 3000 / dt
        ctx.close()
    print(f"{cfg} cos={cos}: graph {res['graph']:8.0f}  eager {res['eager']:8.0f}  un-fused {res.get('unfused', float('nan')):8.0f} steps/s", flush=True)
