"""What does the thermostat chain cost inside kernel B?  The same plan with 0 (fold only, factors = 1), 1, 2, 3 loops per step
(VVIntegrator::setLoopsPerStep): back-to-back launches of kernel B and steps/s from graph replays, headline box and the small ones."""
import importlib, sys, time
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
for cfg in (sys.argv[1:] or ["C3", "C2"]):
    spec = S.make_config(cfg)
    for loops in (1, 2, 3, 4):
        it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001, loopsPerStep=loops)
        if cfg not in ("C1", "C2"):
            it.setMaxDrudeDistance(0.02)
        ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
        ctx.run_graph(2000, 100); ctx.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); ctx.run_graph(6000, 100); ctx.synchronize(); ts.append(time.perf_counter() - t0)
        sps = 6000 / sorted(ts)[1]
        b = min(ctx.time_kernel(1, 200) for _ in range(5)) * 1e3
        a = min(ctx.time_kernel(0, 200) for _ in range(5)) * 1e3
        print(f"{cfg} loops per step {loops}: {sps / 1e3:7.2f} k steps/s ({1e6 / sps:6.3f} us per step)   B back to back {b:5.2f} us   A {a:5.2f} us", flush=True)
        ctx.close()
