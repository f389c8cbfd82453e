import importlib, sys, time
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
for frac in (1.0, 0.5, 0.25, 0.125):
    spec = S.make_config("C3") if frac == 1.0 else S.drude_il(cells=(1, 1, 1), pairs_per_cell=int(3000 * frac))
    for blk in (None,):
        it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
        ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
        ctx.run_graph(400, 100); ctx.synchronize()
        t0 = time.perf_counter(); ctx.run_graph(4000, 100); ctx.synchronize(); dt = time.perf_counter() - t0
        ta = min(ctx.time_kernel(0, 100) for _ in range(3)) * 1e3; tb = min(ctx.time_kernel(1, 100) for _ in range(3)) * 1e3
        print(f"{spec.num_atoms:7d} particles, {ctx.info.num_waves} waves: {4000/dt:9.0f} steps/s = {dt/4000*1e6:.2f} us/step; A {ta:.2f} us, B {tb:.2f} us", flush=True)
        ctx.close()
