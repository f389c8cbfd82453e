"""Large boxes other than the Drude ionic liquid: per-kernel times (in sequence) and steps/s -- flexible water, rigid water (SETTLE), the
ionic liquid with the cos perturbation (three-launch form at this size).  tools/probes/large_variants.py [lib.so]"""
import importlib, os, sys, shutil
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if len(sys.argv) > 1:
    shutil.copy(sys.argv[1], os.path.join(sys.path[0], "openmm-velocityverlet_amd", "lib", "libvvhip.so"))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
import bench
def run(name, spec, cos=0.0, maxd=0.0, T=300.0, dt=0.002, middle=True):
    it = I.VVIntegrator(T, 10, 1.0, 40, dt); it.setMaxDrudeDistance(maxd); it.setCosAcceleration(cos); it.setUseMiddleScheme(middle)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    ctx.run_graph(40, 20); ctx.synchronize()
    import time
    t0 = time.perf_counter(); ctx.run_graph(100, 20); ctx.synchronize(); sps = 100 / (time.perf_counter() - t0)
    a, b, how = bench.kernel_times(ctx, 20, 3) if middle else (0, 0, '')
    print(f"{name}: {spec.num_atoms} particles, periodic={ctx.info.periodic_layout}, {sps:.0f} steps/s, A {a*1e3:.1f} us, B {b*1e3:.1f} us", flush=True)
    ctx.close()
w = S.spce_water(1_000_000, seed=5)
run("flexible water", w)
run("rigid water", S.rigid_water(w))
run("C3x8 + cos", S.make_config("C4", scale=8), cos=0.01, maxd=0.02, T=333.0, dt=0.001)
run("C3x8 classic scheme", S.make_config("C3", scale=8), maxd=0.02, T=333.0, dt=0.001, middle=False)
