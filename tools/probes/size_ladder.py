"""Steps/s of the plan's own choices (layout, launch shape, chain in kernel B or as its own launch) over the sizes from the headline to 8.9 M particles."""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
os.environ.pop("VVHIP_PERIODIC", None)
for nz in [int(x) for x in (sys.argv[1:] or ["3", "6", "12", "18", "24", "30", "36", "48", "72", "120", "240"])]:
    spec = S.bulk_Im21(cells=(2, 2, nz))
    n = max(400, int(4e8 / spec.num_atoms))
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    ctx.run_graph(n // 5, 20); ctx.synchronize()
    t0 = time.perf_counter(); ctx.run_graph(n, 20); ctx.synchronize(); t = time.perf_counter() - t0
    print("%8d particles %6d waves, %s layout: %8.0f steps/s" % (spec.num_atoms, ctx.info.num_waves, "arithmetic" if ctx.info.periodic_layout else "best-fit", n / t))
    ctx.close()
