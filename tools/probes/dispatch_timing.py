import importlib, sys
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
for hb in (False, True):
    spec = S.make_config("C3", hbonds=hb)
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    ctx.run_graph(2000, 100); ctx.synchronize()
    for n in (200, 2000):
        ctx.run_eager(50); ctx.timing(4 * n + 8); ctx.run_eager(n); r = ctx.timing_read(); ctx.timing(False)
        print("hbonds", hb, n, "eager steps: A %.3f us  B %.3f us other %.3f (launches %s)" % (r["ms_a"] / r["launches"][0] * 1e3, r["ms_b"] / r["launches"][1] * 1e3, r["ms_other"] / max(r["launches"][2], 1) * 1e3, r["launches"]))
    print("   back to back A %.3f B %.3f" % (ctx.time_kernel(0, 100) * 1e3, ctx.time_kernel(1, 100) * 1e3))
    ctx.close()
