import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
def rate(spec, n=4000):
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    ctx.run_graph(400, 100); ctx.synchronize()
    t0 = time.perf_counter(); ctx.run_graph(n, 100); ctx.synchronize(); t = time.perf_counter() - t0
    w = ctx.info.num_waves; ctx.close()
    return n / t, w
base = S.make_config("C3")
print("base", rate(base))
for kinds in ((1,), (3,), (0,)):
    for inter in (False, True):
        spec = S.add_virtual_sites(base, kinds=kinds, interleaved=inter)
        print("kinds", kinds, "interleaved", inter, rate(spec))
# the same particles without describing them as sites (massless extras that get no lane): the layout's share
spec = S.add_virtual_sites(base, kinds=(3,), interleaved=False); spec.virtual_sites = []
print("massless extras, not described", rate(spec))
# four-site water: C2's 3 333 rigid molecules (SETTLE) with an averaged M site each
w = S.make_config("C2", hbonds=True)
def rate_w(spec, n=4000):
    it = I.VVIntegrator(300.0, 10, 1.0, 40, 0.002); it.setMaxDrudeDistance(0.0)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    ctx.run_graph(400, 100); ctx.synchronize()
    t0 = time.perf_counter(); ctx.run_graph(n, 100); ctx.synchronize(); t = time.perf_counter() - t0
    ctx.close()
    return n / t
print("rigid water, 3 sites", "%.0f" % rate_w(w), "; with an M site per molecule", "%.0f" % rate_w(S.add_virtual_sites(w, kinds=(1,))))
