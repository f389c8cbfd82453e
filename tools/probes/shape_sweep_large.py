"""Launch shapes in the bandwidth-bound regime (chain as its own launch): block size x blocks per CU of kernels A and B at 4.4 M / 8.9 M particles."""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
def rate(spec, tune, n):
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether", tune=tune)
    ctx.run_graph(n // 5, 20); ctx.synchronize()
    t0 = time.perf_counter(); ctx.run_graph(n, 20); ctx.synchronize(); t = time.perf_counter() - t0
    ctx.close()
    return n / t
for nz in [int(x) for x in (sys.argv[1:] or ["120", "240"])]:
    spec = S.bulk_Im21(cells=(2, 2, nz))
    n = max(300, int(3e8 / spec.num_atoms))
    out = [("own choice", rate(spec, {}, n))]
    for bt in (256, 384, 448):
        for ka, kb in ((8, 4), (4, 4), (4, 2), (6, 3), (8, 2), (2, 2)):
            if (bt // 64) * kb > 16: continue
            out.append(("%d thr A %dx B %dx" % (bt, ka, kb), rate(spec, {"block_threads": bt, "grid_cap_a": 256 * ka, "grid_cap_b": 256 * kb}, n)))
    out2 = sorted(out[1:], key=lambda x: -x[1])
    print("%d particles: own choice %.1f steps/s; best: %s" % (spec.num_atoms, out[0][1], ", ".join("%s %.1f" % o for o in out2[:6])))
