"""Best-fit against arithmetic work-item layout by size (each with the plan's own launch shape): where the automatic switch belongs."""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
def rate(spec, env, n):
    os.environ["VVHIP_PERIODIC"] = env
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    ctx.run_graph(n // 5, 20); ctx.synchronize()
    t0 = time.perf_counter(); ctx.run_graph(n, 20); ctx.synchronize(); t = time.perf_counter() - t0
    w = ctx.info.num_waves; ctx.close()
    return n / t, w
for nz in [int(x) for x in (sys.argv[1:] or ["6", "12", "18", "24", "36", "48", "72", "120"])]:
    spec = S.bulk_Im21(cells=(2, 2, nz))
    n = max(400, int(4e8 / spec.num_atoms))
    r = [rate(spec, e, n) for e in ("0", "1", "0", "1")]
    print("%8d particles: best-fit %6d waves %8.0f %8.0f steps/s | arithmetic %6d waves %8.0f %8.0f" % (spec.num_atoms, r[0][1], r[0][0], r[2][0], r[1][1], r[1][0], r[3][0]))
