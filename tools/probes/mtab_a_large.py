"""Kernel A with the static mass table (A_MTAB, test hook key mass_tab_a) at large N, where the kernel is bound by VALU issue and not by bytes:
steps/s and the kernels' live durations, alternating, same box."""
import importlib, os, sys, time, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
for cfg, nsteps in (("C3x80", 1500), ("C3x8", 10000)):
    spec = S.make_config("C3", scale=float(cfg[3:]))
    res = {0: [], 1: []}
    for rep in range(3):
        for mt in (0, 1):
            it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
            ctx = I.Context(spec, it, precision="mixed", force_provider="tether", tune={"mass_tab_a": mt})
            ctx.run_graph(nsteps // 5, 20); ctx.synchronize()
            t0 = time.perf_counter(); ctx.run_graph(nsteps, 20); ctx.synchronize(); t = time.perf_counter() - t0
            res[mt].append(nsteps / t)
            ctx.close()
    for mt in (0, 1):
        print("%s mass table in A %d: steps/s %s  median %.1f" % (cfg, mt, " ".join("%.1f" % x for x in res[mt]), statistics.median(res[mt])))
