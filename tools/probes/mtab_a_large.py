"""Kernel A with the static mass table (A_MTAB) in the bandwidth-bound regime, where the kernel is co-limited by VALU issue: steps/s with the table
(test hook "mass_tab_a") and without, alternating, same box.  Round 4: +2.6 % at 8.9 M particles; round 5, same probe on the round's code: 2 093.6
against 2 102.3 steps/s at 8.9 M, 5 813 against 6 082 at 3.3 M particles -- it does not reproduce, kernel A keeps forming its masses at every size."""
import importlib, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
for cfg, nsteps in (("C3x80", 1500), ("C3x30", 3000)):
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
        print("%s (%d particles) mass table in kernel A %d: steps/s %s  median %.1f" % (cfg, spec.num_atoms, mt, " ".join("%.1f" % x for x in res[mt]), statistics.median(res[mt])), flush=True)
