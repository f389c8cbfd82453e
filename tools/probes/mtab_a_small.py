"""Kernel A with the static mass table (A_MTAB, test hook key mass_tab_a) at the headline size: C3, C4, C5, alternating, same box."""
import importlib, os, sys, time, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
for cfg in ("C4", "C3", "C5", "C2", "C1"):
    spec = S.make_config(cfg)
    res = {0: [], 1: []}
    for rep in range(3):
        for mt in (0, 1):
            it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02 if len(spec.drude_pairs) else 0.0)
            if cfg == "C4": it.setCosAcceleration(0.02)
            if cfg == "C5": it.setMirrorLocation(float(spec.box[2]) / 2)
            ctx = I.Context(spec, it, precision="mixed", force_provider="tether", tune={"mass_tab_a": mt})
            ctx.run_graph(2000, 100); ctx.synchronize()
            t0 = time.perf_counter(); ctx.run_graph(20000, 100); ctx.synchronize(); t = time.perf_counter() - t0
            res[mt].append(20000 / t)
            ctx.close()
    for mt in (0, 1):
        print("%s mass table in A %d: steps/s %s  median %.0f" % (cfg, mt, " ".join("%.0f" % x for x in res[mt]), statistics.median(res[mt])))
