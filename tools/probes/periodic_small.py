"""The arithmetic work-item layout at the headline size (C3 / C4, 111 000 particles), where analyze() does not choose it by itself:
steps/s of the graph-replayed step for the layouts and kernel paths side by side, alternating, same box."""
import importlib, os, sys, time, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
spec = S.make_config(cfg)
variants = [("best-fit layout, slot words (default)", "0", {}),
            ("arithmetic layout, A + B arithmetic", "1", {}),
            ("arithmetic layout, A arithmetic, B slot words", "1", {"periodic_b": 0}),
            ("arithmetic layout, A + B slot words", "1", {"periodic_kernels": 0})]
res = {v[0]: [] for v in variants}
for rep in range(3):
    for name, env, tune in variants:
        os.environ["VVHIP_PERIODIC"] = env
        it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
        if cfg == "C4": it.setCosAcceleration(0.02)
        ctx = I.Context(spec, it, precision="mixed", force_provider="tether", tune=tune)
        ctx.run_graph(2000, 100); ctx.synchronize()
        t0 = time.perf_counter(); ctx.run_graph(20000, 100); ctx.synchronize(); t = time.perf_counter() - t0
        res[name].append(20000 / t)
        waves = ctx.info.num_waves
        ctx.close()
        res[name + " waves"] = waves
for name, _, _ in variants:
    print("%-50s %5d waves  steps/s %s  median %.0f" % (name, res[name + " waves"], " ".join("%.0f" % x for x in res[name]), statistics.median(res[name])))
