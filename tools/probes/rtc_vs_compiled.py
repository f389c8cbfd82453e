"""The headline step with the compiled kernels and with the same stage sets compiled at run time (VVHIP_RTC=2), alternating, same box."""
import importlib, os, subprocess, sys
code = '''
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
spec = S.make_config(sys.argv[1])
it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
if sys.argv[1] == "C4": it.setCosAcceleration(0.02)
ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
ctx.run_graph(2000, 100); ctx.synchronize()
t0 = time.perf_counter(); ctx.run_graph(20000, 100); ctx.synchronize(); t = time.perf_counter() - t0
print("%.0f" % (20000 / t), I.Context.rtc_stats())
ctx.close()
'''
for cfg in ("C3", "C4"):
    for rep in range(3):
        for mode in ("0", "2"):
            r = subprocess.run([sys.executable, "-c", code, cfg], capture_output=True, text=True, env=dict(os.environ, VVHIP_RTC=mode))
            print(cfg, "VVHIP_RTC=" + mode, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
