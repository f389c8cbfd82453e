import importlib, os, sys, hashlib
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
spec = S.make_config("C3", scale=2)
h = {}
for k in ("1", "0"):
    os.environ["VVHIP_PERIODIC"] = "1"
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether", tune={"periodic_kernels": int(k)})      # 0: slot words loaded, same layout
    ctx.run_graph(20000, 100); ctx.synchronize()
    h[k] = (hashlib.sha1(ctx.getVelm().tobytes()).hexdigest(), hashlib.sha1(ctx.getPosq().tobytes()).hexdigest(), ctx.getGroupTemperatures())
    ctx.close()
print(h["1"][0] == h["0"][0], h["1"][1] == h["0"][1], h["1"][2], h["0"][2])
