"""Is the thermostat wave's time an instruction-cache effect?  Stamps of kernel B right after kernel A (as in a step) and of a second
kernel B launched immediately after the first."""
import importlib, sys
import ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S, H = pkg.integrator, pkg.systems, pkg.vvhip
spec = S.make_config("C3")
it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
ctx.run_graph(200, 100); ctx.synchronize()
FULL = 2048 | 1 | 16 | 512
GHZ = 2.4
def show(tag, out):
    t = np.array(out, dtype=np.int64).reshape(8, 16)
    t0 = t[t > 0].min()
    print(f"  {tag:28s} thermo " + " ".join(f"{(t[7, k] - t0) / GHZ:6.0f}" for k in (0, 1, 4, 2, 3)) + "   tile0 " + " ".join(f"{(t[0, k] - t0) / GHZ:6.0f}" for k in range(6)))
for block in (1, 125):
    for rep in range(3):
        it.step(1)
        ctx.calcForces(); H.check(H.lib.vvhip_step_middle_phase(ctx.plan, 0, 0), ctx.plan)
        o1 = (C.c_longlong * 128)(); o2 = (C.c_longlong * 128)()
        H.check(H.lib.vvhip_debug_timestamps(ctx.plan, FULL, block, C.byref(o1)), ctx.plan)
        H.check(H.lib.vvhip_debug_timestamps(ctx.plan, FULL, block, C.byref(o2)), ctx.plan)
        H.check(H.lib.vvhip_step_middle_phase(ctx.plan, 1, 0), ctx.plan)
    print(f"block {block}")
    show("B after A (as in a step)", o1)
    show("B again right after", o2)
ctx.close()
