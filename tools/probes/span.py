"""Whole-kernel view from the instrumented build: active span (first wave in -> last wave out), gap to the previous launch,
median wave entry delay and median wave lifetime, for back-to-back launches of kernel A / B at the headline size."""
import importlib, sys
import ctypes as C
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S, H = pkg.integrator, pkg.systems, pkg.vvhip
for cfg in ("C3", "C2"):
    spec = S.make_config(cfg)
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02 if cfg == "C3" else 0.0)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    ctx.run_graph(200, 100); ctx.synchronize()
    hw = 512 if cfg == "C3" else 0
    for name, k, f in (("A kick+KE (no store)", 0, 32 | 1024 | (1 << 19)), ("B full + kick", 1, 2048 | 1 | 16 | hw | (1 << 17)), ("B no chain wave", 1, 1 | 16 | hw | (1 << 17))):
        out = (C.c_double * 8)()
        H.check(H.lib.vvhip_debug_span(ctx.plan, k, f, 40, C.byref(out)), ctx.plan)
        ms = C.c_double(0)
        H.check(H.lib.vvhip_time_kernel(ctx.plan, k, f, 100, C.byref(ms)), ctx.plan)
        print(f"{cfg} {name:16s}: launch-to-launch {ms.value * 1e3:5.2f} us | span {out[0]:6.0f} ns, gap to previous {out[1]:6.0f} ns, median entry +{out[2]:5.0f} ns, median wave life {out[3]:6.0f} ns, p90 {out[4]:5.0f}, max {out[5]:5.0f}; last wave out: block {int(out[6])}, entered +{out[7]:4.0f} ns", flush=True)
    ctx.close()
