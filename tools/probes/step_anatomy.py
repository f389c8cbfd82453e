"""Anatomy of a fused step IN SEQUENCE from the instrumented build (VVHIP_LIB=tools/probes/libs/libvvhip_ts.so, -DVV_KERNEL_TIMESTAMPS):
force provider -> kernel A -> kernel B of two consecutive eager steps, every wave stamping the 100 MHz wall clock at entry and exit
(vvhip_debug_step_spans).  Prints, per launch, first / median / last wave in and out [ns after the first entry], the kernel's active
span, and the gap to its predecessor (last wave out -> first wave in): what separates body, ramp, tail and boundary.
    python tools/probes/step_anatomy.py C3 C4 C5 C2 C3+hbonds"""
import importlib, sys
import ctypes as C
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
FUSED = int(os.environ.get("FUSED", "1"))      # 0: force the two-launch step
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S, H = pkg.integrator, pkg.systems, pkg.vvhip
for arg in sys.argv[1:] or ["C3", "C4"]:
    cfg, hb = (arg.split("+") + [""])[:2]
    spec = S.make_config(cfg, hbonds=bool(hb))
    dt = 0.002 if cfg == "C2" else 0.001
    it = I.VVIntegrator(300.0 if cfg == "C2" else 333.0, 10, 1.0, 40, dt)
    if cfg not in ("C1", "C2"): it.setMaxDrudeDistance(0.02)
    if cfg == "C4": it.setCosAcceleration(0.02)
    if cfg == "C5":
        lz = float(spec.box[2]); it.setMirrorLocation(lz / 2); it.setElectricField(2.0 / lz * 2 * 1.602176634e-22)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether", tune={"fused": FUSED})
    ctx.run_graph(400, 100); ctx.synchronize()
    acc = []
    for rep in range(7):
        out = (C.c_double * 36)()
        H.check(H.lib.vvhip_debug_step_spans(ctx.plan, 12, ctx.site.ptr, ctx.k_tether, ctx.k_drude, C.byref(out)), ctx.plan)
        acc.append(np.array(out).reshape(6, 6))
    t = np.median(np.array(acc), axis=0)
    one = ctx.fused_status()[0]
    names = ["force", "A+B", "force", "A+B", "-", "-"] if one else ["force", "A", "B", "force", "A", "B"]
    nl = 4 if one else 6
    print(f"== {arg}: {spec.num_atoms} particles, {ctx.info.num_waves} waves; median of 7 runs; ns")
    print("   launch   first-in  med-in  last-in | first-out med-out last-out |  span  gap-to-previous")
    for l in range(nl):
        gap = t[l, 0] - t[l - 1, 5] if l else float("nan")
        print(f"   {names[l]:6s} {t[l,0]:9.0f} {t[l,1]:7.0f} {t[l,2]:8.0f} | {t[l,3]:9.0f} {t[l,4]:7.0f} {t[l,5]:8.0f} | {t[l,5]-t[l,0]:5.0f}  {gap:6.0f}")
    h = nl // 2
    print(f"   step (first-in of force to first-in of next force): {t[h,0]-t[0,0]:.0f} ns; sum of spans {sum(t[l,5]-t[l,0] for l in range(h,nl)):.0f}, sum of gaps {sum(t[l,0]-t[l-1,5] for l in range(h,nl)):.0f}", flush=True)
    ctx.close()
