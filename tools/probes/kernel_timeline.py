"""In-kernel timelines of the fused step's kernel A and kernel B for any configuration, from the shader-clock stamps of the instrumented
build (VVHIP_LIB=tools/probes/libs/libvvhip_ts.so).  Each sampled block's waves: when they reached each stamp, ns after the block's
earliest stamp (2.4 GHz assumed).  The kernels run in their place (force -> A -> B), the stamped launch replaces the step's own.
    python tools/probes/kernel_timeline.py C3 C4 C5 C2 C3+hbonds
A: entry | loads arrived + extra forces formed | kicked (+ constraints, bias moment) | KE stage done | block sums added
B tile: entry | loads arrived | prep done | scales received | compute done | stores drained ; thermo: entry | acc folded | ke2 | released | chain done | state stored"""
import importlib, sys
import ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S, H = pkg.integrator, pkg.systems, pkg.vvhip
GHZ = 2.4
for arg in sys.argv[1:] or ["C3", "C4"]:
    cfg, hb = (arg.split("+") + [""])[:2]
    spec = S.make_config(cfg, hbonds=bool(hb))
    it = I.VVIntegrator(300.0 if cfg == "C2" else 333.0, 10, 1.0, 40, 0.002 if cfg == "C2" else 0.001)
    if cfg not in ("C1", "C2"): it.setMaxDrudeDistance(0.02)
    if cfg == "C4": it.setCosAcceleration(0.02)
    if cfg == "C5":
        lz = float(spec.box[2]); it.setMirrorLocation(lz / 2); it.setElectricField(2.0 / lz * 2 * 1.602176634e-22)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    ctx.run_graph(400, 100); ctx.synchronize()
    fa, fb = C.c_uint32(0), C.c_uint32(0)
    H.check(H.lib.vvhip_debug_fused_flags(ctx.plan, 0, C.byref(fa)), ctx.plan); H.check(H.lib.vvhip_debug_fused_flags(ctx.plan, 1, C.byref(fb)), ctx.plan)
    nb = max(1, min(251, (ctx.info.num_waves + 6) // 7))
    blocks = sorted({0, 1, nb // 2, nb - 1})
    print(f"== {arg}: {spec.num_atoms} particles, {ctx.info.num_waves} waves, A 0x{fa.value:x} B 0x{fb.value:x}")
    for block in blocks:
        rows_a, rows_b = [], []
        for rep in range(5):
            it.step(1)
            ctx.calcForces()
            out = (C.c_longlong * 128)()
            H.check(H.lib.vvhip_debug_timestamps(ctx.plan, 0x80000000 | fa.value, block, C.byref(out)), ctx.plan)      # kernel A, stamped (real sums for B)
            ta = np.array(out, dtype=np.int64).reshape(8, 16)
            out = (C.c_longlong * 128)()
            H.check(H.lib.vvhip_debug_timestamps(ctx.plan, fb.value, block, C.byref(out)), ctx.plan)                     # kernel B, stamped; the parity is put back
            tb = np.array(out, dtype=np.int64).reshape(8, 16)
            H.check(H.lib.vvhip_step_middle_phase(ctx.plan, 1, 0), ctx.plan)                                             # ... so this is the step's real kernel B
            rows_a.append(ta); rows_b.append(tb)
        ta, tb = rows_a[-1], rows_b[-1]
        wa = [w for w in range(8) if ta[w, 0] > 0]
        if wa:
            t0 = min(ta[w, 0] for w in wa)
            print(f"  A block {block:4d}: " + " | ".join(f"w{w} " + " ".join(f"{(ta[w, k] - t0) / GHZ:5.0f}" for k in range(5)) for w in wa[:3]))
        wb = [w for w in range(7) if tb[w, 0] > 0]
        if wb or tb[7, 0] > 0:
            t0 = min([tb[w, 0] for w in wb] + ([tb[7, 0]] if tb[7, 0] > 0 else []))
            line = f"  B block {block:4d}: " + " | ".join(f"t{w} " + " ".join(f"{(tb[w, k] - t0) / GHZ:5.0f}" if tb[w, k] > 0 else "    -" for k in range(6)) for w in wb[:2])
            if tb[7, 0] > 0:
                line += " | thermo " + " ".join(f"{(tb[7, k] - t0) / GHZ:5.0f}" if tb[7, k] > 0 else "    -" for k in (0, 1, 4, 5, 2, 3))
            print(line)
    sys.stdout.flush()
    ctx.close()
