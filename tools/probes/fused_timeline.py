"""In-kernel timeline of the ONE-launch step (vv_kernel_b<.., SFA>) from the shader-clock stamps of the instrumented build
(VVHIP_LIB=tools/probes/libs/libvvhip_ts.so; ns after the sampled block's first stamp, 2.4 GHz assumed), next to the two-launch step's
kernels (tools/probes/kernel_timeline.py).   python tools/probes/fused_timeline.py C3 C5 C2
tile wave:  entry | loads arrived (+ extra forces) | kick + sums done | partials in LDS | behind barrier 1 | prep done | scales received | compute done | stores drained
thermostat: entry | at barrier 1 | behind it | published | all words held (poll rounds) | folded | ke2 | released | chain done | state stored"""
import importlib, os, sys
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S, H = pkg.integrator, pkg.systems, pkg.vvhip
GHZ = 2.4
for arg in sys.argv[1:] or ["C3"]:
    cfg, hb = (arg.split("+") + [""])[:2]
    spec = S.make_config("C3" if cfg == "C4" else cfg, hbonds=bool(hb))
    it = I.VVIntegrator(300.0 if cfg == "C2" else 333.0, 10, 1.0, 40, 0.002 if cfg == "C2" else 0.001)
    if cfg not in ("C1", "C2"): it.setMaxDrudeDistance(0.02)
    if cfg == "C4": it.setCosAcceleration(0.02)
    if cfg == "C5":
        lz = float(spec.box[2]); it.setMirrorLocation(lz / 2); it.setElectricField(2.0 / lz * 2 * 1.602176634e-22)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    ctx.run_graph(400, 100); ctx.synchronize()
    tiles = 7 if ctx.info.num_waves > 256 else 1
    nb = (ctx.info.num_waves + tiles - 1) // tiles
    print(f"== {arg}: {spec.num_atoms} particles, {ctx.info.num_waves} waves, one launch active {ctx.fused_status()[0]}")
    for block in sorted({0, 1, nb // 2, nb - 1}):
        for rep in range(4):
            ctx.calcForces()
            out = (C.c_longlong * 128)()
            H.check(H.lib.vvhip_debug_timestamps_fused(ctx.plan, block, C.byref(out)), ctx.plan)
            t = np.array(out, dtype=np.int64).reshape(8, 16)
        ws = [w for w in range(7) if t[w, 0] > 0]
        t0 = min([t[w, 0] for w in ws] + ([t[7, 0]] if t[7, 0] > 0 else []))
        f = lambda w, k: f"{(t[w, k] - t0) / GHZ:5.0f}" if t[w, k] > 0 else "    -"
        line = f"  block {block:4d}: " + " | ".join(f"t{w} " + " ".join(f(w, k) for k in (0, 6, 7, 8, 9, 2, 3, 4, 5)) for w in (ws if os.environ.get("ALL_WAVES") else ws[:2]))
        line += " | thermo " + " ".join(f(7, k) for k in (0, 6, 7, 8, 9)) + f" ({t[7, 10] if 0 <= t[7, 10] < 100 else 'shared'} rounds) " + " ".join(f(7, k) for k in (1, 4, 5, 2, 3))
        print(line, flush=True)
    ctx.close()
