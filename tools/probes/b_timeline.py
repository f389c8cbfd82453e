"""Timeline of kernel B at the headline size from shader-clock stamps (instrumented build tools/probes/libvvhip_ts.so, loaded through
VVHIP_LIB).  Prints, per sampled block, when each wave reached each point, in ns after the earliest stamp of that block."""
import importlib, os, sys
import ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S, H = pkg.integrator, pkg.systems, pkg.vvhip
import os
SMALL = int(os.environ.get("SMALL", "0"))          # SMALL=n: n ion pairs only (12 -> one block), to tell chip-wide effects from block-local ones
spec = S.drude_il(cells=(1, 1, 1), pairs_per_cell=SMALL, seed=3) if SMALL else S.make_config("C3")
it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
ctx.run_graph(200, 100); ctx.synchronize()
FULL = 2048 | 1 | 16 | 512
names_t = ["entry", "loads arrived", "prep done", "scales received", "compute done", "stores drained"]
names_c = ["entry", "acc folded", "chain done", "after barrier"]
GHZ = 2.4      # shader clock assumed for the conversion (MI355X boost); relative numbers are what matter
for flags, label in ((FULL, "B full"), (1 | 16 | 512 | (1 << 17), "B without chain wave")):
    print(label)
    for block in ((0,) if SMALL else (0, 1, 125, 250)):
        acc = []
        for rep in range(5):
            it.step(1)
            ctx.calcForces(); H.check(H.lib.vvhip_step_middle_phase(ctx.plan, 0, 0), ctx.plan)      # kernel A, so that B sees real sums
            out = (C.c_longlong * 128)()
            H.check(H.lib.vvhip_debug_timestamps(ctx.plan, flags, block, C.byref(out)), ctx.plan)
            H.check(H.lib.vvhip_step_middle_phase(ctx.plan, 1, 0), ctx.plan)
            t = np.array(out, dtype=np.int64).reshape(8, 16)
            acc.append(t)
        t = acc[-1]
        t0 = min(t[w, 0] for w in (0, 1, 7) if t[w, 0] > 0)
        line = f"  block {block:4d}: "
        for w in (0, 1):
            line += f"tile{w} " + " ".join(f"{(t[w, k] - t0) / GHZ:6.0f}" if 0 < t[w, k] - t0 < 10**7 else "     -" for k in range(6)) + " | "
        line += "thermo " + " ".join(f"{(t[7, k] - t0) / GHZ:6.0f}" if 0 <= t[7, k] - t0 < 10**7 else "     -" for k in (0, 1, 4, 6, 5, 2, 3))
        print(line)
print("A full (kick + KE): entry, velm arrived, kicked + stored, tile loop done, sums added  [ns]")
for block in ((0,) if SMALL else (0, 1, 125, 250)):
    ctx.calcForces()
    out = (C.c_longlong * 128)()
    H.check(H.lib.vvhip_debug_timestamps(ctx.plan, 0x80000000 | 32 | 1024 | (1 << 19), block, C.byref(out)), ctx.plan)      # A_KICK_FULL | A_KE | A_NOSTORE
    H.check(H.lib.vvhip_step_middle_phase(ctx.plan, 1, 0), ctx.plan)
    t = np.array(out, dtype=np.int64).reshape(8, 16)
    t0 = min(t[0, 0], t[1, 0])
    print(f"  block {block:4d}: " + " | ".join("wave%d " % w + " ".join(f"{(t[w, k] - t0) / GHZ:6.0f}" if t[w, k] else "     -" for k in range(5)) for w in (0, 1)))
print("columns tile: " + ", ".join(names_t) + " ; thermo: " + ", ".join(names_c) + "  [ns]")
ctx.close()
