"""Tile waves per block of the ONE-launch step, forced through the test hook, for the configurations below the headline size (C5 338 tile waves, C2 157,
C1 32; +hb: with constraints): steps/s of 20 000-step graph runs, rotated.   python tools/probes/fused_shape_sweep.py C5,C5hb,C2,C2hb [rotations]
(the plan's own choice: seven per block above 256 tile waves -- C3's shape --, one below; FUSED_TUNE=block_threads=... is what this sets)"""
import os, subprocess, sys
cfgs = (sys.argv[1] if len(sys.argv) > 1 else "C5,C5hb,C2hb").split(",")
rot = int(sys.argv[2]) if len(sys.argv) > 2 else 2
here = os.path.dirname(os.path.abspath(__file__))
for cfg in cfgs:
    res = {}
    for r in range(rot):
        for t in (0, 1, 2, 3, 4, 5, 7):
            env = dict(os.environ)
            if t: env["FUSED_TUNE"] = f"block_threads={64 * t}"
            out = subprocess.run([sys.executable, os.path.join(here, "fused_one.py"), cfg, "1", "20000"], env=env, capture_output=True, text=True).stdout.strip().splitlines()
            line = out[-1] if out else "failed"
            try: rate = float(line.split(":")[1].split()[0])
            except Exception: rate = float("nan")
            res.setdefault(t, []).append((rate, "active (True" in line))
    print(cfg + ": " + " | ".join(f"{'own' if t == 0 else str(t) + ' per block'} " + "/".join(f"{x / 1e3:.1f}{'' if ok else '*'}" for x, ok in v) for t, v in res.items()) + "   k steps/s (*: one launch not active)", flush=True)
