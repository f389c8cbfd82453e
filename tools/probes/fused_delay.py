"""One-launch step: steps/s for several settings of its rendezvous wait (test hooks "fused_poll_delay": >= 0 pins the wait in units of 256
clocks, -1 self-tuning; "fused_late_shift": how many late blocks make the wait grow), in rotation with the two-launch step on one box.
    python tools/probes/fused_delay.py C3,C5,C2 "fused_poll_delay=-1;fused_poll_delay=6;fused_poll_delay=-1,fused_late_shift=2" [rotations]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import runpy
argv, sys.argv = sys.argv, sys.argv[:1]
mod = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fused_ab.py"), run_name="lib")
configs = (argv[1] if len(argv) > 1 else "C3").split(",")
variants = (argv[2] if len(argv) > 2 else "fused_poll_delay=-1;fused_poll_delay=6").split(";")
rot = int(argv[3]) if len(argv) > 3 else 2
short = lambda v: v.replace("fused_poll_delay=", "wait ").replace("fused_late_shift=", "shift ")
for cfg in configs:
    ctxs = {}
    for v in variants:
        mod["EXTRA"].clear(); mod["EXTRA"].update({k: int(x) for k, x in (kv.split("=") for kv in v.split(","))})
        ctxs[v] = mod["make"](cfg, True)[0]
    two = mod["make"](cfg, False)[0]
    for r in range(rot):
        row = {v: mod["rate"](ctxs[v], 20000) for v in variants}
        base = mod["rate"](two, 20000)
        print(f"{cfg} rotation {r}: two launches {base:8.0f} | " + " | ".join(f"{short(v)}: {row[v]:8.0f} ({100 * (row[v] / base - 1):+.1f} %)" + (f" [at {ctxs[v].fused_wait_units()}]" if "delay=-1" in v else "") for v in variants), flush=True)
    for c in list(ctxs.values()) + [two]:
        c.close()
