"""One configuration, one mode (one launch / two launches), `steps` steps from 100-step graphs: for rocprofv3 --kernel-trace --stats.
usage: python tools/probes/fused_one.py C3 1 4000"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0]] + sys.argv[1:]
cfg, fused, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 4000
import runpy
sys.argv = sys.argv[:1]
mod = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fused_ab.py"), run_name="fused_ab_lib")
ctx, it = mod["make"](cfg, bool(fused))
r = mod["rate"](ctx, steps)
print(f"{cfg} {'one launch' if fused else 'two launches'}: {r:.0f} steps/s, active {ctx.fused_status()}")
ctx.close()
