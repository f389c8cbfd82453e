"""Dispatch-timestamp clock of eager steps against wall-clock rates, one-launch and two-launch step (why bench.py's kernel_times sees the
one-launch kernel at 11 us in eager steps where rocprofv3 sees 7.5 us in a graph replay).  python tools/probes/fused_eager_clock.py C3"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import runpy
argv, sys.argv = sys.argv, sys.argv[:1]
mod = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fused_ab.py"), run_name="lib")
for cfg in (argv[1] if len(argv) > 1 else "C3").split(","):
    for fused in (True, False):
        ctx, it = mod["make"](cfg, fused)
        ctx.run_graph(2000, 100); ctx.synchronize()
        w0 = ctx.fused_wait_units() if fused else None
        t0 = time.perf_counter(); ctx.run_graph(4000, 100); ctx.synchronize(); g = 4000 / (time.perf_counter() - t0)
        ctx.run_eager(200); ctx.synchronize()
        t0 = time.perf_counter(); ctx.run_eager(2000); ctx.synchronize(); e = 2000 / (time.perf_counter() - t0)
        w1 = ctx.fused_wait_units() if fused else None
        ctx.timing(1000)
        ctx.run_eager(200)
        r = ctx.timing_read(); ctx.timing(0)
        w2 = ctx.fused_wait_units() if fused else None
        print(f"{cfg} {'one launch ' if fused else 'two launches'}: graph {g:8.0f} steps/s, eager {e:8.0f} steps/s; dispatch timestamps in eager steps: "
              f"A {1e3 * r['ms_a'] / max(r['launches'][0], 1):6.2f} us x{r['launches'][0]}, B {1e3 * r['ms_b'] / max(r['launches'][1], 1):6.2f} us x{r['launches'][1]}; wait units {w0} -> {w1} -> {w2}", flush=True)
        ctx.close()
