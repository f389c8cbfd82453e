"""What do the in-kernel constraint sweeps cost?  C3 with its 33 000 HBonds under different solver tolerances (1e30: no corrective
sweep at all, only the hand-over through the LDS page and one check) and, if the library has it, under both solver modes
(VVHIP_SHAKE_MODE=0 Gauss-Seidel sweeps by the central lane, 1 = every peripheral lane its own constraint + direct velocity solve).
steps/s from graph replays; per-kernel averages from HIP events around every launch of eager steps (inflated at this size, but
comparable between the rows)."""
import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
spec = S.make_config(cfg, hbonds=True)
plain = S.make_config(cfg)


def run(spec, tol, label):
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
    it.setMaxDrudeDistance(0.02)
    it.setConstraintTolerance(tol)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    ctx.run_graph(2000, 100); ctx.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); ctx.run_graph(6000, 100); ctx.synchronize(); ts.append(time.perf_counter() - t0)
    sps = 6000 / sorted(ts)[1]
    ctx.run_eager(20); ctx.timing(True); ctx.run_eager(400); r = ctx.timing_read(); ctx.timing(False)
    a, b = r["ms_a"] / max(r["launches"][0], 1) * 1e3, r["ms_b"] / max(r["launches"][1], 1) * 1e3
    bb_a = min(ctx.time_kernel(0, 100) for _ in range(3)) * 1e3
    print(f"{label:44s} {sps / 1e3:7.2f} k steps/s   in sequence A {a:5.2f} B {b:5.2f} us   back to back A {bb_a:5.2f} us", flush=True)
    ctx.close()


run(plain, 1e-5, f"{cfg} without constraints")
for mode in ("0", "1"):
    os.environ["VVHIP_SHAKE_MODE"] = mode
    for tol in (1e-5, 1e-7, 1e-3, 1e30):
        run(spec, tol, f"{cfg} + HBonds, mode {mode}, tolerance {tol:g}")
