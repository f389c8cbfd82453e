"""Long-run sanity of the arithmetic work-item layout: group temperatures over many steps at sizes where it is on by itself
(C3 tiled 2x: in-kernel chain; 8x: stand-alone chain launch), next to the same run with explicit slot words (VVHIP_PERIODIC=0)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
for scale, nsteps in ((2, 40000), (8, 10000)):
    spec = S.make_config("C3", scale=scale)
    out = {}
    for per in ("1", "0"):
        os.environ["VVHIP_PERIODIC"] = per
        it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
        ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
        t0 = time.perf_counter()
        rows = []
        for k in range(4):
            ctx.run_graph(nsteps // 4, 100); ctx.synchronize()
            T = ctx.getGroupTemperatures()
            x = ctx.getPositions()
            d = spec.drude_pairs
            r = np.linalg.norm(x[d[:, 0]] - x[d[:, 1]], axis=1).max()
            rows.append((T[0], T[1], T[2], r, bool(np.isfinite(x).all())))
        out[per] = rows
        print(f"C3x{scale} periodic={ctx.info.periodic_layout}: {nsteps / (time.perf_counter() - t0):.0f} steps/s sustained; " +
              " | ".join(f"T {a:.2f}/{b:.2f}/{c:.3f} K dmax {r:.4f} ok={ok}" for a, b, c, r, ok in rows), flush=True)
        ctx.close()
    # the two layouts sum in different orders: same physics, temperatures agree closely while the trajectories have not yet diverged
    a, b = np.array([r[:3] for r in out["1"]]), np.array([r[:3] for r in out["0"]])
    print(f"  max relative temperature difference between the layouts: {np.abs(a / b - 1).max():.2e}")
