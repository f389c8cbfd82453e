"""Launch shapes (k blocks per CU x t tile waves per block) just above the size that one block per CU holds (1 792 tile waves): steps/s of the
graph-replayed step for every shape pick_launch_shape considers, next to its own choice.  C3 tiled further along z."""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
if os.environ.get("SWEEP_LAYOUT", "bestfit") == "bestfit":
    os.environ["VVHIP_PERIODIC"] = "0"      # the best-fit layout at every size: the shape is the only variable (SWEEP_LAYOUT=auto: the plan's own choice)
def rate(spec, tune, n=6000):
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02 if len(spec.drude_pairs) else 0.0)
    if spec.image_pairs: it.setMirrorLocation(float(spec.box[2]) / 2)
    if os.environ.get("SWEEP_COS", "0") == "1": it.setCosAcceleration(0.02)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether", tune=tune)
    ctx.run_graph(300, 100); ctx.synchronize()
    t0 = time.perf_counter(); ctx.run_graph(n, 100); ctx.synchronize(); t = time.perf_counter() - t0
    w = ctx.info.num_waves; ctx.close()
    return n / t, w
specs = [S.bulk_Im21(cells=tuple(int(x) for x in c.split("x")), hbonds=os.environ.get("SWEEP_HBONDS", "0") == "1")
         for c in os.environ.get("SWEEP_CELLS", "2x2x4,2x2x5,2x3x3,2x3x4").split(",") if c]
specs += [S.make_config(c, hbonds=os.environ.get("SWEEP_HBONDS", "0") == "1") for c in os.environ.get("SWEEP_CONFIGS", "").split(",") if c]
for spec in specs:
    r0, w = rate(spec, {})
    out = []
    for k in (1, 2, 3, 4):
        for t in (1, 2, 3, 4, 5, 6, 7):
            if k * (t + 1) > 16: continue
            r, _ = rate(spec, {"block_threads": 64 * t, "grid_cap_a": 256 * k, "grid_cap_b": 256 * k})
            out.append((r, k, t))
    out.sort(reverse=True)
    print("%d particles, %d waves: own choice %.0f steps/s; best shapes: %s" % (spec.num_atoms, w, r0, ", ".join("k=%d t=%d %.0f" % (k, t, r) for r, k, t in out[:5])))
