"""Launch shapes of the two-launch step in the bandwidth-bound regime (C3x80, 8.9 M particles): kernel A holds 74 VGPRs = 6 waves per SIMD, so its
2 048 blocks of 4 waves run as one full round of 1 536 blocks and a second round of 512 on a third of the machine.  steps/s for grids that are whole
rounds (k blocks per CU), same box, two rotations.  usage: python tools/probes/large_n_shape.py [scale]"""
import importlib, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 80.0
spec = S.make_config("C3", scale=scale)
nsteps = int(120000 / scale)
shapes = [(256, 2048, 512), (256, 1024, 512), (256, 768, 512), (256, 512, 512)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(x) for x in t.split("x")) for t in os.environ["SHAPES"].split(",")]
res = {s: [] for s in shapes}
for rep in range(int(os.environ.get('ROT', '2'))):
    for s in shapes:
        it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
        ctx = I.Context(spec, it, precision="mixed", force_provider="tether", tune={"block_threads": s[0], "grid_cap_a": s[1], "grid_cap_b": s[2]})
        ctx.run_graph(nsteps // 5, 20); ctx.synchronize()
        t0 = time.perf_counter(); ctx.run_graph(nsteps, 20); ctx.synchronize(); t = time.perf_counter() - t0
        res[s].append(nsteps / t)
        ctx.close()
print(f"C3x{scale:g}: {spec.num_atoms} particles")
for s in shapes:
    print("  %3d threads, kernel A <= %4d blocks, kernel B <= %4d blocks: steps/s %s" % (s[0], s[1], s[2], " ".join("%.1f" % x for x in res[s])), flush=True)
