"""The one-launch step of the small configurations: fewer, larger blocks make the rendezvous cheaper (fewer words to poll, more waves to share ten
rows) while a tile's own latency does not care which CU it sits on.  steps/s per tile waves per block, same box, against the plan's own choice.
usage: python tools/probes/small_shape.py [configs: C1,C2,C5,C4/8,C3/8,C4/4]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S, D = pkg.integrator, pkg.systems, pkg.distributed
configs = (sys.argv[1] if len(sys.argv) > 1 else "C1,C2,C5,C4/8,C3/8,C4/4").split(",")


def make(cfg, tune):
    base, _, nr = cfg.partition("/")
    spec = S.make_config("C3" if base == "C4" else base)
    T, dt, maxd = (300.0, 0.002, 0.0) if base == "C2" else ((333.0, 0.001, 0.0) if base == "C1" else (333.0, 0.001, 0.02))
    it = I.VVIntegrator(T, 10, 1.0, 40, dt)
    it.setMaxDrudeDistance(maxd)
    it.setCosAcceleration(0.02 if base == "C4" else 0.0)
    if base == "C5":
        lz = float(spec.box[2]); it.setMirrorLocation(lz / 2); it.setElectricField(2.0 / lz * 2 * 1.602176634e-22)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether", shard=D.shard_bounds(spec, int(nr))[0] if nr else None, tune=tune)
    if nr:
        ctx.mailbox_connect(ctx.mailbox_create(1, 0))
    return ctx


def rate(ctx, n=20000):
    ctx.run_graph(400, 100); ctx.synchronize()
    t0 = time.perf_counter(); ctx.run_graph(n, 100); ctx.synchronize()
    return n / (time.perf_counter() - t0)


for cfg in configs:
    ctx = make(cfg, {})
    nw = ctx.info.num_waves
    out = [("own choice", rate(ctx), ctx.fused_status()[0])]
    ctx.close()
    for t in (1, 2, 3, 4, 5, 6, 7):
        if (nw + t - 1) // t > 256:
            continue
        ctx = make(cfg, {"block_threads": 64 * t, "grid_cap_a": 256, "grid_cap_b": 256})
        out.append((f"{t} tile waves x {(nw + t - 1) // t} blocks", rate(ctx), ctx.fused_status()[0]))
        ctx.close()
    print(f"{cfg}: {nw} waves: " + "; ".join(f"{k}: {v / 1000:.1f} k{'' if a else ' (two launches)'}" for k, v, a in out), flush=True)
