"""Per-step time of ONE rank's share of a sharded run (BASELINE's 1/2/4/8 series is quoted on C4): rank 0's 1/N of the box with the mailbox
exchange set up (its own handle as the only peer: the stage sets and the polling are the sharded run's, the peers' latency is not), one launch
against two, same box in rotation.  usage: python tools/probes/shard_step.py [C4|C3] [ranks, e.g. 8,4,2]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S, D = pkg.integrator, pkg.systems, pkg.distributed
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
ranks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8,4,2").split(",")]
spec = S.make_config("C3" if cfg == "C4" else cfg)


def make(n, fused):
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
    it.setMaxDrudeDistance(0.02)
    it.setCosAcceleration(0.02 if cfg == "C4" else 0.0)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether", shard=D.shard_bounds(spec, n)[0], tune={"fused": int(fused), **({"block_threads": int(os.environ["BLOCK_THREADS"])} if os.environ.get("BLOCK_THREADS") else {})})      # BLOCK_THREADS=64/128/192: tile waves per block
    h = ctx.mailbox_create(1, 0)
    ctx.mailbox_connect(h)
    return ctx


def rate(ctx, n=20000):
    ctx.run_graph(400, 100); ctx.synchronize()
    t0 = time.perf_counter(); ctx.run_graph(n, 100); ctx.synchronize()
    return n / (time.perf_counter() - t0)


for n in ranks:
    ctxs = {f: make(n, f) for f in (True, False)}
    b = D.shard_bounds(spec, n)[0]
    print(f"{cfg} / {n}: rank 0 holds particles {b[0]}..{b[1]} ({ctxs[True].info.num_waves} waves); one launch active: {ctxs[True].fused_status()[0]}", flush=True)
    for r in range(3):
        one, two = rate(ctxs[True]), rate(ctxs[False])
        print(f"  rotation {r}: one launch {one:9.0f} steps/s ({1e6 / one:5.2f} us/step) | two launches {two:9.0f} ({1e6 / two:5.2f} us/step) | {100 * (one / two - 1):+.1f} %", flush=True)
    for c in ctxs.values():
        assert c.status_words() == [0, 0, 0, 0], c.status_words()
        c.close()
