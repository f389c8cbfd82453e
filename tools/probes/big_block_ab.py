"""The one-launch step in blocks of 8-15 tile waves (1 793 .. 3 840 tiles) against the two-launch step of the plan's own shape, same box in rotation.
Round 5 (profiles/r05m_big_block_ab.txt; the kernel variant -- launch bounds 1 024, 128 VGPRs, no spills -- is in the history of this file's commit,
not in the tree): 148 k particles 77.7 against 77.7 k steps/s, 166 k 74.9 / 72.8 k, 185 k 70.3 / 69.9 k, 222 k 63.4 / 66.5 k -- three or four tile waves
per SIMD in front of ONE rendezvous serialise what two launches of two blocks per CU overlap; not taken.  The probe needs the test hook "fused_big".
usage: python tools/probes/big_block_ab.py [cells, e.g. 2x2x4,2x3x3,2x2x5,2x3x4] [water molecules, e.g. 45000]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
cells = [c for c in (sys.argv[1] if len(sys.argv) > 1 else "2x2x4,2x3x3,2x2x5,2x3x4").split(",") if c]


def make(spec, big, maxd):
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
    it.setMaxDrudeDistance(maxd)
    return I.Context(spec, it, precision="mixed", force_provider="tether", tune={"fused_big": int(big)})


def rate(ctx, n=10000):
    ctx.run_graph(400, 100); ctx.synchronize()
    t0 = time.perf_counter(); ctx.run_graph(n, 100); ctx.synchronize()
    return n / (time.perf_counter() - t0)


specs = [(c, S.bulk_Im21(cells=tuple(int(x) for x in c.split("x"))), 0.02) for c in cells]
if len(sys.argv) > 2:
    specs.append((f"water {sys.argv[2]}", S.spce_water(int(sys.argv[2]), seed=5), 0.0))
for name, spec, maxd in specs:
    ctxs = {b: make(spec, b, maxd) for b in (True, False)}
    print(f"{name}: {spec.num_atoms} particles, {ctxs[True].info.num_waves} tile waves; one launch active: {ctxs[True].fused_status()[0]} / {ctxs[False].fused_status()[0]}", flush=True)
    for r in range(3):
        one, two = rate(ctxs[True]), rate(ctxs[False])
        print(f"  rotation {r}: one launch {one:9.0f} steps/s | two launches {two:9.0f} | {100 * (one / two - 1):+.1f} %", flush=True)
    for c in ctxs.values():
        assert c.status_words() == [0, 0, 0, 0], c.status_words()
        c.close()
