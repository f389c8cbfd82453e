"""One-launch step against the two-launch step on ONE box, in rotation (round 5, TUNING_LOG section 13): steps/s of 20 000-step graph
runs with the force provider, and without it (integrator alone), per BASELINE configuration.
usage: python tools/probes/fused_ab.py [configs, e.g. C3,C4,C5,C2,C1,C3hb] [rotations] [steps]      (CLASSIC=1: the classic scheme, two thermostat
applications per step: two launches against four)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
EXTRA = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("FUSED_TUNE", "").split(",") if kv)}      # e.g. FUSED_TUNE=fused_poll_delay=6 (one-launch contexts only)
configs = (sys.argv[1] if len(sys.argv) > 1 else "C3,C4,C5,C2,C1").split(",")
rot = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
CLASSIC = os.environ.get("CLASSIC", "0") == "1"


def make(cfg, fused, provider=True):
    hb = cfg.endswith("hb")
    base = cfg[:-2] if hb else cfg
    cos = 0.02 if base == "C4" else 0.0
    spec = S.make_config("C3" if base == "C4" else base, hbonds=hb)
    T, dt, maxd = (300.0, 0.002, 0.0) if base == "C2" else ((333.0, 0.001, 0.0) if base == "C1" else (333.0, 0.001, 0.02))
    it = I.VVIntegrator(T, 10, 1.0, 40, dt)
    # (integrator alone = free flight: the hard wall is checked as ever but moved out of reach, or its rare hit path runs for most pairs after ~150 steps)
    it.setMaxDrudeDistance(maxd if provider or maxd == 0 else 1.0e3)
    it.setCosAcceleration(cos)
    it.setUseMiddleScheme(not CLASSIC)
    if base == "C5":
        lz = float(spec.box[2])
        it.setMirrorLocation(lz / 2)
        it.setElectricField(2.0 / lz * 2 * 1.602176634e-22)
    return I.Context(spec, it, precision="mixed", force_provider="tether" if provider else "static", tune={"fused": int(fused), **(EXTRA if fused else {})}), it


def rate(ctx, n):
    ctx.run_graph(200, 100); ctx.synchronize()
    t0 = time.perf_counter()
    ctx.run_graph(n, 100); ctx.synchronize()
    return n / (time.perf_counter() - t0)


for cfg in (configs if __name__ == "__main__" else []):
    ctxs = {f: make(cfg, f) for f in (True, False)}
    alone = {f: make(cfg, f, provider=False) for f in (True, False)}      # zero forces throughout: the integrator's launches alone
    print(f"{cfg}: {ctxs[True][0].system.num_atoms} particles, {ctxs[True][0].info.num_waves} waves; one launch active: {ctxs[True][0].fused_status()[0]}{' (classic scheme)' if CLASSIC else ''}", flush=True)
    for r in range(rot):
        row = []
        for f in (True, False):
            ctx = ctxs[f][0]
            row.append((rate(ctx, steps), rate(alone[f][0], steps)))
        print(f"  rotation {r}: one launch {row[0][0]:9.0f} steps/s (integrator alone {row[0][1]:9.0f}) | two launches {row[1][0]:9.0f} ({row[1][1]:9.0f}) | gain {100 * (row[0][0] / row[1][0] - 1):+.1f} % ({100 * (row[0][1] / row[1][1] - 1):+.1f} %)", flush=True)
    for f in (True, False):
        assert ctxs[f][0].status_words() == [0, 0, 0, 0], ctxs[f][0].status_words()
        ctxs[f][0].close()
        alone[f][0].close()
