#!/usr/bin/env python3
"""bench.py -- MD steps/s (ns/day) of the integrator hot path on the 100k-atom Drude ionic-liquid box.

  python bench.py --gpus N --steps K --warmup W      (N > 1: one rank per GPU -- under torch.distributed.run, or plain: the
                                                      command then starts `python -m torch.distributed.run ... bench.py` itself)

A "step" is one VVIntegrator step of the whole box: synthetic force provider (plays OpenMM's calcForcesAndEnergy,
IN the timed region) + the fused middle-scheme integrator path (TGNH thermostat, Drude hard wall).  State is
resident in HBM before the timed region starts.  Rank 0 prints ONE JSON line; see DESIGN.md "Measurement".
"""
import argparse
import importlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic bytes per particle per launch come from the plan (vvhip_algorithmic_bytes; SURVEY.md §8d's accounting).  Mixed precision:
# 94 (A) + 134 (B) = 228 B/atom/step is SURVEY's two-pass floor (kernel A writes the kicked velocities back); 62 + 158 = 220 where A
# keeps them in registers and kernel B repeats the kick from velm + force (the two-launch step of rounds 2-4); 158 for the ONE-launch
# step of round 5 (vv_kernel_b<.., SFA>: velm, force and position read once, velm and position written once, 6 bytes of index).
HBM_PEAK_GBS = 8000.0      # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured float4 copy)
SURVEY_TWO_PASS_FLOOR = {"single": 132, "mixed": 228, "double": 228}      # SURVEY.md §8d (bytes / particle / step; the same for every round)


def widen_hard_wall(ctx):
    """Free flight (zero forces, no provider kernel) lets every Drude particle drift away from its core: at 1 K relative temperature it is 0.02 nm
    away -- at the hard wall -- within ~150 steps, and from then on the wall's rare hit path (K/middle.cu:133-218) runs for most pairs in every
    step, which no step with forces does.  The integrator-alone clocks therefore keep the wall's CHECK (same stage set, same fall-through code as
    the real steps) and move the wall itself out of reach; restore_hard_wall puts it back."""
    it = ctx.integrator
    old = it.getMaxDrudeDistance()
    if old > 0:
        it.setMaxDrudeDistance(1.0e3)
    return old


def restore_hard_wall(ctx, old):
    if old > 0:
        ctx.synchronize()
        ctx.integrator.setMaxDrudeDistance(old)


def kernel_times(ctx, reps, batches, one_launch=None):
    """Average launch duration [ms] of kernel A and kernel B with the fused step's stage bits, two clocks:
      in sequence   every launch of `reps` eager steps (force provider -> A -> B, enqueued from C) timed by the dispatch's own begin /
                    end timestamps (hipExtLaunchKernel start / stop events: nothing is added to the stream; the timestamps rocprofv3's
                    kernel trace reports) -- the kernel in its real place in the step, behind its predecessor's dirty lines.  This is
                    the clock of roofline.frac at EVERY size; profiles/r03*_kernel_stats*.csv (rocprofv3 of the same command) is the
                    cross-check (kernel B: within 1 %);
      back to back  two HIP events around `reps` launches of the same kernel, median of `batches`: kinder to a kernel than its place
                    in the step (it finds its own output in the cache), reported beside the other for comparison only."""
    import statistics
    import numpy as np
    # (`one_launch`: what ALL ranks of a sharded run agreed on -- the two branches launch different numbers of steps, and ranks that disagree
    # would leave the mailbox's sequence numbers out of step)
    if ctx.fused_status()[0] if one_launch is None else one_launch:
        # The one-launch step: ONE integrator kernel per step (an instance of vv_kernel_b that also runs kernel A's stages).  Its clock here is a
        # graph replay of the integrator ALONE (forces resident and zeroed: thermostatted free flight, no provider kernel in the loop): one kernel
        # per step back to back, so 1 / rate is the kernel from its predecessor's end to its own end -- what rocprofv3 reports for a kernel in a
        # serialised replay.  Per-launch start / stop events are NOT used for this kernel: launches that carry them start their blocks so unevenly
        # that the in-kernel rendezvous measures the events (11.3 us against 7.5 us in rocprofv3's trace of the replayed step, C3; the self-tuning
        # wait climbs from 6 to its cap meanwhile: tools/probes/fused_eager_clock.py).
        prov = ctx.force_provider
        ctx.synchronize()
        # (the physical state is saved and put back, whatever happens in between: the free flight scrambles it, and later blocks reuse the context)
        snap = (ctx.getPosq(), ctx.getPosqCorrection(), ctx.getVelm(), ctx.getNHState(), ctx.getForce())
        wall = None
        n = max(200, 20 * reps)
        ts = []
        try:
            ctx.force.upload(np.zeros(3 * ctx.padded, dtype=np.int64))
            ctx.force_provider = "static"
            wall = widen_hard_wall(ctx)
            ctx.run_graph(200, 100); ctx.synchronize()
            for _ in range(max(3, batches)):
                t0 = time.perf_counter(); ctx.run_graph(n, 100); ctx.synchronize(); ts.append((time.perf_counter() - t0) / n)
        finally:
            ctx.force_provider = prov
            if wall is not None:
                restore_hard_wall(ctx, wall)
            ctx.synchronize()
            ctx.posq.upload(snap[0]); ctx.posq_corr.upload(snap[1]); ctx.velm.upload(snap[2]); ctx.setNHState(snap[3]); ctx.force.upload(snap[4])
        return {"A": None, "B": 1e3 * statistics.median(ts), "A_back_to_back": None, "B_back_to_back": None, "one_launch": True,
                "how": "one-launch step: 1 / (steps per second) of a graph replay of the integrator alone (%d steps, median of %d; forces resident and zero, no provider "
                       "kernel, hard wall checked but out of reach of the free flight) = the kernel from its predecessor's end to its own end; launches with start / stop events disturb its in-kernel rendezvous "
                       "and are not used" % (n, max(3, batches))}
    ctx.run_eager(8)
    ctx.timing(4 * reps + 16)
    ctx.run_eager(reps)
    r = ctx.timing_read()
    ctx.timing(0)
    seq_a = r["ms_a"] / max(r["launches"][0], 1)
    seq_b = r["ms_b"] / max(r["launches"][1], 1)
    bb_a = statistics.median(ctx.time_kernel(0, reps) for _ in range(batches))
    bb_b = statistics.median(ctx.time_kernel(1, reps) for _ in range(batches))
    return {"A": seq_a, "B": seq_b, "A_back_to_back": bb_a, "B_back_to_back": bb_b, "one_launch": False,
            "how": "in sequence: dispatch timestamps (hipExtLaunchKernel start/stop events) of every launch of %d eager steps; "
                   "back to back: two HIP events around %d launches of the same kernel, median of %d" % (reps, reps, batches)}


# What a first 1 -> 8 run should show ("the curve to expect"): a rank's own share of the step measured on ONE GPU with this
# round's kernels (profiles/r06h_shard_step.txt, C4: r06w_shard_step_C4.txt; large box: the size ladder) + the exchange as ASSUMED there -- 2.0 us for the xGMI mailbox's
# one remote flight, 20 us for an RCCL all-reduce between two launches (neither measured in the build environment).  us per step.
PREDICTED_RANK_STEP_US = {"C3": {1: 9.55, 2: 8.68, 4: 8.01, 8: 7.64}, "C4": {1: 11.29, 2: 10.73, 4: 10.01, 8: 9.68},
                          "C3x80": {1: 455.0, 2: 217.0, 4: 111.0, 8: 59.0}}
PREDICTED_EXCHANGE_US = {"mailbox": 2.0, "graph": 20.0, "eager": 20.0, "python": 60.0}


def predicted_rate(series, world, exchange):
    """steps/s the model of DESIGN.md section 6 expects for `series` on `world` GPUs with that exchange mechanism, or None outside the table."""
    step = PREDICTED_RANK_STEP_US.get(series, {}).get(world)
    if step is None:
        return None
    if world > 1:
        if exchange != "mailbox":
            step = PREDICTED_RANK_STEP_US[series][world] * 1.2      # (two launches per step: the one-launch step needs the mailbox)
        step += PREDICTED_EXCHANGE_US.get(exchange, 20.0)
    return 1e6 / step


def kill_group(child):
    """End a child started with start_new_session=True together with everything it started, and reap it."""
    import signal
    for sig in (signal.SIGTERM, signal.SIGKILL):
        try:
            os.killpg(child.pid, sig)
        except (ProcessLookupError, PermissionError):
            break
        try:
            child.wait(timeout=10)
            break
        except Exception:                                        # noqa: BLE001
            continue


def launch_ranks(n, argv):
    """`python3 bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, the way the driver's documented
    command does (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`),
    as a CHILD process -- this process has not imported torch or touched the GPU and never will (an exec from a process that has
    initialised the GPU takes the box down; a child does not) -- pass its stdout / stderr through and return its exit code."""
    import socket, subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "bench.py")] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL and the mailbox's hipIpc handles need on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.stderr.write("bench.py: --gpus %d without a launcher: starting %s\n" % (n, " ".join(cmd[1:9])))
    child = subprocess.Popen(cmd, env=env, stdin=subprocess.DEVNULL, start_new_session=True)
    try:
        return child.wait()
    except BaseException:                                        # Ctrl-C / SIGTERM of the parent: take the ranks along
        kill_group(child)
        raise


def rocprof_child(args, extra, seconds=240):
    """Kernel durations of this workload as rocprofv3 reports them, measured live: a child process
        rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py <same workload> --child
    (graph replay, no secondary blocks) started BEFORE this process touches the GPU; its kernel_stats.csv is parsed.  This is the very
    command behind profiles/r*_rocprofv3_kernel_stats_*.csv, so roofline.frac can be recomputed from profiles/ with the same tool.
    In a serialised graph replay rocprofv3's duration of a kernel runs from its predecessor's end to its own end
    (profiles/r03a_trace_timeline_graph_vs_eager.txt).  Returns {kernel name: {"calls", "avg_ns"}} or a string saying why not."""
    import csv, glob, re, shutil, subprocess, tempfile
    if os.environ.get("VVHIP_BENCH_CHILD"):
        return "child"
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.gpus != 1:
        return "multi-GPU run"
    if shutil.which("rocprofv3") is None:
        return "rocprofv3 not on PATH"
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD") or os.environ.get("ROCP_TOOL_LIBRARIES"):
        return "this process already runs under a profiler"
    tmp = tempfile.mkdtemp(prefix="vvbench_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp, "-o", "run", "--", sys.executable, os.path.join(ROOT, "bench.py"),
               "--child", "--precision", args.precision, "--forces", args.forces, "--steps-per-graph", str(args.steps_per_graph)] + extra
        env = dict(os.environ, VVHIP_BENCH_CHILD="1", TMPDIR="/tmp")
        # own session + output to a file: on a time-out the WHOLE group goes (rocprofv3 is a wrapper; its python grandchild would
        # otherwise keep the pipes open and share the GPU with the timed region of this process)
        with open(os.path.join(tmp, "child.log"), "wb") as log:
            child = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = child.wait(timeout=seconds)
            except subprocess.TimeoutExpired:
                kill_group(child)
                return "rocprofv3 child timed out after %d s (its process group was killed)" % seconds
        files = glob.glob(os.path.join(tmp, "**", "*kernel_stats.csv"), recursive=True)
        if not files:
            return "rocprofv3 child wrote no kernel_stats.csv (exit code %d)" % rc
        out = {}
        for row in csv.DictReader(open(files[0])):
            m = re.match(r"void vv::(vv_kernel_\w+)<([^>]*)>", row["Name"])
            if m:
                out[f"{m.group(1)}<{m.group(2)}>"] = {"calls": int(row["Calls"]), "avg_ns": round(float(row["AverageNs"]), 1)}
        return out or "no vv kernels in the child's kernel_stats.csv"
    except Exception as e:                                       # noqa: BLE001 -- a cross-check must never break the bench line
        return f"rocprofv3 child failed ({type(e).__name__}: {e})"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def openmm_cpu_baseline(spec, cfg, dt, seconds):
    """BASELINE.json's named (non-target) baseline: OpenMM's own CPU platform with its built-in NoseHooverIntegrator (no Drude pairs) /
    DrudeNoseHooverIntegrator on this host's cores, same particles, same synthetic tether + Drude-spring forces.  Returns a dict, or a
    string saying why it could not run (OpenMM is not part of this image; the block is written against the OpenMM >= 8.1 Python API
    and has never been executed here)."""
    try:
        import openmm as mm
        from openmm import unit as u
    except Exception as e:                                       # noqa: BLE001
        return f"not installed (import openmm: {type(e).__name__}: {e}); the port above is the only CPU figure of this run"
    try:
        import numpy as np
        system = mm.System()
        for m in spec.masses:
            system.addParticle(float(m))
        lx, ly, lz = (float(b) for b in spec.box)
        system.setDefaultPeriodicBoxVectors(mm.Vec3(lx, 0, 0), mm.Vec3(0, ly, 0), mm.Vec3(0, 0, lz))
        tether = mm.CustomExternalForce("0.5*k*((x-x0)^2+(y-y0)^2+(z-z0)^2)")
        tether.addGlobalParameter("k", 1000.0)
        for name in ("x0", "y0", "z0"):
            tether.addPerParticleParameter(name)
        pos = np.asarray(spec.positions, dtype=np.float64)
        for i in range(spec.num_atoms):
            if spec.masses[i] != 0:
                tether.addParticle(i, [float(c) for c in pos[i]])
        system.addForce(tether)
        pairs = np.asarray(spec.drude_pairs).reshape(-1, 2)
        if len(pairs):
            drude = mm.DrudeForce()
            for d, par in pairs:                                 # k_D = q^2 / (4 pi eps0 alpha): charge and polarisability chosen to give 209 200 kJ/mol/nm^2
                drude.addParticle(int(d), int(par), -1, -1, -1, 1.0, 138.935456 / 209200.0, 0.0, 0.0)
            system.addForce(drude)
            integ = mm.DrudeNoseHooverIntegrator(333.0 * u.kelvin, 10.0 / u.picosecond, 1.0 * u.kelvin, 40.0 / u.picosecond, dt * u.picosecond)
            integ.setMaxDrudeDistance(0.02)
            kind = "DrudeNoseHooverIntegrator"
        else:
            integ = mm.NoseHooverIntegrator((300.0 if cfg == "C2" else 333.0) * u.kelvin, 10.0 / u.picosecond, dt * u.picosecond)
            kind = "NoseHooverIntegrator"
        ctx = mm.Context(system, integ, mm.Platform.getPlatformByName("CPU"))
        ctx.setPositions(pos)
        ctx.setVelocitiesToTemperature(333.0)
        integ.step(5)
        t0 = time.perf_counter(); integ.step(10); per = (time.perf_counter() - t0) / 10
        n = max(10, min(5000, int(seconds / max(per, 1e-6))))
        t0 = time.perf_counter(); integ.step(n); el = time.perf_counter() - t0
        return {"value": round(n / el, 2), "unit": "steps/s", "integrator": kind, "platform": "CPU", "threads": os.cpu_count(),
                "sample": f"{n} steps ({el:.1f} s) incl. OpenMM's own force evaluation of the tether + Drude-spring terms", "openmm_version": mm.version.version}
    except Exception as e:                                       # noqa: BLE001
        return f"installed, but the CPU-platform run failed ({type(e).__name__}: {e})"


def traffic_record(path, cfg, precision):
    """HBM bytes per launch from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the builder, NOT counters
    of this run -- the JSON says so in roofline.traffic_source) if it is for this configuration and precision."""
    try:
        rec = json.load(open(path))
    except Exception:
        return None, None
    if rec.get("config") != cfg or rec.get("precision") != precision:
        return None, None
    src = {"file": os.path.relpath(path, ROOT), "round": rec.get("round"), "commit": rec.get("commit"),
           "what": "separate rocprofv3 --pmc passes on the builder's GPU box, not counters of this run"}
    return rec, src


def rocprof_reference(key, algo, n_local):
    """Cross-check from the committed rocprofv3 --kernel-trace --stats summary of the same workload (profiles/kernel_stats_latest.json,
    written by tools/stamp_profiles.py): average duration of the most-launched variant of kernel A / kernel B, and the fractions they
    give.  In a graph replay rocprofv3's duration of a kernel runs from its predecessor's end to its own end (profiles/r03a_trace_timeline_*)."""
    try:
        st = json.load(open(os.path.join(ROOT, "profiles", "kernel_stats_latest.json")))[key]
    except Exception:
        return None
    out = {"file": st.get("file"), "round": st.get("round"), "commit": st.get("commit"), "avg_launch_us": {}, "frac": {}}
    for k in "AB":
        cand = {n: v for n, v in st["kernels"].items() if n.startswith(f"vv_kernel_{k.lower()}<")}
        if not cand or not algo.get(k):
            continue
        name = max(cand, key=lambda n: cand[n]["calls"])
        us = cand[name]["avg_ns"] * 1e-3
        out["avg_launch_us"][k] = round(us, 3)
        out["frac"][k] = round(algo[k] * n_local / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
    return out


def roofline_block(ctx, n_local, times, rec=None, src=None, note=None, ref_key=None, live=None):
    """roofline objects of the integrator kernel(s) of context `ctx` -- kernel A and kernel B of the two-launch step, or the ONE kernel of
    the one-launch step -- and the dominant one.  Clock of `achieved` / `frac` / `avg_launch_us`, the first that exists (`clock` says which):
      rocprofv3_child          rocprofv3 --kernel-trace --stats of a child run of this very workload, started by this bench run before it touched
                               the GPU (`live` = its kernel_stats; average duration of the most-launched variant of each kernel: the kernel in its
                               place in the replayed step, from its predecessor's end to its own end);
      rocprofv3_committed_csv  the committed summary of the same command (profiles/kernel_stats_latest.json <- profiles/r*_rocprofv3_kernel_stats_*.csv);
      self_clocked             no profiler figure for this workload: dispatch timestamps (two launches) / the integrator-alone replay (one launch).
    The self-clocked figures are always printed beside it (`avg_launch_us_dispatch_timestamps`; for the one-launch step
    `frac_integrator_alone_replay` -- a MODIFIED workload: forces resident and zero, no provider kernel, hard wall out of reach -- round 5
    reported that one as `frac`; the review asked for the figure profiles/ reproduces)."""
    algo = dict(zip("AB", ctx.algorithmic_bytes()))
    kernels = [k for k in "AB" if algo[k] > 0 and times.get(k) is not None]
    one = bool(times.get("one_launch"))
    prof_ms = {}
    if isinstance(live, dict):
        for k in kernels:
            cand = {n: v for n, v in live.items() if n.startswith(f"vv_kernel_{k.lower()}<")}
            if cand:
                prof_ms[k] = cand[max(cand, key=lambda n: cand[n]["calls"])]["avg_ns"] * 1e-6
    ref = rocprof_reference(ref_key, algo, n_local) if ref_key else None
    csv_ms = {k: ref["avg_launch_us"][k] * 1e-3 for k in kernels if ref and k in ref.get("avg_launch_us", {})}
    if len(prof_ms) == len(kernels):
        clock, source = prof_ms, "rocprofv3_child"
    elif len(csv_ms) == len(kernels):
        clock, source = csv_ms, "rocprofv3_committed_csv"
    else:
        clock, source = times, "self_clocked"
    us = lambda v: None if v is None else round(v * 1e3, 3)
    per = {}
    for k in kernels:
        by = algo[k] * n_local
        ach = by / (clock[k] * 1e-3) / 1e9
        per[k] = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                  "traffic": rec.get(f"hbm_bytes_per_launch_{k}") if rec else None, "algorithmic_bytes_per_launch": by,
                  "avg_launch_us": us(clock[k]), "avg_launch_us_dispatch_timestamps": us(times[k]),
                  "avg_launch_us_back_to_back": us(times.get(k + "_back_to_back"))}
    dom = max(kernels, key=lambda k: clock[k])
    out = dict(per[dom])
    out["clock"] = source
    if "B" in prof_ms:
        out["avg_launch_us_rocprofv3_child"] = {k: us(prof_ms[k]) for k in prof_ms}
        out["frac_rocprofv3_child"] = round(algo[dom] * n_local / (prof_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dom in prof_ms else None
    if one:
        out["avg_launch_us_integrator_alone_replay"] = us(times["B"])
        out["frac_integrator_alone_replay"] = round(algo["B"] * n_local / (times["B"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out["integrator_alone_replay_caveat"] = ("a MODIFIED workload, comparison only: graph replay of the integrator alone -- forces resident and zero, no provider kernel in "
                                                 "the loop, hard wall checked but moved out of reach of the free flight; 1 / rate = the kernel from its predecessor's end to its own end")
    name = {k: f"vv_kernel_{k.lower()}" for k in kernels}
    if one:
        name["B"] = "vv_kernel_b<.., SFA> (the one-launch step: kernel A's stages, in-kernel rendezvous, kernel B's stages)"
    how = {"rocprofv3_child": "avg_launch_us / achieved / frac: rocprofv3 --kernel-trace --stats of a child run of this workload (graph replay), started by this bench "
                              "run before it touched the GPU -- the kernel from its predecessor's end to its own end; ",
           "rocprofv3_committed_csv": "avg_launch_us / achieved / frac: the committed rocprofv3 --kernel-trace --stats summary of this workload (%s; no live child run: %s); "
                                      % ((ref or {}).get("file"), live if isinstance(live, str) else "not requested"),
           "self_clocked": "avg_launch_us / achieved / frac: self-clocked (no rocprofv3 figure for this workload: %s) -- %s; "
                           % (live if isinstance(live, str) else "no child run, no committed summary", "the integrator-alone replay" if one else "dispatch timestamps")}[source]
    out.update({"kernel": name[dom], "traffic_source": src, "algorithmic_bytes_per_particle": algo,
                "survey_two_pass_floor_bytes_per_particle": SURVEY_TWO_PASS_FLOOR.get(ctx.precision),
                "launches_per_step": 1 if one else 2,
                "avg_launch_us": {k: per[k]["avg_launch_us"] for k in kernels},
                "avg_launch_us_dispatch_timestamps": {k: per[k]["avg_launch_us_dispatch_timestamps"] for k in kernels},
                "avg_launch_us_back_to_back": {k: per[k]["avg_launch_us_back_to_back"] for k in kernels},
                "launch_timing": how + times["how"],
                "per_kernel": {f"vv_kernel_{k.lower()}": per[k] for k in kernels}})
    if ref_key:
        out["rocprofv3_cross_check"] = ref
    if note:
        out["note"] = note
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4", "C5", "C3x2", "C3x3", "C3x5", "C3x8", "C3x80"])
    ap.add_argument("--precision", default="mixed", choices=["single", "mixed", "double"])
    ap.add_argument("--forces", default="tether", choices=["tether", "static"])
    ap.add_argument("--steps-per-graph", type=int, default=100)
    ap.add_argument("--eager", action="store_true", help="no hipGraph replay (host-launched every step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-GPU code path (process group + exchange) even at N=1")
    ap.add_argument("--dist-mode", default="auto", choices=["auto", "mailbox", "graph", "eager", "python"],
                    help="N>1: mailbox = totals stored into the peers over xGMI by the kernels themselves (default when its trial run passes); "
                         "graph = RCCL all-reduce captured in the hipGraph; eager = RCCL issued from C per step; python = torch.distributed per step")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (gloo + --share-device: tests on a one-GPU box)")
    ap.add_argument("--share-device", action="store_true", help="all ranks use GPU 0 (tests only)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--large-n", default="C3x80", choices=["C3x8", "C3x80", "none"],
                    help="secondary block (N = 1, config C3): the same kernels on the tiled box, where HBM bandwidth is the bound")
    ap.add_argument("--synthetic", action="store_true", help="procedural look-alike systems instead of the reference's example models (C3 / C4 / C5)")
    ap.add_argument("--child", action="store_true", help="(internal) the run rocprofv3 wraps: headline measurement only, no secondary blocks")
    ap.add_argument("--headline-only", action="store_true", help="headline measurement only, no secondary blocks (what the counter passes of tools/profile_round.sh wrap: one variant of each kernel per run)")
    ap.add_argument("--no-rocprof", action="store_true", help="do not spawn the rocprofv3 child runs; roofline.frac then comes from the dispatch-timestamp clock")
    ap.add_argument("--no-other-configs", action="store_true", help="skip config.other_configs (C4, C5, C2, C1 under the driver's flags)")
    ap.add_argument("--hbonds", action="store_true", help="constraints solved in-kernel: HBonds (SHAKE) for the ionic liquids, rigid water (SETTLE) for C2; not the headline workload")
    args = ap.parse_args()

    # ---- N > 1 without a launcher: become the launcher (child process; nothing below runs in this process)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # ---- rocprofv3 child runs, BEFORE this process initialises the GPU (kernel durations as the profiler reports them, measured live)
    prof = {}
    if not args.child and not args.headline_only and not args.no_rocprof and args.gpus == 1 and not args.eager and not args.force_dist:
        base = ["--config", args.config] + (["--synthetic"] if args.synthetic else [])
        short = ["--steps", "200", "--warmup", "40"] if args.config.startswith("C3x") else ["--steps", "4000", "--warmup", "400"]
        # (bounded: a child normally takes 3-15 s; if the first one cannot run, the others are not tried -- same reason, same answer)
        prof["main"] = rocprof_child(args, base + short + (["--hbonds"] if args.hbonds else []), seconds=150 if args.config.startswith("C3x") else 90)
        if isinstance(prof["main"], dict):
            if not args.hbonds and args.config in ("C2", "C3", "C4", "C5"):
                prof["hbonds"] = rocprof_child(args, base + short + ["--hbonds"], seconds=90)
            if args.config == "C3" and args.large_n != "none" and not args.hbonds:
                prof["large_n"] = rocprof_child(args, ["--config", args.large_n, "--steps", "100", "--warmup", "20"] + (["--synthetic"] if args.synthetic else []), seconds=150)
        else:
            prof["hbonds"] = prof["large_n"] = prof["main"]

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path is HIP only (no CPU fallback)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    pkg = importlib.import_module("openmm-velocityverlet_amd")
    I, S, D = pkg.integrator, pkg.systems, pkg.distributed

    # ---- workload (BASELINE.json configs[2] / [3]; C3xK = the same cell tiled K times along z)
    cfg = args.config
    # C3 / C4 / C5 come from the reference's own example models (tests/golden/topo_*.npz <- examples/models/{bulk_Im21,edl_Im21});
    # --synthetic selects the procedural look-alikes of round 1.  --hbonds: the constraints the example scripts put on the System
    # (HBonds -> in-kernel SHAKE; rigidWater for C2 -> SETTLE)
    if cfg.startswith("C3x"):
        spec = S.make_config("C3", float(cfg[3:]), hbonds=args.hbonds, synthetic=args.synthetic)
    else:
        spec = S.make_config(cfg, hbonds=args.hbonds, synthetic=args.synthetic)
    dt = 0.002 if cfg == "C2" else 0.001
    it = I.VVIntegrator(300.0 if cfg == "C2" else 333.0, 10.0, 1.0, 40.0, dt)
    if cfg not in ("C1", "C2"):
        it.setMaxDrudeDistance(0.02)
    if cfg == "C4":
        it.setCosAcceleration(0.02)
    if cfg == "C5":                                   # examples/run-edl.py:82-100: mirror at Lz/2, field V/Lz*2 with V = 2 V
        lz = float(spec.box[2])
        it.setMirrorLocation(lz / 2)
        it.setElectricField(2.0 / lz * 2 * 1.602176634e-22)
    bounds = D.shard_bounds(spec, world)
    # N > 1: three ways to exchange the thermostat sums.  Unless --dist-mode names one, the bench sets up what it
    # can, times a short run of each candidate on all ranks and keeps the faster one:
    #   mailbox  kernel B's head stores the rank's int64 totals into every peer's box over xGMI (hipIpc mappings) and collects
    #            the peers': no collective launch, the sharded step stays two launches inside the hipGraph;
    #   eager    in-core RCCL: ncclAllReduce enqueued from C between kernel A and kernel B of every step (graph = captured; opt-in,
    #            multi-rank capture cannot be tried in the build environment);
    #   python   torch.distributed all-reduce per step (last resort).
    dist_mode = None
    ctx = None
    candidates = {}
    exchange_log = {}          # why each exchange mechanism was (not) usable: goes into config.exchange

    def agree(ok):
        flag = torch.tensor([1 if ok else 0], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def slowest(seconds):
        t = torch.tensor([seconds], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def fresh_context(stream=None):
        it._context = None
        return I.Context(spec, it, precision=args.precision, force_provider=args.forces, shard=bounds[rank], device=local_rank, stream=stream)

    def timed(fn, n):
        """us per step of fn(n) on the slowest rank, or None if it failed on any rank (then every rank gets None)."""
        ok, dt_ = True, 0.0
        try:
            ctx.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            fn(n)
            ctx.synchronize()
            dt_ = time.perf_counter() - t0
        except Exception as e:                                   # noqa: BLE001
            sys.stderr.write(f"[rank {rank}] candidate run failed ({e})\n")
            ok = False
        if not agree(ok):
            return None
        return slowest(dt_) / n * 1e6

    def setup_rccl():
        try:
            idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(ctx.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, src=0)
            ctx.comm_init(bytes(idt.cpu().numpy().tobytes()), world, rank)
            ok = True
        except Exception as e:                                   # noqa: BLE001
            sys.stderr.write(f"[rank {rank}] in-core RCCL unavailable ({e})\n")
            exchange_log["rccl"] = f"unavailable on rank {rank}: {e}"
            ok = False
        ok = agree(ok)
        exchange_log.setdefault("rccl", "communicator up on every rank" if ok else "unavailable on another rank")
        return ok

    def setup_mailbox():
        ok = True
        if world > 1 and not args.share_device:          # the peers' boxes are mapped through hipIpc and written by device stores: needs P2P access
            try:
                H = pkg.vvhip
                denied = []
                for peer in range(world):
                    can = H.C.c_int32(0)
                    if peer != local_rank and (H.lib.vvhip_peer_access(local_rank, peer, H.C.byref(can)) != 0 or not can.value):
                        denied.append(peer)
                exchange_log["peer_access"] = "all peers" if not denied else f"rank {rank}: no peer access to device(s) {denied}"
                if not agree(not denied):
                    exchange_log["mailbox"] = "not tried: hipDeviceCanAccessPeer denies a pair of devices"
                    return False
            except Exception as e:                               # noqa: BLE001
                exchange_log["peer_access"] = f"check failed ({e})"
        try:
            mine = torch.frombuffer(bytearray(ctx.mailbox_create(world, rank)), dtype=torch.uint8).cuda()
            allh = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allh, mine)
            ctx.mailbox_connect(b"".join(bytes(h.cpu().numpy().tobytes()) for h in allh))
        except Exception as e:                                   # noqa: BLE001
            sys.stderr.write(f"[rank {rank}] mailbox exchange unavailable ({e})\n")
            exchange_log["mailbox"] = f"set-up failed on rank {rank}: {e}"
            ok = False
        if not agree(ok):
            exchange_log.setdefault("mailbox", "set-up failed on another rank")
            return False
        try:                                                     # trial: eager steps, then a replayed graph; all ranks must end up
            it.step(4)                                           # with the very same thermostat bits and no wait may have run out
            if not ctx.mailbox_status()[1]:
                ctx.run_graph(2 * args.steps_per_graph, args.steps_per_graph)
            ctx.synchronize()
            active, timed_out = ctx.mailbox_status()
            st = ctx.getNHState()
            mine = torch.tensor(list(st.ke2) + list(st.vscale), dtype=torch.float64, device="cuda")
            every = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            ok = active and not timed_out and all(torch.equal(e.view(torch.int64), mine.view(torch.int64)) for e in every) \
                and bool(torch.isfinite(mine).all())
            if not ok:
                exchange_log["mailbox"] = ("trial run: a wait on the peers' words ran out" if timed_out else
                                           "trial run: the ranks' thermostat bits differ" if active else "trial run: mailbox not active")
        except Exception as e:                                   # noqa: BLE001
            sys.stderr.write(f"[rank {rank}] mailbox trial failed ({e})\n")
            exchange_log["mailbox"] = f"trial run failed on rank {rank}: {e}"
            ok = False
        ok = agree(ok)
        exchange_log.setdefault("mailbox", "trial run passed on every rank" if ok else "trial run failed on another rank")
        return ok

    if use_dist and args.dist_mode != "python":
        want = args.dist_mode
        ctx = fresh_context()
        have_rccl = setup_rccl() if want in ("auto", "eager", "graph") else False
        if have_rccl and want == "auto":
            t_rccl = timed(ctx.run_eager, 100)                   # warm-up, and a first sign of life of the collective
            t_rccl = timed(ctx.run_eager, 1000) if t_rccl is not None else None
            if t_rccl is None:
                have_rccl = False
            else:
                candidates["rccl_eager_us_per_step"] = round(t_rccl, 2)
        have_mb = False
        if want in ("auto", "mailbox"):
            have_mb = setup_mailbox()
            if have_mb and want == "auto":
                t_mb = timed(lambda n: ctx.run_graph(n, args.steps_per_graph), 3 * args.steps_per_graph)
                if t_mb is None:
                    have_mb = False
                else:
                    candidates["mailbox_graph_us_per_step"] = round(t_mb, 2)
            if not have_mb:                                       # a failed trial leaves a void state behind: start over
                if rank == 0:
                    sys.stderr.write("mailbox exchange not usable here\n")
                try:
                    ctx.close()
                except Exception:                                # noqa: BLE001
                    pass
                ctx = fresh_context()
                have_rccl = setup_rccl() if want in ("auto", "eager", "graph") else False
        # (the eager loop's short trial flatters it: over thousands of steps its 3-4 host launches per step become the bound, 14-15 us per
        # step where the trial showed 12; a replayed graph has no such tail -- so the mailbox keeps the run unless it is clearly slower)
        if have_mb and (not have_rccl or want == "mailbox" or
                        candidates.get("mailbox_graph_us_per_step", 0.0) <= 1.15 * candidates.get("rccl_eager_us_per_step", float("inf"))):
            dist_mode = "mailbox"
        elif have_rccl:
            if ctx.mailbox_status()[0]:                          # set up but not chosen: the plan must stop using it
                ctx.mailbox_destroy()
            dist_mode = "graph" if want == "graph" else "eager"
            if dist_mode == "graph":                             # every rank must agree that capture works
                try:
                    ctx.run_graph(2, 2)
                    ctx.synchronize()
                    ok = True
                except Exception as e:                           # noqa: BLE001
                    sys.stderr.write(f"[rank {rank}] graph capture with RCCL failed ({e}); using the C eager loop\n")
                    ok = False
                if not agree(ok):
                    dist_mode = "eager"
        else:
            ctx.close()
            ctx = None
    stepper = None
    if ctx is None:
        stream = torch.cuda.current_stream().cuda_stream if use_dist else None
        ctx = fresh_context(stream)
        if use_dist:
            stepper = D.ShardedStepper(ctx)
            dist_mode = "python"
    # ---- first contact (N > 1): what every rank sees of its device, its peers and the chosen exchange, on stderr BEFORE anything is timed; ranks
    # that disagree end the run with that report and a non-zero exit code instead of a hang or a wrong number
    if use_dist:
        H_ = pkg.vvhip
        peer = None
        if world > 1 and not args.share_device:
            peer = []
            for other in range(world):
                can = H_.C.c_int32(0)
                peer.append(other == local_rank or (H_.lib.vvhip_peer_access(local_rank, other, H_.C.byref(can)) == 0 and bool(can.value)))
        try:
            rccl_n = ctx.comm_count()
        except Exception:                                        # noqa: BLE001
            rccl_n = 0
        import socket
        rec = {"host": socket.gethostname(), "device": local_rank, "device_name": torch.cuda.get_device_name(local_rank), "peer_access": peer, "rccl_ranks": rccl_n, "exchange": dist_mode,
               "mailbox_trial": exchange_log.get("mailbox"), "world": world, "share_device": bool(args.share_device)}
        ok_fc, text_fc = D.first_contact_report(rec)
        if rank == 0 or not ok_fc:
            sys.stderr.write(f"[rank {rank}] first-contact report\n{text_fc}\n")
        exchange_log["first_contact"] = "all ranks agree" if ok_fc else text_fc.splitlines()[-1]
        if not ok_fc:
            raise SystemExit(3)
    use_graph = (not use_dist and not args.eager) or dist_mode in ("graph", "mailbox")
    # The graph that is replayed holds min(--steps-per-graph, K rounded down to even) steps, so that K = 20 (the driver's flags)
    # is one replay of a 20-step graph and K = 20000 is 200 replays of a 100-step graph; a remainder (K odd, or not a multiple)
    # is enqueued step by step from C.  The graph is captured / instantiated / uploaded BEFORE the timed region
    # (vvhip_graph_prepare after the warm-up, for the thermostat parity the warm-up ends on): nothing but hipGraphLaunch and the
    # tail's launches happens between the two fences.
    spg = max(2, min(args.steps_per_graph, args.steps - args.steps % 2)) if args.steps >= 2 else 2
    spg += spg % 2

    def run(n):
        if stepper is not None:
            stepper.step(n)
        elif use_graph:
            ctx.run_graph(n, spg)
        elif dist_mode == "eager":
            ctx.run_eager(n)
        else:
            it.step(n)

    def fence():
        ctx.synchronize()                 # raises if a mailbox wait timed out / an accumulator overflowed (sticky status word)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def healthy_fence():
        """fence() on every rank; False on ALL ranks if any rank's plan reported a failure (the run is void then)."""
        ok = True
        try:
            ctx.synchronize()
        except pkg.vvhip.VVHipError as e:
            sys.stderr.write(f"[rank {rank}] {e}\n")
            ok = False
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
            ok = agree(ok)
        return ok

    run(args.warmup)
    if use_graph and args.steps >= spg:
        ctx.graph_prepare(spg)
    if not healthy_fence():
        raise SystemExit("bench: the warm-up run is void (mailbox time-out or accumulator overflow on some rank)")

    def timed_k_steps():
        # barrier + synchronize on both sides of the K steps; the clock is read between the closing synchronize and the closing barrier:
        # every rank times its own K steps from the common start, the MAX over ranks is the job's time -- the collective that implements the
        # barrier (50-200 us, as much as the 20 steps of the driver's flags take) is not part of the steps
        fence()
        t0 = time.perf_counter()
        run(args.steps)
        ctx.synchronize()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # EXACTLY K steps between two fences (barrier + synchronize), slowest rank.  A K-step region shorter than 50 ms (K = 20 is
    # ~0.3 ms) is dominated by the jitter of one host synchronisation, so it is measured up to 25 times back to back (the simulation
    # simply runs on) and the MEDIAN K-step time is reported; `timed_repeats` and the spread are in the line.
    samples = [timed_k_steps()]
    if samples[0] < 0.05:
        reps = int(min(24, max(4, 0.1 / max(samples[0], 1e-6))))        # the same on every rank: samples[0] is the max over ranks
        for _ in range(reps):
            samples.append(timed_k_steps())
    import statistics
    elapsed = statistics.median(samples)
    steps_per_s = args.steps / elapsed
    if not healthy_fence():
        raise SystemExit("bench: the timed run is void (mailbox time-out or accumulator overflow on some rank)")
    if use_dist and stepper is None:        # every rank must hold the same thermostat bits after the run (exact integer exchange)
        st = ctx.getNHState()
        mine = torch.tensor(list(st.ke2) + list(st.vscale), dtype=torch.float64, device="cuda")
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        if not all(torch.equal(e.view(torch.int64), mine.view(torch.int64)) for e in every):
            raise SystemExit("bench: the ranks' thermostat states differ after the timed run")
    n_replays, n_tail = (args.steps // spg, args.steps % spg) if use_graph and args.steps >= spg else (0, args.steps)

    x = ctx.getPositions()
    if not np.isfinite(x).all():
        raise SystemExit("non-finite positions after the timed run")

    out = None
    if rank == 0:
        n = spec.num_atoms
        out = {
            "metric": "MD steps/sec (ns/day), 100k-atom Drude IL box, 1/2/4/8 MI355X", "value": round(steps_per_s, 1),
            "unit": "steps/s", "ns_per_day": round(steps_per_s * dt * 1e3 * 0.0864, 1), "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 6),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": {"mixed": "f64 velocities, f32+f32 positions (OpenMM 'mixed')", "single": "f32", "double": "f64"}[args.precision],
            "data": "synthetic",
            "config": {"workload": f"{cfg}: {spec.name}, {n} particles, {spec.num_molecules} molecules, "
                                   f"{len(spec.drude_pairs)} Drude pairs; TGNH thermostat ({ctx.info.num_temp_groups} groups), middle scheme, "
                                   f"hard wall 0.02 nm, dt {dt * 1e3:g} fs" + (", cos acceleration 0.02 nm/ps^2" if cfg == "C4" else "")
                                   + (f", {len(spec.constraints)} constraints solved in-kernel ({ctx.info.num_shake_clusters} SHAKE clusters, {ctx.info.num_settle_clusters} SETTLE molecules)" if args.hbonds else "")
                                   + (f", {len(spec.particles_ld)} Langevin particles (device Philox normals), {len(spec.image_pairs)} image pairs, E-field" if cfg == "C5" else ""),
                       "force_provider": f"{args.forces} (synthetic, inside the timed region)" if args.forces == "tether" else "static buffer",
                       "launch": (f"per timed region: {n_replays} replay(s) of a {spg}-step hipGraph captured before the timed region"
                                  + (f" + {n_tail} host-launched step(s)" if n_tail else "")) if use_graph and n_replays else "host-launched per step",
                       "timed_repeats": len(samples), "timed_region_ms": {"median": round(1e3 * elapsed, 4), "min": round(1e3 * min(samples), 4), "max": round(1e3 * max(samples), 4)},
                       "parallelism": ("1 GPU" + (f" (distributed code path forced: {dist_mode})" if use_dist else "")) if world == 1 else f"particle shards on molecule boundaries x{world}, int64 totals exchanged per thermostat application ({dist_mode})",
                       "atom_steps_per_s": round(steps_per_s * n, 1)},
        }
        if candidates:
            out["config"]["exchange_candidates"] = candidates
        if use_dist:
            rccl_ranks = 0
            try:
                rccl_ranks = ctx.comm_count()
            except Exception:                                    # noqa: BLE001
                pass
            chosen = {"mailbox": "xGMI mailbox: kernel B's thermostat waves store / poll the int64 totals in the peers' boxes (no collective launch)",
                      "graph": "ncclAllReduce(int64) captured inside the step graph", "eager": "ncclAllReduce(int64) enqueued from C between kernel A and kernel B",
                      "python": "torch.distributed all-reduce per step (last resort)"}.get(dist_mode, str(dist_mode))
            out["config"]["exchange"] = {"chosen": dist_mode, "what": chosen, "rccl_ranks": rccl_ranks, "process_group_ranks": world,
                                         "candidates_us_per_step": candidates, "log": exchange_log}
            pred = predicted_rate(cfg, world, dist_mode) if not args.hbonds and not args.share_device else None
            out["config"]["exchange"]["predicted"] = {
                "steps_per_s": None if pred is None else round(pred, 1), "measured_over_predicted": None if pred is None else round(steps_per_s / pred, 3),
                "model": "DESIGN.md section 6: rank 0's share of the step as measured on one GPU (profiles/r06h_shard_step.txt) + the exchange as assumed "
                         "there (xGMI mailbox 2.0 us, RCCL all-reduce between two launches 20 us; neither ever measured in the build environment)"
                         + ("" if pred is not None else "; no prediction for this run (ranks sharing a device, constraints, or a size outside the table)")}

    def secondary(c, n=None):
        """steps/s of context c from replays of a --steps-per-graph graph: graph prepared and warmed outside the timed region,
        median of 3 timed regions of n steps (secondary figures only; the headline is measured above)."""
        g = args.steps_per_graph + args.steps_per_graph % 2
        n = n or max(args.steps // 4 // g * g, 10 * g)
        c.run_graph(2 * g, g)
        c.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            c.run_graph(n, g)
            c.synchronize()
            ts.append(time.perf_counter() - t0)
        return n / statistics.median(ts)

    def driver_protocol(c, k=20, regions=25):
        """steps/s of context c under the driver's flags (--steps 20 --warmup 5): one replay of a k-step graph per timed region between two
        synchronisations, median of `regions` regions -- the protocol of the headline `value` when K = 20."""
        c.run_graph(k, k)
        c.graph_prepare(k)
        c.synchronize()
        ts = []
        for _ in range(regions):
            t0 = time.perf_counter()
            c.run_graph(k, k)
            c.synchronize()
            ts.append(time.perf_counter() - t0)
        return k / statistics.median(ts)

    # ---- the integrator path alone: forces resident in HBM (static buffer, zeroed: thermostatted free flight -- the same loads, stores
    # and arithmetic as with any other force values, and nothing that can run away), no provider kernel in the loop.  The physical
    # state is saved and put back: the particles leave their tether sites meanwhile.
    if args.child or args.headline_only:          # the run rocprofv3 wraps: the headline loop is all it needs
        ctx.close()
        print(json.dumps(out), flush=True)
        return
    if world == 1 and rank == 0 and not use_dist and args.forces == "tether" and not args.eager:
        snap = (ctx.getPosq(), ctx.getPosqCorrection(), ctx.getVelm(), ctx.getNHState())
        prov = ctx.force_provider
        ctx.force.upload(np.zeros(3 * ctx.padded, dtype=np.int64))
        ctx.force_provider = "static"
        wall = widen_hard_wall(ctx)
        out["config"]["integrator_only_steps_per_s"] = round(secondary(ctx), 1)
        ctx.force_provider = prov
        restore_hard_wall(ctx, wall)
        ctx.synchronize()
        ctx.posq.upload(snap[0]); ctx.posq_corr.upload(snap[1]); ctx.velm.upload(snap[2]); ctx.setNHState(snap[3])

    # ---- the same box with the constraints the reference's example scripts put on it (HBonds: examples/ommhelper/oplspsffile.py:952-955;
    # rigid water for C2), solved inside the fused kernels: a secondary figure, the headline stays the workload BASELINE.json names
    if world == 1 and rank == 0 and not use_dist and args.forces == "tether" and not args.eager and not args.hbonds and cfg in ("C2", "C3", "C4", "C5"):
        spec_c = S.make_config(cfg, hbonds=True, synthetic=args.synthetic)
        it_c = I.VVIntegrator(it.getTemperature(), 10.0, 1.0, 40.0, dt)
        it_c.setMaxDrudeDistance(it.getMaxDrudeDistance())
        it_c.setCosAcceleration(it.getCosAcceleration())
        it_c.setMirrorLocation(it.getMirrorLocation())
        it_c.setElectricField(it.getElectricField())
        ctx_c = I.Context(spec_c, it_c, precision=args.precision, force_provider="tether", device=local_rank)
        sps_c = secondary(ctx_c)
        g_c = args.steps_per_graph + args.steps_per_graph % 2
        blk_c = {"steps_per_s": round(sps_c, 1),
                 "protocol": f"median of 3 timed regions of {max(args.steps // 4 // g_c * g_c, 10 * g_c)} steps each, replays of a {g_c}-step hipGraph prepared outside the region (the long-run protocol; not the driver's flags)",
                 "steps_per_s_driver_flags": round(driver_protocol(ctx_c), 1),
                 "protocol_driver_flags": "one replay of a 20-step hipGraph per timed region, median of 25 regions (what --steps 20 --warmup 5 measures for the headline)",
                 "constraints": int(len(spec_c.constraints)),
                 "shake_clusters": int(ctx_c.info.num_shake_clusters), "settle_molecules": int(ctx_c.info.num_settle_clusters),
                 "solver": "all constraints of a cluster at once: direct solve (velocities), coupled Newton (positions)" if os.environ.get("VVHIP_SHAKE_MODE", "1") != "0" else "Gauss-Seidel sweeps (VVHIP_SHAKE_MODE=0)"}
        # the same two clocks for the constrained stage sets, and the whole step against the roofline
        rec_c, src_c = traffic_record(os.path.join(ROOT, "profiles", "pmc_latest_hbonds.json"), cfg, args.precision)
        t_c = kernel_times(ctx_c, 100, 5)
        blk_c["roofline"] = roofline_block(ctx_c, spec_c.num_atoms, t_c, rec_c, src_c, ref_key=cfg + "_hbonds", live=prof.get("hbonds"))
        ab_c = sum(ctx_c.algorithmic_bytes()) * spec_c.num_atoms
        blk_c["step"] = {"algorithmic_bytes_per_step": ab_c, "achieved": round(ab_c * sps_c / 1e9, 1), "unit": "GB/s", "frac": round(ab_c * sps_c / 1e9 / HBM_PEAK_GBS, 4)}
        out["config"]["with_constraints"] = blk_c
        ctx_c.close()

    # ---- roofline of the dominant kernel, measured live on the plan's stream (kernel_times: in-sequence clock).  At N > 1 every
    # rank runs the same launches (the mailbox exchange inside kernel B needs its peers), rank 0 reports its own shard; PMC traffic
    # is a N = 1 figure taken from the committed summary (traffic_source says which).  Runs after the headline measurement: the
    # back-to-back batches scramble the physical state.
    if (world == 1 and rank == 0 and not use_dist) or (use_dist and stepper is None):
        one_all = bool(ctx.fused_status()[0])
        mixed = False
        if use_dist:          # the two clocks of kernel_times launch different numbers of steps: all ranks take the same one, or none is taken
            all_one, none_one = agree(one_all), agree(not one_all)
            mixed = not (all_one or none_one)
            one_all = all_one
        times = kernel_times(ctx, 100, 5, one_launch=one_all) if not mixed else None
        n_local = bounds[rank][1] - bounds[rank][0]
        rec, src = traffic_record(os.path.join(ROOT, "profiles", "pmc_latest_hbonds.json" if args.hbonds else "pmc_latest.json"), cfg, args.precision) if world == 1 else (None, None)
        if rank == 0 and times is None:
            out["roofline"] = None
            out["config"]["roofline_skipped"] = "some ranks run the one-launch step and some two launches (different shard shapes): the kernels are not clocked"
        if rank == 0 and times is not None:
            out["roofline"] = roofline_block(ctx, n_local, times, rec, src, ref_key=(cfg + ("_hbonds" if args.hbonds else "")) if world == 1 else None, live=prof.get("main"),
                                             note=(("working set (%.0f MB) is L2 / Infinity-Cache resident at this size: the launch is latency and VALU-issue bound "
                                                    "(two waves per SIMD), see config.large_n for the bandwidth-bound regime" % (228e-6 * n_local)) if n_local < 2_000_000 else
                                                   "bandwidth-bound regime (working set far beyond the 256 MB Infinity Cache)")
                                                  + ("" if world == 1 else f"; rank 0's shard of {n_local} particles"))
            ab = sum(ctx.algorithmic_bytes()) * n_local
            # the whole step (force provider + A + B) against the same roofline: algorithmic bytes of the integrator path / step time
            out["step"] = {"algorithmic_bytes_per_step": ab, "achieved": round(ab * steps_per_s / 1e9, 1), "unit": "GB/s",
                           "frac": round(ab * steps_per_s / 1e9 / HBM_PEAK_GBS, 4), "ms_per_step": round(1e3 * elapsed / args.steps, 6),
                           # (the step's own algorithmic bytes fell from 220 to 158 per particle when the two launches became one: the
                           # same step time against SURVEY's constant two-pass floor, for comparison across rounds)
                           "frac_on_survey_two_pass_floor": round(SURVEY_TWO_PASS_FLOOR[args.precision] * n_local * steps_per_s / 1e9 / HBM_PEAK_GBS, 4)}

    # ---- what the launches of this run were: the one-launch step or two launches, the self-tuned rendezvous wait, and whether any launch fell
    # off the compiled / run-time compiled kernels onto the generic one (15-20 % slower) or any run-time compilation failed
    if rank == 0:
        H = pkg.vvhip
        failed, stats = H.C.c_int64(0), (H.C.c_int64 * 3)()
        secs = H.C.c_double(0)
        H.lib.vvhip_rtc_failures(H.C.byref(failed)); H.lib.vvhip_rtc_stats(H.C.byref(stats), H.C.byref(secs))
        one, nfused = ctx.fused_status()
        out["config"]["step_launches"] = {
            "integrator_launches_per_step": 1 if one else 2,
            "what": ("one launch: kernel A's stages, an in-kernel rendezvous of the co-resident blocks, kernel B's stages (vv_kernel_b<.., SFA>)" if one else
                     "two launches: kernel A (kick + sums), kernel B (thermostat + drift)") + "; + the synthetic force provider's launch",
            "one_launch_step_launches": nfused, "rendezvous_wait_units_of_256_clocks": ctx.fused_wait_units() if one else None,
            "generic_kernel_launches": {"A": ctx.generic_launches()[0][0], "B": ctx.generic_launches()[0][1]},
            "rtc": {"kernels_compiled_at_run_time": int(stats[0]), "failed": int(failed.value), "compile_seconds": round(secs.value, 2)}}

    # ---- every BASELINE configuration under the driver's flags (config.other_configs): one replay of a 20-step graph per timed region,
    # median of 25 regions -- the protocol of the headline when K = 20 -- plus the long-run figure and the integrator kernel's duration
    # (dispatch timestamps) against the roofline.  Secondary figures; the headline stays the configuration the metric is quoted on.
    if world == 1 and rank == 0 and not use_dist and cfg == "C3" and args.forces == "tether" and not args.eager and not args.hbonds and not args.no_other_configs:
        others = {}
        for oc in ("C4", "C5", "C2", "C1"):
            try:
                spec_o = S.make_config("C3" if oc == "C4" else oc, synthetic=args.synthetic)
                dt_o = 0.002 if oc == "C2" else 0.001
                it_o = I.VVIntegrator(300.0 if oc == "C2" else 333.0, 10.0, 1.0, 40.0, dt_o)
                if oc not in ("C1", "C2"):
                    it_o.setMaxDrudeDistance(0.02)
                if oc == "C4":
                    it_o.setCosAcceleration(0.02)
                if oc == "C5":
                    lz_o = float(spec_o.box[2])
                    it_o.setMirrorLocation(lz_o / 2)
                    it_o.setElectricField(2.0 / lz_o * 2 * 1.602176634e-22)
                ctx_o = I.Context(spec_o, it_o, precision=args.precision, force_provider="tether", device=local_rank)
                ctx_o.run_graph(600, 100); ctx_o.synchronize()
                sps_drv = driver_protocol(ctx_o)
                sps_long = secondary(ctx_o, 4000)
                t_o = kernel_times(ctx_o, 100, 3)
                rb_o = roofline_block(ctx_o, spec_o.num_atoms, t_o, ref_key=oc)
                ab_o = sum(ctx_o.algorithmic_bytes()) * spec_o.num_atoms
                others[oc] = {"particles": int(spec_o.num_atoms), "steps_per_s_driver_flags": round(sps_drv, 1), "steps_per_s": round(sps_long, 1),
                              "ns_per_day_driver_flags": round(sps_drv * dt_o * 1e3 * 0.0864, 1),
                              "integrator_launches_per_step": rb_o["launches_per_step"], "kernel": rb_o["kernel"],
                              "roofline": {"bound": "hbm", "frac": rb_o["frac"], "achieved": rb_o["achieved"], "unit": "GB/s", "avg_launch_us": rb_o["avg_launch_us"],
                                           "clock": rb_o["clock"], "launch_timing": rb_o["launch_timing"],
                                           "frac_integrator_alone_replay": rb_o.get("frac_integrator_alone_replay"),
                                           "avg_launch_us_integrator_alone_replay": rb_o.get("avg_launch_us_integrator_alone_replay"),
                                           "rocprofv3_cross_check": rb_o.get("rocprofv3_cross_check"),
                                           "algorithmic_bytes_per_particle": rb_o["algorithmic_bytes_per_particle"]},
                              "step": {"algorithmic_bytes_per_step": ab_o, "frac": round(ab_o * sps_long / 1e9 / HBM_PEAK_GBS, 4)},
                              "generic_kernel_launches": sum(ctx_o.generic_launches()[0])}
                ctx_o.close()
            except Exception as e:                                   # noqa: BLE001 -- a secondary block must never break the bench line
                others[oc] = f"skipped: {type(e).__name__}: {e}"
        try:      # the classic scheme (stepVV, VVIntegrator.cpp:295-336) of the headline box: two thermostat applications per step, one launch each
            it_c = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
            it_c.setMaxDrudeDistance(0.02)
            it_c.setUseMiddleScheme(False)
            ctx_c = I.Context(spec, it_c, precision=args.precision, force_provider="tether", device=local_rank)
            ctx_c.run_graph(600, 100); ctx_c.synchronize()
            n0 = ctx_c.fused_status()[1]
            sps_drv = driver_protocol(ctx_c)
            sps_long = secondary(ctx_c, 4000)
            ctx_c.run_eager(10); ctx_c.synchronize()
            others["C3_classic_scheme"] = {"particles": int(spec.num_atoms), "steps_per_s_driver_flags": round(sps_drv, 1), "steps_per_s": round(sps_long, 1),
                                           "integrator_launches_per_step": 2 if ctx_c.fused_status()[1] > n0 else 4,
                                           "generic_kernel_launches": sum(ctx_c.generic_launches()[0])}
            ctx_c.close()
        except Exception as e:                                       # noqa: BLE001
            others["C3_classic_scheme"] = f"skipped: {type(e).__name__}: {e}"
        out["config"]["other_configs"] = {"protocol": "steps_per_s_driver_flags: one replay of a 20-step hipGraph per timed region, median of 25 regions (the headline's protocol "
                                                      "under --steps 20 --warmup 5); steps_per_s: median of 3 regions of 4000 steps (100-step graphs)", **others}

    # ---- the bandwidth-bound regime of the same kernels: the C3 cell tiled 80x along z (8.88 M particles, ~2 GB working set), where
    # the HBM roofline is the real bound.  Secondary block (config.large_n); the headline stays the workload BASELINE.json names.
    if world == 1 and rank == 0 and not use_dist and cfg == "C3" and args.large_n != "none" and not args.eager and not args.hbonds:
        try:
            spec_l = S.make_config("C3", float(args.large_n[3:]), synthetic=args.synthetic)
            it_l = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, dt)
            it_l.setMaxDrudeDistance(0.02)
            ctx_l = I.Context(spec_l, it_l, precision=args.precision, force_provider="tether", device=local_rank)
            sps_l = secondary(ctx_l, 40)
            t_l = kernel_times(ctx_l, 20, 3)
            nl = spec_l.num_atoms
            rec_l, src_l = traffic_record(os.path.join(ROOT, "profiles", f"pmc_latest_{args.large_n}.json"), args.large_n, args.precision)
            rb = roofline_block(ctx_l, nl, t_l, rec_l, src_l, ref_key=args.large_n, live=prof.get("large_n"))
            blk = {"workload": f"{args.large_n}: {nl} particles, {spec_l.num_molecules} molecules (the C3 cell tiled along z)",
                   "steps_per_s": round(sps_l, 1), "atom_steps_per_s": round(sps_l * nl, 1),
                   "roofline": rb["per_kernel"], "launch_timing": rb["launch_timing"], "traffic_source": src_l, "rocprofv3_cross_check": rb.get("rocprofv3_cross_check"), "algorithmic_bytes_per_particle": rb["algorithmic_bytes_per_particle"]}
            ab_l = sum(ctx_l.algorithmic_bytes()) * nl
            blk["step"] = {"algorithmic_bytes_per_step": ab_l, "achieved": round(ab_l * sps_l / 1e9, 1), "unit": "GB/s", "frac": round(ab_l * sps_l / 1e9 / HBM_PEAK_GBS, 4)}
            out["config"]["large_n"] = blk
            ctx_l.close()
            del spec_l
        except Exception as e:                                       # noqa: BLE001 -- a secondary block must never break the bench line
            sys.stderr.write(f"large-N block skipped: {e}\n")

    # ---- CPU baseline: the oracle (our C restatement of the reference path, OpenMP) on this host's cores, bounded sample
    if world == 1 and rank == 0 and not args.no_cpu_baseline and not use_dist:
        from oracle import oracle as O
        ncpu = os.cpu_count() or 1
        p = O.Params(temperature=it.getTemperature(), drude_temperature=1.0, step_size=dt, max_drude_distance=it.getMaxDrudeDistance(),
                     cos_acceleration=it.getCosAcceleration())
        # pick the OpenMP thread count that is actually fastest on this host (a 256-thread team on 111k-particle loops is
        # slower than 16): short calibration, then one bounded sample at the winner
        best = None
        for threads in sorted({1, 4, 8, 16, 32, 64, min(ncpu, 128)}):
            if threads > ncpu:
                continue
            osys = O.OracleSystem(spec, p, args.precision, force_mode=1 if args.forces == "tether" else 0, num_threads=threads)
            osys.step(2)
            t0 = time.perf_counter()
            osys.step(5)
            per = (time.perf_counter() - t0) / 5
            if best is None or per < best[1]:
                best = (threads, per)
        cores, per = best
        osys = O.OracleSystem(spec, p, args.precision, force_mode=1 if args.forces == "tether" else 0, num_threads=cores)
        osys.step(3)
        nsteps = max(10, min(20000, int(args.cpu_seconds / max(per, 1e-6))))
        t0 = time.perf_counter()
        osys.step(nsteps)
        cpu_elapsed = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(nsteps / cpu_elapsed, 2), "unit": "steps/s", "cores": cores, "kind": "port",
                               "sample": f"{nsteps} steps of the same {cfg} workload ({cpu_elapsed:.1f} s), oracle/vv_oracle.c with OpenMP, "
                                         f"fastest of 1..{min(ncpu, 128)} threads on a {ncpu}-CPU host",
                               # BASELINE.json names OpenMM's CPU platform as the baseline: tried here, and said so when it is absent
                               "openmm": openmm_cpu_baseline(spec, cfg, dt, args.cpu_seconds)}
    # ---- still the baseline leg (the only part of this file that touches oracle/): the reference's own kernel sequence on this GPU
    # (oracle/_ref GPU build; present only if built where /root/reference exists), reported inside the cpu_baseline object
    if world == 1 and rank == 0 and not use_dist and not args.no_cpu_baseline and cfg in ("C2", "C3", "C4") and args.precision == "mixed" and args.forces == "tether" and not args.hbonds:
        try:
            from oracle import oracle as O
            if O.have_ref_gpu():
                p = O.Params(temperature=it.getTemperature(), drude_temperature=1.0, step_size=dt, max_drude_distance=it.getMaxDrudeDistance(),
                             cos_acceleration=it.getCosAcceleration())
                ref = O.RefGpuSystem(spec, p)
                ref.step(200); ref.sync()
                nref = 3000
                t0 = time.perf_counter()
                ref.step(nref); ref.sync()
                tref = time.perf_counter() - t0
                ref.close()
                out["cpu_baseline"]["reference_kernels_on_this_gpu"] = {
                    "value": round(nref / tref, 1), "unit": "steps/s",
                    "what": "the reference's unmodified CUDA kernels compiled for gfx950, launched in the reference's order incl. its blocking "
                            "KE download / host chain / upload per step (oracle/ref_gpu_driver.cpp); same workload and force provider"}
        except Exception as e:                                   # noqa: BLE001 -- a baseline must never break the bench line
            sys.stderr.write(f"reference-kernel baseline skipped: {e}\n")
    # ---- N > 1: further series next to the headline, with the headline's exchange mechanism; every stage is agreed between the ranks, a
    # failure anywhere skips the block on all of them (no collective inside a try block):
    #   config.c4_sharded        BASELINE.json's configs[3]: the same box WITH the cos perturbation (the configuration the 1/2/4/8 series is named
    #                            for; 10 exchanged totals per step instead of 3), timed like the headline (K steps, graph replay where the exchange allows);
    #   config.large_n_sharded   the C3 cell tiled along z, the regime in which sharding pays (at 111 000 particles a step is latency bound per rank).
    def sharded_series(label, spec_l, it_l, n_timed, g_l, note):
        blk, ctx_l, ok = {"workload": None}, None, True
        dbg = (lambda m: sys.stderr.write(f"[rank {rank}] {label}: {m}\n")) if os.environ.get("VVHIP_BENCH_DEBUG") else (lambda m: None)
        try:
            bounds_l = D.shard_bounds(spec_l, world)
            ctx_l = I.Context(spec_l, it_l, precision=args.precision, force_provider=args.forces, shard=bounds_l[rank], device=local_rank,
                              stream=torch.cuda.current_stream().cuda_stream if dist_mode == "python" else None)
        except Exception as e:                                   # noqa: BLE001
            sys.stderr.write(f"[rank {rank}] {label}: set-up failed ({e})\n")
            ok = False
        dbg("context built")
        ok = agree(ok)
        mode_l = dist_mode
        if ok and mode_l == "mailbox":
            try:
                mine = torch.frombuffer(bytearray(ctx_l.mailbox_create(world, rank)), dtype=torch.uint8).cuda()
                allh = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(allh, mine)
                ctx_l.mailbox_connect(b"".join(bytes(h.cpu().numpy().tobytes()) for h in allh))
            except Exception as e:                               # noqa: BLE001
                sys.stderr.write(f"[rank {rank}] {label}: mailbox set-up failed ({e})\n")
                ok = False
            ok = agree(ok)
        elif ok and mode_l in ("eager", "graph"):
            try:
                idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
                if rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(ctx_l.comm_unique_id()), dtype=torch.uint8))
                dist.broadcast(idt, src=0)
                ctx_l.comm_init(bytes(idt.cpu().numpy().tobytes()), world, rank)
            except Exception as e:                               # noqa: BLE001
                sys.stderr.write(f"[rank {rank}] {label}: RCCL set-up failed ({e})\n")
                ok = False
            ok = agree(ok)
        if ok:
            try:
                st_l = D.ShardedStepper(ctx_l) if mode_l == "python" else None

                def run_l(n):
                    if st_l is not None:
                        st_l.step(n)
                    elif mode_l in ("mailbox", "graph"):
                        ctx_l.run_graph(n, g_l)
                    else:
                        ctx_l.run_eager(n)
                dbg(f"exchange {mode_l} set up, warm-up")
                run_l(2 * g_l)
                ctx_l.synchronize(); torch.cuda.synchronize()
                if mode_l == "mailbox" and ctx_l.mailbox_status()[1]:
                    raise RuntimeError("a mailbox wait ran out during the warm-up")
                dbg("warm-up done")
            except Exception as e:                               # noqa: BLE001
                sys.stderr.write(f"[rank {rank}] {label}: warm-up failed ({e})\n")
                ok = False
            ok = agree(ok)                                       # (also the barrier in front of the timed run)
            el_l = 0.0
            if ok:
                try:
                    t0 = time.perf_counter()
                    run_l(n_timed)
                    ctx_l.synchronize(); torch.cuda.synchronize()
                    el_l = time.perf_counter() - t0
                except Exception as e:                           # noqa: BLE001
                    sys.stderr.write(f"[rank {rank}] {label}: run failed ({e})\n")
                    ok = False
            ok = agree(ok)
            if ok:
                el_l = slowest(el_l)
                nl = spec_l.num_atoms
                blk = {"workload": f"{note}: {nl} particles, {bounds_l[rank][1] - bounds_l[rank][0]} on rank 0",
                       "steps_per_s": round(n_timed / el_l, 1), "atom_steps_per_s": round(n_timed / el_l * nl, 1), "n_gpus": world, "exchange": mode_l,
                       "steps": n_timed, "launch": f"replays of a {g_l}-step hipGraph" if mode_l in ("mailbox", "graph") else "host-launched per step",
                       "integrator_launches_per_step": 1 if ctx_l.fused_status()[0] else 2, "scaling": "strong"}
                pred_l = predicted_rate(label.split()[0], world, mode_l) if not args.share_device else None
                if pred_l is not None:
                    blk["predicted_steps_per_s"] = round(pred_l, 1)
                    blk["measured_over_predicted"] = round(n_timed / el_l / pred_l, 3)
        if ctx_l is not None:
            try:
                ctx_l.close()
            except Exception:                                    # noqa: BLE001
                pass
        return blk if ok else "skipped (see stderr)"

    if use_dist and cfg == "C3" and not args.hbonds:
        it_4 = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, dt)
        it_4.setMaxDrudeDistance(0.02)
        it_4.setCosAcceleration(0.02)
        k4 = max(spg, args.steps - args.steps % spg)
        blk4 = sharded_series("C4 sharded block", spec, it_4, k4, spg, "C4: the headline box with cos acceleration 0.02 nm/ps^2 (BASELINE.json configs[3])")
        if rank == 0:
            if isinstance(blk4, dict):
                blk4["note"] = "second series of a scaling run: the configuration BASELINE.json names for the 1/2/4/8 curve; compare with config.other_configs.C4 of the N = 1 line"
            out["config"]["c4_sharded"] = blk4
    if use_dist and cfg == "C3" and args.large_n != "none" and not args.hbonds:
        it_l = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, dt)
        it_l.setMaxDrudeDistance(0.02)
        blkl = sharded_series(f"{args.large_n} sharded block", S.make_config("C3", float(args.large_n[3:]), synthetic=args.synthetic), it_l, 40, 10,
                              f"{args.large_n}: the C3 cell tiled along z")
        if rank == 0:
            if isinstance(blkl, dict):
                blkl["note"] = "third series of a scaling run: the regime in which sharding pays; compare with config.large_n of the N = 1 line"
            out["config"]["large_n_sharded"] = blkl
    ctx.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:          # last thing on stdout (librccl prints a banner of its own when the process group comes up)
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)      # librccl's banner sits in the C stdio buffer until exit otherwise
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
