"""The one-launch step assumes that its blocks are resident together (in-kernel rendezvous).  Another process's kernels
on the device break that: the blocks' bounded wait runs out (sticky word [2], VVHIP_ERR_RENDEZVOUS) and every step behind the failed one has
worked on incomplete sums.  Round 6: the plan-driven loops repair this themselves -- snapshot at the entry of the run call, restore + two
launches per step + repeat at the synchronisation that finds the word raised (include/vvhip.h: vvhip_recovery_count).  Here: a SECOND PROCESS
(tests/cpp/cu_hog.cpp) occupies most compute units for half a second in the middle of a run; the trajectory must equal the two-launch
trajectory bit for bit, no sticky word may be left, and the plan must have given up the one-launch step.
Reference contract this replaces: one context per device, platforms/cuda/src/CudaVVKernelFactory.cpp:68."""
import importlib
import os
import shutil
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S, H = pkg.integrator, pkg.systems, pkg.vvhip

pytestmark = pytest.mark.gpu


def _hog_binary():
    src = os.path.join(ROOT, "tests", "cpp", "cu_hog.cpp")
    out = os.path.join(ROOT, "lib", "cu_hog")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-o", out, src], check=True, capture_output=True, timeout=300)
    return out


def _context(spec, fused, **tune):
    it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
    it.setMaxDrudeDistance(0.02)
    return I.Context(spec, it, precision="mixed", force_provider="tether", tune={"fused": int(fused), **tune})


def _state(ctx):
    st = ctx.getNHState()
    return ctx.getPosq(), ctx.getPosqCorrection(), ctx.getVelm(), np.array(list(st.ke2) + list(st.vscale))


def _equal_bits(a, b):
    return all(np.array_equal(np.ascontiguousarray(x).view(np.uint8), np.ascontiguousarray(y).view(np.uint8)) for x, y in zip(a, b))


@pytest.fixture(scope="module")
def spec():
    return S.make_config("C3")


def test_a_second_process_on_the_gpu_costs_a_repeat_not_the_run(spec):
    hog = _hog_binary()
    steps, warm = 60000, 400
    ref = _context(spec, fused=False)
    ref.run_graph(warm, 100)
    ref.run_graph(steps, 100)
    want = _state(ref)
    ref.close()

    ctx = _context(spec, fused=True)
    ctx.run_graph(warm, 100)
    ctx.synchronize()
    assert ctx.fused_status()[0] and ctx.recovery_count() == 0
    # (224 blocks of 1024 threads with all of a CU's LDS, launched 0.15 s into the run: smaller ones -- 8 to 128 blocks were tried -- the GPU
    # time-slices against our kernels without breaking their residency, profiles/r06h_hog_probe.txt)
    proc = subprocess.Popen([hog, "224", "0.6", "0.15"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    try:
        assert proc.stdout.readline().split()[0] == "ready"
        proc.stdin.write("go\n"); proc.stdin.flush()
        ctx.run_graph(steps, 100)                      # ~0.6 s of work (the call blocks while the queue is full)
        assert proc.stdout.readline().strip() == "launched"
        ctx.synchronize()                              # finds word [2] raised: restores, repeats with two launches per step -- and returns VVHIP_OK
        assert proc.stdout.readline().strip() == "done"
    finally:
        proc.stdin.close()
        proc.wait(timeout=30)
    disturbed = ctx.recovery_count()
    assert ctx.status_words() == [0, 0, 0, 0]
    got = _state(ctx)
    assert _equal_bits(got, want)                      # (whether or not the intruder got in the way: the trajectory is the two-launch one)
    if disturbed == 0:
        ctx.close()
        pytest.skip("the other process's kernel did not break the rendezvous on this box (the GPU time-sliced it): the repair was not exercised")
    assert disturbed == 1
    assert not ctx.fused_status()[0]                   # pinned to two launches from here on
    # ... and the run simply goes on
    ctx.run_graph(200, 100)
    ctx.synchronize()
    assert ctx.recovery_count() == 1 and ctx.status_words() == [0, 0, 0, 0]
    ctx.close()


def test_without_a_snapshot_the_failure_is_reported_and_the_plan_pinned(spec):
    """"recover" = 0 (or a run call shorter than the snapshot threshold): VVHIP_ERR_RENDEZVOUS as in round 5 -- but the later steps of the graph give
    up at once instead of waiting 0.2 s each, the plan takes two launches from then on, and after vvhip_status_clear + a restored state it continues
    bit for bit like a plan that never tried the one-launch step."""
    hog = _hog_binary()
    ctx = _context(spec, fused=True, recover=0)
    ctx.run_graph(400, 100)
    ctx.synchronize()
    snap = _state(ctx)
    nh = ctx.getNHState()
    proc = subprocess.Popen([hog, "224", "0.6", "0.15"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    try:
        assert proc.stdout.readline().split()[0] == "ready"
        t0 = time.perf_counter()
        proc.stdin.write("go\n"); proc.stdin.flush()
        ctx.run_graph(60000, 100)
        assert proc.stdout.readline().strip() == "launched"
        err = None
        try:
            ctx.synchronize()
        except H.VVHipError as e:
            err = e
        elapsed = time.perf_counter() - t0
        assert proc.stdout.readline().strip() == "done"
    finally:
        proc.stdin.close()
        proc.wait(timeout=30)
    if err is None:
        assert ctx.status_words() == [0, 0, 0, 0]
        ctx.close()
        pytest.skip("the other process's kernel did not break the rendezvous on this box (the GPU time-sliced it): the failure path was not exercised")
    assert err.code == H.ERR_RENDEZVOUS
    assert "two launches" in str(err)
    assert elapsed < 5.0, f"{elapsed:.1f} s: the steps behind the failed one waited for their own time-outs"
    assert ctx.status_words()[2] == 1 and not ctx.fused_status()[0] and ctx.recovery_count() == 0      # (word [1] may be up too: sums of garbage overflow)
    ctx.status_clear()
    ctx.posq.upload(snap[0]); ctx.posq_corr.upload(snap[1]); ctx.velm.upload(snap[2]); ctx.setNHState(nh)
    ctx.run_graph(600, 100)
    ctx.synchronize()
    got = _state(ctx)
    ctx.close()
    ref = _context(spec, fused=False)
    ref.run_graph(400, 100)
    ref.run_graph(600, 100)
    want = _state(ref)
    ref.close()
    assert _equal_bits(got, want)
