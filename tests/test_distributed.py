"""N > 1 path: world_size-2 runs over gloo (CPU: the fixed-point exchange protocol; GPU: two ranks sharing the one GPU,
sharded trajectory vs single process).  On the 8-GPU node the same ShardedStepper runs over nccl (= RCCL, xGMI)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _launch(mode, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), WORKER, mode]
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)


def test_fixed_point_exchange_protocol_gloo_world2():
    r = _launch("protocol", 29541)
    assert r.returncode == 0 and "PROTOCOL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_sharded_step_equals_single_process_two_ranks_one_gpu():
    r = _launch("gpu", 29542)
    assert r.returncode == 0 and "GPU DIST OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_mailbox_exchange_two_ranks_one_gpu():
    """The xGMI mailbox path (hipIpc mappings, kernel A publishes / kernel B collects, hipGraph replay) with two processes."""
    r = _launch("mailbox", 29544)
    assert r.returncode == 0 and "MAILBOX OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["mailbox", "graph", "eager", "python"])
def test_nccl_code_path_world1(mode):
    """The exact multi-GPU path of bench.py (process group over nccl = RCCL, accumulator tensor aliasing the plan's
    device memory, all-reduce between kernel A and kernel B on torch's stream) with a single rank."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--dist-mode", mode, "--steps", "300", "--warmup", "30",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["value"] > 100 and line["n_gpus"] == 1
    assert mode in line["config"]["parallelism"], line["config"]["parallelism"]


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu_mailbox():
    """bench.py's own N = 2 logic (handle gather, trial run, agreement between ranks, graph replay) with two processes on GPU 0;
    gloo carries the set-up traffic because RCCL refuses two ranks on one device."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29545", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
           "--steps", "400", "--warmup", "100", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 100
    assert "mailbox" in line["config"]["parallelism"], line["config"]["parallelism"] + r.stderr[-2000:]
