"""cos(2 pi z / Lz) of the cos-acceleration stages: csrc/vv_layout.h: cos_short_range replaces the device library's cosine (a third of
its instructions).  The function uses only correctly rounded operations, so the host build tested here returns the device's bits;
tests/test_gpu_kernels.py / test_gpu_steps.py hold the kernels that use it to the reference goldens."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None or " fma " not in open("/proc/cpuinfo").read().replace("\n", " "), reason="needs g++ and an FMA-capable CPU")
def test_short_range_cosine_is_within_one_ulp(tmp_path):
    exe = str(tmp_path / "cos_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-I", os.path.join(ROOT, "openmm-velocityverlet_amd", "csrc"),
                    "-o", exe, os.path.join(ROOT, "tests", "cpp", "cos_check.cpp"), "-lquadmath"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    f = dict(zip(r.stdout.split()[::2], r.stdout.split()[1::2]))
    assert int(f["samples"]) >= (1 << 22) and float(f["worst_ulp"]) <= 0.8, r.stdout
    # the fallback region (|x| > 1024, or within 2^-36 of a multiple of pi/2) is hit by the adversarial half only
    assert 0 < int(f["fallbacks"]) < int(f["samples"]) // 2, r.stdout
