"""The host analysis (csrc/vv_host.cpp: reference-shaped tables, wave layout, arithmetic layout with its self-check) under AddressSanitizer and
UndefinedBehaviorSanitizer, CPU build (GPU sanitizers are not available on this pool): tests/cpp/host_sanitize.cpp analyses 2 000 random plans --
repeated molecules with Drude pairs, hydrogen constraints, rigid water, molecules larger than a wave, Langevin / image subsets, defects, shards."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_analysis_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I" + os.path.join(ROOT, "include"),
           "-o", exe, os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp"), os.path.join(ROOT, "openmm-velocityverlet_amd", "csrc", "vv_host.cpp")]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and ("asan" in b.stderr or "ubsan" in b.stderr):
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe, "300"], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    m = re.search(r"HOST SANITIZE OK plans=(\d+) periodic=(\d+)", r.stdout)
    assert m and int(m.group(1)) >= 2000 and int(m.group(2)) > 300, r.stdout
