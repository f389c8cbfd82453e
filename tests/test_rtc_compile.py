"""CPU: the run-time compilation path (csrc/vv_rtc.cpp) is complete without a GPU -- hipRTC compiles kernels A and B for gfx950 from the
sources embedded in the library, with the library's options, for stage sets outside the compiled list, all three precision modes and
chain lengths other than the default; the device code is free of host headers.  What the compiled objects compute is the GPU suite's
business (tests/test_gpu_rtc.py: bit-identical to the ahead-of-time kernels)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "openmm-velocityverlet_amd", "csrc")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    subprocess.run(["make", "-s", "-C", CSRC, "vv_rtc_sources.inc"], check=True)
    exe = str(tmp_path_factory.mktemp("rtc") / "rtc_compile_check")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-I/opt/rocm/include", "-o", exe, os.path.join(ROOT, "tests", "cpp", "rtc_compile_check.cpp"),
                    os.path.join(CSRC, "vv_rtc.cpp"), "-ldl"], check=True)
    return exe


# kernel, precision (0 single, 1 mixed, 2 double), stage bits (csrc/vv_args.hpp), chain links
@pytest.mark.parametrize("kind,prec,flags,links", [
    ("A", 1, 0x20 | 0x400 | 0x80000, 3),                       # the headline stage set (kick + sums, velocities kept): also in the compiled list
    ("A", 0, 0x20 | 0x400 | 0x4 | 0x8 | 0x10 | 0x200 | 0x2000 | 0x20000 | 0x10000, 3),   # Langevin + field + cos moments + constraints: not in it
    ("B", 1, 0x1 | 0x10 | 0x200 | 0x800 | 0x10000 | 0x20000, 2),       # two-link thermostat wave
    ("B", 2, 0x1 | 0x2 | 0x1000 | 0x8000 | 0x80 | 0x200 | 0x400 | 0x800 | 0x2000 | 0x100000, 4),     # classic first half with everything, four links
])
def test_hiprtc_compiles_stage_sets_for_gfx950_without_a_gpu(checker, kind, prec, flags, links):
    r = subprocess.run([checker, kind, str(prec), str(flags), str(links)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout[-3000:] + r.stderr[-2000:]
    assert f"vv_kernel_{kind.lower()}" in r.stdout and f"Lj{flags}E" in r.stdout      # the instantiation asked for, by its mangled name
