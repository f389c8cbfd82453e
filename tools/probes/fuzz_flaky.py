"""The deferred-fusion fuzz case "classic scheme + cos perturbation" (tests/test_cpp_plugin.py) N times under extra environment variables:
counts the runs whose fused and staged trajectories differ by more than rounding.  usage: fuzz_flaky.py N [VAR=value ...]"""
import os, sys, pathlib, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_cpp_plugin as T

n = int(sys.argv[1])
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    os.environ[k] = v
bad = 0
worst = 0.0
with tempfile.TemporaryDirectory() as d:
    for i in range(n):
        a = T._fuzz(T.REF_DRIVER, pathlib.Path(d), 0, 0, 0.02, 40, 6, 0, 1)[0]
        b = T._fuzz(T.REF_DRIVER, pathlib.Path(d), 0, 0, 0.02, 40, 6, 0, 0)[0]
        e = max(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max() for k in (7, 8, 9))
        worst = max(worst, e)
        bad += e > 1e-9
print(" ".join(sys.argv[2:]), "-> differing runs:", bad, "of", n, "worst", worst)
