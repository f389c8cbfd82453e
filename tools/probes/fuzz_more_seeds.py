"""The adapter fuzz of tests/test_cpp_plugin.py over MORE seeds than the committed tests hold (a one-off sweep): deferred fusion against stage-by-stage
launches for every configuration of the tests, seeds FIRST..LAST, optionally with every kernel compiled at run time (host stalls).
usage: fuzz_more_seeds.py FIRST LAST [VAR=value ...]"""
import os, sys, pathlib, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_cpp_plugin as T

first, last = int(sys.argv[1]), int(sys.argv[2])
extra = dict(kv.split("=", 1) for kv in sys.argv[3:])
bad = 0
n = 0
with tempfile.TemporaryDirectory() as d:
    for middle, cons, cos in [(1, 0, 0.0), (1, 0, 0.02), (0, 0, 0.0), (0, 0, 0.02), (1, 3, 0.0), (0, 3, 0.0)]:
        for seed in range(first, last + 1):
            for hand in ((0, 1) if middle and cos == 0 and cons == 0 else (0,)):
                a = T._fuzz(T.REF_DRIVER if not hand else T.DRIVER, pathlib.Path(d), middle, cons, cos, 40, seed, hand, 1, **extra)
                b = T._fuzz(T.REF_DRIVER if not hand else T.DRIVER, pathlib.Path(d), middle, cons, cos, 40, seed, hand, 0, **extra)
                n += 1
                for k in (7, 8, 9):
                    x, y = a[0][k], b[0][k]
                    if cos == 0:
                        ok = np.array_equal(x.view(np.uint8), y.view(np.uint8))
                    else:
                        ok = np.abs(x.astype(np.float64) - y.astype(np.float64)).max() <= 1e-12 * max(1.0, np.abs(y).max())
                    if not ok:
                        bad += 1
                        print("DIFFERS", middle, cons, cos, seed, hand, k, np.abs(x.astype(np.float64) - y.astype(np.float64)).max(), flush=True)
                        break
print(f"{n} sequences, {bad} differing", extra)
