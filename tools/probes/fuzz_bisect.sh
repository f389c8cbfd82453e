# Which operation kind of `vv_plugin_driver fuzz` makes deferred fusion and stage-by-stage launches part ways?  VV_FUZZ_MASK = allowed kinds.
D=${D:-oracle/_ref/refplugin/vv_plugin_driver}
ARGS=${ARGS:-"1 0 0.02"}
for mask in ${MASKS:-0 1 2 4 8 16 32 63}; do
  for nops in ${NOPS:-40}; do
  VV_FUZZ_MASK=$mask VVHIP_PLUGIN_DEFER=1 $D fuzz /tmp/a.bin $ARGS $nops ${SEED:-1} 0 > /tmp/a.log 2>&1
  VV_FUZZ_MASK=$mask VVHIP_PLUGIN_DEFER=0 $D fuzz /tmp/b.bin $ARGS $nops ${SEED:-1} 0 > /tmp/b.log 2>&1
  python3 - $mask $nops <<'PY'
import sys, numpy as np
def rd(p):
    out, b = [], open(p, "rb").read(); o = 0
    while o < len(b):
        n = int(np.frombuffer(b, np.int64, 1, o)[0]); o += 8
        out.append((n, o)); 
        break
    return b
def arrays(p):
    b = open(p, "rb").read(); o = 0; res = []
    types = [np.float64, np.float64, np.int32, np.int32, np.int32, np.float64, np.float64, np.float64, np.float32, np.float32, np.float64, np.float64]
    for t in types:
        n = int(np.frombuffer(b, np.int64, 1, o)[0]); o += 8
        res.append(np.frombuffer(b, t, n, o)); o += n * np.dtype(t).itemsize
    return res
a, b = arrays("/tmp/a.bin"), arrays("/tmp/b.bin")
dv = np.abs(a[7] - b[7]).max(); dx = np.abs(a[8].astype(np.float64) - b[8]).max()
print("mask %2s nops %3s: max |dv| %.3e  max |dx| %.3e   %s" % (sys.argv[1], sys.argv[2], dv, dx, open("/tmp/a.log").read().split("FUZZ")[-1].strip()))
PY
  done
done
