#!/usr/bin/env python3
"""Static instruction mix of one kernel in lib/vv_kernels.s (make -C openmm-velocityverlet_amd/csrc asm):
    tools/isa_mix.py _ZN2vv11vv_kernel_aIfdLj1056EEEvNS_5KArgsE [--top 40]
Counts per class (fp64 VALU, other VALU, DPP moves, SALU, LDS, memory) and the most frequent opcodes.  Static counts: a wave that
skips a branch executes fewer; the SQ_INSTS_* counters (tools/pmc_sq.sh) give the dynamic numbers."""
import collections, re, sys
path = "openmm-velocityverlet_amd/lib/vv_kernels.s"
s = open(path).read()
name = sys.argv[1]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 30
a = s.index(name + ":")
b = s.index("s_endpgm", a)
ins = []
for l in s[a:b].splitlines():
    t = l.strip()
    if not l.startswith("\t") or not t or t[0] in ".;": continue
    ins.append(t)
ops = [t.split()[0] for t in ins]
c = collections.Counter(ops)
g = collections.Counter()
for t in ins:
    k = t.split()[0]
    if k.startswith("v_"):
        if "dpp" in t or "row_" in t or "quad_perm" in t: g["valu_dpp"] += 1
        elif "f64" in k: g["valu_f64"] += 1
        else: g["valu_other"] += 1
    elif k.startswith("s_"): g["salu"] += 1
    elif k.startswith("ds_"): g["lds"] += 1
    else: g["mem"] += 1
print(name, "instructions:", len(ins), dict(g))
for k, v in sorted(c.items(), key=lambda x: -x[1])[:top]: print(f"  {v:5d} {k}")
