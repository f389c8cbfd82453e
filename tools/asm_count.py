#!/usr/bin/env python3
"""Static instruction mix of compiled kernels in openmm-velocityverlet_amd/lib/vv_kernels.s (`make -C openmm-velocityverlet_amd/csrc asm`):
   tools/asm_count.py a:665136 b:236051 ...   (kernel a / b, stage-set number as in the kernel's template argument, precision fd = mixed)"""
import re
import sys

s = open(sys.argv[1] if sys.argv[1].endswith(".s") else "openmm-velocityverlet_amd/lib/vv_kernels.s").read()
for arg in [a for a in sys.argv[1:] if not a.endswith(".s")]:
    k, n = arg.split(":")
    m = re.search(r"^_ZN2vv11vv_kernel_%sIfdLj%sEEE[^\n]*:[^\n]*\n(.*?)\n\s+s_endpgm" % (k, n), s, re.S | re.M)
    if not m:
        print(arg, "not found")
        continue
    ins = [l.split()[0] for l in m.group(1).splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    vgpr = re.search(r"_ZN2vv11vv_kernel_%sIfdLj%sEEE.*?\.vgpr_count:\s+(\d+)" % (k, n), s, re.S)
    print(arg, "static:", len(ins), "VALU", sum(i.startswith("v_") for i in ins), "f64", sum("f64" in i for i in ins), "SALU", sum(i.startswith("s_") for i in ins),
          "LDS", sum(i.startswith("ds_") for i in ins), "vmem", sum(i.startswith(("global_", "buffer_", "flat_", "scratch_")) for i in ins),
          "scratch", sum(i.startswith("scratch_") for i in ins))
