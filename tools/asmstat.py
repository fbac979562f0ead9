#!/usr/bin/env python3
"""Instruction-mix summary of the gfx950 code objects: tools/asmstat.py [kernels.hip]."""
import collections, re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "crychic_renderer_amd/csrc/kernels.hip")
out = "/tmp/asmstat.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-I", ROOT + "/include",
                "-I", ROOT + "/crychic_renderer_amd/csrc", "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
name, ins = None, []
def flush():
    if not name: return
    c = collections.Counter(i for i in ins)
    cats = collections.Counter()
    for k, v in c.items():
        if k.startswith(("v_div", "v_rcp", "v_sqrt", "v_rsq")): cats["div/sqrt"] += v
        elif k.startswith(("global_load", "buffer_load")): cats["vmem_ld"] += v
        elif k.startswith(("global_store", "buffer_store")): cats["vmem_st"] += v
        elif k.startswith("ds_"): cats["lds"] += v
        elif k.startswith("v_"): cats["valu"] += v
        elif k.startswith("s_load"): cats["smem"] += v
        elif k.startswith("s_waitcnt"): cats["waitcnt"] += v
        elif k.startswith("s_"): cats["salu"] += v
        else: cats["other"] += v
    print("%-48s %6d  %s" % (name[:48], len(ins), dict(cats)))
    print("      ", c.most_common(12))
for l in lines:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        flush(); name, ins = m.group(1), []
    elif l.startswith("\t") and name and not l.strip().startswith((".", ";")):
        tok = l.split()[0]
        ins.append(tok)
        if tok == "s_endpgm":
            flush(); name = None
