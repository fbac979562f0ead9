#!/usr/bin/env python3
"""Per-section instruction budget of the lighting pass: compiles tools/light_budget.hip (each section of light_core.hpp's
light_pixel as a kernel of its own) for gfx950 and prints, per section, the static instruction mix -- all basic blocks, and the
straight-line path a wavefront of the benchmark frame takes (the wave-uniform fast paths: every s_cbranch that guards a slow
path is followed on its fast side; `--paths` prints the blocks).  The `frame` kernel (three loads, one store) is subtracted.
    python tools/light_budget.py [> profiles/r04_light_isa_budget.txt]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "light_budget.hip")
OUT = "/tmp/light_budget.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-I", ROOT + "/include",
                "-I", ROOT + "/crychic_renderer_amd/csrc", "-S", "--cuda-device-only", SRC, "-o", OUT], check=True, stderr=subprocess.DEVNULL)


def cat(op):
    if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log")): return "trans"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith(("global_load", "buffer_load")): return "vmem_ld"
    if op.startswith(("global_store", "buffer_store")): return "vmem_st"
    if op.startswith("ds_"): return "lds"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_load"): return "smem"
    if op.startswith(("s_waitcnt", "s_nop")): return "wait/nop"
    if op.startswith("s_"): return "salu"
    return "other"


kernels = collections.OrderedDict()
name = None
for l in open(OUT):
    m = re.match(r"^(sec_\w+):", l)
    if m:
        name = m.group(1); kernels[name] = []; continue
    if name and l.startswith("\t") and not l.strip().startswith((".", ";")):
        tok = l.split()[0]
        kernels[name].append(tok)
    elif name and re.match(r"^\.LBB\d+_\d+:", l):
        kernels[name].append(l.strip())
    elif name and l.startswith(".Lfunc_end"):
        name = None

rows = {}
for k, ins in kernels.items():
    c = collections.Counter(cat(i) for i in ins if not i.startswith(".LBB"))
    rows[k] = c
base = rows["sec_frame"]
print("# tools/light_budget.py: static instruction mix per section of light_pixel<true, NoPointLights, false> (gfx950, the product's flags),")
print("# all basic blocks of each section kernel, minus the `frame` kernel (index arithmetic, three 16-byte loads, one 16-byte store).")
print("# valu = one-lane VALU, valu_pk = v_pk_*_f32 (two results per instruction), trans = v_rcp / v_sqrt / v_rsq.")
print("%-14s %6s %8s %6s %8s %8s %6s %6s" % ("section", "valu", "valu_pk", "trans", "vmem_ld", "salu", "smem", "VALU total"))
tot = collections.Counter()
for k, c in rows.items():
    if k == "sec_frame": continue
    d = {x: c[x] - base[x] for x in ("valu", "valu_pk", "trans", "vmem_ld", "salu", "smem")}
    v = d["valu"] + d["valu_pk"] + d["trans"]
    print("%-14s %6d %8d %6d %8d %8d %6d %6d" % (k[4:], d["valu"], d["valu_pk"], d["trans"], d["vmem_ld"], d["salu"], d["smem"], v))
