#!/usr/bin/env python3
"""Instruction histogram of the innermost loop of a kernel: tools/looplist.py <kernel-substring> [asm file]"""
import collections, re, sys
name = sys.argv[1]; path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/asmstat.s"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(name), l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
hdr = [i for i, l in enumerate(body) if "Inner Loop Header" in l]
if not hdr: print("no loop"); sys.exit()
for h in hdr:
    lab = body[h].split(":")[0].strip()
    last = max(i for i, l in enumerate(body) if re.search(r"s_cbranch\w+\s+" + re.escape(lab) + r"\b", l))
    ins = [l.split()[0] for l in body[h:last + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print(lab, "instructions:", len(ins), "valu:", valu, "vmem:", sum(v for k, v in c.items() if k.startswith(("global_", "buffer_"))),
          "lds:", sum(v for k, v in c.items() if k.startswith("ds_")), "salu:", sum(v for k, v in c.items() if k.startswith("s_")))
    print("  ", c.most_common(40))
