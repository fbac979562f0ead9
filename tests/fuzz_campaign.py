#!/usr/bin/env python3
"""One-off wide fuzz run on the GPU box (test infrastructure: it drives the oracle): tests/test_fuzz.py::test_fuzz_device for
seeds [start, start + n).  usage: python tests/fuzz_campaign.py [n=300] [start=5000]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_util, oracle_lib, test_fuzz
from crychic_renderer_amd import _lib

class Built:
    lib = _lib.lib
    check = staticmethod(_lib.check)
    PassConstants = _lib.PassConstants

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
start = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
orc = oracle_lib.load()
bad = 0
for k in range(n):
    seed = start + k - 1000        # test_fuzz_device adds 1000
    try:
        test_fuzz.test_fuzz_device.__wrapped__(Built, orc, seed) if hasattr(test_fuzz.test_fuzz_device, "__wrapped__") else test_fuzz.test_fuzz_device(Built, orc, seed)
    except AssertionError as e:
        bad += 1
        print("seed %d FAILED: %s" % (seed + 1000, str(e)[:300]), flush=True)
    if k % 50 == 49:
        print("%d / %d done, %d failures" % (k + 1, n, bad), flush=True)
print("fuzz campaign: %d cases, %d failures" % (n, bad))
sys.exit(1 if bad else 0)
