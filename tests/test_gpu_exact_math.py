"""The exact reciprocal / square-root / inverse-length sequences of csrc/devmath.hpp against their definitions, for EVERY
binary32 bit pattern, on the GPU (tools/exact_math_probe.hip: v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 + one correction step vs the
IEEE expansions the compiler emits for `1.0f / x` and `sqrtf(x)`).  The hardware seeds are not reproducible on a CPU, so this
is where the claim "the device path equals the oracle's or_rcp / or_len / or_inv_len" is established; the parity tests then
check the whole kernels."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "exact_math_probe")


def build_probe():
    src = EXE + ".hip"
    if not os.path.exists(EXE) or os.path.getmtime(src) > os.path.getmtime(EXE):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-ffp-contract=off", src, "-o", EXE], check=True)
    return EXE


def test_probe_compiles():
    assert os.path.exists(build_probe())


@pytest.mark.gpu
def test_shipped_sequences_equal_their_definitions_exhaustively():
    r = subprocess.run([build_probe(), "shipped"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = re.findall(r"^(\S.*?)\s+inputs \[([0-9a-f]{8,9}), ([0-9a-f]{8,9})\): (\d+) differ", r.stdout, flags=re.M)
    names = {n.strip() for n, *_ in rows}
    assert {"rcp (devmath.hpp) vs or_rcp", "len_from_sq vs or_len", "inv_len_from_sq vs or_inv_len"} <= names, r.stdout
    covered = 0
    for name, lo, hi, bad in rows:
        assert int(bad) == 0, (name, lo, hi, bad)
        if name.strip() == "rcp (devmath.hpp) vs or_rcp":
            covered += int(hi, 16) - int(lo, 16)
    assert covered == 1 << 32                     # every bit pattern
