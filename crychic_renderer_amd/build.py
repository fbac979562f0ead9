"""Builds libcrychic_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m crychic_renderer_amd.build [--force]

-ffp-contract=off is part of the numerical contract: the kernels' arithmetic is defined without
implicit fused multiply-add so that results are bit-identical to the CPU oracle.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcrychic_hip.so")
SOURCES = ["kernels.hip", "raster.hip", "api.cpp", "comm.cpp", "host_constants.cpp", "host_geometry.cpp", "host_textures.cpp"]
HEADERS = ["devmath.hpp", "gamma_pow.inc", "ssao_core.hpp", "blur_tiles.hpp", "light_core.hpp", "raster_core.hpp", "kernels.hpp", "internal.hpp"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
         # the SLP vectoriser packs scalar fp32 pairs into v_pk_* plus the v_mov that pairs their registers: measured 5 % slower
         # lighting / blur (profiles/r01_e_relaxed_math_probe.json); the SSAO tap loop is packed by hand where it pays
         "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "crychic_hip.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    # One builder at a time (several ranks of one node may import the package together): the others wait on the lock, find
    # the library fresh and return.  The link goes to a temporary name and is renamed into place, so a concurrent
    # dlopen never sees a half-written file.
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():
            return LIB
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        tmp = LIB + ".tmp.%d" % os.getpid()
        cmd = [hipcc] + FLAGS + ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
        for s in SOURCES:
            cmd += ["-x", "hip", os.path.join(CSRC, s)]
        cmd += ["-ldl", "-o", tmp]      # comm.cpp binds RCCL at run time (dlopen): no link-time dependency on librccl
        if verbose:
            print("[crychic build]", " ".join(cmd), flush=True)
        try:
            subprocess.run(cmd, check=True)
            os.replace(tmp, LIB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
