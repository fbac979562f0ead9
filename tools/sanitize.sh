#!/bin/bash
# CPU sanitizer tier (the reference's analogue is the CRT leak check + D3D12 debug layer, CRYCHIC.cpp:7-9,
# Common/d3dApp.cpp:417-424): the oracle, the host build of the kernel bodies (tests/hostsim: the text the gfx950 kernels
# inline -- where an indexing bug can be caught off-GPU) and the product's host-side C++ are rebuilt with clang
# AddressSanitizer + UndefinedBehaviourSanitizer (+ float-cast-overflow) and the non-GPU tests that drive them run against
# those builds.  Any report aborts the run.  GPU sanitizers are not available on this pool: CPU only.
set -e
cd "$(dirname "$0")/.."
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$RT" ] || { echo "clang ASan runtime not found"; exit 2; }
export CRYCHIC_SANITIZE=1
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
LD_PRELOAD="$RT" python -m pytest -q -x -m "not gpu" -p no:cacheprovider \
    tests/test_oracle_kat.py tests/test_devmath_host.py tests/test_hostsim_parity.py tests/test_fuzz.py tests/test_golden.py \
    tests/test_numpy_restatements.py tests/test_raster_cpu.py tests/test_constants.py tests/test_culling.py tests/test_textures.py \
    tests/test_point_lights.py tests/test_sanitized_host.py "$@"
