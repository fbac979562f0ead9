#!/bin/bash
# Builds the stand-alone GPU tools against the in-tree library (run from the repo root; hipcc cross-compiles without a GPU):
#   tools/prof_driver       torch-free frame loop for rocprofv3 (PMC passes, tools/pmc_passes.sh)
#   tools/exact_math_probe  exhaustive check of the exact rcp / sqrt / inverse-length sequences against their definitions
#   tools/valu_rate         VALU issue-rate microbenchmark
set -e
cd "$(dirname "$0")/.."
R=$(pwd)
/opt/rocm/bin/hipcc -O2 -std=c++17 -x c++ -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/prof_driver.cpp \
  -L crychic_renderer_amd -lcrychic_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,"$R/crychic_renderer_amd" -o tools/prof_driver
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -ffp-contract=off tools/exact_math_probe.hip -o tools/exact_math_probe
[ -x tools/valu_rate ] && [ tools/valu_rate -nt tools/valu_rate.hip ] || /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/valu_rate.hip -o tools/valu_rate
