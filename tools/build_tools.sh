#!/bin/bash
# builds the torch-free profiling driver against the in-tree library (run from the repo root)
set -e
/opt/rocm/bin/hipcc -O2 -std=c++17 -x c++ -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/prof_driver.cpp \
  -L crychic_renderer_amd -lcrychic_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/root/repo/crychic_renderer_amd -o tools/prof_driver
