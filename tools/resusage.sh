#!/bin/bash
# per-kernel register / scratch / LDS / occupancy figures for csrc/kernels.hip (run from the repo root)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -I include -I crychic_renderer_amd/csrc -c crychic_renderer_amd/csrc/kernels.hip -o /tmp/k.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep remark | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | awk '/Function Name/{printf "\n%s ", $0; next}{printf "| %s ", $0}END{print ""}' | sed -E 's/\| (Bytes|Dynamic|Uses)[^|]*//g; s/_ZN3cry//'
