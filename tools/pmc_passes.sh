#!/bin/bash
# Run on the GPU box: dump the bench scene, then one rocprofv3 pass per counter group on the torch-free driver.
#   tools/pmc_passes.sh <outdir under gpurun_out> "<group1 counters>" "<group2 counters>" ...
set -e
cd "$(dirname "$0")/.."
R=$(pwd)
export TMPDIR=/tmp
out=$1; shift
bash tools/build_tools.sh      # never profile a stale driver
mkdir -p "$out"
[ -d /tmp/scene4k ] || python bench.py --dump-scene /tmp/scene4k --no-cpu-baseline --no-producers > "$out/dump.log" 2>&1
i=0
for grp in "$@"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$R/$out/pass$i" -- "$R/tools/prof_driver" /tmp/scene4k 4 1 > "$R/$out/pass$i.log" 2>&1)
done
python tools/pmc_table.py "$out"
