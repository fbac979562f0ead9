#!/bin/bash
# On the GPU box: light_kernel time against resident wavefronts per SIMD, capped by a dynamic LDS allocation (probe build ldscap).
cd "$(dirname "$0")/../.."
R=$(pwd); export TMPDIR=/tmp
for lds in 0 26000 32000 40000 53000 80000; do
  (cd /tmp && CRY_PROBE_LDS=$lds CRYCHIC_LIB=$R/tools/_probe/lib_ldscap.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/occ_prof -- python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-producers --no-legs $1 > /dev/null 2> $R/gpurun_out/occ.err) || { tail -3 gpurun_out/occ.err; exit 1; }
  python - $lds <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/occ_prof/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "light_kernel" in r["Name"]: print("dynamic LDS %6s B -> light_kernel %.1f us" % (sys.argv[1], float(r["AverageNs"]) / 1e3))
PY
  rm -rf gpurun_out/occ_prof
done
