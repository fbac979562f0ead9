#!/bin/bash
# Probe: rocprofv3 kernel statistics of the producer passes (shadow cascades, normals + depth + G-buffer) as bench.py times them.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$(pwd)
mkdir -p gpurun_out
L="--steps 20 --warmup 5 --no-cpu-baseline --no-legs"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prod_prof -- python $R/bench.py $L > $R/gpurun_out/prod_prof.json 2> $R/gpurun_out/prod_prof.err) || { tail -5 gpurun_out/prod_prof.err; exit 1; }
python - <<PY
import csv, glob, json
f = glob.glob("gpurun_out/prod_prof/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "cry::" in r["Name"] and ("raster" in r["Name"] or "setup" in r["Name"] or "resolve" in r["Name"] or "clear" in r["Name"]):
        print("%-70s calls %5s avg %8.2f us total %9.1f us" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
d = json.load(open("gpurun_out/prod_prof.json"))
print(d["config"]["producer_passes_ms"])
PY
rm -rf gpurun_out/prod_prof
