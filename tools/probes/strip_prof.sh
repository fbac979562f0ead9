#!/bin/bash
# Probe: rocprofv3 kernel statistics of one strip of an 8-way split (what one rank runs per frame).  tools/probes/strip_prof.sh 8:5
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$(pwd)
mkdir -p gpurun_out
L="--steps 200 --warmup 20 --no-cpu-baseline --no-producers --no-legs --strip $1"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/strip_prof -- python $R/bench.py $L > $R/gpurun_out/strip_prof.json 2> $R/gpurun_out/strip_prof.err) || { tail -5 gpurun_out/strip_prof.err; exit 1; }
python - <<PY
import csv, glob, json
f = glob.glob("gpurun_out/strip_prof/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    if "cry::" in r["Name"] and int(r["Calls"]) >= 200:
        per = float(r["TotalDurationNs"]) / 1e3
        print("%-44s calls %5s avg %8.2f us" % (r["Name"].split("(")[0][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
        tot += float(r["AverageNs"]) / 1e3 * (3 if "replay" in r["Name"] else 1)
d = json.load(open("gpurun_out/strip_prof.json"))
print("sum of kernels per frame %.1f us; frame %.1f us" % (tot, d["ms_per_step"] * 1e3))
PY
rm -rf gpurun_out/strip_prof
