#!/bin/bash
# Run on the GPU box: build, one bench line without legs, and the rocprofv3 kernel statistics of the same command.
#   tools/probes/bench_only.sh <tag>
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
tag=$1
R=$(pwd)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${tag}_build.log 2>&1 || { tail -20 gpurun_out/${tag}_build.log; exit 1; }
L="--steps 200 --warmup 20 --no-cpu-baseline --no-producers --no-legs"
timeout -k 10 300 python bench.py $L > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -- python $R/bench.py $L > $R/gpurun_out/${tag}_bench_prof.json 2> $R/gpurun_out/${tag}_prof.err) || { tail -5 gpurun_out/${tag}_prof.err; exit 1; }
cp "$(find gpurun_out/${tag}_prof -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/${tag}_prof
python - <<PY
import csv, json
for r in csv.DictReader(open("gpurun_out/${tag}_kernel_stats.csv")):
    if "cry::" in r["Name"]:
        print("%-44s calls %5s avg %8.2f us  min %8.2f" % (r["Name"].split("(")[0][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
d = json.load(open("gpurun_out/${tag}_bench.json"))
print("ms", d["ms_per_step"], "median", d["config"]["frame_ms_median_hipevent"], "frac", d["roofline"]["frac"], d["config"]["pass_ms"])
PY
