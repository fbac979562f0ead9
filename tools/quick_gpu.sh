#!/bin/bash
# Run on the GPU box: build, the parity tests named on the command line (default: the hot-path ones), one bench line and the
# rocprofv3 kernel statistics of the same command.   tools/quick_gpu.sh <tag> [pytest args...]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${tag}_build.log 2>&1 || { tail -20 gpurun_out/${tag}_build.log; exit 1; }
tests=${@:-tests/test_gpu_parity.py tests/test_golden.py tests/test_fuzz.py}
timeout -k 10 600 python -m pytest $tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -4 gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-producers > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-producers --no-legs > $R/gpurun_out/${tag}_bench_prof.json 2> $R/gpurun_out/${tag}_prof.err) || { tail -5 gpurun_out/${tag}_prof.err; exit 1; }
cp "$(find gpurun_out/${tag}_prof -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/${tag}_prof
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/${tag}_kernel_stats.csv")):
    if "cry::" in r["Name"]:
        print("%-44s calls %5s avg %8.2f us  min %8.2f" % (r["Name"].split("(")[0][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
c = d["config"]
print("value", d["value"], "ms", d["ms_per_step"], "median", c["frame_ms_median_hipevent"], "frac", d["roofline"]["frac"], "pass", c["pass_ms"])
for k in ("throughput_3_in_flight", "pcf_intended", "camera_covered", "cube_mip_chain"):
    print(k, c.get(k))
PY
