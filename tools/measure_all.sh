#!/bin/bash
# Run on the GPU box (gpurun): the round's measurement set, everything under gpurun_out/<tag>/.
#   tools/measure_all.sh <tag>
# 1 bench.py default (the driver's command: JSON line incl. cpu_baseline, producers and the informational legs)
# 2 rocprofv3 kernel trace of the same workload (no legs, so that every cry:: row is the headline configuration)
# 3 PMC passes on the torch-free driver        4 other configs (1080p / blurCount 1; 8K + 64 point lights)
# 5 the eight strips of an 8-GPU run, one at a time on this GPU (one frame at a time, and four in flight)
# 6 the same for 8K + 64 point lights; the kernels' register / LDS / occupancy table
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$(pwd)
out=gpurun_out/$1
mkdir -p "$out"
python -c "import __graft_entry__ as g; g.build()" > "$out/build.log" 2>&1 || exit 1
python bench.py > "$out/bench.json" 2> "$out/bench.err" || exit 1
L="--no-cpu-baseline --no-producers --no-legs"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/prof" -- python "$R/bench.py" --steps 100 --warmup 10 $L > "$R/$out/bench_under_rocprof.json" 2> "$R/$out/prof.err") || exit 1
cp "$(find "$out/prof" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats_all.csv"
python - "$out" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1] + "/kernel_stats_all.csv")) if "cry::" in r["Name"]]
with open(sys.argv[1] + "/kernel_stats.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-producers --no-legs (cry:: kernels only)\n")
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
    for r in rows: w.writerow([r["Name"].split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])
PY
rm -rf "$out/prof" "$out/kernel_stats_all.csv"
for leg in "--camera covered:camera_covered" "--pcf intended:pcf_intended"; do
  flag=${leg%%:*}; name=${leg##*:}
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/prof_$name" -- python "$R/bench.py" --steps 50 --warmup 5 $L $flag > "$R/$out/bench_$name.json" 2>> "$R/$out/prof.err") || exit 1
  python - "$out" "$name" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/prof_" + sys.argv[2] + "/**/*kernel_stats.csv", recursive=True)[0]
with open(sys.argv[1] + "/kernel_stats_" + sys.argv[2] + ".csv", "w") as o:
    w = csv.writer(o); w.writerow(["Name", "Calls", "AverageNs", "MinNs", "MaxNs"])
    for r in csv.DictReader(open(f)):
        if "cry::" in r["Name"]: w.writerow([r["Name"].split("(")[0], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
PY
  rm -rf "$out/prof_$name"
done
tools/pmc_passes.sh "$out/pmc" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" \
    "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
    "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "GRBM_GUI_ACTIVE" > "$out/pmc.log" 2>&1 || exit 1
rm -rf "$out"/pmc/pass*/
python bench.py --steps 100 --warmup 10 $L --width 1920 --height 1080 --blur-count 1 > "$out/bench_c2_1080p.json" 2>> "$out/bench.err" || exit 1
python bench.py --steps 30 --warmup 5 $L --width 7680 --height 4320 --point-lights 8 > "$out/bench_c5_8k_64lights.json" 2>> "$out/bench.err" || exit 1
tools/strips_rehearsal.sh "$out/strips.txt" 1 > /dev/null || exit 1
tools/strips_rehearsal.sh "$out/strips_4_in_flight.txt" 4 > /dev/null || exit 1
bash tools/resusage.sh > "$out/resource_usage.txt" 2>&1
# (6: the eight strips of BASELINE configs[4], 8K + 64 point lights, is a gpurun call of its own -- ~8 minutes:
#   tools/strips_rehearsal.sh gpurun_out/<tag>/strips_8k_64lights.txt 1 "--width 7680 --height 4320 --point-lights 8 --steps 30")
echo "measure_all done"
