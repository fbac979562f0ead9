#!/bin/bash
# Run on the GPU box (gpurun): the round's measurement set, everything under gpurun_out/<tag>/.
#   tools/measure_all.sh <tag>
# 1 bench.py default (JSON line incl. cpu_baseline + producers)   2 rocprofv3 kernel trace of the same command
# 3 PMC passes on the torch-free driver                           4 informational legs (covered camera, intended PCF, 1080p,
#   8K + 64 point lights, 1 / 2 / 4 frames in flight, the eight strips of an 8-GPU run one at a time)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/$1
mkdir -p "$out"
B="python bench.py --steps 100 --warmup 10"
$B > "$out/bench.json" 2> "$out/bench.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-producers > "$out/bench_under_rocprof.json" 2> "$out/prof.err" || exit 1
cp "$(find "$out/prof" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"
tools/pmc_passes.sh "$out/pmc" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" \
    "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
    "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" > "$out/pmc.log" 2>&1 || exit 1
L="--no-cpu-baseline --no-producers"
$B $L --camera covered > "$out/bench_camera_covered.json" 2>> "$out/bench.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_covered" -- python bench.py --steps 50 --warmup 5 $L --camera covered > /dev/null 2>> "$out/prof.err" || exit 1
cp "$(find "$out/prof_covered" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats_camera_covered.csv"
$B $L --pcf intended > "$out/bench_pcf_intended.json" 2>> "$out/bench.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_pcf" -- python bench.py --steps 50 --warmup 5 $L --pcf intended > /dev/null 2>> "$out/prof.err" || exit 1
cp "$(find "$out/prof_pcf" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats_pcf_intended.csv"
for f in 1 2 4; do $B $L --frames-in-flight $f > "$out/bench_${f}_in_flight.json" 2>> "$out/bench.err" || exit 1; done
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_serial" -- python bench.py --steps 100 --warmup 10 $L --frames-in-flight 1 > "$out/bench_1_in_flight_under_rocprof.json" 2>> "$out/prof.err" || exit 1
cp "$(find "$out/prof_serial" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats_1_in_flight.csv"
$B $L --width 1920 --height 1080 --blur-count 1 > "$out/bench_c2_1080p.json" 2>> "$out/bench.err" || exit 1
python bench.py --steps 30 --warmup 5 $L --width 7680 --height 4320 --point-lights 8 > "$out/bench_c5_8k_64lights.json" 2>> "$out/bench.err" || exit 1
: > "$out/strips.txt"
for part in equal balanced; do
  for r in 0 1 2 3 4 5 6 7; do
    python bench.py --steps 100 --warmup 10 $L --strip 8:$r --partition $part --frames-in-flight 4 2>> "$out/bench.err" | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
print('$part strip 8:$r rows', o['config']['strip_rows'], 'ms', o['ms_per_step'])" >> "$out/strips.txt" || exit 1
  done
done
rm -rf "$out/prof" "$out/prof_covered" "$out/prof_pcf" "$out/prof_serial"
echo "measure_all done"
