#!/bin/bash
# Run on the GPU box: what each rank of an 8-GPU run computes per frame, measured one strip at a time on one GPU (no exchange):
#   tools/strips_rehearsal.sh <out.txt> [frames in flight, default 1 = the benchmark's own mode] ["extra bench.py flags", e.g. the 8K + 64 lights config]
cd "$(dirname "$0")/.."
out=$1
F=${2:-1}
X=${3:-}
: > "$out"
L="--no-cpu-baseline --no-producers --no-legs --steps 100 --warmup 10 --frames-in-flight $F $X"
echo "frames in flight: $F   extra flags: ${X:-none}" >> "$out"
full=$(python bench.py $L 2>/dev/null | python -c "import sys, json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
echo "whole frame, one GPU: $full ms" >> "$out"
for part in equal balanced; do
  worst=0
  for r in 0 1 2 3 4 5 6 7; do
    line=$(python bench.py $L --strip 8:$r --partition $part 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
print(o['config']['strip_rows'], o['ms_per_step'], o['config']['pass_ms'])")
    echo "$part strip 8:$r rows/ms/passes $line" >> "$out"
  done
done
python - "$out" <<'PY'
import sys, re
full = None; worst = {}
for l in open(sys.argv[1]):
    if l.startswith("whole"): full = float(l.split(":")[1].split()[0])
    m = re.match(r"(\w+) strip 8:\d rows/ms/passes (\d+) ([\d.]+)", l)
    if m: worst[m.group(1)] = max(worst.get(m.group(1), 0.0), float(m.group(3)))
with open(sys.argv[1], "a") as f:
    for k, v in worst.items():
        f.write("%s: slowest strip %.4f ms => at most %.2fx over one GPU's %.4f ms (before the exchange)\n" % (k, v, full / v, full))
PY
cat "$out"
