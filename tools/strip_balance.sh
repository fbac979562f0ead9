#!/bin/bash
# On a 1-GPU box: time every rank's strip of an N-way split, equal vs balanced partition (what bounds the N-GPU frame rate
# before the gather).  usage: tools/strip_balance.sh N [covered-weight]
cd "$(dirname "$0")/.."
N=$1; CW=${2:-2.3}
for part in equal balanced; do
  line="N=$N $part (weight $CW):"
  for ((r=0; r<N; r++)); do
    python bench.py --strip $N:$r --partition $part --covered-weight $CW --steps 100 --warmup 10 --no-cpu-baseline --no-producers > /tmp/strip.log 2>&1
    ms=$(grep -o '"ms_per_step": [0-9.]*' /tmp/strip.log | cut -d' ' -f2); rows=$(grep -o '"strip_rows": [0-9]*' /tmp/strip.log | cut -d' ' -f2)
    line="$line r$r=${ms}ms/${rows}rows"
  done
  echo "$line"
done
