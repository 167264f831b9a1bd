#!/bin/bash
# Same-box comparison of several library variants on the whole training step: bench.py with each SEGK_LIB in turn, N rounds.
#   tools/ab3_bench.sh 2 "" tools/ubench/bin/libsegk_x.so ...        ("" = the shipped library)
N=$1; shift
for i in $(seq 1 $N); do
  for L in "$@"; do
    line=$(SEGK_LIB=$L python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
    echo "[${L:-shipped}] $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["value"], d["unit"])')"
  done
done
