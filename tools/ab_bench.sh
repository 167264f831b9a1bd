#!/bin/bash
# Same-box A/B of the whole training step: bench.py alternately with the environment of variant A and of variant B.
#   tools/ab_bench.sh "SEGK_HEAD_ON_Z=0" "" [rounds]      (an empty string = the default build)
A="$1"; B="$2"; N=${3:-2}
for i in $(seq 1 $N); do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    line=$(env $E python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1)
    echo "$v [$E] $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["value"], d["unit"])')"
  done
done
