#!/bin/bash
# the headline value against the length of the timed region and of the warm-up before it
mkdir -p gpurun_out/r06
out=gpurun_out/r06/steps_sweep2.txt
: > $out
for wu in 5 20 40 100; do
  for st in 5 20 64; do
    line=$(timeout 300 python3 bench.py --gpus 1 --steps $st --warmup $wu --no-cpu-baseline 2>/dev/null)
    echo "warmup=$wu steps=$st $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("value", d["value"], "ms_per_step", d["ms_per_step"], "total_ms", round(d["ms_per_step"]*d["steps"],3), "verified", d["verified"])')" >> $out
  done
done
cat $out
