#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab_lib.sh <other_lib.so> [repeats]
# (boxes differ by a few per cent, so only numbers from ONE gpurun call are comparable)
L=jrender_amd/csrc/libjrender_hip.so
cp $L /tmp/lib_new.so
show='import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v,4) for k,v in d["phase_ms_per_step"].items()}, round(d["ms_per_step"],4), round(d["value"],1))'
for r in $(seq 1 ${2:-2}); do
  echo -n "new : "; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary | python -c "$show"
  cp $1 $L
  echo -n "other: "; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary | python -c "$show"
  cp /tmp/lib_new.so $L
done
