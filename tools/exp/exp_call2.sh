#!/bin/bash
# round 3, GPU call 2: how does the forward scale with the batch (critical path of the heaviest tile vs throughput)?
O=gpurun_out/r03c2; mkdir -p $O
for v in product no_defer; do
  L=$PWD/jrender_amd/csrc/libjrender_hip.so; [ $v != product ] && L=$PWD/jrender_amd/csrc/libjrender_hip_$v.so
  for b in 1 2 4 8 16 32; do
    echo -n "$v B=$b " >> $O/batch_scaling.txt
    JRENDER_LIB=$L python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-secondary | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v,4) for k,v in d["phase_ms_per_step"].items()}, round(d["ms_per_step"],4))' >> $O/batch_scaling.txt 2>&1
  done
done
cat $O/batch_scaling.txt
