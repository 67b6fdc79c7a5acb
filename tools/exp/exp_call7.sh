#!/bin/bash
O=gpurun_out/r03c7; mkdir -p $O
python tools/ablate/sections.py --heavy > $O/sections_heavy.txt 2>&1
cat $O/sections_heavy.txt
show='import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v,4) for k,v in d["phase_ms_per_step"].items()})'
for pad in 0 3400 10200 22000; do for b in 8 1; do
  echo -n "h0 pad=$pad B=$b " >> $O/pad.txt
  JR_FWD_LDS_PAD=$pad JRENDER_LIB=$PWD/jrender_amd/csrc/libjrender_hip_h0.so timeout 120 python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-secondary | python -c "$show" >> $O/pad.txt 2>&1
done; done
cat $O/pad.txt
