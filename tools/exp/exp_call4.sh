#!/bin/bash
O=gpurun_out/r03c4; mkdir -p $O
python tools/ablate/run.py --rounds 2 --no-parity no_defer diag_nosoftmax diag_nokbuf diag_neither > $O/ablate.log 2>&1
for v in no_defer diag_neither; do for b in 1 4 32; do
  echo -n "$v B=$b " >> $O/scal.txt
  JRENDER_LIB=$PWD/jrender_amd/csrc/libjrender_hip_$v.so python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-secondary | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v,4) for k,v in d["phase_ms_per_step"].items()})' >> $O/scal.txt 2>&1
done; done
grep -v "^{" $O/ablate.log; cat $O/scal.txt
