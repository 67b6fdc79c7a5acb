#!/bin/bash
O=gpurun_out/r03c5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?" > $O/status.txt
tail -15 $O/parity.log
timeout 900 python tools/ablate/run.py --rounds 2 --no-parity product h0 h64 h128 h384 h512 > $O/ablate.log 2>&1
for v in product h0 h64 h128; do for b in 1 2; do
  L=$PWD/jrender_amd/csrc/libjrender_hip.so; [ $v != product ] && L=$PWD/jrender_amd/csrc/libjrender_hip_$v.so
  echo -n "$v B=$b " >> $O/scal.txt
  JRENDER_LIB=$L timeout 120 python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-secondary | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v,4) for k,v in d["phase_ms_per_step"].items()})' >> $O/scal.txt 2>&1
done; done
grep -v "^{" $O/ablate.log; cat $O/scal.txt; cat $O/status.txt
