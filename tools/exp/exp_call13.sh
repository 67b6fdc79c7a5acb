#!/bin/bash
O=gpurun_out/r03c13; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/status.txt
tail -4 $O/pytest_gpu.log
JRENDER_LIB=$PWD/jrender_amd/csrc/libjrender_hip_h128.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz_slice.py -x -q -m gpu > $O/parity_h128.log 2>&1; echo "parity h128 rc=$?" >> $O/status.txt
tail -3 $O/parity_h128.log
timeout 900 python tools/ablate/run.py --rounds 3 --no-parity product no_hint > $O/ablate.log 2>&1
grep -v "^{" $O/ablate.log
show='import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v,4) for k,v in d["phase_ms_per_step"].items()})'
for v in product no_hint; do for b in 1 4; do
  L=$PWD/jrender_amd/csrc/libjrender_hip.so; [ $v != product ] && L=$PWD/jrender_amd/csrc/libjrender_hip_$v.so
  echo -n "$v B=$b " >> $O/scal.txt
  JRENDER_LIB=$L timeout 120 python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-secondary | python -c "$show" >> $O/scal.txt 2>&1
done; done
cat $O/scal.txt; cat $O/status.txt
