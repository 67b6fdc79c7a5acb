#!/bin/bash
O=gpurun_out/r03c11; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?" > $O/status.txt
tail -3 $O/parity.log
JRENDER_LIB=$PWD/jrender_amd/csrc/libjrender_hip_h128.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz_slice.py -x -q -m gpu > $O/parity_h128.log 2>&1; echo "parity h128 rc=$?" >> $O/status.txt
tail -3 $O/parity_h128.log
python tools/ablate/sections.py --heavy > $O/sections_heavy.txt 2>&1
cat $O/sections_heavy.txt
show='import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v,4) for k,v in d["phase_ms_per_step"].items()})'
for v in h512 h1024; do for b in 1 2 4 8; do
  L=$PWD/jrender_amd/csrc/libjrender_hip_$v.so
  echo -n "$v B=$b " >> $O/scal.txt
  JRENDER_LIB=$L timeout 120 python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-secondary | python -c "$show" >> $O/scal.txt 2>&1
done; done
cat $O/scal.txt; cat $O/status.txt
