#!/bin/bash
O=gpurun_out/r03c6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?" > $O/status.txt
tail -5 $O/parity.log
JRENDER_LIB=$PWD/jrender_amd/csrc/libjrender_hip_h64.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/parity_h64.log 2>&1; echo "parity h64 rc=$?" >> $O/status.txt
tail -5 $O/parity_h64.log
timeout 900 python tools/ablate/run.py --rounds 2 --no-parity h0 hnone h768 h1024 h1024d > $O/ablate.log 2>&1
grep -v "^{" $O/ablate.log; cat $O/status.txt
