#!/bin/bash
O=gpurun_out/r03c10; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?" > $O/status.txt
tail -3 $O/parity.log
JRENDER_LIB=$PWD/jrender_amd/csrc/libjrender_hip_h128.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/parity_h128.log 2>&1; echo "parity h128 rc=$?" >> $O/status.txt
tail -3 $O/parity_h128.log
timeout 900 python tools/ablate/run.py --rounds 2 --no-parity h0 product h512 h768 h1024 w5b46 > $O/ablate.log 2>&1
grep -v "^{" $O/ablate.log; cat $O/status.txt
