#!/bin/bash
# round 3, GPU call 1: parity of the new product, one-switch ablations, profiles of old (r2fwd) and new forward, full bench line
mkdir -p gpurun_out/r03c1
O=gpurun_out/r03c1
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/status.txt
python tools/ablate/run.py --rounds 2 --quick-parity product r2fwd no_shift no_defer no_exp1 > $O/ablate1.log 2>&1; echo "ablate rc=$?" >> $O/status.txt
tools/collect_profiles.sh r03_v1 trace fetch write sq; echo "prof v1 rc=$?" >> $O/status.txt
JRENDER_LIB=$PWD/jrender_amd/csrc/libjrender_hip_r2fwd.so tools/collect_profiles.sh r03_v0 trace sq; echo "prof v0 rc=$?" >> $O/status.txt
python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?" >> $O/status.txt
tail -3 $O/pytest_gpu.log; cat $O/ablate1.log | grep -v "^{" | tail -12; cat $O/status.txt
