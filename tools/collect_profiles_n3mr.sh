#!/bin/bash
# NMR workload: kernel trace + PMC passes (separate)
out=gpurun_out/n3mr_prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d $out/trace -o n3mr --output-format csv -- python bench.py --workload n3mr --steps 10 --warmup 2 --no-cpu-baseline > $out/traced.log 2>&1
cp $out/trace/*kernel_stats.csv $out/n3mr_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/$c -o pmc --output-format csv -- python bench.py --workload n3mr --steps 3 --warmup 1 --no-cpu-baseline > $out/$c.log 2>&1
  python tools/pmc_summary.py $(find $out/$c -name '*counter_collection.csv' | head -1) > $out/n3mr_pmc_$c.txt
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD -d $out/sq -o pmc --output-format csv -- python bench.py --workload n3mr --steps 3 --warmup 1 --no-cpu-baseline > $out/sq.log 2>&1
python tools/pmc_summary.py $(find $out/sq -name '*counter_collection.csv' | head -1) > $out/n3mr_pmc_sq.txt
