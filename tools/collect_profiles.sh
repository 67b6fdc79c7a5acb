#!/bin/bash
# Run on the GPU box from the repo root: tools/collect_profiles.sh <tag>
# 1) rocprofv3 --kernel-trace --stats of the default bench command (no PMC in that pass)
# 2) separate --pmc passes: FETCH_SIZE, WRITE_SIZE, and one SQ pass
tag=$1
out=gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d $out/trace -o $tag --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_traced.log 2>&1
python bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err
tools/pmc_run.sh $out/fetch FETCH_SIZE
tools/pmc_run.sh $out/write WRITE_SIZE
tools/pmc_run.sh $out/sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
cp $out/trace/*kernel_stats.csv $out/${tag}_kernel_stats.csv
for d in fetch write sq; do python tools/pmc_summary.py $out/$d/pmc_counter_collection.csv > $out/${tag}_pmc_$d.txt; done
