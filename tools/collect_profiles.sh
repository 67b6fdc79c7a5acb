#!/bin/bash
# Run on the GPU box from the repo root: tools/collect_profiles.sh <tag> [passes: trace bench fetch write sq]
# trace : rocprofv3 --kernel-trace --stats of the default bench command (no PMC in that pass)
# bench : the plain default bench line (what the driver runs)
# fetch / write / sq : separate --pmc passes (FETCH_SIZE | WRITE_SIZE | the SQ instruction / lane counters)
# JRENDER_LIB=<other build of libjrender_hip.so> profiles that build instead of the product; BENCH_ARGS="--batch 1" another workload.
tag=$1; shift
passes=${*:-trace bench fetch write sq}
out=gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for p in $passes; do
  case $p in
    trace) rocprofv3 --kernel-trace --stats -d $out/trace -o $tag --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary $BENCH_ARGS > $out/bench_traced.log 2>&1
           cp $out/trace/*kernel_stats.csv $out/${tag}_kernel_stats.csv ;;
    bench) python bench.py > $out/${tag}_bench.json 2> $out/bench.err ;;
    fetch) tools/pmc_run.sh $out/fetch FETCH_SIZE ;;
    write) tools/pmc_run.sh $out/write WRITE_SIZE ;;
    sq)    tools/pmc_run.sh $out/sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
           tools/pmc_run.sh $out/sq2 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_FLAT ;;
  esac
done
for d in fetch write sq sq2; do
  f=$(find $out/$d -name '*counter_collection.csv' 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" > $out/${tag}_pmc_$d.txt
done
exit 0
