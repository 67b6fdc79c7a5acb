#!/bin/bash
# GPU box: SQ counters of the raster kernels for several ablation variants (one rocprofv3 --pmc pass per
# counter group and variant; no tracing in the same pass).  usage: tools/ablate/pmc_compare.sh <outdir> <variant...>
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
G2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/jrender_amd/csrc/libjrender_hip_$v.so
  [ "$v" = product ] && lib=$GRAFT_REPO_ROOT/jrender_amd/csrc/libjrender_hip.so
  i=0
  for g in "$G1" "$G2"; do
    i=$((i+1))
    JRENDER_LIB=$lib rocprofv3 --pmc $g -d "$out/${v}_g$i" -o pmc --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$out/${v}_g$i.log" 2>&1
    f=$(find "$out/${v}_g$i" -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" | grep -A9 "k_softras_forward\|k_softras_backward" > "$out/${v}_g$i.txt"
  done
done
tail -n +1 "$out"/*.txt
