#!/bin/bash
# usage (on the GPU box, from repo root): tools/pmc_run.sh <outdir> <counter list...>
# One rocprofv3 --pmc pass (kernel-trace not combined) over a short bench run.
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc "$@" -d "$out" -o pmc --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $BENCH_ARGS > "$out/bench.log" 2>&1
