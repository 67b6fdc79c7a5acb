#!/bin/bash
out=gpurun_out/n3mr_prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 90 rocprofv3 --kernel-trace --stats -d $out/trace -o n3mr --output-format csv -- python bench.py --workload n3mr --steps 10 --warmup 2 --no-cpu-baseline > $out/traced.log 2>&1
cp $out/trace/*kernel_stats.csv $out/n3mr_kernel_stats.csv
python - <<'P'
import csv
for r in csv.DictReader(open("gpurun_out/n3mr_prof/n3mr_kernel_stats.csv")):
    print("%-40s calls %4s avg %10.1f ns" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"])))
P
