#!/bin/bash
# The round's evidence in ONE gpurun call, on the sources as they are: tools/round_end.sh <tag>
#   GPU tests, kernel trace + PMC passes of the headline batch and of one view, NMR trace + PMC, profiles/*_latest.json
#   regenerated FROM THIS RUN (stamped with the hash of the kernel sources), then the plain bench line (which therefore
#   prints roofline.profile_stale = false), the gradient-parity table and the other BASELINE configurations.
# Everything lands in gpurun_out/<tag>/; copy into profiles/ (tools/README.md).
tag=$1
out=gpurun_out/$tag
mkdir -p $out
(timeout 1500 python -m pytest tests -m gpu -q > $out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_pytest_gpu.log)
tools/collect_profiles.sh $tag trace fetch write sq
cp gpurun_out/profiles_$tag/${tag}_* $out/ 2>/dev/null
python tools/pmc_to_json.py gpurun_out/profiles_$tag $tag > $out/${tag}_pmc_to_json.log 2>&1
BENCH_ARGS="--batch 1" tools/collect_profiles.sh ${tag}_b1 trace sq
cp gpurun_out/profiles_${tag}_b1/${tag}_b1_* $out/ 2>/dev/null
rm -rf gpurun_out/n3mr_prof; tools/collect_profiles_n3mr.sh
for f in gpurun_out/n3mr_prof/n3mr_*; do cp $f $out/${tag}_$(basename $f); done
python tools/pmc_n3mr_to_json.py gpurun_out/n3mr_prof $tag > $out/${tag}_pmc_n3mr_to_json.log 2>&1
# round 6: both raster kernels' VALU floors from THIS run's counters (instrumented builds count_paths / count_paths_bwd must exist:
# python tools/ablate/build.py count_paths count_paths_bwd) - profiles/min_valu_latest.json, min_valu_bwd_latest.json
if [ -f jrender_amd/csrc/libjrender_hip_count_paths.so ]; then
  (timeout 200 python tools/sim/min_valu.py --measure > /dev/null 2>&1; timeout 200 python tools/sim/min_valu.py > $out/${tag}_min_valu.txt 2>&1)
  cp profiles/r05_path_counts.json $out/${tag}_path_counts_fwd.json 2>/dev/null
fi
if [ -f jrender_amd/csrc/libjrender_hip_count_paths_bwd.so ]; then
  (timeout 200 python tools/sim/min_valu_bwd.py --measure > /dev/null 2>&1; timeout 200 python tools/sim/min_valu_bwd.py > $out/${tag}_min_valu_bwd.txt 2>&1)
  cp profiles/r06_path_counts_bwd.json $out/ 2>/dev/null
fi
cp profiles/traffic_latest.json profiles/valu_latest.json profiles/traffic_n3mr_latest.json profiles/min_valu_latest.json profiles/min_valu_bwd_latest.json $out/ 2>/dev/null
(timeout 300 python __graft_entry__.py --smoke > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?" >> $out/${tag}_smoke.log)
timeout 600 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout 300 python bench.py --scene soup --no-secondary > $out/${tag}_bench_soup.json 2>> $out/${tag}_bench.err
timeout 400 python tools/grad_parity.py > $out/${tag}_grad_parity.txt 2>&1
timeout 400 python tools/time_configs.py > $out/${tag}_time_configs.txt 2>&1
# round 5: randomised sweeps on the final kernels (the SoftRas one draws the launch organisation - bin size, threshold, workgroup size,
# colour path - per case), the automatic launch policy against the 32-pixel reference on every critical-path shape, the N-rank dry run
(timeout 900 python tests/fuzz_parity.py --cases 600 --seed ${FUZZ_SEED:-81} > $out/${tag}_fuzz_softras.log 2>&1; echo "rc=$?" >> $out/${tag}_fuzz_softras.log)
(timeout 300 python tests/fuzz_n3mr.py --cases 1500 --seed ${FUZZ_SEED:-81} > $out/${tag}_fuzz_n3mr.log 2>&1; echo "rc=$?" >> $out/${tag}_fuzz_n3mr.log)
timeout 600 python tools/geometry_sweep.py --quick --bins 32 > $out/${tag}_policy_vs_bin32.txt 2>&1
for n in 2 4 8; do timeout 120 python bench.py --dry-run-ranks $n > $out/${tag}_dry_run_${n}_ranks.json 2>> $out/${tag}_bench.err; done
# kernel trace of the demo2 loop with the chain around the rasteriser on the device
(cd /tmp; export TMPDIR=/tmp; cd - > /dev/null; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/demo2_prof -o demo2 --output-format csv -- python examples/demo2_deform.py --iters 100 --quiet > $out/${tag}_demo2_traced.log 2>&1)
cp $(find gpurun_out/demo2_prof -name "*kernel_stats.csv" | head -1) $out/${tag}_demo2_kernel_stats.csv 2>/dev/null; rm -rf gpurun_out/demo2_prof $out/${tag}_demo2_traced.log
rm -rf gpurun_out/profiles_$tag gpurun_out/profiles_${tag}_b1 gpurun_out/n3mr_prof
tail -n 3 $out/${tag}_pytest_gpu.log; tail -n 1 $out/${tag}_smoke.log; tail -n 2 $out/${tag}_fuzz_softras.log; tail -n 2 $out/${tag}_fuzz_n3mr.log; grep '^##' $out/${tag}_policy_vs_bin32.txt; tail -c 400 $out/${tag}_bench.json
