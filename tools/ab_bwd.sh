#!/bin/bash
# A/B timing experiment (results intentionally wrong in the variants): build the library with -DJR_EXP=<n>, run bench
set -e
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -DJR_EXP=$v -x hip -c jrender_amd/csrc/softras_backward.hip -o /tmp/bwd_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC jrender_amd/csrc/jr_api.o jrender_amd/csrc/binning.o jrender_amd/csrc/softras_forward.o /tmp/bwd_$v.o jrender_amd/csrc/aux_kernels.o -o jrender_amd/csrc/libjrender_hip.so
  echo "== JR_EXP=$v"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['phase_ms_per_step'])"
done
