#!/bin/bash
# A/B timing: tools/ab_file.sh <alt_backward.hip|-> <alt_forward.hip|->   (results may be wrong in variants)
set -e
B=jrender_amd/csrc
bo=$B/softras_backward.o; fo=$B/softras_forward.o
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -I$B -x hip"
if [ "$1" != "-" ]; then /opt/rocm/bin/hipcc $FL -c $1 -o /tmp/ab_b.o; bo=/tmp/ab_b.o; fi
if [ "$2" != "-" ] && [ -n "$2" ]; then /opt/rocm/bin/hipcc $FL -c $2 -o /tmp/ab_f.o; fo=/tmp/ab_f.o; fi
cp $B/libjrender_hip.so /tmp/lib_backup.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/jr_api.o $B/binning.o $fo $bo $B/aux_kernels.o $B/n3mr_kernels.o -o $B/libjrender_hip.so
python bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['phase_ms_per_step'])"
cp /tmp/lib_backup.so $B/libjrender_hip.so
