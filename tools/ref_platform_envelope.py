#!/usr/bin/env python3
"""The reference against itself under a legitimate recompile (CPU only; where oracle/_ref exists).

    python tools/ref_platform_envelope.py [image_size=512] [faces=39000]

Parity of the HIP kernels is pinned to the reference's kernel sources compiled for the host with -ffp-contract=off.
nvcc contracts a*b+c into FMA by default, so the reference's CUDA build is a different float program.  This tool
renders one view with both host builds (oracle/_ref/libsoftras_ref.so, libsoftras_ref_fma.so: build_ref.build_fma)
and prints how far they are apart - the scale against which "ids bit-exact, RGBA 5e-5, gradients 1e-4" of the HIP path
has to be read.  Same function as bench.py's `parity.reference_platform_envelope` (which runs it at 256x256)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import reference_platform_envelope           # noqa: E402

IS = int(sys.argv[1]) if len(sys.argv) > 1 else 512
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 39000
print(json.dumps(reference_platform_envelope(NF, 16, IS), indent=1))
