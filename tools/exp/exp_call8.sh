#!/bin/bash
O=gpurun_out/r03c8; mkdir -p $O
timeout 900 python tools/ablate/run.py --rounds 2 --no-parity h0 w5b44 w5b40 w5b46 w6b36 w6b32 > $O/ablate.log 2>&1
grep -v "^{" $O/ablate.log
