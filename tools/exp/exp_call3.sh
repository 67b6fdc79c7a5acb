#!/bin/bash
O=gpurun_out/r03c3; mkdir -p $O
python tools/ablate/run.py --rounds 2 --no-parity no_defer prio256 prio600 occ3 occ2 > $O/ablate.log 2>&1
python tools/ablate/sections.py > $O/sections_b8.txt 2>&1
grep -v "^{" $O/ablate.log; cat $O/sections_b8.txt
