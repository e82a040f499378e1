#!/bin/bash
cd "$GRAFT_REPO_ROOT/tools/k2lab"
OUT=../../gpurun_out/k2_opad.txt; : > $OUT
for rep in 1 2 3; do for pad in 0 128 2176 640; do for b in pair_s1; do echo "opad $pad" >> $OUT; timeout 120 bin/$b 2 1024 256 8 0 0 $pad >> $OUT 2>&1; done; done; done
grep "opad\|avg" $OUT | paste - - | awk '{print $1,$2,$3,$6,$7,$9,$10}'
