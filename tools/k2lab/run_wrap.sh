#!/bin/bash
cd "$GRAFT_REPO_ROOT/tools/k2lab"
OUT=../../gpurun_out/k2_wrap.txt; : > $OUT
for rep in 1 2 3; do for w in 0 2 64; do for b in pair_s1 prod; do echo "wrap $w" >> $OUT; timeout 120 bin/$b 2 1024 256 8 $w >> $OUT 2>&1; done; done; done
grep "wrap\|avg" $OUT | paste - - | awk '{print $1,$2,$3,$6,$7,$9,$10}'
