#!/bin/bash
cd "$GRAFT_REPO_ROOT/tools/k2lab"
OUT=../../gpurun_out/k2_pad.txt; : > $OUT
for rep in 1 2 3; do for pad in 0 4224; do for b in pair_s1 pair_h8; do echo "pad $pad" >> $OUT; timeout 120 bin/$b 2 1024 256 8 0 $pad >> $OUT 2>&1; done; done; done
grep "pad\|avg" $OUT | paste - - | awk '{print $1,$2,$3,$6,$7,$9,$10}'
