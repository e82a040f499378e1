#!/bin/bash
cd "$GRAFT_REPO_ROOT/tools/k2lab"
OUT=../../gpurun_out/k2_nc2.txt; : > $OUT
for rep in 1 2; do for pad in 0 4224; do for wrap in 0 2; do for b in nc_s1 nc_h8; do echo "pad $pad wrap $wrap" >> $OUT; timeout 120 bin/$b 2 1024 256 8 $wrap $pad >> $OUT 2>&1; done; done; done; done
grep "^pad\|avg" $OUT | paste - - | awk '{print $1,$2,$3,$4,$5,$8,$9,$11,$12}'
