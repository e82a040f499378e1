#!/bin/bash
# on the GPU box: round-3 K2u variants, alternating, stereo config 3 then other channel counts
cd "$GRAFT_REPO_ROOT/tools/k2lab"
OUT=../../gpurun_out/${1:-k2lab_r03}.txt
: > $OUT
for rep in 1 2; do
  for b in old prod pairdma quad_w8 quad_w4; do
    timeout 120 bin/$b 2 1024 256 8 >> $OUT 2>&1 || echo "$b FAILED" >> $OUT
  done
done
echo "== mono" >> $OUT
for rep in 1 2; do for b in old prod; do timeout 120 bin/$b 1 1024 256 8 >> $OUT 2>&1 || echo "$b FAILED" >> $OUT; done; done
echo "== 4 channels (512 x 128)" >> $OUT
for rep in 1 2; do for b in old prod pairdma; do timeout 120 bin/$b 4 512 128 8 >> $OUT 2>&1 || echo "$b FAILED" >> $OUT; done; done
echo "== 8 channels (512 x 128)" >> $OUT
for rep in 1 2; do for b in old prod pairdma; do timeout 120 bin/$b 8 512 128 8 >> $OUT 2>&1 || echo "$b FAILED" >> $OUT; done; done
grep -v "coredump\|pipe\|rocdevice" $OUT
