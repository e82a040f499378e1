#!/bin/bash
# run on the GPU box: every binary in tools/k2lab/bin, 3 s timeout each
cd "$GRAFT_REPO_ROOT/tools/k2lab"
mkdir -p ../../gpurun_out
OUT=../../gpurun_out/${1:-k2lab}.txt
: > $OUT
for b in $(ls bin | sort); do
  case $b in m_*) n=1;; *) n=2;; esac
  timeout 60 bin/$b $n >> $OUT 2>&1 || echo "$b FAILED" >> $OUT
done
cat $OUT
