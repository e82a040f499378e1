#!/bin/bash
# usage: run_list.sh <outfile> <binary>...   (on the GPU box; nch from the m_ prefix)
cd "$GRAFT_REPO_ROOT/tools/k2lab"
OUT=../../gpurun_out/$1.txt; shift
: > $OUT
for b in "$@"; do
  case $b in m_*) n=1;; *) n=2;; esac
  timeout 60 bin/$b $n >> $OUT 2>&1 || echo "$b FAILED" >> $OUT
done
grep -v "coredump\|pipe\|rocdevice\|core\|matches" $OUT
