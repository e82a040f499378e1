#!/bin/bash
# usage: run_list2.sh <outfile> <nch> <clips> <frames> <reps> <binary>...   (on the GPU box; alternating order)
cd "$GRAFT_REPO_ROOT/tools/k2lab"
OUT=../../gpurun_out/$1.txt; N=$2; C=$3; F=$4; R=$5; shift 5
: > $OUT
for rep in $(seq $R); do for b in "$@"; do timeout 120 bin/$b $N $C $F 8 >> $OUT 2>&1 || echo "$b FAILED" >> $OUT; done; done
grep -v "coredump\|pipe\|rocdevice\|matches" $OUT
