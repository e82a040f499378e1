#!/bin/bash
# Run on the GPU box via gpurun: bench + rocprofv3 kernel trace + PMC passes; everything under gpurun_out/.
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/$1
mkdir -p $OUT
python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/trace_bench.json 2> $OUT/trace.err
find $OUT/trace -name '*stats*' | head
for f in $(find $OUT/trace -name '*kernel_stats.csv'); do head -8 $f; done
