#!/bin/bash
# A round's profile set (run on the GPU box through gpurun): PMC passes on the standalone driver, memory-path counters,
# and the rocprofv3 kernel trace of the very command the driver runs (bench.py).  Everything lands in gpurun_out/<tag>*;
# tools/refresh_profiles.py <round> <tag> copies what is committed under profiles/.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-prof}
bash tools/pmc.sh ${TAG}_pmc > gpurun_out/${TAG}_pmc.log 2>&1
bash tools/pmc_mem.sh ${TAG}_pmc_mem > gpurun_out/${TAG}_pmc_mem.log 2>&1
mkdir -p gpurun_out/${TAG}_trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras > gpurun_out/${TAG}_trace_bench.json 2> gpurun_out/${TAG}_trace.err
for f in $(find gpurun_out/${TAG}_trace -name '*kernel_stats.csv'); do head -6 $f; done
python3 bench.py --steps 5 --warmup 1 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.json
cat gpurun_out/${TAG}_pmc/summary.txt | head -60
