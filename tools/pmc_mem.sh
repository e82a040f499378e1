#!/bin/bash
# Memory-path PMC passes for K2 (run on the GPU box): where do the decode kernel's stores wait?
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-pmc_mem}
mkdir -p $OUT
CMD="tools/prof_driver 1024 256 2 1 1"
i=0
# few counters per pass (TCC has 4 counter slots per channel: a larger request aborts the profiler), every pass bounded
for grp in \
  "TCC_EA0_WRREQ_STALL_sum TCC_CYCLE_sum" \
  "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
  "TCC_SRC_FIFO_FULL_sum TCC_TAG_STALL_sum" \
  "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" \
  "TCC_BUSY_sum TCC_REQ_sum" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
  "TCC_WRITE_sum TCC_STREAMING_REQ_sum" \
  "TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum" ; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed: $grp"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'qoa_' not in k: continue
        name = 'encode' if 'encode' in k else ('decode_uniform' if 'uniform' in k else ('decode_frames' if 'frames' in k else 'decode_generic'))
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fo:
    for k, d in agg.items():
        fo.write(f"== {k}\n")
        for c, v in sorted(d.items()):
            fo.write(f"  {c:40s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open(out + '/summary.txt').read())
PY
