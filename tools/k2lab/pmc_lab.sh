#!/bin/bash
# PMC passes on one lab binary (run on the GPU box): pmc_lab.sh <binary> <nch> <tag>
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
BIN=tools/k2lab/bin/$1; NCH=$2; TAG=${3:-pmc_$1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for grp in \
  "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
  "GRBM_GUI_ACTIVE FETCH_SIZE" \
  "WRITE_SIZE" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
  "TCC_REQ_sum TCC_WRITE_sum" ; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- $BIN $NCH 1024 256 2 > $OUT/p$i.log 2>&1 || echo "pass $i failed: $grp"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'decode' not in k: continue
        name = 'uniform' if 'uniform' in k else ('frames' if 'frames' in k else 'other')
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fo:
    for k, d in agg.items():
        fo.write(f"== {k}\n")
        for c, v in sorted(d.items()):
            fo.write(f"  {c:40s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open(out + '/summary.txt').read())
PY
