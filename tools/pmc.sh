#!/bin/bash
# PMC passes for the two hot kernels (run on the GPU box through gpurun).  One counter group per pass,
# --kernel-trace only (no other trace domains), as the microarch guide prescribes.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
CMD="tools/prof_driver 1024 256 2 2 1"
i=0
for grp in \
  "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
  "GRBM_GUI_ACTIVE FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1
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
            fo.write(f"  {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open(out + '/summary.txt').read())
PY
