#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_membench
mkdir -p $OUT
i=0
for grp in "TCC_REQ_sum TCC_WRITE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_WRITE_REQ_sum TCP_TOTAL_WRITE_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- tools/ubench/membench > $OUT/p$i.log 2>&1 || echo "pass $i failed: $grp"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(dict)
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        rows[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
        rows[int(r['Dispatch_Id'])]['k'] = r['Kernel_Name'][:40]
with open(out + '/summary.txt', 'w') as fo:
    for d in sorted(rows):
        fo.write(f"{d:3d} " + " ".join(f"{k}={v:.4g}" if k != 'k' else v for k, v in sorted(rows[d].items())) + "\n")
print(open(out + '/summary.txt').read())
PY
