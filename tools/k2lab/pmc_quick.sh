#!/bin/bash
# quick SQ-level PMC on lab binaries: pmc_quick.sh <nch> <binary>...
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
NCH=$1; shift
for b in "$@"; do
OUT=gpurun_out/pq_$b
mkdir -p $OUT
i=0
for grp in \
  "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
  "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_THREAD_CYCLES_VALU" ; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- tools/k2lab/bin/$b $NCH 1024 256 2 > $OUT/p$i.log 2>&1 || echo "pass $i failed: $grp"
done
python3 - "$OUT" "$b" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list); dur = []
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'uniform' in r['Kernel_Name'] or 'frames_kernel' in r['Kernel_Name'] and 'qoa_lab' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob(out + '/p*/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'qoa_lab' in r['Kernel_Name'] and 'decode' in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
print("==", sys.argv[2], "kernel ms under PMC: min %.3f avg %.3f" % (min(dur), sum(dur) / len(dur)))
for c, v in sorted(agg.items()):
    print(f"  {c:32s} {sum(v)/len(v):.5g}")
PY
done
