#!/bin/bash
# L2 / fabric read counters of K2u for builds of the library: pmc_variants.sh <lib name>...
# The variant is given to the profiling driver through LD_LIBRARY_PATH (a per-variant directory holding a link named
# libqoaudio_amd.so; the driver's RUNPATH yields to it), NOT through LD_PRELOAD: a preloaded HIP library is loaded into every hop of
# the rocprofv3 launcher before the target starts (advisor, round 4).
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for l in product "$@"; do
  pre=""
  if [ $l != product ]; then
    mkdir -p "/tmp/k2lab_variant_$l"; ln -sf "$PWD/tools/k2lab/bin/lib_$l.so" "/tmp/k2lab_variant_$l/libqoaudio_amd.so"
    pre="/tmp/k2lab_variant_$l"
  fi
  for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    rm -rf /tmp/pv; LD_LIBRARY_PATH=$pre${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH} rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pv -- tools/prof_driver 1024 256 2 1 2 > /tmp/pv.log 2>&1
    python3 - "$l" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob('/tmp/pv/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'uniform' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[1], {k: sum(v) / len(v) for k, v in agg.items()})
PY
  done
  LD_LIBRARY_PATH=$pre${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH} tools/prof_driver 1024 256 2 1 6 | grep decode | tail -3
done
