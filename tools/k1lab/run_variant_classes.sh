# signal classes with the product and with lab libraries: bash tools/k1lab/run_variant_classes.sh <lib name> ...
cd "$GRAFT_REPO_ROOT"
for l in "" "$@"; do echo "== ${l:-product}"; QOAUDIO_AMD_LIB=${l:+$PWD/tools/k1lab/bin/lib_$l.so} python bench.py --no-cpu --no-decode --steps 3 --warmup 1 2>/dev/null | python -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("config 3", round(d["ms_per_step"],2), d["bit_exact_vs_oracle"])
for k,v in d["k1_extras"]["signal_classes"].items():
    if isinstance(v,dict): c=v["counters"]; print(" ", k, round(v["kernel_ms"],2), v["bit_exact_vs_oracle"], "spec misses", c["speculation_misses"], "wide declines", c["wide_form_declines"], "on wide", c["wavefront_slices_on_wide_form"], "on 32-bit", c["wavefront_slices_on_32bit_step"])'; done
