timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r04i_gputests.txt 2>&1; tail -2 gpurun_out/r04i_gputests.txt
QOA_FUZZ_SCALE=60 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -1
python bench.py --no-cpu > gpurun_out/r04i_bench.json 2> gpurun_out/r04i_bench.err
python -c '
import json
d=json.loads(open("gpurun_out/r04i_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["bit_exact_vs_oracle"], {k:(round(v["kernel_ms"],2), v["bit_exact_vs_oracle"]) for k,v in d["k1_extras"]["signal_classes"].items() if isinstance(v,dict)})
print(d["k1_extras"]["full_chip"]["samples_per_s"]/1e9)
'
for n in 1 2; do python bench.py --abi-multi --gpus $n --duplicate-devices --clips $((1024 / n)) --steps 5 --warmup 2 2>/dev/null | python -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("abi-multi shards", d["n_gpus"], "clips per shard", d["config"]["clips_per_gpu"], "ms", round(d["ms_per_step"],2), "bit-exact", d["bit_exact_vs_oracle"])'; done
