# do the kernels of two shards on ONE device overlap, and how long does each take?  (bench.py --abi-multi --duplicate-devices)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf /tmp/dup_trace; mkdir -p /tmp/dup_trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/dup_trace -- python3 bench.py --abi-multi --gpus 2 --duplicate-devices --clips 512 --steps 2 --warmup 1 > /tmp/dup_bench.json 2> /tmp/dup.err
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/dup_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'qoa_encode' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
print(len(rows), 'encode launches; first 12 (start us, end us, duration us, queue / stream id):')
for r in rows[:12]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"  {s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {r.get('Queue_Id', '?')} {r.get('Stream_Id', '?')}")
ov = sum(1 for a, b in zip(rows, rows[1:]) if int(b['Start_Timestamp']) < int(a['End_Timestamp']))
print('consecutive launches that overlap in time:', ov, 'of', len(rows) - 1)
PY
