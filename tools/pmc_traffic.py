#!/usr/bin/env python3
"""Turns a tools/pmc.sh summary (gpurun_out/<tag>/summary.txt) into profiles/pmc_traffic.json, the per-launch HBM traffic
bench.py reports as roofline.traffic for config 3.  FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (128-byte read requests tallied at 64).  The sha256 of the
kernel sources the profile was taken on is stored with it: bench.py refuses the number (traffic: null + traffic_stale)
when the sources have changed since.

    python tools/pmc_traffic.py gpurun_out/<tag>/summary.txt profiles/<committed copy of the summary>
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse(path):
    sec, out = None, {}
    for line in open(path):
        m = re.match(r"== (\S+)", line)
        if m:
            sec = out.setdefault(m.group(1), {})
            continue
        m = re.match(r"\s+(\S+)\s+n=\s*\d+\s+mean=(\S+)", line)
        if m and sec is not None:
            sec[m.group(1)] = float(m.group(2))
    return out


def main():
    import bench
    summary, committed = sys.argv[1], sys.argv[2]
    d = parse(summary)
    dec = d.get("decode_uniform") or d.get("decode_frames")
    res = {"workload": "config3: 1024 clips x 2 ch x 256 frames", "source": committed}
    for name, c in (("encode", d["encode"]), ("decode", dec)):
        rd, wr = int(c["FETCH_SIZE"]) * 1024 * 2, int(c["WRITE_SIZE"]) * 1024
        res[name] = {"fetch_size_kb": int(c["FETCH_SIZE"]), "write_size_kb": int(c["WRITE_SIZE"]), "read_bytes_corrected": rd,
                     "write_bytes": wr, "traffic_bytes_per_launch": rd + wr, "source_sha256": bench.kernel_source_sha256(name)}
    # K1's VALU wave-instructions per slice and wavefront for the step form config 3 runs (cheap-quotient register form):
    # bench.py's valu_issue roofline reads it from here, tied to the same source hash as the traffic
    enc = d["encode"]
    if "SQ_INSTS_VALU" in enc and "SQ_WAVES" in enc:
        # per COMPUTE wavefront (config 3: 2048 chains, 4 per wavefront); SQ_WAVES also counts the helper wavefronts of the
        # launch, whose few instructions are part of what a slice costs
        compute_waves = 2048 // 4
        res["encode"]["valu_per_slice"] = {"cheap_quotient": enc["SQ_INSTS_VALU"] / (compute_waves * 256 * 256)}
        res["encode"]["wavefronts_launched"] = int(enc["SQ_WAVES"])
        if "SQ_WAIT_ANY" in enc and "SQ_WAVE_CYCLES" in enc and enc["SQ_WAVES"] == compute_waves:
            res["encode"]["sq_wait_any_frac"] = enc["SQ_WAIT_ANY"] / enc["SQ_WAVE_CYCLES"]   # (meaningless with helpers at barriers)
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
