#!/usr/bin/env python3
"""Generates the "current results" text of BASELINE.md and README.md from ONE bench.py line, so that the documents cannot
quote numbers the bench does not print (round 3's tables contradicted the driver's bench line after a late regression).

    python tools/gen_results_md.py gpurun_out/<bench line>.json [label]

Rewrites the block between `<!-- results:begin -->` and `<!-- results:end -->` in BASELINE.md and README.md.  Every number
below comes from the JSON; the prose around the numbers is fixed text."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def f(x, n=1):
    return f"{x:.{n}f}"


def load(path):
    """The FULL record: bench.py (round 5) prints it first, as {"bench_extras": {...}}, and the contract's <= 6 KB line last;
    --extras-file writes the full record alone; older files hold one line with everything."""
    text = open(path).read().strip()
    lines = [l for l in text.splitlines() if l.startswith("{")]
    for line in lines:
        d = json.loads(line)
        if "bench_extras" in d:
            return d["bench_extras"]
    if lines:
        return json.loads(lines[-1])
    raise SystemExit(f"no JSON line in {path}")


def baseline_block(d, src):
    r, dec, x = d["roofline"], d.get("decode"), d.get("k1_extras", {})
    cpu = d.get("cpu_baseline", {})
    rows = []
    enc = (f"**{f(d['ms_per_step'], 2)} ms per batch = {f(d['value'] / 1e9, 1)} G samples/s** (kernel {f(r['kernel_ms_avg'], 2)} ms; "
           f"{f(r['achieved'], 1)} GB/s algorithmic = {f(100 * r['frac'], 2)} % of the HBM peak; traffic "
           f"{'%.4f GB' % (r['traffic'] / 1e9) if r.get('traffic') else 'n/a (' + str(r.get('traffic_note', ''))[:40] + ')'}")
    vi = r.get("valu_issue", {})
    if vi.get("valu_per_slice"):
        enc += f"; {f(vi['valu_per_slice'], 1)} VALU per slice, {f(100 * vi['frac'], 1)} % of the chip's VALU issue peak"
    enc += f"; {vi.get('wavefronts')} compute wavefronts x {vi.get('chains_per_wavefront')} chains, helper wavefronts: {vi.get('helper_wavefronts')})"
    if "value" in cpu:
        enc += (f".  CPU (oracle ref_style, same clips, same run): 1 thread {f(cpu['value'] / 1e6, 1)} M samples/s -> "
                f"**{f(d['speedup_vs_cpu_1_thread'], 0)}x**")
        if "all_cores_value" in cpu:
            enc += (f"; all {cpu['all_cores_workers']} usable CPUs (of {cpu.get('cores_visible')} visible; cgroup quota {cpu.get('cgroup_cpu_quota')}) "
                    f"{f(cpu['all_cores_value'] / 1e6, 0)} M -> {f(d['value'] / cpu['all_cores_value'], 0)}x (scaling efficiency "
                    f"{f(cpu['scaling_efficiency'], 2)})")
    rows.append(("3 (headline): " + d["config"]["workload"], enc + f"; bit-exact vs oracle: {d.get('bit_exact_vs_oracle')}"))
    if dec:
        rr = dec["roofline"]
        row = (f"**{f(dec['kernel_ms_avg'], 3)} ms** median over {len(dec['kernel_ms_per_placement'])} "
               f"buffer placements ({', '.join(f(v, 3) for v in dec['kernel_ms_per_placement'])}) = {f(dec['value'] / 1e12, 2)} T samples/s = "
               f"**{f(rr['frac'], 3)} of the HBM peak** (best placement {f(rr['frac_best_placement'], 3)}, worst {f(rr['frac_worst_placement'], 3)})")
        if rr.get("traffic"):
            row += (f"; PMC traffic {rr['traffic'] / 1e9:.3f} GB = {rr['traffic'] / rr['algorithmic_bytes_per_launch']:.3f} x the algorithmic bytes")
        if "decode_value" in cpu:
            row += (f".  CPU (oracle decode_all, the same streams, 1 thread): {f(cpu['decode_value'] / 1e6, 0)} M samples/s -> "
                    f"{f(dec['value'] / cpu['decode_value'], 0)}x; PCM equal: {cpu.get('gpu_pcm_equals_cpu_pcm')}")
        rows.append(("3, decode half of the round trip", row))
    k2 = d.get("k2_extras", {})
    if "mono" in k2:
        m = k2["mono"]
        rows.append(("2's shape, decode (" + m["workload"] + ")", f"{f(m['kernel_ms_avg'], 3)} ms median = {f(m['roofline']['frac'], 3)} of the HBM peak "
                     f"({', '.join(f(v, 3) for v in m['kernel_ms_per_placement'])}); bit-exact: {m['bit_exact_vs_oracle']}"))
    for name, g in k2.get("general", {}).items():
        rows.append((f"decode, {name}: {g['workload']}", f"{f(g['kernel_ms_avg'], 3)} ms = {f(g['roofline']['frac'], 3)} of the HBM peak; "
                     f"bit-exact: {g['bit_exact_vs_oracle']}, header mismatches {g['header_mismatches']}"))
    cfg = d.get("configs", {})
    if "cfg2_mono_encode" in cfg:
        c = cfg["cfg2_mono_encode"]
        rows.append(("2: " + c["workload"], f"{f(c['kernel_ms'], 2)} ms = {f(c['samples_per_s'] / 1e9, 1)} G samples/s ({f(c['ns_per_slice'], 0)} ns per slice); "
                     f"bit-exact: {c['bit_exact_vs_oracle']}"))
    if "cfg5_stream" in cfg:
        c = cfg["cfg5_stream"]
        rows.append(("5: " + c["workload"], f"**{f(c['ms_per_call'], 3)} ms per `encode_frame` call** = {f(c['x_realtime'], 0)}x real time; "
                     f"bit-exact vs the oracle's one-shot bytes: {c['bit_exact_vs_oracle']}"))
    if "full_chip" in x:
        c = x["full_chip"]
        rows.append(("K1 with every SIMD busy: " + c["workload"], f"{f(c['kernel_ms_avg'], 2)} ms = {f(c['samples_per_s'] / 1e9, 1)} G samples/s"))
    e2e = d.get("e2e", {})
    if "encode" in e2e:
        rows.append(("3 end to end: " + e2e["workload"],
                     f"encode {f(e2e['encode']['wall_ms'], 1)} ms = **{f(e2e['encode']['samples_per_s'] / 1e9, 1)} G samples/s** "
                     f"({f(e2e['encode']['host_to_device_GBps'], 1)} GB/s of PCM over PCIe: the bound), decode {f(e2e['decode']['wall_ms'], 1)} ms = "
                     f"{f(e2e['decode']['samples_per_s'] / 1e9, 1)} G samples/s; bytes / PCM equal to the device-resident path: "
                     f"{e2e['encode']['bytes_equal_device_path']} / {e2e['decode']['pcm_equal_device_path']} (PCIe-inclusive: never `value`)"))
    out = [f"Generated by `tools/gen_results_md.py` from `{src}` (one `python bench.py` run on one MI355X; host: "
           f"{cpu.get('cores_visible', '?')} hardware threads visible, {cpu.get('cores_usable', '?')} usable by the bench process).", "",
           "| Config | GPU result |", "|---|---|"]
    out += [f"| {a} | {b} |" for a, b in rows]
    sc = x.get("signal_classes", {})
    if sc:
        out += ["", f"Signal classes ({sc.get('workload')}; the batch waits for its slowest wavefront):", "",
                "| input | kernel ms | ns per slice | vs music | bit-exact | which forms it reached (counters of a counting plan) | per wavefront: cycles per slice p50 / max (bound), the slowest one's form mix |", "|---|---|---|---|---|---|---|"]
        for k in ("music", "silence", "fixture", "white", "loud"):
            if k not in sc:
                continue
            c, cn = sc[k], sc[k]["counters"]
            forms = (f"spec misses {cn['speculation_misses']}, cheap-quotient declines {cn['cheap_quotient_declines']}, penalty-form declines "
                     f"{cn.get('penalty_form_declines', 0)}, packed declines {cn['declined_slices']}, wave-slices on the wide form "
                     f"{cn.get('wavefront_slices_on_wide_form', 0)} / on the 32-bit step {cn.get('wavefront_slices_on_32bit_step', 0)} "
                     f"of {cn['slices_total'] // 4}")
            pw = c.get("per_wavefront")
            pws = (f"{f(pw['cycles_per_slice_p50'], 0)} / {f(pw['cycles_per_slice_max'], 0)} ({f(pw['bound_cycles_per_slice'], 0)}); "
                   + ", ".join(f"{kk} {vv}" for kk, vv in pw["slowest_wavefront_form_mix"].items() if kk != "blocks")) if pw else ""
            out.append(f"| {k} | {f(c['kernel_ms'], 2)} | {f(c['ns_per_slice'], 0)} | {f(c.get('vs_music', 1.0), 3)} | {c['bit_exact_vs_oracle']} | {forms} | {pws} |")
    sl = x.get("small_launch", {})
    if sl:
        out += ["", "Small launches (" + sl.get("workload", "") + "): " + ", ".join(
            f"{k.replace('_', ' ')} {f(v['ns_per_slice'], 0)} ns per slice ({v['wavefronts']} x {v['chains_per_wavefront']}, helper {v['helper_wavefronts']})"
            for k, v in sl.items() if isinstance(v, dict)) + "."]
    return "\n".join(out)


def readme_block(d, src):
    dec, x, cfg = d.get("decode"), d.get("k1_extras", {}), d.get("configs", {})
    sc = x.get("signal_classes", {})
    cpu = d.get("cpu_baseline", {})
    s = (f"Measured on one MI355X (`{src}`; method and evidence: `BASELINE.md` §4, `DESIGN.md`, `profiles/`): config 3 (1024 stereo clips x "
         f"256 frames, PCM resident in HBM) encodes at **{f(d['value'] / 1e9, 1)} G samples/s** ({f(d['ms_per_step'], 1)} ms per batch")
    if "value" in cpu:
        s += f"; {f(d['speedup_vs_cpu_1_thread'], 0)}x one host thread running the same algorithm"
        if "all_cores_value" in cpu:
            s += f", {f(d['value'] / cpu['all_cores_value'], 0)}x the {cpu['all_cores_workers']} CPUs the bench process may use"
    if "full_chip" in x:
        s += f"; {f(x['full_chip']['samples_per_s'] / 1e9, 0)} G samples/s once a batch has 4 wavefronts per SIMD"
    s += ")"
    if dec:
        s += (f" and decodes in {f(dec['kernel_ms_avg'], 2)} ms (median over {len(dec['kernel_ms_per_placement'])} buffer placements: "
              f"{f(dec['value'] / 1e12, 2)} T samples/s, {f(dec['roofline']['frac'], 2)} of the HBM peak)")
    s += "."
    if sc and "music" in sc:
        parts = [f"{k} {f(sc[k]['vs_music'], 2)}x" for k in ("fixture", "white", "loud") if k in sc and "vs_music" in sc[k]]
        s += f"  Against the synthetic music class ({f(sc['music']['ns_per_slice'], 0)} ns per slice): " + ", ".join(parts) + "."
    if "cfg5_stream" in cfg:
        s += f"  The frame-at-a-time API on one 8-channel stream: {f(cfg['cfg5_stream']['ms_per_call'], 2)} ms per call."
    e2e = d.get("e2e", {})
    if "encode" in e2e:
        s += (f"  From / to pinned host memory (one call of the host pipeline per direction): {f(e2e['encode']['samples_per_s'] / 1e9, 1)} / "
              f"{f(e2e['decode']['samples_per_s'] / 1e9, 1)} G samples/s, PCIe-bound.")
    return s


def patch(path, block):
    text = open(path).read()
    a, b = "<!-- results:begin -->", "<!-- results:end -->"
    assert a in text and b in text, f"{path}: markers missing"
    i, j = text.index(a) + len(a), text.index(b)
    open(path, "w").write(text[:i] + "\n" + block + "\n" + text[j:])


def main():
    src = sys.argv[1]
    d = load(src)
    label = sys.argv[2] if len(sys.argv) > 2 else os.path.relpath(src, ROOT)
    patch(os.path.join(ROOT, "BASELINE.md"), baseline_block(d, label))
    patch(os.path.join(ROOT, "README.md"), readme_block(d, label))
    print(baseline_block(d, label))


if __name__ == "__main__":
    main()
