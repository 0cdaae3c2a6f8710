#!/usr/bin/env python3
"""Copies the summaries of gpurun_out/final (tools/collect_profiles.sh) into profiles/ under the given name,
refreshes profiles/stft_pmc_traffic.json from the two PMC passes and rewrites the figures table of DESIGN.md section 5
(between the FIGURES markers) from that collection.

    python tools/copy_profiles.py r03      ->  profiles/r03_bench_n1.json, r03_stft_bench_kernel_stats.csv, r03_chains.jsonl, ...
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "final")
P = os.path.join(ROOT, "profiles")
ALG = 603979776


def newest(pat):
    fs = glob.glob(os.path.join(O, pat))
    fs.sort(key=os.path.getmtime)                 # gpurun merges into the local directory: keep the latest run's file
    return fs[-1]


def per_kernel(name):
    """average counter value (KiB) per launch of every jdsp kernel, 65,536-frame launches only"""
    f = newest(f"pmc_{name}/runc/*counter_collection.csv")
    acc = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "stft1024_hop512_kernel" in k or "pcm_touch_kernel" in k:
            acc.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}


def main():
    tag = sys.argv[1]
    fetch, write = per_kernel("fetch"), per_kernel("write")
    # bench.py runs three legs: the reported one (read pass + stft<1>) and two without the pass (stft<2>)
    def pick(d, sub):
        for k, v in d.items():
            if sub in k:
                return v
        return (0, 0.0)
    t_f, s1_f, s2_f = pick(fetch, "pcm_touch"), pick(fetch, "kernel<1>"), pick(fetch, "kernel<2>")
    t_w, s1_w, s2_w = pick(write, "pcm_touch"), pick(write, "kernel<1>"), pick(write, "kernel<2>")
    kib = 1024.0
    d = {
        "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --steps 12 --warmup 6 --no-cpu-baseline",
        "what": "one reported step = pcm_touch_kernel (the read pass) + stft1024_hop512_kernel<1>; FETCH_SIZE / WRITE_SIZE are the "
                "L2's memory-side request counters and count Infinity-Cache hits too (MI355X_MICROARCH.md, HBM), so the transform's "
                "own 64 MiB of reads -- served by the Infinity Cache behind the read pass -- appear in l2_memory_side_bytes_per_step "
                "but not in hbm_bytes_per_step",
        "correction": "FETCH_SIZE doubled (gfx950: exactly half of a wide coalesced streaming read is reported), WRITE_SIZE as is; counters in KiB",
        "launches_averaged": {"pcm_touch_kernel": [t_f[0], t_w[0]], "stft1024_hop512_kernel<1>": [s1_f[0], s1_w[0]],
                              "stft1024_hop512_kernel<2> (legs without the read pass)": [s2_f[0], s2_w[0]]},
        "read_pass_fetch_bytes": 2 * t_f[1] * kib, "read_pass_write_bytes": t_w[1] * kib,
        "transform_fetch_bytes": 2 * s1_f[1] * kib, "transform_write_bytes": s1_w[1] * kib,
        "transform_without_read_pass_fetch_bytes": 2 * s2_f[1] * kib, "transform_without_read_pass_write_bytes": s2_w[1] * kib,
        "algorithmic_bytes_per_launch": ALG,
        "collected_by": "tools/collect_profiles.sh (two separate rocprofv3 --pmc passes), tools/copy_profiles.py",
    }
    d["hbm_bytes_per_launch"] = d["read_pass_fetch_bytes"] + d["read_pass_write_bytes"] + d["transform_write_bytes"]
    d["l2_memory_side_bytes_per_step"] = d["hbm_bytes_per_launch"] + d["transform_fetch_bytes"]
    d["traffic_over_algorithmic"] = d["hbm_bytes_per_launch"] / ALG
    json.dump(d, open(os.path.join(P, "stft_pmc_traffic.json"), "w"), indent=1)
    pre = tag if tag.startswith("r0") else "r02_" + tag      # "r03" -> r03_*, legacy tags "a".."d" -> r02_*_<tag>
    name = (lambda stem, ext: os.path.join(P, f"{pre}_{stem}.{ext}")) if tag.startswith("r0") else \
        (lambda stem, ext: os.path.join(P, f"r02_{stem}_{tag}.{ext}"))
    shutil.copy(os.path.join(O, "bench.json"), name("bench_n1", "json"))
    shutil.copy(newest("prof_bench/runc/*kernel_stats.csv"), name("stft_bench_kernel_stats", "csv"))
    shutil.copy(os.path.join(O, "chains.jsonl"), name("chains", "jsonl"))
    if os.path.exists(os.path.join(O, "chains_warm.jsonl")):
        shutil.copy(os.path.join(O, "chains_warm.jsonl"), name("chains_cache_resident_inputs", "jsonl"))
    shutil.copy(newest("prof_chains/runc/*kernel_stats.csv"), name("chains_kernel_stats", "csv"))
    if os.path.exists(os.path.join(O, "bench_mfcc10k.json")):
        shutil.copy(os.path.join(O, "bench_mfcc10k.json"), name("bench_mfcc10k_n1", "json"))
    b = json.load(open(os.path.join(O, "bench.json")))
    print("bench: %.1f M frames/s, %.2f us per step, frac %.3f; warm %.2f us; PMC hbm %.1f MB (%.4fx), L2 memory side %.1f MB" % (
        b["value"] / 1e6, b["ms_per_step"] * 1e3, b["roofline"]["frac"], b["roofline"]["warm_input"]["kernel_ms"] * 1e3,
        d["hbm_bytes_per_launch"] / 1e6, d["traffic_over_algorithmic"], d["l2_memory_side_bytes_per_step"] / 1e6))
    kstats = {}
    for r in csv.DictReader(open(name("stft_bench_kernel_stats", "csv"))):
        if "stft1024" in r["Name"] or "pcm_touch" in r["Name"]:
            kstats[r["Name"].split("(")[0].replace("void ", "").replace("jdsp::", "")] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
            print("rocprof: %-40s %s calls, %.1f us average" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3))
    cold = [json.loads(l) for l in open(name("chains", "jsonl"))]
    warm = {}
    if os.path.exists(name("chains_cache_resident_inputs", "jsonl")):
        warm = {c["chain"]: c for c in (json.loads(l) for l in open(name("chains_cache_resident_inputs", "jsonl")))}
    for c in cold:
        print("  %-50s %8.1f us %9.1f M/s" % (c["chain"], c["ms"] * 1e3, c["rate_per_s"] / 1e6))
    write_figures(pre, b, kstats, d, cold, warm)


def write_figures(pre, b, kstats, pmc, cold, warm):
    """ONE table per collection, generated -- nothing typed by hand."""
    path = os.path.join(ROOT, "DESIGN.md")
    txt = open(path).read()
    a, z = "<!-- FIGURES:BEGIN -->", "<!-- FIGURES:END -->"
    if a not in txt or z not in txt:
        return
    r = b["roofline"]
    L = []
    L.append("**Collection %s** (one MI355X box, `tools/collect_profiles.sh` + `tools/copy_profiles.py`; files `profiles/%s_*`)." % (pre, pre))
    L.append("")
    L.append("Headline (`bench.py`, defaults; `profiles/%s_bench_n1.json`):" % pre)
    L.append("")
    L.append("| leg | us per 65,536-frame step | frames/s | algorithmic GB/s | of 8 TB/s |")
    L.append("|---|---|---|---|---|")
    L.append("| **reported**: read pass + transform, input rotated over 6 buffers (cold) | %.1f | %.1f M | %.0f | **%.3f** |"
             % (r["kernel_ms"] * 1e3, b["value"] / 1e6, r["achieved"], r["frac"]))
    w, c0 = r.get("warm_input"), r.get("cold_input_without_read_pass")
    if w:
        L.append("| one buffer re-read, no read pass (input in the Infinity Cache) | %.1f | %.1f M | %.0f | %.3f |"
                 % (w["kernel_ms"] * 1e3, w["value"] / 1e6, w["achieved"], w["frac"]))
    if c0:
        L.append("| rotated input WITHOUT the read pass | %.1f | %.1f M | %.0f | %.3f |"
                 % (c0["kernel_ms"] * 1e3, c0["value"] / 1e6, c0["achieved"], c0["frac"]))
    f = r.get("fp64")
    if f:
        L.append("| `roofline.fp64`: the reference's FP64, complex128 out (17,408 B per frame), cold | %.1f | %.1f M | %.0f | **%.3f** |"
                 % (f["kernel_ms"] * 1e3, f["value"] / 1e6, f["achieved"], f["frac"]))
    L.append("")
    if kstats:
        L.append("rocprofv3 `--kernel-trace --stats` of the same command (`profiles/%s_stft_bench_kernel_stats.csv`): " % pre +
                 "; ".join("`%s` %.1f us average over %d launches" % (k, v[1], v[0]) for k, v in sorted(kstats.items())) +
                 " (the profiler's per-dispatch overhead sits on the short kernels).")
    L.append("PMC (`profiles/stft_pmc_traffic.json`): %.1f MB from/to HBM per reported step = %.4fx the algorithmic bytes."
             % (pmc["hbm_bytes_per_launch"] / 1e6, pmc["traffic_over_algorithmic"]))
    cb = b.get("cpu_baseline")
    if cb:
        L.append("CPU baseline (the FP64 oracle on the box's host): %.1f k frames/s on one thread; %.2f M frames/s on %d threads."
                 % (cb["value"] / 1e3, cb["all_cores"]["value"] / 1e6, cb["all_cores"]["threads"]))
    L.append("")
    L.append("Chains (`tools/bench_chains.py`; 65,536 units per call unless the name says otherwise; `profiles/%s_chains*.jsonl`):" % pre)
    L.append("")
    L.append("| chain | us, input from HBM | us, cache-resident input | units/s (HBM) |")
    L.append("|---|---|---|---|")
    for c in cold:
        wv = warm.get(c["chain"])
        L.append("| `%s` | %.1f | %s | %.1f M |" % (c["chain"], c["ms"] * 1e3, "%.1f" % (wv["ms"] * 1e3) if wv else "—", c["rate_per_s"] / 1e6))
    new = txt[:txt.index(a) + len(a)] + "\n" + "\n".join(L) + "\n" + txt[txt.index(z):]
    open(path, "w").write(new)
    print("DESIGN.md: figures table rewritten from collection", pre)


if __name__ == "__main__":
    main()
