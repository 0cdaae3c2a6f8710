#!/usr/bin/env python3
"""Copies the summaries of gpurun_out/final (tools/collect_profiles.sh) into profiles/ under the given tag and
refreshes profiles/stft_pmc_traffic.json from the two PMC passes.

    python tools/copy_profiles.py a      ->  profiles/r02_bench_n1_a.json, r02_stft_bench_kernel_stats_a.csv, ...
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
    shutil.copy(os.path.join(O, "bench.json"), os.path.join(P, f"r02_bench_n1_{tag}.json"))
    shutil.copy(newest("prof_bench/runc/*kernel_stats.csv"), os.path.join(P, f"r02_stft_bench_kernel_stats_{tag}.csv"))
    shutil.copy(os.path.join(O, "chains.jsonl"), os.path.join(P, f"r02_chains_{tag}.jsonl"))
    if os.path.exists(os.path.join(O, "chains_warm.jsonl")):
        shutil.copy(os.path.join(O, "chains_warm.jsonl"), os.path.join(P, f"r02_chains_{tag}_cache_resident_inputs.jsonl"))
    shutil.copy(newest("prof_chains/runc/*kernel_stats.csv"), os.path.join(P, f"r02_chains_{tag}_kernel_stats.csv"))
    b = json.load(open(os.path.join(O, "bench.json")))
    print("bench: %.1f M frames/s, %.2f us per step, frac %.3f; warm %.2f us; PMC hbm %.1f MB (%.4fx), L2 memory side %.1f MB" % (
        b["value"] / 1e6, b["ms_per_step"] * 1e3, b["roofline"]["frac"], b["roofline"]["warm_input"]["kernel_ms"] * 1e3,
        d["hbm_bytes_per_launch"] / 1e6, d["traffic_over_algorithmic"], d["l2_memory_side_bytes_per_step"] / 1e6))
    for r in csv.DictReader(open(os.path.join(P, f"r02_stft_bench_kernel_stats_{tag}.csv"))):
        if "stft1024" in r["Name"] or "pcm_touch" in r["Name"]:
            print("rocprof: %-40s %s calls, %.1f us average" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3))
    for line in open(os.path.join(P, f"r02_chains_{tag}.jsonl")):
        c = json.loads(line)
        print("  %-50s %8.1f us %9.1f M/s" % (c["chain"], c["ms"] * 1e3, c["rate_per_s"] / 1e6))


if __name__ == "__main__":
    main()
