#!/usr/bin/env python3
"""Copies the summaries of gpurun_out/final (tools/collect_profiles.sh) into profiles/ under the given tag
and refreshes profiles/stft_pmc_traffic.json from the two PMC passes.

    python tools/copy_profiles.py v5      ->  profiles/r01_bench_n1_v5.json, r01_stft_bench_kernel_stats_v5.csv, ...
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


def newest(pat):
    fs = glob.glob(os.path.join(O, pat))
    fs.sort(key=os.path.getmtime)                 # gpurun merges into the local directory: keep the latest run's file
    return fs[-1]


def avg(name):
    f = newest(f"pmc_{name}/runc/*counter_collection.csv")
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "stft1024_hop512_kernel" in r["Kernel_Name"]]
    return len(vals), sum(vals) / len(vals)


def main():
    tag = sys.argv[1]
    nf, f = avg("fetch")
    nw, w = avg("write")
    p = os.path.join(P, "stft_pmc_traffic.json")
    d = json.load(open(p))
    d["launches_averaged"] = [nf, nw]
    d["FETCH_SIZE_KiB_per_launch"] = f
    d["WRITE_SIZE_KiB_per_launch"] = w
    d["hbm_read_bytes_per_launch"] = 2 * f * 1024
    d["hbm_write_bytes_per_launch"] = w * 1024
    d["hbm_bytes_per_launch"] = 2 * f * 1024 + w * 1024
    d["traffic_over_algorithmic"] = d["hbm_bytes_per_launch"] / d["algorithmic_bytes_per_launch"]
    json.dump(d, open(p, "w"), indent=1)
    shutil.copy(os.path.join(O, "bench.json"), os.path.join(P, f"r01_bench_n1_{tag}.json"))
    shutil.copy(newest("prof_bench/runc/*kernel_stats.csv"), os.path.join(P, f"r01_stft_bench_kernel_stats_{tag}.csv"))
    shutil.copy(os.path.join(O, "chains.jsonl"), os.path.join(P, f"r01_chains_{tag}.jsonl"))
    shutil.copy(newest("prof_chains/runc/*kernel_stats.csv"), os.path.join(P, f"r01_chains_{tag}_kernel_stats.csv"))
    b = json.load(open(os.path.join(O, "bench.json")))
    print("bench: %.1f M frames/s, %.2f us, frac %.3f; PMC %.1f MB (%.4fx)" % (
        b["value"] / 1e6, b["ms_per_step"] * 1e3, b["roofline"]["frac"], d["hbm_bytes_per_launch"] / 1e6, d["traffic_over_algorithmic"]))
    for r in csv.DictReader(open(os.path.join(P, f"r01_stft_bench_kernel_stats_{tag}.csv"))):
        if "stft1024" in r["Name"]:
            print("rocprof: %s calls, %.1f us average" % (r["Calls"], float(r["AverageNs"]) / 1e3))
    for line in open(os.path.join(P, f"r01_chains_{tag}.jsonl")):
        c = json.loads(line)
        print("  %-46s %8.1f us %9.1f M/s" % (c["chain"], c["ms"] * 1e3, c["rate_per_s"] / 1e6))


if __name__ == "__main__":
    main()
