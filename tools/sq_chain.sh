#!/bin/bash
# Counter pass over one chain (usage: tools/sq_chain.sh <bench_chains --only list>) (GPU box; counters only): wave cycles, issue / wait split, instruction
# counts and GRBM_GUI_ACTIVE (busy cycles: with the dispatch's start/end timestamps it gives the engine clock).
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/sq_chain
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
    --output-format csv -d "$O/p1" -- python3 "$R/tools/bench_chains.py" --only "${1:-denoise}" --iters 3 ${DENOISE_K:+--denoise-k $DENOISE_K} > "$O/p1.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_SALU \
    --output-format csv -d "$O/p2" -- python3 "$R/tools/bench_chains.py" --only "${1:-denoise}" --iters 3 ${DENOISE_K:+--denoise-k $DENOISE_K} > "$O/p2.log" 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for p in ("p1", "p2"):
    for f in glob.glob(f"{O}/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "jdsp::" not in k:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
with open(f"{O}/summary.txt", "w") as out:
    for k in sorted(agg):
        a = {c: sum(v) / len(v) for c, v in agg[k].items()}
        line = k + ": " + ", ".join("%s=%.0f" % (c, a[c]) for c in sorted(a))
        if dur[k]:
            d = sum(dur[k]) / len(dur[k])
            line += ", duration_ns_under_counters=%.0f" % d
        out.write(line + "\n")
print(open(f"{O}/summary.txt").read())
PY
