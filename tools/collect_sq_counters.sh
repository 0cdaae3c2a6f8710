#!/bin/bash
# SQ counter passes over the chain benchmarks (GPU box; counters only, no tracing domains):
#   pass 1: where the wave cycles go (waiting on s_waitcnt / issue-stalled / issuing) and instruction counts
#   pass 2: LDS activity and bank conflicts, memory instruction counts
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/sq
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
    --output-format csv -d "$O/p1" -- python3 "$R/tools/bench_chains.py" --iters 2 > "$O/p1.log" 2>&1
timeout -k 10 500 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SMEM \
    --output-format csv -d "$O/p2" -- python3 "$R/tools/bench_chains.py" --iters 2 > "$O/p2.log" 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2"):
    for f in glob.glob(f"{O}/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "jdsp::" not in k:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
         "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
         "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES"]
with open(f"{O}/sq_counters_by_kernel.csv", "w") as out:
    out.write("kernel,launches," + ",".join(names) + "\n")
    for k in sorted(agg):
        n = max(len(v) for v in agg[k].values())
        out.write(k + "," + str(n) + "," + ",".join("%.0f" % (sum(agg[k][c]) / len(agg[k][c])) if agg[k][c] else "" for c in names) + "\n")
print(open(f"{O}/sq_counters_by_kernel.csv").read()[:6000])
PY
