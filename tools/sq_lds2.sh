#!/bin/bash
# LDS-pipe counters over one chain (GPU box; counters only).  usage: tools/sq_lds2.sh <bench_chains --only list>
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/sq_lds2
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL \
    --output-format csv -d "$O/p1" -- python3 "$R/tools/bench_chains.py" --only "${1:-mfcc}" --iters 3 > "$O/p1.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INSTS_LDS_ATOMIC_BANDWIDTH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES \
    --output-format csv -d "$O/p2" -- python3 "$R/tools/bench_chains.py" --only "${1:-mfcc}" --iters 3 > "$O/p2.log" 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2"):
    for f in glob.glob(f"{O}/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "jdsp::" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{O}/summary.txt", "w") as out:
    for k in sorted(agg):
        a = {c: sum(v) / len(v) for c, v in agg[k].items()}
        out.write(k + ": " + ", ".join("%s=%.0f" % (c, a[c]) for c in sorted(a)) + "\n")
print(open(f"{O}/summary.txt").read())
PY
