#!/bin/bash
# Runs on the GPU box (gpurun -- bash tools/collect_profiles.sh): the full GPU test suite, smoke(), the
# headline bench, its rocprofv3 kernel summary, the two PMC passes for HBM traffic (separate runs, counters
# only), and the chain benchmarks (cold and cache-resident inputs) with their kernel summary.  Everything lands in
# gpurun_out/final/; `python tools/copy_profiles.py r03` then copies the summaries into profiles/r03_* and regenerates DESIGN.md's figures table.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final
rm -rf "$O"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout -k 10 900 python -m pytest "$R/tests" -m gpu -x -q 2>&1 | tail -3 | tee "$O/pytest.log"
echo "== smoke"; (cd "$R" && timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()") 2>&1 | grep smoke | tee "$O/smoke.log"
echo "== bench"; (cd "$R" && timeout -k 10 300 python bench.py) 2>/dev/null | tail -1 > "$O/bench.json"; cut -c1-300 "$O/bench.json"
echo "== bench --workload mfcc10k"; (cd "$R" && timeout -k 10 300 python bench.py --workload mfcc10k --steps 100 --warmup 20) 2>/dev/null | tail -1 > "$O/bench_mfcc10k.json"; cut -c1-300 "$O/bench_mfcc10k.json"
echo "== rocprof bench"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_bench" -- python3 "$R/bench.py" --no-cpu-baseline > "$O/prof_bench.log" 2>&1
echo "== pmc fetch"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 "$R/bench.py" --steps 12 --warmup 6 --no-cpu-baseline > "$O/pmc_fetch.log" 2>&1
echo "== pmc write"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 "$R/bench.py" --steps 12 --warmup 6 --no-cpu-baseline > "$O/pmc_write.log" 2>&1
echo "== chains (cold inputs)"; (cd "$R" && timeout -k 10 600 python tools/bench_chains.py) 2>/dev/null > "$O/chains.jsonl"; wc -l "$O/chains.jsonl"
echo "== chains (cache-resident inputs)"; (cd "$R" && timeout -k 10 600 python tools/bench_chains.py --warm --only denoise,mfcc,fastconv,pitch,mvdr) 2>/dev/null > "$O/chains_warm.jsonl"; wc -l "$O/chains_warm.jsonl"
echo "== rocprof chains"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_chains" -- python3 "$R/tools/bench_chains.py" --iters 8 > "$O/prof_chains.log" 2>&1
find "$O" -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head
echo done
