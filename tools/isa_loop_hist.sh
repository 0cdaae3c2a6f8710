#!/bin/bash
# isa_loop_hist.sh FILE.hip KERNEL_SUBSTRING [-D...]: opcode histogram of the kernel's innermost-loop-bearing body (from the
# first "Loop Header" to s_endpgm) and its register counts -- what a kernel costs per unit before it ever runs.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
f=$1; k=$2; shift 2
mkdir -p "$R/build/isa"
out="$R/build/isa/$(basename "$f" .hip).s"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only "$@" -o "$out" "$R/$f" 2>/dev/null
s=$(grep -n "^_ZN[^ ]*$k[^ ]*:" "$out" | head -1 | cut -d: -f1)
e=$(awk -v s="$s" 'NR>s && /s_endpgm/ {print NR; exit}' "$out")
awk -v s="$s" -v e="$e" 'NR>=s && NR<=e' "$out" > "$R/build/isa/kernel.s"
grep -A8 "\.name: *_ZN[^ ]*$k" "$out" | grep "name\|vgpr_count\|spill" | head -4
l=$(grep -n "Loop Header" "$R/build/isa/kernel.s" | head -1 | cut -d: -f1)
echo "loop from line $l of build/isa/kernel.s"
awk -v l="$l" 'NR>=l' "$R/build/isa/kernel.s" | grep -v "^\s*;\|^\.L\|^\s*$" | awk '{print $1}' | sort | uniq -c | sort -rn | awk '{printf "%s %s, ", $1, $2} END {print ""}'
