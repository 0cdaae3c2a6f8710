#!/bin/bash
# A/B timing of build-time variants of libjdsp.so on the chain benchmarks (GPU box only):
#   bash tools/ab_chains.sh "<bench_chains --only list>" build/variants/a.so build/variants/b.so ...
only=$1; shift
for lib in "$@"; do
  echo "== $lib"
  JDSP_LIB=$PWD/$lib timeout -k 10 300 python tools/bench_chains.py --only "$only" 2>/dev/null \
    | python -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-44s %8.1f us' % (d['chain'], d['ms']*1e3))"
done
