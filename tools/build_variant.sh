#!/bin/bash
# build_variant.sh NAME [-DJDSP_...=.. ...]  ->  build/variants/NAME.so: a complete libjdsp.so with extra flags on top of
# the common and per-file ones (csrc/build_flags.txt), objects under build/obj_NAME/
# (timing-only A/B builds; tools/tune_stft.py, tools/cold_input_probe.py and JDSP_LIB= load them).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$R/build/variants"
cd "$R" && python - "$name" "$@" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
name, extra = sys.argv[1], sys.argv[2:]
g.build_hip(force=True, extra=extra, lib=os.path.join(g.ROOT, "build", "variants", name + ".so"),
            obj_dir=os.path.join(g.ROOT, "build", "obj_" + name))
print("built build/variants/%s.so (%s)" % (name, " ".join(extra)))
PY
