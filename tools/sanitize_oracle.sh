#!/bin/sh
# Host AddressSanitizer/UBSan pass over the CPU checker (GPU sanitizers are not available on the
# pool; the oracle is what the parity tests trust, so it is the piece worth sanitising).
set -e
cd "$(dirname "$0")/.."
gcc -O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -std=c11 -D_GNU_SOURCE \
    -shared -o /tmp/libjdsp_oracle_asan.so oracle/jdsp_oracle.c -lm
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
    python tools/sanitize_oracle.py
