#!/usr/bin/env python3
"""A/B harness for the STFT kernel's build-time variants (GPU box only).

    python tools/tune_stft.py build/variants/*.so [--fpw 8,16,32] [--rounds 5]

Every variant is a complete libjdsp.so built with different -DJDSP_STFT_* flags.
Variants are timed interleaved, round-robin, in one process (guide rule 24) with
HIP events on the stream the kernel runs on; prints median/min microseconds and
the algorithmic GB/s (9,216 B per frame)."""
import argparse
import ctypes as C
import os
import statistics
import sys

import numpy as np
import torch


def load(path):
    lib = C.CDLL(os.path.abspath(path))
    vp, i, l = C.c_void_p, C.c_int, C.c_long
    lib.jdsp_create.argtypes = [i, C.POINTER(vp)]
    lib.jdsp_set_stream.argtypes = [vp, vp]
    lib.jdsp_set_option.argtypes = [vp, C.c_char_p, l]
    lib.jdsp_stft_i16_dev.argtypes = [vp, vp, l, i, i, vp]
    lib.jdsp_last_error.restype = C.c_char_p
    lib.jdsp_last_error.argtypes = [vp]
    h = vp()
    rc = lib.jdsp_create(0, C.byref(h))
    assert rc == 0, lib.jdsp_last_error(None)
    lib.jdsp_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    return lib, h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--fpw", default="0")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--frames", type=int, default=65536)
    ap.add_argument("--sustain-ms", type=float, default=0.0,
                    help="also time every case under a sustained load: run it back to back for this long, "
                         "report the mean of the second half (power management settles after ~2 ms)")
    a = ap.parse_args()
    B = a.frames
    rng = np.random.default_rng(0)
    pcm = torch.from_numpy(np.clip(np.rint(rng.normal(0, 3000, 512 * (B + 1))), -32768, 32767).astype(np.int16)).cuda()
    spec = torch.empty((B, 1024), dtype=torch.complex64, device="cuda")
    ref = None
    cases = []
    for p in a.libs:
        lib, h = load(p)
        for fpw in [int(x) for x in a.fpw.split(",")]:
            cases.append((os.path.basename(p), fpw, lib, h))
    times = {(n, f): [] for n, f, _, _ in cases}

    def run(lib, h, fpw):
        lib.jdsp_set_option(h, b"stft.frames_per_wave", fpw)
        rc = lib.jdsp_stft_i16_dev(h, C.c_void_p(pcm.data_ptr()), B, 1024, 512, C.c_void_p(spec.data_ptr()))
        assert rc == 0, lib.jdsp_last_error(h)

    for n, f, lib, h in cases:          # warm + cross-check every variant against the first
        spec.zero_()
        run(lib, h, f)
        torch.cuda.synchronize()
        if ref is None:
            ref = spec.clone()
        else:
            d = (torch.view_as_real(spec) - torch.view_as_real(ref)).abs().max().item()
            s = torch.view_as_real(ref).abs().max().item()
            if d > 1e-5 * s:
                print("!! %s fpw=%d differs from first variant: %g (scale %g)" % (n, f, d, s))
    for _ in range(a.rounds):
        for n, f, lib, h in cases:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run(lib, h, f)
            e1.record()
            torch.cuda.synchronize()
            times[(n, f)].append(e0.elapsed_time(e1) / a.iters * 1e3)
    sustained = {}
    if a.sustain_ms > 0:
        for n, f, lib, h in cases:
            est = statistics.median(times[(n, f)]) * 1e-3            # ms per launch
            k = max(int(a.sustain_ms / est / 2), 10)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            evs[0].record()
            for _ in range(k):
                run(lib, h, f)
            evs[1].record()
            for _ in range(k):
                run(lib, h, f)
            evs[2].record()
            torch.cuda.synchronize()
            sustained[(n, f)] = (evs[0].elapsed_time(evs[1]) / k * 1e3, evs[1].elapsed_time(evs[2]) / k * 1e3)
            torch.cuda._sleep(int(2e8))                              # let the chip cool off between cases
            torch.cuda.synchronize()
    print("%-28s %5s %9s %9s %9s" % ("variant", "fpw", "med_us", "min_us", "GB/s@med"))
    for (n, f), t in sorted(times.items(), key=lambda kv: statistics.median(kv[1])):
        med = statistics.median(t)
        extra = ""
        if (n, f) in sustained:
            s1, s2 = sustained[(n, f)]
            extra = "   sustained: 1st half %.1f us, 2nd half %.1f us (%.0f GB/s)" % (s1, s2, 9216.0 * B / s2 / 1e3)
        print("%-28s %5d %9.1f %9.1f %9.0f%s" % (n, f, med, min(t), 9216.0 * B / med / 1e3, extra))


if __name__ == "__main__":
    main()
