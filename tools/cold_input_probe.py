#!/usr/bin/env python3
"""What the headline STFT costs when its PCM input is NOT resident in the 256 MiB Infinity Cache (GPU box only).

    python tools/cold_input_probe.py [build/variants/*.so ...] [--fpw 0,1,2,4] [--buffers 6]

Every case is timed twice in steady state, interleaved: every launch re-reading ONE 64 MiB PCM buffer ("warm":
it stays in the Infinity Cache between launches) and launches rotating over --buffers distinct buffers ("cold":
384 MiB at the default, every launch's input comes from HBM).  HIP events on the launch stream."""
import argparse
import ctypes as C
import os
import statistics
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    assert lib.jdsp_create(0, C.byref(h)) == 0, lib.jdsp_last_error(None)
    lib.jdsp_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    return lib, h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*", default=[os.path.join(ROOT, "jeicyboodsp_amd", "libjdsp.so")])
    ap.add_argument("--fpw", default="0")
    ap.add_argument("--read-pass", default="0,8", help="0 = no read pass; n > 0 = read pass with n workgroups per CU")
    ap.add_argument("--buffers", type=int, default=6)
    ap.add_argument("--frames", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=120)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    B = a.frames
    rng = np.random.default_rng(0)
    base = torch.from_numpy(np.clip(np.rint(rng.normal(0, 3000, 512 * (B + 1))), -32768, 32767).astype(np.int16)).cuda()
    pcms = [base] + [torch.roll(base, 512 * 97 * i).contiguous() for i in range(1, a.buffers)]
    spec = torch.empty((B, 1024), dtype=torch.complex64, device="cuda")
    cases = []
    for p in a.libs:
        lib, h = load(p)
        for fpw in [int(x) for x in a.fpw.split(",")]:
            for pw in [int(x) for x in a.read_pass.split(",")]:
                cases.append(("%s read_pass=%d" % (os.path.basename(p), pw), fpw * 1000 + pw, lib, h))

    def run(lib, h, fpw, buf):
        lib.jdsp_set_option(h, b"stft.frames_per_wave", fpw // 1000)
        lib.jdsp_set_option(h, b"stft.read_pass", 1 if fpw % 1000 else 0)
        lib.jdsp_set_option(h, b"stft.read_pass_wg_per_cu", fpw % 1000)
        rc = lib.jdsp_stft_i16_dev(h, C.c_void_p(buf.data_ptr()), B, 1024, 512, C.c_void_p(spec.data_ptr()))
        assert rc == 0, lib.jdsp_last_error(h)

    ref = None
    for n, f, lib, h in cases:                       # every variant must give the first one's spectra
        spec.zero_()
        run(lib, h, f, pcms[0])
        torch.cuda.synchronize()
        if ref is None:
            ref = spec.clone()
        else:
            d = (torch.view_as_real(spec) - torch.view_as_real(ref)).abs().max().item()
            if d != 0.0:
                print("!! %s fpw=%d differs from the first variant by %g" % (n, f // 1000, d))
    for _ in range(600):                             # spin the clocks up
        run(cases[0][2], cases[0][3], cases[0][1], pcms[0])
    torch.cuda.synchronize()
    res = {}
    for _ in range(a.rounds):
        for n, f, lib, h in cases:
            for mode, bufs in (("warm", pcms[:1]), ("cold", pcms)):
                for i in range(30):
                    run(lib, h, f, bufs[i % len(bufs)])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(a.iters):
                    run(lib, h, f, bufs[i % len(bufs)])
                e1.record()
                torch.cuda.synchronize()
                res.setdefault((n, f, mode), []).append(e0.elapsed_time(e1) / a.iters * 1e3)
    print("%-30s %4s %10s %10s %10s %10s" % ("variant", "fpw", "warm_us", "cold_us", "cold GB/s", "cold frac"))
    for n, f, lib, h in cases:
        w = statistics.median(res[(n, f, "warm")])
        c = statistics.median(res[(n, f, "cold")])
        print("%-30s %4d %10.1f %10.1f %10.0f %10.3f" % (n, f // 1000, w, c, 9216.0 * B / c / 1e3, 9216.0 * B / c / 1e3 / 8000.0))


if __name__ == "__main__":
    main()
