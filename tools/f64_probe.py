#!/usr/bin/env python3
"""FP64 STFT (jdsp_stft_i16_f64_dev, 65,536 frames -> 1 GiB of complex128): kernel variant x frames per wave x read pass,
input rotated over 6 buffers (cold) and one buffer (warm).  GPU box only.

    python tools/f64_probe.py [--steps 60]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jeicyboodsp_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--frames", type=int, default=65536)
    a = ap.parse_args()
    eng = jeicyboodsp_amd.Engine(0)
    B, K = a.frames, a.steps
    rng = np.random.default_rng(0)
    base = torch.from_numpy(np.clip(np.rint(rng.normal(0, 3000, 512 * (B + 1))), -32768, 32767).astype(np.int16)).cuda()
    pcms = [base] + [torch.roll(base, 512 * 97 * i).contiguous() for i in range(1, 6)]
    spec = torch.empty((B, 1024), dtype=torch.complex128, device="cuda")
    ref = None

    def run(bufs):
        for i in range(K // 2):
            eng.stft_f64(bufs[i % len(bufs)], B, out=spec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(K):
            eng.stft_f64(bufs[i % len(bufs)], B, out=spec)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / K * 1e3

    alg = 17408 * B
    for kern in (1, 0):
        for fpw in ((0,) if kern == 1 else (0, 1, 2, 4, 8, 32)):
            for rp in (0, 1):
                eng.set_option("stft.f64_kernel", kern)
                eng.set_option("stft.f64_frames_per_wave", fpw)
                eng.set_option("stft.read_pass", rp)
                eng.stft_f64(pcms[0], B, out=spec)
                torch.cuda.synchronize()
                if ref is None:
                    ref = spec[:4096].clone()
                same = bool(((spec[:4096] - ref).abs().max() / ref.abs().max()).item() < 1e-13)
                cold, warm = run(pcms), run(pcms[:1])
                print(json.dumps({"kernel": kern, "frames_per_wave": fpw, "read_pass": rp, "cold_us": round(cold, 1),
                                  "warm_us": round(warm, 1), "cold_frac_of_8TBs": round(alg / cold / 8e6, 3),
                                  "agrees_with_round2_kernel_1e-13": same}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
