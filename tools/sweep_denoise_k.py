#!/usr/bin/env python3
"""Blocks-per-wave sweep of the fused denoise kernel (GPU box only)."""
import os, sys, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jeicyboodsp_amd
eng = jeicyboodsp_amd.Engine(0)
rng = np.random.default_rng(0)
B = 65536
x = np.clip(np.rint(rng.normal(0, 3000, B * 512)), -32768, 32767).astype(np.int16)
x[:12 * 512] = np.clip(np.rint(rng.normal(0, 45, 12 * 512)), -32768, 32767)
t = torch.from_numpy(x).cuda()
for mode in (0, 1):
    for k in (1, 2, 4, 8):
        d = eng.denoiser(mode)
        d.set_option("blocks_per_wave", k)
        d.process(t)
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                d.reset(); d.process(t)
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 10 * 1e3)
        print("mode %d K=%d  %.1f us" % (mode, k, statistics.median(ts)))
        d.close()
