#!/usr/bin/env python3
"""Timing of the generic-hop STFT path (hop 256, 65,536 frames) -- GPU box only."""
import os, sys, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jeicyboodsp_amd
eng = jeicyboodsp_amd.Engine(0)
B = 65536
rng = np.random.default_rng(0)
for hop, nfft in ((256, 1024), (512, 1024), (256, 512)):
    pcm = torch.from_numpy(np.clip(np.rint(rng.normal(0, 3000, hop * (B - 1) + nfft + 8)), -32768, 32767).astype(np.int16)).cuda()
    src = pcm if hop != 512 else pcm[1:]          # hop 512: force the generic path with an unaligned pointer
    spec = torch.empty((B, nfft), dtype=torch.complex64, device="cuda")
    eng.stft(src, B, nfft, hop, out=spec)
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            eng.stft(src, B, nfft, hop, out=spec)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 10 * 1e3)
    print("n_fft %d hop %d generic path: %.1f us per 65,536 frames" % (nfft, hop, statistics.median(ts)))
