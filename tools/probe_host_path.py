#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer STFT entry (jdsp_stft_i16): pageable numpy buffers in,
512 MiB of spectra out.  GPU box only; the figure goes to DESIGN.md, never to bench.py's value."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jeicyboodsp_amd
eng = jeicyboodsp_amd.Engine(0)
B = 65536
rng = np.random.default_rng(0)
pcm = np.clip(np.rint(rng.normal(0, 3000, 512 * (B + 1))), -32768, 32767).astype(np.int16)
eng.stft(pcm[: 512 * 65])
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    spec = eng.stft(pcm)
    ts.append(time.perf_counter() - t0)
print("host path: %.1f ms per 65,536 frames (best of 3) = %.2f M frames/s; output %.0f MiB"
      % (min(ts) * 1e3, B / min(ts) / 1e6, spec.nbytes / 2**20))

import torch
pin_in = torch.empty(pcm.size, dtype=torch.int16).pin_memory()
pin_in.numpy()[:] = pcm
pin_out = torch.empty((B, 1024), dtype=torch.complex64).pin_memory()
eng.stft(pin_in.numpy(), out=pin_out.numpy())
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    eng.stft(pin_in.numpy(), out=pin_out.numpy())
    ts.append(time.perf_counter() - t0)
print("pinned, pipelined: %.1f ms per 65,536 frames (best of 3) = %.2f M frames/s; equal to the plain path: %s"
      % (min(ts) * 1e3, B / min(ts) / 1e6, np.array_equal(pin_out.numpy(), spec)))
