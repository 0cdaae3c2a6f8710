#!/usr/bin/env python3
"""Is jdsp_stft_i16_dev capturable into a hipGraph, and what does replaying K launches from a
graph give against K eager launches?  (GPU box only.)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jeicyboodsp_amd
eng = jeicyboodsp_amd.Engine(0)
B, K = 65536, 50
rng = np.random.default_rng(0)
pcm = torch.from_numpy(np.clip(np.rint(rng.normal(0, 3000, 512 * (B + 1))), -32768, 32767).astype(np.int16)).cuda()
spec = torch.empty((B, 1024), dtype=torch.complex64, device="cuda")
eng.stft(pcm, B, out=spec)
torch.cuda.synchronize()
ref = spec.clone()

def eager():
    for _ in range(K):
        eng.stft(pcm, B, out=spec)

g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eager()                                  # warm up on the side stream
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        eager()
torch.cuda.current_stream().wait_stream(s)
spec.zero_()
g.replay()
torch.cuda.synchronize()
print("graph replay reproduces the eager result:", torch.equal(spec, ref))
for name, fn in (("eager", eager), ("graph", g.replay)):
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / K * 1e3)
    print("%s: %.1f us per launch (min %.1f)" % (name, sorted(ts)[2], min(ts)))
