#!/usr/bin/env python3
"""Can the read pass of batch n+1 run under the transform of batch n?  (GPU box only.)

The headline step is pcm_touch_kernel (64 MiB of PCM streamed into the Infinity Cache, ~9-13 us) followed by the
transform at the chip's store rate (~88 us).  A pipelined caller knows its next batch: this probe issues the read pass
of batch n+1 on a second stream (a second context whose STFT entry issues the pass alone) while batch n's transform
runs with its own pass off, eagerly and in one captured graph of K steps, against the library's default.

The pass alone is not something the shipped library does (it would return spectra nobody wrote): it exists in a
timing-only build.  So:

    tools/build_variant.sh pass_alone -DJDSP_PROBE_READ_PASS_ALONE
    JDSP_LIB=build/variants/pass_alone.so python tools/prefetch_probe.py [--steps 400]
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
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--frames", type=int, default=65536)
    ap.add_argument("--buffers", type=int, default=6)
    a = ap.parse_args()
    eng = jeicyboodsp_amd.Engine(0)
    pre = jeicyboodsp_amd.Engine(0)
    assert "pass_alone" in os.environ.get("JDSP_LIB", ""), "needs the -DJDSP_PROBE_READ_PASS_ALONE build (see the docstring)"
    pre.set_option("stft.read_pass", 1)              # in that build: the read pass and no transform
    B, P, K = a.frames, a.buffers, a.steps
    rng = np.random.default_rng(0)
    base = torch.from_numpy(rng.integers(-20000, 20000, 512 * (B + 1)).astype(np.int16)).cuda()
    pcms = [base] + [torch.roll(base, 512 * 97 * i).contiguous() for i in range(1, P)]
    spec = torch.empty((B, 1024), dtype=torch.complex64, device="cuda")
    eng.stft(pcms[0], B, 1024, 512, out=spec)
    pre.stft(pcms[0], B, 1024, 512, out=spec)
    torch.cuda.synchronize()

    def serial(read_pass):
        eng.set_option("stft.read_pass", read_pass)
        for i in range(K):
            eng.stft(pcms[i % P], B, 1024, 512, out=spec)

    def serial_two_contexts():
        eng.set_option("stft.read_pass", 0)
        for i in range(K):
            pre.stft(pcms[i % P], B, 1024, 512, out=spec)
            eng.stft(pcms[i % P], B, 1024, 512, out=spec)

    def overlapped():
        eng.set_option("stft.read_pass", 0)
        main_s = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        pre.stft(pcms[0], B, 1024, 512, out=spec)
        for i in range(K):
            ev = torch.cuda.Event()
            ev.record(main_s)                        # everything before batch i's transform ...
            side.wait_event(ev)
            with torch.cuda.stream(side):
                pre.stft(pcms[(i + 1) % P], B, 1024, 512, out=spec)     # ... then batch i+1 is read under it
                done = torch.cuda.Event()
                done.record(side)
            eng.stft(pcms[i % P], B, 1024, 512, out=spec)
            main_s.wait_event(done)

    def run(fn, name, graph):
        if graph:
            cap = torch.cuda.Stream()
            cap.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(cap):
                with torch.cuda.graph(g, stream=cap):
                    fn()
            torch.cuda.current_stream().wait_stream(cap)
            go = g.replay
        else:
            go = fn
        go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        go()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / K
        print(json.dumps({"variant": name, "launch": "graph" if graph else "eager", "us_per_step": round(us, 2),
                          "frac_of_8TBps": round(603979776 / us / 8e6, 3)}), flush=True)

    for graph in (True, False):
        run(lambda: serial(-1), "library default: read pass then transform, one stream", graph)
        run(serial_two_contexts, "the same as two calls (pass alone, transform alone), one stream", graph)
        run(overlapped, "read pass of batch n+1 on a second stream under the transform of batch n", graph)
        run(lambda: serial(-1), "library default again", graph)
    eng.set_option("stft.read_pass", -1)


if __name__ == "__main__":
    main()
