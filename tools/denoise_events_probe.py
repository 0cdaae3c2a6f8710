#!/usr/bin/env python3
"""The denoise chain as a function of how much of the stream is non-voice (GPU box only).

bench_chains.py's stream is loud after its first 12 blocks: about ten EstimateNoiseSpectrum events per 65,536 blocks, so
its noise-estimate passes cost next to nothing.  Speech is 40-60 % pauses.  This probe builds streams whose pauses are
runs of quiet, sign-alternating noise (below the energy threshold, ZCR >= 200: non-voice by SS:131-139) between loud
stretches, and times the whole chain per 65,536 blocks for each.

    python tools/denoise_events_probe.py [--only-mode 0]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import jeicyboodsp_amd  # noqa: E402
from bench_chains import timed, pcm_of  # noqa: E402


def stream(rng, n_blocks, block, pause_frac, mean_run):
    """loud Gaussian blocks with pauses: geometric run lengths, `pause_frac` of the blocks quiet"""
    x = pcm_of(rng, n_blocks * block)
    if pause_frac <= 0:
        return x, 0
    quiet = np.zeros(n_blocks, dtype=bool)
    j = 0
    state = False
    while j < n_blocks:
        if state:
            run = 2 + rng.geometric(1.0 / max(mean_run - 2, 1))
        else:
            m = mean_run * (1 - pause_frac) / max(pause_frac, 1e-6)
            run = 1 + rng.geometric(1.0 / max(m, 1)) if pause_frac < 1 else 0
        quiet[j:j + run] = state
        j += run
        state = not state
    q = (np.abs(rng.normal(0, 12, n_blocks * block)) + 3.0) * np.where(np.arange(n_blocks * block) % 2 == 0, 1.0, -1.0)
    mask = np.repeat(quiet, block)
    x[mask] = np.rint(q[mask]).astype(np.int16)
    return x, int(quiet.sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only-mode", type=int, default=-1)
    ap.add_argument("--mvdr", action="store_true", help="the two MVDR chains instead (their VAD is the energy test alone)")
    ap.add_argument("--case", type=int, default=-1, help="only the i-th pause setting (0..4) of the 1024-point denoiser (for rocprofv3)")
    a = ap.parse_args()
    eng = jeicyboodsp_amd.Engine(0)
    rng = np.random.default_rng(5)
    B = 65536
    if a.mvdr:
        for i, (pause_frac, mean_run) in enumerate(((0.0, 0), (0.1, 20), (0.5, 40), (1.0, 1 << 30))):
            if a.case >= 0 and i != a.case:
                continue
            l, n_quiet = stream(rng, B, 512, pause_frac, mean_run)
            r = np.roll(l, 3)
            tl, tr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
            mv = eng.mvdr(0.0)
            mv.process(tl, tr)
            ms = timed(lambda: mv.process(tl, tr), a.iters)
            print(json.dumps({"chain": "mvdr_2mic", "blocks": B, "pause_frac": pause_frac, "quiet_blocks": n_quiet,
                              "us_per_call": round(ms * 1e3, 1)}), flush=True)
            mv.close()
        if a.case >= 0 and a.case < 10:
            return
        nb = 16384
        for i, (pause_frac, mean_run) in enumerate(((0.0, 0), (0.01, 20), (0.1, 20))):
            if a.case >= 10 and i != a.case - 10:                     # --case 10..12: one 8-microphone setting
                continue
            x, n_quiet = stream(rng, nb, 512, pause_frac, mean_run)
            mics = np.stack([np.roll(x, 2 * m) for m in range(8)])
            tm = torch.from_numpy(mics).cuda()
            for n_fft in (1024, 512):
                mv = eng.mvdr_multi(8, None, 1e-3, n_fft=n_fft)
                mv.process(tm)
                ms = timed(lambda: mv.process(tm), 3, rounds=3)
                print(json.dumps({"chain": "mvdr_8mic", "n_fft": n_fft, "samples_per_mic": nb * 512, "pause_frac": pause_frac,
                                  "quiet_1024pt_blocks": n_quiet, "us_per_call": round(ms * 1e3, 1)}), flush=True)
                mv.close()
        return
    for n_fft, block in ((1024, 512), (512, 256)):
        if a.case >= 0 and n_fft != 1024:
            continue
        for ci, (pause_frac, mean_run) in enumerate(((0.0, 0), (0.1, 20), (0.5, 40), (0.5, 6), (1.0, 1 << 30))):
            if a.case >= 0 and ci != a.case:
                continue
            x, n_quiet = stream(rng, B, block, pause_frac, mean_run)
            bufs = [torch.from_numpy(x).cuda() for _ in range(6)]
            for mode in (0, 1):
                if a.only_mode >= 0 and mode != a.only_mode:
                    continue
                d = eng.denoiser(mode, n_fft, block)
                d.process(bufs[0])
                i = [0]

                def step():
                    i[0] += 1
                    d.process(bufs[i[0] % 6])
                ms = timed(step, a.iters)
                flags = d.vad_trace(B, flags_only=True) if hasattr(d, "vad_trace") else None
                nv = int((np.asarray(flags) == 0).sum()) if flags is not None else -1
                print(json.dumps({"n_fft": n_fft, "block": block, "mode": mode, "pause_frac": pause_frac, "mean_pause_run": mean_run,
                                  "quiet_blocks": n_quiet, "non_voice_flags": nv, "us_per_65536_blocks": round(ms * 1e3, 1)}), flush=True)
                d.close()


if __name__ == "__main__":
    main()
