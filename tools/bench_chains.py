#!/usr/bin/env python3
"""Secondary figures: the chains downstream of the STFT, each timed with HIP
events on the stream it runs on, inputs resident in HBM.  One JSON line per chain.

    python tools/bench_chains.py [--iters 20]

These are reported next to, never instead of, bench.py's headline metric."""
import argparse
import json
import os
import statistics
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import jeicyboodsp_amd  # noqa: E402
import oracle_lib  # noqa: E402  (CPU checker: timed here as the per-chain CPU baseline, never used by the product)

HBM_PEAK = 8000.0       # GB/s
FP32_PEAK = 157.3       # TFLOP/s vector


def timed(fn, iters, rounds=5, spin_ms=60.0):
    """Median over `rounds` of the mean launch time of `iters` back-to-back calls, after `spin_ms` of the same
    calls: a GPU that has idled needs tens of milliseconds of load to reach its sustained clocks
    (profiles/r01_bench_warmup_sweep.txt)."""
    import time
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < spin_ms:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters)
    return statistics.median(ts)


def cpu_rate(fn, units):
    """units/s of the CPU oracle (FP64, one thread, structured like the reference) on a bounded sample."""
    import time
    fn()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 1.5:
        fn()
        n += 1
    return units * n / (time.perf_counter() - t0)


def pcm_of(rng, n, sigma=3000.0):
    return np.clip(np.rint(rng.normal(0, sigma, n)), -32768, 32767).astype(np.int16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--warm", action="store_true",
                    help="every call re-reads ONE input buffer (it then lives in the 256 MiB Infinity Cache); default: the calls "
                         "rotate over 6 copies of the input (>= 256 MiB for the 65,536-block chains), so the PCM comes from HBM")
    ap.add_argument("--denoise-k", type=int, default=0, help="denoise blocks_per_wave (0 = auto: one round of resident waves)")
    a = ap.parse_args()
    eng = jeicyboodsp_amd.Engine(0)
    orc = oracle_lib.load_oracle()
    rng = np.random.default_rng(0)
    out = []

    def report(name, ms, units, unit_name, bytes_per_unit, flops_per_unit, note="", cpu=None):
        rate = units / (ms * 1e-3)
        gbs = rate * bytes_per_unit / 1e9
        tf = rate * flops_per_unit / 1e12
        line = {"chain": name, "ms": ms, "units": units, "unit": unit_name, "rate_per_s": rate,
                "input": "re-read from one buffer (cache-resident)" if a.warm else "rotated over 6 copies (from HBM) where the chain's input is a PCM stream",
                "algorithmic_GBps": gbs, "hbm_frac": gbs / HBM_PEAK, "fft_TFLOPs": tf, "fp32_vector_frac": tf / FP32_PEAK,
                "note": note}
        if cpu is not None:
            line["cpu_baseline"] = {"value": cpu, "unit": unit_name + "/s", "cores": 1, "kind": "port"}
        out.append(line)
        print(json.dumps(line), flush=True)

    B = 65536
    want = set(a.only.split(",")) if a.only else None

    class rot:
        """6 device copies of an input, handed out in turn (cold input); --warm: the one buffer every time."""

        def __init__(self, t):
            self.bufs = [t] if a.warm else [t] + [t.clone() for _ in range(5)]
            self.i = 0

        def __call__(self):
            self.i += 1
            return self.bufs[self.i % len(self.bufs)]

    def on(n):
        return want is None or n in want

    if on("stft_half"):
        x = torch.from_numpy(pcm_of(rng, 512 * (B + 1))).cuda()
        for pitch in (513, 514, 516, 520, 528, 576, 1024):
            hs = torch.empty((B, pitch), dtype=torch.complex64, device="cuda")
            ms = timed(lambda: eng.stft_half(x, B, out=hs, pitch=pitch), a.iters)
            report(f"stft_half_spectrum_513_bins_pitch{pitch}", ms, B, "frames", 1024 + 4104, 5 * 512 * 9 + 512 * 14,
                   "extra: bins 0..512 only (5,128 algorithmic bytes per frame); the headline keeps all 1024 bins")
    if on("stft_f64"):
        x = torch.from_numpy(pcm_of(rng, 512 * (B + 1))).cuda()
        xr = rot(x)
        o64 = torch.empty((B, 1024), dtype=torch.complex128, device="cuda")
        ms = timed(lambda: eng.stft_f64(xr(), B, 512, out=o64), a.iters)
        report("stft_1024_hop512_fp64", ms, B, "frames", 1024 + 16384, 5 * 512 * 9 + 512 * 14,
               "the headline analysis in the reference's own precision: FP64 window and transform, complex128 full spectrum "
               "(17,408 algorithmic bytes per frame); flops are FP64")
    if on("denoise"):
        x = pcm_of(rng, B * 512)
        x[:12 * 512] = pcm_of(rng, 12 * 512, 45.0)          # the estimate latches at block 10
        t = torch.from_numpy(x).cuda()
        tr_ = rot(t)
        for mode, nm in ((0, "specsub"), (1, "wiener")):
            d = eng.denoiser(mode)
            d.set_option("blocks_per_wave", a.denoise_k)
            d.process(t)                                        # sizes the workspace

            ms = timed(lambda: d.process(tr_()), a.iters)       # steady state: one stream fed 65,536 blocks per call
            # 512 int16 in + 512 int16 out per block; forward + inverse 1024-pt real transforms
            report("denoise_" + nm, ms, B, "blocks", 2048, 2 * 5 * 512 * 9 + 2 * 512 * 14,
                   "VAD + plan + noise estimate + fused window/FFT/gain/IFFT/OLA, 65,536 blocks of 512",
                   cpu=cpu_rate(lambda: orc.denoise_stream(mode, x[:1024 * 512]), 1024))
            d.close()
        # BASELINE config 3 as worded: 512-point frames, hop 256 (two frames per wave transform), 65,536 blocks of 256
        x5 = pcm_of(rng, B * 256)
        q = (np.abs(rng.normal(0, 45, 24 * 256)) + 14.0) * np.where(np.arange(24 * 256) % 2 == 0, 1.0, -1.0)
        x5[:24 * 256] = np.rint(q).astype(np.int16)         # sign-alternating quiet start: ZCR >= 200, so the estimate latches
        t5 = torch.from_numpy(x5).cuda()
        t5r = rot(t5)
        for mode, nm in ((0, "specsub"), (1, "wiener")):
            d = eng.denoiser(mode, 512, 256)
            d.process(t5)
            ms = timed(lambda: d.process(t5r()), a.iters)
            report("denoise_" + nm + "_512pt_hop256", ms, B, "blocks", 1024, 2 * 5 * 256 * 8 + 2 * 256 * 14,
                   "BASELINE config 3 as worded: FFT_PROCESSING_SIZE 512, BLOCK_LEN 256; 65,536 blocks of 256",
                   cpu=cpu_rate(lambda: orc.denoise_stream(mode, x5[:1024 * 256], block=256), 1024))
            d.close()
    if on("mfcc"):
        x = torch.from_numpy(pcm_of(rng, 512 * (B + 1))).cuda()
        xr = rot(x)
        m = eng.mfcc()
        ms = timed(lambda: m.frames(xr(), B), a.iters)
        xs = x[:512 * 257].cpu().numpy()
        report("mfcc_native_1024_512_38ch", ms, B, "frames", 1024 + 96, 5 * 512 * 9 + 512 * 14 + 2 * 1024 + 2 * 38 * 12,
               "pre-emphasis/Hamming/FFT/mel/ln/DCT/lifter, 65,536 frames, 12 doubles out",
               cpu=cpu_rate(lambda: orc.mfcc_frames(orc.mfcc_native_cfg(), xs, 256), 256))
        m.close()
        x16 = torch.from_numpy(pcm_of(rng, 160 * (B - 1) + 400)).cuda()
        x16r = rot(x16)
        m = eng.mfcc(win_len=400, hop=160, n_fft=512, n_chan=40, n_cep=13, half_rate=8000.0)
        ms = timed(lambda: m.frames(x16r(), B), a.iters)
        report("mfcc_400_160_512fft_40mel", ms, B, "frames", 320 + 104, 5 * 512 * 9 + 512 * 14, "BASELINE config 4 framing")
        m.close()
    if on("mfcc10k"):
        # BASELINE config 4: a 10,000-utterance batch (ragged, 1-6 s at 16 kHz), every utterance framed
        # on its own; on N GPUs jeicyboodsp_amd.sharding.utterance_shard hands out whole utterances
        from jeicyboodsp_amd import sharding
        lens = rng.integers(16000, 96000, 10000)
        m = eng.mfcc(win_len=400, hop=160, n_fft=512, n_chan=40, n_cep=13, half_rate=8000.0)
        fpu = [(int(n) - 400) // 160 + 1 for n in lens]
        offs = np.concatenate([[0], np.cumsum(lens)])
        starts = np.concatenate([offs[u] + 160 * np.arange(fpu[u], dtype=np.int64) for u in range(len(lens))])
        x = torch.from_numpy(pcm_of(rng, int(offs[-1]))).cuda()
        st = torch.from_numpy(starts).cuda()
        ms = timed(lambda: m.frames(x, len(starts), frame_start=st), max(a.iters // 4, 3))
        loads = [sum(fpu[f:f + n]) for f, n in (sharding.utterance_shard(fpu, r, 8) for r in range(8))]
        report("mfcc_10k_utterances_400_160_512fft_40mel", ms, len(starts), "frames", 320 + 104, 5 * 512 * 9 + 512 * 14,
               "10,000 ragged utterances, %d frames; an 8-way utterance split is balanced to %.2f%%"
               % (len(starts), 100.0 * (max(loads) - min(loads)) / max(loads)))
        m.close()
    if on("gmm"):
        # the 10,000-utterance batch's MFCC vectors (12 cepstra, native MFCC configuration
        # framing is irrelevant here) scored where they lie: 25 classes (GMMTest:26) / one 6-state model (Viterbi:26-28)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import gmm_cases as gc
        fpu = rng.integers(98, 598, 10000)
        first = np.concatenate([[0], np.cumsum(fpu)]).astype(np.int64)
        nvec = int(first[-1])
        feats_h = rng.normal(0.0, 3.0, (nvec, 12))
        feats, first_d = torch.from_numpy(feats_h).cuda(), torch.from_numpy(first).cuda()
        classes = gc.gmm_records(1, 25)
        g = eng.gmm(classes)
        ms = timed(lambda: g.score(feats, first_d), max(a.iters // 8, 3))
        # per vector and class: 4 mixtures x (48 MAC + 4 x ~8 flops + 4 exp) + 1 log ~ 550 FP64 flops
        report("gmm_score_25_classes_10k_utterances", ms, nvec, "vectors", 96, 25 * 550,
               "FP64; %d vectors in 10,000 utterances against 25 four-mixture GMMs; flops are FP64" % nvec,
               cpu=cpu_rate(lambda: orc.gmm_classify(feats_h[:2000], classes), 2000))
        g.set_option("evaluation", 1)
        ms = timed(lambda: g.score(feats, first_d), max(a.iters // 8, 3))
        report("gmm_score_25_classes_10k_utterances_fused_evaluation", ms, nvec, "vectors", 96, 25 * 250,
               "opt-in: FMA projections, -0.5/var precomputed, one exp per mixture (jdsp_gmm_set_option evaluation=1)")
        g.close()
        models = gc.hmm_records(2, 1)
        h = eng.hmm(models)
        ms = timed(lambda: h.viterbi(feats, first_d, want_path=True), max(a.iters // 8, 3))
        report("hmm_recursion_6_states_10k_utterances", ms, nvec, "vectors", 96 + 4, 6 * 550,
               "FP64; emission kernel (vector-parallel) + one thread per utterance for the recursion",
               cpu=cpu_rate(lambda: orc.hmm_viterbi(feats_h[:20000], models[0]), 20000))
        h.close()
    if on("fastconv"):
        nb = 4096
        taps = rng.normal(size=7169) * 0.01
        x = torch.from_numpy(pcm_of(rng, nb * 1024, 2000.0)).cuda()
        fc = eng.fastconv(taps, 8192)

        fc.process(x)                                            # steady state: a stream fed 4,096 blocks per call
        ms = timed(lambda: fc.process(x), max(a.iters // 4, 3))
        xs = x[:71 * 1024].cpu().numpy()
        report("fastconv_8192_native", ms, nb, "blocks", 4096, 2 * 5 * 4096 * 12 + 2 * 4096 * 14 + 8192 * 6,
               "reference-native: 7169 taps, 1024-sample blocks, steady-state calls of 4,096 blocks; partitioned "
               "(15 x 512 taps) unless JDSP_FASTCONV_PARTITIONED=0; flops counted for the 8192-point formulation",
               cpu=cpu_rate(lambda: orc.fastconv_stream(xs, taps, 8192), 64))
        fc.close()
        nb = 65536
        h2 = rng.normal(size=(2, 256)) * 0.1
        x = torch.from_numpy(pcm_of(rng, nb * 769, 2000.0)).cuda()
        fc = eng.fastconv(h2, 1024)

        fc.process(x)
        xr = rot(x)
        ms = timed(lambda: fc.process(xr()), a.iters)
        xs = x[:513 * 769].cpu().numpy()
        report("fastconv_1024_hrir_pair", ms, nb, "blocks", 4614, 3 * 5 * 512 * 9 + 3 * 512 * 14 + 2 * 1024 * 6,
               "BASELINE config 2: 256-tap pair, 769-sample blocks, mono in -> 2 ears out",
               cpu=cpu_rate(lambda: [orc.fastconv_stream(xs, h2[0], 1024), orc.fastconv_stream(xs, h2[1], 1024)], 512))
        fc.close()
    if on("fft"):
        n = 65536
        z = torch.from_numpy(rng.normal(size=(n, 512)) + 1j * rng.normal(size=(n, 512))).cuda()
        ms = timed(lambda: eng.fft_process(z), max(a.iters // 4, 3))
        zs = z[:256].cpu().numpy()
        report("fftprocess_f64_512", ms, n, "transforms", 2 * 512 * 16, 5 * 512 * 9, "FFTAlgorithm_ver2 FFTProcess, FP64, batch 65,536",
               cpu=cpu_rate(lambda: orc.fft_process(zs), 256))
    if on("pitch"):
        x = pcm_of(rng, B * 512)
        t = torch.from_numpy(x).cuda()
        tr_ = rot(t)
        ms = timed(lambda: eng.pitch(tr_()), a.iters)
        report("pitch_autocorr", ms, B, "blocks", 1024 + 8, 2 * 5 * 512 * 9 + 2 * 512 * 14,
               "PitchEstimation_method1 CalcPitch: FFT -> |X|^2 -> IFFT -> arg max, 65,536 blocks",
               cpu=cpu_rate(lambda: orc.pitch_stream(x[:1024 * 512]), 1024))
    if on("mvdr"):
        l = pcm_of(rng, B * 512)
        r = pcm_of(rng, B * 512)
        l[:12 * 512] = pcm_of(rng, 12 * 512, 45.0)
        tl, tr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
        mv = eng.mvdr(0.0)
        mv.process(tl, tr)
        tlr, trr = rot(tl), rot(tr)
        ms = timed(lambda: mv.process(tlr(), trr()), a.iters)
        report("mvdr_2mic", ms, B, "blocks", 3072, 3 * 5 * 512 * 9 + 3 * 512 * 14 + 1024 * 40,
               "BeamForming_MVDR_ver1: VAD + correlation + per-bin weights + inverse, 65,536 stereo blocks",
               cpu=cpu_rate(lambda: orc.mvdr_stream(l[:512 * 512], r[:512 * 512]), 512))
        mv.close()
    if on("mvdr8"):
        nbm = 16384
        mics = np.stack([pcm_of(rng, nbm * 512) for _ in range(8)])
        mics[:, :40 * 512] = np.stack([pcm_of(rng, 40 * 512, 30.0) for _ in range(8)])
        tm = torch.from_numpy(mics).cuda()
        mv = eng.mvdr_multi(8, None, 1e-3)
        mv.process(tm)

        ms = timed(lambda: mv.process(tm), max(a.iters // 4, 3))
        small = mics[:, :64 * 512].copy()
        report("mvdr_8mic_per_bin_covariance", ms, nbm, "blocks", 8 * 1024 + 1024, 9 * 5 * 512 * 9 + 9 * 512 * 14 + 1024 * 8 * 8,
               "BASELINE config 5 (generalisation, no reference): 8 microphones, per-bin 8x8 covariance, 16,384 blocks, 39 estimation frames",
               cpu=cpu_rate(lambda: orc.mvdrn_stream(small, None, 1e-3), 64))
        mv.close()
        # BASELINE config 5 as worded: 512-point frames (blocks of 256): 32,768 blocks = the same 8.4 M samples per microphone
        nb5 = 32768
        mv = eng.mvdr_multi(8, None, 1e-3, n_fft=512)
        mv.process(tm)
        ms = timed(lambda: mv.process(tm), max(a.iters // 4, 3))
        small5 = mics[:, :128 * 256].copy()
        report("mvdr_8mic_per_bin_covariance_512pt", ms, nb5, "blocks", 8 * 512 + 512, 4.5 * 5 * 512 * 9 + 1024 * 8 * 8 / 2,
               "BASELINE config 5 as worded (generalisation, no reference): 8 microphones, 512-point frames, per-bin 8x8 "
               "covariance over 257 bins, 32,768 blocks of 256, two microphones per forward and two blocks per inverse transform",
               cpu=cpu_rate(lambda: orc.mvdrn_stream(small5, None, 1e-3, n_fft=512), 128))
        mv.close()
    eng.close()


if __name__ == "__main__":
    main()
