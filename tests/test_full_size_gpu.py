"""GPU parity at BASELINE.json's full batch sizes (65,536 frames / blocks per launch), against the
CPU oracle itself rather than only through size-independent properties: the oracle needs a few
seconds per chain at this size, which is affordable once per suite."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
B = 65536
TOL = 1e-5


@pytest.fixture(scope="module")
def eng():
    import jeicyboodsp_amd
    e = jeicyboodsp_amd.Engine(0)
    yield e
    e.close()


def speechlike(seed, n_blocks):
    """Alternating quiet/loud stretches of random length: thousands of noise frames and latches."""
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 3000, n_blocks * 512)
    b = 0
    while b < n_blocks:
        q = int(rng.integers(1, 40))
        x[b * 512:(b + q) * 512] = rng.normal(0, 45, min(q, n_blocks - b) * 512)[: (min(b + q, n_blocks) - b) * 512]
        b += q + int(rng.integers(1, 60))
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def test_stft_full_batch_vs_oracle_in_chunks(eng, oracle):
    import torch
    rng = np.random.default_rng(0)
    pcm = np.clip(np.rint(rng.normal(0, 3000, 512 * (B + 1))), -32768, 32767).astype(np.int16)
    spec = eng.stft(torch.from_numpy(pcm).cuda(), B)
    torch.cuda.synchronize()
    worst = 0.0
    for f0 in [0, 1, 4095, 4096, 20000, 32767, 49999, B - 512]:
        want = oracle.stft(pcm[512 * f0:512 * (f0 + 513)], 512)
        got = spec[f0:f0 + 512].cpu().numpy().astype(np.complex128)
        worst = max(worst, (np.abs(got - want) / np.abs(want).max(axis=1, keepdims=True)).max())
    assert worst < TOL


@pytest.mark.parametrize("mode", [0, 1])
def test_denoise_full_batch_vs_oracle(eng, oracle, mode):
    import torch
    pcm = speechlike(mode, B)
    o_out, o_pre, flags, noises, ver = oracle.denoise_trace(mode, pcm)
    assert noises.shape[0] > 300 and (flags == 0).sum() > 10000          # a real mix of events
    d = eng.denoiser(mode)
    out, pre = d.process(torch.from_numpy(pcm).cuda(), want_precast=True)
    torch.cuda.synchronize()
    out, pre = out.cpu().numpy(), pre.cpu().numpy()
    v, _, _ = d.vad_trace(B)
    assert np.array_equal(v.astype(np.int32), flags)
    assert out.shape == o_out.shape
    assert np.abs(pre - o_pre).max() < TOL * np.abs(o_pre).max()
    diff = np.abs(out.astype(np.int32) - o_out.astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3
    assert np.abs(d.noise() - noises[-1]).max() <= TOL * noises[-1].max()
    d.close()


def test_mfcc_pitch_fastconv_full_batch_vs_oracle(eng, oracle, golden_dir):
    import os
    import torch
    rng = np.random.default_rng(3)
    pcm = np.clip(np.rint(rng.normal(0, 3000, 512 * (B + 1))), -32768, 32767).astype(np.int16)
    t = torch.from_numpy(pcm).cuda()
    # MFCC: 65,536 frames on the GPU, every 16th checked against the oracle
    m = eng.mfcc()
    feats = m.frames(t, B)
    torch.cuda.synchronize()
    idx = np.arange(0, B, 16)
    cfg = oracle.mfcc_native_cfg()
    want = np.stack([oracle.mfcc_frames(cfg, pcm[512 * f:512 * f + 1024], 1)[0] for f in idx])
    got = feats.cpu().numpy()[idx]
    assert (np.abs(got - want) / np.abs(want).max(axis=1, keepdims=True)).max() < TOL
    m.close()
    # pitch: all 65,536 blocks
    arg, rmax, ac = eng.pitch(t[: B * 512], want_autocorr=True)
    torch.cuda.synchronize()
    o_arg, o_max, o_ac = oracle.pitch_stream(pcm[: B * 512])
    arg, ac = arg.cpu().numpy(), ac.cpu().numpy()
    scale = o_ac[:, 0] + 1.0
    assert (np.abs(ac - o_ac) / scale[:, None]).max() < TOL
    tie = o_max - o_ac[np.arange(B), arg] <= TOL * scale
    assert np.all((arg == o_arg) | tie) and (arg == o_arg).mean() > 0.99
    # overlap-save, native shape: 2,048 blocks = 2 M samples through the 7169-tap response
    g = np.load(os.path.join(golden_dir, "rir_taps.npz"), allow_pickle=False)
    taps = np.zeros(int(g["n_taps"]))
    taps[g["index"]] = g["value"]
    nb = 2048
    fc = eng.fastconv(taps, 8192)
    out, pre = fc.process(t[: nb * 1024], want_precast=True)
    torch.cuda.synchronize()
    o_out, o_pre = oracle.fastconv_stream(pcm[: nb * 1024], taps, 8192)
    assert np.abs(pre[0].cpu().numpy() - o_pre).max() < TOL * np.abs(o_pre).max()
    assert np.abs(out[0].cpu().numpy().astype(np.int32) - o_out.astype(np.int32)).max() <= 1
    fc.close()
