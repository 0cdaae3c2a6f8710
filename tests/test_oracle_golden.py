"""CPU tests: the oracle against the committed golden vectors (outputs of the
reference's FFTAlgorithm_ver2.cpp compiled in the authoring container), against
the compiled reference itself when oracle/_ref is present, and against the
known answers SURVEY.md quotes for the parts no fixture can cover (FFTW-calling
programs: "parity unpinned" at the FFTW boundary)."""
import os

import numpy as np
import pytest

import oracle_lib


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.mark.parametrize("n", [512, 1024])
def test_bitrev_table_bit_exact(oracle, golden_dir, n):
    g = _load(golden_dir, "fftalg_%d.npz" % n)
    got = oracle.bitrev_table(n)
    assert got.dtype == np.int16 and np.array_equal(got, g["bitrev"])
    # the table starts 0,256,128,384,64 for N=512
    if n == 512:
        assert got[:5].tolist() == [0, 256, 128, 384, 64]
    # it is an involutive permutation
    assert np.array_equal(np.sort(got), np.arange(n)) and np.array_equal(got[got], np.arange(n))


def test_bitrev_quirk_bits_from_block_len(oracle):
    # FFTAlgorithm_ver2.cpp:188 takes the bit count from BLOCK_LEN, :202 masks with iFFTLen-1
    t = oracle.bitrev_table(256, block_len=512)
    full = oracle.bitrev_table(512)
    assert np.array_equal(t, full[:256] & 255)


@pytest.mark.parametrize("n", [512, 1024])
def test_fftprocess_bit_exact_vs_golden(oracle, golden_dir, n):
    g = _load(golden_dir, "fftalg_%d.npz" % n)
    frames = g["pcm"].reshape(-1, n).astype(np.complex128)
    fwd = oracle.fft_process(frames, True)
    assert np.array_equal(fwd.view(np.float64), g["fwd"].view(np.float64))
    inv = oracle.fft_process(fwd, False)
    assert np.array_equal(inv.view(np.float64), g["inv"].view(np.float64))
    cf = oracle.fft_process(g["cin"], True)
    assert np.array_equal(cf.view(np.float64), g["cfwd"].view(np.float64))


def test_slow_dft_family_bit_exact_vs_golden(oracle, golden_dir):
    g = _load(golden_dir, "fftalg_512.npz")
    frames = g["pcm"].reshape(-1, 512)
    for i in range(2):
        assert np.array_equal(oracle.dft_process(frames[i]).view(np.float64), g["dft"][i].view(np.float64))
        assert np.array_equal(oracle.idft_process(g["fwd"][i]).view(np.float64), g["idft"][i].view(np.float64))
        assert np.array_equal(oracle.ifft_process(g["fwd"][i]).view(np.float64), g["ifft_n2"][i].view(np.float64))


def test_main_roundtrip_bit_exact_vs_golden(oracle, golden_dir):
    g = _load(golden_dir, "fftalg_512.npz")
    out = oracle.fft_roundtrip_i16(g["pcm"], 512)
    assert np.array_equal(out, g["main_out"])
    # the truncating cast makes the round trip lossy by +-1 on a sizeable share of samples
    d = out.astype(np.int32) - g["pcm"].astype(np.int32)
    assert np.abs(d).max() == 1 and 0.05 < np.mean(d != 0) < 0.5


@pytest.mark.parametrize("n", [512, 1024])
def test_live_reference_agrees_on_fresh_seed(oracle, n):
    ref = oracle_lib.load_ref(n)
    if ref is None:
        pytest.skip("oracle/_ref not built (reference checkout absent)")
    import sys, subprocess
    # run in a child so FFTProcess's per-call printf does not flood the test log
    code = (
        "import sys,os,numpy as np\n"
        "sys.path.insert(0,%r)\n"
        "import oracle_lib\n"
        "os.dup2(os.open(os.devnull,os.O_WRONLY),1)\n"
        "o=oracle_lib.load_oracle(); r=oracle_lib.load_ref(%d)\n"
        "rng=np.random.default_rng(99)\n"
        "for t in range(3):\n"
        "    z=rng.normal(size=%d)*1e3+1j*rng.normal(size=%d)\n"
        "    for fwd in (True,False):\n"
        "        a=o.fft_process(z,fwd); b=r.fft_process(z,fwd)\n"
        "        assert np.array_equal(a.view(np.float64),b.view(np.float64))\n"
        "assert np.array_equal(o.bitrev_table(%d),r.bitrev_table(%d))\n"
        % (os.path.dirname(os.path.abspath(__file__)), n, n, n, n, n))
    subprocess.run([sys.executable, "-c", code], check=True)


@pytest.mark.parametrize("n", [8, 64, 512, 1024, 8192])
def test_dft_contract_vs_numpy_and_pinned_fft(oracle, n):
    rng = np.random.default_rng(n)
    z = rng.normal(size=(2, n)) + 1j * rng.normal(size=(2, n))
    f = oracle.dft_c2c(z, -1)
    b = oracle.dft_c2c(z, +1)
    scale = np.abs(f).max()
    assert np.abs(f - np.fft.fft(z)).max() < 1e-12 * scale
    assert np.abs(b - np.fft.ifft(z) * n).max() < 1e-12 * scale
    # the in-reference radix-2 (pinned above) is the same transform up to its truncated PI (~1e-11)
    p = oracle.fft_process(z, True)
    assert np.abs(p - f).max() < 1e-9 * scale


def test_hamming_and_stft(oracle):
    w = oracle.hamming(1024)
    i = np.arange(1024)
    assert np.allclose(w, 0.54 - 0.46 * np.cos(2 * 3.141592 * i / 1023), rtol=0, atol=1e-15)
    assert abs(w[0] - 0.08) < 1e-12 and w.max() < 1.0
    rng = np.random.default_rng(0)
    pcm = np.clip(np.rint(rng.normal(0, 3000, 512 * 9)), -32768, 32767).astype(np.int16)
    s = oracle.stft(pcm, 8)
    for f in range(8):
        ref = np.fft.fft(pcm[512 * f:512 * f + 1024] * w)
        assert np.abs(s[f] - ref).max() < 1e-10 * np.abs(ref).max()


def test_vad_block(oracle):
    rng = np.random.default_rng(3)
    quiet = np.rint(rng.normal(0, 20, 512)).astype(np.int16)
    loud = np.rint(rng.normal(0, 3000, 512)).astype(np.int16)
    v, e, z = oracle.vad_block(quiet)
    assert not v and e <= 700 and z >= 200
    v, e, z = oracle.vad_block(loud)
    assert v and e > 700
    # few zero crossings => voice even when quiet (SS:147)
    v, e, z = oracle.vad_block(np.full(512, 5, np.int16))
    assert v and z < 200 and e < 700
    # independent recomputation
    w = 0.54 - 0.46 * np.cos(2 * 3.141592 * np.arange(1024) / 1023)
    fr = np.concatenate([np.zeros(512), quiet.astype(np.float64)])
    s = np.trunc(fr * w).astype(np.int64)
    raw_next = np.concatenate([fr[1:], [0]]).astype(np.int64)
    _, e2, z2 = oracle.vad_block(quiet)
    assert abs(e2 - np.sum(s.astype(np.float64) ** 2) / 1024) < 1e-9
    assert z2 == int(np.sum(s * raw_next < 0))


def _denoise_pcm(n_blocks, seed=0, quiet_blocks=12):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 3000, n_blocks * 512)
    x[:quiet_blocks * 512] = rng.normal(0, 45, quiet_blocks * 512)
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("mode", [0, 1])
def test_denoise_stream_structure(oracle, mode):
    pcm = _denoise_pcm(40)
    out, pre = oracle.denoise_stream(mode, pcm)
    assert out.size == (40 - 2) * 512            # SS:260-263: B blocks in -> B-2 out
    assert np.all(np.isfinite(pre))
    o2, p2, flags, noises, ver = oracle.denoise_trace(mode, pcm)
    assert np.array_equal(out, o2)
    assert flags[:12].sum() == 0 and flags[12:].all()
    # the estimate latches exactly once, at run length 10 (block index 9), SS:189-193
    assert noises.shape[0] == 2 and ver[8] == 0 and ver[9] == 1
    # noise recurrence, independent: A += |X|; /2 from run length 3 on (SS:182-187)
    w = oracle.hamming(1024)
    A = np.zeros(1024)
    for n in range(2, 11):
        fr = pcm[(n - 2) * 512:(n) * 512].astype(np.float64) * w
        A += np.abs(np.fft.fft(fr))
        if n >= 3:
            A /= 2
    assert np.abs(A - noises[1]).max() < 1e-9 * A.max()
    # before the latch the estimate is zero, so both algorithms are a plain WOLA identity:
    # emitted block e = 1.08-ish * input (Hamming pairs at 50% overlap sum to 1.08)
    e = 3
    blk = pcm[(e + 1) * 512:(e + 2) * 512].astype(np.float64)
    gain = w[:512] + w[512:]
    assert np.abs(pre[e * 512:(e + 1) * 512] - blk * gain).max() < 1e-6


def test_denoise_gain_formulas(oracle):
    pcm = _denoise_pcm(24, seed=5)
    w = oracle.hamming(1024)
    for mode in (0, 1):
        out, pre, flags, noises, ver = oracle.denoise_trace(mode, pcm)
        N = noises[1]
        ys = {}
        for i in (20, 21):          # call index (0-based block): frame = blocks i-1,i
            X = np.fft.fft(pcm[(i - 1) * 512:(i + 1) * 512] * w)
            if mode == 0:
                Y = X * (1 - N / np.abs(X))
            else:
                Y = X * (1 - np.minimum(1, N ** 2 / np.abs(X) ** 2))
            ys[i] = np.fft.ifft(Y).real
        want = ys[20][512:] + ys[21][:512]
        got = pre[(21 - 2) * 512:(21 - 1) * 512]
        assert np.abs(got - want).max() < 1e-7 * np.abs(want).max()


def test_wiener_zero_over_zero_is_nan_and_casts_to_zero(oracle):
    # WF:204 with |X|^2 == 0 and noise == 0 (digital silence before any estimate): 0/0
    pcm = np.zeros(6 * 512, np.int16)
    out, pre = oracle.denoise_stream(1, pcm)
    assert np.all(np.isnan(pre)) and np.all(out == 0)
    out, pre = oracle.denoise_stream(0, pcm)
    assert np.all(pre == 0) and np.all(out == 0)


def test_fastconv_matches_direct_convolution(oracle, golden_dir):
    g = _load(golden_dir, "rir_taps.npz")
    taps = np.zeros(int(g["n_taps"]))
    taps[g["index"]] = g["value"]
    assert taps.size == 7169 and g["index"].size == 69 and g["index"][0] == 2976 and taps[2976] == 1.0
    assert abs(taps.sum() - 3.8989) < 1e-4        # SURVEY §2 FilterCoefficient.h facts
    rng = np.random.default_rng(1)
    nb = 12
    pcm = np.clip(np.rint(rng.normal(0, 2000, nb * 1024)), -32768, 32767).astype(np.int16)
    out, pre = oracle.fastconv_stream(pcm, taps, 8192)
    assert out.size == (nb - 7) * 1024           # 3D:119-123
    x = pcm.astype(np.float64).copy()
    x[:7 * 1024] = 0                             # queued blocks 1..7 never receive the samples (3D:120)
    full = np.convolve(x, taps)[7 * 1024:nb * 1024]
    assert np.abs(pre - full).max() < 1e-8 * np.abs(full).max()
    # small generic shape (BASELINE config D3: N=1024, M=256, L=769)
    h = rng.normal(size=256)
    pcm2 = np.clip(np.rint(rng.normal(0, 2000, 9 * 769)), -32768, 32767).astype(np.int16)
    out2, pre2 = oracle.fastconv_stream(pcm2, h, 1024)
    x2 = pcm2.astype(np.float64).copy()
    x2[:769] = 0
    full2 = np.convolve(x2, h)[769:9 * 769]
    assert out2.size == 8 * 769 and np.abs(pre2 - full2).max() < 1e-9 * np.abs(full2).max()


def test_mel_tables_known_answers(oracle):
    cfg = oracle.mfcc_native_cfg()
    mel, fi, fb = oracle.mel_init(cfg)
    # SURVEY §8a A15 quotes these edges (Hz) from the reference's own run
    for got, want in zip(mel[:4], [65.357, 136.817, 214.949, 300.375]):
        assert abs(got - want) < 5e-4
    assert abs(mel[-2] - 20107.273) < 5e-4 and abs(mel[-1] - 22050.0) < 1e-9
    counts = np.bincount(fi, minlength=39)
    assert counts[:38].tolist() == [2, 2, 1, 2, 3, 2, 3, 2, 4, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 8, 9, 10, 11, 11,
                                    13, 14, 16, 17, 18, 20, 22, 25, 26, 29, 31, 35, 37, 41]
    assert counts[38] == 46
    assert np.all((fb >= 0) & (fb <= 1))


def test_mfcc_stream(oracle):
    cfg = oracle.mfcc_native_cfg()
    rng = np.random.default_rng(2)
    nb = 6
    pcm = np.clip(np.rint(rng.normal(0, 3000, nb * 1024)), -32768, 32767).astype(np.int16)
    feats = oracle.mfcc_stream(cfg, pcm)
    assert feats.shape == (2 * nb - 1, 12) and np.all(np.isfinite(feats))
    # independent numpy recomputation of one vector (frame index 3 of the padded stream)
    mel, fi, fb = oracle.mel_init(cfg)
    padded = np.concatenate([np.zeros(512), pcm.astype(np.float64)])
    s = padded[3 * 512:3 * 512 + 1024]
    x = np.zeros(1024)
    x[1:] = s[1:] - 0.96 * s[:-1]
    x *= 0.54 - 0.46 * np.cos(2 * 3.141592 * np.arange(1024) / 1023)
    mag = np.abs(np.fft.fft(x))[:512]
    m = np.zeros(38)
    for i in range(512):
        k = fi[i]
        if k == 0:
            m[0] += (1 - fb[i]) * mag[i]
        else:
            m[k - 1] += fb[i] * mag[i]
            if k != 38:
                m[k] += (1 - fb[i]) * mag[i]
    lm = np.log(m)
    c = np.array([np.sum(np.sqrt(2 / 38) * lm * np.cos(3.141592 * i * (np.arange(1, 39) - 0.5) / 38))
                  for i in range(1, 13)])
    c *= 1 + 11 * np.sin(3.141592 * np.arange(1, 13) / 22)
    assert np.abs(feats[2] - c).max() < 1e-9 * np.abs(c).max()
    # generic frames API agrees with the stream API
    fr = oracle.mfcc_frames(cfg, np.concatenate([np.zeros(512, np.int16), pcm]), 2 * nb - 1, first_frame=1)
    assert np.array_equal(fr, feats)


def test_mvdr_estimate_is_the_two_frame_energies(oracle):
    """EstimateSpatialCorrMtx (BF:244-270) sums |L_k|^2 / N, |R_k|^2 / N and the cross terms -Re L Im R + Im L Re R over
    all 1024 bins of two REAL frames.  The cross terms of bins k and N - k cancel and Parseval turns the others into
    sum l^2, sum r^2 -- which is what the device kernel adds (mvdr_corr_kernel).  The restatement of the reference's own
    arithmetic (FP64 transforms) must say so to FP64 rounding, also for full-scale and for silent frames."""
    rng = np.random.default_rng(12)
    for scale in (40, 3000, 32767):
        l = np.clip(np.rint(rng.normal(0, scale, 1024)), -32768, 32767).astype(np.int16)
        r = np.clip(np.rint(rng.normal(0, scale / 2, 1024)), -32768, 32767).astype(np.int16)
        c = oracle.mvdr_estimate(l, r, np.zeros(4))
        el, er = int((l.astype(np.int64) ** 2).sum()), int((r.astype(np.int64) ** 2).sum())
        assert abs(c[0] - el) <= 1e-12 * el and abs(c[3] - er) <= 1e-12 * er
        assert abs(c[1]) <= 1e-12 * el and abs(c[2]) <= 1e-12 * el
    full = np.full(1024, -32768, np.int16)
    c = oracle.mvdr_estimate(full, np.zeros(1024, np.int16), np.array([1.0, 2.0, 3.0, 4.0]))
    assert abs(c[0] - (1.0 + 1024 * 2.0 ** 30)) <= 1e-3 and c[3] == 4.0      # added to the caller's matrix; 2^40 fits FP64 exactly
