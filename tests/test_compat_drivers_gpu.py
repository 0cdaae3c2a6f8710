"""GPU tests of the host side above the C ABI:
  * compat_selftest drives the reference-signature functions (FFTProcess, Bitrev,
    VoiceActivityDetection, EstimateNoiseSpectrum, SpectralSubtraction, WienerFiltering,
    AnalySisFreqDomain, MelFilterBankInit, MFCCFeatureExtraction) one block per call, exactly as
    the reference main()s do;
  * the jdsp_* drivers keep the reference programs' command lines and file formats (44-byte
    header skip or not, stale-tail final block, raw double[12] feature files, list file).
Both are compared with the CPU oracle on the same bytes."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "jeicyboodsp_amd", "compat")


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(os.path.join(COMPAT, "compat_selftest")):
        subprocess.check_call(["make", "-s", "-C", COMPAT])


def run(prog, *args):
    subprocess.run([os.path.join(COMPAT, prog)] + [str(a) for a in args], check=True,
                   stdout=subprocess.DEVNULL, timeout=300)


def speechlike(seed, n_blocks, block=512):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 3000, n_blocks * block)
    q = min(14, n_blocks // 2) * block
    x[:q] = rng.normal(0, 45, q)
    if n_blocks >= 44:
        x[30 * block:30 * block + q] = rng.normal(0, 45, q)
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def rir(golden_dir):
    g = np.load(os.path.join(golden_dir, "rir_taps.npz"), allow_pickle=False)
    taps = np.zeros(int(g["n_taps"]))
    taps[g["index"]] = g["value"]
    return taps


@pytest.mark.parametrize("what,mode", [("ss", 0), ("wf", 1)])
def test_compat_denoise_per_block_calls(tmp_path, oracle, what, mode):
    pcm = speechlike(1, 60)
    pcm.tofile(tmp_path / "in.raw")
    run("compat_selftest", what, tmp_path / "in.raw", tmp_path / "out.bin")
    got = np.fromfile(tmp_path / "out.bin", np.int16)
    want, _ = oracle.denoise_stream(mode, pcm)
    assert got.shape == want.shape
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("n", [512, 12])
def test_compat_slow_dft_family_accumulating(tmp_path, oracle, golden_dir, n):
    """DFTProcess / IDFTProcess / IFFTProcess with the reference's signatures (FFTAlgorithm_ver2.cpp:151-184),
    called with PRE-FILLED outputs (k, -2k): the reference accumulates.  n = 512 against the compiled
    reference's golden outputs, n = 12 (not a power of two) against the oracle."""
    g = np.load(os.path.join(golden_dir, "fftalg_512.npz"), allow_pickle=False)
    pcm = g["pcm"][: 2 * 512] if n == 512 else g["pcm"][: 3 * n]
    pcm.tofile(tmp_path / "in.raw")
    run("compat_selftest", "dft", tmp_path / "in.raw", tmp_path / "out.bin", n)
    got = np.fromfile(tmp_path / "out.bin", np.complex128).reshape(-1, 3, n)
    k = np.arange(n)
    pre = k - 2j * k
    for b in range(got.shape[0]):
        blk = pcm[b * n:(b + 1) * n]
        if n == 512:
            dft = g["dft"][b]
        else:
            dft = oracle.dft_process(blk)
        want_dft = pre + dft
        assert np.abs(got[b, 0] - want_dft).max() < 1e-10 * np.abs(want_dft).max()
        # the selftest feeds ITS DFTProcess output (pre-fill included) to the two inverses
        want_idft = pre + oracle.idft_process(got[b, 0])
        want_ifft = pre + oracle.ifft_process(got[b, 0])
        assert np.abs(got[b, 1] - want_idft).max() < 1e-10 * np.abs(want_idft).max()
        assert np.abs(got[b, 2] - want_ifft).max() < 1e-10 * np.abs(want_ifft).max()


def test_compat_fft_bitrev(tmp_path, oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "fftalg_512.npz"), allow_pickle=False)
    g["pcm"].tofile(tmp_path / "in.raw")
    run("compat_selftest", "fft", tmp_path / "in.raw", tmp_path / "out.bin")
    raw = np.fromfile(tmp_path / "out.bin", np.uint8)
    spec = raw[:4 * 512 * 16].view(np.complex128).reshape(4, 512)
    bits = raw[4 * 512 * 16:].view(np.int16)
    assert np.abs(spec - g["fwd"]).max() < 1e-9 * np.abs(g["fwd"]).max()
    assert np.array_equal(bits, g["bitrev"])


def test_compat_conv_and_mfcc(tmp_path, oracle, golden_dir):
    taps = rir(golden_dir)
    taps.tofile(tmp_path / "taps.f64")
    pcm = speechlike(2, 24, 1024)
    pcm.tofile(tmp_path / "in.raw")
    run("compat_selftest", "conv", tmp_path / "in.raw", tmp_path / "conv.bin", tmp_path / "taps.f64")
    got = np.fromfile(tmp_path / "conv.bin", np.int16)
    want, _ = oracle.fastconv_stream(pcm, taps, 8192)
    assert got.shape == want.shape and np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1
    run("compat_selftest", "mfcc", tmp_path / "in.raw", tmp_path / "mfcc.bin")
    feats = np.fromfile(tmp_path / "mfcc.bin", np.float64).reshape(-1, 12)
    wantf = oracle.mfcc_stream(oracle.mfcc_native_cfg(), pcm)
    assert feats.shape == wantf.shape
    assert (np.abs(feats - wantf) / np.abs(wantf).max(axis=1, keepdims=True)).max() < 1e-5


def test_driver_specsub_wiener_no_header_and_stale_tail(tmp_path, oracle):
    pcm = speechlike(3, 50)
    ragged = np.concatenate([pcm, np.array([7, -9, 11] * 50, np.int16)])          # 150 extra samples
    ragged.tofile(tmp_path / "in.raw")
    # what the reference's fread loop sees: the last block keeps the previous block's tail (SS:94)
    last = pcm[-512:].copy()
    last[:150] = ragged[-150:]
    seen = np.concatenate([pcm, last])
    for prog, mode in (("jdsp_specsub", 0), ("jdsp_wiener", 1)):
        run(prog, tmp_path / "in.raw", tmp_path / "out.raw")
        got = np.fromfile(tmp_path / "out.raw", np.int16)
        want, _ = oracle.denoise_stream(mode, seen)
        assert got.shape == want.shape == ((51 - 2) * 512,)
        assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1


def test_driver_fftalg_and_conv3d_skip_wav_header(tmp_path, oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "fftalg_512.npz"), allow_pickle=False)
    with open(tmp_path / "in.wav", "wb") as f:
        f.write(bytes(range(44)))
        f.write(g["pcm"].tobytes())
    run("jdsp_fftalg", tmp_path / "in.wav", tmp_path / "out.raw")
    got = np.fromfile(tmp_path / "out.raw", np.int16)
    assert got.shape == g["main_out"].shape
    assert np.abs(got.astype(np.int32) - g["main_out"].astype(np.int32)).max() <= 1      # vs the reference's own output
    taps = rir(golden_dir)
    taps.tofile(tmp_path / "taps.f64")
    pcm = speechlike(4, 20, 1024)
    with open(tmp_path / "c.wav", "wb") as f:
        f.write(bytes(44))
        f.write(pcm.tobytes())
    run("jdsp_conv3d", tmp_path / "c.wav", tmp_path / "c.raw", tmp_path / "taps.f64")
    got = np.fromfile(tmp_path / "c.raw", np.int16)
    want, _ = oracle.fastconv_stream(pcm, taps, 8192)
    assert got.shape == want.shape and np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1


def test_driver_mfcc_list_file_carries_state_between_files(tmp_path, oracle):
    cfg = oracle.mfcc_native_cfg()
    a, b = speechlike(5, 6, 1024), speechlike(6, 4, 1024)
    for name, x in (("a.wav", a), ("b.wav", b)):
        with open(tmp_path / name, "wb") as f:
            f.write(bytes(44))
            f.write(x.tobytes())
    with open(tmp_path / "list.txt", "w") as f:
        f.write("%s %s\n%s %s\n" % (tmp_path / "a.wav", tmp_path / "a.mfc", tmp_path / "b.wav", tmp_path / "b.mfc"))
    run("jdsp_mfcc", tmp_path / "list.txt")
    fa = np.fromfile(tmp_path / "a.mfc", np.float64).reshape(-1, 12)
    fb = np.fromfile(tmp_path / "b.mfc", np.float64).reshape(-1, 12)
    wa = oracle.mfcc_stream(cfg, a)                                     # first file: 2B-1 vectors
    # second file: the keep buffer still holds a's last 512 samples and nothing is skipped (MFCC:95,198)
    wb = oracle.mfcc_frames(cfg, np.concatenate([a[-512:], b]), 2 * 4)
    assert fa.shape == wa.shape == (11, 12) and fb.shape == wb.shape == (8, 12)
    for got, want in ((fa, wa), (fb, wb)):
        assert (np.abs(got - want) / np.abs(want).max(axis=1, keepdims=True)).max() < 1e-5


def test_driver_mvdr_and_pitch1(tmp_path, oracle):
    rng = np.random.default_rng(7)
    n = 40 * 512
    src = rng.normal(0, 3000, n)
    L = src + rng.normal(0, 300, n)
    R = 0.6 * src + rng.normal(0, 400, n)
    L[:10 * 512] = rng.normal(0, 45, 10 * 512)
    R[:10 * 512] = rng.normal(0, 60, 10 * 512)
    cv = lambda x: np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    L, R = cv(L), cv(R)
    for name, x in (("l.wav", L), ("r.wav", R)):
        with open(tmp_path / name, "wb") as f:
            f.write(bytes(44))
            f.write(x.tobytes())
    run("jdsp_mvdr", tmp_path / "l.wav", tmp_path / "r.wav", tmp_path / "o.raw")
    got = np.fromfile(tmp_path / "o.raw", np.int16)
    want, _, _, _ = oracle.mvdr_stream(L, R, 0.0)
    assert got.shape == want.shape and np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1
    res = subprocess.run([os.path.join(COMPAT, "jdsp_pitch1"), str(tmp_path / "l.wav")], check=True,
                         capture_output=True, text=True, timeout=120)
    lags = [int(l.split()[2]) for l in res.stdout.splitlines() if l.startswith("Estimation arg")]
    o_arg, o_max, o_ac = oracle.pitch_stream(L)
    lags = np.array(lags)
    assert lags.shape == o_arg.shape
    at_ours = o_ac[np.arange(len(lags)), lags]
    assert np.all((lags == o_arg) | (o_max - at_ours <= 1e-5 * (o_ac[:, 0] + 1)))


def _write_mfc_lists(tmp_path, utterances_per_list):
    """The reference's two-level list format (GMMTest:80-95): a file naming class list files, each naming .mfc files."""
    names, flat = [], []
    for c, utts in enumerate(utterances_per_list):
        lst = tmp_path / f"class{c}.txt"
        paths = []
        for k, x in enumerate(utts):
            f = tmp_path / f"c{c}_u{k}.mfc"
            np.ascontiguousarray(x, np.float64).tofile(f)
            paths.append(str(f))
            flat.append((c, x))
        lst.write_text("\n".join(paths))                                 # no trailing newline, see drivers.cpp
        names.append(str(lst))
    top = tmp_path / "test_list.txt"
    top.write_text("\n".join(names))
    return top, flat


def _run_out(prog, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([os.path.join(COMPAT, prog)] + [str(a) for a in args], check=True, env=e,
                          stdout=subprocess.PIPE, timeout=300).stdout.decode()


def test_driver_gmmtest_and_compat_recognition(tmp_path, oracle):
    import gmm_cases as gc
    classes = gc.gmm_records(41, 25)
    classes.tofile(tmp_path / "params.bin")
    utts = [[gc.vectors(50 + 3 * c + k, 5 + 7 * k + c) for k in range(1 + c % 2)] for c in range(25)]
    top, flat = _write_mfc_lists(tmp_path, utts)
    out = _run_out("jdsp_gmmtest", top, tmp_path / "params.bin").splitlines()
    lines = [ln for ln in out if "class probability" in ln or "-th result" in ln]
    assert len(lines) == len(flat) * 26
    for u, (c, x) in enumerate(flat):
        want, arg = oracle.gmm_classify(x, classes)
        blk = lines[26 * u:26 * (u + 1)]
        for k in range(25):
            assert blk[k] == " %d-th class probability %f " % (k + 1, want[k])      # GMMTest:125
        assert blk[25] == " %d -th result %d " % (c + 1, arg + 1)                    # GMMTest:127
    # per-call compat function, driven like main()'s class loop
    x = flat[3][1]
    x.tofile(tmp_path / "one.mfc")
    run("compat_selftest", "gmm", tmp_path / "one.mfc", tmp_path / "gmm.bin", tmp_path / "params.bin")
    got = np.fromfile(tmp_path / "gmm.bin", np.float64)
    want, arg = oracle.gmm_classify(x, classes)
    assert np.allclose(got[:25], want, rtol=1e-12, atol=0) and int(got[25]) == arg


def test_driver_viterbi_and_compat_hmmrecognition(tmp_path, oracle):
    import gmm_cases as gc
    model = gc.hmm_records_finite(42, 1)
    model.tofile(tmp_path / "hmm.bin")
    utts = [[gc.vectors_near(60 + k, 4 + 5 * k, model["gMMParam"][0, k % 6]) for k in range(3)]]
    top, flat = _write_mfc_lists(tmp_path, utts)
    out = _run_out("jdsp_viterbi", top, tmp_path / "hmm.bin").splitlines()
    pos = 2                                                                        # the two "-th path" lines
    for c, x in flat:
        ret, path, tr = oracle.hmm_viterbi(x, model[0])
        n = len(x)
        for i in range(n - 1, 0, -1):
            assert out[pos] == "max accumulated prob %f " % tr[path[i], i]          # Viterbi:222
            pos += 1
        assert out[pos] == "decoding result ! "
        assert out[pos + 1] == "".join("%d ," % s for s in path[:n - 1])
        assert out[pos + 2] == " 1-th class probability %f " % ret                  # Viterbi:127
        assert out[pos + 3] == " 1 -th result 1 "                                   # Viterbi:129
        pos += 4
    assert pos == len(out)
    x = flat[2][1]
    x.tofile(tmp_path / "one.mfc")
    run("compat_selftest", "hmm", tmp_path / "one.mfc", tmp_path / "hmm_out.bin", tmp_path / "hmm.bin")
    got = np.fromfile(tmp_path / "hmm_out.bin", np.float64)
    ret, _, _ = oracle.hmm_viterbi(x, model[0])
    assert np.allclose(got[0], ret, rtol=1e-12, atol=0)


def test_compat_awgn_analysis_autocorrelation(tmp_path, oracle):
    """AnalysisAdditiveWhiteGaussianNoise.cpp:98-133: the same FFT -> |X|^2 -> inverse chain as CalcPitch
    (the oracle's pitch restatement returns the lags), driven with sigma-10 noise blocks like :138."""
    rng = np.random.default_rng(77)
    noise = np.clip(np.rint(rng.normal(0.0, 10.0, 512 * 9)), -32768, 32767).astype(np.int16)
    noise.tofile(tmp_path / "n.raw")
    run("compat_selftest", "awgn", tmp_path / "n.raw", tmp_path / "ac.bin")
    got = np.fromfile(tmp_path / "ac.bin", np.float64).reshape(9, 512)
    _, _, want = oracle.pitch_stream(noise)
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want[:, 0]).max()
    # white noise: R(0) ~ energy of the two blocks in the frame, other lags far below it (the program's point)
    assert np.all(np.abs(got[1:, 1:]).max(axis=1) < 0.2 * got[1:, 0])


def test_plain_c_example_mfcc_to_gmm_on_the_device(tmp_path, oracle):
    """examples/mfcc_gmm_pipeline.c: the C ABI from C99 -- PCM -> jdsp_mfcc_frames_dev -> jdsp_gmm_score_dev with
    the vectors staying in HBM.  Checked through the Python mirror's MFCC vectors (their own parity is
    test_mfcc_gpu.py's business) scored by the oracle."""
    import gmm_cases as gc
    import jeicyboodsp_amd
    exe = os.path.join(ROOT, "examples", "mfcc_gmm_pipeline")
    if not os.path.exists(exe):
        import __graft_entry__ as ge
        ge.build_examples()
    rng = np.random.default_rng(91)
    utt_blocks, n_utts = 9, 5
    pcm = np.clip(np.rint(rng.normal(0, 3000, 512 * utt_blocks * n_utts)), -32768, 32767).astype(np.int16)
    classes = gc.gmm_records(92, 7)
    classes["mean"] *= 4.0
    classes["covariance"] *= 30.0
    pcm.tofile(tmp_path / "pcm.raw")
    classes.tofile(tmp_path / "params.bin")
    out = subprocess.run([exe, str(tmp_path / "pcm.raw"), str(tmp_path / "params.bin"), "7", str(utt_blocks)],
                         check=True, stdout=subprocess.PIPE, timeout=300).stdout.decode().splitlines()
    assert len(out) == n_utts
    eng = jeicyboodsp_amd.Engine(0)
    m = eng.mfcc()
    for u, line in enumerate(out):
        tok = line.split()
        feats = m.frames(pcm[512 * utt_blocks * u: 512 * utt_blocks * (u + 1)])       # utt_blocks - 1 frames
        want, arg = oracle.gmm_classify(feats, classes)
        got = np.array([float(t) for t in tok[2:]])
        assert int(tok[0]) == u and int(tok[1]) == arg + 1
        assert np.all(np.abs(got - want) <= 1e-12 * np.abs(want))
    m.close()
    eng.close()


def test_compat_mvdr_per_block_functions(tmp_path, oracle):
    """BeamForming_MVDR_ver1.cpp's main() loop (:83-109) on ProcessMVDR / VoiceActivityDetection /
    EstimateSpatialCorrMtx of libjeicyboo_compat_mvdr.so, one block per call."""
    if not os.path.exists(os.path.join(COMPAT, "compat_mvdr_selftest")):
        subprocess.check_call(["make", "-s", "-C", COMPAT])
    rng = np.random.default_rng(93)
    n_blocks = 70
    left = speechlike(94, n_blocks)
    right = np.clip(np.rint(0.8 * left.astype(np.float64) + rng.normal(0, 20, left.size)), -32768, 32767).astype(np.int16)
    left.tofile(tmp_path / "l.raw")
    right.tofile(tmp_path / "r.raw")
    run("compat_mvdr_selftest", tmp_path / "l.raw", tmp_path / "r.raw", tmp_path / "out.bin")
    raw = np.fromfile(tmp_path / "out.bin", np.uint8)
    got = raw[:-32].view(np.int16)
    corr = raw[-32:].view(np.float64)
    want, _, o_corr, _ = oracle.mvdr_stream(left, right)
    assert got.shape == want.shape == ((n_blocks - 1) * 512,)
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1
    assert np.abs(corr - o_corr).max() <= 1e-5 * np.abs(o_corr).max()


def test_compat_mvdr_functions_out_of_order(tmp_path, oracle):
    """ProcessMVDR with a caller-set matrix before any estimate, EstimateSpatialCorrMtx on arbitrary frames into a
    pre-filled matrix, ProcessMVDR again with the result (compat_mvdr_selftest ... ooo): each function must do its
    own job whatever the call order (BeamForming_MVDR_ver1.cpp:124-205, :244-270)."""
    if not os.path.exists(os.path.join(COMPAT, "compat_mvdr_selftest")):
        subprocess.check_call(["make", "-s", "-C", COMPAT])
    rng = np.random.default_rng(95)
    left = speechlike(96, 8)
    right = np.clip(np.rint(0.6 * np.roll(left, 3).astype(np.float64) + rng.normal(0, 200, left.size)), -32768, 32767).astype(np.int16)
    left.tofile(tmp_path / "l.raw")
    right.tofile(tmp_path / "r.raw")
    run("compat_mvdr_selftest", tmp_path / "l.raw", tmp_path / "r.raw", tmp_path / "out.bin", "ooo")
    raw = np.fromfile(tmp_path / "out.bin", np.uint8)
    got = raw[:-32].view(np.int16)
    corr = raw[-32:].view(np.float64)
    blk = lambda x, b: x[b * 512:(b + 1) * 512]
    c0 = np.array([4.0e6, 1.5e5, -2.5e5, 3.0e6])
    c1 = c0
    for a, b in ((3, 4), (7, 5)):
        c1 = oracle.mvdr_estimate(np.concatenate([blk(left, a), blk(left, b)]), np.concatenate([blk(right, a), blk(right, b)]), c1)
    assert np.abs(corr - c1).max() <= 1e-5 * np.abs(c1).max()
    # one ProcessMVDR stream of blocks 0..5: matrix c0 for blocks 0..2, c1 for 3..5 (the keep buffers run through)
    st = oracle.lib.orc_mvdr_create()
    import ctypes as C
    want = []
    ob = np.zeros(512, np.int16)
    for b in range(6):
        c = np.ascontiguousarray(c0 if b < 3 else c1)
        ok = oracle.lib.orc_mvdr_process_block(st, blk(left, b).copy().ctypes.data_as(C.POINTER(C.c_short)),
                                               blk(right, b).copy().ctypes.data_as(C.POINTER(C.c_short)), 0.0,
                                               c.ctypes.data_as(C.POINTER(C.c_double)),
                                               ob.ctypes.data_as(C.POINTER(C.c_short)), None)
        if ok:
            want.append(ob.copy())
    oracle.lib.orc_mvdr_destroy(st)
    want = np.concatenate(want)
    assert got.shape == want.shape == (5 * 512,)
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1


def test_compat_mfcc_sub_steps_and_probability(tmp_path, oracle):
    """MelFilterBank / DCT / Liftering (MFCCFeatureExtraction_auto_version1.cpp:40-42,:154-192) and probability()
    (GMMAlgorithm_Test_Auto_ver2.cpp:43,:164) with the reference's own signatures, one row / one vector per call."""
    import gmm_cases
    rng = np.random.default_rng(97)
    mag = np.abs(rng.normal(0, 4e4, (6, 512))) + 1.0
    mag.tofile(tmp_path / "abs.f64")
    run("compat_selftest", "mfccsteps", tmp_path / "abs.f64", tmp_path / "steps.bin")
    got = np.fromfile(tmp_path / "steps.bin", np.float64).reshape(6, 38 + 12 + 12)
    cfg = oracle.mfcc_native_cfg()
    mel = oracle.mel_filterbank(cfg, mag)
    feat = oracle.liftering(cfg, oracle.dct(cfg, mel))
    pre = oracle.dct(cfg, mel, accumulate_into=np.tile(100.0 + np.arange(12), (6, 1)))
    assert np.abs(got[:, :38] - mel).max() <= 1e-12 * np.abs(mel).max()
    assert np.abs(got[:, 38:50] - feat).max() <= 1e-11 * np.abs(feat).max()
    assert np.abs(got[:, 50:] - pre).max() <= 1e-11 * np.abs(pre).max()
    rec = gmm_cases.gmm_records(8, 1)
    vec = gmm_cases.vectors(9, 50)
    vec.tofile(tmp_path / "v.mfc")
    rec.tofile(tmp_path / "g.bin")
    run("compat_selftest", "prob", tmp_path / "v.mfc", tmp_path / "p.bin", tmp_path / "g.bin")
    p = np.fromfile(tmp_path / "p.bin", np.float64).reshape(50, 4)
    want = np.array([[oracle.gmm_probability(v, rec[0]["mean"][k], rec[0]["covariance"][k], rec[0]["eigenVector"][k])
                      for k in range(4)] for v in vec])
    ok = want > 0
    assert np.abs(p[ok] / want[ok] - 1).max() < 1e-10 and np.array_equal(p == 0, want == 0)


def test_compat_vad_functions_equal_the_compiled_reference(tmp_path, golden_dir):
    """VoiceActivityDetection of libjeicyboo_compat.so (SS:121-156 = WF:261-296) and of libjeicyboo_compat_mvdr.so
    (BF:207-242: frame offset 511, energy only), one block per call, against tests/golden/vad.npz = what the
    reference's own functions returned for the same blocks."""
    subprocess.check_call(["make", "-s", "-C", COMPAT])
    g = np.load(os.path.join(golden_dir, "vad.npz"), allow_pickle=False)
    g["blocks"].tofile(tmp_path / "blocks.raw")
    run("compat_selftest", "vad", tmp_path / "blocks.raw", tmp_path / "ss.bin")
    assert np.array_equal(np.fromfile(tmp_path / "ss.bin", np.uint8), g["ss_flags"])
    assert np.array_equal(g["ss_flags"], g["wf_flags"])
    run("compat_mvdr_selftest", tmp_path / "blocks.raw", tmp_path / "blocks.raw", tmp_path / "bf.bin", "vad")
    assert np.array_equal(np.fromfile(tmp_path / "bf.bin", np.uint8), g["bf_flags"])


@pytest.mark.parametrize("what,mode", [("ss", 0), ("wf", 1)])
def test_compat_denoise_per_block_calls_with_iframecount_256(tmp_path, oracle, what, mode):
    """The per-block signatures carry iFrameCount (SS:58-60): 256 = the programs with BLOCK_LEN / KEEP_LEN 256 and
    FFT_PROCESSING_SIZE 512 (BASELINE config 3 as worded), main()'s loop one block per call."""
    rng = np.random.default_rng(77)
    n_blocks = 90
    alt = np.where(np.arange(256) % 2 == 0, 1.0, -1.0)
    x = rng.normal(0, 3000, n_blocks * 256)
    for b0, n in ((0, 14), (40, 13)):                                     # sign-alternating quiet stretches: ZCR >= 200
        x[b0 * 256:(b0 + n) * 256] = (np.abs(rng.normal(0, 30, (n, 256))) + 14.0).ravel() * np.tile(alt, n)   # E ~ 350 < 700
    pcm = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    pcm.tofile(tmp_path / "in.raw")
    run("compat_selftest", what, tmp_path / "in.raw", tmp_path / "out.bin", 256)
    got = np.fromfile(tmp_path / "out.bin", np.int16)
    want, _ = oracle.denoise_stream(mode, pcm, block=256)
    _, _, flags, noises, _ = oracle.denoise_trace(mode, pcm, block=256)
    assert len(noises) >= 3 and 0 < flags.sum() < n_blocks                # estimates were latched; both VAD outcomes occur
    assert got.shape == want.shape == ((n_blocks - 2) * 256,)
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("block", [512, 256])
def test_compat_estimate_noise_spectrum_by_name(tmp_path, oracle, block):
    """EstimateNoiseSpectrum of the compat layer (SS:159-198: the frame's transform on the GPU in FP32, the running
    magnitude average and the latch on the host in FP64) against the oracle's latched estimates, one by one."""
    rng = np.random.default_rng(5 + block)
    n_blocks = 120
    alt = np.where(np.arange(block) % 2 == 0, 1.0, -1.0)
    x = rng.normal(0, 3000, n_blocks * block)
    for b0, n in ((0, 14), (30, 12), (60, 25), (100, 11)):                 # four quiet runs of >= 10 blocks: four latches
        q = rng.normal(0, 30, (n, block))
        if block == 256:
            q = (np.abs(q) + 14.0) * alt                                  # ZCR >= 200 needs a sign change at nearly every sample
        x[b0 * block:(b0 + n) * block] = q.ravel()
    pcm = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    pcm.tofile(tmp_path / "in.raw")
    run("compat_selftest", "noise", tmp_path / "in.raw", tmp_path / "noise.bin", block)
    got = np.fromfile(tmp_path / "noise.bin", np.float64).reshape(-1, 2 * block)
    _, _, _, noises, _ = oracle.denoise_trace(0, pcm, block=block)
    want = noises[1:]                                                     # [0] is the all-zero start
    assert got.shape == want.shape and want.shape[0] == 4
    assert (np.abs(got - want).max(axis=1) <= 1e-5 * np.abs(want).max(axis=1)).all()
