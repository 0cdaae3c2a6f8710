"""GPU parity: MFCC front end against the CPU oracle.  MelFilterBankInit's tables
are double-precision host work -> exact vs the oracle (and vs the known answers
SURVEY.md quotes); feature vectors within 1e-5 relative to the vector's peak."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def eng():
    import jeicyboodsp_amd
    e = jeicyboodsp_amd.Engine(0)
    yield e
    e.close()


def _pcm(seed, n, sigma=3000.0):
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(rng.normal(0.0, sigma, n)), -32768, 32767).astype(np.int16)


def _check(got, want):
    peak = np.abs(want).max(axis=1, keepdims=True)
    assert (np.abs(got - want) / peak).max() < TOL


def test_mel_tables_exact_and_known_answers(eng, oracle):
    m = eng.mfcc()
    mel, fi, fb = m.tables()
    omel, ofi, ofb = oracle.mel_init(oracle.mfcc_native_cfg())
    assert np.array_equal(mel, omel) and np.array_equal(fi, ofi) and np.array_equal(fb, ofb)
    assert abs(mel[0] - 65.357) < 5e-4 and abs(mel[-1] - 22050.0) < 1e-9
    assert np.bincount(fi, minlength=39)[38] == 46
    m.close()


@pytest.mark.parametrize("n_blocks", [1, 2, 9, 40])
def test_native_stream_matches_oracle(eng, oracle, n_blocks):
    """One file as the reference reads it: blocks of 1024, 2B-1 vectors (MFCC:86-104)."""
    pcm = _pcm(n_blocks, n_blocks * 1024)
    want = oracle.mfcc_stream(oracle.mfcc_native_cfg(), pcm)
    m = eng.mfcc()
    got = m.frames(pcm)                       # hop-512 framing of the file = the 2B-1 kept vectors
    assert got.shape == want.shape == (2 * n_blocks - 1, 12)
    _check(got, want)
    m.close()


def test_speechlike_levels_and_device_path(eng, oracle):
    import torch
    rng = np.random.default_rng(3)
    n = 64 * 1024
    t = np.arange(n)
    x = (6000 * np.sin(2 * np.pi * 220 * t / 44100) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t / 44100))
         + rng.normal(0, 30, n))
    pcm = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    want = oracle.mfcc_stream(oracle.mfcc_native_cfg(), pcm)
    m = eng.mfcc()
    got = m.frames(torch.from_numpy(pcm).cuda())
    torch.cuda.synchronize()
    _check(got.cpu().numpy(), want)
    m.close()


def test_baseline_config_400_160_512fft_40mel(eng, oracle):
    """BASELINE config 4: 25 ms / 10 ms framing at 16 kHz, 512-FFT, 40 mel channels."""
    kw = dict(win_len=400, hop=160, n_fft=512, n_chan=40, n_cep=13, half_rate=8000.0)
    ocfg = oracle.mfcc_cfg(n_bins=256, **kw)
    pcm = _pcm(5, 16000)
    m = eng.mfcc(**kw)
    nf = m.n_frames(pcm.size)
    assert nf == 98
    got = m.frames(pcm)
    want = oracle.mfcc_frames(ocfg, pcm, nf)
    _check(got, want)
    mel, fi, fb = m.tables()
    omel, ofi, ofb = oracle.mel_init(ocfg)
    assert np.array_equal(mel, omel) and np.array_equal(fi, ofi) and np.array_equal(fb, ofb)
    m.close()


def test_utterance_batch_with_frame_starts(eng, oracle):
    """Ragged utterances packed back to back; every utterance framed on its own."""
    cfg = oracle.mfcc_native_cfg()
    lens = [1024, 5000, 2048, 1023, 12345, 3072]
    pcm = _pcm(8, sum(lens))
    starts, want = [], []
    off = 0
    for n in lens:
        nf = (n - 1024) // 512 + 1 if n >= 1024 else 0          # empty for the 1023-sample utterance
        starts += [off + 512 * j for j in range(nf)]
        if nf:
            want.append(oracle.mfcc_frames(cfg, pcm[off:off + n], nf))
        off += n
    m = eng.mfcc()
    got = m.frames(pcm, frame_start=np.array(starts, np.int64))
    _check(got, np.concatenate(want))
    with pytest.raises(Exception):
        m.frames(pcm, frame_start=np.array([pcm.size - 100], np.int64))      # frame runs off the buffer
    m.close()


def test_silent_frame_gives_minus_inf_like_reference(eng, oracle):
    pcm = np.zeros(2048, np.int16)
    want = oracle.mfcc_stream(oracle.mfcc_native_cfg(), pcm)
    m = eng.mfcc()
    got = m.frames(pcm)
    assert np.array_equal(np.isfinite(got), np.isfinite(want)) and not np.isfinite(got).any()
    m.close()


@pytest.mark.parametrize("kw,n_bins", [
    (dict(n_chan=64, n_cep=20), 512),                                   # 65 channel indices: more pieces than lanes,
    (dict(win_len=512, hop=256, n_fft=512, n_chan=64, n_cep=13, half_rate=8000.0), 256),   # the one-frame kernel runs
    (dict(n_chan=12, n_cep=12), 512),                                   # wide channels: several 16-bin pieces each
    (dict(n_chan=1, n_cep=1), 512),
])
def test_filterbank_shapes_and_odd_frame_counts(eng, oracle, kw, n_bins):
    """Both MFCC kernels (two frames per wave with one filterbank piece per lane; one frame per wave when the
    pieces do not fit 64 lanes) and frame counts that leave the last wave half empty."""
    ocfg = oracle.mfcc_cfg(n_bins=n_bins, **kw)
    m = eng.mfcc(**kw)
    win, hop = m.cfg.win_len, m.cfg.hop
    for nf in (1, 2, 7):
        pcm = _pcm(40 + nf, win + hop * (nf - 1))
        got = m.frames(pcm)
        assert got.shape == (nf, m.cfg.n_cep)
        _check(got, oracle.mfcc_frames(ocfg, pcm, nf))
    m.close()
