"""GPU parity: the MFCC kernels over randomly drawn configurations, against the CPU oracle.

The kernels are picked by the configuration -- transform length, whether the filterbank's pieces fit 64 lanes and at
which piece length (8 / 12 / 16 bins), more than 16 cepstra or not, window shorter than the transform, frames that
start at odd samples (the halfword load path) -- so the cases are drawn to hit every combination rather than listed.
Seeds are fixed: a failure names its case."""
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


def _draw(seed):
    rng = np.random.default_rng(1000 + seed)
    n_fft = int(rng.choice([512, 1024]))
    win_len = int(rng.choice([n_fft, n_fft - 1, int(rng.integers(n_fft // 4, n_fft)), 400 if n_fft == 512 else 1000]))
    hop = int(rng.choice([win_len // 2, 160, int(rng.integers(1, win_len)), 2 * int(rng.integers(1, 300)) + 1]))
    n_chan = int(rng.choice([int(rng.integers(1, 65)), 38, 40, 26, 12, 20]))
    n_cep = int(rng.integers(1, min(n_chan, 32) + 1))
    half_rate = float(rng.choice([8000.0, 4000.0, 11025.0]))
    return dict(win_len=win_len, hop=hop, n_fft=n_fft, n_chan=n_chan, n_cep=n_cep, half_rate=half_rate)


@pytest.mark.parametrize("seed", range(48))
def test_random_configuration_matches_oracle(eng, oracle, seed):
    kw = _draw(seed)
    rng = np.random.default_rng(seed)
    nf = int(rng.integers(1, 9))
    lead = int(rng.integers(0, 3))                                  # frames start `lead` samples into the buffer: odd starts too
    n = lead + kw["win_len"] + kw["hop"] * (nf - 1)
    pcm = np.clip(np.rint(rng.normal(0.0, float(rng.choice([30.0, 3000.0, 12000.0])), n)), -32768, 32767).astype(np.int16)
    starts = lead + kw["hop"] * np.arange(nf, dtype=np.int64)
    m = eng.mfcc(**kw)
    got = m.frames(pcm, frame_start=starts)
    ocfg = oracle.mfcc_cfg(n_bins=kw["n_fft"] // 2, **kw)
    want = np.concatenate([oracle.mfcc_frames(ocfg, pcm[s:s + kw["win_len"]], 1) for s in starts])
    m.close()
    assert got.shape == want.shape == (nf, kw["n_cep"])
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin), kw              # ln 0 = -inf of an empty channel spreads the same way
    if fin.any():
        peak = np.abs(np.where(fin, want, 0.0)).max(axis=1, keepdims=True)
        peak[peak == 0] = 1.0
        err = np.abs(np.where(fin, got - want, 0.0)) / peak
        assert err.max() < TOL, (kw, float(err.max()))
