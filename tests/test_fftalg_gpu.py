"""GPU parity for FFTAlgorithm_ver2.cpp's functions: Bitrev table (bit-exact vs
the golden vectors produced by the compiled reference) and batched FFTProcess
(1e-5 relative; in practice ~1e-11, the reference's truncated PI)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import jeicyboodsp_amd
    e = jeicyboodsp_amd.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("n", [512, 1024])
def test_bitrev_table_bit_exact_vs_reference_golden(eng, golden_dir, n):
    g = np.load(os.path.join(golden_dir, "fftalg_%d.npz" % n), allow_pickle=False)
    got = eng.bitrev_table(n)
    assert got.dtype == np.int16 and np.array_equal(got, g["bitrev"])


@pytest.mark.parametrize("n,bl", [(2, 2), (8, 8), (256, 512), (512, 512), (2048, 2048), (8192, 8192), (1024, 512)])
def test_bitrev_table_matches_oracle_incl_quirk(eng, oracle, n, bl):
    assert np.array_equal(eng.bitrev_table(n, bl), oracle.bitrev_table(n, bl))


@pytest.mark.parametrize("n", [512, 1024])
def test_fftprocess_vs_reference_golden(eng, golden_dir, n):
    g = np.load(os.path.join(golden_dir, "fftalg_%d.npz" % n), allow_pickle=False)
    frames = g["pcm"].reshape(-1, n).astype(np.complex128)
    fwd = eng.fft_process(frames, True)
    assert np.abs(fwd - g["fwd"]).max() < 1e-9 * np.abs(g["fwd"]).max()
    inv = eng.fft_process(g["fwd"], False)
    assert np.abs(inv - g["inv"]).max() < 1e-9 * np.abs(g["inv"]).max()
    cf = eng.fft_process(g["cin"], True)
    assert np.abs(cf - g["cfwd"]).max() < 1e-9 * np.abs(g["cfwd"]).max()
    # main()'s round trip (:62-86): (short)(re/N) of IFFT(FFT(x)) -- +-1 LSB of the reference's output
    rt = np.trunc(eng.fft_process(fwd, False).real / n).astype(np.int64).reshape(-1)
    if "main_out" in g:
        assert np.abs(rt - g["main_out"].astype(np.int64)).max() <= 1


@pytest.mark.parametrize("n", [2, 4, 64, 2048, 8192])
def test_fftprocess_sizes_and_device_path(eng, oracle, n):
    import torch
    rng = np.random.default_rng(n)
    z = rng.normal(size=(5, n)) + 1j * rng.normal(size=(5, n))
    want = oracle.fft_process(z, True)
    got = eng.fft_process(z, True)
    assert np.abs(got - want).max() < 1e-9 * np.abs(want).max()
    d = torch.from_numpy(z).cuda()
    back = eng.fft_process(eng.fft_process(d, True), False)
    torch.cuda.synchronize()
    assert np.abs(back.cpu().numpy() / n - z).max() < 1e-12 * n


def test_fftprocess_rejects_non_power_of_two(eng):
    import jeicyboodsp_amd
    with pytest.raises(jeicyboodsp_amd.JdspError):
        eng.fft_process(np.zeros(12, np.complex128))


# ---- A4: DFTProcess / IDFTProcess / IFFTProcess (FFTAlgorithm_ver2.cpp:151-184) -------------------------------
# The golden arrays are the compiled reference's outputs (tests/golden/make_golden.py).  The device evaluates the
# same sums in the same order; what differs is the last place of cos/sin (device libm vs glibc), summed over 512
# terms of magnitude <= 3e4 * 512: observed ~1e-13 of the peak, bound 1e-10.
A4_TOL = 1e-10


def test_slow_dft_family_vs_reference_golden(eng, golden_dir):
    g = np.load(os.path.join(golden_dir, "fftalg_512.npz"), allow_pickle=False)
    frames = g["pcm"].reshape(-1, 512)[:2]
    dft = eng.dft_direct(eng.DFT_I16, frames)
    assert np.abs(dft - g["dft"]).max() < A4_TOL * np.abs(g["dft"]).max()
    idft = eng.dft_direct(eng.IDFT, g["fwd"][:2])
    assert np.abs(idft - g["idft"]).max() < A4_TOL * np.abs(g["idft"]).max()
    ifft = eng.dft_direct(eng.IDFT_OVER_N, g["fwd"][:2])
    assert np.abs(ifft - g["ifft_n2"]).max() < A4_TOL * np.abs(g["ifft_n2"]).max()


def test_slow_dft_family_accumulates_into_the_output(eng, golden_dir):
    """The reference ADDS into its output arrays (:168,:178,:154): a pre-filled output must come back as
    pre-fill + transform, not be overwritten."""
    g = np.load(os.path.join(golden_dir, "fftalg_512.npz"), allow_pickle=False)
    frames = g["pcm"].reshape(-1, 512)[:2]
    k = np.arange(512)
    pre = np.broadcast_to(1e6 * (k - 2j * k), (2, 512)).copy()
    for kind, x, key in ((eng.DFT_I16, frames, "dft"), (eng.IDFT, g["fwd"][:2], "idft"),
                         (eng.IDFT_OVER_N, g["fwd"][:2], "ifft_n2")):
        got = eng.dft_direct(kind, x, accumulate_into=pre)
        assert np.abs(got - (pre + g[key])).max() < A4_TOL * np.abs(pre + g[key]).max(), key
        assert np.abs(got - g[key]).max() > 1e5                                   # really accumulated


@pytest.mark.parametrize("n", [1, 2, 12, 100, 500, 1000])
def test_slow_dft_family_any_length_vs_oracle(eng, oracle, n):
    """Not restricted to powers of two (the reference's loops are not); the oracle's restatement is bit-exact
    against the compiled reference (tests/test_oracle_golden.py)."""
    rng = np.random.default_rng(n)
    s = np.clip(np.rint(rng.normal(0, 3000, n)), -32768, 32767).astype(np.int16)
    z = rng.normal(size=n) + 1j * rng.normal(size=n)
    for kind, x, want in ((eng.DFT_I16, s, oracle.dft_process(s)), (eng.IDFT, z, oracle.idft_process(z)),
                          (eng.IDFT_OVER_N, z, oracle.ifft_process(z))):
        got = eng.dft_direct(kind, x)
        assert np.abs(got - want).max() <= A4_TOL * max(np.abs(want).max(), 1e-300)
