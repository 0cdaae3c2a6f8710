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
