"""The beamformers at the sizes their benchmark figures are quoted on.

tools/bench_chains.py times jdsp_mvdr_process_dev on 65,536 blocks of 512 (2 microphones,
BeamForming_MVDR_ver1.cpp:124-270) and jdsp_mvdrn_process_dev on 32,768 blocks of 256 (8 microphones on 512-point
frames, BASELINE config 5 as worded); the other parity tests stop at 33,000 and 260 blocks.  Here those two calls are
made once and EVERY block is compared with the oracle (pre-cast 1e-5 of the peak, int16 +-1 LSB): the weight table
sized by the call, the XCD-aware block mapping and the event bookkeeping across all 64 / 32 plan tiles are in play.
The streams are the bench's (a quiet head, then loud) plus pauses placed deep into the call and right at its end.
The oracle is single-threaded FP64: ~10 s and ~40 s.
"""
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


def _i16(x):
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def _check(out, pre, o_out, o_pre):
    assert out.shape == o_out.shape and pre.shape == o_pre.shape
    fin = np.isfinite(o_pre)
    assert np.array_equal(np.isfinite(pre), fin)
    assert np.abs(pre[fin] - o_pre[fin]).max() < TOL * np.abs(o_pre[fin]).max()
    assert np.abs(out.astype(np.int32) - o_out.astype(np.int32)).max() <= 1


def test_two_microphone_mvdr_at_65536_blocks(eng, oracle):
    import torch
    nb = 65536
    rng = np.random.default_rng(65536)
    # tiled noise tables (drawing 2 x 33.5 M normal deviates is most of a minute); different periods per channel
    src = np.tile(rng.normal(0, 3000, 1 << 20), 33)[:nb * 512]
    L = src + np.tile(rng.normal(0, 300, (1 << 20) + 4099), 33)[:nb * 512]
    R = 0.7 * np.roll(src, 2) + np.tile(rng.normal(0, 400, (1 << 20) + 977), 33)[:nb * 512]
    for b0, n in ((0, 12), (29990, 14), (40000, 3), (65525, 11)):          # pauses: head, deep inside, too short to latch, at the very end
        L[b0 * 512:(b0 + n) * 512] = rng.normal(0, 45, n * 512)
        R[b0 * 512:(b0 + n) * 512] = rng.normal(0, 60, n * 512)
    L, R = _i16(L), _i16(R)
    o_out, o_pre, o_corr, _ = oracle.mvdr_stream(L, R, 2.5e-4)
    m = eng.mvdr(2.5e-4)
    out, pre = m.process(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda(), want_precast=True)
    torch.cuda.synchronize()
    assert out.numel() == (nb - 1) * 512
    _check(out.cpu().numpy(), pre.cpu().numpy(), o_out, o_pre)
    c = m.corr()
    scale = max(o_corr[0], o_corr[3])
    assert abs(c[0] - o_corr[0]) <= TOL * scale and abs(c[3] - o_corr[3]) <= TOL * scale
    m.close()


def test_eight_microphone_512_point_mvdr_at_32768_blocks(eng, oracle):
    import torch
    nb, n_mics = 32768, 8
    rng = np.random.default_rng(32768)
    n = nb * 256
    src = np.tile(rng.normal(0, 3000, 1 << 19), n // (1 << 19) + 1)[:n]
    interf = np.tile(rng.normal(0, 30, (1 << 19) + 1031), n // (1 << 19) + 1)[:n]
    pcm = np.stack([np.roll(src, m) for m in range(n_mics)])
    for b0, k in ((0, 28), (16000, 24), (32750, 18)):                        # quiet: sources off, sensor noise + interferer stay
        pcm[:, b0 * 256:(b0 + k) * 256] = 0
    noise = np.stack([np.tile(rng.normal(0, 20, (1 << 18) + 17 * (m + 1)), n // (1 << 18) + 1)[:n] for m in range(n_mics)])
    pcm = _i16(pcm + noise + np.stack([np.roll(interf, -2 * m) for m in range(n_mics)]))
    delays = -np.arange(n_mics) / 16000.0
    o_out, o_pre = oracle.mvdrn_stream(pcm, delays, 1e-3, n_fft=512)
    m = eng.mvdr_multi(n_mics, delays, 1e-3, n_fft=512)
    out, pre = m.process(torch.from_numpy(pcm).cuda(), want_precast=True)
    torch.cuda.synchronize()
    assert out.numel() == (nb - 1) * 256
    _check(out.cpu().numpy(), pre.cpu().numpy(), o_out, o_pre)
    m.close()
