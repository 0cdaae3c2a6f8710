"""GPU parity: overlap-save fast convolution against the CPU oracle, at the
reference-native shape (8192-point FFT, the 7169-tap room impulse response of
FilterCoefficient.h, 1024-sample blocks) and at BASELINE config 2's shape
(1024-point FFT, 256-tap HRIR pair, 769-sample blocks).  Pre-cast output within
1e-5 relative to the peak; int16 within +-1 LSB."""
import os

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


@pytest.fixture(params=["partitioned", "8192-point"], autouse=True)
def conv_impl(request, monkeypatch):
    """The 8192-point configuration has two kernels for the same convolution (fastconv_kernels.hip): the
    uniformly partitioned one (default when the block length is a multiple of 512) and the 8192-point
    overlap-save shaped like the reference; the switch is read when a handle is created."""
    monkeypatch.setenv("JDSP_FASTCONV_PARTITIONED", "1" if request.param == "partitioned" else "0")
    return request.param


def _pcm(seed, n, sigma=2000.0):
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(rng.normal(0.0, sigma, n)), -32768, 32767).astype(np.int16)


def rir(golden_dir):
    g = np.load(os.path.join(golden_dir, "rir_taps.npz"), allow_pickle=False)
    taps = np.zeros(int(g["n_taps"]))
    taps[g["index"]] = g["value"]
    return taps


def check(out, pre, o_out, o_pre, floor=0.0):
    """floor: |x|max * sum|h|, the bound on any sample of the segment's circular convolution -- the
    scale FP32 round-off lives on when the kept samples themselves are (nearly) zero."""
    assert out.shape == o_out.shape
    if out.size == 0:
        return
    peak = max(np.abs(o_pre).max(), 0.05 * floor)
    assert np.abs(pre - o_pre).max() < TOL * peak
    assert np.abs(out.astype(np.int32) - o_out.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("n_blocks", [3, 7, 8, 15, 40])
def test_native_8192_rir(eng, oracle, golden_dir, n_blocks):
    taps = rir(golden_dir)
    pcm = _pcm(n_blocks, n_blocks * 1024)
    o_out, o_pre = oracle.fastconv_stream(pcm, taps, 8192)
    fc = eng.fastconv(taps, 8192)
    assert fc.block == 1024 and fc.hist_blocks == 7               # BLOCK_SIZE, MAX_QUEUE_SIZE
    out, pre = fc.process(pcm, want_precast=True)
    assert out.shape == (1, max(n_blocks - 7, 0) * 1024)
    check(out[0], pre[0], o_out, o_pre, floor=np.abs(pcm).max() * np.abs(taps).sum())
    fc.close()


def test_native_8192_dense_filter_and_chunked_calls(eng, oracle):
    rng = np.random.default_rng(1)
    taps = rng.normal(size=7169) * np.exp(-np.arange(7169) / 1500.0) * 0.05
    pcm = _pcm(2, 33 * 1024)
    o_out, o_pre = oracle.fastconv_stream(pcm, taps, 8192)
    fc = eng.fastconv(taps, 8192)
    outs, pres = [], []
    pos = 0
    for n in [1, 1, 4, 1, 2, 9, 15]:                               # the reference calls once per block
        o, p = fc.process(pcm[pos * 1024:(pos + n) * 1024], want_precast=True)
        outs.append(o[0]); pres.append(p[0]); pos += n
    assert pos == 33
    check(np.concatenate(outs), np.concatenate(pres), o_out, o_pre)
    fc.reset()
    o, p = fc.process(pcm, want_precast=True)
    check(o[0], p[0], o_out, o_pre)
    fc.close()


def test_config_1024_hrir_pair(eng, oracle):
    rng = np.random.default_rng(3)
    hl = rng.normal(size=256) * np.exp(-np.arange(256) / 40.0)
    hr = rng.normal(size=256) * np.exp(-np.arange(256) / 55.0)
    nb = 41
    pcm = _pcm(4, nb * 769)
    fc = eng.fastconv(np.stack([hl, hr]), 1024)
    assert fc.block == 769 and fc.hist_blocks == 1
    out, pre = fc.process(pcm, want_precast=True)
    for ear, h in enumerate((hl, hr)):
        o_out, o_pre = oracle.fastconv_stream(pcm, h, 1024)
        check(out[ear], pre[ear], o_out, o_pre)
    # per-block calls give the same stream
    fc.reset()
    pieces = [fc.process(pcm[b * 769:(b + 1) * 769])[0] for b in range(nb)]
    got = np.concatenate(pieces)
    assert np.abs(got.astype(np.int32) - out[0].astype(np.int32)).max() <= 1
    fc.close()


@pytest.mark.parametrize("n_taps", [1, 2, 512, 1024])
def test_1024_tap_count_edges(eng, oracle, n_taps):
    rng = np.random.default_rng(n_taps)
    h = rng.normal(size=n_taps) / np.sqrt(n_taps)
    block = 1024 - n_taps + 1
    nb = 12 if block > 1 else 1200
    pcm = _pcm(n_taps, nb * block)
    o_out, o_pre = oracle.fastconv_stream(pcm, h, 1024)
    fc = eng.fastconv(h, 1024)
    out, pre = fc.process(pcm, want_precast=True)
    check(out[0], pre[0], o_out, o_pre)
    fc.close()


def test_device_path_linearity_full_size(eng):
    """Size-independent property at a large batch: conv(a+b) = conv(a) + conv(b) before the cast."""
    import torch
    rng = np.random.default_rng(0)
    taps = rng.normal(size=7169) * 0.01
    nb = 2048
    a = _pcm(1, nb * 1024, 1500.0)
    b = _pcm(2, nb * 1024, 1500.0)
    fc = eng.fastconv(taps, 8192)
    res = []
    for x in (a, b, (a.astype(np.int32) + b).astype(np.int16)):
        fc.reset()
        _, p = fc.process(torch.from_numpy(x).cuda(), want_precast=True)
        torch.cuda.synchronize()
        res.append(p[0].double())
    err = (res[2] - res[0] - res[1]).abs().max().item()
    assert err < 1e-5 * res[2].abs().max().item()
    fc.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_convolution_equals_single(eng, oracle, golden_dir, world):
    """Multi-GPU path on one GPU: every simulated rank convolves its own blocks plus the
    history halo; the concatenation must equal the single-GPU stream and the oracle."""
    import torch
    from jeicyboodsp_amd import sharding
    taps = rir(golden_dir)
    nb = 45
    pcm = _pcm(world, nb * 1024)
    o_out, _ = oracle.fastconv_stream(pcm, taps, 8192)
    t = torch.from_numpy(pcm).cuda()
    parts = []
    for r in range(world):
        fc = eng.fastconv(taps, 8192)
        got = sharding.fastconv_sharded(fc, t, nb, r, world)
        if got is not None:
            parts.append(got[0])
        fc.close()
    torch.cuda.synchronize()
    res = torch.cat(parts).cpu().numpy()
    assert res.shape == o_out.shape
    assert np.abs(res.astype(np.int32) - o_out.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("n_taps,block", [(6657, 1536), (7681, 512), (513, 7680), (1, 8192), (7000, 1193)])
def test_8192_other_tap_counts_and_a_filter_pair(eng, oracle, n_taps, block):
    """Block lengths that are multiples of 512 run partitioned (1 to 16 partitions, a last partition of one
    tap, several sub-blocks per block); 7000 taps (block 1193) always takes the 8192-point kernel."""
    rng = np.random.default_rng(n_taps)
    taps = rng.normal(size=(2, n_taps)) * np.exp(-np.arange(n_taps) / 900.0) * 0.05
    hist = -(-(n_taps - 1) // block)
    n_blocks = hist + 5
    pcm = _pcm(n_taps + 1, n_blocks * block)
    fc = eng.fastconv(taps, 8192)
    assert fc.block == block and fc.hist_blocks == hist
    outs, pres = [], []
    for lo, hi in [(0, 2), (2, n_blocks)]:                           # two calls: state carried across them
        o, p = fc.process(pcm[lo * block:hi * block], want_precast=True)
        outs.append(o); pres.append(p)
    out, pre = np.concatenate(outs, axis=1), np.concatenate(pres, axis=1)
    for f in range(2):
        o_out, o_pre = oracle.fastconv_stream(pcm, taps[f], 8192)
        check(out[f], pre[f], o_out, o_pre, floor=np.abs(pcm).max() * np.abs(taps[f]).sum())
    fc.close()


@pytest.mark.parametrize("n_fft,n_taps", [(1024, 200), (8192, 7169), (8192, 7000)])
def test_three_filters_share_one_forward_transform(eng, oracle, n_fft, n_taps):
    """More filters than the HRIR pair: every kernel loops its filters over one forward spectrum (the partitioned
    one reloads the partition spectra into LDS per filter, with workgroup barriers in between)."""
    rng = np.random.default_rng(n_taps)
    taps = rng.normal(size=(3, n_taps)) * np.exp(-np.arange(n_taps) / 700.0) * 0.05
    block = n_fft - n_taps + 1
    hist = -(-(n_taps - 1) // block)
    pcm = _pcm(n_taps + 3, (hist + 6) * block)
    fc = eng.fastconv(taps, n_fft)
    out, pre = fc.process(pcm, want_precast=True)
    assert out.shape == (3, 6 * block)
    for f in range(3):
        o_out, o_pre = oracle.fastconv_stream(pcm, taps[f], n_fft)
        check(out[f], pre[f], o_out, o_pre, floor=np.abs(pcm).max() * np.abs(taps[f]).sum())
    fc.close()
