"""GPU vs CPU restatement: MVDR generalised to n microphones with a per-bin covariance (BASELINE
config 5).  There is NO reference for this algorithm (the reference has 2
microphones and one 2x2 matrix for all bins -- tests/test_mvdr_gpu.py covers that): parity here is
against the build's own FP64 restatement, i.e. unpinned by construction.  1e-5 of the peak before
the cast, +-1 LSB after it."""
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


def array_scene(seed, n_mics, n_blocks, src_delay=1, quiet=((0, 14), (30, 12))):
    rng = np.random.default_rng(seed)
    n = n_blocks * 512
    src = rng.normal(0, 3000, n)
    interf = rng.normal(0, 30, n)
    pcm = np.stack([np.roll(src, src_delay * m) for m in range(n_mics)])
    for b0, nb in quiet:
        if b0 + nb <= n_blocks:
            pcm[:, b0 * 512:(b0 + nb) * 512] = 0
    pcm = pcm + rng.normal(0, 20, (n_mics, n)) + np.stack([np.roll(interf, -2 * m) for m in range(n_mics)])
    return np.clip(np.rint(pcm), -32768, 32767).astype(np.int16), src, -src_delay * np.arange(n_mics) / 16000.0


def check(out, pre, o_out, o_pre):
    assert out.shape == o_out.shape
    fin = np.isfinite(o_pre)
    assert np.array_equal(np.isfinite(pre), fin)
    if fin.any():
        assert np.abs(pre[fin] - o_pre[fin]).max() < TOL * max(np.abs(o_pre[fin]).max(), 1.0)
    assert np.abs(out.astype(np.int32) - o_out.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("n_mics", [2, 3, 5, 7, 8])
def test_matches_cpu_restatement(eng, oracle, n_mics):
    pcm, _, delays = array_scene(n_mics, n_mics, 50)
    o_out, o_pre = oracle.mvdrn_stream(pcm, delays, 1e-3)
    m = eng.mvdr_multi(n_mics, delays, 1e-3)
    out, pre = m.process(pcm, want_precast=True)
    check(out, pre, o_out, o_pre)
    m.close()


def test_distortionless_towards_the_steered_source_and_chunked_calls(eng, oracle):
    import torch
    pcm, src, delays = array_scene(5, 8, 60)
    o_out, o_pre = oracle.mvdrn_stream(pcm, delays, 1e-3)
    m = eng.mvdr_multi(8, delays, 1e-3)
    t = torch.from_numpy(pcm).cuda()
    outs, pres, pos = [], [], 0
    for n in [1, 3, 16, 40]:
        o, p = m.process(t[:, pos * 512:(pos + n) * 512].contiguous(), want_precast=True)
        outs.append(o); pres.append(p); pos += n
    torch.cuda.synchronize()
    out = torch.cat(outs).cpu().numpy()
    pre = torch.cat(pres).cpu().numpy()
    check(out, pre, o_out, o_pre)
    b = 50                                   # a loud block after both quiet stretches
    got = pre[(b - 1) * 512:b * 512].astype(np.float64)
    assert np.corrcoef(got, src[b * 512:(b + 1) * 512])[0, 1] > 0.995        # w^H c = 1: the source passes
    m.close()


def test_singular_until_enough_noise_frames_without_loading(eng, oracle):
    pcm, _, delays = array_scene(9, 4, 24, quiet=((0, 3),))    # only 2 estimation frames < 4 microphones
    o_out, o_pre = oracle.mvdrn_stream(pcm, delays, 0.0)
    m = eng.mvdr_multi(4, delays, 0.0)
    out, pre = m.process(pcm, want_precast=True)
    assert out.shape == o_out.shape
    # a rank-deficient matrix: both sides produce non-finite or meaningless weights; what is pinned is
    # that nothing crashes and the block count is right
    assert out.size == 23 * 512
    m.close()


# ---- BASELINE config 5 as worded: 8-mic array on 512-point frames (blocks of 256, KEEP_LEN 255, 257 bins) ----------
def array_scene256(seed, n_mics, n_blocks, src_delay=1, quiet=((0, 28), (60, 24))):
    """The same scene cut into 256-sample blocks.  No reference counterpart: unpinned by construction, checked
    against the build's own FP64 restatement (orc_mvdrn_stream2, n_fft = 512)."""
    rng = np.random.default_rng(seed)
    n = n_blocks * 256
    src = rng.normal(0, 3000, n)
    interf = rng.normal(0, 30, n)
    pcm = np.stack([np.roll(src, src_delay * m) for m in range(n_mics)])
    for b0, nb in quiet:
        if b0 + nb <= n_blocks:
            pcm[:, b0 * 256:(b0 + nb) * 256] = 0
    pcm = pcm + rng.normal(0, 20, (n_mics, n)) + np.stack([np.roll(interf, -2 * m) for m in range(n_mics)])
    return np.clip(np.rint(pcm), -32768, 32767).astype(np.int16), src, -src_delay * np.arange(n_mics) / 16000.0


@pytest.mark.parametrize("n_mics,n_blocks", [(8, 100), (8, 1), (8, 2), (8, 3), (2, 61), (3, 100), (5, 77), (7, 100)])
def test_512_point_frames_match_cpu_restatement(eng, oracle, n_mics, n_blocks):
    """Two microphones per forward transform, two blocks per inverse transform: odd microphone counts leave the
    last forward transform half empty, odd block counts the last inverse."""
    pcm, _, delays = array_scene256(40 + n_mics, n_mics, n_blocks)
    o_out, o_pre = oracle.mvdrn_stream(pcm, delays, 1e-3, n_fft=512)
    m = eng.mvdr_multi(n_mics, delays, 1e-3, n_fft=512)
    assert m.block == 256
    out, pre = m.process(pcm, want_precast=True)
    assert out.size == max(n_blocks - 1, 0) * 256
    if out.size:
        check(out, pre, o_out, o_pre)
    else:
        assert o_out.size == 0
    m.close()


def test_512_point_frames_chunked_device_calls_and_distortionless_response(eng, oracle):
    import torch
    pcm, src, delays = array_scene256(6, 8, 120)
    o_out, o_pre = oracle.mvdrn_stream(pcm, delays, 1e-3, n_fft=512)
    m = eng.mvdr_multi(8, delays, 1e-3, n_fft=512)
    t = torch.from_numpy(pcm).cuda()
    outs, pres, pos = [], [], 0
    for n in [1, 3, 16, 33, 67]:                      # odd chunk sizes: the block pairs of a call never line up with the last call's
        o, p = m.process(t[:, pos * 256:(pos + n) * 256].contiguous(), want_precast=True)
        outs.append(o); pres.append(p); pos += n
    torch.cuda.synchronize()
    out = torch.cat(outs).cpu().numpy()
    pre = torch.cat(pres).cpu().numpy()
    check(out, pre, o_out, o_pre)
    b = 100                                           # a loud block after both quiet stretches
    got = pre[(b - 1) * 256:b * 256].astype(np.float64)
    assert np.corrcoef(got, src[b * 256:(b + 1) * 256])[0, 1] > 0.99         # w^H c = 1: the steered source passes
    m.close()


def test_512_point_frames_singular_block_does_not_poison_its_pair(eng, oracle):
    """Without loading the per-bin matrix is singular until n_mics estimation frames have been seen: those blocks are
    NaN (-> 0 after the cast) like the reference's before its first estimate -- and ONLY those, although two blocks
    share one inverse transform."""
    pcm, _, delays = array_scene256(11, 2, 40, quiet=((5, 6),))      # blocks 1..5 have no estimate yet
    o_out, o_pre = oracle.mvdrn_stream(pcm, delays, 0.0, n_fft=512)
    m = eng.mvdr_multi(2, delays, 0.0, n_fft=512)
    out, pre = m.process(pcm, want_precast=True)
    assert not np.isfinite(o_pre[:256]).any() and np.isfinite(o_pre[-256:]).all()
    check(out, pre, o_out, o_pre)
    m.close()


def test_mvdrn_cfg_rejects_other_frame_lengths(eng):
    import jeicyboodsp_amd
    with pytest.raises(jeicyboodsp_amd.JdspError):
        eng.mvdr_multi(8, None, 0.0, n_fft=256)


@pytest.mark.parametrize("n_fft", [1024, 512])
def test_many_estimation_frames_per_call(eng, oracle, n_fft):
    """More events than the covariance update has chunks (128): several events per chunk, chunk sums, the prefix over
    chunks and the per-chunk walks all in play; the weights of every version are used by some block."""
    block = n_fft // 2
    n_blocks = 260
    quiet = tuple((b0, 17) for b0 in range(2, n_blocks - 20, 23))        # 11 pauses of 17 blocks: ~180 events
    pcm, _, delays = array_scene(21, 4, n_blocks * block // 512 + 1, quiet=())
    pcm = pcm[:, :n_blocks * block].copy()
    for b0, nb in quiet:
        pcm[:, b0 * block:(b0 + nb) * block] = np.clip(np.rint(np.random.default_rng(b0).normal(0, 25, (4, nb * block))), -200, 200)
    o_out, o_pre = oracle.mvdrn_stream(pcm, delays, 1e-3, n_fft=n_fft)
    m = eng.mvdr_multi(4, delays, 1e-3, n_fft=n_fft)
    out, pre = m.process(pcm, want_precast=True)
    check(out, pre, o_out, o_pre)
    m.close()


@pytest.mark.parametrize("n_quiet", [31, 33, 65, 127, 128, 129, 130, 257, 258])
def test_covariance_chunk_geometry_edges(eng, oracle, n_quiet):
    """Event counts on both sides of the chunked update's boundaries (128 chunks: one event per chunk up to 128, two from
    129, three from 257; a run of n quiet blocks is n - 1 or n events)."""
    n_blocks = n_quiet + 12
    pcm, _, delays = array_scene(70 + n_quiet, 3, n_blocks, quiet=((4, n_quiet),))
    o_out, o_pre = oracle.mvdrn_stream(pcm, delays, 1e-3)
    m = eng.mvdr_multi(3, delays, 1e-3)
    out, pre = m.process(pcm, want_precast=True)
    check(out, pre, o_out, o_pre)
    m.close()
