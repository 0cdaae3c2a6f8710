"""GPU parity: two-microphone MVDR beamformer (BeamForming_MVDR_ver1.cpp)
against the CPU oracle.  The spatial-correlation matrix within 1e-5 relative (its off-diagonal is
pure round-off in the reference too: only its magnitude is bounded), the pre-cast output within
1e-5 of the peak, int16 within +-1 LSB."""
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


def stereo(seed, n_blocks, quiet=((0, 8), (20, 5), (40, 12))):
    rng = np.random.default_rng(seed)
    n = n_blocks * 512
    src = rng.normal(0, 3000, n)
    L = src + rng.normal(0, 300, n)
    R = 0.7 * np.roll(src, 2) + rng.normal(0, 400, n)
    for b0, nb in quiet:
        if b0 + nb <= n_blocks:
            L[b0 * 512:(b0 + nb) * 512] = rng.normal(0, 45, nb * 512)
            R[b0 * 512:(b0 + nb) * 512] = rng.normal(0, 60, nb * 512)
    cv = lambda x: np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    return cv(L), cv(R)


def check(out, pre, o_out, o_pre):
    assert out.shape == o_out.shape
    if out.size == 0:
        return
    fin = np.isfinite(o_pre)
    assert np.array_equal(np.isfinite(pre), fin)
    if fin.any():
        assert np.abs(pre[fin] - o_pre[fin]).max() < TOL * max(np.abs(o_pre[fin]).max(), 1.0)
    assert np.abs(out.astype(np.int32) - o_out.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("n_blocks", [1, 2, 9, 64])
@pytest.mark.parametrize("d_time", [0.0, 2.5e-4])
def test_mvdr_stream_matches_oracle(eng, oracle, n_blocks, d_time):
    L, R = stereo(n_blocks, n_blocks)
    o_out, o_pre, o_corr, _ = oracle.mvdr_stream(L, R, d_time)
    m = eng.mvdr(d_time)
    out, pre = m.process(L, R, want_precast=True)
    check(out, pre, o_out, o_pre)
    c = m.corr()
    scale = max(o_corr[0], o_corr[3], 1.0)
    assert abs(c[0] - o_corr[0]) <= TOL * scale and abs(c[3] - o_corr[3]) <= TOL * scale
    assert abs(c[1]) <= 1e-6 * scale and abs(c[2]) <= 1e-6 * scale      # sum_k Im(L conj R) = 0 for real signals
    m.close()


def test_mvdr_is_a_scalar_mix_at_zero_delay(eng, oracle):
    """With dTime = 0 the whole chain collapses to out = w0*L + w1*R with w = R^-1 1 / (1^T R^-1 1)."""
    L, R = stereo(3, 40)
    m = eng.mvdr(0.0)
    out, pre = m.process(L, R, want_precast=True)
    _, _, _, trace = oracle.mvdr_stream(L, R, 0.0)
    for b in (15, 30, 39):
        r00, r11 = trace[b][0], trace[b][3]
        mix = (r11 * L[b * 512:(b + 1) * 512] + r00 * R[b * 512:(b + 1) * 512]) / (r00 + r11)
        assert np.abs(pre[(b - 1) * 512:b * 512] - mix).max() < TOL * np.abs(mix).max()
    m.close()


def test_mvdr_before_any_estimate_and_chunked_calls(eng, oracle):
    L, R = stereo(4, 50, quiet=((25, 6),))          # starts loud: R = 0 -> singular -> NaN -> (short) 0
    o_out, o_pre, _, _ = oracle.mvdr_stream(L, R, 0.0)
    assert np.isnan(o_pre[:20 * 512]).all() and np.all(o_out[:20 * 512] == 0)
    m = eng.mvdr(0.0)
    outs, pres = [], []
    pos = 0
    for n in [1, 1, 3, 20, 2, 23]:                  # the reference calls once per block
        o, p = m.process(L[pos * 512:(pos + n) * 512], R[pos * 512:(pos + n) * 512], want_precast=True)
        outs.append(o); pres.append(p); pos += n
    check(np.concatenate(outs), np.concatenate(pres), o_out, o_pre)
    m.reset()
    o, p = m.process(L, R, want_precast=True)
    check(o, p, o_out, o_pre)
    m.close()


def test_mvdr_device_path(eng, oracle):
    import torch
    L, R = stereo(5, 300)
    o_out, o_pre, _, _ = oracle.mvdr_stream(L, R, 1e-4)
    m = eng.mvdr(1e-4)
    out, pre = m.process(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda(), want_precast=True)
    torch.cuda.synchronize()
    check(out.cpu().numpy(), pre.cpu().numpy(), o_out, o_pre)
    m.close()


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_mvdr_sharded_equals_single(eng, oracle, world):
    """Multi-GPU path on one GPU: simulated ranks, all-gathers = concatenations."""
    import torch
    from jeicyboodsp_amd import sharding
    nb = 150
    L, R = stereo(9, nb, quiet=((0, 8), (20, 5), (40, 12), (70, 30), (120, 9)))
    o_out, _, _, _ = oracle.mvdr_stream(L, R, 1e-4)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    ranks = []
    for r in range(world):
        b0, cnt = sharding.split_even(nb, r, world)
        ext0 = max(b0 - 1, 0)
        ranks.append(dict(m=eng.mvdr(1e-4), ext0=ext0, b0=b0, b1=b0 + cnt,
                          l=tl[ext0 * 512:(b0 + cnt) * 512].clone(), r=tr[ext0 * 512:(b0 + cnt) * 512].clone()))
    flags = torch.cat([k["m"].shard_vad(k["l"], k["r"], k["ext0"], k["b0"], k["b1"], nb) for k in ranks]).contiguous()
    sums = torch.stack([k["m"].shard_summary(flags) for k in ranks]).contiguous()
    outs = [k["m"].shard_finish(sums, world, r) for r, k in enumerate(ranks)]
    torch.cuda.synchronize()
    got = torch.cat(outs).cpu().numpy()
    for k in ranks:
        k["m"].close()
    assert got.shape == o_out.shape
    assert np.abs(got.astype(np.int32) - o_out.astype(np.int32)).max() <= 1


def test_estimate_and_apply_on_their_own_in_any_order(eng, oracle):
    """EstimateSpatialCorrMtx (:244-270) and ProcessMVDR (:124-205) as separately callable entries
    (jdsp_mvdr_estimate_corr / jdsp_mvdr_apply): the first ADDS a frame's contribution to the caller's matrix,
    the second takes its weights from the caller's matrix -- in main()'s order or not."""
    L, R = stereo(31, 12)
    m = eng.mvdr(2.5e-4)
    # apply with a hand-set matrix, no estimate ever called
    corr = np.array([4.0e6, 1.5e5, -2.5e5, 3.0e6])
    out, pre = m.apply(L[:5 * 512], R[:5 * 512], corr, want_precast=True)
    o_out, o_pre = oracle.mvdr_apply(L[:5 * 512], R[:5 * 512], corr, 2.5e-4)
    check(out, pre, o_out, o_pre)
    assert out.size == 4 * 512                                               # :201-204: the first call returns false
    # estimate accumulates into whatever the caller holds; frames need not be consecutive blocks
    frames_l = np.stack([np.concatenate([L[3 * 512:4 * 512], L[4 * 512:5 * 512]]), np.concatenate([L[7 * 512:8 * 512], L[5 * 512:6 * 512]])])
    frames_r = np.stack([np.concatenate([R[3 * 512:4 * 512], R[4 * 512:5 * 512]]), np.concatenate([R[7 * 512:8 * 512], R[5 * 512:6 * 512]])])
    got = m.estimate_corr(frames_l, frames_r, corr)
    want = oracle.mvdr_estimate(frames_l[1], frames_r[1], oracle.mvdr_estimate(frames_l[0], frames_r[0], corr))
    assert np.abs(got - want).max() <= TOL * np.abs(want).max()
    assert np.abs(got - corr).max() > 1e3                                     # something was added
    one = m.estimate_corr(frames_l[0], frames_r[0], np.zeros(4))
    assert np.abs(one - oracle.mvdr_estimate(frames_l[0], frames_r[0], np.zeros(4))).max() <= TOL * np.abs(one).max()
    # the sums are integers (two frame energies): full-scale frames, the largest they get, come out exactly
    fs_l = np.full((3, 1024), -32768, np.int16)
    fs_r = np.full((3, 1024), 32767, np.int16)
    fs = m.estimate_corr(fs_l, fs_r, np.zeros(4))
    assert fs[0] == 3 * 1024 * 2.0 ** 30 and fs[3] == 3 * 1024 * 32767.0 ** 2 and fs[1] == 0.0 and fs[2] == 0.0
    # the stream continues (keep buffers carried) with the new matrix
    out2, pre2 = m.apply(L[5 * 512:9 * 512], R[5 * 512:9 * 512], got, want_precast=True)
    st_out, st_pre = oracle.mvdr_apply(L[4 * 512:9 * 512], R[4 * 512:9 * 512], want, 2.5e-4)    # block 4 primes the keep buffers
    check(out2, pre2, st_out, st_pre)
    # a singular matrix gives the reference's NaN -> 0 samples
    z, zp = m.apply(L[9 * 512:11 * 512], R[9 * 512:11 * 512], np.zeros(4), want_precast=True)
    assert not np.isfinite(zp).any() and not z.any()
    m.close()


def test_pause_heavy_stream_uses_the_parallel_prefix(eng, oracle):
    """Thousands of EstimateSpatialCorrMtx events in one call: the running matrix comes from the tiled scan (1,024 events
    per workgroup, then the tiles' bases), not from the four-thread walk of short event lists."""
    nb = 3000
    quiet = tuple((b0, 19) for b0 in range(0, nb - 25, 24))             # ~2,300 quiet blocks
    L, R = stereo(33, nb, quiet=quiet)
    o_out, o_pre, o_corr, trace = oracle.mvdr_stream(L, R, 0.0)
    m = eng.mvdr(0.0)
    out, pre = m.process(L, R, want_precast=True)
    check(out, pre, o_out, o_pre)
    c = m.corr()
    scale = max(o_corr[0], o_corr[3], 1.0)
    assert abs(c[0] - o_corr[0]) <= TOL * scale and abs(c[3] - o_corr[3]) <= TOL * scale
    m.close()


def test_running_matrix_agrees_with_the_fp64_restatement_to_1e12(eng, oracle):
    """EstimateSpatialCorrMtx's sums over the 1024 bins of two real frames are, by Parseval and the Hermitian symmetry of
    the spectra, (sum l^2, 0, 0, sum r^2): the device adds those integers.  The restatement of the reference's
    own arithmetic (FP64 transforms, :244-270) must then agree to FP64 rounding -- over a stream, a handful of events and
    thousands -- and its cross terms must be nothing but that rounding."""
    for nb, quiet in ((64, ((0, 8), (20, 5), (40, 12))), (3000, tuple((b0, 19) for b0 in range(0, 2975, 24)))):
        L, R = stereo(91 + nb, nb, quiet=quiet)
        _, _, o_corr, _ = oracle.mvdr_stream(L, R, 0.0)
        m = eng.mvdr(0.0)
        m.process(L, R)
        c = m.corr()
        m.close()
        scale = max(o_corr[0], o_corr[3])
        assert scale > 1e6
        assert abs(c[0] - o_corr[0]) <= 1e-12 * scale and abs(c[3] - o_corr[3]) <= 1e-12 * scale
        assert c[1] == 0.0 and c[2] == 0.0
        assert abs(o_corr[1]) <= 1e-12 * scale and abs(o_corr[2]) <= 1e-12 * scale


@pytest.mark.parametrize("world", [2, 3])
def test_mvdr_sharded_pause_heavy(eng, oracle, world):
    """Sharded run whose ranks each hold more than a tile (1,024) of estimation events: the tiled prefix with a rank's
    event range and the earlier ranks' sums carried in."""
    import torch
    from jeicyboodsp_amd import sharding
    nb = 4200
    quiet = tuple((b0, 21) for b0 in range(0, nb - 25, 24))
    L, R = stereo(71, nb, quiet=quiet)
    o_out, _, _, trace = oracle.mvdr_stream(L, R, 1e-4)
    updated = np.diff(trace[:, 0]) != 0                                  # blocks whose matrix moved = events
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    ranks = []
    for r in range(world):
        b0, cnt = sharding.split_even(nb, r, world)
        assert updated[b0:b0 + cnt].sum() > 1024
        ext0 = max(b0 - 1, 0)
        ranks.append(dict(m=eng.mvdr(1e-4), ext0=ext0, b0=b0, b1=b0 + cnt,
                          l=tl[ext0 * 512:(b0 + cnt) * 512].clone(), r=tr[ext0 * 512:(b0 + cnt) * 512].clone()))
    flags = torch.cat([k["m"].shard_vad(k["l"], k["r"], k["ext0"], k["b0"], k["b1"], nb) for k in ranks]).contiguous()
    sums = torch.stack([k["m"].shard_summary(flags) for k in ranks]).contiguous()
    outs = [k["m"].shard_finish(sums, world, r) for r, k in enumerate(ranks)]
    torch.cuda.synchronize()
    got = torch.cat(outs).cpu().numpy()
    for k in ranks:
        k["m"].close()
    assert got.shape == o_out.shape
    assert np.abs(got.astype(np.int32) - o_out.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("n_quiet,nb", [(1021, 1061), (1024, 1064), (1027, 1067), (1500, 5990), (1500, 6010),
                                        (8189, 33000), (8195, 33000)])
def test_weight_table_boundary(eng, oracle, n_quiet, nb):
    """A call reads its per-bin weights from the per-version table (filled 1,024 versions per pass) when it has fewer than
    1,024 events, or fewer than 8,192 and fewer than a quarter of its blocks; otherwise it computes them per block.  Both
    sides of each of the three boundaries against the oracle (one quiet run of n blocks = n - 1 or n events)."""
    L, R = stereo(50 + n_quiet, nb, quiet=((20, n_quiet),))
    o_out, o_pre, o_corr, trace = oracle.mvdr_stream(L, R, 2.5e-4)
    m = eng.mvdr(2.5e-4)
    out, pre = m.process(L, R, want_precast=True)
    check(out, pre, o_out, o_pre)
    m.close()
