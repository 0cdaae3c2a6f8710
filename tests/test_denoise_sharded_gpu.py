"""Multi-GPU path of spectral subtraction / Wiener, exercised on ONE GPU: W ranks
are simulated by W stream objects stepped phase by phase, the three all-gathers are plain
concatenations.  The concatenated shard outputs must equal the single-GPU stream (and the
oracle) to +-1 LSB, for every way of cutting the stream -- including cuts inside a noise run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import jeicyboodsp_amd
    e = jeicyboodsp_amd.Engine(0)
    yield e
    e.close()


def speechlike(seed, n_blocks, pattern):
    rng = np.random.default_rng(seed)
    x = np.zeros(n_blocks * 512)
    b, quiet, i = 0, True, 0
    while b < n_blocks:
        n = min(pattern[i % len(pattern)], n_blocks - b)
        x[b * 512:(b + n) * 512] = rng.normal(0, 45 if quiet else 3000, n * 512)
        b += n
        quiet = not quiet
        i += 1
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def run_sharded(eng, mode, pcm, world, n_fft=1024):
    import torch
    from jeicyboodsp_amd import sharding
    B = n_fft // 2
    n_total = pcm.size // B
    t = torch.from_numpy(pcm).cuda()
    ranks = []
    for r in range(world):
        ext0, b0, b1 = sharding.denoise_shard_range(n_total, r, world)
        ranks.append(dict(d=eng.denoiser(mode, n_fft, B), ext0=ext0, b0=b0, b1=b1, pcm=t[ext0 * B:b1 * B].clone()))
    flags = [k["d"].shard_vad(k["pcm"], k["ext0"], k["b0"], k["b1"], n_total) for k in ranks]
    flags_all = torch.cat(flags).contiguous()
    summ = torch.stack([k["d"].shard_summary(flags_all) for k in ranks]).contiguous()
    last = torch.stack([k["d"].shard_rows(summ, world, r) for r, k in enumerate(ranks)]).contiguous()
    outs = [k["d"].shard_finish(last, world, r) for r, k in enumerate(ranks)]
    torch.cuda.synchronize()
    res = torch.cat(outs).cpu().numpy()
    for k in ranks:
        k["d"].close()
    return res


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_sharded_equals_single_and_oracle(eng, oracle, mode, world):
    pcm = speechlike(11, 331, [12, 9, 3, 4, 15, 7, 1, 2, 11, 30, 25, 6])
    want, _ = oracle.denoise_stream(mode, pcm)
    d = eng.denoiser(mode)
    single = d.process(pcm)
    d.close()
    got = run_sharded(eng, mode, pcm, world)
    assert got.shape == want.shape == single.shape
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1
    assert np.abs(got.astype(np.int32) - single.astype(np.int32)).max() <= 1


def test_sharded_tiny_and_uneven(eng, oracle):
    for n_blocks, world in ((3, 2), (5, 8), (17, 4), (64, 5)):
        pcm = speechlike(n_blocks, n_blocks, [11, 2, 12, 3])
        want, _ = oracle.denoise_stream(1, pcm)
        got = run_sharded(eng, 1, pcm, world)
        assert got.shape == want.shape
        if got.size:
            assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1


def test_denoise_sharded_driver_world1(eng, oracle):
    import torch
    from jeicyboodsp_amd import sharding
    pcm = speechlike(5, 120, [12, 9, 3, 4, 15])
    want, _ = oracle.denoise_stream(0, pcm)
    d = eng.denoiser(0)
    ext0, b0, b1 = sharding.denoise_shard_range(120, 0, 1)
    out = sharding.denoise_sharded(d, torch.from_numpy(pcm).cuda(), ext0, b0, b1, 120, 0, 1, lambda t, c: t)
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy().astype(np.int32) - want.astype(np.int32)).max() <= 1
    d.close()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pause_heavy_stream(eng, oracle, world):
    """Thousands of noise-estimation events per rank: the ranks' events go through the same chunked average as the
    one-GPU path (several events per chunk, latches in the middle of chunks), then the ranks' maps are folded."""
    n_blocks = 13000
    pcm = speechlike(23, n_blocks, [40, 3, 25, 1, 90, 2, 11, 5])
    want, _ = oracle.denoise_stream(0, pcm)
    got = run_sharded(eng, 0, pcm, world)
    assert got.shape == want.shape
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1


# ---- BASELINE config 3 as worded, on N GPUs: 512-point frames, hop 256 (SS:53-55 with the macros halved) ---------------
def speechlike256(seed, n_blocks, pattern):
    """Quiet / loud stretches in 256-sample blocks; the quiet ones alternate in sign (ZCR ~ 255 >= the reference's
    threshold of 200, SS:49, which a 256-sample block can only reach that way) so that the non-voice path is taken."""
    rng = np.random.default_rng(seed)
    x = np.zeros(n_blocks * 256)
    alt = np.where(np.arange(256) % 2 == 0, 1.0, -1.0)
    b, i, quiet = 0, 0, True
    while b < n_blocks:
        n = min(pattern[i % len(pattern)], n_blocks - b)
        if quiet:
            x[b * 256:(b + n) * 256] = (np.abs(rng.normal(0, 45, (n, 256))) + 14.0).ravel() * np.tile(alt, n)
        else:
            x[b * 256:(b + n) * 256] = rng.normal(0, 3000, n * 256)
        b += n
        quiet = not quiet
        i += 1
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_sharded_512_point_frames_equal_single_and_oracle(eng, oracle, mode, world):
    pcm = speechlike256(41, 333, [12, 9, 3, 4, 15, 7, 1, 2, 11, 30, 25, 6])
    want, _ = oracle.denoise_stream(mode, pcm, block=256)
    d = eng.denoiser(mode, 512, 256)
    single = d.process(pcm)
    d.close()
    got = run_sharded(eng, mode, pcm, world, n_fft=512)
    assert got.shape == want.shape == single.shape
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1
    assert np.abs(got.astype(np.int32) - single.astype(np.int32)).max() <= 1


def test_sharded_512_point_tiny_uneven_and_pause_heavy(eng, oracle):
    for n_blocks, world in ((3, 2), (5, 8), (17, 4), (64, 5)):
        pcm = speechlike256(n_blocks, n_blocks, [11, 2, 12, 3])
        want, _ = oracle.denoise_stream(1, pcm, block=256)
        got = run_sharded(eng, 1, pcm, world, n_fft=512)
        assert got.shape == want.shape
        if got.size:
            assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1
    # thousands of events per rank: odd cuts put a rank's first event on either side of a frame pair
    pcm = speechlike256(29, 13001, [40, 3, 25, 1, 90, 2, 11, 5])
    want, _ = oracle.denoise_stream(0, pcm, block=256)
    for world in (2, 3):
        got = run_sharded(eng, 0, pcm, world, n_fft=512)
        assert got.shape == want.shape
        assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1
