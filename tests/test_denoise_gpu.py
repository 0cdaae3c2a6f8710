"""GPU parity: spectral subtraction / Wiener filter (libjdsp.so through the C ABI)
against the CPU oracle's block-by-block state machine on identical PCM.

Bars (BASELINE.json / SURVEY.md §8d): VAD decisions, energies and zero-crossing
counts are integer work -> bit-exact; the noise estimate and the pre-cast output
within 1e-5 relative to the peak; int16 output within +-1 LSB (the (short) cast
truncates, so a 1e-7 error near an integer flips one LSB)."""
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


def speechlike(seed, n_blocks, pattern=None):
    """Alternating quiet (sigma 45: non-voice) and loud (sigma 3000) stretches."""
    rng = np.random.default_rng(seed)
    x = np.zeros(n_blocks * 512)
    b = 0
    pattern = pattern or [12, 9, 3, 4, 15, 7, 1, 2, 11, 30]
    quiet = True
    i = 0
    while b < n_blocks:
        n = min(pattern[i % len(pattern)], n_blocks - b)
        x[b * 512:(b + n) * 512] = rng.normal(0, 45 if quiet else 3000, n * 512)
        b += n
        quiet = not quiet
        i += 1
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def check_stream(out, pre, o_out, o_pre):
    assert out.shape == o_out.shape
    if out.size == 0:
        return
    fin = np.isfinite(o_pre)
    assert np.array_equal(np.isfinite(pre), fin)
    peak = max(np.abs(o_pre[fin]).max(), 1.0) if fin.any() else 1.0
    assert np.abs(pre[fin] - o_pre[fin]).max() < TOL * peak
    assert np.abs(out.astype(np.int32) - o_out.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n_blocks", [1, 2, 3, 5, 40, 333])
def test_denoise_stream_matches_oracle(eng, oracle, mode, n_blocks):
    pcm = speechlike(100 + n_blocks, n_blocks)
    o_out, o_pre, flags, noises, ver = oracle.denoise_trace(mode, pcm)
    d = eng.denoiser(mode)
    d.set_option("vad_trace", 1)            # keep energies and ZCR: the traced VAD kernel
    out, pre = d.process(pcm, want_precast=True)
    check_stream(out, pre, o_out, o_pre)
    v, e, z = d.vad_trace(n_blocks)
    assert np.array_equal(v.astype(np.int32), flags)
    want_noise = noises[-1]
    assert np.abs(d.noise() - want_noise).max() <= TOL * max(want_noise.max(), 1.0)
    d.close()


def test_long_stream_crosses_plan_tiles(eng, oracle):
    """> 65,536 blocks: the run-length plan walks the batch in tiles and carries its counters."""
    n_blocks = 66000
    pcm = speechlike(42, n_blocks, pattern=[12, 9, 3, 4, 15, 7, 1, 2, 11, 30, 64, 100, 10, 5])
    o_out, o_pre, flags, noises, ver = oracle.denoise_trace(1, pcm)
    d = eng.denoiser(1)
    out, pre = d.process(pcm, want_precast=True)
    check_stream(out, pre, o_out, o_pre)
    v = d.vad_trace(n_blocks, flags_only=True)          # the flags-only VAD kernel
    assert np.array_equal(v.astype(np.int32), flags)
    assert noises.shape[0] > 100                       # many latches, on both sides of the tile boundary
    assert np.abs(d.noise() - noises[-1]).max() <= TOL * noises[-1].max()
    d.close()


def test_vad_energy_and_zcr_bit_exact(eng, oracle):
    pcm = speechlike(7, 64, pattern=[3, 2, 5, 1])
    d = eng.denoiser(0)
    d.set_option("vad_trace", 1)            # keep energies and ZCR: the traced VAD kernel
    d.process(pcm)
    v, e, z = d.vad_trace(64)
    for b in range(64):
        ov, oe, oz = oracle.vad_block(pcm[b * 512:(b + 1) * 512])
        assert v[b] == int(ov) and z[b] == oz and e[b] == int(round(oe * 1024))
    d.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_chunked_streaming_equals_one_shot_and_per_block_calls(eng, oracle, mode):
    """The reference calls its function once per block; any batching must give the same stream."""
    n_blocks = 97
    pcm = speechlike(5, n_blocks)
    o_out, o_pre, *_ = oracle.denoise_trace(mode, pcm)
    d = eng.denoiser(mode)
    pieces, pres = [], []
    pos = 0
    for n in [1, 1, 1, 2, 7, 1, 30, 1, 1, 52]:
        o, p = d.process(pcm[pos * 512:(pos + n) * 512], want_precast=True)
        pieces.append(o)
        pres.append(p)
        pos += n
    assert pos == n_blocks
    check_stream(np.concatenate(pieces), np.concatenate(pres), o_out, o_pre)
    d.reset()
    out2, pre2 = d.process(pcm, want_precast=True)
    check_stream(out2, pre2, o_out, o_pre)
    d.close()


@pytest.mark.parametrize("k", [1, 2, 4, 8])
def test_blocks_per_wave_variants_agree(eng, oracle, k):
    pcm = speechlike(9, 130)
    o_out, o_pre, *_ = oracle.denoise_trace(1, pcm)
    d = eng.denoiser(1)
    d.set_option("blocks_per_wave", k)
    out, pre = d.process(pcm, want_precast=True)
    check_stream(out, pre, o_out, o_pre)
    d.close()


def test_wiener_zero_over_zero_and_silence(eng, oracle):
    pcm = np.zeros(8 * 512, np.int16)            # digital silence before any estimate: WF:204 is 0/0
    for mode in (0, 1):
        o_out, o_pre = oracle.denoise_stream(mode, pcm)
        d = eng.denoiser(mode)
        out, pre = d.process(pcm, want_precast=True)
        assert np.array_equal(np.isnan(pre), np.isnan(o_pre))
        assert np.array_equal(out, o_out)
        d.close()


def test_device_path_full_batch_properties(eng):
    """BASELINE size (65,536 blocks): device path, determinism across batch splits, and the
    WOLA identity before any estimate latches (gain 1 => out = 1.08*in for Hamming at 50 %)."""
    import torch
    n_blocks = 65536
    rng = np.random.default_rng(0)
    pcm = np.clip(np.rint(rng.normal(0, 3000, n_blocks * 512)), -32768, 32767).astype(np.int16)
    d = eng.denoiser(0)
    t = torch.from_numpy(pcm).cuda()
    out, pre = d.process(t, want_precast=True)
    torch.cuda.synchronize()
    assert out.numel() == (n_blocks - 2) * 512
    i = np.arange(1024)
    w = 0.54 - 0.46 * np.cos(2 * 3.141592 * i / 1023)
    gain = torch.from_numpy((w[:512] + w[512:]).astype(np.float32)).cuda()
    want = t[512:-512].float().view(-1, 512) * gain            # emitted block e <-> input block e+1
    err = (pre.view(-1, 512) - want).abs().max().item()
    assert err < 1e-5 * 32768 * 1.1
    # same stream in two halves
    d.reset()
    a = d.process(t[: 512 * 30001])
    b = d.process(t[512 * 30001:])
    torch.cuda.synchronize()
    # The halo frame and the in-loop frames are separately inlined copies of the same arithmetic;
    # the compiler may contract them differently, so a different batch split can flip the
    # truncating (short) cast on a value within 1e-7 of an integer: +-1 LSB, and rare.
    diff = (torch.cat([a, b]).int() - out.int()).abs()
    assert diff.max().item() <= 1 and (diff != 0).float().mean().item() < 1e-4
    d.close()


# ---- BASELINE config 3 as worded: 512-point frames, hop 256 (FFT_PROCESSING_SIZE 512, BLOCK_LEN = KEEP_LEN 256) ----
def speechlike256(seed, n_blocks, pattern=None):
    """Quiet / loud stretches in 256-sample blocks.  The reference's ZCR threshold stays 200 (SS:49) while a
    256-sample block has at most 256 sign changes, so `dZCR < 200` calls everything voice except signals that change
    sign at nearly every sample: the quiet stretches here alternate in sign (ZCR ~ 255) so that the non-voice path
    -- run lengths, running average, latches -- is exercised at this frame size too."""
    rng = np.random.default_rng(seed)
    x = np.zeros(n_blocks * 256)
    pattern = pattern or [12, 9, 3, 4, 15, 7, 1, 2, 11, 30]
    b, i, quiet = 0, 0, True
    alt = np.where(np.arange(256) % 2 == 0, 1.0, -1.0)
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
@pytest.mark.parametrize("n_blocks", [1, 2, 3, 6, 7, 8, 15, 80, 667])
def test_denoise_512_point_frames_match_oracle(eng, oracle, mode, n_blocks):
    """Two 512-sample frames per wave transform (z = a + j b): every block count that puts the last block at a
    different place of the 7-block wave, a quiet start so the estimate latches, VAD flags bit-exact."""
    pcm = speechlike256(300 + n_blocks, n_blocks, pattern=[13, 5, 2, 3, 11, 4, 1, 1, 16, 14])
    o_out, o_pre, flags, noises, ver = oracle.denoise_trace(mode, pcm, block=256)
    d = eng.denoiser(mode, 512, 256)
    d.set_option("vad_trace", 1)            # keep energies and ZCR: the traced VAD kernel
    assert d.block == 256
    out, pre = d.process(pcm, want_precast=True)
    check_stream(out, pre, o_out, o_pre)
    v, e, z = d.vad_trace(n_blocks)
    assert np.array_equal(v.astype(np.int32), flags)
    want_noise = noises[-1]
    assert d.noise().shape == (512,)
    assert np.abs(d.noise() - want_noise).max() <= TOL * max(want_noise.max(), 1.0)
    if n_blocks >= 80:
        assert noises.shape[0] > 2 and (flags == 0).sum() > 20
    d.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_denoise_512_point_frames_chunked_calls_equal_one_call(eng, oracle, mode):
    """State carried between calls (previous block, overlap tail, run length, running average): any chunking --
    down to the reference's one block per call -- gives the stream of a single call.  NOT bit for bit at this frame
    size: two frames share one complex transform (z = a + j b), so which frame a frame is paired with -- and with it
    the last bits of its FP32 spectrum -- depends on where the call boundaries fall; what is required is the
    tolerance every stream is held to: +-1 LSB of the one-call stream and of the oracle."""
    import torch
    n_blocks = 97
    pcm = speechlike256(77, n_blocks, pattern=[13, 5, 2, 3, 11, 4, 1, 1, 16, 14])
    o_out, o_pre = oracle.denoise_stream(mode, pcm, block=256)
    whole = eng.denoiser(mode, 512, 256)
    w_out = whole.process(pcm)
    whole.close()
    assert np.abs(w_out.astype(np.int32) - o_out.astype(np.int32)).max() <= 1
    for chunks in ([1] * n_blocks, [2, 5, 1, 7, 14, 3, 65], [96, 1]):
        d = eng.denoiser(mode, 512, 256)
        got, b = [], 0
        for c in chunks:
            got.append(d.process(torch.from_numpy(pcm[b * 256:(b + c) * 256]).cuda()).cpu().numpy())
            b += c
        d.close()
        got = np.concatenate(got)
        assert got.shape == w_out.shape, chunks
        assert np.abs(got.astype(np.int32) - w_out.astype(np.int32)).max() <= 1, chunks
        assert np.abs(got.astype(np.int32) - o_out.astype(np.int32)).max() <= 1, chunks


def test_denoise_512_point_full_batch(eng, oracle):
    """65,536 blocks of 256 samples in one launch (config 3's batch), against the oracle sample for sample."""
    import torch
    n_blocks = 65536
    pcm = speechlike256(5, n_blocks, pattern=[12, 9, 3, 4, 15, 7, 1, 2, 11, 30, 64, 100, 10, 5])
    o_out, o_pre, flags, noises, ver = oracle.denoise_trace(0, pcm, block=256)
    d = eng.denoiser(0, 512, 256)
    out, pre = d.process(torch.from_numpy(pcm).cuda(), want_precast=True)
    torch.cuda.synchronize()
    check_stream(out.cpu().numpy(), pre.cpu().numpy(), o_out, o_pre)
    v = d.vad_trace(n_blocks, flags_only=True)          # the flags-only VAD kernel
    assert np.array_equal(v.astype(np.int32), flags) and noises.shape[0] > 100
    d.close()


def test_denoise_cfg_rejects_other_shapes(eng):
    import jeicyboodsp_amd
    for n_fft, hop in ((512, 512), (1024, 256), (2048, 1024), (256, 128)):
        with pytest.raises(jeicyboodsp_amd.JdspError):
            eng.denoiser(0, n_fft, hop)
    d = eng.denoiser(0)
    with pytest.raises(jeicyboodsp_amd.JdspError):
        d.set_option("blocks_per_wave", 3)
    d.close()


def test_wiener_512_point_zero_over_zero_stays_in_its_own_frame(eng, oracle):
    """WF:204's 0/0 (an all-zero frame before any noise estimate) makes that frame's inverse transform NaN in the
    reference, i.e. two output blocks; with two frames per transform the NaN frame must not leak into its partner."""
    rng = np.random.default_rng(3)
    pcm = np.clip(np.rint(rng.normal(0, 3000, 20 * 256)), -32768, 32767).astype(np.int16)
    pcm[6 * 256:8 * 256] = 0                      # frame [block 6, block 7] is all zero
    o_out, o_pre = oracle.denoise_stream(1, pcm, block=256)
    assert 0 < (~np.isfinite(o_pre)).sum() <= 3 * 256
    d = eng.denoiser(1, 512, 256)
    out, pre = d.process(pcm, want_precast=True)
    check_stream(out, pre, o_out, o_pre)
    d.close()


@pytest.mark.parametrize("n_fft,block", [(1024, 512), (512, 256)])
def test_flags_only_vad_equals_the_oracle_at_the_thresholds(eng, oracle, n_fft, block):
    """The flags-only VAD kernel (clamped 32-bit energy sum by DPP, zero crossings counted on the scalar unit) must
    make the reference's decision bit for bit: blocks whose energy straddles 700 and whose zero-crossing count
    straddles 200, plus blocks with one huge sample (a lane's partial sum clamps) and all-zero blocks."""
    rng = np.random.default_rng(123 + block)
    n_blocks = 3000
    x = np.zeros((n_blocks, block))
    for b in range(n_blocks):
        kind = b % 5
        if kind == 0:                                   # energy near the threshold, white: ZCR ~ block / 2
            x[b] = rng.normal(0, rng.uniform(40, 90), block)
        elif kind == 1:                                 # smoothed noise: fewer zero crossings, loud enough to matter
            w = rng.normal(0, 1, block + 8)
            k = int(rng.integers(1, 4))
            x[b] = np.convolve(w, np.ones(k) / k, mode="same")[:block] * rng.uniform(30, 120)
        elif kind == 2:                                 # sign-alternating: ZCR near the maximum, quiet
            x[b] = (np.abs(rng.normal(0, 30, block)) + rng.uniform(0, 20)) * np.where(np.arange(block) % 2 == 0, 1.0, -1.0)
            flip = rng.integers(0, block, int(rng.integers(0, 120)))
            x[b, flip] *= -1                            # knock the count down towards 200
        elif kind == 3:                                 # one huge sample in silence
            x[b, int(rng.integers(0, block))] = rng.choice([-32768, 32767, 20000])
        # kind 4: all zeros
    pcm = np.clip(np.rint(x), -32768, 32767).astype(np.int16).ravel()
    want = np.array([oracle.vad_block(pcm[b * block:(b + 1) * block])[0] for b in range(n_blocks)], np.uint8)
    assert 0.03 < want.mean() < 0.97                    # both decisions represented (256-sample blocks: ZCR < 200 says voice almost always)
    d = eng.denoiser(0, n_fft, block)
    d.process(pcm)
    fast = d.vad_trace(n_blocks, flags_only=True)
    assert np.array_equal(fast, want)
    d.reset()
    d.set_option("vad_trace", 1)
    d.process(pcm)
    traced, e, z = d.vad_trace(n_blocks)
    assert np.array_equal(traced, want)
    import jeicyboodsp_amd
    d.set_option("vad_trace", 0)
    d.process(pcm)
    with pytest.raises(jeicyboodsp_amd.JdspError):
        d.vad_trace(n_blocks)                           # energies / ZCR were not kept
    d.close()


# ---- pause-heavy streams: thousands of EstimateNoiseSpectrum events per call (noise_accum / noise_combine kernels) ----
@pytest.mark.parametrize("n_fft,block", [(1024, 512), (512, 256)])
def test_pause_heavy_stream_matches_oracle(eng, oracle, n_fft, block):
    """More events than noise_accum_kernel has chunks (4,096): several events per chunk, 64 groups of chunks, latches
    in the middle of chunks.  Speech is 40-60 % pauses; here nine blocks in ten are."""
    n_blocks = 9500
    gen = speechlike if block == 512 else speechlike256
    pcm = gen(77, n_blocks, pattern=[40, 3, 25, 1, 90, 2, 11, 5])
    o_out, o_pre, flags, noises, ver = oracle.denoise_trace(0, pcm, block=block)
    assert (flags == 0).sum() > 4096 + 2000 and noises.shape[0] > 50
    d = eng.denoiser(0, n_fft, block)
    out, pre = d.process(pcm, want_precast=True)
    check_stream(out, pre, o_out, o_pre)
    assert np.array_equal(d.vad_trace(n_blocks, flags_only=True).astype(np.int32), flags)
    assert np.abs(d.noise() - noises[-1]).max() <= TOL * noises[-1].max()
    # (nearly) every block quiet: runs of hundreds of events, the average halving all the way
    pcm = gen(78, 5000, pattern=[5000])
    o_out, o_pre, flags, noises, ver = oracle.denoise_trace(1, pcm, block=block)
    assert (flags == 0).mean() > 0.85
    d2 = eng.denoiser(1, n_fft, block)
    out, pre = d2.process(pcm, want_precast=True)
    check_stream(out, pre, o_out, o_pre)
    assert np.abs(d2.noise() - noises[-1]).max() <= TOL * noises[-1].max()
    d.close()
    d2.close()


def test_small_calls_keep_the_sequential_average_bit_for_bit(eng):
    """Up to 64 events per call the chunked average is the sequential one bit for bit (one event per chunk, one chunk
    per group; a power-of-two factor commutes with rounding): per-block calls and a 60-block call agree exactly."""
    n_blocks = 60
    pcm = speechlike(9, n_blocks, pattern=[14, 2, 25, 3, 16])
    a = eng.denoiser(0)
    out_a, pre_a = a.process(pcm, want_precast=True)
    b = eng.denoiser(0)
    outs, pres = [], []
    for j in range(n_blocks):
        o, p = b.process(pcm[j * 512:(j + 1) * 512], want_precast=True)
        outs.append(o)
        pres.append(p)
    assert np.array_equal(np.concatenate(outs), out_a)
    assert np.array_equal(np.concatenate(pres), pre_a)
    assert np.array_equal(a.noise(), b.noise())
    a.close()
    b.close()


@pytest.mark.parametrize("n_events_target", [63, 64, 65, 66, 4090, 4101, 4104, 4200])
def test_noise_average_chunk_geometry_edges(eng, oracle, n_events_target):
    """Event counts on both sides of the chunked average's boundaries: 64 events (one chunk per group / several), 4,096
    (one event per chunk / two).  One long quiet run gives n - 1 events (run length >= 2 from its second block on)."""
    n_quiet = n_events_target + 1
    n_blocks = n_quiet + 6
    pcm = speechlike(900 + n_events_target, n_blocks, pattern=[n_quiet, 6])
    o_out, o_pre, flags, noises, ver = oracle.denoise_trace(0, pcm)
    # the quiet stretch is Gaussian: a handful of its blocks may come out as voice and split the run -- what matters is
    # that the number of events is at the boundary or within a few of it
    n_events = int(sum(max(0, r - 1) for r in _runs_of_zero(flags)))
    assert abs(n_events - n_events_target) <= 12
    d = eng.denoiser(0)
    out, pre = d.process(pcm, want_precast=True)
    check_stream(out, pre, o_out, o_pre)
    assert np.abs(d.noise() - noises[-1]).max() <= TOL * max(noises[-1].max(), 1.0)
    d.close()


def _runs_of_zero(flags):
    runs, n = [], 0
    for f in flags:
        if f == 0:
            n += 1
        else:
            if n:
                runs.append(n)
            n = 0
    if n:
        runs.append(n)
    return runs


@pytest.mark.parametrize("seed", range(12))
def test_random_streams_and_call_cuts(eng, oracle, seed):
    """Seeded random stream lengths, pause patterns and call boundaries, both frame sizes and modes: the plan's event
    list, the chunked average, the run kernels' halo logic and the state hand-over all at once, against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    block = 512 if seed % 2 == 0 else 256
    n_fft = 2 * block
    mode = (seed // 2) % 2
    n_blocks = int(rng.integers(1, 900))
    pattern = [int(v) for v in rng.integers(1, 40, size=int(rng.integers(2, 9)))]
    gen = speechlike if block == 512 else speechlike256
    pcm = gen(2000 + seed, n_blocks, pattern=pattern)
    o_out, o_pre, flags, noises, ver = oracle.denoise_trace(mode, pcm, block=block)
    d = eng.denoiser(mode, n_fft, block)
    cuts = sorted(set(int(v) for v in rng.integers(0, n_blocks + 1, size=int(rng.integers(0, 6)))) | {0, n_blocks})
    outs, pres = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        o, p = d.process(pcm[a * block:b * block], want_precast=True)
        outs.append(o)
        pres.append(p)
    check_stream(np.concatenate(outs), np.concatenate(pres), o_out, o_pre)
    assert np.abs(d.noise() - noises[-1]).max() <= TOL * max(noises[-1].max(), 1.0)
    d.close()
