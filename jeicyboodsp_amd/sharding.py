"""Data-parallel sharding of the spectral path over the GPUs of one node.

Frames, convolution blocks and utterances are independent once each shard carries
its input halo, so there is NO data-path collective: rank r transforms its own
contiguous slice.  The only collective is the optional gather of the outputs
(RCCL all_gather when the caller wants the whole buffer on every rank).

Pure index arithmetic + torch.distributed calls; works with any backend
(tests run it on CPU with gloo, world_size 2).
"""
from collections import namedtuple

Shard = namedtuple("Shard", "first count sample_first sample_count")


def split_even(n, rank, world):
    """Contiguous, balanced split of n units: the first n % world ranks get one more."""
    base, rem = divmod(n, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def stft_shard(n_frames, rank, world, n_fft=1024, hop=512):
    """Frames [first, first+count) and the PCM slice they need: frame f covers
    samples [hop*f, hop*f + n_fft), so neighbouring shards overlap by n_fft - hop samples."""
    first, count = split_even(n_frames, rank, world)
    if count == 0:
        return Shard(first, 0, hop * first, 0)
    return Shard(first, count, hop * first, hop * (count - 1) + n_fft)


def fastconv_shard(n_out_blocks, rank, world, block, n_taps):
    """Output blocks [first, first+count) of an overlap-save stream (numbered from the first
    EMITTED block) and the input samples they need: n_taps - 1 samples of history in front."""
    first, count = split_even(n_out_blocks, rank, world)
    if count == 0:
        return Shard(first, 0, first * block, 0)
    return Shard(first, count, first * block - (n_taps - 1), count * block + (n_taps - 1))


def utterance_shard(frames_per_utt, rank, world):
    """Whole utterances per rank, contiguous, balanced by frame count (prefix-sum cut points).
    Returns (first_utt, n_utt)."""
    prefix = [0]
    for f in frames_per_utt:
        prefix.append(prefix[-1] + f)
    total = prefix[-1]
    cuts = [0]
    u = 0
    for r in range(1, world):
        target = total * r / world
        while u < len(frames_per_utt) and abs(prefix[u + 1] - target) <= abs(prefix[u] - target):
            u += 1
        cuts.append(u)
    cuts.append(len(frames_per_utt))
    return cuts[rank], cuts[rank + 1] - cuts[rank]


def all_gather_rows(local, counts, dist, group=None):
    """Gathers row-sharded tensors of unequal length: local is [counts[rank], ...]; returns
    [sum(counts), ...] on every rank.  Shards are padded to the longest so a single
    all_gather_into_tensor moves everything (one large collective, not world_size small ones)."""
    import torch
    world = len(counts)
    longest = max(counts)
    tail = tuple(local.shape[1:])
    padded = local
    if local.shape[0] != longest:
        padded = torch.zeros((longest,) + tail, dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    out = torch.empty((world * longest,) + tail, dtype=local.dtype, device=local.device)
    real = padded.is_complex()
    if real:
        dist.all_gather_into_tensor(torch.view_as_real(out), torch.view_as_real(padded.contiguous()), group=group)
    else:
        dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if all(c == longest for c in counts):
        return out
    return torch.cat([out[r * longest: r * longest + counts[r]] for r in range(world)])


def denoise_shard_range(n_blocks, rank, world):
    """Blocks [b0, b1) a rank owns and the first block it must be given (two halo blocks)."""
    b0, count = split_even(n_blocks, rank, world)
    return max(b0 - 2, 0), b0, b0 + count


def denoise_sharded(denoiser, pcm_ext, ext0, b0, b1, n_total, rank, world, gather):
    """One rank's part of a spectral-subtraction / Wiener run over ONE global stream.

    pcm_ext: torch int16 CUDA tensor with global blocks [ext0, b1).  gather(t, counts) must return
    the concatenation over ranks of each rank's t (counts[r] rows of it): with torch.distributed,
    ``lambda t, c: all_gather_rows(t, c, dist)``.  Three small exchanges (voice flags, one
    (alpha, beta[1024]) pair per rank, one latched estimate per rank); the heavy kernels see
    only the rank's own blocks.  Returns the rank's emitted int16 blocks."""
    own = [split_even(n_total, r, world)[1] for r in range(world)]
    flags_own = denoiser.shard_vad(pcm_ext, ext0, b0, b1, n_total)
    flags_all = gather(flags_own.reshape(-1, 1), own).reshape(-1).contiguous()
    summary = denoiser.shard_summary(flags_all)
    summaries = gather(summary.reshape(1, -1), [1] * world).contiguous()
    last = denoiser.shard_rows(summaries, world, rank)
    lasts = gather(last.reshape(1, -1), [1] * world).contiguous()
    return denoiser.shard_finish(lasts, world, rank)


def fastconv_shard_blocks(n_blocks, hist_blocks, rank, world):
    """Input blocks a rank must be given to produce its share of an overlap-save stream.
    The stream of n_blocks input blocks emits n_blocks - hist_blocks output blocks (the first
    hist_blocks inputs only prime the history); emitted block e is input block e + hist_blocks.
    Returns (first_input_block, n_input_blocks, first_emitted, n_emitted): the rank reads
    hist_blocks halo blocks in front of its own."""
    n_out = max(n_blocks - hist_blocks, 0)
    e0, cnt = split_even(n_out, rank, world)
    if cnt == 0:
        return e0, 0, e0, 0
    return e0, cnt + hist_blocks, e0, cnt


def fastconv_sharded(conv, pcm, n_blocks, rank, world):
    """One rank's emitted blocks of conv applied to the whole stream `pcm` (int16 tensor/array the
    rank can slice: only its own blocks plus the halo are touched).  [n_filters, n_emitted*block]."""
    first_in, n_in, e0, cnt = fastconv_shard_blocks(n_blocks, conv.hist_blocks, rank, world)
    if cnt == 0:
        return None
    conv.set_position(first_in)
    out = conv.process(pcm[first_in * conv.block:(first_in + n_in) * conv.block])
    # position first_in >= hist_blocks: every fed block emits; the first hist_blocks of them saw a silent history
    drop = out.shape[1] - cnt * conv.block
    return out[:, drop:]


def utterance_batch_shard(utt_first, rank, world):
    """GMM scoring / HMM recursion over a batch of utterances (jdsp_gmm_score_dev, jdsp_hmm_viterbi_dev):
    whole utterances per rank, balanced by vector count; no halo, no collective on the data path.
    utt_first: the n_utts + 1 offsets of the whole batch (host sequence).  Returns
    (first_utt, n_utt, vec_lo, vec_hi, local_first): this rank scores vectors [vec_lo, vec_hi) with the
    offsets local_first (rebased to 0); its rows of the [n_utts, n_classes] result are
    [first_utt, first_utt + n_utt).  all_gather_rows() assembles the full table if every rank wants it."""
    lens = [int(utt_first[u + 1]) - int(utt_first[u]) for u in range(len(utt_first) - 1)]
    first_utt, n_utt = utterance_shard(lens, rank, world)
    lo = int(utt_first[first_utt])
    hi = int(utt_first[first_utt + n_utt])
    local_first = [int(utt_first[first_utt + k]) - lo for k in range(n_utt + 1)]
    return first_utt, n_utt, lo, hi, local_first


def mfcc_utterance_shard(utt_sample_first, win_len, hop, rank, world):
    """MFCC over a batch of utterances packed back to back (BASELINE config 4: "10 000-utterance batch sharded across
    1/2/4/8 GPUs"): whole utterances per rank, contiguous, balanced by FRAME count; no halo (an utterance is framed on
    its own, MFCCFeatureExtraction_auto_version1.cpp:68-101 opens one file per utterance), no data-path collective.
    utt_sample_first: the n_utts + 1 sample offsets of the whole batch (host sequence).
    Returns (first_utt, n_utt, sample_lo, sample_hi, frame_lo, frame_hi, local_starts): the rank is given samples
    [sample_lo, sample_hi), computes frames [frame_lo, frame_hi) of the batch's frame list, and local_starts (a numpy
    int64 array) are its frames' first samples relative to sample_lo -- what jdsp_mfcc_frames_dev takes as frame_start."""
    import numpy as np
    offs = np.asarray(utt_sample_first, dtype=np.int64)
    lens = offs[1:] - offs[:-1]
    nf = np.where(lens >= win_len, (lens - win_len) // hop + 1, 0)
    first_utt, n_utt = utterance_shard([int(v) for v in nf], rank, world)
    frame_first = np.concatenate([[0], np.cumsum(nf)])
    lo, hi = int(offs[first_utt]), int(offs[first_utt + n_utt])
    starts = [offs[u] - lo + hop * np.arange(nf[u], dtype=np.int64) for u in range(first_utt, first_utt + n_utt)]
    local_starts = np.concatenate(starts) if starts else np.zeros(0, np.int64)
    return first_utt, n_utt, lo, hi, int(frame_first[first_utt]), int(frame_first[first_utt + n_utt]), local_starts
