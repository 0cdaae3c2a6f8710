"""BASELINE config 4's multi-GPU split on one GPU: "10 000-utterance batch sharded across 1/2/4/8 GPUs".

Every simulated rank gets whole utterances (sharding.mfcc_utterance_shard: contiguous, balanced by frame count),
runs jdsp_mfcc_frames_dev on its own PCM slice with its own frame list, and the concatenation of the ranks'
vectors must be the single-GPU batch.  There is no halo and no collective: an utterance is framed on its own
(MFCCFeatureExtraction_auto_version1.cpp:68-101 opens one file per utterance).

How equal: the 512-FFT kernel puts frames 2j and 2j+1 OF THE CALL's frame list into one complex transform
(mfcc512_pair_kernel), so a frame's partner -- and with it the FP32 rounding that leaks between the two spectra,
6e-8 of the partner's magnitudes -- depends on the parity of the frame's position in the call.  A rank whose first
frame has an even global index reproduces the batch bit for bit (up to its last frame, if it holds an odd number of
them: that one is alone in its transform); one that starts on an odd index pairs every frame with its other neighbour
and agrees to <= 2e-5 of each vector's peak (two FP32 results, each within north_star's 1e-5 of the FP64 oracle).
Both are asserted, and every rank is held to the oracle on a sample of utterances.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KW = dict(win_len=400, hop=160, n_fft=512, n_chan=40, n_cep=13, half_rate=8000.0)


@pytest.fixture(scope="module")
def eng():
    import jeicyboodsp_amd
    e = jeicyboodsp_amd.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def batch(eng):
    """The 10,000-utterance ragged batch of test_mfcc_gpu.py (1-6 s each, packed) and its single-GPU vectors."""
    import torch
    rng = np.random.default_rng(2024)
    n_utts = 10000
    lens = rng.integers(16000, 6 * 16000 + 1, n_utts)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = int(offs[-1])
    table = np.clip(np.rint(rng.normal(0, 3000, 1 << 22)), -32768, 32767).astype(np.int16)
    pcm = np.tile(table, total // table.size + 1)[:total]
    nf = (lens - 400) // 160 + 1
    starts = np.concatenate([offs[u] + 160 * np.arange(nf[u], dtype=np.int64) for u in range(n_utts)])
    m = eng.mfcc(**KW)
    d_pcm = torch.from_numpy(pcm).cuda()
    whole = m.frames(d_pcm, frame_start=torch.from_numpy(starts).cuda())
    torch.cuda.synchronize()
    yield dict(pcm=pcm, d_pcm=d_pcm, offs=offs, nf=nf, whole=whole, m=m)
    m.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_utterance_sharded_mfcc_equals_the_single_gpu_batch(eng, oracle, batch, world):
    import torch
    from jeicyboodsp_amd import sharding
    m, whole, offs, nf = batch["m"], batch["whole"], batch["offs"], batch["nf"]
    ocfg = oracle.mfcc_cfg(n_bins=256, **KW)
    covered_utts = covered_frames = 0
    exact_ranks = 0
    for r in range(world):
        u0, nu, lo, hi, f0, f1, local = sharding.mfcc_utterance_shard(offs, 400, 160, r, world)
        assert u0 == covered_utts and f0 == covered_frames                 # contiguous, nothing skipped, nothing twice
        covered_utts += nu
        covered_frames = f1
        assert abs((f1 - f0) - whole.shape[0] / world) < 600                # balanced by frames: within one utterance
        shard_pcm = batch["d_pcm"][lo:hi].clone()                           # the rank's own buffer: only its samples
        got = m.frames(shard_pcm, frame_start=torch.from_numpy(local).cuda())
        torch.cuda.synchronize()
        want = whole[f0:f1]
        assert got.shape == want.shape
        err = ((got - want).abs() / want.abs().amax(dim=1, keepdim=True)).max().item()
        # whatever the pairing: both results are FP32 chains held to 1e-5 of the FP64 oracle, so 2e-5 of each other.  (Over
        # 3.5 M noise frames the largest difference seen is 9.7e-6: a frame whose low channels happen to be 40 dB below
        # its own average takes the transform's rounding floor relative to those channels, ln() divides by them.)
        assert err < 2e-5, err
        if f0 % 2 == 0:
            # same pairs as the batch -> bit for bit; except the shard's LAST frame when its length is odd: alone in
            # its transform here, paired with the next rank's first frame in the batch
            n_same = (f1 - f0) & ~1
            assert torch.equal(got[:n_same].view(torch.int64), want[:n_same].view(torch.int64))
            exact_ranks += 1
        # and against the oracle, first and last utterance of the shard
        for u in (u0, u0 + nu - 1):
            a, b = int(np.sum(nf[u0:u])), int(np.sum(nf[u0:u + 1]))
            o = oracle.mfcc_frames(ocfg, batch["pcm"][offs[u]:offs[u + 1]], int(nf[u]))
            g = got[a:b].cpu().numpy()
            assert (np.abs(g - o) / np.abs(o).max(axis=1, keepdims=True)).max() < 1e-5
    assert covered_utts == 10000 and covered_frames == whole.shape[0]
    assert exact_ranks >= 1                                                 # rank 0 at least starts on frame 0
