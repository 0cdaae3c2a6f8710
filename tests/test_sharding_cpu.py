"""CPU tests of the multi-GPU path's host logic (gloo, world_size 2): every rank works on its
own shard + halo, the gathered result must equal the single-process result bit for bit.  The
per-shard compute is done by the CPU oracle here -- it stands in for the device kernels, which
have their own parity tests; what is under test is partitioning, halos and the gather."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from jeicyboodsp_amd import sharding  # noqa: E402


def test_split_even_covers_everything_once():
    for n in (0, 1, 7, 8, 9, 65536, 65537):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                first, count = sharding.split_even(n, r, world)
                got += list(range(first, first + count))
            assert got == list(range(n))
            sizes = [sharding.split_even(n, r, world)[1] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_stft_shard_halo():
    for world in (1, 2, 4, 8):
        prev_end = 0
        for r in range(world):
            s = sharding.stft_shard(1000, r, world)
            assert s.first == prev_end and s.sample_first == 512 * s.first
            assert s.sample_count == 512 * (s.count - 1) + 1024
            prev_end = s.first + s.count
        assert prev_end == 1000


def test_utterance_shard_balances_by_frames():
    rng = np.random.default_rng(0)
    frames = rng.integers(1, 400, 1000).tolist()
    for world in (2, 4, 8):
        loads, seen = [], []
        for r in range(world):
            first, n = sharding.utterance_shard(frames, r, world)
            seen += list(range(first, first + n))
            loads.append(sum(frames[first:first + n]))
        assert seen == list(range(1000))
        assert max(loads) - min(loads) <= 2 * max(frames)   # every cut is within half an utterance of ideal


def _worker(rank, world, port, n_frames, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    orc = oracle_lib.load_oracle()
    rng = np.random.default_rng(0)
    pcm = np.clip(np.rint(rng.normal(0, 3000, 512 * (n_frames + 1))), -32768, 32767).astype(np.int16)
    # --- STFT: frame shards with a 512-sample halo, no collective until the gather
    s = sharding.stft_shard(n_frames, rank, world)
    local = orc.stft(pcm[s.sample_first:s.sample_first + s.sample_count], s.count) if s.count else np.zeros((0, 1024), complex)
    counts = [sharding.split_even(n_frames, r, world)[1] for r in range(world)]
    full = sharding.all_gather_rows(torch.from_numpy(local), counts, dist)
    # --- fast convolution: block shards with n_taps-1 samples of history
    taps = rng.normal(size=256)
    block, nb = 769, 12
    x = np.clip(np.rint(rng.normal(0, 2000, nb * block)), -32768, 32767).astype(np.int16)
    n_out = nb - 1
    c = sharding.fastconv_shard(n_out, rank, world, block, 256)
    xz = x.astype(np.float64).copy()
    xz[:block] = 0                                         # the reference never sees its first hist block
    lo = c.sample_first + block                            # emitted block e is input block e+1
    seg = np.zeros(c.sample_count)
    src_lo, src_hi = max(lo, 0), lo + c.sample_count
    seg[src_lo - lo:] = xz[src_lo:src_hi]
    loc = np.convolve(seg, taps)[255:255 + c.count * block]
    ccounts = [sharding.split_even(n_out, r, world)[1] * block for r in range(world)]
    conv_full = sharding.all_gather_rows(torch.from_numpy(loc).reshape(-1, 1), ccounts, dist)
    if rank == 0:
        want = orc.stft(pcm, n_frames)
        ret["stft_equal"] = bool(np.array_equal(full.numpy(), want))
        _, pre = orc.fastconv_stream(x, taps, 1024)
        ret["conv_err"] = float(np.abs(conv_full.numpy().ravel() - pre).max() / np.abs(pre).max())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [37, 64])
def test_two_rank_gloo_sharded_equals_single(n_frames):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, n_frames, ret), nprocs=2, join=True)
    assert ret["stft_equal"] is True
    assert ret["conv_err"] < 1e-9


def test_utterance_batch_shard_covers_every_utterance_once():
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 400, 97).tolist()
    first = np.concatenate([[0], np.cumsum(lens)])
    for world in (1, 2, 3, 8):
        seen, vecs = [], 0
        for r in range(world):
            u0, n, lo, hi, local = sharding.utterance_batch_shard(first, r, world)
            seen += list(range(u0, u0 + n))
            assert local[0] == 0 and local[-1] == hi - lo and len(local) == n + 1
            assert [local[k + 1] - local[k] for k in range(n)] == lens[u0:u0 + n]
            vecs += hi - lo
        assert seen == list(range(97)) and vecs == first[-1]


def test_mfcc_utterance_shard_edges():
    from jeicyboodsp_amd import sharding
    offs = [0, 1000, 1000, 1300, 5000]                                      # an empty and a too-short utterance (no frames)
    seen = []
    for r in range(3):
        u0, nu, lo, hi, f0, f1, local = sharding.mfcc_utterance_shard(offs, 400, 160, r, 3)
        seen.append((u0, nu, f0, f1))
        assert local.size == f1 - f0 and (local.size == 0 or (local.min() >= 0 and local.max() + 400 <= hi - lo))
    assert seen[0][0] == 0 and sum(s[1] for s in seen) == 4 and seen[-1][3] == 4 + 0 + 0 + 21
