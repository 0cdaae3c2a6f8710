#!/usr/bin/env python3
"""bench.py -- headline benchmark: STFT frames/s (1024-pt, 50 % overlap, Hamming)
at batch = 65,536 frames per GPU (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path (jdsp_stft_i16_dev: int16 framing + window
+ forward transform, full 1024-bin complex64 spectrum) over one batch of
synthetic PCM that is already resident in HBM.  For N > 1 the driver launches
one rank per GPU with torch.distributed.run; frames are independent, so every
rank transforms its own shard of N*65,536 frames (weak scaling, no data-path
collective); `--gather` additionally times an RCCL all_gather of the spectra and
reports it separately (never part of `value`).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_FFT, HOP, FRAMES_PER_GPU = 1024, 512, 65536
BYTES_PER_FRAME = 512 * 2 + 1024 * 8          # 9,216 B algorithmic
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: 8.0 TB/s spec


def synth_pcm(rank, n_frames, world=1):
    """SURVEY.md §8d synthetic input: normal(0, 3000) rounded and clipped to int16, (B+1)*512 samples
    per rank = B frames + the 512-sample halo (jeicyboodsp_amd.sharding.stft_shard).  Rank 0 draws
    from default_rng(0) exactly as SURVEY specifies; rank r draws its shard of the global stream from
    default_rng(r), so that an 8-rank job does not have every rank generate 8 shards of noise."""
    import numpy as np
    from jeicyboodsp_amd import sharding
    s = sharding.stft_shard(n_frames * world, rank, world, N_FFT, HOP)
    assert s.count == n_frames and s.sample_count == HOP * (n_frames - 1) + N_FFT
    rng = np.random.default_rng(rank)
    return np.clip(np.rint(rng.normal(0.0, 3000.0, s.sample_count)), -32768, 32767).astype(np.int16)


def cpu_baseline(seconds=10.0):
    """The oracle (CPU restatement of the reference algorithm, FP64, per-frame
    window recomputation, single thread like the reference) on a bounded sample."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    orc = oracle_lib.load_oracle()
    chunk = 4096
    pcm = synth_pcm(0, chunk)
    orc.stft(pcm, 64)                              # warm the twiddle cache
    done, t0 = 0, time.perf_counter()
    while True:
        orc.stft(pcm, chunk)
        done += chunk
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    one = done / dt
    # several host cores: frames are independent, static partition over threads (ctypes drops the
    # GIL).  Capped at this job's CPU share of the GPU box (16 threads for one GPU).
    import threading
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    counts = [0] * cores
    stop = time.perf_counter() + min(seconds, 8.0)

    def work(i):
        while time.perf_counter() < stop:
            orc.stft(pcm, chunk)
            counts[i] += chunk

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    [t.start() for t in th]
    [t.join() for t in th]
    allc = sum(counts) / (time.perf_counter() - t0)
    return {"value": one, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames (chunks of %d from the same synthetic stream), FP64 oracle, single thread" % (done, chunk),
            "multi_thread": {"value": allc, "cores": cores, "host_cores_visible": avail}}


def read_traffic():
    """HBM bytes per launch from the committed PMC pass (profiles/r01_stft_pmc.json), or None."""
    p = os.path.join(ROOT, "profiles", "stft_pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU per step")
    ap.add_argument("--gather", action="store_true", help="time the RCCL all_gather of the spectra even with one rank")
    ap.add_argument("--no-gather", action="store_true", help="skip the (separately reported) output all_gather at N > 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of one hipGraph replay of the K steps")
    ap.add_argument("--spinup-ms", type=float, default=60.0,
                    help="untimed load before the W warmup steps so that the GPU is at its sustained clocks whatever W is "
                         "(0 disables; reported in config.spinup_ms)")
    args = ap.parse_args()

    import torch
    import jeicyboodsp_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or "RANK" in os.environ:            # launched by torch.distributed.run: one rank per GPU, RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = torch.device("cuda", local_rank)

    eng = jeicyboodsp_amd.Engine(local_rank)
    B = args.frames
    pcm = torch.from_numpy(synth_pcm(rank, B, world)[: 512 * (B + 1)].copy()).to(dev)
    spec = torch.empty((B, N_FFT), dtype=torch.complex64, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    eng.stft(pcm, B, N_FFT, HOP, out=spec)            # builds the handle's constant tables (first call only)
    barrier()
    # Timed region: EXACTLY K launches, captured once into a hipGraph (jdsp_stft_i16_dev only
    # enqueues: no allocation, no sync) and replayed, so the host's per-launch overhead is not in
    # the way; bracketed by HIP events on the launch stream and by barriers for the wall clock.
    # The capture comes BEFORE the warmup, so that the W warmup launches run right up to the barrier that
    # opens the timed region.  The defaults (W = 500, K = 1000: 48 ms + 96 ms of GPU time) are sized for the
    # GPU's clock management: after an idle period it takes tens of milliseconds of continuous load to
    # reach its sustained clocks -- on one box W/K = 20/200 gave 96.4 us per launch, 200/500 90.3,
    # 500/1000 89.7, 2000/4000 88.3 (profiles/r01_bench_warmup_sweep.txt).
    graph = None
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for _ in range(args.steps):
                        eng.stft(pcm, B, N_FFT, HOP, out=spec)
            torch.cuda.current_stream().wait_stream(side)
        except Exception as exc:                      # capture unsupported: fall back to eager launches
            print("bench: graph capture failed (%s), timing eager launches" % exc, file=sys.stderr)
            graph = None
    # Clock spin-up, independent of W: a GPU that has idled needs tens of milliseconds of continuous load to reach
    # its sustained clocks (profiles/r01_bench_warmup_sweep.txt); a caller that passes a small W would otherwise
    # time the ramp.  Untimed, like the warmup, and reported in config.spinup_ms.
    if args.spinup_ms > 0:
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
            for _ in range(32):
                eng.stft(pcm, B, N_FFT, HOP, out=spec)
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        eng.stft(pcm, B, N_FFT, HOP, out=spec)
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    if graph is not None:
        graph.replay()
    else:
        for _ in range(args.steps):
            eng.stft(pcm, B, N_FFT, HOP, out=spec)
    e1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = e0.elapsed_time(e1) / max(args.steps, 1)      # average launch duration (incl. launch gaps)

    # whole-job timing first: MAX over ranks of the wall clock and of the per-launch duration
    t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kern_ms = float(t[0]), float(t[1])

    # The only collective of the path: gathering the spectra (RCCL all_gather over xGMI).  Timed
    # AFTER and OUTSIDE the timed region, reported separately, never part of `value` (SURVEY §8e asks for
    # both figures at N > 1: at 8 GPUs the gather costs far more than the transform).  On by default when
    # there is more than one rank.  Every collective of this phase is launched asynchronously and polled
    # against a deadline, so that a stuck gather costs the gather figure, never the bench line.
    gather_ms = None
    stuck = False

    def guarded(launch, deadline_s):
        work = launch()
        t_end = time.perf_counter() + deadline_s
        while not work.is_completed():
            if time.perf_counter() > t_end:
                return False
            time.sleep(0.0005)
        torch.cuda.synchronize()
        return True

    if dist is not None and (args.gather or (world > 1 and not args.no_gather)):
        try:
            flat = torch.view_as_real(spec)
            gathered = torch.empty((world,) + tuple(flat.shape), dtype=flat.dtype, device=dev)
            launch = lambda: dist.all_gather_into_tensor(gathered, flat, async_op=True)   # noqa: E731
            stuck = not guarded(launch, 120.0)
            if not stuck:
                t1 = time.perf_counter()
                for _ in range(3):
                    stuck = stuck or not guarded(launch, 60.0)
                local_ms = (time.perf_counter() - t1) / 3 * 1e3
            if not stuck:
                g = torch.tensor([local_ms], dtype=torch.float64, device=dev)
                stuck = not guarded(lambda: dist.all_reduce(g, op=dist.ReduceOp.MAX, async_op=True), 60.0)
                if not stuck:
                    gather_ms = float(g[0])
            if stuck:
                print("bench: output gather did not complete in time, reported as null", file=sys.stderr)
        except Exception as exc:
            print("bench: output gather skipped (%s)" % exc, file=sys.stderr)
            gather_ms = None

    if rank == 0:
        frames_total = float(B) * world * args.steps
        value = frames_total / elapsed
        ach = BYTES_PER_FRAME * B / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "STFT frames/s (1024-pt, 50% OLA)",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "STFT analysis n_fft=1024 hop=512 Hamming, batch=%d frames/GPU, int16 PCM in HBM -> complex64 full spectrum in HBM" % B,
                       "frames_per_gpu": B, "parallelism": "frame-sharded x%d, no collective" % world,
                       "launch": "hipGraph replay of the K steps" if graph is not None else "eager",
                       "spinup_ms": args.spinup_ms},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": read_traffic(),
                         "kernel": "stft1024_hop512_kernel<2>", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": BYTES_PER_FRAME * B},
        }
        if gather_ms is not None:
            line["gather"] = {"ms": gather_ms, "bytes_per_rank": B * N_FFT * 8,
                              "frames_per_s_including_gather": float(B) * world / (elapsed / args.steps + gather_ms * 1e-3)}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)

    eng.close()
    if stuck:                                         # a collective is still pending: do not wait on it
        sys.stdout.flush()
        os._exit(0)
    if dist is not None:
        if not guarded(lambda: dist.barrier(async_op=True), 60.0):
            sys.stdout.flush()
            os._exit(0)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
