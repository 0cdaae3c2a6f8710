#!/usr/bin/env python3
"""bench.py -- headline benchmark: STFT frames/s (1024-pt, 50 % overlap, Hamming)
at batch = 65,536 frames per GPU (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path (jdsp_stft_i16_dev: int16 framing + window
+ forward transform, full 1024-bin complex64 spectrum) over one batch of
synthetic PCM that is already resident in HBM.  Successive steps read DIFFERENT
PCM buffers (--pcm-buffers, default 6 x 64 MiB = 384 MiB > the 256 MiB Infinity
Cache), so the input of every timed launch comes from HBM, not from a cache that
the previous launch warmed; the same-buffer figure is reported beside it
(`roofline.warm_input`).

N > 1: one rank per GPU over RCCL.  Either an external launcher started the ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: RANK
is set) or this script starts them itself: `python bench.py --gpus N` spawns
`torch.distributed.run` as a CHILD process before anything touches the GPU,
relays rank 0's JSON line and exits with the children's status.  Frames are
independent, so every rank transforms its own shard of N*65,536 frames (weak
scaling, no data-path collective); the output all_gather is timed separately and
never part of `value`.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_FFT, HOP, FRAMES_PER_GPU = 1024, 512, 65536
BYTES_PER_FRAME = 512 * 2 + 1024 * 8          # 9,216 B algorithmic
BYTES_PER_FRAME_F64 = 512 * 2 + 1024 * 16     # the same analysis in the reference's FP64: complex128 spectrum
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: 8.0 TB/s spec
TRAFFIC_FILE = os.path.join("profiles", "stft_pmc_traffic.json")
EXIT_STUCK_COLLECTIVE = 3


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU per step")
    ap.add_argument("--pcm-buffers", type=int, default=6,
                    help="distinct PCM input buffers the steps rotate over (1 = every step re-reads the same buffer)")
    ap.add_argument("--gather", action="store_true", help="time the all_gather of the spectra even with one rank")
    ap.add_argument("--no-gather", action="store_true", help="skip the (separately reported) output all_gather at N > 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp64-leg", action="store_true",
                    help="skip roofline.fp64 (the same K steps through jdsp_stft_i16_f64_dev, timed after the reported leg)")
    ap.add_argument("--workload", choices=["stft", "mfcc10k"], default="stft",
                    help="stft (default): the headline metric.  mfcc10k: BASELINE config 4 -- the 10,000-utterance MFCC batch "
                         "(400/160 framing, 512-FFT, 40 mel) sharded by utterance over the ranks (strong scaling); a secondary "
                         "line, never the headline")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of one hipGraph replay of the K steps")
    ap.add_argument("--spinup-ms", type=float, default=60.0,
                    help="untimed load before the W warmup steps so that the GPU is at its sustained clocks whatever W is "
                         "(0 disables; reported in config.spinup_ms)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend: nccl (= RCCL over xGMI, the product path) or gloo (host-staged "
                         "gather; for rehearsing N ranks on a box with fewer GPUs -- ranks then share GPUs)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="no GPU work at all: ranks rendezvous, barrier, reduce the timings and gather a dummy buffer, "
                         "so the N > 1 launch path can be tested on a CPU-only host (the line says so; value is null)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------ N > 1: start the ranks ourselves
def rank_command(args, argv, port):
    """The child command for `python bench.py --gpus N` (N > 1, no RANK in the environment)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(args, argv):
    """Parent of an N-rank run.  Never imports torch / touches HIP itself (a process that has
    initialised the GPU must not spawn the job that uses it by exec; this one spawns a child and
    waits).  stdout/stderr are inherited, so rank 0's JSON line is this process's JSON line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(rank_command(args, argv, port), env=env, stdin=subprocess.DEVNULL).returncode


# ------------------------------------------------------------------ inputs, baselines
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


def host_cpu_share():
    """(cores visible to this process, cgroup CPU quota in cores or None)."""
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()
        if q != "max":
            quota = float(q) / float(period)
    except Exception:
        pass
    return visible, quota


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
    # every host core this job may use: frames are independent, static partition over threads (ctypes
    # drops the GIL).  Threads = the cores visible to the process (nproc), cut to the cgroup's CPU quota
    # when there is one (more runnable threads than the quota only adds contention) -- both are reported.
    import threading
    visible, quota = host_cpu_share()
    cores = max(1, min(visible, int(quota) if quota and quota >= 1 else visible))
    small = 512                                    # shorter calls so that 256 threads all finish inside the window
    pcm_s = pcm[: 512 * (small + 1)]
    counts = [0] * cores
    stop = time.perf_counter() + min(seconds, 8.0)

    def work(i):
        while time.perf_counter() < stop:
            orc.stft(pcm_s, small)
            counts[i] += small

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    [t.start() for t in th]
    [t.join() for t in th]
    allc = sum(counts) / (time.perf_counter() - t0)
    return {"value": one, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames (chunks of %d from the same synthetic stream), FP64 oracle, single thread" % (done, chunk),
            "all_cores": {"value": allc, "unit": "frames/s", "threads": cores, "nproc": visible,
                          "cgroup_cpu_quota": quota,
                          "note": "same oracle, one thread per usable host core (nproc cut to the cgroup CPU quota if any)"}}


def read_traffic():
    """HBM bytes per launch from the committed PMC pass (a constant of that collection, NOT measured by
    this run: rocprofv3 --pmc needs its own passes), or None."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            return json.load(f).get("hbm_bytes_per_launch")
    except Exception:
        return None


# ------------------------------------------------------------------ one rank
def init_dist(args):
    """(dist module or None, rank, world, local_rank)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and "RANK" not in os.environ:
        return None, 0, 1, 0
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if args.backend == "nccl" and not args.launcher_selftest:
        n_dev = max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank % n_dev)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank % n_dev))
    else:
        dist.init_process_group("gloo")
    return dist, rank, world, local_rank


def guarded(work_launch, deadline_s, sync=None):
    """Launches an async collective and polls it against a deadline: False = still pending."""
    work = work_launch()
    t_end = time.perf_counter() + deadline_s
    while not work.is_completed():
        if time.perf_counter() > t_end:
            return False
        time.sleep(0.0005)
    if sync:
        sync()
    return True


def selftest_rank(args):
    """--launcher-selftest: everything of an N-rank run except the GPU (rendezvous, barriers, MAX
    reduction of the timings, the output gather on a dummy buffer, the JSON line, the shutdown)."""
    import torch
    dist, rank, world, _ = init_dist(args)
    assert world == args.gpus, "world size %d != --gpus %d" % (world, args.gpus)
    B = args.frames
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64)
    gather_ms = None
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        flat = torch.full((B, N_FFT, 2), float(rank), dtype=torch.float32)
        out = torch.empty((world * B,) + tuple(flat.shape[1:]), dtype=flat.dtype)     # concatenated along dim 0
        t1 = time.perf_counter()
        dist.all_gather_into_tensor(out, flat)
        gather_ms = (time.perf_counter() - t1) * 1e3
        assert all(float(out[r * B, 0, 0]) == float(r) for r in range(world))
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test (no GPU work, not a measurement)", "value": None,
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": float(t[0]) / max(args.steps, 1) * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none",
                          "config": {"workload": "launcher self-test", "backend": "gloo", "frames_per_gpu": B},
                          "gather": {"ms": gather_ms, "bytes_per_rank": B * N_FFT * 8}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def run_rank(args):
    import torch
    import jeicyboodsp_amd

    dist, rank, world, local_rank = init_dist(args)
    assert world == args.gpus, "world size %d != --gpus %d (start the ranks with torch.distributed.run, or let " \
                               "`python bench.py --gpus N` start them)" % (world, args.gpus)
    n_dev = torch.cuda.device_count()
    assert n_dev >= 1, "bench.py needs a GPU (no CPU fallback)"
    if args.backend == "nccl":
        assert world <= n_dev, "nccl: %d ranks need %d GPUs, %d visible (rehearse with --backend gloo)" % (world, world, n_dev)
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    eng = jeicyboodsp_amd.Engine(dev_index)
    B = args.frames
    P = max(1, args.pcm_buffers)
    base = torch.from_numpy(synth_pcm(rank, B, world)[: 512 * (B + 1)].copy()).to(dev)
    # P distinct buffers (distinct addresses are what defeats the cache; the content is the same noise rotated
    # by a whole number of frames, so every buffer is the SURVEY §8d stream: no all-zero frames)
    pcms = [base] + [torch.roll(base, 512 * 97 * i).contiguous() for i in range(1, P)]
    spec = torch.empty((B, N_FFT), dtype=torch.complex64, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(i, bufs):
        eng.stft(bufs[i % len(bufs)], B, N_FFT, HOP, out=spec)

    step(0, pcms)                                     # builds the handle's constant tables (first call only)
    barrier()

    # Timed region: EXACTLY K launches, captured once into a hipGraph (jdsp_stft_i16_dev only
    # enqueues: no allocation, no sync) and replayed, so the host's per-launch overhead is not in
    # the way; bracketed by HIP events on the launch stream and by barriers for the wall clock.
    # The capture comes BEFORE the warmup, so that the W warmup launches run right up to the barrier that
    # opens the timed region.  The defaults (W = 500, K = 1000: 48 ms + 96 ms of GPU time) are sized for the
    # GPU's clock management: after an idle period it takes tens of milliseconds of continuous load to
    # reach its sustained clocks (profiles/r01_bench_warmup_sweep.txt).
    def capture(bufs, step=step):
        if args.no_graph:
            return None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    for i in range(args.steps):
                        step(i, bufs)
            torch.cuda.current_stream().wait_stream(side)
            return g
        except Exception as exc:                      # capture unsupported: fall back to eager launches
            print("bench: graph capture failed (%s), timing eager launches" % exc, file=sys.stderr)
            return None

    def timed(bufs, graph, step=step):
        for i in range(args.warmup):
            step(i, bufs)
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        if graph is not None:
            graph.replay()
        else:
            for i in range(args.steps):
                step(i, bufs)
        e1.record()
        barrier()
        elapsed = time.perf_counter() - t0
        return elapsed, e0.elapsed_time(e1) / max(args.steps, 1)   # wall clock; average launch duration (incl. gaps)

    # Three legs, each its own captured graph of the K steps:
    #   reported : library defaults ("stft.read_pass" auto: a read-only launch streams each slab's PCM into the
    #              Infinity Cache, then the transform runs), input ROTATED over P buffers -- PCM comes from HBM
    #   warm     : every step re-reads ONE buffer, read pass off -- round 1's figure (input served by the cache)
    #   cold, no read pass : what the transform alone does with its input in HBM
    eng.set_option("stft.read_pass", -1)
    graph = capture(pcms)
    eng.set_option("stft.read_pass", 0)
    graph_warm = capture(pcms[:1]) if P > 1 else None
    graph_cold0 = capture(pcms) if P > 1 else None
    eng.set_option("stft.read_pass", -1)
    # Clock spin-up, independent of W: a GPU that has idled needs tens of milliseconds of continuous load to reach
    # its sustained clocks; a caller that passes a small W would otherwise time the ramp.  Untimed, like the warmup,
    # and reported in config.spinup_ms.
    if args.spinup_ms > 0:
        t_spin = time.perf_counter()
        i = 0
        while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
            for _ in range(32):
                step(i, pcms)
                i += 1
            torch.cuda.synchronize()
    warm = cold0 = None
    if P > 1:                                         # secondary figures first
        eng.set_option("stft.read_pass", 0)           # (the eager warmup launches follow the leg's setting too)
        warm = timed(pcms[:1], graph_warm)
        cold0 = timed(pcms, graph_cold0)
        eng.set_option("stft.read_pass", -1)
    elapsed, kern_ms = timed(pcms, graph)             # the reported figure: library defaults, input rotated over P buffers

    # The reference computes in FP64 (SS:205-206 fftw_complex): the same K steps through jdsp_stft_i16_f64_dev
    # (complex128 out, 17,408 algorithmic bytes per frame), AFTER the reported leg so that `value` is untouched.
    f64 = None
    if not args.no_fp64_leg:
        spec = None                                   # 512 MiB back before the 1 GiB of complex128
        spec64 = torch.empty((B, N_FFT), dtype=torch.complex128, device=dev)
        def step64(i, bufs):
            eng.stft_f64(bufs[i % len(bufs)], B, HOP, out=spec64)

        step64(0, pcms)
        barrier()
        f64 = timed(pcms, capture(pcms, step64), step64)
        del spec64
        spec = torch.empty((B, N_FFT), dtype=torch.complex64, device=dev)
        eng.stft(pcms[0], B, N_FFT, HOP, out=spec)    # the gather below moves real spectra

    # whole-job timing: MAX over ranks of the wall clock and of the per-launch duration
    t = torch.tensor([elapsed, kern_ms] + (list(warm) + list(cold0) if warm else [0.0] * 4) + (list(f64) if f64 else [0.0] * 2),
                     dtype=torch.float64, device=dev)
    if dist is not None:
        tt = t.cpu() if args.backend == "gloo" else t
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = tt
    elapsed, kern_ms, warm_elapsed, warm_kern_ms, cold0_elapsed, cold0_kern_ms, f64_elapsed, f64_kern_ms = (float(x) for x in t)

    # The only collective of the path: gathering the spectra (RCCL all_gather over xGMI).  Timed
    # AFTER and OUTSIDE the timed region, reported separately, never part of `value` (SURVEY §8e asks for
    # both figures at N > 1: at 8 GPUs the gather costs far more than the transform).  On by default when
    # there is more than one rank.  Every collective of this phase is launched asynchronously and polled
    # against a deadline, so that a stuck gather costs the gather figure, never the bench line -- and the
    # process then exits non-zero.
    gather_ms = None
    stuck = False
    gather_note = None
    if dist is not None and (args.gather or (world > 1 and not args.no_gather)):
        try:
            flat = torch.view_as_real(spec)
            if args.backend == "gloo":                # gloo has no device all_gather: stage through the host
                src = flat.cpu()
                gathered = torch.empty((world * B,) + tuple(src.shape[1:]), dtype=src.dtype)
                sync = None
                gather_note = "gloo, host-staged (rehearsal transport, not xGMI)"
            else:
                src = flat
                gathered = torch.empty((world * B,) + tuple(flat.shape[1:]), dtype=flat.dtype, device=dev)
                sync = torch.cuda.synchronize
                gather_note = "RCCL all_gather_into_tensor"
            launch = lambda: dist.all_gather_into_tensor(gathered, src, async_op=True)   # noqa: E731
            stuck = not guarded(launch, 120.0, sync)
            if not stuck:
                t1 = time.perf_counter()
                for _ in range(3):
                    stuck = stuck or not guarded(launch, 60.0, sync)
                local_ms = (time.perf_counter() - t1) / 3 * 1e3
            if not stuck:
                g = torch.tensor([local_ms], dtype=torch.float64, device=None if args.backend == "gloo" else dev)
                stuck = not guarded(lambda: dist.all_reduce(g, op=dist.ReduceOp.MAX, async_op=True), 60.0, sync)
                if not stuck:
                    gather_ms = float(g[0])
                    # the gathered buffer really holds every rank's spectra: rank r's first bin of frame 0
                    mine = gathered[rank * B, 0].to("cpu")
                    assert torch.equal(mine, flat[0, 0].to("cpu")), "gathered shard differs from the local spectra"
            if stuck:
                print("bench: output gather did not complete in time", file=sys.stderr)
        except AssertionError:
            raise
        except Exception as exc:
            print("bench: output gather skipped (%s)" % exc, file=sys.stderr)
            gather_ms = None

    if rank == 0:
        frames_total = float(B) * world * args.steps
        value = frames_total / elapsed
        alg = BYTES_PER_FRAME * B
        ach = alg / (kern_ms * 1e-3) / 1e9
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
                       "spinup_ms": args.spinup_ms,
                       "pcm_buffers": P, "pcm_bytes_total": P * int(base.numel()) * 2,
                       "input": ("step i reads PCM buffer i %% %d (%d MiB in all, larger than the 256 MiB Infinity Cache): cold input"
                                 % (P, P * int(base.numel()) * 2 >> 20)) if P > 1 else "every step re-reads the same PCM buffer",
                       "backend": args.backend if dist is not None else None},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": read_traffic(),
                         "traffic_source": TRAFFIC_FILE + " (committed rocprofv3 --pmc passes of this command; a constant of "
                                           "that collection, not measured by the run that printed this line)",
                         "kernel": "stft1024_hop512_kernel<1> behind pcm_touch_kernel (the read pass): one step = both "
                                   "launches, kernel_ms = HIP-event time per step = the sum of their two durations + the gap",
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg},
        }
        if warm is not None:
            w_ach = alg / (warm_kern_ms * 1e-3) / 1e9
            line["roofline"]["warm_input"] = {"kernel_ms": warm_kern_ms, "achieved": w_ach, "frac": w_ach / HBM_PEAK_GBS,
                                              "value": frames_total / warm_elapsed,
                                              "note": "same K steps, every step re-reading ONE 64 MiB PCM buffer (input served by the "
                                                      "Infinity Cache), read pass off: stft1024_hop512_kernel<2> alone -- round 1's figure"}
            c_ach = alg / (cold0_kern_ms * 1e-3) / 1e9
            line["roofline"]["cold_input_without_read_pass"] = {
                "kernel_ms": cold0_kern_ms, "achieved": c_ach, "frac": c_ach / HBM_PEAK_GBS, "value": frames_total / cold0_elapsed,
                "note": "input rotated like the reported figure, read pass off: what stft1024_hop512_kernel<2> alone does with PCM in HBM"}
        if f64 is not None:
            alg64 = BYTES_PER_FRAME_F64 * B
            a64 = alg64 / (f64_kern_ms * 1e-3) / 1e9
            line["roofline"]["fp64"] = {
                "kernel_ms": f64_kern_ms, "achieved": a64, "frac": a64 / HBM_PEAK_GBS, "unit": "GB/s",
                "value": frames_total / f64_elapsed, "algorithmic_bytes_per_launch": alg64, "dtype": "f64",
                "kernel": "stft1024_f64_v2_kernel behind pcm_touch_kernel (jdsp_stft_i16_f64_dev): the reference's own "
                          "arithmetic (FP64 window, transform and split; complex128 full spectrum, 17,408 B per frame)",
                "note": "same K steps, same rotated input, timed after the reported leg; never part of `value`"}
        if gather_ms is not None:
            line["gather"] = {"ms": gather_ms, "bytes_per_rank": B * N_FFT * 8, "transport": gather_note,
                              "frames_per_s_including_gather": float(B) * world / (elapsed / args.steps + gather_ms * 1e-3)}
        elif stuck:
            line["gather"] = {"stuck": True}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)

    eng.close()
    if stuck:                                         # a collective is still pending: do not wait on it, and say so
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(EXIT_STUCK_COLLECTIVE)
    if dist is not None:
        if not guarded(lambda: dist.barrier(async_op=True), 60.0):
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(EXIT_STUCK_COLLECTIVE)
        dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------ BASELINE config 4 (secondary line)
MFCC_KW = dict(win_len=400, hop=160, n_fft=512, n_chan=40, n_cep=13, half_rate=8000.0)
MFCC_BYTES_PER_FRAME = 160 * 2 + 13 * 8       # hop new samples in, n_cep doubles out


def mfcc10k_batch():
    """The 10,000-utterance ragged batch (1-6 s at 16 kHz, packed back to back): sample offsets and a 4 M-sample
    noise table the PCM is tiled from (seeded: every rank builds the same batch and keeps its own shard)."""
    import numpy as np
    rng = np.random.default_rng(2024)
    lens = rng.integers(16000, 6 * 16000 + 1, 10000)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    table = np.clip(np.rint(rng.normal(0, 3000, 1 << 22)), -32768, 32767).astype(np.int16)
    return offs, table


def run_rank_mfcc(args):
    """`--workload mfcc10k`: whole utterances per rank (sharding.mfcc_utterance_shard), no halo, no collective on the
    data path; STRONG scaling (the batch is fixed, N ranks split it).  value = frames of the whole batch per second."""
    import numpy as np
    import torch
    import jeicyboodsp_amd
    from jeicyboodsp_amd import sharding

    dist, rank, world, local_rank = init_dist(args)
    assert world == args.gpus, "world size %d != --gpus %d" % (world, args.gpus)
    n_dev = torch.cuda.device_count()
    assert n_dev >= 1, "bench.py needs a GPU (no CPU fallback)"
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    eng = jeicyboodsp_amd.Engine(dev_index)
    offs, table = mfcc10k_batch()
    u0, nu, lo, hi, f0, f1, local = sharding.mfcc_utterance_shard(offs, MFCC_KW["win_len"], MFCC_KW["hop"], rank, world)
    total_frames = sum(sharding.mfcc_utterance_shard(offs, MFCC_KW["win_len"], MFCC_KW["hop"], r, world)[5] -
                       sharding.mfcc_utterance_shard(offs, MFCC_KW["win_len"], MFCC_KW["hop"], r, world)[4] for r in range(world))
    T = table.size
    pcm = np.tile(table, (hi - lo) // T + 2)[lo % T: lo % T + (hi - lo)]
    d_pcm = torch.from_numpy(np.ascontiguousarray(pcm)).to(dev)
    d_st = torch.from_numpy(local).to(dev)
    nf = int(local.size)
    feats = torch.empty((max(nf, 1), MFCC_KW["n_cep"]), dtype=torch.float64, device=dev)
    m = eng.mfcc(**MFCC_KW)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        if nf:
            m.frames(d_pcm, nf, frame_start=d_st, out=feats)

    step()
    barrier()
    for _ in range(args.warmup):
        step()
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = e0.elapsed_time(e1) / max(args.steps, 1)
    t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=dev)
    if dist is not None:
        tt = t.cpu() if args.backend == "gloo" else t
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = tt
    elapsed, kern_ms = float(t[0]), float(t[1])
    if rank == 0:
        alg = MFCC_BYTES_PER_FRAME * nf
        ach = alg / (kern_ms * 1e-3) / 1e9
        line = {"metric": "MFCC frames/s (25 ms / 10 ms framing, 512-FFT, 40 mel + DCT; 10,000-utterance batch)",
                "value": float(total_frames) * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "MFCC 400/160 framing, 512-FFT, 40 mel, 13 cepstra (FP64 out); 10,000 ragged utterances "
                                       "(1-6 s at 16 kHz, %d frames), int16 PCM in HBM -> double[13] vectors in HBM" % total_frames,
                           "parallelism": "utterance-sharded x%d (balanced by frame count), no collective" % world,
                           "frames_rank0": nf, "utterances_rank0": nu, "launch": "eager",
                           "backend": args.backend if dist is not None else None},
                "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                             "traffic": None, "kernel": "mfcc512_pair_kernel (+ mfcc_kernel for the listed unequal pairs)",
                             "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg,
                             "note": "424 B per frame against ~28 kflop: this chain is VALU/LDS-issue-bound, the HBM fraction "
                                     "is reported for completeness (DESIGN.md 3.6); rank 0's launch"}}
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            orc = oracle_lib.load_oracle()
            ocfg = orc.mfcc_cfg(n_bins=256, **MFCC_KW)
            done, t1 = 0, time.perf_counter()
            u = 0
            while time.perf_counter() - t1 < 10.0:
                n = int((offs[u + 1] - offs[u] - 400) // 160 + 1)
                orc.mfcc_frames(ocfg, pcm[offs[u] - lo: offs[u + 1] - lo], n)
                done += n
                u += 1
            line["cpu_baseline"] = {"value": done / (time.perf_counter() - t1), "unit": "frames/s", "cores": 1, "kind": "port",
                                    "sample": "the batch's first %d utterances (%d frames), FP64 oracle, single thread" % (u, done)}
        print(json.dumps(line), flush=True)
    m.close()
    eng.close()
    if dist is not None:
        if not guarded(lambda: dist.barrier(async_op=True), 60.0):
            os._exit(EXIT_STUCK_COLLECTIVE)
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        # BEFORE importing torch or touching the GPU: this process only starts the ranks and waits for them
        return launch_ranks(args, argv)
    if args.launcher_selftest:
        return selftest_rank(args)
    if args.workload == "mfcc10k":
        return run_rank_mfcc(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
