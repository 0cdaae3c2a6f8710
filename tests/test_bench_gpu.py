"""GPU: bench.py end to end on the box's one MI355X -- the JSON contract of the N = 1 line, RCCL really
executing (init_process_group("nccl") + all_gather_into_tensor at world size 1), and the self-launched
N = 2 path with real kernels (both ranks share the one GPU, so the rehearsal transport is gloo: RCCL refuses
two ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _one_json_line(r):
    assert r.returncode == 0, (r.stdout.decode()[-1500:], r.stderr.decode()[-3000:])
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_default_line_small_batch_carries_roofline_and_cold_input_figures():
    r = subprocess.run([sys.executable, BENCH, "--frames", "4096", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                       env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    line = _one_json_line(r)
    assert line["metric"] == "STFT frames/s (1024-pt, 50% OLA)" and line["n_gpus"] == 1 and line["value"] > 0
    assert line["config"]["pcm_buffers"] == 6
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0 < roof["frac"] < 1.2
    assert "traffic_source" in roof and roof["warm_input"]["kernel_ms"] > 0
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * roof["achieved"]


def test_rccl_runs_at_world_size_one_with_the_output_gather():
    """`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 --gather`: the nccl (= RCCL) process group is
    initialised on the MI355X and all_gather_into_tensor of the spectra executes; the line reports it."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29541", BENCH, "--gpus", "1", "--gather", "--frames", "4096", "--steps", "4",
           "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    line = _one_json_line(r)
    assert line["n_gpus"] == 1 and line["config"]["backend"] == "nccl"
    g = line["gather"]
    assert g.get("stuck") is not True and g["ms"] > 0 and g["bytes_per_rank"] == 4096 * 1024 * 8
    assert "RCCL" in g["transport"]


def test_gpus_two_starts_its_own_ranks_and_runs_the_kernels():
    """`python bench.py --gpus 2` with nothing in the environment: the parent spawns torch.distributed.run, two
    ranks run the real STFT kernel (sharing cuda:0 on this one-GPU box, hence --backend gloo), the barrier +
    MAX-over-ranks timing and the output gather run, rank 0 prints one line, the parent exits 0."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--frames", "256", "--steps", "2",
                        "--warmup", "1", "--spinup-ms", "5"],
                       env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    line = _one_json_line(r)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["frames_per_gpu"] == 256 and "cpu_baseline" not in line
    assert line["gather"]["ms"] > 0 and "gloo" in line["gather"]["transport"]


def test_default_line_carries_the_fp64_leg():
    """roofline.fp64: the same K steps through jdsp_stft_i16_f64_dev (the reference's own precision, complex128 out,
    17,408 algorithmic bytes per frame), timed after the reported leg."""
    r = subprocess.run([sys.executable, BENCH, "--frames", "4096", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"],
                       env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    f = _one_json_line(r)["roofline"]["fp64"]
    assert f["dtype"] == "f64" and f["algorithmic_bytes_per_launch"] == 4096 * (512 * 2 + 1024 * 16)
    assert f["kernel_ms"] > 0 and abs(f["achieved"] - f["algorithmic_bytes_per_launch"] / (f["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * f["achieved"]
    r = subprocess.run([sys.executable, BENCH, "--frames", "4096", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-fp64-leg"],
                       env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert "fp64" not in _one_json_line(r)["roofline"]


@pytest.mark.parametrize("gpus", [1, 2])
def test_mfcc10k_workload_is_rank_aware(gpus):
    """`--workload mfcc10k` = BASELINE config 4: the 10,000-utterance MFCC batch split by whole utterances over the
    ranks (strong scaling).  One rank, and two self-launched ranks sharing this box's one GPU (gloo): the same total
    frame count, rank 0 holding all of it or about half."""
    cmd = [sys.executable, BENCH, "--workload", "mfcc10k", "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    if gpus > 1:
        cmd += ["--backend", "gloo"]
    r = subprocess.run(cmd, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    line = _one_json_line(r)
    assert line["n_gpus"] == gpus and line["scaling"] == "strong" and line["unit"] == "frames/s" and line["value"] > 0
    assert "10,000" in line["config"]["workload"] and "utterance-sharded x%d" % gpus in line["config"]["parallelism"]
    total = 3_000_000
    if gpus == 1:
        assert line["config"]["utterances_rank0"] == 10000 and 3_000_000 < line["config"]["frames_rank0"] < 4_000_000
    else:
        assert 4000 < line["config"]["utterances_rank0"] < 6000 and total / 2 * 0.9 < line["config"]["frames_rank0"] < 2_100_000
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["algorithmic_bytes_per_launch"] == 424 * line["config"]["frames_rank0"]
