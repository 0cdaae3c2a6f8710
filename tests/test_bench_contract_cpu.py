"""CPU: bench.py, __graft_entry__.py and the tools compile, and bench.py's JSON line carries every key the
driver's contract names (checked in the source: running it needs a GPU)."""
import os
import py_compile
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_entry_points_compile():
    files = ["bench.py", "__graft_entry__.py"] + [os.path.join("tools", f) for f in os.listdir(os.path.join(ROOT, "tools"))
                                                  if f.endswith(".py")]
    for f in files:
        py_compile.compile(os.path.join(ROOT, f), doraise=True)


def test_bench_line_has_the_contract_keys():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]:
        assert re.search(r'"%s"' % key, src), key
    for key in ["bound", "achieved", "peak", "frac", "traffic"]:                       # roofline object
        assert re.search(r'"%s"' % key, src), key
    for key in ["cores", "kind", "sample"]:                                             # cpu_baseline object
        assert re.search(r'"%s"' % key, src), key
    assert '"--gpus"' in src and '"--steps"' in src and '"--warmup"' in src
    assert "STFT frames/s (1024-pt, 50% OLA)" in src                                   # BASELINE.json's metric


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_n_builds_a_torch_distributed_run_child_command():
    """`python bench.py --gpus N` (no RANK in the environment) must start the ranks itself: one
    torch.distributed.run child, N local ranks, rendezvous on 127.0.0.1, the same flags passed through."""
    import sys
    bench = _load_bench()
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = bench.rank_command(bench.parse_args(argv), argv, 29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv
    # the parent must decide BEFORE importing torch / touching HIP
    src = open(os.path.join(ROOT, "bench.py")).read()
    main_src = src[src.index("def main("):]
    assert main_src.index("launch_ranks(args, argv)") < main_src.index("run_rank(args)")
    assert "import torch" not in src[:src.index("def synth_pcm")]


def test_self_launch_two_ranks_gloo_end_to_end():
    """The N > 1 path for real, on the CPU: `python bench.py --gpus 2 --launcher-selftest` spawns two ranks
    (gloo), which rendezvous, barrier, MAX-reduce their timings, all_gather a 256-frame dummy spectrum buffer
    and print exactly one JSON line; the parent relays it and exits 0."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest",
                        "--frames", "256", "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] is None and "self-test" in line["metric"]
    assert line["gather"]["bytes_per_rank"] == 256 * 1024 * 8 and line["gather"]["ms"] > 0


def test_stuck_collective_is_not_reported_as_success():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os._exit(0)" not in src and "os._exit(EXIT_STUCK_COLLECTIVE)" in src
    assert re.search(r"EXIT_STUCK_COLLECTIVE\s*=\s*[1-9]", src)
    assert '"stuck": True' in src
