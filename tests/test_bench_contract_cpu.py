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
