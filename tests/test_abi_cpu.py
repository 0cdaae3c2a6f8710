"""CPU tests of the drop-in boundary: libjdsp.so loads, exports every symbol that
include/jdsp.h declares, reports errors through return codes, and refuses to run
without a GPU (no fallback).  No compute is attempted here."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "jdsp.h")
LIB = os.path.join(ROOT, "jeicyboodsp_amd", "libjdsp.so")


def declared_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(jdsp_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build_hip()
    return C.CDLL(LIB)


def test_header_is_plain_c():
    src = "#include \"%s\"\nint main(void){return JDSP_ABI_VERSION==0;}\n" % HEADER
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-x", "c", "-fsyntax-only", "-"],
                   input=src.encode(), check=True)


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (jdsp_[a-z0-9_]+)", out))
    assert set(names) <= exported
    # nothing that looks like a CPU fallback or the oracle is linked in
    assert not re.search(r"orc_|oracle", out)
    deps = subprocess.run(["ldd", LIB], capture_output=True, text=True).stdout
    assert "libamdhip64" in deps and "oracle" not in deps and "torch" not in deps


def test_abi_version_and_null_handling(lib):
    lib.jdsp_abi_version.restype = C.c_int
    assert lib.jdsp_abi_version() == 2            # include/jdsp.h: JDSP_ABI_VERSION
    lib.jdsp_last_error.restype = C.c_char_p
    for fn in ("jdsp_destroy", "jdsp_denoise_destroy", "jdsp_mfcc_destroy", "jdsp_fastconv_destroy"):
        f = getattr(lib, fn)
        f.argtypes = [C.c_void_p]
        assert f(None) == 0
    lib.jdsp_synchronize.argtypes = [C.c_void_p]
    assert lib.jdsp_synchronize(None) == -1            # JDSP_EINVAL
    lib.jdsp_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    assert lib.jdsp_create(0, None) == -1


def test_no_gpu_means_loud_failure_not_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    lib.jdsp_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    rc = lib.jdsp_create(0, C.byref(h))
    assert rc == -4 and not h.value                    # JDSP_ENODEV
    lib.jdsp_last_error.restype = C.c_char_p
    assert b"no CPU fallback" in lib.jdsp_last_error(None)
    import jeicyboodsp_amd
    with pytest.raises(jeicyboodsp_amd.JdspError):
        jeicyboodsp_amd.Engine(0)


def test_product_never_touches_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may load anything under oracle/."""
    pkg = os.path.join(ROOT, "jeicyboodsp_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in txt.lower(), os.path.join(dp, f)


def test_per_file_build_flags_name_existing_sources():
    """csrc/build_flags.txt (per-file hipcc flags read by build() and tools/build_variant.sh): every entry names a source
    that exists, and the flags are what the notes beside them say -- a renamed file must not silently lose its flags."""
    import __graft_entry__ as g
    flags = g.file_flags()
    csrc = os.path.join(ROOT, "jeicyboodsp_amd", "csrc")
    assert flags, "build_flags.txt parsed to nothing"
    for name, fl in flags.items():
        assert os.path.exists(os.path.join(csrc, name)), name
        assert fl and all(f.startswith("-") for f in fl), (name, fl)
    assert "-fno-slp-vectorize" in flags.get("mvdr_kernels.hip", [])
    assert "-DJDSP_XCHG_UNPAIRED=0" in flags.get("stft_kernels.hip", [])
