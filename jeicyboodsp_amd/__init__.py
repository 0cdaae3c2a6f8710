"""jeicyboodsp_amd -- MI355X (gfx950) engine for JeicybooDSP's FFT-based spectral
path.  The product is libjdsp.so (hand-written HIP kernels behind the C ABI in
include/jdsp.h); this package is the thin Python mirror used by tests and
bench.py.  PyTorch is used only to own device memory and streams."""
from ._lib import JdspError, LIB_PATH  # noqa: F401
from .engine import GMM_PARAM, HMM_PARAM, Denoiser, Engine, FastConv, Gmm, Hmm, Mfcc, Mvdr, MvdrMulti  # noqa: F401

__all__ = ["Engine", "Denoiser", "FastConv", "Mfcc", "Mvdr", "MvdrMulti", "Gmm", "Hmm", "GMM_PARAM", "HMM_PARAM",
           "JdspError", "LIB_PATH"]
