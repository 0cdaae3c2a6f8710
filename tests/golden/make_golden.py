#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  Run in the authoring container only
(needs /root/reference and oracle/_ref built by `make -C oracle`):

    python tests/golden/make_golden.py

What is stored is DATA: seeded inputs and the outputs the compiled reference
produced for them -- never reference source text.

  fftalg_512.npz    FFTAlgorithm_ver2.cpp built as it stands (BLOCK_LEN 512):
                    Bitrev table, FFTProcess fwd/inv, DFTProcess, IDFTProcess,
                    IFFTProcess on seeded frames, and main()'s WAV round trip.
  fftalg_1024.npz   same translation unit with -DBLOCK_LEN=1024 (the only edit:
                    the #define at :16), Bitrev table + FFTProcess fwd/inv.
  stft_1024.npz     the headline path pinned to the reference's OWN transform: seeded int16 frames
  stft_512.npz      (hop n/2) x the applications' Hamming window (SpectralSubtraction_final.cpp:226,
                    PI 3.141592) -> FFTProcess of the compiled FFTAlgorithm_ver2.cpp (BLOCK_LEN = n),
                    i.e. SS:218-230 with the reference's in-tree FFT in FFTW's place (north_star:
                    "FFTAlgorithm_ver2 feeding ...").  n = 1024 (BASELINE metric) and 512 (config 3 as worded).
  rir_taps.npz      the 69 non-zero taps (index, value) of FilterCoefficient.h's
                    rgdFirLPF_coefficients[7169] -- the filter DATA the native
                    fast-convolution configuration runs with.
"""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402

REF = "/root/reference"


class quiet_stdout:
    """FFTProcess printf()s an op count per call (:148): send C stdout to /dev/null."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        self.null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self.null, 1)

    def __exit__(self, *a):
        C.CDLL(None).fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.null)
        os.close(self.saved)


def seeded_pcm(seed, n, sigma=3000.0):
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(rng.normal(0.0, sigma, n)), -32768, 32767).astype(np.int16)


def run_ref_main(lib_path, pcm):
    """The reference main(): 44-byte header skipped, per-block FFT->IFFT->(short)."""
    with tempfile.TemporaryDirectory() as d:
        src, dst = os.path.join(d, "in.wav"), os.path.join(d, "out.raw")
        with open(src, "wb") as f:
            f.write(bytes(44))
            f.write(pcm.tobytes())
        code = ("import ctypes as C,sys\n"
                "L=C.CDLL(%r)\n"
                "argv=(C.c_char_p*3)(b'ref',%r.encode(),%r.encode())\n"
                "L.ref_main(3,argv)\n" % (lib_path, src, dst))
        subprocess.run([sys.executable, "-c", code], stdin=subprocess.DEVNULL,
                       stdout=subprocess.DEVNULL, check=True)
        return np.fromfile(dst, np.int16)


def fftalg(block_len, n_frames, with_slow):
    ref = oracle_lib.load_ref(block_len)
    assert ref is not None and ref.block_len == block_len
    n = block_len
    pcm = seeded_pcm(1000 + block_len, n_frames * n)
    frames = pcm.reshape(n_frames, n)
    out = {"pcm": pcm}
    with quiet_stdout():
        out["bitrev"] = ref.bitrev_table(n)
        fwd = np.stack([ref.fft_process(fr.astype(np.complex128), True) for fr in frames])
        inv = np.stack([ref.fft_process(s, False) for s in fwd])
        out["fwd"], out["inv"] = fwd, inv
        # a complex (not purely real) input too
        rng = np.random.default_rng(7)
        z = rng.normal(size=n) + 1j * rng.normal(size=n)
        out["cin"] = z
        out["cfwd"] = ref.fft_process(z, True)
        if with_slow:
            out["dft"] = np.stack([ref.dft_process(fr) for fr in frames[:2]])
            out["idft"] = np.stack([ref.idft_process(s) for s in fwd[:2]])
            out["ifft_n2"] = np.stack([ref.ifft_process(s) for s in fwd[:2]])
    if with_slow:
        out["main_out"] = run_ref_main(oracle_lib.ref_path(block_len), pcm)
    np.savez_compressed(os.path.join(HERE, "fftalg_%d.npz" % block_len), **out)
    print("fftalg_%d.npz" % block_len, {k: v.shape for k, v in out.items()})


def stft(n, n_frames):
    """SS:218-230 with FFTAlgorithm_ver2's FFTProcess as the transform."""
    import math
    ref = oracle_lib.load_ref(n)
    assert ref is not None and ref.block_len == n
    hop = n // 2
    pcm = seeded_pcm(2000 + n, hop * (n_frames - 1) + n)
    PI = 3.141592                                                      # SS:52
    w = np.array([0.54 - 0.46 * math.cos(2 * PI * i / (n - 1)) for i in range(n)])   # SS:226, libm cos
    spec = np.zeros((n_frames, n), np.complex128)
    with quiet_stdout():
        for f in range(n_frames):
            fr = pcm[hop * f:hop * f + n].astype(np.float64) * w      # fcInputBefFFT[i][0] *= (...)
            spec[f] = ref.fft_process(fr.astype(np.complex128), True)
    np.savez_compressed(os.path.join(HERE, "stft_%d.npz" % n), pcm=pcm, hop=np.int32(hop), spec=spec)
    print("stft_%d.npz" % n, pcm.shape, spec.shape)


def rir_taps():
    txt = open(os.path.join(REF, "FilterCoefficient.h"), "rb").read().decode("latin-1")
    n = int(re.search(r"#define\s+FILTER_LENGTH\s+(\d+)", txt).group(1))
    body = txt[txt.index("{") + 1:txt.rindex("}")]
    vals = np.array([float(t) for t in body.replace("\r", "").replace("\n", "").split(",") if t.strip()])
    assert vals.size == n, (vals.size, n)
    idx = np.nonzero(vals)[0].astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "rir_taps.npz"), n_taps=np.int32(n), index=idx, value=vals[idx])
    print("rir_taps.npz", n, idx.size, idx[0], idx[-1], vals.sum())


if __name__ == "__main__":
    fftalg(512, 4, True)
    fftalg(1024, 2, False)
    stft(1024, 6)
    stft(512, 6)
    rir_taps()
