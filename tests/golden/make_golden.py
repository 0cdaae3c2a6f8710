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
  mfcc_tail.npz     MFCCFeatureExtraction_auto_version1.cpp:13-42,116-192 compiled as it lies (oracle/Makefile,
                    libref_mfcc_tail.so): MelFilterBankInit's three tables; seeded |X| rows (one with an empty
                    channel, one all zero) -> MelFilterBank's 38 ln-sums -> DCT's 12 cepstra (from zero and
                    accumulating into a pre-filled vector) -> Liftering.
  vad.npz           VoiceActivityDetection of SpectralSubtraction_final.cpp:121-156, WienerFilter_final.cpp:261-296
                    and BeamForming_MVDR_ver1.cpp:207-242 (libref_vad_{ss,wf,bf}.so) on seeded blocks at levels
                    around both thresholds (E = 700, Z = 200) plus constant / alternating / full-scale blocks:
                    return value, and the energy and zero-crossing count the function printf()s.  The value its
                    one-past-the-end read (SS:139) finds is painted by the caller: 0 (what the oracle defines),
                    +1 and -1, each recorded.
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


def mfcc_tail():
    ref = oracle_lib.load_ref_mfcc_tail()
    assert ref is not None
    mel, fi, fb = ref.tables()
    rng = np.random.default_rng(31)
    nb = ref.n_bins
    rows = []
    for sigma in (3000.0, 300.0, 20.0):                       # |X| of windowed int16 frames at three levels
        x = np.clip(np.rint(rng.normal(0, sigma, 2 * nb)), -32768, 32767)
        w = 0.54 - 0.46 * np.cos(2 * 3.141592 * np.arange(2 * nb) / (2 * nb - 1))
        rows.append(np.abs(np.fft.fft(x * w))[:nb])
    rows.append(np.abs(rng.normal(0, 1.0, nb)) * np.exp(-np.arange(nb) / 60.0) * 1e5)   # steep spectral tilt
    r = rows[0].copy()
    r[(fi == 7) | (fi == 8)] = 0.0                            # channel 7 gets nothing: ln 0 = -inf (MFCC:171)
    rows.append(r)
    rows.append(np.zeros(nb))                                 # digital silence: every channel -inf
    rows.append(np.full(nb, 1.0))
    rows.append(np.abs(rng.normal(0, 1e-3, nb)))              # tiny magnitudes: negative logarithms
    mag = np.stack(rows)
    with np.errstate(all="ignore"):
        mel_out = ref.mel_filterbank(mag)
        cep = ref.dct(mel_out)
        acc_in = rng.normal(0, 5.0, cep.shape)
        cep_acc = ref.dct(mel_out, accumulate_into=acc_in)
        lift = ref.liftering(cep)
    np.savez_compressed(os.path.join(HERE, "mfcc_tail.npz"),
                        consts=np.array([ref.n_cep, ref.n_bins, ref.n_chan, ref.lifter, ref.block_len], np.int32),
                        half_rate=np.float64(ref.half_rate), mel_freqs=mel, fi_bins=fi, filter_bank=fb,
                        mag=mag, mel=mel_out, cep=cep, cep_acc_in=acc_in, cep_acc=cep_acc, liftered=lift)
    print("mfcc_tail.npz", mag.shape, mel_out.shape, cep.shape, "non-finite mel:", int((~np.isfinite(mel_out)).sum()))


def vad_blocks():
    """Seeded 512-sample blocks around the two thresholds + special shapes."""
    rng = np.random.default_rng(41)
    out = []
    for sigma in np.linspace(45.0, 75.0, 61):                 # E crosses 700 near sigma = 59 (white: Z ~ 256 >= 200)
        out.append(rng.normal(0, sigma, 512))
    for a in np.linspace(0.0, 0.75, 76):                      # AR(1) low-pass at a quiet level: Z sweeps ~256 -> ~120
        e = rng.normal(0, 40.0, 512 + 64)
        y = np.zeros_like(e)
        for i in range(1, e.size):
            y[i] = a * y[i - 1] + e[i]
        y = y[64:]
        out.append(y * (40.0 / max(y.std(), 1e-9)))
    for sigma in (3.0, 10.0, 30.0, 300.0, 3000.0, 12000.0):
        out.append(rng.normal(0, sigma, 512))
    blocks = [np.clip(np.rint(b), -32768, 32767).astype(np.int16) for b in out]
    alt = np.where(np.arange(512) % 2 == 0, 1, -1)
    for v in (np.zeros(512), np.full(512, 32767), np.full(512, -32768), alt, alt * 40, alt * 300, alt * 32767,
              np.where(np.arange(512) % 4 < 2, 50, -50), np.full(512, 26), np.full(512, -27)):
        blocks.append(np.asarray(v).astype(np.int16))
    return np.stack(blocks)


def vad():
    blocks = vad_blocks()
    out = {"blocks": blocks}
    for which in ("ss", "wf", "bf"):
        ref = oracle_lib.load_ref_vad(which)
        assert ref is not None and ref.block_len == 512
        out[which + "_consts"] = np.array([ref.thr_energy, ref.thr_zcr, ref.keep_len, ref.block_len, ref.n_fft, ref.pi])
        for fill, tag in ((0, ""), (1, "_fill_pos"), (-1, "_fill_neg")):
            flags, energy, zcr = ref.run(blocks, fill)
            out[which + "_flags" + tag] = flags.astype(np.uint8)
            out[which + "_zcr" + tag] = zcr
            if fill == 0:
                out[which + "_energy_printed"] = energy       # "%f": 6 decimals of sum / 1024
                esum = np.rint(energy * ref.n_fft).astype(np.int64)
                assert np.abs(esum / ref.n_fft - energy).max() < 1e-6
                out[which + "_energy_sum"] = esum             # the integer sum of squares, exact
            else:
                assert np.array_equal(energy, out[which + "_energy_printed"])
        f, z = out[which + "_flags"], out[which + "_zcr"]
        print("vad", which, "voice", int(f.sum()), "of", f.size, "| Z in [195,205]:", int(((z >= 195) & (z <= 205)).sum()),
              "| E in [650,750]:", int(((out[which + "_energy_printed"] > 650) & (out[which + "_energy_printed"] < 750)).sum()),
              "| flag flips under fill:", int((out[which + "_flags_fill_pos"] != f).sum()),
              int((out[which + "_flags_fill_neg"] != f).sum()))
    np.savez_compressed(os.path.join(HERE, "vad.npz"), **out)


if __name__ == "__main__":
    fftalg(512, 4, True)
    fftalg(1024, 2, False)
    stft(1024, 6)
    stft(512, 6)
    rir_taps()
    mfcc_tail()
    vad()
