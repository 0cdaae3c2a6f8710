"""ctypes bindings for the CPU oracle (oracle/libjdsp_oracle.so) and, when it
has been built in the authoring container, the compiled reference translation
unit oracle/_ref/libref_fftalg_*.so.

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

_c_short_p = C.POINTER(C.c_short)
_c_double_p = C.POINTER(C.c_double)
_c_int_p = C.POINTER(C.c_int)


class MfccCfg(C.Structure):
    _fields_ = [("win_len", C.c_int), ("hop", C.c_int), ("n_fft", C.c_int), ("n_bins", C.c_int),
                ("n_chan", C.c_int), ("n_cep", C.c_int), ("lifter", C.c_int),
                ("half_rate", C.c_double), ("preemph", C.c_double)]


# the reference's parameter records (GMMAlgorithm_Test_Auto_ver2.cpp:29-34, Viterbi_version1.cpp:30-40)
GMM_PARAM = np.dtype([("alpa", "<f8", (4,)), ("mean", "<f8", (4, 12)), ("covariance", "<f8", (4, 12, 12)),
                      ("eigenVector", "<f8", (4, 12, 4))])
HMM_PARAM = np.dtype([("gMMParam", GMM_PARAM, (6,)), ("transProb", "<f8", (6, 6))])


def _p(a, t):
    return a.ctypes.data_as(t)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libjdsp_oracle.so"])


class Oracle:
    SPECSUB, WIENER = 0, 1

    def __init__(self, path):
        L = self.lib = C.CDLL(path)
        L.orc_bitrev_table.argtypes = [C.c_int, C.c_int, _c_short_p]
        L.orc_fft_process.argtypes = [_c_double_p, _c_double_p, C.c_int, C.c_int, C.c_int]
        L.orc_dft_process.argtypes = [_c_short_p, _c_double_p, C.c_int]
        L.orc_idft_process.argtypes = [_c_double_p, _c_double_p, C.c_int]
        L.orc_ifft_process.argtypes = [_c_double_p, _c_double_p, C.c_int]
        L.orc_fft_roundtrip_i16.argtypes = [_c_short_p, C.c_int, C.c_int, _c_short_p]
        L.orc_dft_c2c.argtypes = [_c_double_p, _c_double_p, C.c_int, C.c_int]
        L.orc_hamming.argtypes = [C.c_int, _c_double_p]
        L.orc_stft.argtypes = [_c_short_p, C.c_long, C.c_int, C.c_int, _c_double_p]
        L.orc_vad_block.argtypes = [_c_short_p, C.c_int, _c_double_p, _c_int_p]
        L.orc_vad_block.restype = C.c_int
        L.orc_mvdr_vad_block.argtypes = [_c_short_p, C.c_int, _c_double_p, _c_int_p]
        L.orc_mvdr_vad_block.restype = C.c_int
        L.orc_denoise_stream.argtypes = [C.c_int, _c_short_p, C.c_long, _c_short_p, _c_double_p]
        L.orc_denoise_stream.restype = C.c_long
        L.orc_denoise_stream2.argtypes = [C.c_int, C.c_int, _c_short_p, C.c_long, _c_short_p, _c_double_p]
        L.orc_denoise_stream2.restype = C.c_long
        L.orc_denoise_create2.argtypes = [C.c_int, C.c_int]
        L.orc_denoise_create2.restype = C.c_void_p
        L.orc_denoise_create.argtypes = [C.c_int]
        L.orc_denoise_create.restype = C.c_void_p
        L.orc_denoise_destroy.argtypes = [C.c_void_p]
        L.orc_denoise_block.argtypes = [C.c_void_p, _c_short_p, _c_short_p, _c_double_p]
        L.orc_denoise_block.restype = C.c_int
        L.orc_denoise_noise.argtypes = [C.c_void_p]
        L.orc_denoise_noise.restype = _c_double_p
        L.orc_denoise_last_voice.argtypes = [C.c_void_p]
        L.orc_denoise_last_voice.restype = C.c_int
        L.orc_fastconv_stream.argtypes = [_c_short_p, C.c_long, _c_double_p, C.c_int, C.c_int,
                                          _c_short_p, _c_double_p]
        L.orc_fastconv_stream.restype = C.c_long
        L.orc_pitch_stream.argtypes = [_c_short_p, C.c_long, _c_int_p, _c_double_p, _c_double_p]
        L.orc_mvdr_stream.argtypes = [_c_short_p, _c_short_p, C.c_long, C.c_double, _c_short_p, _c_double_p,
                                      _c_double_p, _c_double_p]
        L.orc_mvdr_stream.restype = C.c_long
        L.orc_mvdr_estimate.argtypes = [_c_short_p, _c_short_p, _c_double_p]
        L.orc_mvdr_create.restype = C.c_void_p
        L.orc_mvdr_destroy.argtypes = [C.c_void_p]
        L.orc_mvdr_process_block.argtypes = [C.c_void_p, _c_short_p, _c_short_p, C.c_double, _c_double_p, _c_short_p,
                                             _c_double_p]
        L.orc_mvdr_process_block.restype = C.c_int
        L.orc_mvdrn_stream.argtypes = [_c_short_p, C.c_long, C.c_int, C.c_long, _c_double_p, C.c_double, _c_short_p,
                                       _c_double_p]
        L.orc_mvdrn_stream.restype = C.c_long
        L.orc_mvdrn_stream2.argtypes = [_c_short_p, C.c_long, C.c_int, C.c_long, _c_double_p, C.c_double, C.c_int,
                                        _c_short_p, _c_double_p]
        L.orc_mvdrn_stream2.restype = C.c_long
        L.orc_gmm_probability.argtypes = [_c_double_p, _c_double_p, _c_double_p, _c_double_p]
        L.orc_gmm_probability.restype = C.c_double
        L.orc_gmm_recognition.argtypes = [_c_double_p, C.c_long, C.c_void_p]
        L.orc_gmm_recognition.restype = C.c_double
        L.orc_gmm_classify.argtypes = [_c_double_p, C.c_long, C.c_void_p, C.c_int, _c_double_p]
        L.orc_gmm_classify.restype = C.c_int
        L.orc_hmm_viterbi.argtypes = [_c_double_p, C.c_long, C.c_void_p, _c_int_p, _c_double_p]
        L.orc_hmm_viterbi.restype = C.c_double
        L.orc_mfcc_native_cfg.argtypes = [C.POINTER(MfccCfg)]
        L.orc_mel_init.argtypes = [C.POINTER(MfccCfg), _c_double_p, _c_int_p, _c_double_p]
        L.orc_mel_filterbank.argtypes = [C.POINTER(MfccCfg), _c_int_p, _c_double_p, _c_double_p, _c_double_p]
        L.orc_dct.argtypes = [C.POINTER(MfccCfg), _c_double_p, _c_double_p]
        L.orc_liftering.argtypes = [C.POINTER(MfccCfg), _c_double_p]
        L.orc_mfcc_frame.argtypes = [C.POINTER(MfccCfg), _c_int_p, _c_double_p, _c_short_p, _c_double_p]
        L.orc_mfcc_stream.argtypes = [C.POINTER(MfccCfg), _c_short_p, C.c_long, _c_double_p]
        L.orc_mfcc_stream.restype = C.c_long

    # --- GMM scoring / HMM recursion (records: numpy arrays of GMM_PARAM / HMM_PARAM) ------------
    def gmm_probability(self, x, mean, cov, eig):
        x = np.ascontiguousarray(x, np.float64)
        mean = np.ascontiguousarray(mean, np.float64)
        cov = np.ascontiguousarray(cov, np.float64)
        eig = np.ascontiguousarray(eig, np.float64)
        return self.lib.orc_gmm_probability(_p(x, _c_double_p), _p(mean, _c_double_p), _p(cov, _c_double_p),
                                            _p(eig, _c_double_p))

    def gmm_classify(self, feats, classes):
        """(scores [n_classes], arg) for one utterance [n, 12]."""
        feats = np.ascontiguousarray(feats, np.float64)
        classes = np.ascontiguousarray(classes, GMM_PARAM).reshape(-1)
        scores = np.zeros(len(classes), np.float64)
        arg = self.lib.orc_gmm_classify(_p(feats, _c_double_p), feats.shape[0], classes.ctypes.data_as(C.c_void_p),
                                        len(classes), _p(scores, _c_double_p))
        return scores, arg

    def hmm_viterbi(self, feats, model):
        """(returned value, path [n], trellis [6, n]) for one utterance and one HMM_PARAM record."""
        feats = np.ascontiguousarray(feats, np.float64)
        model = np.ascontiguousarray(model, HMM_PARAM).reshape(-1)
        n = feats.shape[0]
        path = np.zeros(max(n, 1), np.int32)
        trellis = np.zeros((6, max(n, 1)), np.float64)
        ret = self.lib.orc_hmm_viterbi(_p(feats, _c_double_p), n, model.ctypes.data_as(C.c_void_p), _p(path, _c_int_p),
                                       _p(trellis, _c_double_p))
        return ret, path[:n], trellis[:, :n]

    # --- FFTAlgorithm_ver2 --------------------------------------------------
    def bitrev_table(self, n_fft, block_len=None):
        t = np.zeros(n_fft, np.int16)
        self.lib.orc_bitrev_table(n_fft, block_len or n_fft, _p(t, _c_short_p))
        return t

    def fft_process(self, x, forward=True, block_len=None):
        x = np.ascontiguousarray(x, np.complex128)
        out = np.zeros_like(x)
        n = x.shape[-1]
        for xi, oi in zip(x.reshape(-1, n), out.reshape(-1, n)):
            self.lib.orc_fft_process(_p(xi, _c_double_p), _p(oi, _c_double_p), n, int(forward), block_len or n)
        return out

    def dft_process(self, s):
        s = np.ascontiguousarray(s, np.int16)
        out = np.zeros(s.shape[-1], np.complex128)
        self.lib.orc_dft_process(_p(s, _c_short_p), _p(out, _c_double_p), s.shape[-1])
        return out

    def idft_process(self, x):
        x = np.ascontiguousarray(x, np.complex128)
        out = np.zeros_like(x)
        self.lib.orc_idft_process(_p(x, _c_double_p), _p(out, _c_double_p), x.shape[-1])
        return out

    def ifft_process(self, x):
        x = np.ascontiguousarray(x, np.complex128)
        out = np.zeros_like(x)
        self.lib.orc_ifft_process(_p(x, _c_double_p), _p(out, _c_double_p), x.shape[-1])
        return out

    def fft_roundtrip_i16(self, pcm, n_fft):
        pcm = np.ascontiguousarray(pcm, np.int16)
        nb = pcm.size // n_fft
        out = np.zeros(nb * n_fft, np.int16)
        self.lib.orc_fft_roundtrip_i16(_p(pcm, _c_short_p), nb, n_fft, _p(out, _c_short_p))
        return out

    def dft_c2c(self, x, sign=-1):
        x = np.ascontiguousarray(x, np.complex128)
        out = np.zeros_like(x)
        n = x.shape[-1]
        for xi, oi in zip(x.reshape(-1, n), out.reshape(-1, n)):
            self.lib.orc_dft_c2c(_p(xi, _c_double_p), _p(oi, _c_double_p), n, sign)
        return out

    # --- applications -------------------------------------------------------
    def hamming(self, n):
        w = np.zeros(n, np.float64)
        self.lib.orc_hamming(n, _p(w, _c_double_p))
        return w

    def stft(self, pcm, n_frames, n=1024, hop=512):
        pcm = np.ascontiguousarray(pcm, np.int16)
        assert pcm.size >= (n_frames - 1) * hop + n
        out = np.zeros((n_frames, n), np.complex128)
        self.lib.orc_stft(_p(pcm, _c_short_p), n_frames, n, hop, _p(out, _c_double_p))
        return out

    def vad_block(self, block):
        block = np.ascontiguousarray(block, np.int16)
        e = C.c_double()
        z = C.c_int()
        v = self.lib.orc_vad_block(_p(block, _c_short_p), block.size, C.byref(e), C.byref(z))
        return bool(v), e.value, z.value

    def mvdr_vad_block(self, block):
        """BF:207-242 on one block of n_fft/2 samples: (voice, energy, zcr)."""
        block = np.ascontiguousarray(block, np.int16)
        e = C.c_double()
        z = C.c_int()
        v = self.lib.orc_mvdr_vad_block(_p(block, _c_short_p), 2 * block.size, C.byref(e), C.byref(z))
        return bool(v), e.value, z.value

    def denoise_stream(self, mode, pcm, block=512):
        pcm = np.ascontiguousarray(pcm, np.int16)
        nb = pcm.size // block
        out = np.zeros(max(nb, 1) * block, np.int16)
        pre = np.zeros(max(nb, 1) * block, np.float64)
        n = self.lib.orc_denoise_stream2(mode, block, _p(pcm, _c_short_p), nb, _p(out, _c_short_p), _p(pre, _c_double_p))
        return out[:n * block].copy(), pre[:n * block].copy()

    def denoise_trace(self, mode, pcm, block=512):
        """Block-by-block run that also returns the VAD flags and every latched noise estimate."""
        pcm = np.ascontiguousarray(pcm, np.int16)
        nb = pcm.size // block
        h = self.lib.orc_denoise_create2(mode, block)
        out, pre, flags, noises, ver = [], [], [], [np.zeros(2 * block)], []
        ob = np.zeros(block, np.int16)
        pb = np.zeros(block, np.float64)
        for b in range(nb):
            blk = pcm[b * block:(b + 1) * block]
            ok = self.lib.orc_denoise_block(h, _p(blk, _c_short_p), _p(ob, _c_short_p), _p(pb, _c_double_p))
            cur = np.ctypeslib.as_array(self.lib.orc_denoise_noise(h), shape=(2 * block,)).copy()
            if not np.array_equal(cur, noises[-1]):
                noises.append(cur)
            ver.append(len(noises) - 1)
            flags.append(self.lib.orc_denoise_last_voice(h))
            if ok:
                out.append(ob.copy())
                pre.append(pb.copy())
        self.lib.orc_denoise_destroy(h)
        cat = lambda l, dt: np.concatenate(l) if l else np.zeros(0, dt)
        return cat(out, np.int16), cat(pre, np.float64), np.array(flags, np.int32), np.stack(noises), np.array(ver, np.int32)

    def fastconv_stream(self, pcm, taps, n_fft):
        pcm = np.ascontiguousarray(pcm, np.int16)
        taps = np.ascontiguousarray(taps, np.float64)
        block = n_fft - taps.size + 1
        nb = pcm.size // block
        out = np.zeros(max(nb, 1) * block, np.int16)
        pre = np.zeros(max(nb, 1) * block, np.float64)
        n = self.lib.orc_fastconv_stream(_p(pcm, _c_short_p), nb, _p(taps, _c_double_p), taps.size, n_fft,
                                         _p(out, _c_short_p), _p(pre, _c_double_p))
        return out[:n * block].copy(), pre[:n * block].copy()

    def pitch_stream(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.int16)
        nb = pcm.size // 512
        arg = np.zeros(nb, np.int32)
        rmax = np.zeros(nb, np.float64)
        ac = np.zeros((nb, 512), np.float64)
        self.lib.orc_pitch_stream(_p(pcm, _c_short_p), nb, _p(arg, _c_int_p), _p(rmax, _c_double_p), _p(ac, _c_double_p))
        return arg, rmax, ac

    def mvdr_stream(self, left, right, d_time=0.0):
        left = np.ascontiguousarray(left, np.int16)
        right = np.ascontiguousarray(right, np.int16)
        nb = left.size // 512
        out = np.zeros(max(nb, 1) * 512, np.int16)
        pre = np.zeros(max(nb, 1) * 512, np.float64)
        corr = np.zeros(4, np.float64)
        trace = np.zeros((max(nb, 1), 4), np.float64)
        n = self.lib.orc_mvdr_stream(_p(left, _c_short_p), _p(right, _c_short_p), nb, d_time, _p(out, _c_short_p),
                                     _p(pre, _c_double_p), _p(corr, _c_double_p), _p(trace, _c_double_p))
        return out[:n * 512].copy(), pre[:n * 512].copy(), corr, trace[:nb]

    def mvdr_estimate(self, temp_l, temp_r, corr):
        """EstimateSpatialCorrMtx on one 1024-sample frame per channel: returns corr + the frame's contribution."""
        tl = np.ascontiguousarray(temp_l, np.int16)
        tr = np.ascontiguousarray(temp_r, np.int16)
        assert tl.size == 1024 and tr.size == 1024
        c = np.ascontiguousarray(corr, np.float64).reshape(4).copy()
        self.lib.orc_mvdr_estimate(_p(tl, _c_short_p), _p(tr, _c_short_p), _p(c, _c_double_p))
        return c

    def mvdr_apply(self, left, right, corr, d_time=0.0):
        """ProcessMVDR block by block with ONE caller matrix: (int16 out, pre-cast) of the blocks it returns true for."""
        left = np.ascontiguousarray(left, np.int16)
        right = np.ascontiguousarray(right, np.int16)
        c = np.ascontiguousarray(corr, np.float64).reshape(4)
        st = self.lib.orc_mvdr_create()
        outs, pres = [], []
        ob, pb = np.zeros(512, np.int16), np.zeros(512, np.float64)
        for b in range(left.size // 512):
            ok = self.lib.orc_mvdr_process_block(st, _p(left[b * 512:(b + 1) * 512].copy(), _c_short_p),
                                                 _p(right[b * 512:(b + 1) * 512].copy(), _c_short_p), d_time,
                                                 _p(c, _c_double_p), _p(ob, _c_short_p), _p(pb, _c_double_p))
            if ok:
                outs.append(ob.copy())
                pres.append(pb.copy())
        self.lib.orc_mvdr_destroy(st)
        cat = lambda l, dt: np.concatenate(l) if l else np.zeros(0, dt)
        return cat(outs, np.int16), cat(pres, np.float64)

    def mvdrn_stream(self, pcm, delays=None, loading=0.0, n_fft=1024):
        """pcm: int16 [n_mics, n_samples] (planar); blocks of n_fft / 2 samples."""
        pcm = np.ascontiguousarray(pcm, np.int16)
        m, n = pcm.shape
        B = n_fft // 2
        nb = n // B
        out = np.zeros(max(nb, 1) * B, np.int16)
        pre = np.zeros(max(nb, 1) * B, np.float64)
        d = np.ascontiguousarray(delays if delays is not None else np.zeros(m), np.float64)
        k = self.lib.orc_mvdrn_stream2(_p(pcm, _c_short_p), n, m, nb, _p(d, _c_double_p), float(loading), n_fft,
                                       _p(out, _c_short_p), _p(pre, _c_double_p))
        return out[:k * B].copy(), pre[:k * B].copy()

    def mfcc_native_cfg(self):
        c = MfccCfg()
        self.lib.orc_mfcc_native_cfg(C.byref(c))
        return c

    def mfcc_cfg(self, **kw):
        c = self.mfcc_native_cfg()
        for k, v in kw.items():
            setattr(c, k, v)
        return c

    def mel_init(self, cfg):
        mel = np.zeros(cfg.n_chan + 1, np.float64)
        fi = np.zeros(cfg.n_bins, np.int32)
        fb = np.zeros(cfg.n_bins, np.float64)
        self.lib.orc_mel_init(C.byref(cfg), _p(mel, _c_double_p), _p(fi, _c_int_p), _p(fb, _c_double_p))
        return mel, fi, fb

    def mel_filterbank(self, cfg, mag):
        """MelFilterBank (MFCC:154-174) per row of |X| [rows, n_bins] -> [rows, n_chan]."""
        mag = np.ascontiguousarray(np.atleast_2d(mag), np.float64)
        _, fi, fb = self.mel_init(cfg)
        out = np.zeros((mag.shape[0], cfg.n_chan), np.float64)
        for r in range(mag.shape[0]):
            self.lib.orc_mel_filterbank(C.byref(cfg), _p(fi, _c_int_p), _p(fb, _c_double_p), _p(mag[r], _c_double_p),
                                        _p(out[r], _c_double_p))
        return out

    def dct(self, cfg, mel, accumulate_into=None):
        mel = np.ascontiguousarray(np.atleast_2d(mel), np.float64)
        out = np.zeros((mel.shape[0], cfg.n_cep), np.float64) if accumulate_into is None else \
            np.ascontiguousarray(np.atleast_2d(accumulate_into), np.float64).copy()
        for r in range(mel.shape[0]):
            self.lib.orc_dct(C.byref(cfg), _p(mel[r], _c_double_p), _p(out[r], _c_double_p))
        return out

    def liftering(self, cfg, cep):
        cep = np.ascontiguousarray(np.atleast_2d(cep), np.float64).copy()
        for r in range(cep.shape[0]):
            self.lib.orc_liftering(C.byref(cfg), _p(cep[r], _c_double_p))
        return cep

    def mfcc_frames(self, cfg, pcm, n_frames, first_frame=0):
        """frame j = pcm[hop*(first_frame+j) : +win_len]"""
        pcm = np.ascontiguousarray(pcm, np.int16)
        _, fi, fb = self.mel_init(cfg)
        out = np.zeros((n_frames, cfg.n_cep), np.float64)
        for j in range(n_frames):
            fr = np.ascontiguousarray(pcm[cfg.hop * (first_frame + j):cfg.hop * (first_frame + j) + cfg.win_len])
            assert fr.size == cfg.win_len
            self.lib.orc_mfcc_frame(C.byref(cfg), _p(fi, _c_int_p), _p(fb, _c_double_p), _p(fr, _c_short_p),
                                    _p(out[j], _c_double_p))
        return out

    def mfcc_stream(self, cfg, pcm):
        pcm = np.ascontiguousarray(pcm, np.int16)
        nb = pcm.size // cfg.win_len
        out = np.zeros((max(2 * nb - 1, 1), cfg.n_cep), np.float64)
        n = self.lib.orc_mfcc_stream(C.byref(cfg), _p(pcm, _c_short_p), nb, _p(out, _c_double_p))
        return out[:max(n, 0)].copy()


def load_oracle():
    path = os.path.join(ORACLE_DIR, "libjdsp_oracle.so")
    if not os.path.exists(path):
        build_oracle()
    return Oracle(path)


class RefFftAlg:
    """FFTAlgorithm_ver2.cpp compiled from the reference checkout (oracle/_ref)."""

    def __init__(self, path):
        L = self.lib = C.CDLL(path)
        L.ref_Bitrev_table.argtypes = [C.c_int, _c_short_p]
        L.ref_FFTProcess.argtypes = [_c_double_p, _c_double_p, C.c_int, C.c_int]
        L.ref_DFTProcess.argtypes = [_c_short_p, _c_double_p, C.c_int]
        L.ref_IDFTProcess.argtypes = [_c_double_p, _c_double_p, C.c_int]
        L.ref_IFFTProcess.argtypes = [_c_double_p, _c_double_p, C.c_int]
        L.ref_block_len.restype = C.c_int
        self.block_len = L.ref_block_len()

    def bitrev_table(self, n_fft):
        t = np.zeros(n_fft, np.int16)
        self.lib.ref_Bitrev_table(n_fft, _p(t, _c_short_p))
        return t

    def fft_process(self, x, forward=True):
        x = np.ascontiguousarray(x, np.complex128)
        out = np.zeros_like(x)
        self.lib.ref_FFTProcess(_p(x, _c_double_p), _p(out, _c_double_p), x.size, int(forward))
        return out

    def dft_process(self, s):
        s = np.ascontiguousarray(s, np.int16)
        out = np.zeros(s.size, np.complex128)
        self.lib.ref_DFTProcess(_p(s, _c_short_p), _p(out, _c_double_p), s.size)
        return out

    def idft_process(self, x):
        x = np.ascontiguousarray(x, np.complex128)
        out = np.zeros_like(x)
        self.lib.ref_IDFTProcess(_p(x, _c_double_p), _p(out, _c_double_p), x.size)
        return out

    def ifft_process(self, x):
        x = np.ascontiguousarray(x, np.complex128)
        out = np.zeros_like(x)
        self.lib.ref_IFFTProcess(_p(x, _c_double_p), _p(out, _c_double_p), x.size)
        return out


class capture_c_stdout:
    """Sends the C library's stdout (fd 1) to a temporary file for the duration; .text afterwards."""

    def __enter__(self):
        import sys
        import tempfile
        sys.stdout.flush()
        C.CDLL(None).fflush(None)
        self.tmp = tempfile.TemporaryFile()
        self.saved = os.dup(1)
        os.dup2(self.tmp.fileno(), 1)
        return self

    def __exit__(self, *a):
        C.CDLL(None).fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)
        self.tmp.seek(0)
        self.text = self.tmp.read().decode()
        self.tmp.close()


class RefMfccTail:
    """MelFilterBankInit / MelFilterBank / DCT / Liftering and the three global tables of
    MFCCFeatureExtraction_auto_version1.cpp (:13-42, :116-192), compiled from the reference checkout."""

    def __init__(self, path):
        L = self.lib = C.CDLL(path)
        L.ref_mfcc_consts.argtypes = [_c_int_p]
        L.ref_mfcc_half_rate.restype = C.c_double
        L.ref_mfcc_tables.argtypes = [_c_double_p, _c_int_p, _c_double_p]
        L.ref_MelFilterBank.argtypes = [_c_double_p, _c_double_p]
        L.ref_DCT.argtypes = [_c_double_p, _c_double_p]
        L.ref_Liftering.argtypes = [_c_double_p]
        c = np.zeros(5, np.int32)
        L.ref_mfcc_consts(_p(c, _c_int_p))
        self.n_cep, self.n_bins, self.n_chan, self.lifter, self.block_len = (int(v) for v in c)
        self.half_rate = L.ref_mfcc_half_rate()
        L.ref_MelFilterBankInit()

    def tables(self):
        mel = np.zeros(self.n_chan + 1, np.float64)
        fi = np.zeros(self.n_bins, np.int32)
        fb = np.zeros(self.n_bins, np.float64)
        self.lib.ref_mfcc_tables(_p(mel, _c_double_p), _p(fi, _c_int_p), _p(fb, _c_double_p))
        return mel, fi, fb

    def mel_filterbank(self, mag):
        mag = np.ascontiguousarray(np.atleast_2d(mag), np.float64)
        assert mag.shape[1] == self.n_bins
        out = np.zeros((mag.shape[0], self.n_chan), np.float64)
        for r in range(mag.shape[0]):
            self.lib.ref_MelFilterBank(_p(mag[r], _c_double_p), _p(out[r], _c_double_p))
        return out

    def dct(self, mel, accumulate_into=None):
        mel = np.ascontiguousarray(np.atleast_2d(mel), np.float64)
        out = np.zeros((mel.shape[0], self.n_cep), np.float64) if accumulate_into is None else \
            np.ascontiguousarray(np.atleast_2d(accumulate_into), np.float64).copy()
        for r in range(mel.shape[0]):
            self.lib.ref_DCT(_p(mel[r], _c_double_p), _p(out[r], _c_double_p))
        return out

    def liftering(self, cep):
        cep = np.ascontiguousarray(np.atleast_2d(cep), np.float64).copy()
        for r in range(cep.shape[0]):
            self.lib.ref_Liftering(_p(cep[r], _c_double_p))
        return cep


class RefVad:
    """VoiceActivityDetection of SpectralSubtraction_final.cpp ('ss'), WienerFilter_final.cpp ('wf') or
    BeamForming_MVDR_ver1.cpp ('bf'), compiled from the reference checkout."""

    def __init__(self, path):
        L = self.lib = C.CDLL(path)
        L.ref_vad_paint.argtypes = [C.c_short]
        L.ref_VoiceActivityDetection.argtypes = [_c_short_p, C.c_int, C.c_short]
        L.ref_VoiceActivityDetection.restype = C.c_int
        L.ref_vad_consts.argtypes = [_c_double_p]
        c = np.zeros(6, np.float64)
        L.ref_vad_consts(_p(c, _c_double_p))
        self.thr_energy, self.thr_zcr, self.keep_len, self.block_len, self.n_fft, self.pi = c

    def run(self, blocks, fill=0):
        """blocks int16 [n, block_len] -> (flags int32 [n], printed energy float64 [n], printed ZCR int32 [n]).
        `fill`: the value painted over the callee's stack depth before every call = what SS:139's over-read finds."""
        blocks = np.ascontiguousarray(np.atleast_2d(blocks), np.int16)
        flags = np.zeros(blocks.shape[0], np.int32)
        with capture_c_stdout() as cap:
            for i, b in enumerate(blocks):
                flags[i] = self.lib.ref_VoiceActivityDetection(_p(b, _c_short_p), b.size, fill)
        import re
        rows = re.findall(r"dEnergy\s+(\S+)\s*,\s*dZCR\s+(-?\d+)", cap.text)
        assert len(rows) == blocks.shape[0], (len(rows), blocks.shape)
        return flags, np.array([float(r[0]) for r in rows]), np.array([int(r[1]) for r in rows], np.int32)


def ref_slice_path(name):
    return os.path.join(ORACLE_DIR, "_ref", "libref_%s.so" % name)


def load_ref_mfcc_tail():
    p = ref_slice_path("mfcc_tail")
    return RefMfccTail(p) if os.path.exists(p) else None


def load_ref_vad(which):
    p = ref_slice_path("vad_" + which)
    return RefVad(p) if os.path.exists(p) else None


def ref_path(block_len=512):
    return os.path.join(ORACLE_DIR, "_ref", "libref_fftalg_%d.so" % block_len)


def load_ref(block_len=512):
    p = ref_path(block_len)
    return RefFftAlg(p) if os.path.exists(p) else None
