"""Python mirror of the C ABI: one Engine == one jdsp_ctx.

Host arrays (numpy) go through the host entry points (copy in, run, copy out);
torch CUDA tensors go through the *_dev entry points on torch's current stream,
so torch.cuda.Event timing and stream ordering see the kernels.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import JdspError

L = _lib.lib


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class Engine:
    def __init__(self, device=0):
        h = C.c_void_p()
        rc = L.jdsp_create(int(device), C.byref(h))
        if rc != 0:
            raise JdspError(rc, L.jdsp_last_error(None).decode())
        self._h = h
        self._children = []           # stream objects that hold a pointer to this ctx: destroyed first
        self.device = int(device)
        n_cu, hbm = C.c_int(), C.c_size_t()
        name = C.create_string_buffer(96)
        self._ck(L.jdsp_device_info(h, C.byref(n_cu), C.byref(hbm), name, 96))
        self.n_cu, self.hbm_bytes, self.name = n_cu.value, hbm.value, name.value.decode()

    def close(self):
        if getattr(self, "_h", None):
            for c in list(self._children):
                c.close()
            L.jdsp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise JdspError(rc, L.jdsp_last_error(self._h).decode())

    def _use_torch_stream(self):
        import torch
        self._ck(L.jdsp_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def set_option(self, name, value):
        self._ck(L.jdsp_set_option(self._h, name.encode(), int(value)))

    def fastconv(self, taps, n_fft):
        return FastConv(self, taps, n_fft)

    def mvdr_multi(self, n_mics, delays=None, loading=0.0, n_fft=1024):
        return MvdrMulti(self, n_mics, delays, loading, n_fft)

    def mvdr(self, d_time=0.0):
        return Mvdr(self, d_time)

    def mfcc(self, **cfg):
        return Mfcc(self, **cfg)

    def gmm(self, classes):
        return Gmm(self, classes)

    def hmm(self, models):
        return Hmm(self, models)

    def denoiser(self, mode, n_fft=1024, hop=512):
        return Denoiser(self, mode, n_fft, hop)

    def synchronize(self):
        self._ck(L.jdsp_synchronize(self._h))

    # ---- VoiceActivityDetection on its own (SS:121-156 = WF:261-296; BF:207-242) ----
    def vad_blocks(self, blocks, variant="denoise"):
        """blocks int16 [n, 512 | 256] (host) -> (voice uint8 [n], energy sums int64 [n], zero crossings int32 [n]).
        variant "denoise": energy > 700 or ZCR < 200; "mvdr": BeamForming_MVDR_ver1.cpp's frame offset, energy only."""
        blocks = np.ascontiguousarray(np.atleast_2d(blocks), np.int16)
        n, bl = blocks.shape
        v = np.zeros(n, np.uint8)
        e = np.zeros(n, np.int64)
        z = np.zeros(n, np.int32)
        self._ck(L.jdsp_vad_blocks_ex(self._h, {"denoise": 0, "mvdr": 1}[variant], bl, _vp(blocks), n, _vp(v), _vp(e), _vp(z)))
        return v, e, z

    # ---- FFTAlgorithm_ver2.cpp ------------------------------------------------
    def bitrev_table(self, n_fft, block_len=None):
        """Bitrev table (FFTAlgorithm_ver2.cpp:186-202) computed on the device; int16[n_fft]."""
        t = np.zeros(n_fft, np.int16)
        self._ck(L.jdsp_bitrev_table(self._h, n_fft, block_len or n_fft, t.ctypes.data_as(C.c_void_p)))
        return t

    def fft_process(self, x, forward=True):
        """Batched FFTProcess (FFTAlgorithm_ver2.cpp:94-149): complex128 [..., n_fft], unnormalised."""
        if _is_torch(x):
            import torch
            assert x.is_cuda and x.dtype == torch.complex128 and x.is_contiguous()
            out = torch.empty_like(x)
            n = x.shape[-1]
            self._use_torch_stream()
            self._ck(L.jdsp_fft_process_f64_dev(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), n,
                                                x.numel() // n, int(forward)))
            return out
        x = np.ascontiguousarray(x, np.complex128)
        out = np.empty_like(x)
        n = x.shape[-1]
        self._ck(L.jdsp_fft_process_f64(self._h, x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), n,
                                        x.size // n, int(forward)))
        return out

    DFT_I16, IDFT, IDFT_OVER_N = 0, 1, 2

    def dft_direct(self, kind, x, accumulate_into=None):
        """DFTProcess / IDFTProcess / IFFTProcess (FFTAlgorithm_ver2.cpp:162-173 / :175-184 / :151-160): the
        O(N^2) sums as the reference writes them, any length, ADDED to `accumulate_into` (default: zeros).
        x: int16 [..., n] for DFT_I16, complex128 [..., n] otherwise.  Returns complex128 [..., n]."""
        x = np.ascontiguousarray(x, np.int16 if kind == self.DFT_I16 else np.complex128)
        n = x.shape[-1]
        out = np.zeros(x.shape, np.complex128) if accumulate_into is None else \
            np.ascontiguousarray(accumulate_into, np.complex128).copy()
        assert out.shape == x.shape
        self._ck(L.jdsp_dft_direct_f64(self._h, int(kind), x.ctypes.data_as(C.c_void_p),
                                       out.ctypes.data_as(C.c_void_p), n, x.size // n))
        return out

    def gmm_probability(self, feats, mean, cov, eig):
        """probability() (GMMAlgorithm_Test_Auto_ver2.cpp:164-236) of every 12-double vector of feats under one
        mixture component (mean [12], cov [12, 12], eig [12, 4])."""
        feats = np.ascontiguousarray(np.atleast_2d(feats), np.float64)
        mean = np.ascontiguousarray(mean, np.float64)
        cov = np.ascontiguousarray(cov, np.float64)
        eig = np.ascontiguousarray(eig, np.float64)
        assert feats.shape[1] == 12 and mean.size == 12 and cov.size == 144 and eig.size == 48
        out = np.empty(feats.shape[0], np.float64)
        self._ck(L.jdsp_gmm_probability(self._h, _vp(feats), feats.shape[0], _vp(mean), _vp(cov), _vp(eig), _vp(out)))
        return out

    # ---- PitchEstimation_method1.cpp ---------------------------------------------
    def pitch(self, pcm, prev_block=None, want_autocorr=False):
        """CalcPitch (PitchEstimation_method1.cpp:69-116) for every 512-sample block of pcm:
        returns (arg int32[nb], rmax float32[nb][, autocorr float32[nb,512]])."""
        if _is_torch(pcm):
            import torch
            assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.is_contiguous() and pcm.numel() % 512 == 0
            nb = pcm.numel() // 512
            arg = torch.empty(nb, dtype=torch.int32, device=pcm.device)
            rmax = torch.empty(nb, dtype=torch.float32, device=pcm.device)
            ac = torch.empty((nb, 512), dtype=torch.float32, device=pcm.device) if want_autocorr else None
            self._use_torch_stream()
            self._ck(L.jdsp_pitch_autocorr_dev(self._h, C.c_void_p(pcm.data_ptr()), nb,
                                               C.c_void_p(prev_block.data_ptr()) if prev_block is not None else None,
                                               C.c_void_p(arg.data_ptr()), C.c_void_p(rmax.data_ptr()),
                                               C.c_void_p(ac.data_ptr()) if want_autocorr else None))
            return (arg, rmax, ac) if want_autocorr else (arg, rmax)
        pcm = np.ascontiguousarray(pcm, np.int16)
        assert pcm.size % 512 == 0
        nb = pcm.size // 512
        arg = np.zeros(nb, np.int32)
        rmax = np.zeros(nb, np.float32)
        ac = np.zeros((nb, 512), np.float32) if want_autocorr else None
        pb = np.ascontiguousarray(prev_block, np.int16) if prev_block is not None else None
        self._ck(L.jdsp_pitch_autocorr(self._h, pcm.ctypes.data_as(C.c_void_p), nb,
                                       pb.ctypes.data_as(C.c_void_p) if pb is not None else None,
                                       arg.ctypes.data_as(C.c_void_p), rmax.ctypes.data_as(C.c_void_p),
                                       ac.ctypes.data_as(C.c_void_p) if want_autocorr else None))
        return (arg, rmax, ac) if want_autocorr else (arg, rmax)

    def stft_half(self, pcm, n_frames=None, out=None, pitch=513):
        """Bins 0..512 only: complex64 [n_frames, pitch >= 513], columns past 512 untouched."""
        import torch
        assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.is_contiguous()
        if n_frames is None:
            n_frames = (pcm.numel() - 1024) // 512 + 1 if pcm.numel() >= 1024 else 0
        if out is None:
            out = torch.empty((n_frames, pitch), dtype=torch.complex64, device=pcm.device)
        assert out.is_contiguous() and out.shape[-1] == pitch
        self._use_torch_stream()
        self._ck(L.jdsp_stft_half_i16_dev(self._h, C.c_void_p(pcm.data_ptr()), n_frames, C.c_void_p(out.data_ptr()),
                                          pitch))
        return out

    # ---- STFT analysis (SS:218-230 / WF:181-193 for a whole batch) ---------
    def stft(self, pcm, n_frames=None, n_fft=1024, hop=512, out=None):
        """pcm: int16 numpy array (host path) or torch CUDA int16 tensor (device path).
        Returns complex64 [n_frames, n_fft]."""
        if _is_torch(pcm):
            import torch
            assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.is_contiguous()
            total = pcm.numel()
            if n_frames is None:
                n_frames = (total - n_fft) // hop + 1 if total >= n_fft else 0
            assert n_frames == 0 or hop * (n_frames - 1) + n_fft <= total, "pcm too short"
            if out is None:
                out = torch.empty((n_frames, n_fft), dtype=torch.complex64, device=pcm.device)
            assert out.is_contiguous() and out.dtype == torch.complex64 and out.numel() >= n_frames * n_fft
            self._use_torch_stream()
            self._ck(L.jdsp_stft_i16_dev(self._h, C.c_void_p(pcm.data_ptr()), n_frames, n_fft, hop,
                                         C.c_void_p(out.data_ptr())))
            return out
        pcm = np.ascontiguousarray(pcm, np.int16)
        nf = C.c_long()
        total = pcm.size
        want = (total - n_fft) // hop + 1 if total >= n_fft else 0
        if out is not None:              # e.g. a pinned buffer: the library then pipelines the PCIe copies
            res = out
            assert res.dtype == np.complex64 and res.flags.c_contiguous and res.shape == (max(want, 0), n_fft)
        else:
            res = np.empty((max(want, 0), n_fft), np.complex64)
        self._ck(L.jdsp_stft_i16(self._h, pcm.ctypes.data_as(C.c_void_p), total, n_fft, hop,
                                 res.ctypes.data_as(C.c_void_p), C.byref(nf)))
        assert nf.value == want
        return res

    def stft_f64(self, pcm, n_frames=None, hop=512, out=None):
        """The 1024-point analysis in FP64 (jdsp_stft_i16_f64*): complex128 [n_frames, 1024]."""
        n_fft = 1024
        if _is_torch(pcm):
            import torch
            assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.is_contiguous()
            total = pcm.numel()
            if n_frames is None:
                n_frames = (total - n_fft) // hop + 1 if total >= n_fft else 0
            assert n_frames == 0 or hop * (n_frames - 1) + n_fft <= total, "pcm too short"
            if out is None:
                out = torch.empty((n_frames, n_fft), dtype=torch.complex128, device=pcm.device)
            assert out.is_contiguous() and out.dtype == torch.complex128 and out.numel() >= n_frames * n_fft
            self._use_torch_stream()
            self._ck(L.jdsp_stft_i16_f64_dev(self._h, C.c_void_p(pcm.data_ptr()), n_frames, n_fft, hop,
                                             C.c_void_p(out.data_ptr())))
            return out
        pcm = np.ascontiguousarray(pcm, np.int16)
        nf = C.c_long()
        want = (pcm.size - n_fft) // hop + 1 if pcm.size >= n_fft else 0
        res = np.empty((max(want, 0), n_fft), np.complex128)
        self._ck(L.jdsp_stft_i16_f64(self._h, pcm.ctypes.data_as(C.c_void_p), pcm.size, n_fft, hop,
                                     res.ctypes.data_as(C.c_void_p), C.byref(nf)))
        assert nf.value == want
        return res


class Denoiser:
    """One SS/Wiener audio stream (jdsp_denoise): mirrors main()'s loop of
    SpectralSubtraction_final.cpp:92-113 / WienerFilter_final.cpp:91-112 for batches of blocks."""
    SPECSUB, WIENER = 0, 1

    def __init__(self, engine, mode, n_fft=1024, hop=512):
        self.eng = engine
        h = C.c_void_p()
        engine._ck(L.jdsp_denoise_create_cfg(engine._h, int(mode), int(n_fft), int(hop), C.byref(h)))
        self._h = h
        self.n_fft, self.block = int(n_fft), L.jdsp_denoise_block_len(h)
        engine._children.append(self)

    def close(self):
        if getattr(self, "_h", None):
            L.jdsp_denoise_destroy(self._h)
            self._h = None
            if self in self.eng._children:
                self.eng._children.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.eng._ck(L.jdsp_denoise_reset(self._h))

    def set_option(self, name, value):
        self.eng._ck(L.jdsp_denoise_set_option(self._h, name.encode(), int(value)))

    def blocks_out(self, n_blocks):
        return L.jdsp_denoise_blocks_out(self._h, n_blocks)

    def process(self, pcm, want_precast=False):
        """pcm: int16, a whole number of blocks (self.block samples: 512, or 256 for 512-point frames).
        numpy -> host path, torch CUDA -> device path.  Returns out (int16) or (out, precast float32)."""
        B = self.block
        if _is_torch(pcm):
            import torch
            assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.is_contiguous() and pcm.numel() % B == 0
            nb = pcm.numel() // B
            n_out = self.blocks_out(nb)
            out = torch.empty(max(n_out, 1) * B, dtype=torch.int16, device=pcm.device)
            pre = torch.empty(max(n_out, 1) * B, dtype=torch.float32, device=pcm.device) if want_precast else None
            self.eng._use_torch_stream()
            self.eng._ck(L.jdsp_denoise_process_dev(self._h, C.c_void_p(pcm.data_ptr()), nb, C.c_void_p(out.data_ptr()),
                                                    C.c_void_p(pre.data_ptr()) if want_precast else None, None))
            out = out[:n_out * B]
            return (out, pre[:n_out * B]) if want_precast else out
        pcm = np.ascontiguousarray(pcm, np.int16)
        assert pcm.size % B == 0
        nb = pcm.size // B
        n_out = self.blocks_out(nb)
        out = np.zeros(max(n_out, 1) * B, np.int16)
        pre = np.zeros(max(n_out, 1) * B, np.float32) if want_precast else None
        got = C.c_long()
        self.eng._ck(L.jdsp_denoise_process(self._h, pcm.ctypes.data_as(C.c_void_p), nb, out.ctypes.data_as(C.c_void_p),
                                            pre.ctypes.data_as(C.c_void_p) if want_precast else None, C.byref(got)))
        assert got.value == n_out
        return (out[:n_out * B], pre[:n_out * B]) if want_precast else out[:n_out * B]

    # ---- one rank's share of a global stream (jdsp_denoise_shard_*; driver: sharding.denoise_sharded)
    def shard_vad(self, pcm_ext, ext0, b0, b1, n_total):
        import torch
        flags = torch.empty(max(b1 - b0, 1), dtype=torch.uint8, device=pcm_ext.device)
        self.eng._use_torch_stream()
        self.eng._ck(L.jdsp_denoise_shard_vad_dev(self._h, C.c_void_p(pcm_ext.data_ptr()), ext0, b0, b1, n_total,
                                                  C.c_void_p(flags.data_ptr())))
        return flags[: b1 - b0]

    def shard_summary(self, flags_all):
        import torch
        out = torch.empty(1025, dtype=torch.float32, device=flags_all.device)
        self.eng._use_torch_stream()
        self.eng._ck(L.jdsp_denoise_shard_summary_dev(self._h, C.c_void_p(flags_all.data_ptr()), C.c_void_p(out.data_ptr())))
        return out

    def shard_rows(self, summaries_all, world, rank):
        import torch
        out = torch.empty(1025, dtype=torch.float32, device=summaries_all.device)
        self.eng._use_torch_stream()
        self.eng._ck(L.jdsp_denoise_shard_rows_dev(self._h, C.c_void_p(summaries_all.data_ptr()), world, rank,
                                                   C.c_void_p(out.data_ptr())))
        return out

    def shard_finish(self, last_all, world, rank, want_precast=False):
        import torch
        n_out = L.jdsp_denoise_shard_blocks_out(self._h)
        B = self.block
        out = torch.empty(max(n_out, 1) * B, dtype=torch.int16, device=last_all.device)
        pre = torch.empty(max(n_out, 1) * B, dtype=torch.float32, device=last_all.device) if want_precast else None
        self.eng._use_torch_stream()
        self.eng._ck(L.jdsp_denoise_shard_finish_dev(self._h, C.c_void_p(last_all.data_ptr()), world, rank,
                                                     C.c_void_p(out.data_ptr()),
                                                     C.c_void_p(pre.data_ptr()) if want_precast else None, None))
        return (out[: n_out * B], pre[: n_out * B]) if want_precast else out[: n_out * B]

    def apply(self, pcm, noise, want_precast=False):
        """SpectralSubtraction / WienerFiltering (SS:201-264 / WF:162-235) over whole blocks with the CALLER's
        pdEstimatedNoiseSpec (n_fft doubles) instead of the handle's own VAD + estimate (jdsp_denoise_apply; host)."""
        B = self.block
        pcm = np.ascontiguousarray(pcm, np.int16)
        noise = np.ascontiguousarray(noise, np.float64)
        assert pcm.size % B == 0 and noise.size == self.n_fft
        nb = pcm.size // B
        n_out = self.blocks_out(nb)
        out = np.zeros(max(n_out, 1) * B, np.int16)
        pre = np.zeros(max(n_out, 1) * B, np.float32) if want_precast else None
        self.eng._ck(L.jdsp_denoise_apply(self._h, _vp(pcm), nb, _vp(noise), _vp(out), _vp(pre), None))
        return (out[:n_out * B], pre[:n_out * B]) if want_precast else out[:n_out * B]

    def noise(self):
        n = np.zeros(self.n_fft, np.float64)
        self.eng._ck(L.jdsp_denoise_noise(self._h, n.ctypes.data_as(C.c_void_p)))
        return n

    def vad_trace(self, n, flags_only=False):
        """(voice, energy sums, ZCR) of the last call's first n blocks; energies / ZCR need set_option("vad_trace", 1)
        before that call (flags_only=True asks for the flags alone, which are always kept)."""
        if flags_only:
            v = np.zeros(n, np.uint8)
            self.eng._ck(L.jdsp_denoise_vad_trace(self._h, n, v.ctypes.data_as(C.c_void_p), None, None))
            return v
        v = np.zeros(n, np.uint8)
        e = np.zeros(n, np.int64)
        z = np.zeros(n, np.int32)
        self.eng._ck(L.jdsp_denoise_vad_trace(self._h, n, v.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p),
                                              z.ctypes.data_as(C.c_void_p)))
        return v, e, z


class Mfcc:
    """MFCC front end (jdsp_mfcc): MFCCFeatureExtraction_auto_version1.cpp for batches of frames.
    Keyword arguments override the reference-native configuration (jdsp_mfcc_native_cfg)."""

    def __init__(self, engine, **kw):
        self.eng = engine
        cfg = _lib.MfccCfg()
        engine._ck(L.jdsp_mfcc_native_cfg(C.byref(cfg)))
        for k, v in kw.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        h = C.c_void_p()
        engine._ck(L.jdsp_mfcc_create(engine._h, C.byref(cfg), C.byref(h)))
        self._h = h
        engine._children.append(self)

    def close(self):
        if getattr(self, "_h", None):
            L.jdsp_mfcc_destroy(self._h)
            self._h = None
            if self in self.eng._children:
                self.eng._children.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tables(self):
        """MelFilterBankInit's rgdMelFreqs, rgdFiBins, rgdFilterBank."""
        nb = self.cfg.n_fft // 2
        mel = np.zeros(self.cfg.n_chan + 1, np.float64)
        fi = np.zeros(nb, np.int32)
        fb = np.zeros(nb, np.float64)
        self.eng._ck(L.jdsp_mfcc_tables(self._h, mel.ctypes.data_as(C.c_void_p), fi.ctypes.data_as(C.c_void_p),
                                        fb.ctypes.data_as(C.c_void_p)))
        return mel, fi, fb

    def n_frames(self, n_samples):
        return (n_samples - self.cfg.win_len) // self.cfg.hop + 1 if n_samples >= self.cfg.win_len else 0

    # ---- the sub-steps on their own (MFCC:154-192), FP64, rows = independent frames
    def mel_filterbank(self, mag):
        """MelFilterBank: |X| [rows, n_fft/2] -> ln channel sums [rows, n_chan]."""
        mag = np.ascontiguousarray(np.atleast_2d(mag), np.float64)
        assert mag.shape[1] == self.cfg.n_fft // 2
        out = np.empty((mag.shape[0], self.cfg.n_chan), np.float64)
        self.eng._ck(L.jdsp_mfcc_melfilterbank(self._h, _vp(mag), mag.shape[0], _vp(out)))
        return out

    def dct(self, mel, accumulate_into=None):
        """DCT: [rows, n_chan] -> [rows, n_cep], ADDED to accumulate_into (default zeros) like the reference."""
        mel = np.ascontiguousarray(np.atleast_2d(mel), np.float64)
        assert mel.shape[1] == self.cfg.n_chan
        out = np.zeros((mel.shape[0], self.cfg.n_cep), np.float64) if accumulate_into is None else \
            np.ascontiguousarray(np.atleast_2d(accumulate_into), np.float64).copy()
        assert out.shape == (mel.shape[0], self.cfg.n_cep)
        self.eng._ck(L.jdsp_mfcc_dct(self._h, _vp(mel), mel.shape[0], _vp(out)))
        return out

    def liftering(self, cep):
        cep = np.ascontiguousarray(np.atleast_2d(cep), np.float64).copy()
        assert cep.shape[1] == self.cfg.n_cep
        self.eng._ck(L.jdsp_mfcc_liftering(self._h, _vp(cep), cep.shape[0]))
        return cep

    def frames(self, pcm, n_frames=None, frame_start=None, out=None):
        """Feature vectors [n_frames, n_cep] float64; frame j starts at frame_start[j] (default hop*j).
        out (device path only): a caller-owned [n_frames, n_cep] float64 CUDA tensor (no allocation: graph capture)."""
        if _is_torch(pcm):
            import torch
            assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.is_contiguous()
            if n_frames is None:
                n_frames = len(frame_start) if frame_start is not None else self.n_frames(pcm.numel())
            if out is None:
                out = torch.empty((n_frames, self.cfg.n_cep), dtype=torch.float64, device=pcm.device)
            assert out.is_cuda and out.dtype == torch.float64 and out.is_contiguous() and out.numel() >= n_frames * self.cfg.n_cep
            if frame_start is not None:
                assert frame_start.is_cuda and frame_start.dtype == torch.int64 and frame_start.numel() >= n_frames
            self.eng._use_torch_stream()
            self.eng._ck(L.jdsp_mfcc_frames_dev(self._h, C.c_void_p(pcm.data_ptr()),
                                                C.c_void_p(frame_start.data_ptr()) if frame_start is not None else None,
                                                n_frames, C.c_void_p(out.data_ptr())))
            return out
        pcm = np.ascontiguousarray(pcm, np.int16)
        if frame_start is not None:
            frame_start = np.ascontiguousarray(frame_start, np.int64)
        if n_frames is None:
            n_frames = len(frame_start) if frame_start is not None else self.n_frames(pcm.size)
        out = np.zeros((n_frames, self.cfg.n_cep), np.float64)
        self.eng._ck(L.jdsp_mfcc_frames(self._h, pcm.ctypes.data_as(C.c_void_p), pcm.size,
                                        frame_start.ctypes.data_as(C.c_void_p) if frame_start is not None else None,
                                        n_frames, out.ctypes.data_as(C.c_void_p)))
        return out


# numpy views of the reference's parameter records (GMMAlgorithm_Test_Auto_ver2.cpp:29-34,
# Viterbi_version1.cpp:30-40): np.fromfile(path, GMM_PARAM) reads a parameter file
GMM_PARAM = np.dtype([("alpa", "<f8", (4,)), ("mean", "<f8", (4, 12)), ("covariance", "<f8", (4, 12, 12)),
                      ("eigenVector", "<f8", (4, 12, 4))])
HMM_PARAM = np.dtype([("gMMParam", GMM_PARAM, (6,)), ("transProb", "<f8", (6, 6))])


def _vp(a):
    if a is None:
        return None
    if _is_torch(a):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)


class _Child:
    _destroy = None

    def close(self):
        if getattr(self, "_h", None):
            type(self)._destroy(self._h)
            self._h = None
            if self in self.eng._children:
                self.eng._children.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Gmm(_Child):
    """GMM scoring (jdsp_gmm): Recognition() of GMMAlgorithm_Test_Auto_ver2.cpp for batches of utterances.
    `classes`: numpy array of GMM_PARAM records, one per class."""
    _destroy = staticmethod(lambda h: L.jdsp_gmm_destroy(h))

    def __init__(self, engine, classes):
        self.eng = engine
        classes = np.ascontiguousarray(classes, GMM_PARAM).reshape(-1)
        self.n_classes = len(classes)
        h = C.c_void_p()
        engine._ck(L.jdsp_gmm_create(engine._h, _vp(classes), self.n_classes, C.byref(h)))
        self._h = h
        engine._children.append(self)

    def set_option(self, name, value):
        self.eng._ck(L.jdsp_gmm_set_option(self._h, name.encode(), int(value)))

    def score(self, feats, utt_first):
        """feats [n_vectors, 12] float64, utt_first [n_utts + 1] int64 -> (scores [n_utts, n_classes], best [n_utts]).
        torch CUDA tensors stay on the device; numpy arrays go through the host entry."""
        n_utts = len(utt_first) - 1
        if _is_torch(feats):
            import torch
            assert feats.is_cuda and feats.dtype == torch.float64 and feats.is_contiguous()
            assert utt_first.is_cuda and utt_first.dtype == torch.int64
            scores = torch.empty((n_utts, self.n_classes), dtype=torch.float64, device=feats.device)
            best = torch.empty((n_utts,), dtype=torch.int32, device=feats.device)
            self.eng._use_torch_stream()
            self.eng._ck(L.jdsp_gmm_score_dev(self._h, _vp(feats), feats.shape[0], _vp(utt_first), n_utts, _vp(scores),
                                              _vp(best)))
            return scores, best
        feats = np.ascontiguousarray(feats, np.float64)
        utt_first = np.ascontiguousarray(utt_first, np.int64)
        scores = np.empty((n_utts, self.n_classes), np.float64)
        best = np.empty((n_utts,), np.int32)
        self.eng._ck(L.jdsp_gmm_score(self._h, _vp(feats), _vp(utt_first), n_utts, _vp(scores), _vp(best)))
        return scores, best


class Hmm(_Child):
    """Six-state HMM recursion (jdsp_hmm): HMMRecognition() of Viterbi_version1.cpp for batches of utterances.
    `models`: numpy array of HMM_PARAM records."""
    _destroy = staticmethod(lambda h: L.jdsp_hmm_destroy(h))

    def __init__(self, engine, models):
        self.eng = engine
        models = np.ascontiguousarray(models, HMM_PARAM).reshape(-1)
        self.n_models = len(models)
        h = C.c_void_p()
        engine._ck(L.jdsp_hmm_create(engine._h, _vp(models), self.n_models, C.byref(h)))
        self._h = h
        engine._children.append(self)

    def set_option(self, name, value):
        self.eng._ck(L.jdsp_hmm_set_option(self._h, name.encode(), int(value)))

    def viterbi(self, feats, utt_first, want_path=True, want_trellis=False):
        """-> (scores [n_utts, n_models], best [n_utts], path [n_models, n_vectors] or None[, trellis])."""
        n_utts = len(utt_first) - 1
        n_frames = feats.shape[0]
        if _is_torch(feats):
            import torch
            assert feats.is_cuda and feats.dtype == torch.float64 and feats.is_contiguous()
            assert utt_first.is_cuda and utt_first.dtype == torch.int64
            dev = feats.device
            scores = torch.empty((n_utts, self.n_models), dtype=torch.float64, device=dev)
            best = torch.empty((n_utts,), dtype=torch.int32, device=dev)
            path = torch.zeros((self.n_models, n_frames), dtype=torch.int32, device=dev) if want_path else None
            trellis = torch.zeros((self.n_models, 6, n_frames), dtype=torch.float64, device=dev) if want_trellis else None
            self.eng._use_torch_stream()
            self.eng._ck(L.jdsp_hmm_viterbi_dev(self._h, _vp(feats), n_frames, _vp(utt_first), n_utts, _vp(scores),
                                                _vp(best), _vp(path), _vp(trellis)))
            return (scores, best, path, trellis) if want_trellis else (scores, best, path)
        feats = np.ascontiguousarray(feats, np.float64)
        utt_first = np.ascontiguousarray(utt_first, np.int64)
        scores = np.empty((n_utts, self.n_models), np.float64)
        best = np.empty((n_utts,), np.int32)
        path = np.zeros((self.n_models, n_frames), np.int32) if want_path else None
        trellis = np.zeros((self.n_models, 6, n_frames), np.float64) if want_trellis else None
        self.eng._ck(L.jdsp_hmm_viterbi(self._h, _vp(feats), _vp(utt_first), n_utts, _vp(scores), _vp(best), _vp(path),
                                        _vp(trellis)))
        return (scores, best, path, trellis) if want_trellis else (scores, best, path)


class FastConv:
    """Overlap-save convolver (jdsp_fastconv): AnalySisFreqDomain of
    Fast_Convolution_Based_3DAudio_Impl.cpp:102-177 for batches of blocks.
    taps: [n_taps] or [n_filters, n_taps] float64."""

    def __init__(self, engine, taps, n_fft):
        self.eng = engine
        taps = np.ascontiguousarray(np.atleast_2d(np.asarray(taps, np.float64)))
        self.n_filters, self.n_taps = taps.shape
        h = C.c_void_p()
        engine._ck(L.jdsp_fastconv_create(engine._h, taps.ctypes.data_as(C.c_void_p), self.n_taps, self.n_filters,
                                          int(n_fft), C.byref(h)))
        self._h = h
        self.block = L.jdsp_fastconv_block_len(h)
        self.hist_blocks = L.jdsp_fastconv_hist_blocks(h)
        engine._children.append(self)

    def close(self):
        if getattr(self, "_h", None):
            L.jdsp_fastconv_destroy(self._h)
            self._h = None
            if self in self.eng._children:
                self.eng._children.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.eng._ck(L.jdsp_fastconv_reset(self._h))

    def set_position(self, blocks_consumed):
        self.eng._ck(L.jdsp_fastconv_set_position(self._h, int(blocks_consumed)))

    def blocks_out(self, n_blocks):
        return L.jdsp_fastconv_blocks_out(self._h, n_blocks)

    def process(self, pcm, want_precast=False):
        """pcm: int16, whole blocks of self.block samples.  Returns out [n_filters, n_out*block] (and precast)."""
        if _is_torch(pcm):
            import torch
            assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.is_contiguous() and pcm.numel() % self.block == 0
            nb = pcm.numel() // self.block
            n_out = self.blocks_out(nb)
            shape = (self.n_filters, max(n_out, 1) * self.block)
            out = torch.empty(shape, dtype=torch.int16, device=pcm.device)
            pre = torch.empty(shape, dtype=torch.float32, device=pcm.device) if want_precast else None
            if n_out == 0:
                out, pre = out[:, :0], (pre[:, :0] if want_precast else None)
            self.eng._use_torch_stream()
            self.eng._ck(L.jdsp_fastconv_process_dev(self._h, C.c_void_p(pcm.data_ptr()), nb, C.c_void_p(out.data_ptr()),
                                                     C.c_void_p(pre.data_ptr()) if want_precast else None, None))
            return (out, pre) if want_precast else out
        pcm = np.ascontiguousarray(pcm, np.int16)
        assert pcm.size % self.block == 0
        nb = pcm.size // self.block
        n_out = self.blocks_out(nb)
        out = np.zeros((self.n_filters, n_out * self.block), np.int16)
        pre = np.zeros((self.n_filters, n_out * self.block), np.float32) if want_precast else None
        dummy = np.zeros(1, np.int16)
        self.eng._ck(L.jdsp_fastconv_process(self._h, pcm.ctypes.data_as(C.c_void_p), nb,
                                             (out if out.size else dummy).ctypes.data_as(C.c_void_p),
                                             pre.ctypes.data_as(C.c_void_p) if (want_precast and pre.size) else None, None))
        return (out, pre) if want_precast else out


class Mvdr:
    """Two-microphone MVDR beamformer (jdsp_mvdr): main()'s loop of BeamForming_MVDR_ver1.cpp:169-231
    for batches of 512-sample blocks per channel."""

    def __init__(self, engine, d_time=0.0):
        self.eng = engine
        h = C.c_void_p()
        engine._ck(L.jdsp_mvdr_create(engine._h, float(d_time), C.byref(h)))
        self._h = h
        engine._children.append(self)

    def close(self):
        if getattr(self, "_h", None):
            L.jdsp_mvdr_destroy(self._h)
            self._h = None
            if self in self.eng._children:
                self.eng._children.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.eng._ck(L.jdsp_mvdr_reset(self._h))

    def blocks_out(self, n_blocks):
        return L.jdsp_mvdr_blocks_out(self._h, n_blocks)

    def corr(self):
        c = np.zeros(4, np.float64)
        self.eng._ck(L.jdsp_mvdr_corr(self._h, c.ctypes.data_as(C.c_void_p)))
        return c

    def estimate_corr(self, left_frames, right_frames, corr):
        """EstimateSpatialCorrMtx (BeamForming_MVDR_ver1.cpp:244-270): frames [n, 1024] per channel, returns
        corr (4 doubles, row-major) + every frame's contribution."""
        lf = np.ascontiguousarray(np.atleast_2d(left_frames), np.int16)
        rf = np.ascontiguousarray(np.atleast_2d(right_frames), np.int16)
        assert lf.shape == rf.shape and lf.shape[1] == 1024
        c = np.ascontiguousarray(corr, np.float64).reshape(4).copy()
        self.eng._ck(L.jdsp_mvdr_estimate_corr(self._h, _vp(lf), _vp(rf), lf.shape[0], _vp(c)))
        return c

    def apply(self, left, right, corr, want_precast=False):
        """ProcessMVDR (:124-205) for whole blocks with the CALLER's matrix (4 doubles, row-major)."""
        left = np.ascontiguousarray(left, np.int16)
        right = np.ascontiguousarray(right, np.int16)
        c = np.ascontiguousarray(corr, np.float64).reshape(4)
        nb = left.size // 512
        n_out = self.blocks_out(nb)
        out = np.zeros(max(n_out, 1) * 512, np.int16)
        pre = np.zeros(max(n_out, 1) * 512, np.float32) if want_precast else None
        self.eng._ck(L.jdsp_mvdr_apply(self._h, _vp(left), _vp(right), nb, _vp(c), _vp(out), _vp(pre), None))
        return (out[:n_out * 512], pre[:n_out * 512]) if want_precast else out[:n_out * 512]

    # ---- one rank's share of a global stream (jdsp_mvdr_shard_*)
    def shard_vad(self, left_ext, right_ext, ext0, b0, b1, n_total):
        import torch
        flags = torch.empty(max(b1 - b0, 1), dtype=torch.uint8, device=left_ext.device)
        self.eng._use_torch_stream()
        self.eng._ck(L.jdsp_mvdr_shard_vad_dev(self._h, C.c_void_p(left_ext.data_ptr()), C.c_void_p(right_ext.data_ptr()),
                                               ext0, b0, b1, n_total, C.c_void_p(flags.data_ptr())))
        return flags[: b1 - b0]

    def shard_summary(self, flags_all):
        import torch
        out = torch.empty(4, dtype=torch.float64, device=flags_all.device)
        self.eng._use_torch_stream()
        self.eng._ck(L.jdsp_mvdr_shard_summary_dev(self._h, C.c_void_p(flags_all.data_ptr()), C.c_void_p(out.data_ptr())))
        return out

    def shard_finish(self, sums_all, world, rank):
        import torch
        n_out = L.jdsp_mvdr_shard_blocks_out(self._h)
        out = torch.empty(max(n_out, 1) * 512, dtype=torch.int16, device=sums_all.device)
        self.eng._use_torch_stream()
        self.eng._ck(L.jdsp_mvdr_shard_finish_dev(self._h, C.c_void_p(sums_all.data_ptr()), world, rank,
                                                  C.c_void_p(out.data_ptr()), None, None))
        return out[: n_out * 512]

    def process(self, left, right, want_precast=False):
        if _is_torch(left):
            import torch
            assert left.is_cuda and right.is_cuda and left.dtype == right.dtype == torch.int16
            assert left.is_contiguous() and right.is_contiguous() and left.numel() == right.numel() and left.numel() % 512 == 0
            nb = left.numel() // 512
            n_out = self.blocks_out(nb)
            out = torch.empty(max(n_out, 1) * 512, dtype=torch.int16, device=left.device)
            pre = torch.empty(max(n_out, 1) * 512, dtype=torch.float32, device=left.device) if want_precast else None
            self.eng._use_torch_stream()
            self.eng._ck(L.jdsp_mvdr_process_dev(self._h, C.c_void_p(left.data_ptr()), C.c_void_p(right.data_ptr()), nb,
                                                 C.c_void_p(out.data_ptr()),
                                                 C.c_void_p(pre.data_ptr()) if want_precast else None, None))
            out = out[:n_out * 512]
            return (out, pre[:n_out * 512]) if want_precast else out
        left = np.ascontiguousarray(left, np.int16)
        right = np.ascontiguousarray(right, np.int16)
        assert left.size == right.size and left.size % 512 == 0
        nb = left.size // 512
        n_out = self.blocks_out(nb)
        out = np.zeros(max(n_out, 1) * 512, np.int16)
        pre = np.zeros(max(n_out, 1) * 512, np.float32) if want_precast else None
        self.eng._ck(L.jdsp_mvdr_process(self._h, left.ctypes.data_as(C.c_void_p), right.ctypes.data_as(C.c_void_p), nb,
                                         out.ctypes.data_as(C.c_void_p),
                                         pre.ctypes.data_as(C.c_void_p) if want_precast else None, None))
        return (out[:n_out * 512], pre[:n_out * 512]) if want_precast else out[:n_out * 512]


class MvdrMulti:
    """MVDR generalised to n_mics <= 8 with a per-bin covariance (jdsp_mvdrn, BASELINE config 5).
    pcm: int16 [n_mics, n_blocks * block] (planar); block = n_fft / 2 (512, or 256 for 512-point frames)."""

    def __init__(self, engine, n_mics, delays=None, loading=0.0, n_fft=1024):
        self.eng = engine
        self.n_mics = int(n_mics)
        d = np.ascontiguousarray(delays if delays is not None else np.zeros(n_mics), np.float64)
        assert d.size == n_mics
        h = C.c_void_p()
        engine._ck(L.jdsp_mvdrn_create_cfg(engine._h, self.n_mics, d.ctypes.data_as(C.c_void_p), float(loading), int(n_fft),
                                           C.byref(h)))
        self._h = h
        self.block = L.jdsp_mvdrn_block_len(h)
        engine._children.append(self)

    def close(self):
        if getattr(self, "_h", None):
            L.jdsp_mvdrn_destroy(self._h)
            self._h = None
            if self in self.eng._children:
                self.eng._children.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.eng._ck(L.jdsp_mvdrn_reset(self._h))

    def blocks_out(self, n_blocks):
        return L.jdsp_mvdrn_blocks_out(self._h, n_blocks)

    def process(self, pcm, want_precast=False):
        B = self.block
        if _is_torch(pcm):
            import torch
            assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.is_contiguous() and pcm.dim() == 2
            assert pcm.shape[0] == self.n_mics and pcm.shape[1] % B == 0
            nb = pcm.shape[1] // B
            n_out = self.blocks_out(nb)
            out = torch.empty(max(n_out, 1) * B, dtype=torch.int16, device=pcm.device)
            pre = torch.empty(max(n_out, 1) * B, dtype=torch.float32, device=pcm.device) if want_precast else None
            self.eng._use_torch_stream()
            self.eng._ck(L.jdsp_mvdrn_process_dev(self._h, C.c_void_p(pcm.data_ptr()), pcm.shape[1], nb,
                                                  C.c_void_p(out.data_ptr()),
                                                  C.c_void_p(pre.data_ptr()) if want_precast else None, None))
            out = out[:n_out * B]
            return (out, pre[:n_out * B]) if want_precast else out
        pcm = np.ascontiguousarray(pcm, np.int16)
        assert pcm.ndim == 2 and pcm.shape[0] == self.n_mics and pcm.shape[1] % B == 0
        nb = pcm.shape[1] // B
        n_out = self.blocks_out(nb)
        out = np.zeros(max(n_out, 1) * B, np.int16)
        pre = np.zeros(max(n_out, 1) * B, np.float32) if want_precast else None
        self.eng._ck(L.jdsp_mvdrn_process(self._h, pcm.ctypes.data_as(C.c_void_p), pcm.shape[1], nb,
                                          out.ctypes.data_as(C.c_void_p),
                                          pre.ctypes.data_as(C.c_void_p) if want_precast else None, None))
        return (out[:n_out * B], pre[:n_out * B]) if want_precast else out[:n_out * B]
