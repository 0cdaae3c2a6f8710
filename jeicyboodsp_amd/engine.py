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
        self.device = int(device)
        n_cu, hbm = C.c_int(), C.c_size_t()
        name = C.create_string_buffer(96)
        self._ck(L.jdsp_device_info(h, C.byref(n_cu), C.byref(hbm), name, 96))
        self.n_cu, self.hbm_bytes, self.name = n_cu.value, hbm.value, name.value.decode()

    def close(self):
        if getattr(self, "_h", None):
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

    def synchronize(self):
        self._ck(L.jdsp_synchronize(self._h))

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
        res = np.empty((max(want, 0), n_fft), np.complex64)
        self._ck(L.jdsp_stft_i16(self._h, pcm.ctypes.data_as(C.c_void_p), total, n_fft, hop,
                                 res.ctypes.data_as(C.c_void_p), C.byref(nf)))
        assert nf.value == want
        return res
