"""ctypes binding of libjdsp.so (the C ABI in include/jdsp.h).

There is no fallback: if the HIP library has not been built, importing this
module raises, and if there is no gfx950 GPU, creating an Engine raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# JDSP_LIB: a differently built libjdsp.so (tools/ A/B timing of build-time variants); default: the in-tree build
LIB_PATH = os.environ.get("JDSP_LIB") or os.path.join(_HERE, "libjdsp.so")

OK, EINVAL, EHIP, ENOMEM, ENODEV = 0, -1, -2, -3, -4


class MfccCfg(C.Structure):
    """jdsp_mfcc_cfg (include/jdsp.h)"""
    _fields_ = [("win_len", C.c_int), ("hop", C.c_int), ("n_fft", C.c_int), ("n_chan", C.c_int),
                ("n_cep", C.c_int), ("lifter", C.c_int), ("half_rate", C.c_double), ("preemph", C.c_double)]


class JdspError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("jdsp error %d: %s" % (code, text))
        self.code = code


def _load():
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7
    # (same SONAME as /opt/rocm's).  Load torch's first so that libjdsp.so binds
    # to the runtime torch's allocator and streams live in; loading them in the
    # other order leaves torch unable to see the GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "jeicyboodsp_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i, l, sz = C.c_void_p, C.c_int, C.c_long, C.c_size_t
    sig = {
        "jdsp_abi_version": (i, []),
        "jdsp_create": (i, [i, C.POINTER(vp)]),
        "jdsp_destroy": (i, [vp]),
        "jdsp_last_error": (C.c_char_p, [vp]),
        "jdsp_set_stream": (i, [vp, vp]),
        "jdsp_use_own_stream": (i, [vp]),
        "jdsp_set_option": (i, [vp, C.c_char_p, l]),
        "jdsp_synchronize": (i, [vp]),
        "jdsp_device_info": (i, [vp, C.POINTER(i), C.POINTER(sz), C.c_char_p, sz]),
        "jdsp_malloc": (i, [vp, sz, C.POINTER(vp)]),
        "jdsp_free": (i, [vp, vp]),
        "jdsp_host_alloc": (i, [vp, sz, C.POINTER(vp)]),
        "jdsp_host_free": (i, [vp, vp]),
        "jdsp_memcpy_h2d": (i, [vp, vp, vp, sz]),
        "jdsp_memcpy_d2h": (i, [vp, vp, vp, sz]),
        "jdsp_bitrev_table": (i, [vp, i, i, vp]),
        "jdsp_fft_process_f64": (i, [vp, vp, vp, i, l, i]),
        "jdsp_fft_process_f64_dev": (i, [vp, vp, vp, i, l, i]),
        "jdsp_dft_direct_f64": (i, [vp, i, vp, vp, i, l]),
        "jdsp_dft_direct_f64_dev": (i, [vp, i, vp, vp, i, l]),
        "jdsp_denoise_create": (i, [vp, i, C.POINTER(vp)]),
        "jdsp_denoise_create_cfg": (i, [vp, i, i, i, C.POINTER(vp)]),
        "jdsp_denoise_block_len": (i, [vp]),
        "jdsp_denoise_destroy": (i, [vp]),
        "jdsp_denoise_reset": (i, [vp]),
        "jdsp_denoise_set_option": (i, [vp, C.c_char_p, l]),
        "jdsp_denoise_blocks_out": (l, [vp, l]),
        "jdsp_denoise_reserve": (i, [vp, l]),
        "jdsp_denoise_process_dev": (i, [vp, vp, l, vp, vp, C.POINTER(l)]),
        "jdsp_denoise_process": (i, [vp, vp, l, vp, vp, C.POINTER(l)]),
        "jdsp_denoise_noise": (i, [vp, vp]),
        "jdsp_denoise_vad_trace": (i, [vp, l, vp, vp, vp]),
        "jdsp_gmm_create": (i, [vp, vp, i, C.POINTER(vp)]),
        "jdsp_gmm_destroy": (i, [vp]),
        "jdsp_gmm_set_option": (i, [vp, C.c_char_p, l]),
        "jdsp_gmm_score_dev": (i, [vp, vp, l, vp, l, vp, vp]),
        "jdsp_gmm_score": (i, [vp, vp, vp, l, vp, vp]),
        "jdsp_hmm_create": (i, [vp, vp, i, C.POINTER(vp)]),
        "jdsp_hmm_destroy": (i, [vp]),
        "jdsp_hmm_set_option": (i, [vp, C.c_char_p, l]),
        "jdsp_hmm_reserve": (i, [vp, l]),
        "jdsp_hmm_viterbi_dev": (i, [vp, vp, l, vp, l, vp, vp, vp, vp]),
        "jdsp_hmm_viterbi": (i, [vp, vp, vp, l, vp, vp, vp, vp]),
        "jdsp_mfcc_native_cfg": (i, [vp]),
        "jdsp_mfcc_create": (i, [vp, vp, C.POINTER(vp)]),
        "jdsp_mfcc_destroy": (i, [vp]),
        "jdsp_mfcc_tables": (i, [vp, vp, vp, vp]),
        "jdsp_mfcc_frames_dev": (i, [vp, vp, vp, l, vp]),
        "jdsp_mfcc_melfilterbank": (i, [vp, vp, l, vp]),
        "jdsp_mfcc_dct": (i, [vp, vp, l, vp]),
        "jdsp_mfcc_liftering": (i, [vp, vp, l]),
        "jdsp_gmm_probability": (i, [vp, vp, l, vp, vp, vp, vp]),
        "jdsp_mfcc_frames": (i, [vp, vp, l, vp, l, vp]),
        "jdsp_fastconv_create": (i, [vp, vp, i, i, i, C.POINTER(vp)]),
        "jdsp_fastconv_destroy": (i, [vp]),
        "jdsp_fastconv_reserve": (i, [vp, l]),
        "jdsp_fastconv_reset": (i, [vp]),
        "jdsp_fastconv_set_position": (i, [vp, l]),
        "jdsp_fastconv_block_len": (i, [vp]),
        "jdsp_fastconv_hist_blocks": (i, [vp]),
        "jdsp_fastconv_blocks_out": (l, [vp, l]),
        "jdsp_fastconv_process_dev": (i, [vp, vp, l, vp, vp, C.POINTER(l)]),
        "jdsp_fastconv_process": (i, [vp, vp, l, vp, vp, C.POINTER(l)]),
        "jdsp_vad_blocks": (i, [vp, vp, l, vp, vp, vp]),
        "jdsp_vad_blocks_ex": (i, [vp, i, i, vp, l, vp, vp, vp]),
        "jdsp_denoise_apply": (i, [vp, vp, l, vp, vp, vp, C.POINTER(l)]),
        "jdsp_pitch_autocorr_dev": (i, [vp, vp, l, vp, vp, vp, vp]),
        "jdsp_pitch_autocorr": (i, [vp, vp, l, vp, vp, vp, vp]),
        "jdsp_mvdr_create": (i, [vp, C.c_double, C.POINTER(vp)]),
        "jdsp_mvdr_destroy": (i, [vp]),
        "jdsp_mvdr_reset": (i, [vp]),
        "jdsp_mvdr_blocks_out": (l, [vp, l]),
        "jdsp_mvdr_process_dev": (i, [vp, vp, vp, l, vp, vp, C.POINTER(l)]),
        "jdsp_mvdr_process": (i, [vp, vp, vp, l, vp, vp, C.POINTER(l)]),
        "jdsp_mvdr_corr": (i, [vp, vp]),
        "jdsp_mvdr_estimate_corr": (i, [vp, vp, vp, l, vp]),
        "jdsp_mvdr_apply": (i, [vp, vp, vp, l, vp, vp, vp, C.POINTER(l)]),
        "jdsp_denoise_shard_vad_dev": (i, [vp, vp, l, l, l, l, vp]),
        "jdsp_denoise_shard_summary_dev": (i, [vp, vp, vp]),
        "jdsp_denoise_shard_rows_dev": (i, [vp, vp, i, i, vp]),
        "jdsp_denoise_shard_blocks_out": (l, [vp]),
        "jdsp_denoise_shard_finish_dev": (i, [vp, vp, i, i, vp, vp, C.POINTER(l)]),
        "jdsp_mvdr_shard_vad_dev": (i, [vp, vp, vp, l, l, l, l, vp]),
        "jdsp_mvdr_shard_summary_dev": (i, [vp, vp, vp]),
        "jdsp_mvdr_shard_blocks_out": (l, [vp]),
        "jdsp_mvdr_shard_finish_dev": (i, [vp, vp, i, i, vp, vp, C.POINTER(l)]),
        "jdsp_mvdrn_create": (i, [vp, i, vp, C.c_double, C.POINTER(vp)]),
        "jdsp_mvdrn_create_cfg": (i, [vp, i, vp, C.c_double, i, C.POINTER(vp)]),
        "jdsp_mvdrn_block_len": (i, [vp]),
        "jdsp_mvdrn_destroy": (i, [vp]),
        "jdsp_mvdrn_reset": (i, [vp]),
        "jdsp_mvdrn_blocks_out": (l, [vp, l]),
        "jdsp_mvdrn_process_dev": (i, [vp, vp, l, l, vp, vp, C.POINTER(l)]),
        "jdsp_mvdrn_process": (i, [vp, vp, l, l, vp, vp, C.POINTER(l)]),
        "jdsp_stft_half_i16_dev": (i, [vp, vp, l, vp, l]),
        "jdsp_stft_i16_dev": (i, [vp, vp, l, i, i, vp]),
        "jdsp_stft_i16": (i, [vp, vp, l, i, i, vp, C.POINTER(l)]),
        "jdsp_stft_i16_f64_dev": (i, [vp, vp, l, i, i, vp]),
        "jdsp_stft_i16_f64": (i, [vp, vp, l, i, i, vp, C.POINTER(l)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()
