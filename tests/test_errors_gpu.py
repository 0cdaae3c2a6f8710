"""GPU-box tests of the C ABI's error behaviour: bad arguments come back as negative codes
with a message, never as a crash or a silent fallback."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import jeicyboodsp_amd
    e = jeicyboodsp_amd.Engine(0)
    yield e
    e.close()


def test_bad_sizes_and_alignment(eng):
    import torch
    from jeicyboodsp_amd import JdspError, _lib
    L = _lib.lib
    pcm = torch.zeros(4096, dtype=torch.int16, device="cuda")
    spec = torch.empty((4, 1024), dtype=torch.complex64, device="cuda")
    h = eng._h
    assert L.jdsp_stft_i16_dev(h, C.c_void_p(pcm.data_ptr()), 4, 2048, 512, C.c_void_p(spec.data_ptr())) == _lib.EINVAL
    assert b"n_fft" in L.jdsp_last_error(h)
    assert L.jdsp_stft_i16_dev(h, C.c_void_p(pcm.data_ptr()), 4, 1024, 0, C.c_void_p(spec.data_ptr())) == _lib.EINVAL
    assert L.jdsp_stft_i16_dev(h, None, 4, 1024, 512, C.c_void_p(spec.data_ptr())) == _lib.EINVAL
    assert L.jdsp_stft_i16_dev(h, C.c_void_p(pcm.data_ptr()), 4, 1024, 512, C.c_void_p(spec.data_ptr() + 8)) == _lib.EINVAL
    assert L.jdsp_stft_i16_dev(h, C.c_void_p(pcm.data_ptr()), 0, 1024, 512, None) == 0          # nothing to do
    assert L.jdsp_set_option(h, b"no.such.option", 1) == _lib.EINVAL
    with pytest.raises(JdspError):
        eng.fft_process(np.zeros((2, 16384), np.complex128))        # > 8192
    with pytest.raises(JdspError):
        eng.denoiser(7)
    with pytest.raises(JdspError):
        eng.fastconv(np.ones(2000), 1024)                           # more taps than the transform
    with pytest.raises(JdspError):
        eng.fastconv(np.ones(10), 4096)                             # unsupported transform size
    with pytest.raises(JdspError):
        eng.mfcc(n_fft=256)
    with pytest.raises(JdspError):
        eng.mfcc(win_len=2000)
    d = eng.denoiser(0)
    bad = torch.zeros(512 * 3 + 8, dtype=torch.int16, device="cuda")[4:4 + 512 * 3]   # 8 bytes off
    out = torch.empty(512, dtype=torch.int16, device="cuda")
    assert L.jdsp_denoise_process_dev(d._h, C.c_void_p(bad.data_ptr()), 3, C.c_void_p(out.data_ptr()), None, None) == _lib.EINVAL
    assert L.jdsp_denoise_process_dev(d._h, C.c_void_p(pcm.data_ptr()), -1, None, None, None) == _lib.EINVAL
    assert L.jdsp_denoise_shard_vad_dev(d._h, C.c_void_p(pcm.data_ptr()), 5, 4, 8, 8, C.c_void_p(out.data_ptr())) == _lib.EINVAL
    d.close()


def test_zero_length_calls_are_ok(eng):
    d = eng.denoiser(1)
    assert d.process(np.zeros(0, np.int16)).size == 0
    assert d.process(np.zeros(512, np.int16)).size == 0              # first block of a stream: nothing out
    fc = eng.fastconv(np.ones(8), 1024)
    assert fc.process(np.zeros(0, np.int16)).shape == (1, 0)
    m = eng.mfcc()
    assert m.frames(np.zeros(100, np.int16)).shape == (0, 12)
    arg, rmax = eng.pitch(np.zeros(0, np.int16))
    assert arg.size == 0
    for o in (d, fc, m):
        o.close()


def test_two_handles_are_independent(eng):
    """One handle per audio stream: interleaving two streams must not mix their state."""
    rng = np.random.default_rng(0)
    a = np.clip(np.rint(rng.normal(0, 3000, 20 * 512)), -32768, 32767).astype(np.int16)
    b = np.clip(np.rint(rng.normal(0, 800, 20 * 512)), -32768, 32767).astype(np.int16)
    da, db, ref_a, ref_b = eng.denoiser(1), eng.denoiser(1), eng.denoiser(1), eng.denoiser(1)
    outs_a, outs_b = [], []
    for k in range(0, 20, 5):
        outs_a.append(da.process(a[k * 512:(k + 5) * 512]))
        outs_b.append(db.process(b[k * 512:(k + 5) * 512]))
    assert np.array_equal(np.concatenate(outs_a), ref_a.process(a))
    assert np.array_equal(np.concatenate(outs_b), ref_b.process(b))
    for o in (da, db, ref_a, ref_b):
        o.close()


def test_vad_trace_follows_the_option_at_the_time_of_the_call(eng):
    """jdsp_denoise_vad_trace hands out energies / zero-crossing counts only if "vad_trace" was set when the traced
    call RAN (ABI version 2): setting it afterwards must not expose buffers that call never filled."""
    import jeicyboodsp_amd
    rng = np.random.default_rng(3)
    pcm = np.clip(np.rint(rng.normal(0, 3000, 8 * 512)), -32768, 32767).astype(np.int16)
    d = eng.denoiser(0)
    d.process(pcm)                                                    # option off: flags only
    assert d.vad_trace(8, flags_only=True).shape == (8,)
    d.set_option("vad_trace", 1)                                      # ... switched on AFTER the call
    with pytest.raises(jeicyboodsp_amd.JdspError):
        d.vad_trace(8)
    d.process(pcm)                                                    # a call that ran with it on
    v, e, z = d.vad_trace(8)
    assert v.shape == e.shape == z.shape == (8,) and (e > 0).all()
    d.set_option("vad_trace", 0)
    v2, e2, z2 = d.vad_trace(8)                                       # still that call's trace
    assert np.array_equal(e2, e)
    d.close()
    assert jeicyboodsp_amd._lib.lib.jdsp_abi_version() == 2


def test_stft_options_take_documented_values_only(eng):
    for name, bad in (("stft.read_pass", 2), ("stft.f64_kernel", 2), ("stft.f64_frames_per_wave", -1)):
        with pytest.raises(Exception):
            eng.set_option(name, bad)
    eng.set_option("stft.f64_kernel", 1)
    eng.set_option("stft.f64_kernel", 0)
    eng.set_option("stft.f64_frames_per_wave", 0)
