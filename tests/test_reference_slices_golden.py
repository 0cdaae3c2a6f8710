"""The FFTW-free reference functions, pinned to the reference itself.

Five functions of the application programs never call FFTW and compile from their own line ranges
(oracle/Makefile: libref_mfcc_tail.so, libref_vad_{ss,wf,bf}.so -- no stand-in header, no copied source):

  MelFilterBankInit / MelFilterBank / DCT / Liftering   MFCCFeatureExtraction_auto_version1.cpp:116-192
  VoiceActivityDetection                                SpectralSubtraction_final.cpp:121-156,
                                                        WienerFilter_final.cpp:261-296,
                                                        BeamForming_MVDR_ver1.cpp:207-242

tests/golden/mfcc_tail.npz and vad.npz hold seeded inputs and what those compiled functions returned
(tests/golden/make_golden.py).  Compared here:

  CPU (not gpu): the oracle's restatements vs the fixtures -- tables, flags, energy sums, zero-crossing counts
                 exact; MelFilterBank / DCT / Liftering <= 1e-12 relative (observed: bit-identical), with the
                 same -inf / NaN pattern for empty channels.
  GPU (gpu)    : the C ABI's entries vs the fixtures -- jdsp_mfcc_tables exact; jdsp_mfcc_melfilterbank / _dct /
                 _liftering (FP64 entries) <= 1e-12 relative; jdsp_vad_blocks / jdsp_vad_blocks_ex flags, energy
                 sums and zero-crossing counts exact; the FP32 production chain (jdsp_mfcc_frames) is held to
                 north_star's 1e-5 against the same tail fed with its own spectra elsewhere (test_mfcc_gpu.py).

The VAD reads one short past its frame (SS:139).  The fixture records the outcome with that slot painted 0 (the
value the oracle and the kernels define), +1 and -1: it can only matter when the frame's last windowed sample is
non-zero, and it moves the count by at most one.
"""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def tail(golden_dir):
    return np.load(os.path.join(golden_dir, "mfcc_tail.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def vad(golden_dir):
    return np.load(os.path.join(golden_dir, "vad.npz"), allow_pickle=False)


def _same_nonfinite(got, want):
    return (np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.isposinf(got), np.isposinf(want))
            and np.array_equal(np.isneginf(got), np.isneginf(want)))


def _rel(got, want):
    """max |got - want| / row peak over the finite entries (rows without a finite entry contribute 0)"""
    fin = np.isfinite(want)
    d = np.where(fin, np.abs(np.where(fin, got, 0.0) - np.where(fin, want, 0.0)), 0.0)
    peak = np.where(fin, np.abs(want), 0.0).max(axis=1, keepdims=True)
    return (d / np.maximum(peak, 1e-300)).max()


# ------------------------------------------------------------------------------------------ CPU: oracle vs fixture
def test_fixture_is_the_native_configuration(tail, oracle):
    cfg = oracle.mfcc_native_cfg()
    n_cep, n_bins, n_chan, lifter, block_len = (int(v) for v in tail["consts"])
    assert (cfg.n_cep, cfg.n_bins, cfg.n_chan, cfg.lifter, cfg.win_len) == (n_cep, n_bins, n_chan, lifter, block_len)
    assert cfg.half_rate == float(tail["half_rate"])


def test_oracle_mel_tables_equal_the_reference_tables(tail, oracle):
    mel, fi, fb = oracle.mel_init(oracle.mfcc_native_cfg())
    assert np.array_equal(fi, tail["fi_bins"])
    assert np.array_equal(mel.view(np.uint64), tail["mel_freqs"].view(np.uint64))
    assert np.array_equal(fb.view(np.uint64), tail["filter_bank"].view(np.uint64))


def test_oracle_melfilterbank_dct_liftering_match_the_reference(tail, oracle):
    cfg = oracle.mfcc_native_cfg()
    with np.errstate(all="ignore"):
        mel = oracle.mel_filterbank(cfg, tail["mag"])
        assert _same_nonfinite(mel, tail["mel"]) and _rel(mel, tail["mel"]) <= 1e-12
        assert np.isneginf(tail["mel"][4, 7]) and np.isneginf(tail["mel"][5]).all()      # the empty-channel rows are there
        cep = oracle.dct(cfg, tail["mel"])
        assert _same_nonfinite(cep, tail["cep"]) and _rel(cep, tail["cep"]) <= 1e-12
        acc = oracle.dct(cfg, tail["mel"], accumulate_into=tail["cep_acc_in"])
        assert _same_nonfinite(acc, tail["cep_acc"]) and _rel(acc, tail["cep_acc"]) <= 1e-12
        lift = oracle.liftering(cfg, tail["cep"])
        assert _same_nonfinite(lift, tail["liftered"]) and _rel(lift, tail["liftered"]) <= 1e-12


@pytest.mark.parametrize("which", ["ss", "wf"])
def test_oracle_vad_equals_the_reference_vad(vad, oracle, which):
    blocks = vad["blocks"]
    thr_e, thr_z, keep, block_len, n_fft, pi = vad[which + "_consts"]
    assert (thr_e, thr_z, keep, block_len, n_fft, pi) == (700.0, 200.0, 512.0, 512.0, 1024.0, 3.141592)
    got = [oracle.vad_block(b) for b in blocks]
    flags = np.array([g[0] for g in got], np.uint8)
    esum = np.rint(np.array([g[1] for g in got]) * n_fft).astype(np.int64)
    zcr = np.array([g[2] for g in got], np.int32)
    assert np.array_equal(flags, vad[which + "_flags"])
    assert np.array_equal(esum, vad[which + "_energy_sum"])
    assert np.array_equal(zcr, vad[which + "_zcr"])
    # both thresholds are exercised from both sides
    e = vad[which + "_energy_printed"]
    assert (e > 700).any() and ((e < 700) & (e > 600)).any()
    assert (zcr == 199).any() or ((zcr < 200).any() and (zcr >= 200).any())


def test_the_overread_slot_moves_the_count_by_at_most_one(vad):
    """What SS:139 finds past the frame is the caller's: record how far it reaches."""
    for which in ("ss", "wf", "bf"):
        z0 = vad[which + "_zcr"].astype(np.int64)
        for tag in ("_fill_pos", "_fill_neg"):
            d = vad[which + "_zcr" + tag].astype(np.int64) - z0
            assert d.min() >= 0 and d.max() <= 1
            assert np.array_equal(vad[which + "_flags" + tag], vad[which + "_flags"])   # no decision hangs on it here
        if which == "bf":                     # frame[1023] = 0 in the beamformer's frame: the slot never matters
            assert np.array_equal(vad["bf_zcr_fill_pos"], z0) and np.array_equal(vad["bf_zcr_fill_neg"], z0)


def test_oracle_mvdr_vad_equals_the_reference_vad(vad, oracle):
    blocks = vad["blocks"]
    thr_e, _, keep, block_len, n_fft, pi = vad["bf_consts"]
    assert (thr_e, keep, block_len, n_fft, pi) == (700.0, 511.0, 512.0, 1024.0, 3.141592)
    got = [oracle.mvdr_vad_block(b) for b in blocks]
    assert np.array_equal(np.array([g[0] for g in got], np.uint8), vad["bf_flags"])
    assert np.array_equal(np.rint(np.array([g[1] for g in got]) * n_fft).astype(np.int64), vad["bf_energy_sum"])
    assert np.array_equal(np.array([g[2] for g in got], np.int32), vad["bf_zcr"])
    # the frame offset matters: the same blocks through the denoisers' VAD give other energies
    assert not np.array_equal(vad["bf_energy_sum"], vad["ss_energy_sum"])


def test_live_reference_slices_reproduce_the_fixtures(tail, vad):
    """Authoring container only: the fixtures are what the compiled reference slices give now."""
    import oracle_lib
    ref = oracle_lib.load_ref_mfcc_tail()
    if ref is None or oracle_lib.load_ref_vad("ss") is None:
        pytest.skip("oracle/_ref not built (reference checkout absent)")
    mel, fi, fb = ref.tables()
    assert np.array_equal(fi, tail["fi_bins"]) and np.array_equal(fb, tail["filter_bank"]) and np.array_equal(mel, tail["mel_freqs"])
    with np.errstate(all="ignore"):
        assert np.array_equal(ref.mel_filterbank(tail["mag"]), tail["mel"], equal_nan=True)
        assert np.array_equal(ref.dct(tail["mel"]), tail["cep"], equal_nan=True)
        assert np.array_equal(ref.liftering(tail["cep"]), tail["liftered"], equal_nan=True)
    for which in ("ss", "wf", "bf"):
        flags, energy, zcr = oracle_lib.load_ref_vad(which).run(vad["blocks"], 0)
        assert np.array_equal(flags.astype(np.uint8), vad[which + "_flags"])
        assert np.array_equal(energy, vad[which + "_energy_printed"]) and np.array_equal(zcr, vad[which + "_zcr"])


# ------------------------------------------------------------------------------------------ GPU: C ABI vs fixture
@pytest.fixture(scope="module")
def eng():
    import jeicyboodsp_amd
    e = jeicyboodsp_amd.Engine(0)
    yield e
    e.close()


@pytest.mark.gpu
def test_gpu_mfcc_tables_equal_the_reference_tables(eng, tail):
    m = eng.mfcc()
    mel, fi, fb = m.tables()
    assert np.array_equal(fi, tail["fi_bins"])
    assert np.array_equal(mel.view(np.uint64), tail["mel_freqs"].view(np.uint64))
    assert np.array_equal(fb.view(np.uint64), tail["filter_bank"].view(np.uint64))
    m.close()


@pytest.mark.gpu
def test_gpu_melfilterbank_dct_liftering_match_the_reference(eng, tail):
    m = eng.mfcc()
    with np.errstate(all="ignore"):
        mel = m.mel_filterbank(tail["mag"])
        assert _same_nonfinite(mel, tail["mel"]) and _rel(mel, tail["mel"]) <= 1e-12
        cep = m.dct(tail["mel"])
        assert _same_nonfinite(cep, tail["cep"]) and _rel(cep, tail["cep"]) <= 1e-12
        acc = m.dct(tail["mel"], accumulate_into=tail["cep_acc_in"])
        assert _same_nonfinite(acc, tail["cep_acc"]) and _rel(acc, tail["cep_acc"]) <= 1e-12
        lift = m.liftering(tail["cep"])
        assert _same_nonfinite(lift, tail["liftered"]) and _rel(lift, tail["liftered"]) <= 1e-12
        # the whole tail chained on the device entries, from |X| to the liftered vector
        chain = m.liftering(m.dct(m.mel_filterbank(tail["mag"])))
        assert _same_nonfinite(chain, tail["liftered"]) and _rel(chain, tail["liftered"]) <= 1e-11
    m.close()


@pytest.mark.gpu
def test_gpu_production_mfcc_tail_within_1e5_of_the_reference_tail(eng, tail, oracle):
    """jdsp_mfcc_frames (FP32 transform, FP32 filterbank, FP64 DCT) against the REFERENCE's tail fed with an FP64 |X|
    of the same frames (numpy's DFT where the reference calls FFTW): whatever the kernel's own tail does differently
    shows here."""
    import oracle_lib
    rng = np.random.default_rng(5)
    pcm = np.clip(np.rint(rng.normal(0, 2500, 1024 * 6)), -32768, 32767).astype(np.int16)
    m = eng.mfcc()
    got = m.frames(pcm)                                   # hop 512: 11 frames
    cfg = oracle.mfcc_native_cfg()
    raw = np.stack([pcm[512 * j:512 * j + 1024].astype(np.float64) for j in range(got.shape[0])])
    frames = np.zeros_like(raw)
    frames[:, 1:] = raw[:, 1:] - 0.96 * raw[:, :-1]       # pre-emphasis, sample 0 left at 0 (MFCC:207-209)
    frames *= oracle.hamming(1024)                        # MFCC:211-213
    mag = np.abs(np.fft.fft(frames, axis=1))[:, :512]     # MFCC:215-219 (FFTW's contract: the unnormalised DFT)
    ref = oracle_lib.load_ref_mfcc_tail()
    if ref is not None:                                   # authoring container: straight through the compiled reference
        want = ref.liftering(ref.dct(ref.mel_filterbank(mag)))
    else:                                                 # GPU box: the oracle's tail, which the CPU tests hold bit-equal to it
        want = oracle.liftering(cfg, oracle.dct(cfg, oracle.mel_filterbank(cfg, mag)))
    assert (np.abs(got - want) / np.abs(want).max(axis=1, keepdims=True)).max() < 1e-5
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["ss", "wf", "bf"])
def test_gpu_vad_equals_the_reference_vad(eng, vad, which):
    v, e, z = eng.vad_blocks(vad["blocks"], "mvdr" if which == "bf" else "denoise")
    assert np.array_equal(v, vad[which + "_flags"])
    assert np.array_equal(e, vad[which + "_energy_sum"])
    assert np.array_equal(z, vad[which + "_zcr"])


@pytest.mark.gpu
def test_gpu_stream_vad_flags_equal_the_reference_flags(eng, vad):
    """The flags-only kernel the batched denoise / MVDR chains run, through their handles."""
    blocks = vad["blocks"]
    pcm = blocks.reshape(-1)
    d = eng.denoiser(0)
    d.process(pcm)
    assert np.array_equal(d.vad_trace(blocks.shape[0], flags_only=True), vad["ss_flags"])
    d.close()
    import torch
    mv = eng.mvdr()
    t = torch.from_numpy(pcm.copy()).cuda()
    n = blocks.shape[0]
    flags = mv.shard_vad(t, t, 0, 0, n, n)                # the chain's own VAD launch over blocks [0, n)
    torch.cuda.synchronize()
    assert np.array_equal(flags.cpu().numpy(), vad["bf_flags"])
    mv.close()
