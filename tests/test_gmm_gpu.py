"""GPU parity: GMM scoring (GMMAlgorithm_Test_Auto_ver2.cpp) and the HMM recursion (Viterbi_version1.cpp) on
device-resident MFCC vectors against the CPU oracle, through the C ABI.  FP64 on both sides; the only
differences allowed are the last-place ones of exp/log and of the per-utterance sum's association:
1e-12 relative on every finite value, identical NaN / -inf patterns, identical arg-max indices and states."""
import numpy as np
import pytest

import gmm_cases as gc

pytestmark = pytest.mark.gpu

RTOL = 1e-12


@pytest.fixture(scope="module")
def eng():
    import jeicyboodsp_amd
    e = jeicyboodsp_amd.Engine(0)
    yield e
    e.close()


def _same(got, want):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want))
    inf = np.isinf(want)
    assert np.array_equal(got[inf], want[inf])
    fin = np.isfinite(want)
    assert np.all(np.abs(got[fin] - want[fin]) <= RTOL * np.abs(want[fin]) + 1e-300)


LENGTHS = [1, 2, 63, 64, 65, 300, 7, 128, 1, 511]


def test_gmm_scores_and_argmax_over_a_ragged_batch(eng, oracle):
    import torch
    classes = gc.gmm_records(11, 25)                                  # NUM_OF_CLASS 25 (GMMTest:26)
    first = gc.offsets(LENGTHS)
    x = gc.vectors(12, int(first[-1]))
    g = eng.gmm(classes)
    scores, best = g.score(torch.from_numpy(x).cuda(), torch.from_numpy(first).cuda())
    scores, best = scores.cpu().numpy(), best.cpu().numpy()
    for u in range(len(LENGTHS)):
        want, arg = oracle.gmm_classify(x[first[u]:first[u + 1]], classes)
        _same(scores[u], want)
        assert best[u] == arg
    hs, hb = g.score(x, first)                                        # host entry = device entry
    assert np.array_equal(hs, scores) and np.array_equal(hb, best)
    g.close()


def test_gmm_fused_evaluation_option(eng, oracle):
    """jdsp_gmm_set_option("evaluation", 1): same densities with FMA projections and one exp per mixture;
    held to 1e-11 relative (a few 1e-16 per operation times the size of the exponent) and the same classes."""
    import jeicyboodsp_amd
    classes = gc.gmm_records(31, 25)
    first = gc.offsets(LENGTHS)
    x = gc.vectors(32, int(first[-1]))
    g = eng.gmm(classes)
    ref_scores, ref_best = g.score(x, first)
    g.set_option("evaluation", 1)
    scores, best = g.score(x, first)
    for u in range(len(LENGTHS)):
        want, arg = oracle.gmm_classify(x[first[u]:first[u + 1]], classes)
        assert np.all(np.abs(scores[u] - want) <= 1e-11 * np.abs(want))
        assert best[u] == arg
    assert not np.array_equal(scores, ref_scores)                    # it really is a different evaluation
    assert np.array_equal(best, ref_best)
    far = np.full((4, 12), 1e6)                                       # total underflow: -inf either way
    s, b = g.score(far, np.array([0, 4], np.int64))
    assert np.all(np.isneginf(s)) and b[0] == 0
    g.set_option("evaluation", 0)
    again, _ = g.score(x, first)
    assert np.array_equal(again, ref_scores)
    with pytest.raises(jeicyboodsp_amd.JdspError):
        g.set_option("evaluation", 2)
    with pytest.raises(jeicyboodsp_amd.JdspError):
        g.set_option("no_such_option", 1)
    g.close()


@pytest.mark.parametrize("n_classes", [1, 3, 4, 5, 64, 256])
def test_gmm_class_counts(eng, oracle, n_classes):
    classes = gc.gmm_records(20 + n_classes, n_classes)
    first = gc.offsets([40, 3])
    x = gc.vectors(13, 43)
    g = eng.gmm(classes)
    scores, best = g.score(x, first)
    for u in range(2):
        want, arg = oracle.gmm_classify(x[first[u]:first[u + 1]], classes)
        _same(scores[u], want)
        assert best[u] == arg
    g.close()


def test_gmm_ties_underflow_and_empty_utterances(eng, oracle):
    classes = gc.gmm_records(14, 6)
    classes[4] = classes[1]                                           # exact tie: the earlier class must win if maximal
    x = np.concatenate([gc.vectors(15, 10), np.full((5, 12), 1e6), gc.vectors(16, 3)])
    first = np.array([0, 10, 15, 15, 18], np.int64)                   # [normal, all densities 0, empty, normal]
    g = eng.gmm(classes)
    scores, best = g.score(x, first)
    for u in range(4):
        want, arg = oracle.gmm_classify(x[first[u]:first[u + 1]], classes)
        _same(scores[u], want)
        assert best[u] == arg
    assert np.all(np.isneginf(scores[1])) and np.all(np.isnan(scores[2])) and best[1] == 0 and best[2] == 0
    assert scores[0][1] == scores[0][4]
    g.close()


def test_mfcc_to_gmm_without_leaving_the_device(eng, oracle):
    """PCM -> jdsp_mfcc_frames_dev -> jdsp_gmm_score_dev; the scorer is checked on the vectors the MFCC
    kernel produced (its own parity is tests/test_mfcc_gpu.py's business)."""
    import torch
    rng = np.random.default_rng(17)
    pcm = np.clip(np.rint(rng.normal(0, 3000, 512 * 41)), -32768, 32767).astype(np.int16)
    m = eng.mfcc()
    feats = m.frames(torch.from_numpy(pcm).cuda())                    # [40, 12] float64 on the device
    assert feats.shape == (40, 12)
    lens = [15, 25]
    first = gc.offsets(lens)
    classes = gc.gmm_records(18, 25)
    classes["mean"] *= 4.0                                            # MFCC-sized means
    classes["covariance"] *= 30.0
    g = eng.gmm(classes)
    scores, best = g.score(feats, torch.from_numpy(first).cuda())
    f = feats.cpu().numpy()
    for u in range(2):
        want, arg = oracle.gmm_classify(f[first[u]:first[u + 1]], classes)
        _same(scores[u].cpu().numpy(), want)
        assert int(best[u]) == arg
    g.close()
    m.close()


def _check_hmm(eng, oracle, models, x, first):
    import torch
    h = eng.hmm(models)
    scores, best, path, trellis = h.viterbi(torch.from_numpy(x).cuda(), torch.from_numpy(first).cuda(),
                                            want_trellis=True)
    scores, best, path, trellis = (t.cpu().numpy() for t in (scores, best, path, trellis))
    for u in range(len(first) - 1):
        a, b = int(first[u]), int(first[u + 1])
        rets = []
        for mdl in range(len(models)):
            ret, p, tr = oracle.hmm_viterbi(x[a:b], models[mdl])
            _same(trellis[mdl][:, a:b], tr)
            assert np.array_equal(path[mdl][a:b], p)
            _same(scores[u, mdl], ret)
            rets.append(ret)
        arg, mx = 0, rets[0]
        for mdl in range(1, len(models)):                             # Viterbi:119-126
            if mx < rets[mdl]:
                mx, arg = rets[mdl], mdl
        assert best[u] == arg
    hs, hb, hp = h.viterbi(x, first)                                  # host entry = device entry
    assert np.array_equal(hs, scores, equal_nan=True) and np.array_equal(hb, best) and np.array_equal(hp, path)
    h.close()
    return scores, path


def test_hmm_recursion_finite_regime(eng, oracle):
    models = gc.hmm_records_finite(21, 3)
    lens = [20, 1, 2, 65, 7]
    first = gc.offsets(lens)
    x = np.concatenate([gc.vectors_near(30 + i, n, models["gMMParam"][i % 3, (2 * i) % 6]) for i, n in enumerate(lens)])
    scores, path = _check_hmm(eng, oracle, models, x, first)
    assert np.isfinite(scores[0]).any() and path.any()
    assert np.all(scores[1] == 0.0)                                   # one vector: dTempProb keeps its initial 0


def test_hmm_fused_evaluation_option(eng, oracle):
    models = gc.hmm_records_finite(41, 2)
    lens = [9, 33]
    first = gc.offsets(lens)
    x = np.concatenate([gc.vectors_near(50 + i, n, models["gMMParam"][i % 2, i % 6]) for i, n in enumerate(lens)])
    h = eng.hmm(models)
    ref = h.viterbi(x, first, want_trellis=True)
    h.set_option("evaluation", 1)
    got = h.viterbi(x, first, want_trellis=True)
    fin = np.isfinite(ref[3])
    assert fin.any() and np.array_equal(np.isfinite(got[3]), fin)
    assert np.all(np.abs(got[3][fin] - ref[3][fin]) <= 1e-10 * np.abs(ref[3][fin]))       # trellis
    assert np.array_equal(got[2], ref[2]) and np.array_equal(got[1], ref[1])                # states, best model
    h.close()


def test_hmm_recursion_nan_regime(eng, oracle):
    """Ordinary parameters: the accumulated log probability is negative and Viterbi_version1.cpp:196's
    log() of it makes the trellis NaN from the second vector on; the device path must do the same."""
    models = gc.hmm_records(22, 1)                                    # NUM_OF_CLASS 1 (Viterbi:26)
    first = gc.offsets([12, 30])
    x = gc.vectors(23, 42)
    scores, path = _check_hmm(eng, oracle, models, x, first)
    assert np.all(np.isnan(scores)) and not path.any()


def test_offsets_past_the_end_are_clamped_on_the_device(eng, oracle):
    import torch
    classes = gc.gmm_records(24, 3)
    x = gc.vectors(25, 50)
    g = eng.gmm(classes)
    first = torch.tensor([0, 20, 10_000_000], dtype=torch.int64).cuda()   # second utterance claims too much
    scores, best = g.score(torch.from_numpy(x).cuda(), first)
    want, arg = oracle.gmm_classify(x[20:50], classes)
    _same(scores[1].cpu().numpy(), want)
    assert int(best[1]) == arg
    g.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_utterance_sharded_scoring_equals_the_single_gpu_batch(eng, world):
    """§8(e) for rank 4: whole utterances per rank, no collective; simulated ranks on one GPU."""
    from jeicyboodsp_amd import sharding
    lens = [5, 300, 1, 64, 77, 0, 130, 9, 256, 31, 2]
    first = gc.offsets(lens)
    x = gc.vectors(26, int(first[-1]))
    classes = gc.gmm_records(27, 25)
    models = gc.hmm_records_finite(28, 2)
    g, h = eng.gmm(classes), eng.hmm(models)
    want_s, want_b = g.score(x, first)
    want_hs, want_hb, want_hp = h.viterbi(x, first)
    rows_s, rows_b, rows_hs, rows_hb, cols_hp = [], [], [], [], []
    for r in range(world):
        u0, n, lo, hi, local = sharding.utterance_batch_shard(first, r, world)
        s, b = g.score(x[lo:hi], np.asarray(local, np.int64))
        hs, hb, hp = h.viterbi(x[lo:hi], np.asarray(local, np.int64))
        rows_s.append(s); rows_b.append(b); rows_hs.append(hs); rows_hb.append(hb); cols_hp.append(hp)
    assert np.array_equal(np.concatenate(rows_s), want_s, equal_nan=True)
    assert np.array_equal(np.concatenate(rows_b), want_b)
    assert np.array_equal(np.concatenate(rows_hs), want_hs, equal_nan=True)
    assert np.array_equal(np.concatenate(rows_hb), want_hb)
    assert np.array_equal(np.concatenate(cols_hp, axis=1), want_hp)
    g.close()
    h.close()


def test_gmm_hmm_errors(eng):
    import jeicyboodsp_amd
    with pytest.raises(jeicyboodsp_amd.JdspError):
        eng.gmm(gc.gmm_records(1, 257))
    with pytest.raises(jeicyboodsp_amd.JdspError):
        eng.gmm(np.zeros(0, jeicyboodsp_amd.GMM_PARAM))
    g = eng.gmm(gc.gmm_records(1, 2))
    with pytest.raises(jeicyboodsp_amd.JdspError):
        g.score(gc.vectors(1, 10), np.array([0, 7, 5], np.int64))     # decreasing offsets
    s, b = g.score(np.zeros((0, 12)), np.array([0], np.int64))        # no utterances
    assert s.shape == (0, 2) and b.shape == (0,)
    g.close()
    h = eng.hmm(gc.hmm_records(2, 1))
    with pytest.raises(jeicyboodsp_amd.JdspError):
        h.viterbi(gc.vectors(1, 10), np.array([2, 10], np.int64))     # host table must start at 0
    h.close()


def test_probability_on_its_own_matches_the_oracle(eng, oracle):
    """probability() (GMMAlgorithm_Test_Auto_ver2.cpp:43,:164-236) as a separately callable entry: every vector
    under every mixture component of a record, against the oracle's restatement (1e-12: exp's last place)."""
    import gmm_cases
    rec = gmm_cases.gmm_records(5, 1)[0]
    vec = gmm_cases.vectors(6, 200)
    for k in range(4):
        got = eng.gmm_probability(vec, rec["mean"][k], rec["covariance"][k], rec["eigenVector"][k])
        want = np.array([oracle.gmm_probability(v, rec["mean"][k], rec["covariance"][k], rec["eigenVector"][k]) for v in vec])
        assert np.array_equal(np.isfinite(got), np.isfinite(want))
        ok = np.isfinite(want) & (want != 0)
        assert np.abs(got[ok] - want[ok]).max() <= 1e-12 * np.abs(want[ok]).max()
        assert (np.abs(got[ok] / want[ok] - 1) < 1e-10).all()
