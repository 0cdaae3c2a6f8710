"""CPU: the oracle's GMM / HMM restatement against closed forms written independently in
numpy, and the reference's documented behaviours.  The reference holds no fixture for these programs and they
need Eigen to compile (absent here): parity unpinned at the Eigen boundary, see oracle/jdsp_oracle.h."""
import numpy as np

import gmm_cases as gc

PI = 3.141592


def _prob(x, g, k):
    y = x @ g["eigenVector"][k]
    c = np.diag(g["covariance"][k])[:4]
    d = y - g["mean"][k][:4]
    return np.prod((1.0 / np.sqrt(2.0 * PI)) * (1.0 / np.sqrt(c)) * np.exp(-0.5 * d * d / c))


def test_probability_matches_the_closed_form(oracle):
    g = gc.gmm_records(1, 3)
    x = gc.vectors(2, 20)
    for r in g:
        for k in range(4):
            for v in x:
                got = oracle.gmm_probability(v, r["mean"][k], r["covariance"][k], r["eigenVector"][k])
                want = _prob(v, r, k)
                assert abs(got - want) <= 1e-13 * abs(want) + 1e-300


def test_recognition_and_class_argmax(oracle):
    g = gc.gmm_records(3, 25)
    x = gc.vectors(4, 57)
    scores, arg = oracle.gmm_classify(x, g)
    want = np.array([np.mean([np.log(sum(r["alpa"][k] * _prob(v, r, k) for k in range(4))) for v in x]) for r in g])
    assert np.allclose(scores, want, rtol=1e-12, atol=0)
    assert arg == int(np.argmax(want))


def test_class_argmax_keeps_the_first_maximum_and_ignores_nan(oracle):
    g = gc.gmm_records(5, 4)
    g[2] = g[0]                                                # an exact tie: the earlier class stays
    x = gc.vectors(6, 9)
    scores, arg = oracle.gmm_classify(x, g)
    assert scores[0] == scores[2] and arg == int(np.argmax(scores))
    far = np.full((3, 12), 1e6)                                # every density underflows: log(0) = -inf everywhere
    scores, arg = oracle.gmm_classify(far, g)
    assert np.all(np.isneginf(scores)) and arg == 0
    scores, arg = oracle.gmm_classify(np.zeros((0, 12)), g)    # 0/0 (GMMTest:161)
    assert np.all(np.isnan(scores)) and arg == 0


def test_hmm_recursion_in_the_regime_where_the_reference_is_finite(oracle):
    h = gc.hmm_records_finite(7, 1)[0]
    x = np.concatenate([gc.vectors_near(8 + s, 5, h["gMMParam"][s]) for s in (0, 3, 5, 1)])
    ret, path, tr = oracle.hmm_viterbi(x, h)
    n = len(x)
    b = np.array([[sum(h["gMMParam"][m]["alpa"][k] * _prob(v, h["gMMParam"][m], k) for k in range(4))
                   for m in range(6)] for v in x])
    with np.errstate(all="ignore"):
        want = np.zeros((6, n))
        want[:, 0] = np.log(b[0]) + np.log(1.0 / 6)
        for i in range(1, n):
            for m in range(6):
                cand = np.log(want[:, i - 1]) + np.log(h["transProb"][:, m]) + np.log(b[i, m])
                want[m, i] = cand[0]
                for u in range(1, 6):
                    if want[m, i] < cand[u]:
                        want[m, i] = cand[u]
    fin = np.isfinite(want)
    assert fin.all() and np.isfinite(tr).all() and len(set(path[1:].tolist())) > 1
    assert np.allclose(tr[fin], want[fin], rtol=1e-12)
    assert path[0] == 0 and ret == tr[:, 1].max()
    for i in range(1, n):
        col = tr[:, i]
        assert path[i] == (0 if np.isnan(col[0]) else int(np.argmax(np.where(np.isnan(col), -np.inf, col))))


def test_hmm_recursion_goes_nan_when_the_accumulated_log_probability_is_negative(oracle):
    """Viterbi_version1.cpp:196 takes log() of the previous column, which is itself a log probability."""
    h = gc.hmm_records(9, 1)[0]
    x = gc.vectors(10, 12)
    ret, path, tr = oracle.hmm_viterbi(x, h)
    assert np.all(tr[:, 0] < 0) and np.all(np.isnan(tr[:, 1:]))
    assert np.isnan(ret) and not path.any()
    ret1, path1, _ = oracle.hmm_viterbi(x[:1], h)              # one vector: the back-pass loop never runs
    assert ret1 == 0.0 and path1.tolist() == [0]
