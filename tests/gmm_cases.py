"""Synthetic GMM / HMM parameter records and MFCC-like vectors for the rank-4 tests (the reference ships no
parameter or feature files, so there is nothing of its own to load)."""
import numpy as np

from oracle_lib import GMM_PARAM, HMM_PARAM


def gmm_records(seed, n, var_lo=0.5, var_hi=4.0):
    rng = np.random.default_rng(seed)
    g = np.zeros(n, GMM_PARAM)
    a = rng.uniform(0.2, 1.0, (n, 4))
    g["alpa"] = a / a.sum(axis=1, keepdims=True)
    g["mean"] = rng.normal(0.0, 2.0, (n, 4, 12))
    cov = rng.normal(0.0, 0.1, (n, 4, 12, 12))                 # off-diagonal entries are never read
    idx = np.arange(12)
    cov[:, :, idx, idx] = rng.uniform(var_lo, var_hi, (n, 4, 12))
    g["covariance"] = cov
    q = np.linalg.qr(rng.normal(0.0, 1.0, (n, 4, 12, 4)))[0]   # orthonormal columns, like eigenvectors
    g["eigenVector"] = q
    return g


def hmm_records(seed, n, **kw):
    rng = np.random.default_rng(seed + 1000)
    h = np.zeros(n, HMM_PARAM)
    h["gMMParam"] = gmm_records(seed, 6 * n, **kw).reshape(n, 6)
    t = rng.uniform(0.05, 1.0, (n, 6, 6))
    h["transProb"] = t / t.sum(axis=2, keepdims=True)
    return h


def hmm_records_finite(seed, n):
    """Models whose six states are small perturbations of one tight GMM: every state's density stays far above
    6 for vectors_near(), the only regime in which Viterbi_version1.cpp:196's log(log-probability) is finite."""
    rng = np.random.default_rng(seed + 2000)
    h = hmm_records(seed, n)
    for i in range(n):
        base = gmm_records(seed + 17 * i, 1, var_lo=1e-4, var_hi=2e-4)[0]
        for s in range(6):
            r = base.copy()
            r["mean"][:, :4] += rng.normal(0.0, 0.004, (4, 4))
            h["gMMParam"][i, s] = r
    return h


def vectors(seed, n, scale=3.0):
    return np.random.default_rng(seed).normal(0.0, scale, (n, 12))


def vectors_near(seed, n, gmm_record, spread=0.003):
    """Vectors whose projection lands next to one of the record's mixture means: densities far above 1."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 12))
    for i in range(n):
        k = rng.integers(0, 4)
        E = gmm_record["eigenVector"][k]                       # [12, 4]
        target = gmm_record["mean"][k][:4] + rng.normal(0.0, spread, 4)
        out[i] = E @ np.linalg.solve(E.T @ E, target)
    return out


def offsets(lengths):
    return np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
