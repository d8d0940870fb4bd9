"""Pre-pivoted blocked UDT in one launch (csrc/qrb.hip; n = 256) against the oracle.

The reference's udt_AVX_pivot! (src/linalg/UDT.jl:192-306) searches the largest trailing column at every step; the device
kernel takes the column order once, from the norms of the input.  Two gates:
  * factor level: U, D, T, pivot against the ORACLE RUN WITH THE SAME PRE-SORTED ORDER (oracle study switch
    orc_set_udt_presort) - same Householder vectors, so the factors agree to rounding;
  * the reference's contracts (test/slice_matrices.jl:202-234) and, through the engine, G / HS-field parity against the
    reference-rule oracle (tests/test_gpu_parity_r3.py, test_gpu_dqmc.py run on this path by default).
"""
import os

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


CONTRACTS_ONLY = (4,)


def graded_cases(n, rng):
    X = rng.standard_normal((6, n, n))
    X[1] *= np.exp(rng.uniform(-20, 20, size=n))[None, :]                      # graded columns (slice-sequence builds)
    X[2] = (X[2] * np.exp(np.linspace(18, -18, n))[:, None]) * np.exp(np.linspace(15, -15, n))[None, :]  # rows and columns
    q, _ = np.linalg.qr(X[3])
    X[3] = q + np.diag(np.exp(np.linspace(12, -12, n)))                       # orthogonal + diagonal (stack.jl:368)
    # graded singular values behind a dense mixing matrix: numerically rank deficient (cond 1e13, no graded columns), so the
    # trailing part of D is rounding noise in ANY implementation - contracts only, no factor comparison (CONTRACTS_ONLY)
    X[4] = X[4] @ np.diag(np.exp(np.linspace(15, -15, n))) @ rng.standard_normal((n, n)) / n
    X[5] = np.eye(n)[:, rng.permutation(n)] * np.exp(rng.uniform(-3, 3, size=n))[None, :]  # sparse, distinct norms
    return X


@pytest.mark.parametrize("apply_pivot", [True, False])
def test_blocked_udt_factors_against_presorted_oracle(gpu, O, apply_pivot):
    n = 256
    rng = np.random.default_rng(7)
    X = graded_cases(n, rng)
    U, D, T, piv = gpu.udt_AVX_pivot(X, apply_pivot)
    O.lib().orc_set_udt_presort(1)
    try:
        for i in range(X.shape[0]):
            Uo, Do, To, po = O.udt_pivot(X[i], apply_pivot)
            assert sorted(piv[i]) == list(range(1, n + 1))
            if not np.array_equal(po, piv[i]):
                # a near-tie of two input norms went the other way (different summation order): must be a tie
                nrm = (X[i] ** 2).sum(axis=0)
                d = np.nonzero(po != piv[i])[0]
                assert np.all(np.abs(nrm[po[d] - 1] / nrm[piv[i][d] - 1] - 1) < 1e-12)
                continue
            if i in CONTRACTS_ONLY:
                continue
            assert np.abs(D[i] / Do - 1).max() < 1e-10          # element by element, also the smallest ones
            assert relerr(U[i], Uo) < 1e-10
            Tt, Tto = (T[i], To) if apply_pivot else (np.triu(T[i]), np.triu(To))
            assert relerr(Tt, Tto) < 1e-10
    finally:
        O.lib().orc_set_udt_presort(0)


@pytest.mark.parametrize("apply_pivot", [True, False])
@pytest.mark.parametrize("batch", [1, 9, 32])
def test_blocked_udt_contracts(gpu, apply_pivot, batch):
    """test/slice_matrices.jl:202-234: U*Diagonal(D)*T = X (Val(true)); U*D*UpperTriangular(T)*P = X with P[i, pivot[i]] = 1"""
    n = 256
    rng = np.random.default_rng(batch)
    X = np.concatenate([graded_cases(n, rng) for _ in range((batch + 5) // 6)])[:batch]
    U, D, T, piv = gpu.udt_AVX_pivot(X, apply_pivot)
    for i in range(batch):
        assert relerr(U[i].T @ U[i], np.eye(n)) < 1e-12
        assert np.all(D[i] > 0)
        assert sorted(piv[i]) == list(range(1, n + 1))
        if apply_pivot:
            rec = (U[i] * D[i]) @ T[i]
        else:
            P = np.zeros((n, n)); P[np.arange(n), piv[i] - 1] = 1
            rec = (U[i] * D[i]) @ np.triu(T[i]) @ P
        scale = np.abs(X[i]).max(axis=0)
        assert (np.abs(rec - X[i]) / scale[None, :]).max() < 1e-12
        # the pivot order is the descending order of the input's column norms
        nrm = (X[i] ** 2).sum(axis=0)[piv[i] - 1]
        assert np.all(np.diff(nrm) <= 1e-12 * nrm[:-1])


def test_blocked_udt_singular_and_tied_columns(gpu):
    """a zero column (UDT.jl:136-138: tau = 0, the column stays as it is, D = 0) sorts last and nothing hangs or spreads;
    two equal columns keep their input order (first maximum first, UDT.jl:151-168)"""
    n = 256
    rng = np.random.default_rng(5)
    X = rng.standard_normal((3, n, n))
    X[0][:, 7] = 0.0
    X[1][:, 9] = X[1][:, 200]
    U, D, T, piv = gpu.udt_AVX_pivot(X, False)
    assert piv[0][255] == 8 and D[0][255] == 0.0
    assert np.all(np.isfinite(D[0])) and np.all(D[0][:255] > 0) and np.all(np.isfinite(U[0]))
    assert relerr(U[0].T @ U[0], np.eye(n)) < 1e-12
    P = np.zeros((n, n)); P[np.arange(n), piv[0] - 1] = 1
    R = np.triu(T[0])[:255] * D[0][:255, None]          # (row 255 of T is 0 / 0)
    assert np.abs(U[0][:, :255] @ R @ P - X[0]).max() < 1e-12
    i9, i200 = list(piv[1]).index(10), list(piv[1]).index(201)
    assert i200 == i9 + 1
    P = np.zeros((n, n)); P[np.arange(n), piv[1] - 1] = 1
    assert relerr((U[1] * D[1]) @ np.triu(T[1]) @ P, X[1]) < 1e-12
    assert D[1][i200] < 1e-12 * D[1][i9]                 # the second copy has nothing left once the first is eliminated
