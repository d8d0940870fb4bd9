"""Comparison of an independent sampler with the statistical goldens the reference holds for its own seeded runs
(test/integration_tests.jl:29-185).  The goldens are means of ONE run of 100 measurements under Julia's
MersenneTwister stream, which cannot be reproduced here; next to every mean the reference also stores the
std_error of that run.  Rules (tolerances are the reference's own; the one widened limit is explained in rule 1):

  1. every element: |ours - golden| <= Z * sqrt(se_golden^2 + se_ours^2)   (Z = 4.5; exact zeros must be exact).
     For the 400 pairing-correlation elements Z = 6.5 - the one place where a limit is wider than 4.5 sigma.  Their
     published std_errors go down to 1e-5 (0.4 % of the value, from 100 measurements after 10 thermalisation sweeps)
     and a 6000-sample run resolves offsets of 1e-4 there - 300 times below the reference's own atol of 0.04, which
     rule 2 holds for all 400 elements.  The offsets belong to the GOLDEN, not to our pairing kernel: an independent
     CPU run of the oracle with the python formulas of oracle/ref_test_oracle.py (6 chains x 3000 measurements,
     tools/pc_offsets.py, profiles/r03_pc_offsets.txt) shows the same elements at the same distance (max |z| 6.2, nine
     elements above 4.5 - e.g. (dir 14, 4, 3): golden -0.005327 +- 0.000022, oracle -0.005193 +- 0.000005, device
     -0.005194) while oracle and device agree with each other (mean z^2 1.4 with the oracle's error alone).  The
     outliers sit on symmetry-related elements with nearly equal golden values, i.e. they are one fluctuation of the
     reference's single 100-measurement run (its 10 thermalisation sweeps from a random field included), counted
     several times, with a std_error estimated from those same 100 correlated samples.
  2. where the reference test carries an atol A: every element agrees within A itself (all_at_atol: CDC, SDC, PC,
     magnetisations), or - for the Green's function, whose diagonal has a published std_error of 0.019 against
     A = 0.04, and for the recorded HS field (std_error 0.1) - every element whose own published standard error
     allows it (3 se_golden <= A)
  3. the mean of z^2 over the elements of an observable stays below 3 (the two samplers draw from one distribution;
     the published std_error is itself an estimate from 100 samples and symmetry-equivalent elements are correlated)
"""
import json
import os

import numpy as np

Z = 4.5


def load(name):
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", name)))


def golden_arrays(ent, shape=None, order="F"):
    m = np.array(ent["mean"], dtype=float)
    s = np.array(ent["std_error"], dtype=float) if "std_error" in ent else None
    if shape is not None and m.ndim == 1:
        m = m.reshape(shape, order=order)
        s = s.reshape(shape, order=order) if s is not None else None
    return m, s


def binned_error(samples, nbins=20):
    """standard error of the mean of a (possibly autocorrelated) series from the scatter of bin means"""
    x = np.asarray(samples, dtype=float)
    n = (x.shape[0] // nbins) * nbins
    b = x[:n].reshape((nbins, n // nbins) + x.shape[1:]).mean(axis=1)
    return b.std(axis=0, ddof=1) / np.sqrt(nbins)


def check(label, ours, ours_se, gold_mean, gold_se, atol, all_at_atol=False, zmax=Z):
    ours, ours_se = np.asarray(ours, float), np.asarray(ours_se, float)
    assert ours.shape == gold_mean.shape, (label, ours.shape, gold_mean.shape)
    diff = np.abs(ours - gold_mean)
    se = np.sqrt(gold_se ** 2 + ours_se ** 2)
    exact = se == 0.0
    assert np.all(diff[exact] < 1e-12), (label, "exact zeros differ", diff[exact].max())
    z = diff[~exact] / se[~exact]
    assert z.size == 0 or z.max() <= zmax, (label, "z-score", float(z.max()))
    assert z.size < 8 or np.mean(z ** 2) < 3.0, (label, "mean z^2", float(np.mean(z ** 2)))
    n_atol = 0
    if atol is not None:
        strict = np.ones(diff.shape, bool) if all_at_atol else 3.0 * gold_se <= atol
        n_atol = int(strict.sum())
        assert np.all(diff[strict] <= atol), (label, "reference atol", float(diff[strict].max()), atol)
    return dict(max_abs=float(diff.max()), max_z=float(z.max()) if z.size else 0.0, n=int(diff.size), n_at_ref_atol=n_atol)
