#!/usr/bin/env python3
"""Generates tests/golden/*.json|npz.  Run in the build container (needs /root/reference).

Two kinds of fixtures:
  (a) DATA copied out of the reference's own tests (numbers only, no source text):
      - the SquareLattice(4) checkerboard table   test/flavortests_DQMC.jl:22-24
      - mean Green's functions of the end-to-end runs   test/integration_tests.jl:46-49, 115-118
  (b) vectors produced HERE by the independent restatements of the reference's test
      oracles (oracle/ref_test_oracle.py: scipy dgeqp3 UDT chain, numpy ED).
"""
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def julia_matrix(txt):
    rows = [r.split() for r in txt.strip().strip("[]").split(";")]
    return [[float(x) for x in r] for r in rows]


def main():
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(REF, "test/flavortests_DQMC.jl")).read()
    m = re.search(r"build_checkerboard\(sq\) == \(\[(.*?)\], UnitRange\[(.*?)\], (\d+)\)", src, re.S)
    cb = julia_matrix(m.group(1))
    groups = [[int(a), int(b)] for a, b in re.findall(r"(\d+):(\d+)", m.group(2))]
    json.dump({"source": "test/flavortests_DQMC.jl:22-24", "checkerboard": cb, "groups": groups,
               "n_groups": int(m.group(3))}, open(os.path.join(OUT, "checkerboard_square4.json"), "w"))

    src = open(os.path.join(REF, "test/integration_tests.jl")).read()
    att = src[src.index("attractive HubbardModel Simulation"):src.index("repulsive HubbardModel Simulation")]
    rep = src[src.index("repulsive HubbardModel Simulation"):]
    g_att = re.search(r"# Greens\s*@test \[(.*?)\]\s*≈ \(measured\[:G\] \|> mean\)", att, re.S).group(1)
    g_att = [float(x) for x in g_att.replace("\n", " ").split(",")]
    g_rep = re.search(r"# Greens\s*@test \[(.*?)\]\s*≈ measured\[:G\] \|> mean", rep, re.S).group(1)
    g_rep = julia_matrix(g_rep.replace("\n", " "))
    def vec(block, pat):
        t = re.search(pat, block, re.S).group(1)
        return [float(x) for x in t.replace("\n", " ").split(",")]
    cdc_att = vec(att, r"# Charge Density Correlation\s*@test \[(.*?)\]\[:\] ≈ mean\(measured\[:CDC\]\)")
    sdcx_att = vec(att, r"# Spin density correlations \(x, y, z\)\s*@test \[(.*?)\]\[:\] ≈ mean\(measured\[:SDCx\]\)")

    def all_goldens(block):
        """every `@test [numbers] ≈ <observable>` of a testset: {observable: {mean, std_error, has_atol}} (numbers
        only; `conf` is the mean of the recorded configurations, Julia matrix literal = rows separated by ';')"""
        out = {}
        for mm in re.finditer(r"@test \[(.*?)\]\s*(?:\[:\])?\s*≈\s*([^\n]*)", block, re.S):
            body, rhs = re.sub(r"\+ 0\.0im", "", mm.group(1)), mm.group(2)
            rows = [[float(x) for x in re.split(r"[,\s]+", r.strip()) if x] for r in body.replace("\n", " ").split(";")]
            key = re.search(r"measured\[:(\w+)\]", rhs)
            name = key.group(1) if key else "conf"
            ent = out.setdefault(name, {})
            ent["std_error" if "std_error" in rhs else "mean"] = rows if len(rows) > 1 else rows[0]
            ent["has_atol"] = "atol" in rhs
        return out

    json.dump({"source": "test/integration_tests.jl:29-94 (attractive 4x4, beta=1, 10+1000 sweeps, "
                         "measure_rate 10 = 100 measurements; atol = 4*dtau^2 = 0.04); `all` = every golden of the "
                         "testset (mean and std_error of G, CDC, SDCx/y/z, PC with K=5; mean of the recorded HS "
                         "fields), numbers only", "L": 4, "beta": 1.0, "atol": 0.04, "n_measurements": 100,
               "G_mean_colmajor": g_att, "CDC_mean": cdc_att, "SDCx_mean": sdcx_att, "all": all_goldens(att)},
              open(os.path.join(OUT, "integration_attractive_4x4.json"), "w"))
    json.dump({"source": "test/integration_tests.jl:98-185 (repulsive 2x2, beta=1, 10+1000 sweeps = 100 measurements; "
                         "atol = 2*dtau^2 = 0.02 where the reference gives one - CDC, Mz, SDC and PC are compared "
                         "with Julia's default isapprox there, i.e. bit-for-bit regression values of its own RNG "
                         "stream); `all` = every golden of the testset", "L": 2, "beta": 1.0, "atol": 0.02,
               "n_measurements": 100, "G_mean": g_rep, "all": all_goldens(rep[:rep.index("# TODO")])},
              open(os.path.join(OUT, "integration_repulsive_2x2.json"), "w"))

    # ---- (b) vectors from the independent restatements
    from oracle import oracle as O
    from oracle import ref_test_oracle as R
    # exact diagonalisation, parameters of test/ED/ED_tests.jl:91-95
    neighs = O.square_neighs(2)
    ed = {"source": "oracle/ref_test_oracle.py restating test/ED/ED.jl:68-120,497-518 with the models of "
                    "test/ED/ED_tests.jl:91-95; tolerance there: atol = rtol = 2*dtau^2 = 0.02",
          "repulsive_U1_t1": R.ed_hubbard_greens(neighs, 4, 1.0, 1.0, 0.0, 1.0).tolist(),
          "attractive_U1_mu1_t1": R.ed_hubbard_greens(neighs, 4, -1.0, 1.0, 1.0, 1.0).tolist()}
    json.dump(ed, open(os.path.join(OUT, "ed_hubbard_2x2.json"), "w"))
    # dgeqp3 Green's functions (test/testfunctions.jl:80-118) on seeded HS fields
    fx = {}
    for name, L, beta, mu, s, slices_at in (("a4", 4, 1.0, 0.0, 10, (0, 1, 5, 10)),
                                            ("a8", 8, 4.0, 0.0, 10, (0, 1, 17, 40)),
                                            ("a8mu", 8, 5.0, 0.5, 5, (0, 7, 25, 50))):
        T = O.hopping_square(L, 1.0, mu)
        eT, eTinv, eT2, eTinv2 = O.hopping_exponentials(T, 0.1)
        M = int(round(beta / 0.1))
        conf = O.random_conf(1000 + L, L * L, M)
        lam = np.arccosh(np.exp(0.5 * 1.0 * 0.1))
        B = lambda k: R.slice_matrix(eT2, lam, conf, k)
        fx[name + "_conf"] = conf
        fx[name + "_meta"] = np.array([L, beta, mu, s], dtype=np.float64)
        fx[name + "_slices"] = np.array(slices_at)
        fx[name + "_G"] = np.stack([R.calculate_greens_and_logdet(B, M, k, s) for k in slices_at])
    np.savez_compressed(os.path.join(OUT, "greens_dgeqp3.npz"), **fx)
    # lattice tables
    json.dump({"source": "src/lattices/square.jl:25-60 restated (oracle/dqmc_oracle.c orc_square_neighs)",
               **{"L%d" % L: O.square_neighs(L).tolist() for L in (2, 3, 4, 8)}},
              open(os.path.join(OUT, "square_neighs.json"), "w"))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
