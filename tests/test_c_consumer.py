"""A compiled C99 consumer of include/dqmc_hip.h (tests/c_consumer/consumer.c): the header is valid C, its structs need no
hand-made mirror, and the sweep path can be driven from compiled code the way a Julia `ccall` would (SURVEY section 7
step 2; the Python side re-declares the structs in montecarlo.jl_amd/_lib.py - this test is the check that nothing in that
mirror hides a layout disagreement between header and library)."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from conftest import relerr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_consumer", "consumer.c")
PKG = os.path.join(ROOT, "montecarlo.jl_amd")


@pytest.fixture(scope="module")
def consumer(tmp_path_factory, mc_amd):
    """gcc -std=c99 -pedantic -Werror against the header, linked with the shipped library"""
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path_factory.mktemp("cc") / "consumer")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), SRC,
           "-o", exe, "-L", PKG, "-ldqmc_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath-link,/opt/rocm/lib",
           "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return exe


def _env():
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return env


def test_c_consumer_compiles_links_and_probes(consumer, mc_amd):
    """no device work: the library loads into a C program, the invalid dqmc_params is refused with DQMC_ERR_INVALID (the
    reference throws InexactError at stack.jl:115), struct sizes as ctypes sees them"""
    r = subprocess.run([consumer, "--probe"], capture_output=True, text=True, timeout=120, env=_env())
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "create(invalid) -1 handle null" in r.stdout and "divisible by safe_mult" in r.stdout
    import ctypes
    import importlib
    L = importlib.import_module(mc_amd.__name__ + "._lib")   # the hand-made mirror of the structs agrees with the compiler's
    assert "sizeof dqmc_params %d dqmc_stats %d" % (ctypes.sizeof(L.Params), ctypes.sizeof(L.Stats)) in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["attractive", "repulsive"])
def test_c_consumer_runs_the_sweep_path_against_the_oracle(consumer, gpu, O, kind, tmp_path):
    """4x4 Hubbard, beta = 1, 2 walkers, 2 sweeps on host-supplied uniforms: HS field bit-exact, G within 1e-10, the same
    number of uniforms drawn (the conditional rule of DQMC.jl:573), the same counters"""
    Lx, beta, dtau, safe, walkers, sweeps, nu = 4, 1.0, 0.1, 5, 2, 2, 4000
    n, M = Lx * Lx, 10
    nb = 2 if kind == "repulsive" else 1
    rng = np.random.default_rng(11)
    refs = []
    for w in range(walkers):
        o = O.OracleDQMC(Lx, kind, beta=beta, delta_tau=dtau, safe_mult=safe, U=1.0)
        refs.append(o)
    T = refs[0].T
    exps = O.hopping_exponentials(T, dtau)
    blob = struct.pack("<8i2d", n, 1 if kind == "repulsive" else 0, M, safe, walkers, sweeps, nu, nb, dtau, 1.0)
    for e in exps:
        e = np.asarray(e, dtype=np.float64)
        mats = [e] * nb if e.ndim == 2 else list(e)
        for m in mats:
            blob += np.asfortranarray(m).tobytes(order="F")
    confs, us = [], []
    for w in range(walkers):
        c = (2 * rng.integers(0, 2, size=(n, M)) - 1).astype(np.int8)
        u = rng.random(nu)
        confs.append(c); us.append(u)
        blob += np.asfortranarray(c).tobytes(order="F") + u.tobytes()
    prob, out = tmp_path / "problem.bin", tmp_path / "out.bin"
    prob.write_bytes(blob)
    r = subprocess.run([consumer, str(prob), str(out)], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, (r.stdout, r.stderr)
    raw = out.read_bytes()
    cs, direction = struct.unpack_from("<2i", raw, 0)
    off = 8
    for w, o in enumerate(refs):
        o.set_conf(confs[w]); o.set_uniforms(us[w]); o.prepare(); o.sweeps(sweeps)
        conf = np.frombuffer(raw, dtype=np.int8, count=n * M, offset=off).reshape((n, M), order="F"); off += n * M
        G = np.frombuffer(raw, dtype=np.float64, count=nb * n * n, offset=off); off += 8 * nb * n * n
        used, prop, acc = struct.unpack_from("<Qqq", raw, off); off += 24
        assert np.array_equal(conf, o.conf())
        for b, g0 in enumerate(o.greens_eff()):
            assert relerr(G[b * n * n:(b + 1) * n * n].reshape((n, n), order="F"), g0) < 1e-10
        st = o.stats()
        assert (used, prop, acc) == (o.uniforms_used(), st.prop_local, st.acc_local)
    assert (cs, direction) == (refs[0].current_slice, refs[0].direction)
    assert off == len(raw)
