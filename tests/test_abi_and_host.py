"""C-ABI surface and host logic, no GPU: the library loads, exports every symbol that
include/dqmc_hip.h declares, validates its arguments, and the product path neither
links nor imports the CPU oracle."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "dqmc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dqmc_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(mc_amd):
    from montecarlo_jl_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "symbol %s declared in include/dqmc_hip.h is not exported" % n
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding and header disagree"


def test_library_is_built_from_the_sources_in_the_tree(mc_amd):
    """dqmc_build_source_hash() (csrc/Makefile: SOURCE_HASH) restated: sha256 over the sorted csrc/*.hip, *.h, *.inl, engine.cpp,
    then the Makefile and the assembly patcher.  A library left over from other sources (the .so is not in git: it travels to
    the GPU box as built) fails here instead of being measured; the same hash decides whether bench.py calls a committed
    profile stale"""
    import glob, hashlib
    from montecarlo_jl_amd import _lib
    csrc = os.path.join(ROOT, "montecarlo.jl_amd", "csrc")
    files = sorted(set(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) +
                       glob.glob(os.path.join(csrc, "*.inl")) + [os.path.join(csrc, "engine.cpp")]), key=os.path.basename)
    h = hashlib.sha256()
    for f in files + [os.path.join(csrc, "Makefile"), os.path.join(ROOT, "tools", "scan_mfma_war.py")]:
        h.update(open(f, "rb").read())
    assert _lib.lib().dqmc_build_source_hash().decode() == h.hexdigest()[:12]


def test_no_device_and_argument_errors(mc_amd):
    from montecarlo_jl_amd import _lib
    lib = _lib.lib()
    n = 16
    eye = np.eye(n).reshape(-1).copy()
    def params(**kw):
        d = dict(n_sites=n, model_kind=0, slices=10, safe_mult=10, n_walkers=1, device_id=0,
                 check_propagation_error=1, check_sign_problem=1, delta_tau=0.1, U=1.0)
        d.update(kw)
        return _lib.Params(d["n_sites"], d["model_kind"], d["slices"], d["safe_mult"], d["n_walkers"], d["device_id"],
                           d["check_propagation_error"], d["check_sign_problem"], d["delta_tau"], d["U"],
                           _lib.dptr(eye), _lib.dptr(eye), _lib.dptr(eye), _lib.dptr(eye))
    h = C.c_void_p()
    # stack.jl:115: slices must be divisible by safe_mult
    assert lib.dqmc_create(C.byref(params(slices=15)), C.byref(h)) == _lib.ERR_INVALID
    assert b"safe_mult" in lib.dqmc_last_error(None)
    assert lib.dqmc_create(C.byref(params(model_kind=7)), C.byref(h)) == _lib.ERR_INVALID
    assert lib.dqmc_create(C.byref(params(U=-1.0)), C.byref(h)) == _lib.ERR_INVALID
    if lib.dqmc_device_count() == 0:
        # no CPU fallback: without a device creation fails loudly
        assert lib.dqmc_create(C.byref(params()), C.byref(h)) == _lib.ERR_NO_DEVICE
        with pytest.raises(_lib.DQMCError):
            mc_amd.DQMC(mc_amd.HubbardModelAttractive(4, 2), beta=1.0)
        with pytest.raises(_lib.DQMCError):
            mc_amd.vmul(np.eye(4), np.eye(4))


def test_missing_library_fails_loudly(tmp_path):
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import importlib.util, os\n"
        "import __graft_entry__ as g\n"
        "spec = importlib.util.spec_from_file_location('x_lib', os.path.join(g.PKG_DIR, '_lib.py'))\n"
        "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
        "m.LIB_PATH = %r\n"
        "try:\n    m.lib()\nexcept ImportError as e:\n    print('LOUD', e)\n" % (ROOT, str(tmp_path / "nope.so")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "LOUD" in out.stdout and "no CPU fallback" in out.stdout


def test_product_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "montecarlo.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert "dqmc_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f
    # the shared library does not link the oracle either
    out = subprocess.run(["ldd", os.path.join(pkg, "libdqmc_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_host_hopping_exponentials(mc_amd, O):
    """init_hopping_matrix_exp (stack.jl:167-181): eT*eTinv = I, eT2 = eT*eT; same as the oracle's host code"""
    T = mc_amd.HubbardModelAttractive(4, 2, mu=0.3).hopping_matrix()[0]
    eT, eTinv, eT2, eTinv2 = mc_amd.hopping_exponentials(T, 0.1)
    assert np.abs(eT @ eTinv - np.eye(16)).max() < 1e-14
    assert np.abs(eT2 - eT @ eT).max() == 0.0
    import scipy.linalg as sla
    assert np.abs(eT - sla.expm(-0.05 * T)).max() < 1e-14
    o = O.hopping_exponentials(T, 0.1)
    assert np.array_equal(o[0], eT) and np.array_equal(o[3], eTinv2)


def test_rand_conf_layout(mc_amd):
    rng = np.random.Generator(np.random.Philox(key=5))
    c = mc_amd.rand_conf(rng, 16, 10)
    assert c.shape == (16, 10) and c.dtype == np.int8 and set(np.unique(c)) == {-1, 1} and c.flags.f_contiguous


def test_sharding(mc_amd):
    first, ids = mc_amd.walker_range(3, 8, 32)
    assert first == 96 and ids == list(range(96, 128))
    seeds = [s for r in range(4) for s in mc_amd.walker_seeds(123, r, 4, 2)]
    assert seeds == list(range(123, 131))


def test_compress_layout_host(mc_amd):
    """BitArray(conf .== 1): element i (1-based, column-major) -> bit (i-1) % 64 of chunk (i-1) / 64"""
    rng = np.random.Generator(np.random.Philox(key=3))
    c = mc_amd.rand_conf(rng, 9, 15)   # 135 elements: ragged last chunk
    cc = mc_amd.compress(c)
    assert cc.chunks.dtype == np.uint64 and cc.chunks.size == 3
    flat = c.reshape(-1, order="F")
    for i in range(135):
        assert ((int(cc.chunks[i // 64]) >> (i % 64)) & 1) == (flat[i] == 1)
    assert int(cc.chunks[2]) >> (135 - 128) == 0
    assert np.array_equal(mc_amd.decompress(cc), c)
    r = mc_amd.ConfigRecorder(rate=3)
    assert len(r) == 0 and len(mc_amd.Discarder()) == 0


def test_pair_iterator_contracts(mc_amd, R):
    """test/lattices.jl:52-58,112-140: directions sorted by norm; every pair's minimal-image
    displacement equals the direction it is filed under; N^2 pairs.  The product's table equals the
    independent restatement in oracle/ref_test_oracle.py."""
    for L in (3, 4):
        l = mc_amd.SquareLattice(L)
        it = mc_amd.EachSitePairByDistance(l)
        norms = [np.linalg.norm(d) for d in it.directions]
        assert all(norms[i - 1] < norms[i] + 1e-5 for i in range(1, len(norms)))
        trip = list(it)
        assert len(trip) == len(it) == (L * L) ** 2 and it.ndirections() == L * L
        dirs, table = R.square_pair_directions(L)
        assert np.array_equal(table, it.dir_of)
        for d, s, t in trip[:: 7]:
            i1, j1 = (s - 1) % L, (s - 1) // L
            i2, j2 = (t - 1) % L, (t - 1) // L
            dv = it.directions[d - 1]
            assert (i1 - i2 - dv[0]) % L == 0 and (j1 - j2 - dv[1]) % L == 0
    assert list(mc_amd.EachSitePairByDistance(mc_amd.SquareLattice(4)).directions[0]) == [0.0, 0.0]


def test_each_local_quad_by_distance(mc_amd, R):
    """EachLocalQuadByDistance{K} (lattice_iterators.jl:258-353): length, grouping and the
    geometric meaning of every quadruple; K defaults to 1 + number of nearest neighbours"""
    L = 4
    l = mc_amd.SquareLattice(L)
    it = mc_amd.EachLocalQuadByDistance(l)
    assert it.K == 5 and it.ndirections() == (16, 5, 5)
    assert len(it) == (16 * 5) ** 2  # (sum of targets per source)^2, lattice_iterators.jl:281
    dirs = it.pairs_by_dir.directions
    pos = [np.array([i + 1.0, j + 1.0]) for j in range(L) for i in range(L)]

    def same(a, b):  # equal up to lattice vectors
        d = (a - b) / L
        return np.abs(d - np.round(d)).max() < 1e-9

    count, last = 0, 0
    for lin, s1, t1, s2, t2 in it:
        assert lin >= last  # sorted by the linear index of (dir12, dir1, dir2)
        last = lin
        d12, d1, d2 = (lin - 1) % 16, ((lin - 1) // 16) % 5, (lin - 1) // 80
        assert same(pos[s1 - 1] - pos[s2 - 1], dirs[d12])
        assert same(pos[s1 - 1] - pos[t1 - 1], dirs[d1]) and same(pos[s2 - 1] - pos[t2 - 1], dirs[d2])
        count += 1
    assert count == len(it)
    # first direction is on-site, the next four are the nearest neighbours
    assert np.all(it.trg_of[:, 0] == np.arange(16))
    assert sorted(np.linalg.norm(d) for d in dirs[1:5]) == [1.0] * 4
    # the independent restatement in the oracle sees the same target table
    _, table = R.square_pair_directions(L)
    for s in range(16):
        for k in range(5):
            assert table[s, it.trg_of[s, k]] == k
    with pytest.raises(ValueError):
        mc_amd.EachLocalQuadByDistance(mc_amd.SquareLattice(2), K=5)  # a 2x2 torus has 4 directions
