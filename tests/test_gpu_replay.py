"""configuration recording / replay.  compress = BitArray(conf .== 1)
(HubbardModel.jl:56-59, src/configurations.jl:24-43), replay! = calculate_greens(mc, 0) per
recorded configuration (DQMC.jl:605-697; reference tests: test/FileIO.jl:153-175,
test/flavortests_DQMC.jl:62-69)."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


def test_conf_bits_roundtrip_and_layout(gpu):
    mc = gpu.DQMC(gpu.HubbardModelAttractive(4, 2), beta=1.3, delta_tau=0.1, safe_mult=1, n_walkers=3, seed=11)
    for w in range(3):
        c = mc.conf(w)                       # 16 x 13 = 208 elements: 3 full chunks + a ragged one
        cc = mc.conf_bits(w)
        assert cc == gpu.compress(c)         # device packing == numpy restatement of Julia's BitArray layout
        flat = c.reshape(-1, order="F")
        for i in (0, 1, 63, 64, 127, 207):
            assert ((int(cc.chunks[i // 64]) >> (i % 64)) & 1) == (1 if flat[i] == 1 else 0)
        assert int(cc.chunks[-1]) >> (208 - 192) == 0   # trailing bits of the last chunk are zero
        mc.set_conf_bits((w + 1) % 3, cc)
        assert np.array_equal(mc.conf((w + 1) % 3), c)
    mc.close()


def test_replay_matches_oracle(gpu, O):
    """record during a run, replay with a different batch shape: G(slice 0) from scratch equals the
    oracle's calculate_greens(mc, 0) on the decompressed configuration; sums equal the direct sums"""
    L, beta = 4, 1.0
    mc = gpu.DQMC(gpu.HubbardModelRepulsive(L, 2), beta=beta, n_walkers=1, seed=5, thermalization=2, sweeps=10,
                  measure_rate=2)
    rec = gpu.ConfigRecorder(rate=2)
    mc.run(recorder=rec)
    assert len(rec) == 5
    mc.close()
    mc2 = gpu.DQMC(gpu.HubbardModelRepulsive(L, 2), beta=beta, n_walkers=2, seed=99)
    acc = mc2.replay(rec)                    # 5 configurations on 2 walkers: batches 2 + 2 + 1
    assert acc[-1] == 5
    n = L * L
    ref = np.zeros_like(acc)
    for cc in rec:
        o = O.OracleDQMC(L, "repulsive", beta=beta)
        o.set_conf(gpu.decompress(cc))
        o.set_greens_eff(o.calculate_greens_at(0))
        G = o.greens()
        g = np.concatenate([x.reshape(-1, order="F") for x in G])
        ref[:2 * n * n] += g
        ref[2 * n * n:4 * n * n] += g * g
        for b in range(2):
            ref[4 * n * n + b * n:4 * n * n + (b + 1) * n] += 1.0 - np.diag(G[b])
        ref[-1] += 1
    assert relerr(acc, ref) < 1e-10
    # single configuration: mc.s.greens after replay_greens(0) is the effective G at slice 0
    mc2.set_conf_bits(0, rec[0]); mc2.set_conf_bits(1, rec[1])
    mc2.replay_greens(0)
    for w in range(2):
        o = O.OracleDQMC(L, "repulsive", beta=beta)
        o.set_conf(gpu.decompress(rec[w]))
        for b in range(2):
            assert relerr(mc2.greens_eff(w)[b], o.calculate_greens_at(0)[b]) < 1e-10
    mc2.close()
