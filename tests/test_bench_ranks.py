"""The multi-rank path of bench.py under the driver's eyes (walkers sharded over ranks, one reduction).

* CPU: walker_block / walker_range deal every global walker id to exactly one rank for the world sizes of the scaling
  bench (config 4: 256 repulsive walkers over 2 / 4 / 8 GPUs), and the per-walker seeds do not depend on the sharding.
* GPU: `bench.py --gpus 2` as a FRESH child process (the bench parent makes no GPU call before it starts its ranks) with
  BENCH_ONE_DEVICE=1 - both ranks on the one device of the test box, gloo process group, host-mediated export -> all_reduce
  -> import reduction - must print ONE JSON line with the two-rank shape."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [1, 2, 4, 8, 3])
def test_every_walker_has_exactly_one_rank(mc_amd, world):
    total = 256
    seen = []
    for rank in range(world):
        lo, hi = mc_amd.walker_block(rank, world, total)
        assert 0 <= lo <= hi <= total
        seen += list(range(lo, hi))
        assert hi - lo in (total // world, total // world + 1)
    assert seen == list(range(total))
    # weak scaling (config 3): rank r holds walkers r * w .. r * w + w - 1, seeds follow the GLOBAL walker id
    per = 32
    ids, seeds = [], []
    for rank in range(world):
        first, mine = mc_amd.walker_range(rank, world, per)
        assert first == rank * per
        ids += mine
        seeds += mc_amd.walker_seeds(123, rank, world, per)
    assert ids == list(range(world * per))
    assert seeds == [123 + w for w in range(world * per)]


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device(gpu):
    env = dict(os.environ, BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--walkers", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["warmup"] == 0
    assert d["config"]["total_walkers"] == 4 and d["config"]["walkers_per_gpu"] == 2
    assert "export" in d["config"]["reduction"] and "gloo" in d["config"]["reduction"] and "import" in d["config"]["reduction"]
    assert d["config"]["qr_fallbacks"] == 0 and d["config"]["device_errors"] == 0
    assert d["value"] > 0 and d["unit"] == "walker-sweeps/s" and "roofline" in d
