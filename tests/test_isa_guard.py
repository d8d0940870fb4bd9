"""Guards on the generated gfx950 code of two hot kernels (no GPU needed: hipcc cross-compiles).  Round 3 found, with in-kernel
stamps, requests that the compiler had serialised invisibly - a bounded load sunk into a branch of its own and waited for at
the merge, load/LDS-store pairs waited for one by one, LDS operands read with flat loads.  The cures are
source idioms (clamped unconditional loads behind one scheduling barrier, requests-then-stores, offsets instead of pointer
choices) that a later edit can undo without any test noticing; these checks notice.  tools/scan_isa.py runs the same scan
over every kernel."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "montecarlo.jl_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only"]


def _kernels(src, extra=()):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    subprocess.run(["hipcc"] + FLAGS + list(extra) + [os.path.join(CSRC, src), "-o", out], check=True,
                   stderr=subprocess.DEVNULL, timeout=600)
    res, kern, seq, flat = {}, None, [], 0
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern, seq, flat = m.group(1), [], 0
            continue
        t = line.strip()
        if t.startswith(("global_load", "flat_load", "buffer_load")):
            seq.append("L")
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            seq.append("W")
        if t.startswith(("flat_load", "flat_store")):
            flat += 1
        if t.startswith(".Lfunc_end") and kern:
            runs = [len(r) // 2 for r in re.findall(r"(?:LW){4,}", "".join(seq))]
            res[kern] = (max(runs) if runs else 0, flat)
            kern = None
    return res


def test_trsm_requests_are_not_serialised():
    k = _kernels("trsm_rl.hip", ["-mllvm", "-pragma-unroll-threshold=4000000"])
    hot = {n: v for n, v in k.items() if "trsm_rl_kernel" in n}
    assert len(hot) == 2
    for name, (longest_run, flat) in hot.items():
        assert longest_run == 0, (name, "consecutive load -> s_waitcnt vmcnt(0) pairs", longest_run)
        assert flat == 0, (name, "flat memory operations", flat)


def test_slab_kernel_stages_x0_with_all_requests_in_flight():
    k = _kernels("slab.hip")
    hot = {n: v for n, v in k.items() if "slab_chain_kernel" in n}
    assert len(hot) == 1
    for name, (longest_run, flat) in hot.items():
        assert longest_run < 4, (name, longest_run)
        assert flat == 0, (name, flat)
