"""Guards on the gfx950 code that SHIPS: the device code objects are taken out of montecarlo.jl_amd/libdqmc_hip.so
(llvm-objdump --offloading) and disassembled; no GPU needed.  Two families of checks, per kernel on the default path:

* serialised requests and flat operations (round 3, found with in-kernel stamps): runs of `global_load -> s_waitcnt vmcnt(0)`
  pairs - a bounded load sunk into a branch of its own, a kernel at its register limit requesting one element at a time - and
  flat_load / flat_store from pointers whose address space the compiler did not know.  The cures are source idioms that a
  later edit can undo without any numerical test noticing; every kernel has a budget = what it is known to contain today
  (0 for the kernels that were cleaned, the measured value for those that still carry such runs in cold code: the GEMM's
  generic-mode epilogue, the checkerboard factor loads).  tools/scan_isa.py prints the same figures from a fresh compile.
* MFMA source-C write-after-read (round 4, tools/scan_mfma_war.py): a load that the register allocator placed in the old
  accumulator registers of an out-of-place v_mfma_f64 right behind it.  Produced wrong results in the one-launch UDT; the
  build patches the assembly of the files that show the pattern.  No shipped kernel may contain it."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "montecarlo.jl_amd", "libdqmc_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
sys.path.insert(0, os.path.join(ROOT, "tools"))

# kernel-name fragment (demangled) -> (longest allowed run of load->wait(0) pairs, allowed flat operations)
BUDGET = {
    "slab_chain_kernel": (1, 0),                 # (the write-back scalings are requested ahead of the k-loop)
    "trsm_rl_kernel<true>": (1, 0),
    "trsm_rl_kernel<false>": (2, 0),
    "trsm_diag_inv_kernel": (1, 0),
    "sweep_fused_kernel<1, 8, 2>": (2, 11),      # config 3: elimination next to the previous chunk's flush
    "sweep_fused_kernel<2, 8, 2>": (2, 15),      # repulsive model (config 4)
    "sweep_lu4_kernel<1, true>": (1, 1),
    "sweep_lu4_kernel<2, true>": (1, 1),
    "sweep_flush_lu_kernel<true, 8, 1>": (1, 0),
    "sweep_flush_lu_kernel<true, 8, 2>": (1, 0),
    "qrb_udt_kernel": (1, 0),                    # one-launch UDT (a hand-off poll is one load -> wait by nature)
    "qr_coop_kernel": (1, 0),
    "qr_tail_kernel<128, 4>": (3, 0),
    "qr_panel_kernel<9>": (1, 0),
    "qr_tile256_kernel<true>": (2, 0),
    "udt_finish_kernel": (1, 0),
    # GEMM: k-loop clean; the plain and the array-scaled epilogues (everything the default path launches) request their
    # loads in batches - the longest run there is the 5-6 of the prologue.  What is left (35) is the generic epilogue behind
    # vs_get's run-time mode dispatch (conf-derived / inverted / clamped scales): only reached by the non-slab fallbacks
    "gemm_kernel<false, false, true, 0>": (35, 0),
    "gemm_kernel<false, true, true, 0>": (35, 0),
    "gemm_kernel<true, false, true, 0>": (35, 0),
    "gemm_kernel<false, false, false, -1>": (35, 0),
    "cb_apply_kernel<32>": (22, 0),
}


def _load_shipped():
    """{demangled kernel name: [instruction text]} of every function in the shipped library's gfx950 code objects"""
    if not os.path.exists(LIB) or not os.path.exists(OBJDUMP) or shutil.which("c++filt") is None:
        pytest.skip("library or llvm-objdump missing")
    tmp = tempfile.mkdtemp()
    shutil.copy(LIB, os.path.join(tmp, "lib.so"))
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True, timeout=120)
    funcs = {}
    for f in sorted(os.listdir(tmp)):
        if "gfx950" not in f:
            continue
        out = subprocess.run([OBJDUMP, "-d", f], cwd=tmp, check=True, capture_output=True, text=True, timeout=600).stdout
        cur = None
        for line in out.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
            if m:
                cur = m.group(1)
                funcs[cur] = []
            elif cur and line.startswith("\t"):
                funcs[cur].append(line.split("//")[0].strip())
    names = list(funcs)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    shutil.rmtree(tmp, ignore_errors=True)
    return {d: funcs[n] for n, d in zip(names, dem)}


@pytest.fixture(scope="module")
def shipped():
    return _load_shipped()


def _runs_and_flat(ins):
    seq, flat = [], 0
    for t in ins:
        if t.startswith(("global_load", "flat_load", "buffer_load")):
            seq.append("L")
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            seq.append("W")
        if t.startswith(("flat_load", "flat_store", "flat_atomic")):
            flat += 1
    runs = [len(r) // 2 for r in re.findall(r"(?:LW)+", "".join(seq))]
    return (max(runs) if runs else 0), flat


def test_default_path_kernels_stay_within_their_budgets(shipped):
    missing = []
    for frag, (max_run, max_flat) in BUDGET.items():
        hits = {n: v for n, v in shipped.items() if frag in n and "(" in n}
        if not hits:
            missing.append(frag)
            continue
        for name, ins in hits.items():
            run, flat = _runs_and_flat(ins)
            assert run <= max_run, (name[:90], "consecutive load -> s_waitcnt vmcnt(0) pairs", run, "budget", max_run)
            assert flat <= max_flat, (name[:90], "flat memory operations", flat, "budget", max_flat)
    assert not missing, ("kernels named in the budget table are not in the shipped library", missing)


def test_no_mfma_source_c_write_after_read_in_the_shipped_library(shipped):
    import scan_mfma_war as W
    bad = []
    n_mfma = 0
    for name, ins in shipped.items():
        if not any(t.startswith("v_mfma") for t in ins):
            continue
        n_mfma += 1
        lines = ["\t" + t for t in ins]
        for ln, ln2, _ in W.hazards(lines):
            bad.append((name[:80], ins[ln], ins[ln2]))
    assert n_mfma >= 10          # (the scan really saw the MFMA kernels)
    assert not bad, bad[:5]


def test_the_scanner_sees_the_pattern_and_the_patch_removes_it():
    import scan_mfma_war as W
    asm = ["\tv_mfma_f64_16x16x4_f64 v[90:97], v[22:23], v[82:83], v[98:105]",
           "\tds_read2_b64 v[104:107], v29 offset0:64 offset1:96",
           "\tv_mfma_f64_16x16x4_f64 v[90:97], v[32:33], v[84:85], v[90:97]"]
    assert len(W.hazards(asm)) == 1
    fixed, n = W.patch(asm)
    assert n == 1 and not W.hazards(fixed) and sum("s_nop" in x for x in fixed) >= 5
    # in-place accumulation, or a load into other registers, is not a hazard
    assert not W.hazards(["\tv_mfma_f64_16x16x4_f64 v[90:97], v[22:23], v[82:83], v[90:97]", "\tds_read_b64 v[96:97], v1"])
    assert not W.hazards([asm[0], "\tds_read_b64 v[110:111], v1"])
