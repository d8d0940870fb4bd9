"""Compile every kernel file to gfx950 assembly and report, per kernel, two patterns that cost a trip to the L2 each and are
invisible in the source (found in round 3 with in-kernel stamps, tools/tr_stamps.py):
  * runs of `global_load -> s_waitcnt vmcnt(0)` pairs: a conditional load (`ok ? p[i] : 0.0`) is sunk into an exec-mask
    branch of its own and waited for at the merge; a kernel at its register limit requests, waits and uses one element at a
    time; a load followed by an LDS store through a generic pointer is waited for on the spot;
  * flat_load / flat_store: a device pointer whose address space the compiler does not know (argument of a __noinline__
    function, member of a by-value struct, run-time choice between two LDS pointers) - counts in vmcnt AND lgkmcnt.
Usage: python tools/scan_isa.py [min_run]"""
import os, re, subprocess, sys, tempfile
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "montecarlo.jl_amd", "csrc")
min_run = int(sys.argv[1]) if len(sys.argv) > 1 else 4
flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-mllvm",
         "-pragma-unroll-threshold=4000000", "-S", "--cuda-device-only"]
for f in sorted(os.listdir(root)):
    if f.endswith(".hip") and len(sys.argv) > 2 and f not in sys.argv[2:]:
        continue
    if not f.endswith(".hip"):
        continue
    out = os.path.join(tempfile.gettempdir(), f + ".s")
    subprocess.run(["hipcc"] + flags + [os.path.join(root, f), "-o", out], check=True, stderr=subprocess.DEVNULL)
    kern, seq, flat = None, [], 0
    for l in open(out):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern, seq, flat = m.group(1), [], 0
            continue
        t = l.strip()
        if t.startswith(("global_load", "flat_load", "buffer_load")):
            seq.append("L")
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            seq.append("W")
        if t.startswith(("flat_load", "flat_store", "flat_atomic")):
            flat += 1
        if t.startswith(".Lfunc_end") and kern:
            runs = [len(r) // 2 for r in re.findall(r"(?:LW){%d,}" % min_run, "".join(seq))]
            if runs or flat > 4:
                name = subprocess.run(["c++filt", kern], capture_output=True, text=True).stdout.strip()[:100]
                print("%-12s %-100s load->wait(0) runs %s, flat ops %d" % (f, name, runs, flat))
            kern = None
