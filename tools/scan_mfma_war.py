"""MFMA source-C write-after-read: scan and patch (round 4).

v_mfma_f64_16x16x4_f64 reads its accumulator input (src C) pass by pass over the 64 cycles it occupies the matrix pipe.
When the compiler accumulates OUT OF PLACE (vdst != src C: it renames an accumulator instead of copying it) the old src C
registers are free as soon as the instruction has issued, and a ds_read / global_load / scratch_load issued right behind it
may be given those registers as its destination.  The load's data return is asynchronous; if it lands before the matrix
pipe has read the last registers of src C, the product is computed from the loaded bytes.  Seen on gfx950 / ROCm 7.2 in
qrb.hip (`v_mfma v[90:97], .., .., v[98:105]` followed by `ds_read2_b64 v[104:107]`: rows 4 r + kq with r = 3 of one
accumulator tile wrong, differently from run to run).  The compiler's hazard recogniser covers VALU writes behind an MFMA,
not memory return data.

A pair counts as a hazard when fewer than SAFE_MFMAS other MFMAs lie between the two instructions (a later MFMA is accepted
by the pipe only when the earlier one has issued all its passes) and fewer than SAFE_CYCLES cycles of s_nop.
  scan:   python tools/scan_mfma_war.py file.s [...]          exit code 1 if a hazard is found
  patch:  python tools/scan_mfma_war.py --patch in.s out.s    inserts s_nop in front of the load of every hazard
The build patches the assembly of the kernel files that show the pattern (csrc/Makefile) and scans the result;
tests/test_isa_guard.py scans the disassembly of the shipped library."""
import re, sys

WINDOW = 64          # instructions looked at behind an out-of-place MFMA
SAFE_MFMAS = 2
SAFE_CYCLES = 80
rng = re.compile(r"([va])\[(\d+):(\d+)\]|([va])(\d+)")
LOADS = ("ds_read", "ds_load", "global_load", "scratch_load", "buffer_load", "flat_load")


def regs(tok):
    m = rng.fullmatch(tok.strip())
    if not m:
        return None
    if m.group(1):
        return m.group(1), int(m.group(2)), int(m.group(3))
    return m.group(4), int(m.group(5)), int(m.group(5))


def overlap(a, b):
    return a and b and a[0] == b[0] and a[1] <= b[2] and b[1] <= a[2]


def instructions(lines):
    out = []
    for i, l in enumerate(lines):
        t = l.strip()
        if not t or t.startswith((".", ";", "//")) or t.endswith(":") or not (l.startswith("\t") or l.startswith(" ")):
            continue
        out.append((i, t))
    return out


def nop_cycles(t):
    m = re.match(r"s_nop\s+(\d+)", t)
    return int(m.group(1)) + 1 if m else 0


def hazards(lines):
    """[(mfma line, load line, cycles missing)]"""
    ins = instructions(lines)
    hits = []
    for k, (ln, t) in enumerate(ins):
        if not t.startswith("v_mfma"):
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        if len(ops) < 4:
            continue
        dst, srcc = regs(ops[0]), regs(ops[3].split()[0])
        if not srcc or not dst or dst == srcc:
            continue
        mfmas, cyc = 0, 0
        for ln2, t2 in ins[k + 1:k + 1 + WINDOW]:
            if mfmas >= SAFE_MFMAS or cyc >= SAFE_CYCLES:
                break
            if t2.startswith("v_mfma"):
                mfmas += 1
                continue
            if t2.startswith(("s_endpgm", "s_setpc", "s_branch", "s_cbranch")):
                break  # (control flow: the allocator does not reuse a register across it for a load in flight)
            cyc += nop_cycles(t2)
            if t2.startswith(LOADS):
                d2 = regs(t2.split(None, 1)[1].split(",")[0])
                if overlap(d2, srcc) and not overlap(d2, dst):
                    hits.append((ln, ln2, SAFE_CYCLES - cyc))
    return hits


def patch(lines):
    need = {}
    for ln, ln2, missing in hazards(lines):
        need[ln2] = max(need.get(ln2, 0), missing)
    out = []
    for i, l in enumerate(lines):
        if i in need:
            c = need[i]
            while c > 0:
                n = min(c, 16)
                out.append("\ts_nop %d ; mfma src C write-after-read guard (tools/scan_mfma_war.py)" % (n - 1))
                c -= n
        out.append(l)
    return out, len(need)


if __name__ == "__main__":
    if sys.argv[1] == "--patch":
        lines = open(sys.argv[2]).read().split("\n")
        out, n = patch(lines)
        left = hazards(out)
        open(sys.argv[3], "w").write("\n".join(out))
        print("%s: %d loads guarded, %d hazards left" % (sys.argv[2], n, len(left)))
        sys.exit(1 if left else 0)
    bad = 0
    for p in sys.argv[1:]:
        lines = open(p).read().split("\n")
        for ln, ln2, missing in hazards(lines):
            bad += 1
            print("%s:%d: %s\n    -> %d: %s" % (p, ln + 1, lines[ln].strip(), ln2 + 1, lines[ln2].strip()))
    print("%d MFMA src-C write-after-read hazards" % bad)
    sys.exit(1 if bad else 0)
