"""per-kernel table (launches, mean us) of ONE sweep of config 3 from a rocprofv3 kernel-trace timeline written by
tools/trace_overlap_parse.py"""
import collections, sys, statistics
ev = [l.split() for l in open(sys.argv[1])]
sw = [i for i, e in enumerate(ev) if e[3].startswith('sweep_lu4')]
first = sw[len(sw) - 160] if len(sw) >= 160 else sw[0]
# one sweep = from the wrap before that elimination; approximate: start at `first`
d = collections.defaultdict(list)
for a, b, q, n in ev[first:]:
    d[n[:64]].append(float(b) - float(a))
tot = 0
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("%8.2f ms  %5d x %7.1f us  %s" % (sum(v) / 1e3, len(v), sum(v) / len(v), n)); tot += sum(v)
q1 = [(float(a), float(b)) for a, b, q, n in ev[first:]]
gaps = [q1[i + 1][0] - q1[i][1] for i in range(len(q1) - 1)]
print("kernel time %.2f ms; gaps %.2f ms (median %.2f us, %d above 3 us); span %.2f ms" % (tot / 1e3, sum(g for g in gaps if g > 0) / 1e3,
      statistics.median(gaps), sum(1 for g in gaps if g > 3), (q1[-1][1] - q1[0][0]) / 1e3))
