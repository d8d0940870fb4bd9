"""kernel-trace CSV -> compact timeline (start us, end us, queue, kernel) + how much device time had kernels of two
queues in flight at once"""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]) for r in rows)
t0 = ev[0][0]
with open(sys.argv[2], "w") as o:
    for s, e, q, nm in ev:
        nm = nm.replace("void ", "").replace("dqmc::", "")
        o.write("%.1f %.1f %s %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, q, nm[:48].replace(" ", "_")))
print("kernels", len(ev))
