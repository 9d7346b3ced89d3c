"""Idle-gap listing for the last WINDOW ms of a rocprofv3 kernel trace (csv)."""
import csv, sys, glob, collections
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
win = float(sys.argv[2]) * 1e6
thr = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 20e3
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:]) for r in csv.DictReader(open(path))]
rows.sort()
end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= end - win]
t0 = rows[0][0]
cur_e, prev = rows[0][1], rows[0][2]
gaps = []
for s, e, n in rows[1:]:
    if s > cur_e:
        gaps.append((s - cur_e, (cur_e - t0) / 1e6, prev, n))
    if e > cur_e:
        cur_e, prev = e, n
tot = sum(g[0] for g in gaps)
print("window %.1f ms, idle %.2f ms in %d gaps; gaps >= %.0f us: %.2f ms" % (win / 1e6, tot / 1e6, len(gaps), thr / 1e3, sum(g[0] for g in gaps if g[0] >= thr) / 1e6))
# cluster: contiguous regions where gaps are dense
for g in gaps:
    if g[0] >= thr:
        print("%8.1f us at %7.2f ms  after %-48s before %s" % (g[0] / 1e3, g[1], g[2], g[3]))
# idle per 2-ms bucket
b = collections.Counter()
for g in gaps:
    b[int(g[1] // 2)] += g[0]
print("idle per 2 ms bucket (us):")
print(" ".join("%d:%d" % (k * 2, v / 1e3) for k, v in sorted(b.items()) if v > 100e3))
