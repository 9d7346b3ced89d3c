"""Summarise a rocprofv3 kernel trace csv: per-kernel totals over the last WINDOW ms + idle analysis."""
import csv, sys, glob, collections
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 380e6
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= end - win]
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in rows:
    n = n.split("(")[0][:60]
    tot[n] += e - s; cnt[n] += 1
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("window ms %.1f union-busy %.1f%% sum-kernels ms %.1f n=%d" % (win / 1e6, 100 * busy / win, sum(tot.values()) / 1e6, len(rows)))
for n, t in tot.most_common(25):
    print("%9.2f ms %6d  %s" % (t / 1e6, cnt[n], n))
