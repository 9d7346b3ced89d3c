import json, gzip, collections, sys
d = json.load(gzip.open('gpurun_out/trace.json.gz'))
ev = [e for e in d['traceEvents'] if e.get('ph') == 'X']
cats = collections.Counter(e.get('cat') for e in ev); print(cats)
k = sorted([e for e in ev if e.get('cat') == 'kernel'], key=lambda e: e['ts'])
cpu = sorted([e for e in ev if e.get('cat') in ('cpu_op', 'user_annotation', 'python_function')], key=lambda e: e['ts'])
rt = sorted([e for e in ev if e.get('cat') in ('cuda_runtime', 'cuda_driver')], key=lambda e: e['ts'])
t0 = k[0]['ts']; tend = max(e['ts'] + e['dur'] for e in k)
print('kernels', len(k), 'span ms', (tend - t0) / 1e3, 'sum ms', sum(e['dur'] for e in k) / 1e3)
# launch -> kernel correlation
by_corr = {}
for e in rt:
    c = e.get('args', {}).get('correlation')
    if c is not None: by_corr[c] = e
# gaps
cur = k[0]['ts'] + k[0]['dur']; prev = k[0]
gaps = []
for e in k[1:]:
    if e['ts'] > cur:
        gaps.append((e['ts'] - cur, cur, prev, e))
    if e['ts'] + e['dur'] > cur:
        cur = e['ts'] + e['dur']; prev = e
print('idle ms', sum(g[0] for g in gaps) / 1e3, 'n', len(gaps))
import bisect
cpu_ts = [e['ts'] for e in cpu]
def cpu_at(ts):
    # innermost-ish cpu ops active at ts (by thread)
    out = []
    i = bisect.bisect_right(cpu_ts, ts)
    for e in cpu[max(0, i - 400):i]:
        if e['ts'] <= ts <= e['ts'] + e['dur']:
            out.append(e['name'][:40])
    return out[-4:]
b = collections.Counter()
for g in gaps: b[int((g[1] - t0) // 2000)] += g[0]
print(' '.join('%d:%d' % (kk * 2, v) for kk, v in sorted(b.items()) if v > 150))
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 40
for g in gaps:
    if g[0] >= thr:
        nxt = g[3]; c = nxt.get('args', {}).get('correlation'); l = by_corr.get(c)
        lat = (nxt['ts'] - l['ts']) if l else -1
        print('%6.0f us at %6.2f ms  after %-28s before %-28s launch->start %5.0f us  cpu@gapstart: %s' % (
            g[0], (g[1] - t0) / 1e3, g[2]['name'].split('(')[0][-28:], nxt['name'].split('(')[0][-28:], lat, cpu_at(g[1])))
