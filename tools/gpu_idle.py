"""GPU idle time inside the training step, from a rocprofv3 --kernel-trace csv.

    rocprofv3 --kernel-trace -d out --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pointwise
    python tools/gpu_idle.py out/*/*_kernel_trace.csv

Steps are delimited by the last `sgd_kernel` launch of each iteration.  For every step: wall span, the
union of all kernel intervals (busy), idle = span - busy, and the largest gaps with the kernels either side, so a
gap can be attributed (host synchronisation point, host-bound launch sequence, stream join).
"""
import csv
import json
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    sgd_end = [e for s, e, n in rows if n.startswith("sgd_kernel")]
    # group consecutive sgd launches (one per sub-model) into one boundary per step: a new step starts when the
    # distance to the previous sgd launch exceeds 5 ms
    bounds = []
    for e in sgd_end:
        if not bounds or e - bounds[-1] > 5e6:
            bounds.append(e)
        else:
            bounds[-1] = e
    report = []
    gap_by_pair = defaultdict(lambda: [0, 0.0])
    for k in range(1, len(bounds)):
        lo, hi = bounds[k - 1], bounds[k]
        ks = [(s, e, n) for s, e, n in rows if s >= lo and e <= hi]
        if not ks:
            continue
        busy, cur_s, cur_e, cur_n = 0, ks[0][0], ks[0][1], ks[0][2]
        gaps = [(ks[0][0] - lo, "<step boundary>", ks[0][2])]
        for s, e, n in ks[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                gaps.append((s - cur_e, cur_n, n))
                cur_s, cur_e, cur_n = s, e, n
            elif e > cur_e:
                cur_e, cur_n = e, n
        busy += cur_e - cur_s
        span = hi - lo
        gaps.sort(reverse=True)
        for g, a, b in gaps:
            if g > 20000:
                key = (a[:60], b[:60])
                gap_by_pair[key][0] += 1
                gap_by_pair[key][1] += g / 1e3
        report.append({"step": k, "span_ms": round(span / 1e6, 2), "busy_ms": round(busy / 1e6, 2),
                       "idle_ms": round((span - busy) / 1e6, 2), "kernels": len(ks),
                       "gaps_over_20us": sum(1 for g in gaps if g[0] > 20000),
                       "idle_in_gaps_over_20us_ms": round(sum(g[0] for g in gaps if g[0] > 20000) / 1e6, 2),
                       "top_gaps_us": [[round(g / 1e3, 1), a[:70], b[:70]] for g, a, b in gaps[:top]]})
    pairs = sorted(gap_by_pair.items(), key=lambda kv: -kv[1][1])[:25]
    print(json.dumps({"steps": report,
                      "gap_pairs_all_steps(us_total,count,before,after)":
                          [[round(v[1], 1), v[0], k[0], k[1]] for k, v in pairs]}, indent=1))


if __name__ == "__main__":
    main()
