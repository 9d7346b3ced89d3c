"""GPU idle time inside the training step, from a rocprofv3 --kernel-trace csv.

    rocprofv3 --kernel-trace -d out --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pointwise
    python tools/gpu_idle.py out/*/*_kernel_trace.csv

Steps are delimited by the last `sgd_kernel` / `sgd_multi_kernel` launch of each iteration.  For every step: wall span, the
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
    sgd_end = [e for s, e, n in rows if n.startswith("sgd_kernel") or n.startswith("sgd_multi_kernel")]
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
    exposed_by_name = defaultdict(lambda: [0, 0.0, 0.0])
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
        # exposed small-kernel time: the union of the intervals of kernels shorter than 20 us minus the union of the
        # longer ones -- stretches of the step where the GPU runs nothing but host-glue kernels
        def union(iv):
            out, cs, ce = [], None, None
            for s_, e_ in sorted(iv):
                if cs is None:
                    cs, ce = s_, e_
                elif s_ <= ce:
                    ce = max(ce, e_)
                else:
                    out.append((cs, ce))
                    cs, ce = s_, e_
            if cs is not None:
                out.append((cs, ce))
            return out
        small = union([(s_, e_) for s_, e_, n_ in ks if e_ - s_ < 20000])
        big = union([(s_, e_) for s_, e_, n_ in ks if e_ - s_ >= 20000])
        exposed, bi = 0, 0
        for s_, e_ in small:
            cur = s_
            while bi < len(big) and big[bi][1] <= cur:
                bi += 1
            j = bi
            while cur < e_:
                if j >= len(big) or big[j][0] >= e_:
                    exposed += e_ - cur
                    break
                if big[j][0] > cur:
                    exposed += big[j][0] - cur
                cur = max(cur, big[j][1])
                j += 1
        small_total = sum(e_ - s_ for s_, e_, n_ in ks if e_ - s_ < 20000)
        # the same per kernel name (each small kernel against the long-kernel union; overlaps among small ones count
        # once per kernel, so these add up to a little more than the exposed total)
        bj = 0
        for s_, e_, n_ in ks:
            if e_ - s_ >= 20000:
                continue
            while bj < len(big) and big[bj][1] <= s_:
                bj += 1
            cur, j, ex = s_, bj, 0
            while cur < e_:
                if j >= len(big) or big[j][0] >= e_:
                    ex += e_ - cur
                    break
                if big[j][0] > cur:
                    ex += big[j][0] - cur
                cur = max(cur, big[j][1])
                j += 1
            rec = exposed_by_name[n_[:80]]
            rec[0] += 1
            rec[1] += (e_ - s_) / 1e3
            rec[2] += ex / 1e3
        gaps.sort(reverse=True)
        for g, a, b in gaps:
            if g > 20000:
                key = (a[:60], b[:60])
                gap_by_pair[key][0] += 1
                gap_by_pair[key][1] += g / 1e3
        report.append({"step": k, "span_ms": round(span / 1e6, 2), "busy_ms": round(busy / 1e6, 2),
                       "idle_ms": round((span - busy) / 1e6, 2), "kernels": len(ks),
                       "kernels_under_20us": sum(1 for s_, e_, n_ in ks if e_ - s_ < 20000),
                       "small_kernel_sum_ms": round(small_total / 1e6, 2),
                       "small_kernel_exposed_ms": round(exposed / 1e6, 2),
                       "gaps_over_20us": sum(1 for g in gaps if g[0] > 20000),
                       "idle_in_gaps_over_20us_ms": round(sum(g[0] for g in gaps if g[0] > 20000) / 1e6, 2),
                       "top_gaps_us": [[round(g / 1e3, 1), a[:70], b[:70]] for g, a, b in gaps[:top]]})
    pairs = sorted(gap_by_pair.items(), key=lambda kv: -kv[1][1])[:25]
    nsteps = max(len(report), 1)
    names = sorted(exposed_by_name.items(), key=lambda kv: -kv[1][2])[:40]
    print(json.dumps({"steps": report,
                      "small_kernels_per_step(calls,sum_us,exposed_us,name)":
                          [[round(v[0] / nsteps, 1), round(v[1] / nsteps, 1), round(v[2] / nsteps, 1), k] for k, v in names],
                      "gap_pairs_all_steps(us_total,count,before,after)":
                          [[round(v[1], 1), v[0], k[0], k[1]] for k, v in pairs]}, indent=1))


if __name__ == "__main__":
    main()
