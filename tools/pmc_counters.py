#!/usr/bin/env python3
"""Per-kernel averages of the counters of a rocprofv3 --pmc pass (--output-format csv).

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \\
        --output-format csv -d /tmp/pmc -- python3 tools/pointwise_roofline.py --big-only --only <kernels> --reps 3
    python tools/pmc_counters.py /tmp/pmc/*/*counter_collection.csv [name-substring ...]

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md); the ratios printed
are fractions of SQ_WAVE_CYCLES: wait_any = parked on s_waitcnt / barrier (memory latency), wait_inst = issue stalls,
active_valu = cycles a wave had a vector-ALU instruction executing."""
import collections
import csv
import gzip
import re
import sys


def short(name):
    """kernel name with its template arguments (the conv instances differ only there), without the parameter list"""
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else name[:40]


def main():
    path, subs = sys.argv[1], [a for a in sys.argv[2:] if "=" not in a]
    op = gzip.open if path.endswith(".gz") else open
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    with op(path, "rt") as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if subs and not any(s in k for s in subs):
                continue
            a = acc[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for k, cs in sorted(acc.items()):
        mean = {c: v[1] / v[0] for c, v in cs.items()}
        n = max(v[0] for v in cs.values())
        line = "%-44s %4d dispatches " % (k, n)
        wc = mean.get("SQ_WAVE_CYCLES")
        for c in sorted(mean):
            line += " %s=%.4g" % (c.replace("SQ_", ""), mean[c])
        if wc:
            for c, lab in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst"), ("SQ_ACTIVE_INST_VALU", "active_valu"),
                           ("SQ_ACTIVE_INST_ANY", "active_any")):
                if c in mean:
                    line += "  %s/wave_cycles=%.3f" % (lab, mean[c] / wc)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and len(sys.argv) > 2 and sys.argv[-1].startswith("waves_per_simd="):
                # MFMA_BUSY counts cycles per SIMD, WAVE_CYCLES quad-cycles per wave: with w resident waves per SIMD for the
                # whole launch, mfma_busy = MFMA_BUSY / (4 * WAVE_CYCLES / w)
                w = float(sys.argv[-1].split("=")[1])
                line += "  mfma_busy(at %g waves/SIMD)=%.3f" % (w, mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc / w))
            if "SQ_INSTS_VALU" in mean and "SQ_BUSY_CYCLES" in mean:
                line += "  valu_insts_per_busy_cycle=%.3f" % (mean["SQ_INSTS_VALU"] / mean["SQ_BUSY_CYCLES"])
        print(line)


if __name__ == "__main__":
    main()
