"""Prints the kernels of one training step between two marker kernels from a rocprofv3 --kernel-trace CSV, with the gap
in front of each (same queue or not): where the GPU idles inside the graph tier.

    python tools/trace_window.py <kernel_trace.csv> [first-marker-substring] [last-marker-substring]
"""
import csv
import sys


def main():
    path = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "gn_apply_kernel"
    last = sys.argv[3] if len(sys.argv) > 3 else "dynconv"
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"),
                         r.get("Stream_Id", "?")))
    rows.sort()
    # last occurrence of the `last` marker, and the closest `first` marker in front of it that follows a big conv
    idx_last = max(i for i, r in enumerate(rows) if last in r[2] and "bwd" not in r[2])
    i0 = idx_last
    while i0 > 0 and not (first in rows[i0][2]):
        i0 -= 1
    t_prev_end = rows[i0][1]
    busy_until = t_prev_end
    print("window: %d kernels, %.1f us" % (idx_last - i0, (rows[idx_last][0] - rows[i0][1]) / 1e3))
    idle = 0.0
    for s, e, name, q, st in rows[i0:idx_last + 1]:
        gap = (s - busy_until) / 1e3
        if gap > 0:
            idle += gap
        print("%8.1f us gap  %8.1f us  q%s s%s  %s" % (gap, (e - s) / 1e3, q, st, name[:90]))
        busy_until = max(busy_until, e)
    print("idle inside the window: %.1f us" % idle)


if __name__ == "__main__":
    main()
