#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --output-format csv).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python tools/pmc_summarize.py <fetch counter_collection.csv[.gz]> <write ...csv[.gz]> > profiles/rNN_pmc_traffic.json

Counter unit is KiB.  Per /opt/skills/guides/MI355X_MICROARCH.md FETCH_SIZE under-reports wide coalesced reads by
2x on gfx950, so hbm_read_bytes = 2 * FETCH_SIZE * 1024 (checked on the dynamic-conv forward: 89.6 MB counted vs
89.4 MB algorithmic).  Values are averages per launch over all launches of the kernel group in the run."""
import collections
import csv
import gzip
import json
import sys

def _conv_groups():
    """(substring, key) for the template instances of the split-operand conv kernels, in their mangled and demangled spelling;
    keys = the symbols bench.py's roofline names"""
    out = []
    for np_ in (3, 2):
        for bn, th, nt, ks, tpb, gl in ((256, 16, 512, 3, 1, True), (256, 16, 1024, 3, 1, True), (256, 16, 1024, 3, 1, False),
                                        (256, 16, 512, 3, 1, False), (128, 16, 512, 3, 1, True), (128, 16, 512, 3, 1, False),
                                        (128, 16, 1024, 3, 3, True), (128, 16, 1024, 3, 3, False), (128, 16, 1024, 3, 1, True),
                                        (128, 16, 1024, 3, 1, False), (128, 16, 512, 3, 3, False), (64, 16, 512, 3, 1, True),
                                        (64, 8, 256, 3, 1, True), (64, 16, 512, 3, 1, False), (64, 16, 256, 3, 1, False),
                                        (64, 8, 256, 3, 3, False), (64, 8, 256, 3, 1, False), (128, 16, 512, 1, 1, False),
                                        (64, 8, 256, 1, 1, False)):
            key = "conv_split_kernel<%d,%d,%d,%d,%d%s>" % (np_, bn, th, nt, ks, ",%d,true" % tpb if gl else ("" if tpb == 1 else ",%d" % tpb))
            out.append(("conv_split_kernelILi%dELi%dELi%dELi%dELi%dELi%dELb%dEE" % (np_, bn, th, nt, ks, tpb, int(gl)), key))
            out.append(("conv_split_kernel<%d, %d, %d, %d, %d, %d, %s>" % (np_, bn, th, nt, ks, tpb, "true" if gl else "false"), key))
        wk = 32 if np_ == 3 else 64
        for tom, tcw in ((2, 4), (4, 2)):
            key = "conv_wgrad_v6_kernel<%d,%d,3>" % (np_, wk)
            out.append(("conv_wgrad_v6_kernelILi%dELi%dELi3ELi%dELi%dEE" % (np_, wk, tom, tcw), key))
            out.append(("conv_wgrad_v6_kernel<%d, %d, 3, %d, %d>" % (np_, wk, tom, tcw), key))
        out.append(("conv_wgrad_v4_kernelILi%dELi3E" % np_, "conv_wgrad_v4_kernel<%d,3,1>" % np_))
        out.append(("conv_wgrad_v4_kernel<%d, 3" % np_, "conv_wgrad_v4_kernel<%d,3,1>" % np_))
        out.append(("conv_wgrad_v4_kernelILi%dELi1E" % np_, "conv_wgrad_v4_kernel<%d,1,S>" % np_))
        out.append(("conv_wgrad_v4_kernel<%d, 1" % np_, "conv_wgrad_v4_kernel<%d,1,S>" % np_))
        out.append(("conv_smallcin_kernelILi%dE" % np_, "conv_smallcin_kernel<%d>" % np_))
        out.append(("conv_smallcin_kernel<%d," % np_, "conv_smallcin_kernel<%d>" % np_))
    return out


GROUPS = _conv_groups() + [  # (substring of the kernel name, group key)
    ("gconv_taps_kernel", "gconv_taps"), ("gconv_gather_kernel", "gconv_gather"), ("gconv_bwd_kernel", "gconv_bwd"),
    ("weight_split_batched_kernel", "weight_split_batched"), ("sgd_multi_kernel", "sgd_multi"),
    ("conv3x3_bf16x3_kernelILi256ELi16ELi512ELi3", "conv3x3_bf16x3_kernel<256,16,512>"),
    ("conv3x3_bf16x3_kernelILi128ELi16ELi512ELi3", "conv3x3_bf16x3_kernel<128,16,512>"),
    ("conv3x3_bf16x3_kernelILi64ELi8ELi256ELi3", "conv3x3_bf16x3_kernel<64,8,256>"),
    ("fcos_assign_kernel", "fcos_assign"), ("fcos_compact_kernel", "fcos_compact"), ("fcos_nodes_kernel", "fcos_nodes"),
    ("upsample2x_add_kernel", "upsample2x_add"), ("downsample2x_sum_kernel", "downsample2x_sum"),
    ("dbscan_neighbors_kernel", "dbscan_neighbors"), ("slab_bias_reduce_kernel", "slab_bias_reduce"),
    ("conv_igemm_kernel<0", "conv_igemm_kernel<0,4>"), ("conv_igemm_kernel<1", "conv_igemm_kernel<1,4>"),
    ("conv_igemm_kernelILi0", "conv_igemm_kernel<0,4>"), ("conv_igemm_kernelILi1", "conv_igemm_kernel<1,4>"),
    ("conv_wgrad_kernel", "conv_wgrad_kernel"),
    ("gn_stats_kernel", "gn_stats"), ("gn_apply_kernel", "gn_apply"), ("gn_bwd_reduce_kernel", "gn_bwd_reduce"),
    ("gn_bwd_apply_kernel", "gn_bwd_apply"), ("relu_bwd_kernel", "relu_bwd"),
    ("maxpool2_fwd_kernel", "maxpool2_fwd"), ("maxpool2_bwd_kernel", "maxpool2_bwd"),
    ("dynconv_fwd_kernel", "dynconv_fwd"), ("dynconv_bwd_kernel", "dynconv_bwd"),
    ("focal_fwd_kernel", "sigmoid_focal_fwd"), ("focal_bwd_kernel", "sigmoid_focal_bwd"),
    ("sfl_kernelILb0", "softmax_focal_fwd"), ("sfl_kernel<false", "softmax_focal_fwd"),
    ("sfl_kernelILb1", "softmax_focal_bwd"), ("sfl_kernel<true", "softmax_focal_bwd"),
    ("cka_fwd", "cka_bce_fwd"), ("cka_bwd_kernel", "cka_bce_bwd"), ("scale_kernel", "grl_scale"),
    ("sgd_kernel", "sgd_momentum"), ("weight_split_kernel", "weight_split"),
]


def read(path, counter):
    op = gzip.open if path.endswith(".gz") else open
    acc = collections.defaultdict(lambda: [0, 0.0])
    with op(path, "rt") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            for sub, key in GROUPS:
                if sub in name:
                    acc[key][0] += 1
                    acc[key][1] += float(r["Counter_Value"])
                    break
    return acc


def meta():
    """what the passes are valid for: the commit checked out (if a git tree is here) and the hash of the kernel sources
    (bench.py reports the traffic only while that hash matches the build it runs on)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    commit = os.environ.get("SCAN_COMMIT")
    if not commit:
        try:
            commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        except Exception:
            commit = None
    return {"commit": commit, "csrc_sha1": bench.csrc_sha1()}


def main():
    fe, wr = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
    out = {"_note": __doc__.split("\n\n")[-1].replace("\n", " "), "_meta": meta()}
    for _, key in GROUPS:
        if key not in fe or key not in wr or key in out:
            continue
        n = fe[key][0]
        f_kib, w_kib = fe[key][1] / n, wr[key][1] / max(1, wr[key][0])
        rd, wb = int(2 * f_kib * 1024), int(w_kib * 1024)
        out[key] = {"launches": n, "FETCH_SIZE_KiB_raw": round(f_kib, 1), "WRITE_SIZE_KiB": round(w_kib, 1),
                    "hbm_read_bytes": rd, "hbm_write_bytes": wb, "hbm_bytes": rd + wb}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
