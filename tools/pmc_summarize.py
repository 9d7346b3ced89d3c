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

GROUPS = [  # (substring of the kernel name, group key); conv keys = the symbols bench.py's roofline names
    ("conv_bf16x3_v2_kernelILi256ELi16ELi1024ELi3", "conv_bf16x3_v2_kernel<256,16,1024,3>"),
    ("conv_bf16x3_v2_kernelILi256ELi16ELi512ELi3ELi1ELb1E", "conv_bf16x3_v2_kernel<256,16,512,3,1,true>"),
    ("conv_bf16x3_v2_kernelILi256ELi16ELi512ELi3", "conv_bf16x3_v2_kernel<256,16,512,3>"),
    ("conv_bf16x3_v2_kernelILi128ELi16ELi1024ELi3", "conv_bf16x3_v2_kernel<128,16,1024,3>"),
    ("conv_bf16x3_v2_kernelILi128ELi16ELi512ELi3", "conv_bf16x3_v2_kernel<128,16,512,3>"),
    ("conv_bf16x3_v2_kernelILi64ELi8ELi256ELi3", "conv_bf16x3_v2_kernel<64,8,256,3>"),
    ("conv_bf16x3_v2_kernelILi64ELi16ELi256ELi3", "conv_bf16x3_v2_kernel<64,16,256,3>"),
    ("gconv_taps_kernel", "gconv_taps"), ("gconv_gather_kernel", "gconv_gather"), ("gconv_bwd_kernel", "gconv_bwd"),
    ("weight_split_batched_kernel", "weight_split_batched"), ("sgd_multi_kernel", "sgd_multi"),
    ("conv_bf16x3_v2_kernelILi128ELi16ELi512ELi1", "conv_bf16x3_v2_kernel<128,16,512,1>"),
    ("conv_bf16x3_v2_kernelILi64ELi8ELi256ELi1", "conv_bf16x3_v2_kernel<64,8,256,1>"),
    ("conv3x3_bf16x3_kernelILi128ELi16ELi512ELi3", "conv3x3_bf16x3_fwd_dgrad_bn128"),
    ("conv3x3_bf16x3_kernelILi64ELi8ELi256ELi3", "conv3x3_bf16x3_fwd_bn64"),
    ("conv3x3_bf16x3_kernelILi128ELi16ELi512ELi1", "conv1x1_bf16x3_fwd_dgrad_bn128"),
    ("conv3x3_bf16x3_kernelILi64ELi8ELi256ELi1", "conv1x1_bf16x3_fwd_dgrad_bn64"),
    ("conv_wgrad_bf16x3_v6_kernel", "conv_wgrad_bf16x3_v6_kernel<3>"),
    ("conv_wgrad_bf16x3_v4_kernelILi3", "conv_wgrad_bf16x3_v4_kernel<3,1>"), ("conv_wgrad_bf16x3_v4_kernel<3", "conv_wgrad_bf16x3_v4_kernel<3,1>"),
    ("conv_wgrad_bf16x3_v4_kernelILi1", "conv_wgrad_bf16x3_v4_kernel<1,S>"), ("conv_wgrad_bf16x3_v4_kernel<1", "conv_wgrad_bf16x3_v4_kernel<1,S>"),
    ("fcos_assign_kernel", "fcos_assign"), ("fcos_compact_kernel", "fcos_compact"), ("fcos_nodes_kernel", "fcos_nodes"),
    ("upsample2x_add_kernel", "upsample2x_add"), ("downsample2x_sum_kernel", "downsample2x_sum"),
    ("conv_wgrad_bf16x3_v2_kernelILi3", "conv_wgrad_bf16x3_v2_kernel<3,1>"), ("conv_wgrad_bf16x3_v2_kernel<3", "conv_wgrad_bf16x3_v2_kernel<3,1>"),
    ("conv_wgrad_bf16x3_v2_kernelILi1", "conv_wgrad_bf16x3_v2_kernel<1,S>"), ("conv_wgrad_bf16x3_v2_kernel<1", "conv_wgrad_bf16x3_v2_kernel<1,S>"),
    ("conv3x3_wgrad_bf16x3_kernelILi3", "conv3x3_wgrad_bf16x3_kernel<3,1>"), ("conv3x3_wgrad_bf16x3_kernel<3", "conv3x3_wgrad_bf16x3_kernel<3,1>"),
    ("conv3x3_wgrad_bf16x3_kernelILi1", "conv3x3_wgrad_bf16x3_kernel<1,S>"), ("conv3x3_wgrad_bf16x3_kernel<1", "conv3x3_wgrad_bf16x3_kernel<1,S>"),
    ("conv_smallcin_kernel", "conv_smallcin_kernel"), ("dbscan_neighbors_kernel", "dbscan_neighbors"),
    ("slab_bias_reduce_kernel", "slab_bias_reduce"),
    ("conv_igemm_kernel<0", "conv_igemm_kernel<0,4>"), ("conv_igemm_kernel<1", "conv_igemm_kernel<1,4>"),
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
