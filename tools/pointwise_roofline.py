#!/usr/bin/env python
"""HBM roofline of the pointwise / reduction kernels of the SCAN hot path (north star: "rocprof HBM GB/s
for the pointwise kernels against CDNA4 peak").

Every kernel is launched through the C ABI on pre-allocated device buffers and timed with HIP events on the launch
stream, at two sizes: M = 2^24 rows (the microbenchmark SURVEY 8d prescribes; working sets far beyond the 256 MB
last-level cache) and M = 601,608 rows = the 8 frames per GPU of BASELINE.json configs[4] (1333x2666 padded to
1344x2688: 75,201 locations per frame).  achieved = ALGORITHMIC bytes (the formula is printed with every record: what
a perfectly fused kernel must read and write once) / average launch time; peak = 8 TB/s (MI355X_MICROARCH.md).

    python tools/pointwise_roofline.py [--out profiles/r02_pointwise_roofline.json] [--small]

bench.py imports measure() for the `roofline_pointwise` object of its JSON line; `rocprofv3 --kernel-trace --stats`
of this script gives the matching per-kernel averages (profiles/r02_pointwise_kernel_stats.csv).
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0
M_CFG5 = 8 * 75201  # 8 frames/GPU at 1344x2688: levels 168x336, 84x168, 42x84, 21x42, 11x21
M_BIG = 1 << 24


def _time(fn, reps, torch):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(torch.cuda.current_stream())
    for _ in range(reps):
        fn()
    e.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps  # us per launch


def measure(dev, sizes=(M_CFG5, M_BIG), reps=10, K=9, only=None):
    import torch
    from scan_amd import ops
    from scan_amd._lib import call, query
    C = K - 1
    P = ops._ptr
    out = []

    def st():
        return ops._stream()

    def rec(name, M, nbytes, formula, fn, unit="row"):
        if only and name not in only:
            return
        us = _time(fn, reps, torch)
        gbs = nbytes / us * 1e-3
        out.append({"kernel": name, "M": int(M), "bytes": int(nbytes), "bytes_per_%s" % unit: round(nbytes / M, 2),
                    "formula": formula, "us": round(us, 2), "GBps": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4),
                    "fits_llc": bool(nbytes < 256e6)})

    g = torch.Generator(device=dev).manual_seed(0)
    for M in sizes:
        # ---- sigmoid focal loss (a13): logits [M,C], targets int32 [M]
        x = (torch.randn((M, C), device=dev, generator=g) * 3 - 2)
        t = torch.randint(-1, C + 1, (M,), device=dev, generator=g, dtype=torch.int32)
        s1 = torch.zeros(1, device=dev)
        d = torch.empty_like(x)
        rec("sigmoid_focal_fwd_sum", M, 4 * (M * C + M), "4(MC+M): logits + targets read, sum in registers",
            lambda: call("scan_sigmoid_focal_loss_forward", P(x), P(t), M, C, 2.0, 0.25, None, P(s1), st()))
        rec("sigmoid_focal_fwd_elem", M, 4 * (2 * M * C + M), "4(2MC+M): + element-wise losses written (the _C entry point)",
            lambda: call("scan_sigmoid_focal_loss_forward", P(x), P(t), M, C, 2.0, 0.25, P(d), None, st()))
        rec("sigmoid_focal_bwd", M, 4 * (2 * M * C + M), "4(2MC+M): logits + targets read, d_logits written (scalar upstream grad)",
            lambda: call("scan_sigmoid_focal_loss_backward", P(x), P(t), None, 0.5, M, C, 2.0, 0.25, P(d), st()))
        # ---- softmax focal act loss (a9): logits [M,K], labels int64 [M]
        z = torch.randn((M, K), device=dev, generator=g) * 2
        lab = torch.randint(0, K, (M,), device=dev, generator=g)
        dz = torch.empty_like(z)
        rec("softmax_focal_fwd", M, M * (4 * K + 8), "M(4K+8): logits + int64 labels read",
            lambda: call("scan_softmax_focal_forward", P(z), P(lab), M, K, 2.0, P(s1), st()))
        rec("softmax_focal_bwd", M, M * (8 * K + 8), "M(8K+8): + d_logits written",
            lambda: call("scan_softmax_focal_backward", P(z), P(lab), M, K, 2.0, 1.0 / M, P(dz), st()))
        # ---- CKA class-weighted BCE (a15): logits [M,C], act maps [M,K]
        act = torch.softmax(z, 1).contiguous()
        o2 = torch.zeros(2 * C, device=dev)
        gd = torch.rand(C, device=dev)
        rec("cka_bce_fwd", M, 4 * M * (C + K), "4M(C+K): logits + act maps read",
            lambda: call("scan_cka_bce_forward", P(x), P(act), M, C, 1.0, P(o2), st()))
        rec("cka_bce_bwd", M, 4 * M * (2 * C + K), "4M(2C+K): + d_logits written",
            lambda: call("scan_cka_bce_backward", P(x), P(act), M, C, 1.0, P(gd), P(d), st()))
        del x, t, d, z, lab, dz, act
        # ---- IoU loss (a14) / centerness BCE on P positives (P = M here: the kernels do not care)
        pr = torch.rand((M, 4), device=dev, generator=g) * 60 + 0.5
        tg = torch.rand((M, 4), device=dev, generator=g) * 60 + 0.5
        w = torch.rand((M,), device=dev, generator=g)
        o = torch.zeros(2, device=dev)
        dp = torch.empty_like(pr)
        gn = torch.ones(1, device=dev)
        rec("iou_loss_fwd", M, 36 * M, "36P: pred + target (16 B each) + weight read", unit="positive",
            fn=lambda: call("scan_iou_loss_forward", P(pr), P(tg), P(w), M, P(o), st()))
        rec("iou_loss_bwd", M, 52 * M, "52P: + d_pred written", unit="positive",
            fn=lambda: call("scan_iou_loss_backward", P(pr), P(tg), P(w), M, P(gn), P(dp), st()))
        rec("bce_logits_fwd", M, 8 * M, "8P: logits + targets read", unit="positive",
            fn=lambda: call("scan_bce_logits_forward", P(w), P(w), 0.0, None, 0, M, P(o), st()))
        del pr, tg, dp
        # ---- GRL / SGD on flat buffers
        n = M * 16
        a = torch.randn(n, device=dev, generator=g)
        b = torch.empty_like(a)
        rec("grl_scale", n, 8 * n, "8 B/element: read + write", unit="element",
            fn=lambda: call("scan_scale", P(a), -0.02, P(b), n, st()))
        m_ = torch.zeros_like(a)
        rec("sgd_momentum", n, 20 * n, "20 B/parameter: p, g, m read; p, m written", unit="element",
            fn=lambda: call("scan_sgd_momentum", P(a), P(b), P(m_), n, 1e-9, 1e-4, 0.9, 0, st()))
        del a, b, m_
        # ---- dynamic conv + softmax (a4): feat [M,256], kernels [K,256]
        feat = torch.randn((M, 256), device=dev, generator=g)
        kern = torch.randn((K, 256), device=dev, generator=g) * 0.05
        lg = torch.empty((M, K), device=dev)
        pb = torch.empty((M, K), device=dev)
        rec("dynconv_softmax_fwd", M, M * (1024 + 8 * K), "M(1024+8K): features read, logits + probabilities written",
            lambda: call("scan_dynconv_softmax_forward", P(feat), P(kern), M, 256, K, P(lg), P(pb), st()))
        dl = torch.randn((M, K), device=dev, generator=g)
        dpb = torch.randn((M, K), device=dev, generator=g)
        dfe = torch.empty_like(feat)
        dk = torch.empty_like(kern)
        ws = torch.empty((query("scan_dynconv_ws_floats", M, 256, K),), device=dev)
        rec("dynconv_softmax_bwd", M, M * (2048 + 12 * K), "M(2048+12K): features, probs, d_logits, d_probs read; d_features written",
            lambda: call("scan_dynconv_softmax_backward", P(feat), P(kern), P(pb), P(dl), P(dpb), M, 256, K, P(dfe),
                         P(dk), P(ws), st()))
        del lg, pb, dl, dpb, dk, ws
        # ---- GroupNorm(32)+ReLU on a one-level pyramid of 8 images (a3/a11/a15 towers)
        n_img = 8
        hw = M // n_img
        h = 1
        while h * h * 2 < hw:
            h += 1
        wd_ = hw // h
        Mg = n_img * h * wd_
        shape = ops.PyramidShape(n_img, [(h, wd_)])
        gam, bet = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev)
        stats = torch.empty((n_img * 64,), device=dev)
        wsg = torch.empty((query("scan_groupnorm_ws_floats", shape.ref(), 256, 32) // 2 + 1,), dtype=torch.float64, device=dev)
        xg, yg = feat[:Mg], dfe[:Mg]
        rec("groupnorm_stats", Mg, 1024 * Mg, "1024M: x read (towers get the sums from the conv epilogue instead)",
            lambda: call("scan_groupnorm_stats", P(xg), shape.ref(), 256, 32, 1e-5, P(stats), P(wsg), st()))
        rec("groupnorm_relu_apply", Mg, 2048 * Mg, "2048M: x read, y written",
            lambda: call("scan_groupnorm_relu_forward", P(xg), shape.ref(), 256, 32, P(stats), P(gam), P(bet), 1, P(yg), st()))
        dyg = torch.randn((Mg, 256), device=dev, generator=g)
        dxg = torch.empty_like(dyg)
        dg_, db_ = torch.empty(256, device=dev), torch.empty(256, device=dev)
        rec("groupnorm_relu_bwd", Mg, 5120 * Mg, "5120M: reduce pass reads x, dy; apply pass reads x, dy, writes dx",
            lambda: call("scan_groupnorm_relu_backward", P(xg), P(bet), P(dyg), shape.ref(), 256, 32, P(stats), P(gam), 1,
                         P(dxg), P(dg_), P(db_), 0, P(wsg), st()))
        del feat, dfe, dyg, dxg
        torch.cuda.empty_cache()
        # ---- CKA class branches' output conv (a15): x [Mc, 8 x 128] -> one channel per class, 8 images of one level
        Mc_target = min(M, 1 << 21)  # 8.6 GB of input at 2^21 rows: far beyond the last-level cache already
        hw = Mc_target // n_img
        h = 1
        while h * h * 2 < hw:
            h += 1
        wd_ = hw // h
        Mc = n_img * h * wd_
        shape = ops.PyramidShape(n_img, [(h, wd_)])
        xc = torch.relu(torch.randn((Mc, 1024), device=dev, generator=g))
        wc = torch.randn((8, 9, 1024), device=dev, generator=g) * 0.03
        bc = torch.randn(8, device=dev, generator=g)
        yc = torch.empty((Mc, 8), device=dev)
        wsc = torch.empty((query("scan_gconv3x3_to1_ws_floats", shape.ref(), 8, 128),), device=dev)
        rec("gconv3x3_to1_fwd", Mc, Mc * (4096 + 32 + 2 * 288), "M(4096+32+576): x read, y written, 72 tap products written and read",
            lambda: call("scan_gconv3x3_to1_forward", P(xc), shape.ref(), 8, 128, P(wc), P(bc), P(yc), 8, P(wsc), st()))
        dyc = torch.randn((Mc, 8), device=dev, generator=g)
        dxc = torch.empty_like(xc)
        dwc = torch.zeros_like(wc)
        rec("gconv3x3_to1_bwd", Mc, Mc * (8192 + 32), "M(8192+32): x and dy read, dx written (dw: 36 KB per workgroup)",
            lambda: call("scan_gconv3x3_to1_backward", P(xc), P(dyc), 8, shape.ref(), 8, 128, P(wc), 1, P(dxc), P(dwc), 0,
                         P(wsc), st()))
        del xc, dxc, yc, dyc, wsc
        torch.cuda.empty_cache()
        # ---- 2x2 max-pool (VGG stages 3-5) and the FPN top-down join, on 8 single-level images of 256 channels
        hp, wp = 2 * (h // 2), 2 * (wd_ // 2)
        Mp = n_img * hp * wp
        xp = torch.randn((Mp, 256), device=dev, generator=g)
        yp = torch.empty((Mp // 4, 256), device=dev)
        rec("maxpool2x2_fwd", Mp, Mp * 1280, "1280M: x read (1024 B / input pixel), y written (256 B / input pixel)",
            lambda: call("scan_maxpool2x2_forward", P(xp), n_img, hp, wp, 256, P(yp), st()))
        dyp = torch.randn((Mp // 4, 256), device=dev, generator=g)
        dxp = torch.empty_like(xp)
        rec("maxpool2x2_bwd", Mp, Mp * 2560, "2560M: x, y, dy read; dx written",
            lambda: call("scan_maxpool2x2_backward", P(xp), P(yp), P(dyp), n_img, hp, wp, 256, P(dxp), 1, st()))
        rec("upsample2x_add", Mp, Mp * 2304, "2304M: lateral read, coarse read (1/4), sum written",
            lambda: call("scan_upsample2x_add", P(xp), P(yp), n_img, hp // 2, wp // 2, 256, P(dxp), st()))
        rec("downsample2x_sum", Mp, Mp * 1280, "1280M: g read, 2x2 sums written",
            lambda: call("scan_downsample2x_sum", P(xp), n_img, hp // 2, wp // 2, 256, P(yp), st()))
        del xp, yp, dyp, dxp
        torch.cuda.empty_cache()
    # ---- bf16 hi / lo planes of all conv weights of a training iteration in one launch (csrc/batched.hip): the C2F
    # model's job mix (3x3 weights, both plane orientations for the trainable ones), M = fp32 weight elements split
    jobs, keep, off, elems = [], [], 0, 0
    for (O, Cs, modes) in [(64, 64, (0,)), (128, 64, (0,)), (128, 128, (0,))] + [(256, 128, (0, 1))] + \
            [(256, 256, (0, 1))] * 26 + [(512, 256, (0, 1))] + [(512, 512, (0, 1))] * 5 + [(1024, 264, (0, 1))] * 5:
        w = torch.randn((O, 9, Cs), device=dev, generator=g)
        for mode in modes:
            rows_, csw = (O, ops._round32(Cs)) if mode == 0 else (Cs, ops._round32(O))
            wh = torch.empty((rows_, 9, csw), dtype=torch.bfloat16, device=dev)
            wm, wl = torch.empty_like(wh), torch.empty_like(wh)
            jobs.append([w.data_ptr(), wh.data_ptr(), wm.data_ptr(), O, 9, Cs, mode, rows_, csw, off, wl.data_ptr()])
            off += query("scan_weight_split_job_blocks", O, 9, Cs, mode, csw)
            keep.append((w, wh, wm, wl))
            elems += O * 9 * Cs
    table = torch.tensor(jobs, dtype=torch.int64).to(dev)
    rec("weight_split_batched", elems, 10 * elems, "10 B per weight element and orientation: fp32 read, three bf16 pieces written",
        lambda: call("scan_weight_split_batched", P(table), len(jobs), int(table.shape[1]), off, st()), unit="element")
    del keep, table
    # ---- class-aware NMS on one image's candidate set (a18): latency-bound, reported as time (bytes = the 20 B / box read)
    for n_box in (1000, 4000, 8192):
        xy = torch.rand((n_box, 2), device=dev, generator=g) * 1000
        wh_ = torch.rand((n_box, 2), device=dev, generator=g) * 120 + 4
        boxes = torch.cat([xy, xy + wh_], 1).contiguous()
        scores = torch.rand((n_box,), device=dev, generator=g)
        labels = torch.randint(1, 9, (n_box,), device=dev, generator=g).float()
        wsn = torch.empty((query("scan_nms_ws_bytes", n_box) + 15) // 16 * 2, dtype=torch.float64, device=dev)
        keepn = torch.empty((n_box,), dtype=torch.int64, device=dev)
        cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
        rec("nms_by_label_n%d" % n_box, n_box, 24 * n_box, "24 B / box read (sort + 64x64 IoU bit-mask tiles + scan: latency-bound, see us)",
            lambda: call("scan_nms", P(boxes), P(scores), P(labels), n_box, 0.6, 1, P(keepn), P(cnt), P(wsn), st()), unit="box")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--small", action="store_true", help="only the configs[4] size")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="", help="comma-separated kernel names (for counter passes over a few kernels)")
    ap.add_argument("--big-only", action="store_true", help="only the M = 2^24 size")
    a = ap.parse_args()
    import torch
    dev = torch.device("cuda:0")
    sizes = (M_CFG5,) if a.small else ((M_BIG,) if a.big_only else (M_CFG5, M_BIG))
    res = measure(dev, sizes, a.reps, only=set(a.only.split(",")) if a.only else None)
    for r in res:
        print("%-26s M=%9d %8.1f us %8.1f GB/s  frac %.3f%s" % (r["kernel"], r["M"], r["us"], r["GBps"], r["frac"],
                                                               "  (fits LLC)" if r["fits_llc"] else ""))
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"peak_GBps": PEAK_HBM_GBS, "records": res}, f, indent=1)


if __name__ == "__main__":
    main()
