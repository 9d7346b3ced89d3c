#!/usr/bin/env python
"""bench.py -- SCAN hot path on MI355X: train images/sec, VGG16 C2F, 1024x2048 synthetic frames.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full domain-adaptation iteration (reference fcos_core/engine/trainer.py:266-424:
source forward + losses, 5 CKA discriminators on source, target forward, 5 CKA discriminators on
target, three backward passes, SGD step of all 8 sub-models) on a per-GPU batch of 2 source + 2
target frames (BASELINE.json configs[1]).  N > 1: one process per GPU (torch.distributed, RCCL),
same per-GPU batch (weak scaling), one gradient all-reduce per sub-model flat buffer.

Prints ONE JSON line on rank 0.  `value` = source/target image PAIRS per second over all GPUs
(frames/s = 2x, in config); `roofline` = fp32-MFMA implicit-GEMM conv kernels timed live with HIP
events on their stream during the timed steps; `cpu_baseline` = the torch-CPU restatement
(oracle/scan_ref.py, a port of the reference's CPU path) on this host's cores, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks.  fp32 kernels: v_mfma_f32_32x32x2_f32 157.3 TFLOP/s.
# bf16x3 kernels issue THREE v_mfma_f32_32x32x16_bf16 per fp32-equivalent product (hi*hi + hi*lo + lo*hi), so
# their ceiling in algorithmic (fp32-equivalent) FLOPs is the 2.5 PFLOP/s dense bf16 peak / 3.
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_BF16X3_TFLOPS = 2500.0 / 3.0


def pmc_traffic(kernel_name):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
    (profiles/r01_pmc_traffic.json; FETCH_SIZE doubled per the gfx950 correction).  None if not recorded."""
    key = {"conv3x3_wgrad_bf16x3_kernel<3,1>": "conv3x3_bf16x3_wgrad",
           "conv3x3_bf16x3_kernel<128,16,512,3>": "conv3x3_bf16x3_fwd_dgrad_bn128",
           "conv3x3_bf16x3_kernel<64,8,256,3>": "conv3x3_bf16x3_fwd_bn64", "conv_igemm_kernel<0,4>": "conv_igemm_fwd",
           "conv_igemm_kernel<1,4>": "conv_igemm_dgrad", "conv_wgrad_kernel": "conv_wgrad_fp32"}.get(kernel_name)
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            rec = json.load(f).get(key)
        return rec["hbm_bytes"] if rec else None
    except Exception:
        return None


# timer record name (scan_amd/ops.py) -> kernel symbol as rocprofv3 lists it: forward and data-gradient launches of
# a conv are the SAME kernel (dgrad = forward on dY with flipped/transposed weights)
SYMBOL = {"conv3x3_bf16x3_fwd_bn128": "conv3x3_bf16x3_kernel<128,16,512,3>",
          "conv3x3_bf16x3_dgrad_bn128": "conv3x3_bf16x3_kernel<128,16,512,3>",
          "conv3x3_bf16x3_fwd_bn64": "conv3x3_bf16x3_kernel<64,8,256,3>",
          "conv3x3_bf16x3_dgrad_bn64": "conv3x3_bf16x3_kernel<64,8,256,3>",
          "conv1x1_bf16x3_fwd_bn128": "conv3x3_bf16x3_kernel<128,16,512,1>",
          "conv1x1_bf16x3_dgrad_bn128": "conv3x3_bf16x3_kernel<128,16,512,1>",
          "conv1x1_bf16x3_fwd_bn64": "conv3x3_bf16x3_kernel<64,8,256,1>",
          "conv1x1_bf16x3_dgrad_bn64": "conv3x3_bf16x3_kernel<64,8,256,1>",
          "conv1x1_bf16x3_wgrad": "conv3x3_wgrad_bf16x3_kernel<1,S>",
          "conv3x3_bf16x3_wgrad": "conv3x3_wgrad_bf16x3_kernel<3,1>",
          "conv_smallcin_bf16x3": "conv_smallcin_kernel",
          "conv_igemm_fwd": "conv_igemm_kernel<0,4>", "conv_igemm_dgrad": "conv_igemm_kernel<1,4>",
          "conv_wgrad": "conv_wgrad_kernel"}


def by_symbol(ksum):
    out = {}
    for name, r in ksum.items():
        g = out.setdefault(SYMBOL.get(name, name), {"launches": 0, "total_ms": 0.0, "flops": 0.0})
        g["launches"] += r["launches"]
        g["total_ms"] += r["total_ms"]
        g["flops"] += r["flops"]
    for g in out.values():
        g["avg_ms"] = g["total_ms"] / g["launches"]
        g["tflops"] = g["flops"] / (g["total_ms"] * 1e-3) / 1e12 if g["total_ms"] > 0 else 0.0
    return out


def peak_for(kernel_name):
    return PEAK_BF16X3_TFLOPS if "bf16x3" in kernel_name else PEAK_FP32_MFMA_TFLOPS


def cpu_baseline(h, w):
    """oracle port timed on the host cores: 1 (src,tgt) pair, one DA iteration."""
    import torch
    from oracle import scan_ref
    from scan_amd import synth
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # torch-CPU conv backward degrades badly past a few dozen threads (256 threads: 80x slower than 8 here);
    # use what it scales to and report that number as `cores`
    cores = min(cores, 16)
    torch.set_num_threads(cores)
    sds = synth.all_state_dicts(9)
    P = {k: scan_ref.params(v, frozen_prefixes=("body.features.0.", "body.features.2.", "body.features.5.",
                                                "body.features.7.")) for k, v in sds.items()}
    st = scan_ref.PrototypeState(sds["middle_head"]["prototype"])
    imgs_s, imgs_t = synth.synth_images(1, h, w, 1234), synth.synth_images(1, h, w, 2234)
    tg = synth.synth_targets(1, h, w, 8, 12, 4321)
    bufs = {}
    t0 = time.time()
    scan_ref.da_iteration(P, st, imgs_s, tg, imgs_t)
    scan_ref.sgd_step(P, bufs)
    dt = time.time() - t0
    return dt, cores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=2, help="source (= target) frames per GPU per step")
    ap.add_argument("--model", choices=("c2f", "s2c", "k2c", "k2c_r50"), default="c2f",
                    help="which shipped yaml's model (engine.CONFIGS); the headline metric is c2f")
    ap.add_argument("--forward-target", action="store_true",
                    help="target pass with DBSCAN node sampling + GST losses (reference: once val AP50 > INITIAL_AP50); "
                         "the headline metric uses False like the reference's first phase")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial-streams", action="store_true",
                    help="run everything on one stream (no side-stream overlap): what the per-kernel roofline "
                         "figures and the rocprof summaries under profiles/ are taken with")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ  # started by torch.distributed.run
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from scan_amd import engine, ops, synth
    mcfg = engine.CONFIGS[a.model]
    body = mcfg.get("conv_body", "VGG-16-FPN-RETINANET")
    model = engine.build_model(mcfg["num_classes"], mcfg["test_mode"], device=dev, transfer_cfg=mcfg["transfer_cfg"],
                               conv_body=body)
    engine.load_procedural_weights(model, mcfg["num_classes"], body)
    # under torch.distributed.run the data-parallel path (flat-buffer all-reduce on the side stream, paradigm
    # all-reduce) is exercised even with a single rank
    trainer = engine.Trainer(model, distributed=True if dist.is_initialized() else None)

    def set_serial(flag):
        trainer.overlap_target = not flag
        if flag:
            trainer._saved_dis = trainer.dis_streams
            trainer.dis_streams = {}
        elif hasattr(trainer, "_saved_dis"):
            trainer.dis_streams = trainer._saved_dis

    if a.serial_streams:
        set_serial(True)
    H, W, B = a.height, a.width, a.batch
    # frames go through the collator's zero padding to /32 (structures.to_image_list), e.g. 1333x2666 -> 1344x2688
    imgs_s = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * B, 1234 + 100 * rank)], 32)
    imgs_t = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * B, 2234 + 100 * rank)], 32)
    tg = [(b.to(dev), l.to(dev))
          for b, l in synth.synth_targets(B, H, W, mcfg["num_classes"] - 1, 12, 4321 + 100 * rank)]

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        trainer.step(imgs_s, tg, imgs_t, forward_target=a.forward_target)
    barrier()
    t0 = time.time()
    for _ in range(a.steps):
        losses = trainer.step(imgs_s, tg, imgs_t, forward_target=a.forward_target)
    barrier()
    dt = time.time() - t0
    # per-kernel roofline figures: HIP events around every conv launch on its stream.  With the side-stream overlap
    # of the timed region an event pair also spans whatever co-runs on the other streams, so the kernels are timed
    # in two extra steps with the overlap switched off (same kernels, same shapes; rocprof: profiles/*serial*).
    set_serial(True)
    ops.kernel_timer.enabled = True
    ops.kernel_timer.reset()
    roof_steps = 2
    t0r = time.time()
    for _ in range(roof_steps):
        trainer.step(imgs_s, tg, imgs_t, forward_target=a.forward_target)
    torch.cuda.synchronize()
    dtr = time.time() - t0r
    ops.kernel_timer.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ksum = ops.kernel_timer.summary()
    finite = all(bool(torch.isfinite(v)) for v in losses.values())

    if rank == 0:
        pairs = B * world * a.steps
        value = pairs / dt
        dom = max(by_symbol(ksum).items(), key=lambda kv: kv[1]["total_ms"]) if ksum else None
        roof = None
        if dom:
            name, r = dom
            roof = {"bound": "mfma", "kernel": name, "achieved": round(r["tflops"], 2), "peak": round(peak_for(name), 1),
                    "unit": "TFLOP/s", "frac": round(r["tflops"] / peak_for(name), 4), "traffic": pmc_traffic(name),
                    "traffic_note": "HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes (profiles/r01_pmc_traffic.json)",
                    "peak_note": "algorithmic fp32-equivalent FLOPs; bf16x3 kernels spend 3 bf16 MFMAs per product, "
                                 "peak = 2.5 PFLOP/s dense bf16 / 3",
                    "launches": r["launches"], "avg_launch_ms": round(r["avg_ms"], 4),
                    "measured": "HIP events on the launch stream, %d steps with side-stream overlap off "
                                "(%.1f ms/step serial vs %.1f ms/step overlapped)" % (roof_steps, dtr / roof_steps * 1e3,
                                                                                     dt / a.steps * 1e3),
                    "all_conv_kernels": {k: {"tflops": round(v["tflops"], 2), "avg_ms": round(v["avg_ms"], 4),
                                             "launches": v["launches"],
                                             "share_of_serial_step": round(v["total_ms"] / (dtr * 1e3), 3)}
                                         for k, v in ksum.items()}}
        cpu = None
        if not a.no_cpu_baseline and a.model == "c2f" and world == 1:  # rank 0 at N=1 only
            sh, sw = H, W  # one full-size pair: ~10 s on 16 host threads
            cdt, cores = cpu_baseline(sh, sw)
            scale = (sh * sw) / float(H * W)
            cpu = {"value": round(scale / cdt, 5), "unit": "pairs/s", "cores": cores, "kind": "port",
                   "sample": "1 (src,tgt) pair of %dx%d frames, one full DA iteration + SGD (the GPU step does 2 pairs); "
                             "%.1f s of CPU work" % (sh, sw, cdt)}
        line = {
            "metric": "train images/sec (whole node), VGG16 C2F 1024x2048", "value": round(value, 4),
            "unit": "image pairs/s (1 source + 1 target frame per pair)", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SCAN %s %s DA iteration, %d src + %d tgt frames/GPU at %dx%d, "
                                   "forward_target=%s, procedural weights" % (a.model.upper(), body, B, B, H, W, a.forward_target),
                       "arithmetic": "fp32 storage and accumulation; 3x3 convs split each fp32 operand hi+lo into "
                                     "2 x bf16 and issue 3 bf16 MFMAs per product (1.7e-6 rel on the losses vs fp32)",
                       "global_batch_pairs": B * world, "frames_per_s": round(2 * value, 4), "parallelism": "dp%d" % world,
                       "losses_finite": finite},
            "roofline": roof, "cpu_baseline": cpu,
        }
        # RCCL prints a version banner through C stdio, which is still buffered here when stdout is a pipe: push it
        # out first so the JSON line is the LAST line on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
