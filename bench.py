#!/usr/bin/env python
"""bench.py -- SCAN hot path on MI355X: train images/sec, VGG16 C2F, 1024x2048 synthetic frames.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full domain-adaptation iteration (reference fcos_core/engine/trainer.py:266-424:
source forward + losses, 5 CKA discriminators on source, target forward, 5 CKA discriminators on
target, three backward passes, SGD step of all 8 sub-models) on a per-GPU batch of 2 source + 2
target frames (BASELINE.json configs[1]).

N > 1: one process per GPU (torch.distributed, RCCL over xGMI).  Started by `torch.distributed.run`
(RANK / WORLD_SIZE in the environment) this process is one rank; started plainly with `--gpus N`
it is only a LAUNCHER: it checks that N devices are visible, starts `python -m torch.distributed.run
--nproc-per-node N bench.py ...` as a child BEFORE touching the GPU and exits with its code.
`--scaling weak` (default): `--batch` source (= target) frames per GPU.  `--scaling strong`: a fixed
GLOBAL batch (`--global-batch`, default 16 = BASELINE.json configs[2]) split over the ranks like the
reference's loaders do (data/build.py:181-188).  Collectives per step: three gradient all-reduces over
contiguous arena ranges on a side stream, the 9x257 paradigm all-reduce, one loss-scalar reduce.

Prints ONE JSON line on rank 0.  `value` = source/target image PAIRS per second over all GPUs
(frames/s = 2x, in config); `roofline` = the conv kernel with the largest share, timed live with HIP
events on its stream; `roofline_pointwise` = HBM GB/s of the pointwise kernels (tools/pointwise_roofline.py);
`cpu_baseline` = the torch-CPU restatement (oracle/scan_ref.py, a port of the reference's CPU path) on this
host's cores: 1 warm-up + 3 timed iterations at the best thread count of a short sweep.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks.  fp32 kernels: v_mfma_f32_32x32x2_f32 157.3 TFLOP/s.
# bf16x3 kernels issue THREE v_mfma_f32_32x32x16_bf16 per fp32-equivalent product (hi*hi + hi*lo + lo*hi), so
# their ceiling in algorithmic (fp32-equivalent) FLOPs is the 2.5 PFLOP/s dense bf16 peak / 3.
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_BF16X3_TFLOPS = 2500.0 / 3.0


def csrc_sha1():
    """sha1 over the kernel sources (scan_amd/csrc/*.hip, *.h, *.cpp): what a committed PMC pass is valid for."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "scan_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "scan_amd", "csrc", "*.h"))
                    + glob.glob(os.path.join(ROOT, "scan_amd", "csrc", "*.cpp"))):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()


def pmc_traffic(kernel_name):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command (profiles/rNN_pmc_traffic.json,
    written by tools/pmc_summarize.py; FETCH_SIZE doubled per the gfx950 correction), keyed by the kernel symbol.
    The file is stamped with the commit and the kernel-source hash it was taken at: a pass taken on OTHER kernel sources
    is not reported (traffic = None, the note says why), so the figure cannot go stale silently.
    Returns (bytes or None, provenance dict)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, {"traffic_source": None}
    path = files[-1]
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception:
        return None, {"traffic_source": os.path.relpath(path, ROOT), "traffic_note": "unreadable"}
    meta = d.get("_meta", {})
    prov = {"traffic_source": os.path.relpath(path, ROOT), "traffic_commit": meta.get("commit"),
            "traffic_csrc_sha1": meta.get("csrc_sha1")}
    if meta.get("csrc_sha1") != csrc_sha1():
        prov["traffic_note"] = "PMC pass is from other kernel sources than this build (csrc sha1 differs): not reported"
        return None, prov
    rec = d.get(kernel_name)
    prov["traffic_note"] = "HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command at that commit"
    return (rec["hbm_bytes"] if rec else None), prov


# timer record name (scan_amd/ops.py) -> kernel symbol as rocprofv3 lists it: forward and data-gradient launches of
# a conv are the SAME kernel (dgrad = forward on dY with flipped/transposed weights); the _bnNNN suffix is the
# output-channel tile of the instance the launch took (scan_conv3x3_bf16x3_instance)
def symbol_of(name):
    import re
    m = re.match(r"conv(3x3|1x1)_bf16x3_(fwd|dgrad)_bn(\d+)$", name)
    if m:
        bn = int(m.group(3))
        th, nt = (8, 256) if bn == 64 else (16, 512)
        if bn > 2000:  # the 8-wave LDS-DMA instance of the 256-channel tile
            return "conv_bf16x3_v2_kernel<%d,%d,512,3,1,true>" % (bn - 2000, th)
        if bn > 1000:
            bn, nt = bn - 1000, 1024
        return "conv_bf16x3_v2_kernel<%d,%d,%d,%d>" % (bn, th, nt, 3 if m.group(1) == "3x3" else 1)
    # the _g1 / _g2 suffix is the round-2 generation choice by input channel count (scan_conv_wgrad_bf16x3_generation);
    # since round 3 every bf16x3 weight-gradient launch runs conv_wgrad_bf16x3_v6_kernel (3x3) / _v4_kernel (1x1)
    return {"conv1x1_bf16x3_wgrad_g1": "conv_wgrad_bf16x3_v4_kernel<1,S>",
            "conv1x1_bf16x3_wgrad_g2": "conv_wgrad_bf16x3_v4_kernel<1,S>",
            "conv3x3_bf16x3_wgrad_g1": "conv_wgrad_bf16x3_v6_kernel<3>",
            "conv3x3_bf16x3_wgrad_g2": "conv_wgrad_bf16x3_v6_kernel<3>",
            "conv_smallcin_bf16x3": "conv_smallcin_kernel",
            "conv_igemm_fwd": "conv_igemm_kernel<0,4>", "conv_igemm_dgrad": "conv_igemm_kernel<1,4>",
            "conv_wgrad": "conv_wgrad_kernel"}.get(name, name)


def by_symbol(ksum):
    out = {}
    for name, r in ksum.items():
        g = out.setdefault(symbol_of(name), {"launches": 0, "total_ms": 0.0, "flops": 0.0})
        g["launches"] += r["launches"]
        g["total_ms"] += r["total_ms"]
        g["flops"] += r["flops"]
    for g in out.values():
        g["avg_ms"] = g["total_ms"] / g["launches"]
        g["tflops"] = g["flops"] / (g["total_ms"] * 1e-3) / 1e12 if g["total_ms"] > 0 else 0.0
    return out


def peak_for(kernel_name):
    return PEAK_BF16X3_TFLOPS if "bf16x3" in kernel_name else PEAK_FP32_MFMA_TFLOPS


def cpu_baseline(h, w, timed=3):
    """oracle port timed on the host cores: one (src, tgt) pair per iteration, full DA iteration + SGD.
    Thread count: best of a short sweep on quarter-size frames (torch-CPU conv backward stops scaling, then
    collapses, past a few dozen threads), then 1 warm-up + `timed` timed full-size iterations."""
    import platform
    import torch
    from oracle import scan_ref
    from scan_amd import synth
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    sds = synth.all_state_dicts(9)
    frozen = ("body.features.0.", "body.features.2.", "body.features.5.", "body.features.7.")

    def one(hh, ww, P, st, bufs):
        imgs_s, imgs_t = synth.synth_images(1, hh, ww, 1234), synth.synth_images(1, hh, ww, 2234)
        tg = synth.synth_targets(1, hh, ww, 8, 12, 4321)
        t0 = time.time()
        for pd in P.values():
            for v in pd.values():
                v.grad = None
        scan_ref.da_iteration(P, st, imgs_s, tg, imgs_t)
        scan_ref.sgd_step(P, bufs)
        return time.time() - t0

    def fresh():
        return ({k: scan_ref.params(v, frozen_prefixes=frozen) for k, v in sds.items()},
                scan_ref.PrototypeState(sds["middle_head"]["prototype"]), {})

    physical = avail // 2 if avail >= 4 else avail  # SMT siblings do not help the conv kernels
    sweep = {}
    for n in sorted({c for c in (8, 16, 32, 64, physical) if 1 <= c <= avail}):
        torch.set_num_threads(n)
        P, st, bufs = fresh()
        one(h // 4, w // 4, P, st, bufs)  # warm-up (thread pool, allocator)
        sweep[n] = one(h // 4, w // 4, P, st, bufs)
        if sweep[n] > 4 * min(sweep.values()):
            break  # past the collapse: larger counts only get worse
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    P, st, bufs = fresh()
    one(h, w, P, st, bufs)  # warm-up
    times = [one(h, w, P, st, bufs) for _ in range(timed)]
    cpu_model = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    return {"times_s": [round(t, 2) for t in times], "best_s": min(times), "cores": cores, "visible_threads": avail,
            "cpu_model": cpu_model, "sweep_quarter_size_s": {str(k): round(v, 2) for k, v in sweep.items()}}


def _free_port():
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def visible_gpus():
    """GPUs this process tree may use, counted WITHOUT opening the GPU: kfd topology nodes with SIMDs (CPU nodes have
    simd_count 0), cut down by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES.  None if sysfs has
    no kfd topology (then the caller falls back to torch.cuda.device_count())."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(l.split(None, 1) for l in f.read().splitlines() if " " in l)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except Exception:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(a):
    """`python bench.py --gpus N` without a torch.distributed.run environment: become the launcher.  Nothing here
    opens the GPU (devices are counted from sysfs), the ranks are fresh child processes."""
    n_vis = visible_gpus()
    if n_vis is None:
        import torch
        n_vis = torch.cuda.device_count()
    if n_vis < a.gpus and not a.launch_check:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible -- refusing to measure fewer ranks than asked for"
                         % (a.gpus, n_vis))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=2, help="source (= target) frames per GPU per step (weak scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--global-batch", type=int, default=16,
                    help="--scaling strong: global source (= target) frames per step, split over the ranks "
                         "(SOLVER.IMS_PER_BATCH semantics, reference data/build.py:181-188)")
    ap.add_argument("--no-pointwise", action="store_true", help="skip the pointwise HBM roofline leg")
    ap.add_argument("--launch-check", action="store_true",
                    help="test hook for the launcher (tests/test_launcher.py, runs without GPUs): the ranks only "
                         "rendezvous over gloo, all-reduce their rank and rank 0 prints the world size it saw")
    ap.add_argument("--model", choices=("c2f", "s2c", "k2c", "k2c_r50"), default="c2f",
                    help="which shipped yaml's model (engine.CONFIGS); the headline metric is c2f")
    ap.add_argument("--forward-target", action="store_true",
                    help="target pass with DBSCAN node sampling + GST losses (reference: once val AP50 > INITIAL_AP50); "
                         "the headline metric uses False like the reference's first phase")
    ap.add_argument("--ft-positives", type=float, default=None,
                    help="with --forward-target: keep this fraction of the (pixel, class) act-map entries that pass the 0.05 "
                         "threshold as clustering candidates (SURVEY.md 8d second series: 0.01).  A random-init model passes "
                         "ALL entries -- the worst case -- a trained one a small fraction; measurement switch only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-companions", action="store_true",
                    help="skip the companion legs of the line (strict fp32-MFMA steps, the three-phase schedule, inference)")
    ap.add_argument("--strict-steps", type=int, default=10, help="timed steps of the strict fp32-MFMA companion leg")
    ap.add_argument("--serial-streams", action="store_true",
                    help="run everything on one stream (no side-stream overlap): what the per-kernel roofline "
                         "figures and the rocprof summaries under profiles/ are taken with")
    a = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ  # started by torch.distributed.run
    if not launched and a.gpus > 1:
        raise SystemExit(launch_ranks(a))
    if a.launch_check:
        import torch
        import torch.distributed as dist
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world != a.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.tensor([float(dist.get_rank())])
            dist.all_reduce(t)
            assert float(t) == world * (world - 1) / 2
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world,
                              "ranks_in_process_group": dist.get_world_size() if world > 1 else 1}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from scan_amd import comm, engine, ops, synth
    if a.ft_positives is not None:
        if not a.forward_target:
            raise SystemExit("--ft-positives needs --forward-target")
        from scan_amd.modeling import condgraph
        condgraph.FT_CANDIDATE_FRACTION = float(a.ft_positives)
    mcfg = engine.CONFIGS[a.model]
    body = mcfg["conv_body"]
    model = engine.build_model(device=dev, settings=mcfg)
    engine.load_procedural_weights(model, mcfg["num_classes"], body)
    # under torch.distributed.run the data-parallel path (flat-buffer all-reduce on the side stream, paradigm
    # all-reduce) is exercised even with a single rank
    trainer = engine.Trainer(model, settings=mcfg, distributed=True if dist.is_initialized() else None)

    def set_serial(flag):
        trainer.overlap_target = not flag
        if flag:
            trainer._saved_dis = trainer.dis_streams
            trainer.dis_streams = {}
            model["middle_head"].out_stream = None
            ops.WGRAD_STREAM = None
        elif hasattr(trainer, "_saved_dis"):
            trainer.dis_streams = trainer._saved_dis
            model["middle_head"].out_stream = trainer.out_stream
            ops.WGRAD_STREAM = trainer.wgrad_stream

    if a.serial_streams:
        set_serial(True)
    H, W = a.height, a.width
    B = a.batch if a.scaling == "weak" else comm.images_per_gpu(a.global_batch, world)
    # frames go through the collator's zero padding to /32 (structures.to_image_list), e.g. 1333x2666 -> 1344x2688
    imgs_s = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * B, 1234 + 100 * rank)], 32)
    imgs_t = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * B, 2234 + 100 * rank)], 32)
    # ground truth as the collator delivers it: host tensors (the plan kernels upload it on their own stream)
    tg = synth.synth_targets(B, H, W, mcfg["num_classes"] - 1, 12, 4321 + 100 * rank)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        losses = trainer.step(imgs_s, tg, imgs_t, forward_target=a.forward_target)
        if world > 1:  # the reference reduces the loss scalars to rank 0 for its meters (engine/trainer.py:76-98)
            comm.reduce_loss_dict(losses)
        return losses

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.time()
    for _ in range(a.steps):
        losses = step()
    barrier()
    dt = time.time() - t0
    # per-kernel roofline figures: HIP events around every conv launch on its stream.  With the side-stream overlap
    # of the timed region an event pair also spans whatever co-runs on the other streams, so the kernels are timed
    # in two extra steps with the overlap switched off (same kernels, same shapes; rocprof: profiles/*serial*).
    set_serial(True)
    # one untimed step in this schedule first: its allocation pattern differs from the overlapped one, and a first pass
    # through the caching allocator (hipMalloc) would otherwise land in the serial figure quoted below
    trainer.step(imgs_s, tg, imgs_t, forward_target=a.forward_target)
    torch.cuda.synchronize()
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

    # ---- companion legs (N = 1, headline model only): same frames, same trainer, outside the timed region
    strict = three_phase = infer = None
    if world == 1 and not a.no_companions and not a.forward_target:
        # (a) the SAME step with every convolution on the exact fp32 matrix-core kernels (v_mfma_f32_32x32x2_f32 /
        # 16x16x4_f32 = an fp32 fma chain, the reference's arithmetic): what the headline costs without the bf16 split
        set_serial(a.serial_streams)
        ops.CONV_MODE = "fp32"
        try:
            for _ in range(2):
                trainer.step(imgs_s, tg, imgs_t)
            torch.cuda.synchronize()
            t0s = time.time()
            for _ in range(a.strict_steps):
                ls = trainer.step(imgs_s, tg, imgs_t)
            torch.cuda.synchronize()
            dts = (time.time() - t0s) / a.strict_steps
            set_serial(True)
            trainer.step(imgs_s, tg, imgs_t)
            torch.cuda.synchronize()
            ops.kernel_timer.enabled = True
            ops.kernel_timer.reset()
            trainer.step(imgs_s, tg, imgs_t)
            torch.cuda.synchronize()
            ops.kernel_timer.enabled = False
            ks = by_symbol(ops.kernel_timer.summary())
            sdom = max(ks.items(), key=lambda kv: kv[1]["total_ms"])
            strict = {"dtype": "f32 (exact fp32 MFMA, v_mfma_f32_32x32x2_f32 / 16x16x4_f32)",
                      "ms_per_step": round(dts * 1e3, 2), "pairs_per_s": round(B / dts, 4), "steps": a.strict_steps,
                      "dominant_kernel": sdom[0], "dominant_tflops": round(sdom[1]["tflops"], 2),
                      "frac_of_157.3": round(sdom[1]["tflops"] / PEAK_FP32_MFMA_TFLOPS, 4),
                      "dominant_avg_launch_ms": round(sdom[1]["avg_ms"], 4), "dominant_launches": sdom[1]["launches"],
                      "losses_finite": all(bool(torch.isfinite(v)) for v in ls.values())}
        finally:
            ops.CONV_MODE = "bf16x3"
            set_serial(a.serial_streams)
        # (b) the reference's three-phase schedule (source forward/backward, target forward/backward as separate
        # pyramids): what do_train runs when source and target batches pad to different sizes
        trainer.paired = False
        try:
            for _ in range(2):
                trainer.step(imgs_s, tg, imgs_t)
            torch.cuda.synchronize()
            t0p = time.time()
            for _ in range(a.steps):
                trainer.step(imgs_s, tg, imgs_t)
            torch.cuda.synchronize()
            dtp = (time.time() - t0p) / a.steps
            three_phase = {"ms_per_step": round(dtp * 1e3, 2), "pairs_per_s": round(B / dtp, 4), "steps": a.steps,
                           "note": "Trainer.step with paired=False: the schedule real ragged batches take"}
        finally:
            trainer.paired = True
        # (c) inference on the same frames (engine.inference: backbone + middle head + FCOS head + post-processing +
        # batched NMS, TEST.MODE of the yaml), and the NMS launch alone on the last image's candidate set
        import scan_amd.modeling.fcos as fcos_mod
        frames = imgs_t
        for _ in range(2):
            dets = engine.inference(model, frames)
        torch.cuda.synchronize()
        n_inf = max(3, a.steps)
        t0i = time.time()
        for _ in range(n_inf):
            dets = engine.inference(model, frames, static_weights=True)  # a dataset loop: nothing trains between batches
        torch.cuda.synchronize()
        dti = (time.time() - t0i) / n_inf
        nms_rec = fcos_mod.last_nms_record()
        infer = {"images_per_s": round(B / dti, 3), "ms_per_batch": round(dti * 1e3, 2), "batch": B,
                 "test_mode": mcfg["test_mode"], "detections_per_image": [int(len(d[0])) for d in dets]}
        if nms_rec is not None:
            boxes_n, scores_n, labels_n, thr = nms_rec
            n_c = int(boxes_n.shape[0])
            if n_c > 0:
                for _ in range(3):
                    ops.nms_by_label(boxes_n, scores_n, labels_n, thr)
                torch.cuda.synchronize()
                t0n = time.time()
                for _ in range(20):
                    ops.nms_by_label(boxes_n, scores_n, labels_n, thr)
                torch.cuda.synchronize()
                infer["nms_us_per_image"] = round((time.time() - t0n) / 20 * 1e6, 1)
            infer["n_candidates"] = n_c
        for m_ in model.values():
            m_.train()

    if rank == 0:
        pairs = B * world * a.steps
        value = pairs / dt
        dom = max(by_symbol(ksum).items(), key=lambda kv: kv[1]["total_ms"]) if ksum else None
        roof = None
        if dom:
            name, r = dom
            traffic, prov = pmc_traffic(name)
            roof = {"bound": "mfma", "kernel": name, "achieved": round(r["tflops"], 2), "peak": round(peak_for(name), 1),
                    "unit": "TFLOP/s", "frac": round(r["tflops"] / peak_for(name), 4), "traffic": traffic,
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
            roof.update(prov)
        pointwise = None
        if not a.no_pointwise and world == 1:
            from tools import pointwise_roofline
            del trainer, model, imgs_s, imgs_t
            torch.cuda.empty_cache()
            recs = pointwise_roofline.measure(dev, reps=5, K=9)
            pointwise = {"bound": "hbm", "peak": pointwise_roofline.PEAK_HBM_GBS, "unit": "GB/s",
                         "sizes": {"cfg5": pointwise_roofline.M_CFG5, "microbench": pointwise_roofline.M_BIG},
                         "note": "algorithmic bytes / HIP-event time per launch; cfg5-size working sets under 256 MB "
                                 "can be served by the last-level cache (fits_llc)",
                         "kernels": [{k: r[k] for k in ("kernel", "M", "bytes", "us", "GBps", "frac", "fits_llc")}
                                     for r in recs]}
        cpu = None
        if not a.no_cpu_baseline and a.model == "c2f" and world == 1:  # rank 0 at N=1 only
            cb = cpu_baseline(H, W)
            cpu = {"value": round(1.0 / cb["best_s"], 5), "unit": "pairs/s", "cores": cb["cores"], "kind": "port",
                   "cpu_model": cb["cpu_model"], "visible_threads": cb["visible_threads"],
                   "iterations_s": cb["times_s"], "thread_sweep_quarter_size_s": cb["sweep_quarter_size_s"],
                   "sample": "1 (src,tgt) pair of %dx%d frames per iteration, full DA iteration + SGD (the GPU step does "
                             "%d pairs); 1 warm-up + %d timed iterations, best %.1f s"
                             % (H, W, B, len(cb["times_s"]), cb["best_s"])}
        line = {
            "metric": "train images/sec (whole node), %s %s %dx%d" % (
                "VGG16" if body.startswith("VGG") else body.split("-FPN")[0], a.model.split("_")[0].upper(), H, W),
            "value": round(value, 4),
            "unit": "image pairs/s (1 source + 1 target frame per pair)", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True,
            "scaling": a.scaling, "vs_baseline": None,
            "dtype": "bf16x3 (fp32 storage + fp32 accumulate; every conv operand split hi+lo into 2 bf16, 3 bf16 MFMAs per "
                     "product = 16 significand bits per operand; strict fp32-MFMA figure: strict_fp32)",
            "data": "synthetic",
            "config": {"workload": "SCAN %s %s DA iteration, %d src + %d tgt frames/GPU at %dx%d, "
                                   "forward_target=%s%s, procedural weights" % (
                           a.model.upper(), body, B, B, H, W, a.forward_target,
                           "" if a.ft_positives is None else " (%.3g of the act-map entries kept as clustering candidates)" % a.ft_positives),
                       "arithmetic": "fp32 storage and accumulation; 3x3 convs split each fp32 operand hi+lo into "
                                     "2 x bf16 and issue 3 bf16 MFMAs per product (1.7e-6 rel on the losses vs fp32)",
                       "global_batch_pairs": B * world, "frames_per_s": round(2 * value, 4), "parallelism": "dp%d" % world,
                       "ranks_in_process_group": dist.get_world_size() if dist.is_initialized() else 1,
                       "collective_backend": dist.get_backend() if dist.is_initialized() else None,
                       "losses_finite": finite},
            "roofline": roof, "roofline_pointwise": pointwise, "cpu_baseline": cpu,
            "strict_fp32": strict, "three_phase_schedule": three_phase, "inference": infer,
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
