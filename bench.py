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
# The split-operand kernels (scan_amd/csrc/conv_split.h) issue SIX ("bf16x6": three bf16 pieces per operand, all 24
# significand bits -- the reference's fp32 arithmetic) or THREE ("bf16x3": two pieces, 16 bits) bf16 MFMAs per fp32-equivalent
# product, so their ceilings in algorithmic (fp32-equivalent) FLOPs are the 2.5 PFLOP/s dense bf16 peak / 6 and / 3.
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_BF16X6_TFLOPS = PEAK_BF16_TFLOPS / 6.0
PEAK_BF16X3_TFLOPS = PEAK_BF16_TFLOPS / 3.0
HEADLINE_MODE = "bf16x6"
XGMI_LINK_GBS = 153.0  # per link and direction; 7 links per GPU (MI355X_MICROARCH.md)
RCCL_MAX_NCHANNELS = 32  # launcher default for N > 1 (profiles/r06_dp_emulation.txt: 16-64 resident workgroups cost the step the same, 8 more), see launch_ranks


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
# output-channel tile of the instance the launch took (scan_conv3x3_bf16x6_instance / _bf16x3_instance)
def symbol_of(name):
    import re
    m = re.match(r"conv(3x3|1x1)_bf16x(6|3)_(fwd|dgrad)_bn(\d+)$", name)
    if m:
        np_, bn, ks = (3 if m.group(2) == "6" else 2), int(m.group(4)), (3 if m.group(1) == "3x3" else 1)
        th, nt, tail = (8, 256, "") if bn == 64 else (16, 512, "")
        if np_ == 3 and ks == 1:  # scan_conv1x1_bf16x6_instance: 1128 / 1256 = LDS-DMA weight tiles (scan_tune conv1x1)
            return "conv_split_kernel<3,%d,%d,%d,1%s>" % (bn % 1000, 8 if bn == 64 else 16, 256 if bn == 64 else 512, ",1,true" if bn > 1000 else "")
        if np_ == 3:  # three pieces: weight tiles by LDS-DMA on every 3x3 instance (scan_conv3x3_bf16x6_instance)
            if bn in (1064, 2064):
                return "conv_split_kernel<3,64,%d,512,3,1,true>" % (16 if bn == 1064 else 32)
            return "conv_split_kernel<3,%d,%d,%d,%d%s>" % (bn, th, nt, ks, ",1,true" if ks == 3 else "")
        if bn > 2000:  # the 8-wave LDS-DMA instance of the 256-channel tile
            return "conv_split_kernel<2,%d,%d,512,3,1,true>" % (bn - 2000, th)
        if bn > 1000:
            bn, nt = bn - 1000, 1024
        return "conv_split_kernel<2,%d,%d,%d,%d>" % (bn, th, nt, ks)
    return {"conv1x1_bf16x3_wgrad": "conv_wgrad_v4_kernel<2,1,S>", "conv1x1_bf16x6_wgrad": "conv_wgrad_v4_kernel<3,1,S>",
            "conv3x3_bf16x3_wgrad": "conv_wgrad_v6_kernel<2,64,3>", "conv3x3_bf16x6_wgrad": "conv_wgrad_v6_kernel<3,32,3>",
            "conv_smallcin_bf16x3": "conv_smallcin_kernel<2>", "conv_smallcin_bf16x6": "conv_smallcin_kernel<3>",
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
    """dense-MFMA ceiling of a conv kernel symbol in fp32-equivalent TFLOP/s"""
    import re
    m = re.match(r"conv_(split|wgrad_v[46]|smallcin)_kernel<(\d)", kernel_name)
    if m:
        return PEAK_BF16X6_TFLOPS if m.group(2) == "3" else PEAK_BF16X3_TFLOPS
    return PEAK_FP32_MFMA_TFLOPS


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
        if n != physical and sweep and sweep[max(sweep)] > 4 * min(sweep.values()):
            continue  # past the collapse: larger counts only get worse (the physical-core count is always measured)
        torch.set_num_threads(n)
        P, st, bufs = fresh()
        one(h // 4, w // 4, P, st, bufs)  # warm-up (thread pool, allocator)
        sweep[n] = one(h // 4, w // 4, P, st, bufs)
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    P, st, bufs = fresh()
    one(h, w, P, st, bufs)  # warm-up
    times = [one(h, w, P, st, bufs) for _ in range(timed)]
    # SURVEY.md 8(d) asks for the host's physical cores: ONE full-size iteration at that thread count beside the best-of-sweep
    # figure (the pool is warm from the sweep).  Skipped -- and said so -- when the quarter-size sweep puts it beyond ~2 minutes.
    phys = {"cores": physical, "quarter_size_s": round(sweep[physical], 2)}
    if physical == cores:
        phys.update(iteration_s=round(min(times), 2), value=round(1.0 / min(times), 5), note="the best-of-sweep count")
    elif sweep[physical] / sweep[cores] * min(times) <= 120.0:
        torch.set_num_threads(physical)
        P, st, bufs = fresh()
        t_ph = one(h, w, P, st, bufs)
        phys.update(iteration_s=round(t_ph, 2), value=round(1.0 / t_ph, 5), note="one full-size iteration, no warm-up beyond the sweep")
    else:
        est = sweep[physical] / sweep[cores] * min(times)
        phys.update(iteration_s=None, value=round(1.0 / est, 5),
                    note="not run at full size: the quarter-size ratio to the best count puts it at ~%.0f s per iteration" % est)
    cpu_model = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    return {"times_s": [round(t, 2) for t in times], "best_s": min(times), "cores": cores, "visible_threads": avail,
            "cpu_model": cpu_model, "sweep_quarter_size_s": {str(k): round(v, 2) for k, v in sweep.items()},
            "physical_cores": phys}


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
    # a container may expose only some GPUs through device cgroups while sysfs still lists every GPU of the host: the
    # render nodes this process can open bound the count too (one /dev/dri/renderD* per GPU)
    try:
        nodes = [f for f in os.listdir("/dev/dri") if f.startswith("renderD")]
        usable = sum(1 for f in nodes if os.access(os.path.join("/dev/dri", f), os.R_OK | os.W_OK))
        if nodes:
            n = min(n, usable)
    except OSError:
        pass
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
    # RCCL's CU footprint beside the one-workgroup-per-CU MFMA kernels: one workgroup per channel stays resident for the whole
    # collective.  Default from the emulation (tools/dp_emulate.py, profiles/r06_dp_emulation.txt); an explicit setting wins.
    env.setdefault("NCCL_MAX_NCHANNELS", str(RCCL_MAX_NCHANNELS))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
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
    ap.add_argument("--conv-mode", choices=("bf16x6", "bf16x3", "fp32"), default=HEADLINE_MODE,
                    help="arithmetic of the timed region (default: the headline, bf16x6).  Another value is for profiling the "
                         "companion arithmetics under rocprofv3 (use with --no-companions); the line then says so in dtype")
    ap.add_argument("--three-phase", action="store_true",
                    help="time the three-phase schedule (Trainer.paired = False: what ragged batches take) as the main region; "
                         "a profiling / A-B aid, use with --no-companions")
    ap.add_argument("--no-companions", action="store_true",
                    help="skip the companion legs of the line (strict fp32-MFMA steps, the three-phase schedule, inference)")
    ap.add_argument("--strict-steps", type=int, default=5, help="timed steps of the strict fp32-MFMA companion leg")
    ap.add_argument("--allow-exp-lib", action="store_true",
                    help="run although SCAN_HIP_LIB selects another build than scan_amd/libscan_hip.so (timing experiments, "
                         "csrc/Makefile exp_*: WRONG results by construction); the line is then marked as not a measurement "
                         "of the product")
    ap.add_argument("--surface", choices=("layers", "none"), default="layers",
                    help="companion leg `extra.drop_in`: the same DA iteration through scan_amd.surface -- NCHW tensors, one module "
                         "call per level, scan_amd.layers on the C++ autograd operators, i.e. the call shape of the reference's module "
                         "files (rpn/fcos/fcos.py:66-114, condgraph.py:86-119) -- ms/step, launches/step and a per-operator table "
                         "beside the engine's (tools/surface_bench.py)")
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
    # each rank on the CPUs local to ITS GPU, disjoint from its neighbours' (scan_amd/comm.py: sysfs kfd topology -> NUMA node ->
    # cpulist; SCAN_RANK_BINDING=0 switches it off).  A single rank keeps the whole affinity mask: the CPU baseline needs it.
    from scan_amd import comm as _comm
    binding = _comm.bind_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world))) if world > 1 else None
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:  # started by the driver's own torch.distributed.run: the launcher's default applies here too
            os.environ.setdefault("NCCL_MAX_NCHANNELS", str(RCCL_MAX_NCHANNELS))
        dist.init_process_group("nccl", device_id=dev)

    from scan_amd import _lib, comm, engine, ops, synth
    ident = _lib.lib_identity()
    if not ident["product_library"] and not a.allow_exp_lib:
        raise SystemExit("bench.py: SCAN_HIP_LIB selects %s, not the product library scan_amd/libscan_hip.so -- no headline is "
                         "printed from an experiment build (pass --allow-exp-lib for an A/B timing run)" % _lib.LIB_PATH)
    ops.CONV_MODE = a.conv_mode
    if a.conv_mode != HEADLINE_MODE and not a.no_companions:
        raise SystemExit("--conv-mode %s is a profiling aid: combine it with --no-companions" % a.conv_mode)
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
    if a.three_phase:
        if not a.no_companions:
            raise SystemExit("--three-phase is a profiling aid: combine it with --no-companions")
        trainer.paired = False

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
    dt_local = dt
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
    strict = fast = three_phase = infer = dp1 = drop_in = None

    def note(msg):  # progress on stderr: the JSON line is the only thing on stdout
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    def mode_leg(conv_mode, steps, what):
        """the SAME step with every convolution in another arithmetic (ops.CONV_MODE), timed over `steps` steps, plus one
        serial step under the kernel timer for its dominant kernel"""
        set_serial(a.serial_streams)
        ops.CONV_MODE = conv_mode
        try:
            for _ in range(2):
                trainer.step(imgs_s, tg, imgs_t)
            torch.cuda.synchronize()
            t0s = time.time()
            for _ in range(steps):
                ls = trainer.step(imgs_s, tg, imgs_t)
            torch.cuda.synchronize()
            dts = (time.time() - t0s) / steps
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
            return {"dtype": what, "ms_per_step": round(dts * 1e3, 2), "pairs_per_s": round(B / dts, 4), "steps": steps,
                    "dominant_kernel": sdom[0], "dominant_tflops": round(sdom[1]["tflops"], 2),
                    "dominant_peak_tflops": round(peak_for(sdom[0]), 1),
                    "dominant_frac": round(sdom[1]["tflops"] / peak_for(sdom[0]), 4),
                    "dominant_avg_launch_ms": round(sdom[1]["avg_ms"], 4), "dominant_launches": sdom[1]["launches"],
                    "losses_finite": all(bool(torch.isfinite(v)) for v in ls.values())}
        finally:
            ops.CONV_MODE = HEADLINE_MODE
            set_serial(a.serial_streams)

    if world == 1 and not a.no_companions and not a.forward_target:
        # (a) the exact fp32 matrix-core kernels (v_mfma_f32_32x32x2_f32 / 16x16x4_f32 = an fp32 fma chain): the same
        # arithmetic as the headline on the 16x narrower pipe; and the two-piece split (16 significand bits per operand:
        # NARROWER than the reference's arithmetic, 2e-6 on the losses) -- what giving up the third piece would buy
        note("headline done: %.2f ms/step; companion legs" % (dt / a.steps * 1e3))
        strict = mode_leg("fp32", a.strict_steps, "f32 (exact fp32 MFMA, v_mfma_f32_32x32x2_f32 / 16x16x4_f32)")
        fast = mode_leg("bf16x3", a.steps, "bf16x3 (two bf16 pieces per operand = 16 significand bits, 3 bf16 MFMAs per "
                                           "product: narrower than the reference's fp32 multiply; not the headline)")
        note("strict %.1f ms, two-piece %.1f ms; three-phase schedule" % (strict["ms_per_step"], fast["ms_per_step"]))
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
        note("three-phase %.1f ms; inference" % three_phase["ms_per_step"])
        import scan_amd.modeling.fcos as fcos_mod
        # TEST.IMS_PER_BATCH frames per batch (reference data/build.py:119-124: images_per_gpu = TEST.IMS_PER_BATCH // num_gpus; 4 in
        # every scan yaml), of the workload's size; rounds 1-4 timed the B = 2 target frames batch by batch: kept as `two_frames`
        n_test = int(mcfg.get("test_ims_per_batch", 4))

        def time_inference(frames_):
            for _ in range(2):
                d_ = engine.inference(model, frames_)
            torch.cuda.synchronize()
            n_ = max(3, a.steps)
            t0_ = time.time()
            for _ in range(n_):
                d_ = engine.inference(model, frames_, static_weights=True)  # a dataset loop: nothing trains between batches
            torch.cuda.synchronize()
            per_call = (time.time() - t0_) / n_
            # the dataset loop (engine.inference_stream: what validation() and inference_distributed() run): batch k + 1 is
            # queued before batch k's candidate counts are read, so the post-processing round trips hide behind the next forward
            t0_ = time.time()
            for ds_ in engine.inference_stream(model, (frames_ for _ in range(n_)), static_weights=True):
                pass
            torch.cuda.synchronize()
            looped = (time.time() - t0_) / n_
            assert all(torch.equal(x, y) for d0, d1 in zip(d_, ds_) for x, y in zip(d0, d1))
            return per_call, looped, d_

        frames = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * n_test, 2234 + 100 * rank)], 32)
        dti, dts, dets = time_inference(frames)
        nms_rec = fcos_mod.last_nms_record()
        infer = {"images_per_s": round(n_test / dts, 3), "ms_per_batch": round(dts * 1e3, 2), "batch": n_test,
                 "loop": "engine.inference_stream (one batch of look-ahead), TEST.IMS_PER_BATCH frames per batch",
                 "batch_by_batch": {"images_per_s": round(n_test / dti, 3), "ms_per_batch": round(dti * 1e3, 2),
                                    "note": "engine.inference called per batch, detections read before the next call"},
                 "test_mode": mcfg["test_mode"], "detections_per_image": [int(len(d[0])) for d in dets]}
        if n_test != B:
            dti2, dts2, _ = time_inference(imgs_t)
            infer["two_frames"] = {"batch": B, "images_per_s": round(B / dts2, 3), "ms_per_batch": round(dts2 * 1e3, 2),
                                   "batch_by_batch_images_per_s": round(B / dti2, 3),
                                   "note": "the step's B target frames; batch by batch = the figure of rounds 1-4"}
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
        # (d) the data-parallel machinery with ONE rank on RCCL (gradient hooks, buckets, side-stream all-reduces, the
        # 1 / world scale, paradigm all-reduce, loss reduce): a regression there shows without a multi-GPU box
        note("inference %.1f ms per batch in the dataset loop, %.1f batch by batch; one-rank RCCL leg"
             % (infer["ms_per_batch"], infer["batch_by_batch"]["ms_per_batch"]))
        if not dist.is_initialized():
            try:
                dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                                        device_id=dev)
                tr_dp = engine.Trainer(model, settings=mcfg, distributed=True)

                def use(tr):  # the two trainers share the model: per-trainer stream roles live in model / ops attributes
                    model["middle_head"].out_stream = tr.out_stream
                    ops.WGRAD_STREAM = tr.wgrad_stream

                def timed(tr, n, reduce):
                    use(tr)
                    for _ in range(2):
                        ls_ = tr.step(imgs_s, tg, imgs_t)
                        if reduce:
                            comm.reduce_loss_dict(ls_)
                    torch.cuda.synchronize()
                    t0_ = time.time()
                    for _ in range(n):
                        ls_ = tr.step(imgs_s, tg, imgs_t)
                        if reduce:
                            comm.reduce_loss_dict(ls_)
                    torch.cuda.synchronize()
                    return (time.time() - t0_) / n

                # A B A B in this leg, minutes after the headline region: the difference to the PLAIN step of the same leg is the
                # figure (the process has warmed up since the headline; box drift is 1-2 % over a run)
                n_half = max(3, a.steps // 2)
                t_plain, t_dp = [], []
                for _ in range(2):
                    t_plain.append(timed(trainer, n_half, False))
                    t_dp.append(timed(tr_dp, n_half, True))
                dtd, dtp = sum(t_dp) / 2, sum(t_plain) / 2
                dp1 = {"ms_per_step": round(dtd * 1e3, 2), "pairs_per_s": round(B / dtd, 4), "steps": 2 * n_half,
                       "plain_ms_same_leg": round(dtp * 1e3, 2), "delta_ms": round((dtd - dtp) * 1e3, 2),
                       "blocks_ms": {"plain": [round(t * 1e3, 2) for t in t_plain], "dp1": [round(t * 1e3, 2) for t in t_dp]},
                       "collective_backend": dist.get_backend(), "ranks_in_process_group": dist.get_world_size(),
                       "gradient_allreduces_per_step": len(tr_dp.collective_log), "dp_policy": tr_dp.dp_policy,
                       "streams": "comm = side stream s2 (no fifth stream), head_out share on s1",
                       "note": "same step through engine.Trainer(distributed=True) on a one-rank RCCL group, alternating with the plain "
                               "step in the same leg (A B A B): delta_ms is the data-parallel machinery's cost at one rank"}
                ops.WGRAD_STREAM = trainer.wgrad_stream
                model["middle_head"].out_stream = trainer.out_stream
                note("one-rank RCCL leg %.1f ms" % dp1["ms_per_step"])
                del tr_dp
            except Exception as e:  # a broken collective path must not take the headline down with it
                dp1 = {"error": repr(e)}
            finally:
                if dist.is_initialized():
                    dist.destroy_process_group()

        # (e) the drop-in operator surface: what a reference-shaped NCHW, per-level module graph gets from the same kernels
        if a.surface == "layers":
            note("drop-in surface leg (scan_amd.surface on scan_amd.layers)")
            try:
                from tools import surface_bench
                ops.WGRAD_STREAM = trainer.wgrad_stream
                model["middle_head"].out_stream = trainer.out_stream
                drop_in = surface_bench.measure(trainer, imgs_s, tg, imgs_t, steps=max(3, a.steps // 4))
                drop_in["engine_ms_per_step"] = round(dt / a.steps * 1e3, 2)
                drop_in["ratio_to_engine"] = round(drop_in["ms_per_step"] / (dt / a.steps * 1e3), 3)
                note("drop-in surface %.1f ms/step, %d launches (engine %d)" % (drop_in["ms_per_step"], drop_in["launches_per_step"],
                                                                              drop_in["engine_launches_per_step"]))
            except Exception as e:  # a companion leg never costs the run its line
                drop_in = {"error": repr(e)}

    # gradient buckets of the data-parallel step: bytes and a per-link-bound ring model of their all-reduce over xGMI
    def bucket_plan(n):
        out = []
        for name, rngs, *_ in trainer._buckets():
            nbytes = 4 * sum(hi - lo for lo, hi in rngs)
            ring = 2.0 * (n - 1) / n * nbytes / (XGMI_LINK_GBS * 1e9) if n > 1 else 0.0
            out.append({"bucket": name, "bytes": nbytes, "allreduces": len(rngs),
                        "ring_ms_one_link": round(ring * 1e3, 3), "ring_ms_seven_links": round(ring * 1e3 / 7, 3)})
        return out

    n_model = world if world > 1 else 8
    buckets = bucket_plan(n_model)
    trainer_policy = trainer.dp_policy
    rank_ms = None
    if world > 1:  # per-rank step time of the timed region (before the MAX): a straggler shows as max >> min
        t = torch.tensor([dt_local / a.steps * 1e3], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        rank_ms = [round(float(x.item()), 2) for x in allt]

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
                    "peak_note": "algorithmic fp32-equivalent FLOPs; bf16x6 kernels spend 6 bf16 MFMAs per product: "
                                 "peak = 2.5 PFLOP/s dense bf16 / 6 = 416.7",
                    "launches": r["launches"], "avg_launch_ms": round(r["avg_ms"], 4),
                    "measured": "HIP events on the launch stream, %d steps with side-stream overlap off "
                                "(%.1f ms/step serial vs %.1f ms/step overlapped)" % (roof_steps, dtr / roof_steps * 1e3,
                                                                                     dt / a.steps * 1e3),
                    "all_conv_kernels": {k: {"tflops": round(v["tflops"], 2), "avg_ms": round(v["avg_ms"], 4),
                                             "launches": v["launches"],
                                             "share_of_serial_step": round(v["total_ms"] / (dtr * 1e3), 3)}
                                         for k, v in ksum.items()}}
            roof.update(prov)
            if world == 1 and not a.no_companions:
                # what the bf16 matrix pipe of THIS board sustains under its package power cap (register-only MFMA loop on
                # random operands, csrc/mfma_peak.hip): context for `frac`, which stays priced against the nominal peak
                import ctypes
                from scan_amd import _lib
                tf = ctypes.c_double(0.0)
                note("board-sustained bf16 MFMA rate (2 s register-only loop)")
                try:
                    _lib.call("scan_mfma_sustained_bf16", 2.0, 1, ctypes.byref(tf), ops._stream())
                except RuntimeError as e:  # context only: never costs the run its line
                    note("board-sustained measurement failed: %s" % e)
                per = PEAK_BF16_TFLOPS / peak_for(name)  # bf16 MFMA FLOPs per algorithmic FLOP of this kernel (6, 3, ...)
                roof["board_sustained"] = {
                    "bf16_tflops": round(tf.value, 1), "frac_of_nominal_peak": round(tf.value / PEAK_BF16_TFLOPS, 4),
                    "kernel_frac_of_it": round(r["tflops"] * per / tf.value, 4) if tf.value > 0 else None,
                    "how": "v_mfma_f32_16x16x32_bf16 from registers only, two waves per SIMD on every CU, random-sign / "
                           "random-mantissa operands, 2 s: the socket sits at its 1.4 kW cap at 2.0-2.2 GHz "
                           "(profiles/r04_mfma_peak.txt); the conv kernels run at the same cap (r04_kernel_power_clock.txt)"}
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
                   "physical_cores": cb["physical_cores"],
                   "cores_note": "value / cores = the best thread count of the sweep (torch-CPU's conv backward collapses beyond a "
                                 "few dozen threads); physical_cores = the same iteration on every physical core of the host",
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
            "dtype": "f32 (fp32 storage, fp32 multiply, fp32 accumulate -- the reference's arithmetic; the convolutions run it on "
                     "the bf16 matrix cores as bf16x6: every fp32 operand cut into 3 bf16 pieces = all 24 significand bits, "
                     "6 bf16 MFMAs per product)",
            "data": "synthetic",
            "config": {"workload": "SCAN %s %s DA iteration, %d src + %d tgt frames/GPU at %dx%d, "
                                   "forward_target=%s%s, procedural weights" % (
                           a.model.upper(), body, B, B, H, W, a.forward_target,
                           "" if a.ft_positives is None else " (%.3g of the act-map entries kept as clustering candidates)" % a.ft_positives),
                       "arithmetic": "bf16x6: exact 3-piece split of both operands, the six piece products >= 2^-24 of the "
                                     "product accumulated in fp32 (dropped terms <= 2^-23 per product, below fp32 rounding); "
                                     "distance from an fp64 conv <= the exact fp32-MFMA kernels' (tests/test_gpu_kernels.py::"
                                     "test_conv_error_vs_fp64)",
                       "lib": ident["lib"], "lib_sha1": ident["lib_sha1"], "csrc_sha1": csrc_sha1(),
                       "scan_tune_non_default": ident["scan_tune_non_default"],
                       "global_batch_pairs": B * world, "frames_per_s": round(2 * value, 4), "parallelism": "dp%d" % world,
                       "ranks_in_process_group": dist.get_world_size() if dist.is_initialized() else 1,
                       "collective_backend": dist.get_backend() if dist.is_initialized() else None,
                       "losses_finite": finite, "rank_ms_per_step": rank_ms,
                       "dp_policy": trainer_policy, "rank0_cpu_binding": binding,
                       "rccl_env": {k: os.environ[k] for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS") if k in os.environ},
                       "gradient_buckets": buckets, "gradient_buckets_modelled_for_ranks": n_model},
            "roofline": roof, "roofline_pointwise": pointwise, "cpu_baseline": cpu,
            "strict_fp32": strict, "bf16x3_two_piece": fast, "three_phase_schedule": three_phase, "inference": infer,
            "dp1_nccl": dp1, "extra": {"drop_in": drop_in},
        }
        # RCCL prints a version banner through C stdio, which is still buffered here when stdout is a pipe: push it
        # out first so the JSON line is the LAST line on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        if a.three_phase:
            line["config"]["workload"] += " [THREE-PHASE SCHEDULE, not the headline]"
        if not ident["product_library"]:
            line["metric"] = "EXPERIMENT LIBRARY %s (results wrong by construction), NOT A MEASUREMENT OF THE PRODUCT: " % ident["lib"] \
                             + line["metric"]
        elif ident["scan_tune_non_default"]:
            line["config"]["workload"] += " [SCAN_TUNE off default: %s]" % ident["scan_tune_non_default"]
        if a.conv_mode != HEADLINE_MODE:  # a profiling run of a companion arithmetic: never to be read as the headline
            what = {"bf16x3": "bf16x3: two bf16 pieces per operand = 16 significand bits, narrower than the reference",
                    "fp32": "exact fp32-MFMA kernels (v_mfma_f32_32x32x2_f32)"}[a.conv_mode]
            line["dtype"] = "COMPANION ARITHMETIC, NOT THE HEADLINE (--conv-mode %s): %s" % (a.conv_mode, what)
            line["config"]["arithmetic"] = what
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
