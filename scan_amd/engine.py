"""Domain-adaptation training / inference engine for the SCAN hot path on MI355X.

Mirrors the caller contract of the reference: ``foward_detector``
(fcos_core/engine/trainer.py:20-72) and the three-phase DA iteration of ``do_train``
(trainer.py:266-424): (1) generator on source (backward retain_graph), (2) CKA discriminators on
source through the GRL, (3) target pass + CKA discriminators on target, then one SGD step per
sub-model (solver/build.py:7-43: bias lr x2 / wd 0, momentum 0.9, constant 1/3 warm-up for 1000
iterations, solver/lr_scheduler.py:39-52).

MI355X-native pieces: every sub-model's parameters, gradients and momentum live in ONE flat fp32
buffer each (weights first, biases after), so the optimizer is two fused-SGD launches per
sub-model and data parallelism is one RCCL all-reduce per flat gradient buffer, issued on a side
HIP stream as soon as that sub-model's last backward contribution is final.
"""
import os

import torch
import torch.distributed as dist

from . import config, ops, synth
from .structures import ImageList, to_image_list  # noqa: F401
from .modeling.backbone import build_backbone
from .modeling.condgraph import build_condgraph
from .modeling.discriminator import FCOSDiscriminator_con
from .modeling import fcos as fcos_mod
from .modeling.fcos import build_fcos
from .modeling.resnet import build_resnet_fpn_backbone

# Parameters used on the side HIP streams (P4..P7 discriminators, FCOS head) get their torch-tier gradients from
# AccumulateGrad nodes that run on those streams; Trainer._join_streams orders them before anything consumes the
# gradient buffers, so autograd's "stream does not match" advice does not apply here.
if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)

LEVELS = ("P3", "P4", "P5", "P6", "P7")
DIS_ORDER = ("P7", "P6", "P5", "P4", "P3")  # order the reference builds / iterates them


# the shipped experiment definitions (scan_amd/configs/*.yaml, parsed by scan_amd/config.py: same keys as the
# reference's configs/scan/*.yaml) as the flat settings dict build_model / Trainer consume.  "k2c_r50" = BASELINE.json
# configs[3]: the K2C yaml with MODEL.BACKBONE.CONV_BODY R-50-FPN-RETINANET.
CONFIGS = {name: config.settings(config.load(name)) for name in ("c2f", "s2c", "k2c", "k2c_r50")}


def build_model(num_classes=9, test_mode="precision", device="cuda", attn_dropout=0.1, transfer_cfg=("NODES", "ADJ"),
                conv_body="VGG-16-FPN-RETINANET", settings=None):
    """dict MODEL{backbone, middle_head, fcos, dis_P*_CON} like tools/train_net_da.py:43-48,223-274.
    settings: a config.settings(cfg) dict (e.g. engine.CONFIGS["s2c"]); it overrides the four model keywords."""
    s = dict(CONFIGS["c2f"])
    s.update(num_classes=num_classes, test_mode=test_mode, transfer_cfg=tuple(transfer_cfg), conv_body=conv_body)
    if settings is not None:
        s.update(settings)
    conv_body = s["conv_body"]
    if conv_body == "VGG-16-FPN-RETINANET":
        backbone = build_backbone()
    elif conv_body in ("R-50-FPN-RETINANET", "R-101-FPN-RETINANET"):
        backbone = build_resnet_fpn_backbone(conv_body[:-len("-FPN-RETINANET")])
    else:
        raise ValueError("conv_body %r is not built" % conv_body)
    model = {
        "backbone": backbone,
        "middle_head": build_condgraph(s, 256, s["num_classes"], s["transfer_cfg"]),
        "fcos": build_fcos(s, s["num_classes"], s["test_mode"]),
    }
    model["middle_head"].multihead_attn.dropout.p = attn_dropout
    model["middle_head"].multihead_attn.attn_dropout.p = attn_dropout
    for lvl in DIS_ORDER:
        model["dis_%s_CON" % lvl] = FCOSDiscriminator_con(num_convs=s["dis_num_convs"][lvl], num_classes=s["num_classes"],
                                                          grad_reverse_lambda=s["grl_weight"][lvl])
    for m in model.values():
        m.to(device)
        for p in m.parameters():
            if p.dim() == 4:
                p.data = p.data.contiguous(memory_format=torch.channels_last)
    return model


def load_state_dicts(model, sds):
    ops.invalidate_weight_planes()  # in-place copies keep data_ptr: cached bf16 planes of the old values must go
    for k, m in model.items():
        missing, unexpected = m.load_state_dict(sds[k], strict=False)
        if missing or unexpected:
            raise RuntimeError("state_dict mismatch for %s: missing %s unexpected %s" % (k, missing, unexpected))
        for p in m.parameters():
            if p.dim() == 4 and not p.data.is_contiguous(memory_format=torch.channels_last):
                p.data = p.data.contiguous(memory_format=torch.channels_last)


def load_procedural_weights(model, num_classes=9, conv_body="VGG-16-FPN-RETINANET"):
    load_state_dicts(model, synth.all_state_dicts(num_classes, conv_body))


# ----------------------------------------------------------------------------- forward
def _plan_stream(device):
    """The stream of the ground-truth plan: side stream 0 of the process (ops.borrow_side_streams), which the P4
    discriminator uses much later in the step.  HIP multiplexes its streams onto FOUR hardware queues and a fifth ACTIVE
    queue costs the step 17 % (GPU_MAX_HW_QUEUES >= 5: 93 -> 109 ms, profiles/r05_hw_queues.txt; the same cliff as round
    3's "high-priority plan stream", profiles/r03_host_runahead.txt), so the whole process stays at the null stream + three
    side streams and every role is mapped onto those explicitly instead of by the runtime's aliasing."""
    return ops.borrow_side_streams(3)[0]


def forward_detector(model, images, targets=None, mode="source", forward_target=False):
    """reference engine/trainer.py:20-72.  images: [N,3,H,W] on the GPU or an ImageList (structures.to_image_list:
    ragged images zero-padded to a common /32 size, true sizes kept for the box clipping of inference); targets
    list of (boxes, labels).  Training: (losses, features{P3..P7: rows}, act_maps{P3..P7: rows}, shape).
    Eval: detections."""
    il = to_image_list(images)
    images = il.tensors
    in_rows = getattr(il, "rows", None)  # data.BatchCollator: the batch already in the first conv's NHWC4 layout
    dev = in_rows.device if in_rows is not None else images.device
    plan_here = bool(targets) and mode == "source" and model["middle_head"].training and dev.type == "cuda"
    if plan_here:
        inputs_ready = torch.cuda.Event()
        inputs_ready.record(torch.cuda.current_stream())
    rows, shape = model["backbone"](images, in_rows, getattr(il, "shape", None))
    if plan_here:
        # the backbone is queued (tens of ms of GPU work, ~1 ms of host time): derive everything that depends on
        # the ground truth alone on a side stream now, so its host round trips hide behind the convolutions
        fcos_mod.target_plan(shape, targets, dev, side_stream=_plan_stream(dev), after=inputs_ready)
    losses = {}
    feats, loss_graph, loss_act, maps = model["middle_head"](rows, shape, targets=targets, mode=mode,
                                                             forward_target=forward_target)
    if loss_graph is not None:
        node_loss, consistency_loss = loss_graph
        if consistency_loss is not None and not (isinstance(consistency_loss, (int, float)) and consistency_loss == 0):
            losses["consistency_loss"] = consistency_loss
        if node_loss is not None:
            losses["node_loss"] = node_loss
    if loss_act is not None:
        losses["act_loss"] = loss_act
    sizes = il.image_sizes
    proposals, proposal_losses = model["fcos"](sizes, feats, shape, targets=targets, act_maps=maps)
    if model["fcos"].training:
        losses.update(proposal_losses)
        f = dict(zip(LEVELS, ops.split_levels(feats, shape)))
        a = dict(zip(LEVELS, ops.split_levels(maps, shape)))
        return losses, f, a, shape
    return proposals


# ----------------------------------------------------------------------------- flat parameters
class FlatGroup:
    """All trainable parameters of one sub-model in one flat fp32 buffer (weights | biases), with
    matching flat gradient and momentum buffers.  Parameters and .grad become views into them."""

    @staticmethod
    def numel(module, skip=()):
        return sum(p.numel() for n, p in module.named_parameters() if p.requires_grad and not n.startswith(tuple(skip)))

    def __init__(self, module, lr, bias_lr_factor=2.0, wd=1e-4, wd_bias=0.0, momentum=0.9, flat_g=None, skip=()):
        """flat_g: optional pre-allocated, zeroed gradient storage (a slice of the Trainer's gradient arena).
        skip: name prefixes of trainable parameters that never receive a gradient on this path (middle head:
        cond_2.*, unused in RNN mode, reference condgraph.py:237 vs 315-319).  torch.optim.SGD skips parameters whose
        .grad is None -- no weight decay, no momentum buffer -- so they stay out of the flat buffers and are never
        touched; they still appear (stateless) in optimizer_state_dict like in the reference's checkpoint."""
        params = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        self.skipped = [n for n, _ in params if n.startswith(tuple(skip))] if skip else []
        live = [(n, p) for n, p in params if n not in self.skipped]
        wts = [(n, p) for n, p in live if "bias" not in n]
        bss = [(n, p) for n, p in live if "bias" in n]
        self.n_w = sum(p.numel() for _, p in wts)
        self.n_b = sum(p.numel() for _, p in bss)
        dev = params[0][1].device
        for n, p in params:
            if n in self.skipped:
                p.grad = None
        self.flat_p = torch.empty(self.n_w + self.n_b, device=dev)
        self.flat_g = torch.zeros_like(self.flat_p) if flat_g is None else flat_g
        assert self.flat_g.numel() == self.flat_p.numel()
        self.flat_m = torch.zeros_like(self.flat_p)
        self._order = wts + bss  # layout of the flat buffers
        self._named_order = [n for n, _ in params]  # the reference optimizer's parameter order
        off = 0
        self.offset = {}  # parameter name -> (offset, numel) in the flat buffers
        for pname, p in wts + bss:
            n = p.numel()
            self.offset[pname] = (off, n)
            stride = p.data.stride()
            view = self.flat_p[off:off + n].as_strided(p.shape, stride)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_g[off:off + n].as_strided(p.shape, stride)
            p._scan_flat = True  # ops.* backward kernels accumulate straight into p.grad (no autograd add)
            off += n
        self.lr, self.bias_lr_factor, self.wd, self.wd_bias, self.momentum = lr, bias_lr_factor, wd, wd_bias, momentum
        self.first = True

    def zero_grad(self):
        self.flat_g.zero_()

    # ---- optimizer state in torch.optim.SGD's state_dict layout, as the reference's make_optimizer builds it
    # (solver/build.py:7-43: ONE param group per trainable parameter, named_parameters order) -- what
    # DetectronCheckpointer stores under "optimizer_<sub-model>" (utils/checkpoint.py:201-245)
    def _logical(self):
        off = 0
        for name, p in self._order:
            n = p.numel()
            yield name, p, self.flat_m[off:off + n].as_strided(p.shape, p.data.stride())
            off += n

    def optimizer_state_dict(self, lr_factor=1.0):
        """lr_factor: the scheduler's current factor -- torch's param_groups hold the SCHEDULED lr next to
        ``initial_lr`` (what WarmupMultiStepLR leaves in a checkpoint the reference writes)."""
        by_name = {name: (p, m) for name, p, m in self._logical()}
        state, groups = {}, []
        for i, name in enumerate(self._named_order):
            m = by_name[name][1] if name in by_name else None
            bias = "bias" in name
            if not self.first and m is not None:
                state[i] = {"momentum_buffer": m.detach().clone().contiguous().cpu()}
            base = self.lr * (self.bias_lr_factor if bias else 1.0)
            groups.append({"lr": base * lr_factor, "initial_lr": base,
                           "weight_decay": self.wd_bias if bias else self.wd, "momentum": self.momentum, "dampening": 0,
                           "nesterov": False, "params": [i]})
        return {"state": state, "param_groups": groups}

    def load_optimizer_state_dict(self, sd):
        by_name = {name: m for name, p, m in self._logical()}
        if len(sd["param_groups"]) != len(self._named_order):
            raise ValueError("optimizer state has %d parameter groups, the sub-model has %d trainable parameters"
                             % (len(sd["param_groups"]), len(self._named_order)))
        self.flat_m.zero_()
        for i, name in enumerate(self._named_order):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if name not in by_name:
                continue
            if st is not None and st.get("momentum_buffer") is not None:
                by_name[name].copy_(st["momentum_buffer"].to(self.flat_m.device))
        self.first = len(sd["state"]) == 0

    def segments(self, lr_factor=1.0):
        """(p, g, buf, lr, wd, first_step) of the weight range and of the bias range, for ops.sgd_momentum_multi_"""
        lr = self.lr * lr_factor
        return [(self.flat_p[:self.n_w], self.flat_g[:self.n_w], self.flat_m[:self.n_w], lr, self.wd, self.first),
                (self.flat_p[self.n_w:], self.flat_g[self.n_w:], self.flat_m[self.n_w:], lr * self.bias_lr_factor,
                 self.wd_bias, self.first)]

    def step(self, lr_factor=1.0):
        lr = self.lr * lr_factor
        ops.sgd_momentum_(self.flat_p[:self.n_w], self.flat_g[:self.n_w], self.flat_m[:self.n_w], lr, self.wd,
                          self.momentum, self.first)
        if self.n_b:
            ops.sgd_momentum_(self.flat_p[self.n_w:], self.flat_g[self.n_w:], self.flat_m[self.n_w:],
                              lr * self.bias_lr_factor, self.wd_bias, self.momentum, self.first)
        self.first = False


def _padded_shape(il):
    if getattr(il, "rows", None) is not None:
        return (il.shape.n_images,) + tuple(il.shape.sizes[0])
    return (il.tensors.shape[0],) + tuple(il.tensors.shape[-2:])


def warmup_factor(iteration, warmup_iters=1000, factor=1.0 / 3, steps=(60000, 80000), gamma=0.1, method="constant"):
    """WarmupMultiStepLR.get_lr / base_lr at scheduler step ``iteration`` (reference solver/lr_scheduler.py:39-52;
    the scheduler is stepped after the optimizer, engine/trainer.py:418-424, so iteration i of the loop runs with
    last_epoch = i).  gamma ** bisect_right(steps, i) == gamma ** #{s <= i}."""
    if method not in ("constant", "linear"):
        raise ValueError("Only 'constant' or 'linear' warmup_method accepted, got %r" % (method,))
    f = 1.0
    if iteration < warmup_iters:
        if method == "constant":
            f = factor
        else:
            alpha = float(iteration) / warmup_iters
            f = factor * (1 - alpha) + alpha
    return f * gamma ** sum(1 for s in steps if s <= iteration)


def _transfer_active(model):
    """True when a forward_target iteration CAN yield the GST loss (TRANSFER_CFG set).  Whether it does depends on the
    rank's own target batch (DBSCAN may sample no node), and comm.reduce_loss_dict stacks the scalars in sorted-key
    order: a key present on one rank only would mis-size the reduce.  The step therefore always emits
    ``consistency_loss_gt`` in that configuration -- a zero scalar (no gradient) when this rank sampled nothing."""
    t = model["middle_head"].transfer_cfg
    return bool(t) and t[0] is not None


def _merge_ranges(rngs):
    out = []
    for a, b in sorted(rngs):
        if out and a <= out[-1][1]:
            out[-1] = (out[-1][0], max(out[-1][1], b))
        else:
            out.append((a, b))
    return out


def _complement(rngs, lo, hi):
    out, cur = [], lo
    for a, b in rngs:
        if a > cur:
            out.append((cur, a))
        cur = max(cur, b)
    if cur < hi:
        out.append((cur, hi))
    return out


# trainable parameters of a sub-model that get no gradient on this path (see FlatGroup: skip)
NO_GRAD_PARAMS = {"middle_head": ("cond_2.",)}


DP_POLICIES = ("overlap", "coarse", "tail")
DEFAULT_DP_POLICY = "overlap"


class Trainer:
    """One process per GPU.  world_size > 1: per-rank shard of the batch, local graph / normalisers,
    gradients averaged by one all-reduce per sub-model flat buffer."""

    def __init__(self, model, base_lr=None, con_dis_lambda=None, distributed=None, settings=None, dp_policy=None):
        """settings: a config.settings(cfg) dict (engine.CONFIGS[name]) -- per-sub-model SGD / WarmupMultiStepLR
        settings (SOLVER.{BACKBONE,FCOS,MIDDLE_HEAD,DIS}.*, reference solver/build.py:7-84) and CON_DIS_LAMBDA;
        default: the C2F yaml.  base_lr / con_dis_lambda override it."""
        self.model = model
        self.settings = settings = dict(settings or CONFIGS["c2f"])
        self.con_dis_lambda = settings["con_dis_lambda"] if con_dis_lambda is None else con_dis_lambda
        # ONE gradient arena for all sub-models, ordered so that what becomes final together is contiguous: the FCOS
        # head, the discriminators, then middle head and backbone.  Gradient zeroing is one fill and data parallelism
        # is one all-reduce per contiguous range (two per iteration) instead of one per sub-model.
        order = [k for k in model if k == "fcos"] + [k for k in model if k.startswith("dis_")] + \
            [k for k in model if k != "fcos" and not k.startswith("dis_")]
        sizes = {k: FlatGroup.numel(model[k], NO_GRAD_PARAMS.get(k, ())) for k in order}
        pad = lambda n: (n + 63) // 64 * 64  # keep every sub-model's base 256-byte aligned (float4 kernels)
        dev0 = next(next(iter(model.values())).parameters()).device
        self.grad_arena = torch.zeros(sum(pad(n) for n in sizes.values()), device=dev0)
        self.arena_range, off = {}, 0
        for k in order:
            self.arena_range[k] = (off, off + pad(sizes[k]))
            off += pad(sizes[k])
        self.groups, self.sched = {}, {}
        for k in order:
            sv = settings["solver"]["dis" if k.startswith("dis_") else k]
            self.groups[k] = FlatGroup(model[k], sv["lr"] if base_lr is None else base_lr, sv["bias_lr_factor"], sv["wd"],
                                       sv["wd_bias"], sv["momentum"], skip=NO_GRAD_PARAMS.get(k, ()),
                                       flat_g=self.grad_arena[self.arena_range[k][0]:self.arena_range[k][0] + sizes[k]])
            self.sched[k] = dict(warmup_iters=sv["warmup_iters"], factor=sv["warmup_factor"], steps=sv["steps"],
                                 gamma=sv["gamma"], method=sv["warmup_method"])
        self.iteration = 0
        self.distributed = dist.is_initialized() and dist.get_world_size() > 1 if distributed is None else distributed
        # gradient all-reduce policy (SCAN_DP_POLICY / dp_policy; the same on every rank -- it fixes the collective sequence):
        #   "overlap": every bucket as soon as it is final, beside the rest of the backward (six ranges per step)
        #   "coarse":  two -- FCOS head + discriminators (114 MB) once the middle head's backward starts, the rest at the end
        #   "tail":    ONE all-reduce of the whole arena after the backward, in front of the optimizer
        # Default from profiles/r06_dp_emulation.txt (tools/dp_emulate.py: a stand-in kernel with RCCL's footprint on the comm
        # stream beside the MFMA kernels, policy x workgroups, ms/step).
        self.dp_policy = (dp_policy or os.environ.get("SCAN_DP_POLICY", DEFAULT_DP_POLICY)).lower()
        if self.dp_policy not in DP_POLICIES:
            raise ValueError("dp_policy %r: one of %s" % (self.dp_policy, ", ".join(DP_POLICIES)))
        self.comm_stream = None  # set below, from the side-stream pool: the all-reduce lives INSIDE the four-queue budget
        self.comm_hook = None  # callable(lo, hi) run on the comm stream behind every all-reduce (tools/dp_emulate.py)
        if hasattr(model["backbone"], "record_grad_marks"):
            model["backbone"].record_grad_marks = bool(self.distributed)
        self._pending = []
        self._seeds = {}
        self._bucket_list = None
        self._issued, self._ready = 0, set()
        self.collective_log = []  # (lo, hi) of every gradient all-reduce issued, in order (tests; cleared per step)
        self._split_plan = ops.SplitPlan()  # the conv weights re-split at the start of every iteration, in one launch
        # The five CKA discriminators are independent; P4..P7 have 16 K ... 256 pixel rows, far too few tiles to
        # fill 256 CUs, so they run on side HIP streams next to P3 (autograd replays each backward on the stream
        # of its forward, so the backward overlaps the same way).
        self.device = next(next(iter(model.values())).parameters()).device
        on_gpu = self.device.type == "cuda"
        # Stream budget: the null stream + THREE side streams for the whole process (ops.borrow_side_streams), one hardware
        # queue each (see _plan_stream).  Roles: s0 = P4 discriminator, ground-truth plan, target forward of the
        # three-phase schedule / FCOS head beside the discriminators; s1 = P5; s2 = P6 + P7, head_out's feature share; inference
        # borrows the same three (NMS chains on s0 / s1, candidate selection on s2).  This is the mapping the runtime's own
        # aliasing of seven streams onto four queues produced in rounds 2-4, now independent of creation order.
        # SCAN_DIS_STREAMS = number of side streams the four small levels use (default 3; 2: P4 + P6 | P5 + P7; 1: all on s0)
        n_side = max(1, min(3, int(os.environ.get("SCAN_DIS_STREAMS", "3"))))
        pool = ops.borrow_side_streams(3) if on_gpu else []
        slot = {3: (0, 1, 2, 2), 2: (0, 1, 0, 1), 1: (0, 0, 0, 0)}[n_side]
        self.dis_streams = {lvl: pool[slot[i]] for i, lvl in enumerate(("P4", "P5", "P6", "P7"))} if on_gpu else {}
        self.tgt_stream = pool[0] if on_gpu else None  # (three-phase step: s0 98.2-98.5 ms, s1 99.0-99.2, s2 100.2-100.4)
        # head_out's feature share (97 % of that conv) beside the graph tier's tiny launches, forward and backward
        # (an existing side stream, idle at that point of the step -- the P7 discriminator's)
        self.out_stream = self.dis_streams.get("P7") if on_gpu and os.environ.get("SCAN_OUT_STREAM", "1") != "0" else None
        if self.distributed and on_gpu:
            # Data parallel: NO fifth stream (a fifth active hardware queue costs the step 17 %, profiles/r05_hw_queues.txt; a
            # fifth HIP stream aliases another one's queue and serialises with it).  The collectives take s2: its other role,
            # the P6 + P7 discriminators, has queued its last backward kernel before the first bucket becomes final (the
            # discriminator bucket is final when ALL discriminators have back-propagated) and its next forward comes after the
            # optimizer, which waits for the collectives -- the two roles never interleave.  head_out's feature share, the one
            # s2 role that would (it runs during the middle head's backward, beside the first buckets' all-reduce), moves to s1.
            # RCCL launches on the stream it is called on (ProcessGroupNCCL runs async_op=False collectives on the CURRENT
            # stream), so nothing else appears: null stream + three side streams, as without data parallelism.
            self.comm_stream = pool[2]
            if self.out_stream is not None:
                self.out_stream = pool[1]
        if "middle_head" in model and hasattr(model["middle_head"], "out_stream"):
            model["middle_head"].out_stream = self.out_stream
        # weight gradients of flat-buffer parameters on a stream of their own (ops.WGRAD_STREAM)
        self.wgrad_stream = torch.cuda.Stream() if on_gpu and os.environ.get("SCAN_WGRAD_STREAM", "0") != "0" else None
        ops.WGRAD_STREAM = self.wgrad_stream
        # three-phase schedule: the target forward on a side stream beside the source backward (SCAN_TGT_STREAM=0 switches it off)
        self.overlap_target = os.environ.get("SCAN_TGT_STREAM", "1") != "0"
        # paired step: the FCOS head on a side stream beside the discriminators.  SCAN_FCOS_STREAM=1 / 0 forces it; unset, it
        # follows the conv arithmetic at step time: on for the two-piece kernels (their shorter launches leave tails to fill:
        # 61 -> 60 ms/step in round 3), off for bf16x6 and fp32, whose kernels run at the package power cap -- co-running MFMA
        # kernels share one power budget and cost each other cache and LDS (same box, A B A B: 95.05 / 95.32 -> 93.98 / 94.45 ms)
        self._fcos_stream_env = os.environ.get("SCAN_FCOS_STREAM")
        self.merge_source_backward = True
        # source and target frames as ONE batch through backbone / middle head / discriminators and one backward
        # (step_paired): same losses and gradients as the three phases, larger launches.  Used when both batches
        # have the same padded size.
        self.paired = True
        self.throttle = os.environ.get("SCAN_THROTTLE", "none")
        self._throttle_ev = None

    @property
    def fcos_beside_dis(self):
        if self._fcos_stream_env is not None:
            return self._fcos_stream_env != "0"
        return ops.split_pieces() == 2

    def _allreduce_async(self, keys, after_side_streams=False):
        """all-reduce the flat gradient buffers of sub-models whose gradients are final, on the side stream."""
        if not self.distributed:
            return
        lo = min(self.arena_range[k][0] for k in keys)
        hi = max(self.arena_range[k][1] for k in keys)
        assert sum(b - a for a, b in (self.arena_range[k] for k in keys)) == hi - lo, "keys must be contiguous in the arena"
        self._allreduce_range(lo, hi, after_side_streams)

    def _allreduce_range(self, lo, hi, after_side_streams=False):
        """average grad_arena[lo:hi] over the ranks on the side stream.  The kernels that wrote the range (weight
        gradients go straight into the arena, bypassing AccumulateGrad) were launched on the current stream -- or on
        the side streams (after_side_streams) -- before this call: the comm stream is ordered after them."""
        if not self.distributed or hi <= lo:
            return
        ws = dist.get_world_size()
        self.collective_log.append((lo, hi))
        if self.comm_stream is None:  # host tensors (the gloo tests of the bucket logic): no streams to order
            g = self.grad_arena[lo:hi]
            g.div_(ws)
            dist.all_reduce(g)
            return
        self.comm_stream.wait_stream(torch.cuda.current_stream())
        if after_side_streams:  # kernels that accumulate into these buffers were queued on the side streams
            for s in self.dis_streams.values():
                self.comm_stream.wait_stream(s)
            if self.tgt_stream is not None:
                self.comm_stream.wait_stream(self.tgt_stream)
            if self.out_stream is not None:
                self.comm_stream.wait_stream(self.out_stream)
        if self.wgrad_stream is not None:  # a bucket declared final: its weight gradients may still sit on that stream
            self.comm_stream.wait_stream(self.wgrad_stream)
        with torch.cuda.stream(self.comm_stream):
            g = self.grad_arena[lo:hi]
            if dist.get_backend() == "nccl":
                # RCCL: the mean in the collective itself (no scaling pass), launched synchronously = on THIS stream
                # (ProcessGroupNCCL: async_op=False collectives run on the current stream, no internal stream, no host wait)
                dist.all_reduce(g, op=dist.ReduceOp.AVG)
            else:  # gloo with device tensors (tests/test_gpu_dp.py): staged through the host, completed at _flush_buckets
                g.div_(ws)
                self._pending.append(dist.all_reduce(g, async_op=True))
            if self.comm_hook is not None:
                self.comm_hook(lo, hi)

    # ---- gradient buckets: the ranges of the arena in the order they become final during the backward, the SAME list
    # on every rank and in both schedules (a collective sequence must not depend on a rank's batch shapes)
    def _buckets(self):
        """[(name, [(lo, hi), ...], after_side_streams)]: FCOS head | discriminators | middle head | backbone by
        stage, last first (conv5, conv4 + FPN, then conv3 + all biases).  A backbone without stage marks (ResNet) is one
        bucket."""
        if self._bucket_list is not None:
            return self._bucket_list
        self._bucket_list = self._policy_buckets(self._fine_buckets())
        return self._bucket_list

    def _policy_buckets(self, fine):
        """the bucket list of self.dp_policy out of the fine-grained one.  Entries: (name, ranges, after_side_streams,
        needs) -- needs = the hook names (fine bucket names) that must ALL have fired before the entry is final."""
        if self.dp_policy == "overlap":
            return [(n, r, s_, frozenset([n])) for n, r, s_ in fine]
        total = (min(r[0] for _, rs, _ in fine for r in rs), max(r[1] for _, rs, _ in fine for r in rs))
        if self.dp_policy == "tail":
            return [("all", [total], True, frozenset(n for n, _, _ in fine))]
        heads = [(n, rs) for n, rs, _ in fine if n in ("fcos", "dis")]
        cut = max(r[1] for _, rs in heads for r in rs)  # the arena is ordered FCOS head | discriminators | middle head | backbone
        assert min(r[0] for _, rs in heads for r in rs) == total[0]
        return [("heads", [(total[0], cut)], True, frozenset(n for n, _ in heads)),
                ("rest", [(cut, total[1])], True, frozenset(n for n, _, _ in fine if n not in ("fcos", "dis")))]

    def _fine_buckets(self):
        ar = self.arena_range
        dis = [k for k in self.groups if k.startswith("dis_")]
        out = [("fcos", [ar["fcos"]], True)]
        if dis:
            out.append(("dis", [(min(ar[k][0] for k in dis), max(ar[k][1] for k in dis))], True))
        rest = [k for k in self.groups if k != "fcos" and k not in dis and k != "backbone"]
        for k in rest:
            # the middle head's output conv back-propagates its feature share on a side stream (condgraph._out_features)
            out.append((k, [ar[k]], True))
        if "backbone" in self.groups:
            g, base = self.groups["backbone"], ar["backbone"][0]
            stages = getattr(self.model["backbone"], "grad_stage_params", None)
            if stages:  # [(mark name, [parameter names final when the mark's gradient arrives])], backward order
                for mark, names in stages:
                    rngs = _merge_ranges([(base + g.offset[n][0], base + g.offset[n][0] + g.offset[n][1])
                                          for n in names if n in g.offset])
                    if rngs:
                        out.append(("backbone:" + mark, rngs, False))
                covered = _merge_ranges([r for nm, rs, _ in out if nm.startswith("backbone:") for r in rs])
                restr = _complement(covered, ar["backbone"][0], ar["backbone"][1])
                out.append(("backbone:rest", restr, False))
            else:
                out.append(("backbone:rest", [ar["backbone"]], False))
        return out

    def _begin_buckets(self):
        self._issued = 0
        self._ready = set()

    def _bucket_ready(self, name):
        """a bucket's gradients are final.  Issue it -- and any following bucket that became ready out of order -- only
        when every earlier bucket has been issued: the collective sequence is then the canonical one on every rank
        whichever hooks fire (a hook that never fires is made up for by _flush_buckets after the backward)."""
        if not self.distributed:
            return
        self._ready.add(name)
        bl = self._buckets()
        while self._issued < len(bl) and bl[self._issued][3] <= self._ready:
            _, rngs, side, _ = bl[self._issued]
            for lo, hi in rngs:
                self._allreduce_range(lo, hi, side)
            self._issued += 1

    def _flush_buckets(self):
        """after the backward (and _join_streams): everything not reduced yet, in canonical order."""
        if not self.distributed:
            return
        bl = self._buckets()
        self.issued_before_flush = self._issued  # buckets the backward's own hooks fired (tests: all of them in the paired step)
        while self._issued < len(bl):
            _, rngs, side, _ = bl[self._issued]
            for lo, hi in rngs:
                self._allreduce_range(lo, hi, side)
            self._issued += 1
        for w in self._pending:
            w.wait()
        self._pending = []
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def _hook_ready(self, tensor, name):
        """issue bucket ``name`` when the gradient w.r.t. ``tensor`` has been computed (every backward node between the
        losses and ``tensor`` has been launched by then)."""
        if not self.distributed or tensor is None or not tensor.requires_grad:
            return

        def _h(grad):
            self._bucket_ready(name)
            return grad

        tensor.register_hook(_h)

    def _hook_backbone_marks(self):
        """hooks on the marks the backbone's last forward left (modeling/backbone.py: VGG16FPN.grad_marks)."""
        bb = self.model["backbone"]
        for mark, t in getattr(bb, "grad_marks", {}).items():
            if mark.startswith("node:"):
                # a tensor whose producer is the last node of the backward: nothing below it gets a gradient, so a tensor
                # hook (which fires BEFORE the producer's backward runs) would be too early -- hook the node itself; its
                # post-hook runs once its backward, weight-gradient launch included, has been queued
                if self.distributed and t is not None and t.grad_fn is not None:
                    def _node_done(grad_inputs, grad_outputs, _n="backbone:" + mark[5:]):
                        # weight / bias gradients written straight into the arena by the node (ops._Conv2d: flat-buffer
                        # parameters) come back as None; one that is still on its way to an AccumulateGrad node is not
                        # final yet -- then the bucket waits for _flush_buckets
                        if all(g is None for g in grad_inputs[1:3]):
                            self._bucket_ready(_n)
                    t.grad_fn.register_hook(_node_done)
                continue
            self._hook_ready(t, "middle_head" if mark == "out" else "backbone:" + mark)
        if hasattr(bb, "grad_marks"):
            bb.grad_marks = {}

    def _discriminators(self, feats, maps, shape, label, domain, tag):
        """con_dis_lambda * dis_CON(feat[l], label, act_maps[l]) for the five levels (reference trainer.py:314-333,
        373-376); the small levels run concurrently on side streams."""
        main = torch.cuda.current_stream()
        ld = {}
        for lvl in DIS_ORDER:
            i = LEVELS.index(lvl)
            side = self.dis_streams.get(lvl)
            if side is None:
                ld["loss_adv_%s_CON_%s" % (lvl, tag)] = self.con_dis_lambda * self.model["dis_%s_CON" % lvl](
                    feats[lvl], label, maps[lvl], domain=domain, shape=shape.level(i))
                continue
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ld["loss_adv_%s_CON_%s" % (lvl, tag)] = self.con_dis_lambda * self.model["dis_%s_CON" % lvl](
                    feats[lvl], label, maps[lvl], domain=domain, shape=shape.level(i))
        for side in self.dis_streams.values():
            main.wait_stream(side)
        return ld

    def _backward_terms(self, terms):
        """backward of sum_i w_i * t_i without forming the sum: every term is a root of ONE autograd pass seeded with its weight
        (a cached 0-dim constant).  Same gradients as (sum_i w_i * t_i).backward() -- the seeds are the exact partial
        derivatives -- and the same node order (the engine orders by creation sequence whatever the roots)."""
        roots, seeds = [], []
        for t, w in terms:
            if t.requires_grad:
                key = (str(t.device), float(w))
                sd = self._seeds.get(key)
                if sd is None:
                    sd = self._seeds[key] = torch.full((), float(w), device=t.device)
                roots.append(t)
                seeds.append(sd)
        torch.autograd.backward(roots, seeds)

    def _join_streams(self):
        """Backward kernels that accumulate straight into the flat gradient buffers run on the stream of their
        forward op and bypass autograd's AccumulateGrad (and with it the engine's end-of-backward stream sync):
        make the main stream wait for every side stream before gradients are reduced / consumed."""
        main = torch.cuda.current_stream()
        for s in self.dis_streams.values():
            main.wait_stream(s)
        if self.tgt_stream is not None:
            main.wait_stream(self.tgt_stream)
        if self.out_stream is not None:
            main.wait_stream(self.out_stream)
        if self.wgrad_stream is not None:
            main.wait_stream(self.wgrad_stream)

    def _optimizer_step(self):
        """optimizer.step() + scheduler.step() of every sub-model (reference engine/trainer.py:418-424)."""
        moms = {g.momentum for g in self.groups.values()}
        if len(moms) == 1 and ops.BATCHED:
            # every (weights | biases) range of every sub-model in ONE launch, each with its own lr / weight decay
            segs = []
            for k, g in self.groups.items():
                segs.extend(g.segments(warmup_factor(self.iteration, **self.sched[k])))
                g.first = False
            ops.sgd_momentum_multi_(segs, moms.pop())
        else:
            for k, g in self.groups.items():
                g.step(warmup_factor(self.iteration, **self.sched[k]))
        self.iteration += 1
        # parameters changed in place (same data_ptr): the cached bf16 hi/lo weight planes are stale from here on,
        # whoever runs next -- the next step, an in-loop validation or inference()
        ops.invalidate_weight_planes()

    def lr_of(self, sub_model, bias=False):
        """learning rate the NEXT step applies (what scheduler.get_last_lr() reports in the reference)."""
        g = self.groups[sub_model]
        return g.lr * (g.bias_lr_factor if bias else 1.0) * warmup_factor(self.iteration, **self.sched[sub_model])

    def save_checkpoint(self, save_dir, name):
        """reference DetectronCheckpointer.save (utils/checkpoint.py:141-301): the model state_dicts plus
        optimizer_<sub-model> (torch.optim.SGD layout) and the iteration, so training resumes where it stopped."""
        from . import checkpoint
        extra = {"optimizer_" + k: g.optimizer_state_dict(warmup_factor(self.iteration, **self.sched[k]))
                 for k, g in self.groups.items()}
        # scheduler_<sub-model> like DetectronCheckpointer (utils/checkpoint.py:247-295): WarmupMultiStepLR.state_dict()
        # is the scheduler's __dict__ minus the optimizer; load_checkpoint reads last_epoch back
        for k, g in self.groups.items():
            sc = self.sched[k]
            n_groups = len(g._named_order)
            extra["scheduler_" + k] = {
                "milestones": list(sc["steps"]), "gamma": sc["gamma"], "warmup_factor": sc["factor"],
                "warmup_iters": sc["warmup_iters"], "warmup_method": sc["method"], "last_epoch": self.iteration,
                "base_lrs": [g.lr * (g.bias_lr_factor if "bias" in n else 1.0) for n in g._named_order],
                "_step_count": self.iteration + 1,
                "_last_lr": [g.lr * (g.bias_lr_factor if "bias" in n else 1.0)
                             * warmup_factor(self.iteration, **sc) for n in g._named_order][:n_groups]}
        mh = self.model["middle_head"]
        return checkpoint.save(self.model, save_dir, name, iteration=self.iteration,
                               proto_counter=mh.counter_rnn.counter, **extra)

    def load_checkpoint(self, path, load_dis=True, load_opt_sch=True):
        """reference DetectronCheckpointer.load(f, load_dis, load_opt_sch) (utils/checkpoint.py:303-415).  A file written
        by the reference holds the model state_dicts plus ``optimizer_dis_P*_CON`` / ``scheduler_dis_P*_CON`` only
        (save() writes no optimizer for backbone / fcos / middle head and drops its keyword arguments, :201-295); files
        written by save_checkpoint here add ``optimizer_<sub-model>`` for every sub-model, ``iteration`` and the
        paradigm counter.  Optimizer entries found are loaded; without an ``iteration`` entry the schedulers'
        ``last_epoch`` (= optimizer steps taken) restores it.  load_opt_sch=False: weights only, like the reference's
        own call (tools/train_net_da.py:552).  Returns the entries not consumed."""
        from . import checkpoint
        ops.invalidate_weight_planes()
        rest = checkpoint.load(self.model, path, load_dis=load_dis)
        # load_state_dict copies into the parameters in place, so they stay views of the flat buffers
        sched_epochs = []
        for k, g in self.groups.items():
            sd = rest.pop("optimizer_" + k, None)
            sc = rest.pop("scheduler_" + k, None)
            if not load_opt_sch or (k.startswith("dis_") and not load_dis):
                continue
            if sd is not None:
                g.load_optimizer_state_dict(sd)
            if sc is not None and "last_epoch" in sc:
                sched_epochs.append(int(sc["last_epoch"]))
        it = rest.pop("iteration", None)
        pc = rest.pop("proto_counter", None)
        if load_opt_sch:
            if it is not None:
                self.iteration = int(it)
            elif sched_epochs:
                self.iteration = max(sched_epochs)
            if pc is not None:
                self.model["middle_head"].counter_rnn.counter = pc
        return rest

    def step_paired(self, il_s, targets_s, il_t, forward_target=False):
        """The DA iteration with the source and the target frames in one pyramid (frames [0, B) source, [B, 2B)
        target).  Per image nothing changes -- convolutions, GroupNorm, dynamic conv and the CKA towers treat images
        independently -- and everything that is per DOMAIN keeps its own rows: node sampling, paradigm update, act
        loss and the FCOS head see the source rows, each discriminator takes its source loss (label 1) and its target
        loss (label 0) on the two halves of a level.  The reference's three backward calls (trainer.py:299,343,377)
        accumulate into the same .grad, so ONE backward of the summed losses leaves identical gradients."""
        model, lam = self.model, self.con_dis_lambda
        self._throttle()
        ops.begin_weight_epoch(self._split_plan, self.device)
        fcos_mod.reset_target_plan()
        for m in model.values():
            if not m.training:  # walking every sub-module costs ~0.1 ms of host time per model
                m.train()
        self.grad_arena.zero_()
        B = len(il_s.image_sizes)
        inputs_ready = torch.cuda.Event()
        inputs_ready.record(torch.cuda.current_stream())
        if getattr(il_s, "rows", None) is not None and getattr(il_t, "rows", None) is not None:
            in_rows = torch.cat([il_s.rows, il_t.rows], 0)
            dev = in_rows.device
            rows, shape = model["backbone"](None, in_rows, ops.PyramidShape(2 * B, il_s.shape.sizes))
        else:
            images = torch.cat([il_s.tensors, il_t.tensors], 0)
            dev = images.device
            rows, shape = model["backbone"](images)
        shape_src = ops.PyramidShape(B, shape.sizes)
        fcos_mod.target_plan(shape_src, targets_s, dev, side_stream=_plan_stream(dev), after=inputs_ready)
        feats, node_loss, act_loss, maps, consistency = model["middle_head"].forward_pair(
            rows, shape, targets_s, B, forward_target=forward_target)
        losses = {"node_loss_gs": node_loss, "act_loss_gs": act_loss}
        if consistency is not None:
            losses["consistency_loss_gt"] = consistency
        elif forward_target and _transfer_active(model):
            losses["consistency_loss_gt"] = feats.new_zeros(())  # rank-invariant key set, see _transfer_active
        main = torch.cuda.current_stream()
        # the FCOS head (source rows) is independent of the discriminators: it takes the side stream the three-phase
        # schedule uses for the target forward and fills the tails of the P3 discriminator's kernels (~1.2 ms)
        fstream = self.tgt_stream if self.overlap_target and self.fcos_beside_dis else None  # (property, see __init__)
        if fstream is not None:
            fstream.wait_stream(main)
        with torch.cuda.stream(fstream if fstream is not None else main):
            src_feats, _ = ops.take_images(feats, shape, 0, B)
            _, fl = model["fcos"](il_s.image_sizes, src_feats, shape_src, targets=targets_s)
        losses.update({k + "_gs": v for k, v in fl.items()})
        # the five levels go to their discriminators behind the discriminators' gradient-reversal layers: split + GRL
        # as one node per tensor whose backward scales each level's gradient straight into its rows
        lams = [model["dis_%s_CON" % lvl].grad_reverse.lambda_ for lvl in LEVELS]
        f, a = ops.split_levels_grl(feats, shape, lams), ops.split_levels_grl(maps, shape, lams)
        adv_names, adv_raw = [], []
        for lvl in DIS_ORDER:
            i = LEVELS.index(lvl)
            side = self.dis_streams.get(lvl)
            if side is not None:
                side.wait_stream(main)
            with torch.cuda.stream(side if side is not None else main):
                ls, lt = model["dis_%s_CON" % lvl].forward_pair(f[i], a[i], shape.level(i), B, grl_applied=True)
                adv_names += ["loss_adv_%s_CON_ds" % lvl, "loss_adv_%s_CON_dt" % lvl]
                adv_raw += [ls, lt]
        for side in self.dis_streams.values():
            main.wait_stream(side)
        if fstream is not None:
            main.wait_stream(fstream)
        # the ten adversarial terms enter the objective as con_dis_lambda * loss (trainer.py:314-333,373-376).  The reported values
        # are formed in two launches (stack, scale) instead of ten, and the backward is seeded per term (_backward_terms) instead
        # of walking mul / add nodes of a summed scalar: 26 forward and 11 backward launches of one element each less, same
        # products bit for bit (lam * loss; d(lam * loss) = lam).
        terms = [(v, 1.0) for v in losses.values()] + [(v, lam) for v in adv_raw]
        scaled = torch.stack([v.detach() for v in adv_raw]) * lam
        for n, v in zip(adv_names, scaled.unbind(0)):
            losses[n] = v
        if self.distributed:
            # the backward reaches `feats` (and `maps`) only after every consumer -- the five discriminators and the
            # FCOS head -- has back-propagated: their gradient buffers (114 of the 199 MB) are final then and are
            # reduced while the middle head and the backbone still run their backward; the middle head's buffer
            # follows when the gradient reaches the backbone's output, the backbone's stage by stage (conv5, then
            # conv4 + FPN) as the marks the backbone left on its stage outputs receive theirs
            del self.collective_log[:]
            self._begin_buckets()
            pending = {"n": 2}

            def _heads_done(grad):
                pending["n"] -= 1
                if pending["n"] == 0:
                    self._bucket_ready("fcos")
                    self._bucket_ready("dis")
                return grad

            feats.register_hook(_heads_done)
            maps.register_hook(_heads_done)
            if not getattr(model["backbone"], "grad_marks", None):
                self._hook_ready(rows, "middle_head")
            self._hook_backbone_marks()
        self._mark("mid")
        self._backward_terms(terms)
        self._join_streams()
        losses["zero_gt"] = feats.new_zeros(())
        self._flush_buckets()
        self._optimizer_step()
        self._mark("end")
        return losses

    # ---- host run-ahead knob (SCAN_THROTTLE, default "none"): the step can wait, at its top, for a point of the PREVIOUS
    # iteration -- "mid" = its forward pass has been executed, "end" = all of it.  In the default configuration the one
    # host read of the ground-truth plan already keeps the host within one iteration of the GPU; the knob exists for the
    # experiments of profiles/r03_host_runahead.txt (no setting beat the default).
    def _mark(self, which):
        if self.tgt_stream is None or self.throttle == "none":
            return
        if which == self.throttle:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._throttle_ev = ev

    def _throttle(self):
        ev = getattr(self, "_throttle_ev", None)
        if ev is not None:
            ev.synchronize()
            self._throttle_ev = None

    def step(self, images_s, targets_s, images_t, forward_target=False):
        """One DA iteration; returns the loss dict (0-dim GPU tensors, reference key names)."""
        if self.paired and self.tgt_stream is not None:
            il_s, il_t = to_image_list(images_s), to_image_list(images_t)
            if _padded_shape(il_s) == _padded_shape(il_t):
                return self.step_paired(il_s, targets_s, il_t, forward_target)
        model, lam = self.model, self.con_dis_lambda
        ops.begin_weight_epoch(self._split_plan, self.device)  # parameters change once per iteration: reuse their bf16 planes within it
        fcos_mod.reset_target_plan()
        for m in model.values():
            if not m.training:  # walking every sub-module costs ~0.1 ms of host time per model
                m.train()
        self.grad_arena.zero_()
        out = {}
        # (1) generator on source
        loss_dict, feat_s, maps_s, shape = forward_detector(model, images_s, targets_s, mode="source")
        loss_dict = {k + "_gs": v for k, v in loss_dict.items()}
        # (3a) the target forward does not depend on the source backward passes (parameters are only updated
        # at the end of the iteration, the paradigm buffer was updated by the source forward): issue it now on a
        # side stream so it fills the tails of the source backward kernels
        tgt = None
        if self.overlap_target and self.tgt_stream is not None:
            main = torch.cuda.current_stream()
            self.tgt_stream.wait_stream(main)
            with torch.cuda.stream(self.tgt_stream):
                tgt = forward_detector(model, images_t, None, mode="target", forward_target=forward_target)
        # (2) discriminators on source (GRL pushes -lambda*grad into backbone / middle head).
        # The reference calls backward twice on the source graph (losses_gs with retain_graph, then the adversarial
        # losses; trainer.py:299,343).  Gradients are linear in the loss, so ONE backward of the sum leaves exactly
        # the same accumulated .grad and walks the shared backbone / middle-head graph once instead of twice.
        ld = self._discriminators(feat_s, maps_s, shape, 1.0, "source", "ds")
        if self.merge_source_backward:
            self._backward_terms([(v, 1.0) for v in loss_dict.values()] + [(v, 1.0) for v in ld.values()])
        else:
            sum(loss_dict.values()).backward(retain_graph=True)
            sum(ld.values()).backward()
        self._join_streams()
        out.update(loss_dict)
        out.update(ld)
        del loss_dict, feat_s, maps_s
        if self.distributed:
            del self.collective_log[:]
            self._begin_buckets()
            self._bucket_ready("fcos")  # the target pass adds nothing to the FCOS head
        # (3) target pass + discriminators on target
        if tgt is None:
            tgt = forward_detector(model, images_t, None, mode="target", forward_target=forward_target)
        else:
            torch.cuda.current_stream().wait_stream(self.tgt_stream)
        loss_dict, feat_t, maps_t, shape = tgt
        ld = {k + "_gt": v for k, v in loss_dict.items()}
        if forward_target and _transfer_active(model) and "consistency_loss_gt" not in ld:
            ld["consistency_loss_gt"] = feat_t["P3"].new_zeros(())  # rank-invariant key set, see _transfer_active
        ld.update(self._discriminators(feat_t, maps_t, shape, 0.0, "target", "dt"))
        if self.distributed:
            # the same bucket list in the same order as step_paired: ranks holding differently shaped batches may take
            # different schedules without mismatching a collective.  In this schedule the last contributions arrive
            # with the target backward, so the hooks sit on the target pass's tensors.
            pending = {"n": 2 * len(LEVELS)}

            def _heads_done(grad):
                pending["n"] -= 1
                if pending["n"] == 0:
                    self._bucket_ready("dis")
                return grad

            for lvl in LEVELS:
                for t in (feat_t[lvl], maps_t[lvl]):
                    if t.requires_grad:
                        t.register_hook(_heads_done)
                    else:
                        pending["n"] -= 1
            self._hook_backbone_marks()
        self._backward_terms([(v, 1.0) for k, v in ld.items() if k != "zero_gt"])
        self._join_streams()
        out.update(ld)
        self._flush_buckets()
        self._optimizer_step()
        return out


@torch.no_grad()
def inference(model, images, static_weights=False, deferred=False):
    """reference engine/inference.py:15-37 on one batch: list of (boxes, scores, labels) per image.

    static_weights=True is the caller's promise that no parameter changed since the previous inference() call (a
    dataset loop, serving): the bf16 weight planes split in that call are reused instead of being split again (31
    launches per batch).  Anything that updates parameters through this package (optimizer step, checkpoint /
    state-dict load) drops the planes regardless, so the flag only matters for writes from outside.
    deferred=True: everything is queued up to the candidate counts of the post-processing and an object is returned
    whose finish() does the rest (NMS, top-100) and returns the list -- inference_stream() overlaps batches with it."""
    for m in model.values():
        m.eval()
    if not (static_weights and ops.SPLIT_EPOCH is not None):
        ops.invalidate_weight_planes()  # weights may have been updated / loaded since the planes were cached
        ops.begin_weight_epoch()
    ops.CACHE_PLAIN_PARAMS = True  # an inference-only model holds plain nn.Parameters: their planes live as long as the epoch
    selector = model["fcos"].box_selector_test
    selector.deferred = bool(deferred)
    try:
        return forward_detector(model, images, None)
    finally:
        ops.CACHE_PLAIN_PARAMS = False
        selector.deferred = False


@torch.no_grad()
def inference_stream(model, batches, static_weights=False):
    """The dataset loop of reference engine/inference.py:15-37 (compute_on_dataset) as a generator: yields the detections of
    every batch of ``batches`` (an iterable of image batches) in order, exactly what inference() returns for it.  One batch
    of look-ahead: batch k + 1 is queued on the GPU before the host reads batch k's candidate counts, so the two host round
    trips of the post-processing and its single-workgroup NMS chains run behind / beside the next batch's convolutions
    instead of leaving the GPU idle between batches.  static_weights as in inference() for the FIRST batch; the following
    ones reuse the weight planes (nothing trains inside the loop)."""
    pending = None
    for k, images in enumerate(batches):
        nxt = inference(model, images, static_weights=static_weights or k > 0, deferred=True)
        if pending is not None:
            yield pending.finish()
        pending = nxt
    if pending is not None:
        yield pending.finish()


@torch.no_grad()
def inference_distributed(model, batches):
    """reference engine/inference.py:62-120 inference(): every rank runs compute_on_dataset (:15-37) over ITS shard --
    ``batches`` yields (images, image_ids) -- then the predictions of all ranks are accumulated on rank 0
    (_accumulate_predictions_from_multiple_gpus, :40-58) with comm.gather_detections.  Returns on rank 0 the list of
    (boxes, scores, labels) ordered by image id (CPU tensors), None on the other ranks."""
    from . import comm
    results, ids = [], []
    dev = next(next(iter(model.values())).parameters()).device
    def frames():
        for images, image_ids in batches:
            ids.extend(int(i) for i in image_ids)
            yield images

    for out in inference_stream(model, frames()):  # nothing trains between the batches of one pass
        results.extend(out)
    merged = comm.gather_detections(results, ids, device=dev)
    if merged is None:
        return None
    return [merged[i] for i in sorted(merged)]


@torch.no_grad()
def validation(model, dataset, batch_size=4, size_divisible=32, output_folder=None):
    """The in-loop validation of the DA trainer (reference engine/trainer.py:100-122 validataion() ->
    engine/inference.py:62-120 inference() -> do_coco_validation): every rank runs the detector over its share of
    ``dataset`` (items r, r + world, ... as DistributedSampler(shuffle=False) deals them, samplers/distributed.py:
    43-56), detections are taken back to each frame's original size (BoxList.resize, coco_eval.py:78-82), gathered
    on rank 0 and scored.  Returns (COCOResults, coco result dicts) on rank 0, None elsewhere; the models are left in
    eval mode (the trainer switches them back, trainer.py:484-485)."""
    from . import coco_eval, comm, data, datasets
    collate = data.BatchCollator(size_divisible)
    rank, world = comm.get_rank(), comm.get_world_size()
    mine = list(range(rank, len(dataset), world))
    dev = next(next(iter(model.values())).parameters()).device
    results, ids = [], []
    meta = []  # (image sizes, dataset indices) of the batches in flight, in order

    def frames():
        for k in range(0, len(mine), batch_size):
            il, _, idxs = collate([dataset[i] for i in mine[k:k + batch_size]])
            meta.append((il.image_sizes, idxs))
            yield il

    for dets in inference_stream(model, frames()):
        sizes, idxs = meta.pop(0)
        for (boxes, scores, labels), (h, w), idx in zip(dets, sizes, idxs):
            info = dataset.get_img_info(idx)
            boxes = datasets.resize_detections(boxes, (w, h), (info["width"], info["height"]))
            results.append((boxes, scores, labels))
            ids.append(int(idx))
    merged = comm.gather_detections(results, ids, device=dev)
    if merged is None:
        return None
    predictions = []
    for i in range(len(dataset)):
        info = dataset.get_img_info(i)
        b, s, l = merged[i]
        predictions.append((b, s, l, (info["width"], info["height"])))
    return coco_eval.do_coco_validation(dataset, predictions, output_folder)


def do_train(trainer, loader_source, loader_target, max_iter, val_dataset=None, gate=None, save_dir=None,
             checkpoint_period=0, val_batch_size=4, size_divisible=32, log=None):
    """The DA training loop around Trainer.step (reference engine/trainer.py:124-500, the with_DA branch): source and
    target batches in lock step (:269-271), ``forward_target`` from the validation gate (:350), every sub-model stepped
    once per iteration (inside Trainer.step), the loss scalars reduced to rank 0 for the meters (:76-98), validation
    every VAL_ITER iterations with a checkpoint on a new best (:465-479) -- or a checkpoint every
    ``checkpoint_period`` iterations when validation is off (:486-488) -- and ``model_final`` at the end (:489-490).

    loader_*: iterables of (ImageList, targets, ids) as data.BatchCollator returns them.  gate: coco_eval.TargetGate.
    Returns the list of per-iteration reduced loss dicts (python floats; rank 0 only, empty elsewhere)."""
    from . import comm
    from .modeling import condgraph
    if condgraph.FT_CANDIDATE_FRACTION is not None:
        raise RuntimeError("condgraph.FT_CANDIDATE_FRACTION is a measurement switch of bench.py (--ft-positives): node "
                           "sampling with it set differs from the reference's; it must be None in a training run")
    history = []
    start = trainer.iteration
    for it, ((il_s, tg_s, _), (il_t, _, _)) in enumerate(zip(loader_source, loader_target), start + 1):
        if it > max_iter:
            break
        forward_target = gate.forward_target if gate is not None else False
        dev = il_s.tensors.device if il_s.tensors is not None else il_s.rows.device
        # the ground truth stays on the host: the plan kernels upload it on their own stream (fcos._build_plan_device)
        losses = trainer.step(il_s, tg_s, il_t, forward_target=forward_target)
        reduced = comm.reduce_loss_dict(losses)
        if comm.is_main_process():
            names = sorted(reduced)  # one device->host copy for all the scalars
            vals = torch.stack([reduced[k].detach().float().reshape(()) for k in names]).tolist()
            rec = dict(zip(names, vals))
            rec["iteration"], rec["forward_target"] = it, forward_target
            history.append(rec)
            if log is not None:
                log(rec)
        if gate is not None and val_dataset is not None and gate.due(it):
            out = validation(trainer.model, val_dataset, batch_size=val_batch_size, size_divisible=size_divisible)
            best = False
            if out is not None:
                best = gate.update(out[0])
            if comm.get_world_size() > 1:  # every rank must take the same branch next iteration
                flag = torch.tensor([gate.ap50_emp, float(best)], dtype=torch.float64, device=dev)
                dist.broadcast(flag, 0)
                gate.ap50_emp, best = float(flag[0]), bool(flag[1] > 0)
                gate.best = max(gate.best, gate.ap50_emp) if best else gate.best
            if best and save_dir is not None and comm.is_main_process():
                trainer.save_checkpoint(save_dir, "model_{}_{:07d}".format(gate.best, it))
            for m in trainer.model.values():
                m.train()
        elif (gate is None or not getattr(gate, "adapt_val_on", True)) and checkpoint_period \
                and it % checkpoint_period == 0 and save_dir is not None and comm.is_main_process():
            trainer.save_checkpoint(save_dir, "model_{:07d}".format(it))
        if it == max_iter:
            if save_dir is not None and comm.is_main_process():
                trainer.save_checkpoint(save_dir, "model_final")
            break
    return history
