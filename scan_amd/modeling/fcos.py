"""FCOS head, loss and post-processor on pyramid activations.

Mirrors the reference's fcos_core/modeling/rpn/fcos/{fcos.py, loss.py, inference.py}:
same module/parameter names (head.cls_tower.N, head.bbox_tower.N, head.cls_logits,
head.bbox_pred, head.centerness, head.scales.L.scale), same loss composition.
Because a pyramid's rows are already in the reference's flatten order (level-major,
image, y, x; rpn/fcos/loss.py:191-202) the losses consume the conv outputs with no
permute/reshape/cat copies.
"""
import contextlib
import ctypes
import math

import torch
from torch import nn

from .. import ops
from ..layers import IOULoss, Scale, SigmoidFocalLoss
from .backbone import conv_holder

INF = 100000000  # reference rpn/fcos/loss.py:22
SIZES_OF_INTEREST = ((-1, 64), (64, 128), (128, 256), (256, 512), (512, INF))  # loss.py:41-47
FPN_STRIDES = (8, 16, 32, 64, 128)


def make_tower(n, c=256):
    layers = []
    for _ in range(n):
        layers += [conv_holder(c, c, 3), nn.GroupNorm(32, c), nn.ReLU()]
    return nn.Sequential(*layers)


def run_tower(tower, rows, shape, n, out_buf=None):
    """n x [conv3x3 (MFMA), GroupNorm(32)+ReLU (fused HIP)] sharing weights across the levels:
    one launch per layer covers the whole pyramid.  out_buf: a wider [M, C + e] matrix whose first C columns receive
    the tower's output (ops.groupnorm_relu(out_buf=...)); the returned rows are that column slice."""
    for i in range(n):
        conv, gn = tower[3 * i], tower[3 * i + 1]
        rows = ops.conv2d(rows, conv.weight, conv.bias, shape, 3, 1, gn_sums=True)  # GN sums from the conv epilogue
        rows = ops.groupnorm_relu(rows, gn.weight, gn.bias, shape, relu=True, eps=gn.eps,
                                  out_buf=out_buf if i == n - 1 else None)
    return rows


def compute_locations(shape, device, strides=FPN_STRIDES):
    """reference fcos.py:234-258 / condgraph.py:631-655: x-fastest grid + stride // 2, one per level."""
    locs = []
    for (h, w), s in zip(shape.sizes, strides):
        xs = torch.arange(0, w * s, step=s, dtype=torch.float32, device=device)
        ys = torch.arange(0, h * s, step=s, dtype=torch.float32, device=device)
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        locs.append(torch.stack((xx.reshape(-1), yy.reshape(-1)), dim=1) + s // 2)
    return locs


class TargetPlan:
    """Everything the source pass derives from the ground truth alone (no network output involved): the FCOS
    assignment (reference loss.py:40-126, and its copy at :262-343 used by the middle head), the positive rows
    with their regression / centerness targets (loss.py:128-133,197-222) and the graph-node sampling index
    (loss.py:428-463).  It is built once per batch -- by engine.forward_detector on a side stream while the
    backbone convolutions are queued on the main one, so its host round trips (nonzero) cost no GPU idle time."""
    __slots__ = ("key", "targets", "labels", "labels_i32", "reg_targets", "pos_inds", "n_pos", "reg_pos", "ctr_pos",
                 "node_index", "node_labels", "ready")


_plan = [None]


def reset_target_plan():
    """engine.Trainer calls this at the top of every iteration: a plan never outlives its batch."""
    _plan[0] = None


def source_node_index(labels, shape):
    """Row indices and labels of the graph nodes the source branch samples (reference loss.py:428-463): per level,
    positives in row order; negatives = floor(linspace(0, n_neg-2, n_pos)) of the background rows (all of them
    when n_pos > n_neg); final order [all neg, all pos]."""
    import numpy as np
    pos, neg = [], []
    for l in range(shape.n_levels):
        r0, r1 = shape.row_off[l], shape.row_off[l + 1]
        lab = labels[r0:r1]
        pi = torch.nonzero(lab > 0).squeeze(1)
        ni = torch.nonzero(lab == 0).squeeze(1)
        n_pos, n_neg = pi.numel(), ni.numel()
        if n_pos <= n_neg:
            idx = np.floor(np.linspace(0, n_neg - 2, n_pos)).astype(np.int64)
            ni = ni[torch.from_numpy(idx).to(ni.device)]
        pos.append(pi + r0)
        neg.append(ni + r0)
    pos, neg = torch.cat(pos, 0), torch.cat(neg, 0)
    return torch.cat([neg, pos], 0), torch.cat([labels.new_zeros(neg.numel()), labels[pos]], 0)


def target_plan(shape, targets, device, side_stream=None, after=None):
    """The TargetPlan of (shape, targets); cached for the current batch.  With ``side_stream`` the plan is built
    there (after event ``after``) and the calling stream is made to wait for it."""
    key = (id(targets), tuple(shape.sizes), shape.n_images, str(device))
    p = _plan[0]
    if p is not None and p.key == key and p.targets is targets:
        return p
    cur = torch.cuda.current_stream() if device.type == "cuda" else None
    if side_stream is not None and cur is not None:
        host_targets = bool(targets) and all(not b.is_cuda and not l.is_cuda for b, l in targets)
        if after is not None and not host_targets:  # device-resident ground truth was produced on the calling stream
            side_stream.wait_event(after)
        with torch.cuda.stream(side_stream):
            p = _build_plan(shape, targets, device)
            p.ready = torch.cuda.Event()
            p.ready.record(side_stream)
        cur.wait_event(p.ready)
        for name in TargetPlan.__slots__:
            t = getattr(p, name, None)
            if isinstance(t, torch.Tensor):
                t.record_stream(cur)
    else:
        p = _build_plan(shape, targets, device)
    p.key, p.targets = key, targets
    _plan[0] = p
    return p


DEVICE_PLAN = True  # False: the torch spelling below also on the GPU (A/B, cross-checks)
_plan_staging = {}  # (N, device) -> pinned host buffers of the packed ground truth, grown (never shrunk) to a power of two of boxes


def _build_plan_device(shape, targets, device):
    """The plan in three launches of csrc/targets.hip and ONE host read (the five per-level positive counts) instead of
    ~115 torch launches with a dozen host round trips; bit-identical to the torch spelling below."""
    p = TargetPlan()
    p.ready = None
    N, L, M = shape.n_images, shape.n_levels, shape.rows
    G = max(1, max(int(b.shape[0]) for b, _ in targets))
    if all(not b.is_cuda and not l.is_cuda for b, l in targets):
        # ground truth as the collator hands it over (host tensors): packed in pinned staging buffers and uploaded on
        # THIS stream -- the plan then depends on nothing the main stream has queued, and the one host read below waits
        # for three small kernels instead of for the previous iteration (tools/host_profile.py: 33 of 52 ms per step)
        # one staging set per (batch size, device): the largest box count of a batch changes nearly every iteration on
        # real data, and a set per count would page-lock a new buffer (milliseconds, on the hot path) each time and never
        # give it back.  The kernels' loop is bounded by each image's own count (ng), so a wider G costs nothing.
        key = (N, str(device))
        stage = _plan_staging.get(key)
        if stage is None or stage[0].shape[1] < G:
            G_cap = 1 << max(4, (G - 1).bit_length())
            stage = _plan_staging[key] = (torch.zeros((N, G_cap, 4), dtype=torch.float32).pin_memory(),
                                          torch.zeros((N, G_cap), dtype=torch.int64).pin_memory(),
                                          torch.zeros((N,), dtype=torch.int32).pin_memory())
        hb, hl, hn = stage
        G = hb.shape[1]
        hb.zero_()
        hl.zero_()
        for i, (b, l) in enumerate(targets):
            g = int(b.shape[0])
            hb[i, :g] = b.float()
            hl[i, :g] = l.long()
            hn[i] = g
        boxes, glab, ng = (t.to(device, non_blocking=True) for t in (hb, hl, hn))
    else:
        boxes = torch.zeros((N, G, 4), dtype=torch.float32, device=device)
        glab = torch.zeros((N, G), dtype=torch.int64, device=device)
        for i, (b, l) in enumerate(targets):
            g = int(b.shape[0])
            if g:
                boxes[i, :g] = b.to(device=device, dtype=torch.float32)
                glab[i, :g] = l.to(device=device, dtype=torch.int64)
        ng = torch.tensor([int(b.shape[0]) for b, _ in targets], dtype=torch.int32).to(device)
    st = ops._stream()
    p.labels = torch.empty((M,), dtype=torch.int64, device=device)
    p.labels_i32 = torch.empty((M,), dtype=torch.int32, device=device)
    p.reg_targets = torch.empty((M, 4), dtype=torch.float32, device=device)
    level_pos = torch.empty((8,), dtype=torch.int32, device=device)
    pos_list = torch.empty((M,), dtype=torch.int32, device=device)
    neg_list = torch.empty((M,), dtype=torch.int32, device=device)
    strides_h = (ctypes.c_int32 * L)(*FPN_STRIDES[:L])
    soi_h = (ctypes.c_float * (2 * L))(*[float(v) for l in range(L) for v in SIZES_OF_INTEREST[l]])
    ops.call("scan_fcos_assign", shape.ref(), strides_h, soi_h, ops._ptr(boxes), ops._ptr(glab), ops._ptr(ng), G,
             ops._ptr(p.labels), ops._ptr(p.labels_i32), ops._ptr(p.reg_targets), ops._ptr(level_pos), st)
    ops.call("scan_fcos_compact", shape.ref(), ops._ptr(p.labels), ops._ptr(pos_list), ops._ptr(neg_list), st)
    counts = level_pos[:L].tolist()  # the one host round trip of the plan
    cnt_h = (ctypes.c_int32 * L)(*counts)
    n_nodes = ops.query("scan_fcos_nodes_count", shape.ref(), cnt_h)
    p.n_pos = int(sum(counts))
    p.node_index = torch.empty((n_nodes,), dtype=torch.int64, device=device)
    p.node_labels = torch.empty((n_nodes,), dtype=torch.int64, device=device)
    p.pos_inds = torch.empty((p.n_pos,), dtype=torch.int64, device=device)
    p.reg_pos = torch.empty((p.n_pos, 4), dtype=torch.float32, device=device)
    ctr = torch.empty((p.n_pos,), dtype=torch.float32, device=device)
    ops.call("scan_fcos_nodes", shape.ref(), cnt_h, ops._ptr(p.labels), ops._ptr(p.reg_targets), ops._ptr(pos_list),
             ops._ptr(neg_list), ops._ptr(p.node_index), ops._ptr(p.node_labels), ops._ptr(p.pos_inds),
             ops._ptr(p.reg_pos), ops._ptr(ctr), st)
    p.ctr_pos = ctr if p.n_pos > 0 else None
    return p


def _build_plan(shape, targets, device):
    if DEVICE_PLAN and device.type == "cuda" and targets and all(int(b.shape[0]) > 0 for b, _ in targets):
        return _build_plan_device(shape, targets, device)
    p = TargetPlan()
    p.ready = None
    p.labels, p.reg_targets = assign_targets(compute_locations(shape, device), targets)
    p.labels_i32 = p.labels.int()
    p.node_index, p.node_labels = source_node_index(p.labels, shape)
    p.pos_inds = torch.nonzero(p.labels > 0).squeeze(1)
    p.n_pos = p.pos_inds.numel()
    p.reg_pos = p.reg_targets[p.pos_inds]
    p.ctr_pos = centerness_targets(p.reg_pos) if p.n_pos > 0 else None
    return p


def assign_targets(locations, targets):
    """FCOS location -> GT assignment (reference loss.py:40-126; PrototypeComputation has an identical
    copy at :262-343).  targets: list of (boxes [G,4] xyxy, labels [G] int64) per image.
    Returns labels [M] int64 and reg targets [M,4] in pyramid row order (level-major, image, y, x)."""
    npl = [len(l) for l in locations]
    dev = locations[0].device
    soi = torch.cat([torch.tensor(SIZES_OF_INTEREST[l], dtype=torch.float32, device=dev)[None].expand(n, -1)
                     for l, n in enumerate(npl)], 0)
    pts = torch.cat(locations, 0)
    xs, ys = pts[:, 0], pts[:, 1]
    labels, regs = [], []
    for boxes, lab in targets:
        boxes = boxes.to(dev).float()
        lab = lab.to(dev)
        area = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)  # BoxList.area(): +1 widths
        l = xs[:, None] - boxes[:, 0][None]
        t = ys[:, None] - boxes[:, 1][None]
        r = boxes[:, 2][None] - xs[:, None]
        b = boxes[:, 3][None] - ys[:, None]
        reg = torch.stack([l, t, r, b], dim=2)
        inside = reg.min(dim=2)[0] > 0
        mx = reg.max(dim=2)[0]
        cared = (mx >= soi[:, [0]]) & (mx <= soi[:, [1]])
        a = area[None].repeat(len(pts), 1)
        a[~inside] = INF
        a[~cared] = INF
        amin, gi = a.min(dim=1)
        reg = reg[torch.arange(len(pts), device=dev), gi]
        lb = lab[gi].clone()
        lb[amin == INF] = 0
        labels.append(torch.split(lb, npl, 0))
        regs.append(torch.split(reg, npl, 0))
    nl = len(locations)
    lab_rows = torch.cat([torch.cat([li[l] for li in labels], 0) for l in range(nl)], 0)
    reg_rows = torch.cat([torch.cat([ri[l] for ri in regs], 0) for l in range(nl)], 0)
    return lab_rows, reg_rows


def centerness_targets(reg):
    """reference loss.py:128-133."""
    lr = reg[:, [0, 2]]
    tb = reg[:, [1, 3]]
    return torch.sqrt((lr.min(-1)[0] / lr.max(-1)[0]) * (tb.min(-1)[0] / tb.max(-1)[0]))


class FCOSHead(nn.Module):
    """reference fcos.py:13-114 with REG_CTR_ON True."""

    def __init__(self, num_classes=9, num_convs=4, prior_prob=0.01):
        super().__init__()
        self.num_fg = num_classes - 1
        self.num_convs = num_convs
        self.cls_tower = make_tower(num_convs)
        self.bbox_tower = make_tower(num_convs)
        self.cls_logits = conv_holder(256, self.num_fg, 3)
        self.bbox_pred = conv_holder(256, 4, 3)
        self.centerness = conv_holder(256, 1, 3)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, std=0.01)
                nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.cls_logits.bias, -math.log((1 - prior_prob) / prior_prob))
        self.scales = nn.ModuleList([Scale(init_value=1.0) for _ in range(5)])

    def forward(self, rows, shape, need_cls=True):
        """-> logits [M,C] (or None), bbox_reg [M,4] = exp(scale_l * bbox_pred), centerness [M]."""
        logits = None
        if need_cls:
            ct = run_tower(self.cls_tower, rows, shape, self.num_convs)
            logits = ops.conv2d(ct, self.cls_logits.weight, self.cls_logits.bias, shape, 3, 1)[:, :self.num_fg]
        rt = run_tower(self.bbox_tower, rows, shape, self.num_convs)
        # bbox_pred (4) and centerness (1) read the same tower: one skinny conv with 5 outputs
        w = torch.cat([self.bbox_pred.weight, self.centerness.weight], 0)
        b = torch.cat([self.bbox_pred.bias, self.centerness.bias], 0)
        out = ops.conv2d(rt, w, b, shape, 3, 1, cout_s=8)
        scale_rows = torch.cat([self.scales[l].scale.expand(shape.row_off[l + 1] - shape.row_off[l])
                                for l in range(shape.n_levels)], 0)
        bbox_reg = torch.exp(out[:, :4] * scale_rows[:, None])
        return logits, bbox_reg, out[:, 4]


class FCOSLossComputation:
    """reference loss.py:25-230."""

    def __init__(self, gamma=2.0, alpha=0.25):
        self.cls_loss_func = SigmoidFocalLoss(gamma, alpha)
        self.box_reg_loss_func = IOULoss()

    def __call__(self, shape, box_cls, box_regression, centerness, targets):
        N = shape.n_images
        plan = target_plan(shape, targets, box_cls.device)
        pos_inds, reg_targets, ctr_t = plan.pos_inds, plan.reg_pos, plan.ctr_pos
        cls_loss = self.cls_loss_func(box_cls.contiguous(), plan.labels_i32) / (plan.n_pos + N)
        box_regression = box_regression[pos_inds]
        centerness = centerness[pos_inds]
        if plan.n_pos > 0:
            reg_loss = self.box_reg_loss_func(box_regression, reg_targets, ctr_t)
            ctr_loss = ops.bce_with_logits_mean(centerness, ctr_t)
        else:
            reg_loss = box_regression.sum()
            ctr_loss = centerness.sum()
        return cls_loss, reg_loss, ctr_loss


_last_nms = [None]


def last_nms_record():
    """(boxes, scores, labels, threshold) of the most recent per-image NMS launch of the post-processor -- bench.py
    re-times the NMS kernel alone on a real candidate set with it.  None before the first inference."""
    return _last_nms[0]


class _PendingDetections:
    """Second half of FCOSPostProcessor.__call__: everything behind the one host round trip of the selection (the candidate
    counts).  The counts travel to pinned host memory asynchronously and an event marks them, so a dataset loop can queue
    the NEXT batch's forward before it reads this batch's counts (engine.inference_stream): the round trips of batch k
    then hide behind the convolutions of batch k + 1 and its NMS chains run beside them on the side streams."""

    def __init__(self, proc, ok, det, val, lab, n_images, deferred):
        self.proc, self.ok, self.det, self.val, self.lab, self.N, self.deferred = proc, ok, det, val, lab, n_images, deferred
        self.main = torch.cuda.current_stream() if det.is_cuda else None
        self.results = None
        self.inputs = None
        if self.main is not None:
            self.counts = torch.empty((n_images,), dtype=torch.int64, pin_memory=True)
            self.counts.copy_(ok.sum(1), non_blocking=True)
            self.ready = torch.cuda.Event()
            self.ready.record(self.main)  # also what the side streams wait for: nothing queued on main after it
        else:
            self.counts = ok.sum(1)

    def finish(self):
        """per image (boxes [k,4], scores [k], labels [k]); idempotent."""
        if self.results is not None:
            return self.results
        proc, ok, det, val, lab, N, main = self.proc, self.ok, self.det, self.val, self.lab, self.N, self.main
        dev = det.device
        if main is not None:
            self.ready.synchronize()  # the one host round trip of the selection
        counts = self.counts.tolist()
        # Phase 1, no host read: every image's candidates are gathered (their number is known from `counts`, so
        # nonzero_static needs no round trip) and its NMS is queued -- image i on side stream i % 2, since one NMS is a chain
        # of single-workgroup kernels that leaves the GPU to the next image's.  Phase 2 reads the kept counts.
        use_side = main is not None and (N > 1 or self.deferred)
        if use_side and proc._nms_streams is None:
            proc._nms_streams = ops.borrow_side_streams(2)  # the trainer's, when there is one in the process
        now = torch.cuda.current_stream() if main is not None else None  # == main unless the caller switched streams
        pending = []
        for i in range(N):
            if counts[i] == 0:
                pending.append(None)
                continue
            side = proc._nms_streams[i % 2] if use_side else None
            if side is not None:
                side.wait_event(self.ready)
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                sel = ok[i].nonzero_static(size=counts[i]).squeeze(1)
                boxes, scores, labels = det[i][sel], torch.sqrt(val[i][sel]), lab[i][sel]
                # per-class NMS (reference inference.py:160-176 loops classes and calls boxlist_nms on each) as ONE
                # class-aware launch: a box is suppressed only by a kept, higher-scored box of the SAME label, which is
                # exactly greedy NMS run per class; output order = class-major, original index ascending within a class
                _last_nms[0] = (boxes, scores, labels, proc.nms_thresh)
                if side is not None:  # kept beyond this call (bench.py times the NMS on them on the main stream)
                    for t in (boxes, scores, labels):
                        t.record_stream(now)
                finish = ops.nms_by_label_async(boxes, scores, labels, proc.nms_thresh)
            pending.append((boxes, scores, labels, finish, side))
        results = []
        for i in range(N):
            if pending[i] is None:
                results.append((det.new_zeros((0, 4)), det.new_zeros((0,)), lab.new_zeros((0,))))
                continue
            boxes, scores, labels, finish, side = pending[i]
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                keep = finish().to(dev)
                keep = keep[torch.argsort(labels[keep], stable=True)]
                rb, rs, rl = boxes[keep], scores[keep], labels[keep]
                n = len(rs)
                if n > proc.fpn_post_nms_top_n > 0:
                    th, _ = torch.kthvalue(rs, n - proc.fpn_post_nms_top_n + 1)
                    k = torch.nonzero(rs >= th).squeeze(1)
                    rb, rs, rl = rb[k], rs[k], rl[k]
            if side is not None:
                now.wait_stream(side)
                for t in (rb, rs, rl):
                    t.record_stream(now)
            results.append((rb, rs, rl))
        self.results = results
        self.ok = self.det = self.val = self.lab = self.inputs = None  # (released only now: the side streams read them until here)
        return results


class FCOSPostProcessor:
    """reference inference.py:20-194; per-class NMS runs on the device (scan_nms)."""

    def __init__(self, pre_nms_thresh=0.05, pre_nms_top_n=1000, nms_thresh=0.6, fpn_post_nms_top_n=100, min_size=0,
                 num_classes=9, mode="common"):
        self.pre_nms_thresh = pre_nms_thresh
        self.pre_nms_top_n = pre_nms_top_n
        self.nms_thresh = nms_thresh
        self.fpn_post_nms_top_n = fpn_post_nms_top_n
        self.min_size = min_size
        self.num_classes = num_classes
        self.mode = mode
        self._nms_streams = None  # two side streams for the per-image NMS chains, made on first use with a batch > 1
        self.deferred = False  # True: __call__ returns the pending object instead of finishing it (engine.inference_stream)
        self._select_stream = None  # deferred mode: the stream the selection is queued on

    def __call__(self, shape, box_cls, box_regression, centerness, image_sizes):
        """see _select.  With self.deferred on a GPU the selection itself is queued on a side stream of its own (its ~400
        small launches -- one top-k, sort and gather per level -- then run beside the NEXT batch's convolutions instead of
        in front of them) and the pending object keeps the head's outputs alive until finish()."""
        if not (self.deferred and box_cls.is_cuda):
            return self._select(shape, box_cls, box_regression, centerness, image_sizes)
        main = torch.cuda.current_stream()
        if self._select_stream is None:
            self._select_stream = ops.borrow_side_streams(3)[2]
        self._select_stream.wait_stream(main)
        with torch.cuda.stream(self._select_stream):
            pend = self._select(shape, box_cls, box_regression, centerness, image_sizes)
        pend.inputs = (box_cls, box_regression, centerness)  # read on the side stream: not to be reused by main before finish()
        # ... and should the pending object be dropped without finish() (an exception unwinding the dataset loop), the caching
        # allocator must still know who touched what: the head outputs (allocated on main) were read on the selection stream,
        # the selection's results (allocated there) are read on main and the NMS streams
        for t in pend.inputs:
            t.record_stream(self._select_stream)
        for t in (pend.ok, pend.det, pend.val, pend.lab):
            t.record_stream(main)
            for s_ in (self._nms_streams or ops.borrow_side_streams(2)):
                t.record_stream(s_)
        return pend

    def _select(self, shape, box_cls, box_regression, centerness, image_sizes):
        """box_cls [M,C] (logits for 'common', fused probabilities otherwise), box_regression [M,4],
        centerness [M] logits.  Returns per image (boxes [k,4], scores [k], labels [k]) -- or, with self.deferred, the
        _PendingDetections whose finish() returns them.

        Same selection as the reference (inference.py:49-118: candidates = score > pre_nms_thresh, at most
        pre_nms_top_n per image and level by score x centerness, decode, clip, min-size; :140-194: per-class NMS,
        top fpn_post_nms_top_n by kthvalue) but batched on the device: one top-k per level over all images, one
        decode over everything selected, one class-aware NMS launch per image -- no host loop over levels x images x
        classes and one host round trip for the candidate counts instead of one per (level, image)."""
        N, C = shape.n_images, box_cls.shape[1]
        dev = box_cls.device
        locs = compute_locations(shape, dev)
        prob = box_cls.sigmoid() if self.mode == "common" else box_cls
        score = torch.where(prob > self.pre_nms_thresh, prob * centerness.sigmoid()[:, None], prob.new_full((), -1.0))
        sel_score, sel_cls, sel_row, sel_loc = [], [], [], []
        for l in range(shape.n_levels):
            r0, r1 = shape.row_off[l], shape.row_off[l + 1]
            hw = (r1 - r0) // N
            flat = score[r0:r1].reshape(N, hw * C)
            k = min(self.pre_nms_top_n, hw * C)
            val, idx = flat.topk(k, dim=1, sorted=False)
            # candidates in flattened (location, class) order like the reference's nonzero(); non-candidates last
            idx, perm = torch.sort(torch.where(val > 0, idx, idx.new_full((), hw * C)), dim=1)
            val = val.gather(1, perm)
            loc = torch.div(idx, C, rounding_mode="floor").clamp(max=hw - 1)
            sel_score.append(val)
            sel_cls.append(idx % C + 1)
            sel_row.append(r0 + torch.arange(N, device=dev)[:, None] * hw + loc)
            sel_loc.append(locs[l][loc])
        val = torch.cat(sel_score, 1)  # [N, K]
        lab = torch.cat(sel_cls, 1)
        rg = box_regression[torch.cat(sel_row, 1)]  # [N, K, 4]
        lc = torch.cat(sel_loc, 1)  # [N, K, 2]
        det = torch.stack([lc[..., 0] - rg[..., 0], lc[..., 1] - rg[..., 1], lc[..., 0] + rg[..., 2],
                           lc[..., 1] + rg[..., 3]], -1)
        # BoxList.clip_to_image (TO_REMOVE = 1) against each image's true size
        lim = torch.tensor([[w - 1, h - 1, w - 1, h - 1] for h, w in image_sizes], dtype=det.dtype, pin_memory=det.is_cuda)
        lim = lim.to(dev, non_blocking=True)
        det = torch.minimum(det.clamp(min=0), lim[:, None, :])
        ws, hs = det[..., 2] - det[..., 0] + 1, det[..., 3] - det[..., 1] + 1
        ok = (val > 0) & (ws >= self.min_size) & (hs >= self.min_size)
        pend = _PendingDetections(self, ok, det, val, lab, N, self.deferred)
        return pend if self.deferred else pend.finish()


class FCOSModule(nn.Module):
    """model["fcos"] (reference fcos.py:117-258)."""

    def __init__(self, num_classes=9, mode="precision", cfg=None):
        super().__init__()
        c = cfg or {}
        if c.get("num_convs_cls", 4) != c.get("num_convs_reg", 4):
            raise ValueError("MODEL.FCOS.NUM_CONVS_CLS != NUM_CONVS_REG is not built")
        self.head = FCOSHead(num_classes, c.get("num_convs_cls", 4), c.get("prior_prob", 0.01))
        self.loss_evaluator = FCOSLossComputation(c.get("loss_gamma", 2.0), c.get("loss_alpha", 0.25))
        # reference rpn/fcos/inference.py:197-217 make_fcos_postprocessor
        self.box_selector_test = FCOSPostProcessor(
            pre_nms_thresh=c.get("inference_th", 0.05), pre_nms_top_n=c.get("pre_nms_top_n", 1000),
            nms_thresh=c.get("nms_th", 0.6), fpn_post_nms_top_n=c.get("detections_per_img", 100), min_size=0,
            num_classes=num_classes, mode=mode)
        self.mode = mode

    def forward(self, image_sizes, rows, shape, targets=None, act_maps=None):
        if self.training:
            if targets is None:
                # reference fcos.py:215-220: the target pass only yields an identically-zero loss whose
                # gradients are zero, so the head is not run at all (SURVEY.md 8d "dead work")
                return None, {"zero": rows.new_zeros(())}
            logits, reg, ctr = self.head(rows, shape)
            lc, lr, lctr = self.loss_evaluator(shape, logits, reg, ctr, targets)
            return None, {"loss_cls": lc, "loss_reg": lr, "loss_centerness": lctr}
        logits, reg, ctr = self.head(rows, shape, need_cls=self.mode != "light")
        if self.mode == "light":
            logits = act_maps[:, 1:]
        elif self.mode == "precision":
            logits = 0.5 * logits.sigmoid() + 0.5 * act_maps[:, 1:]
        return self.box_selector_test(shape, logits, reg, ctr, image_sizes), {}


def build_fcos(cfg=None, num_classes=9, mode="precision"):
    """reference rpn/rpn.py:201 build_rpn(cfg, in_channels) for FCOS_ON; cfg: a config.settings dict or None."""
    if cfg is not None:
        return FCOSModule(cfg["num_classes"], cfg["test_mode"], cfg)
    return FCOSModule(num_classes, mode)
