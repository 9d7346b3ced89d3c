"""torch.autograd wrappers around the C ABI (include/scan_hip.h).

PyTorch is plumbing here: it owns device memory, the current HIP stream and the
autograd tape; all arithmetic of the ops below happens in libscan_hip.so.
Every op requires CUDA(HIP) tensors and raises otherwise -- there is no CPU
fallback in the product path.

Activations are "pyramids": a contiguous fp32 matrix [M, Cs] (rows = pixels in
level-major / image / y / x order, Cs = channel stride, a multiple of 4) plus a
``PyramidShape`` describing how rows split into levels (see include/scan_hip.h).
"""
import contextlib
import ctypes
import os
import weakref

import torch

from . import _lib
from ._lib import PyramidDesc, call, query


# ----------------------------------------------------------------------------- plumbing
def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    # the raw handle of torch's current stream on the current device; the Python-level torch.cuda.current_stream() builds a
    # Stream object per call (~10 us, 900 calls per training step)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("scan_amd ops run only on the GPU (HIP) -- got a %s tensor; no CPU fallback" % t.device)
        if t.dtype == torch.float32 and not t.is_contiguous():
            raise RuntimeError("scan_amd op got a non-contiguous tensor")


def pad4(c):
    return (c + 3) // 4 * 4


class KernelTimer:
    """Optional HIP-event timing of the MFMA conv launches on the stream they run on (bench.py uses it
    for the live roofline figure).  Disabled by default: no events are recorded."""

    def __init__(self):
        self.enabled = False
        self.records = {}

    def begin(self, name, flops):
        if not self.enabled:
            return None
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.setdefault(name, []).append((s, e, flops))
        s.record(torch.cuda.current_stream())
        return e

    def end(self, e):
        if e is not None:
            e.record(torch.cuda.current_stream())

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in recs)
            fl = sum(f for _, _, f in recs)
            out[name] = {"launches": len(recs), "total_ms": ms, "avg_ms": ms / len(recs), "flops": fl,
                         "tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0}
        return out

    def reset(self):
        self.records = {}


kernel_timer = KernelTimer()


class PyramidShape:
    """Level table of a pyramid activation (mirrors scan_pyramid_t)."""

    def __init__(self, n_images, sizes):
        self.n_images = int(n_images)
        self.sizes = [(int(h), int(w)) for h, w in sizes]
        self.row_off = [0]
        for h, w in self.sizes:
            self.row_off.append(self.row_off[-1] + self.n_images * h * w)
        d = PyramidDesc()
        d.n_levels = len(self.sizes)
        d.n_images = self.n_images
        for i, (h, w) in enumerate(self.sizes):
            d.h[i], d.w[i] = h, w
        for i, o in enumerate(self.row_off):
            d.row_off[i] = o
        self.desc = d

    @property
    def rows(self):
        return self.row_off[-1]

    @property
    def n_levels(self):
        return len(self.sizes)

    def level(self, l):
        return PyramidShape(self.n_images, [self.sizes[l]])

    def conv_out(self, ksize, stride):
        pad = ksize // 2
        return PyramidShape(self.n_images, [((h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1)
                                            for h, w in self.sizes])

    def ref(self):
        return ctypes.byref(self.desc)

    def __eq__(self, o):
        return isinstance(o, PyramidShape) and self.n_images == o.n_images and self.sizes == o.sizes

    def __repr__(self):
        return "PyramidShape(N=%d, %s)" % (self.n_images, self.sizes)


def nchw_to_rows(x, cs=None):
    """[N,C,H,W] -> ([N*H*W, Cs] rows, PyramidShape); zero-pads channels to Cs."""
    N, C, H, W = x.shape
    cs = cs or pad4(C)
    rows = x.permute(0, 2, 3, 1).reshape(N * H * W, C)
    if cs != C:
        rows = torch.nn.functional.pad(rows, (0, cs - C))
    return rows.contiguous(), PyramidShape(N, [(H, W)])


def rows_to_nchw(rows, shape, l=0, c=None):
    """view of level l as a logical [N,C,H,W] tensor (channels_last strides)."""
    h, w = shape.sizes[l]
    r = rows[shape.row_off[l]:shape.row_off[l + 1]]
    if c is not None:
        r = r[:, :c]
    return r.view(shape.n_images, h, w, r.shape[1]).permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------- weights
def pack_weight(weight, cs):
    """[Cout,Cin,k,k] parameter -> [Cout, k*k, Cs] fp32 ("OHWI").  Zero-copy when the
    parameter is stored channels_last and Cin == Cs."""
    co, ci, kh, kw = weight.shape
    w = weight.permute(0, 2, 3, 1)  # [Cout,k,k,Cin]
    if ci == cs and w.is_contiguous():
        return w.reshape(co, kh * kw, ci)
    out = weight.new_zeros(co, kh * kw, cs)
    out[:, :, :ci] = w.reshape(co, kh * kw, ci)
    return out


def unpack_weight_grad(dw, weight):
    """[Cout, T, Cs] -> gradient with the parameter's logical shape (channels_last strides)."""
    co, ci, kh, kw = weight.shape
    return dw[:, :, :ci].reshape(co, kh, kw, ci).permute(0, 3, 1, 2)


# How the 3x3 / stride-1 and 1x1 convolutions multiply (csrc/conv_split.h):
#   "bf16x6" (default): on the bf16 matrix cores with every fp32 operand cut into THREE bf16 pieces (all 24 significand
#            bits) and six piece products per product -- the reference's fp32 multiply / fp32 accumulate up to summation order;
#   "bf16x3": two pieces (16 significand bits per operand), three piece products: 2e-6 on the losses of a DA iteration, 2x
#            the rate of bf16x6;
#   "fp32":  everything on v_mfma_f32_32x32x2_f32 (bit-for-bit an fp32 fma chain), the slowest.
CONV_MODE = "bf16x6"
SPLIT_MODES = {"bf16x6": 3, "bf16x3": 2}


def split_pieces():
    """bf16 pieces per fp32 operand in the current CONV_MODE (0: the exact fp32-MFMA kernels)."""
    return SPLIT_MODES.get(CONV_MODE, 0)


def _round32(c):
    return (c + 31) // 32 * 32


def _round8(c):
    return (c + 7) // 8 * 8


# bf16 hi/lo planes of the parameters are valid for one optimizer step: engine.Trainer bumps SPLIT_EPOCH at the
# start of every iteration and the planes of flat-buffer parameters are reused by the source / target passes.
# SCAN_BATCHED=0 in the environment (or ops.BATCHED = False): the one-launch-per-tensor paths this build replaced --
# per-weight plane splits, per-group SGD launches, the torch construction of the discriminators' stacked weights, the
# separate GroupNorm statistics finalisation -- for same-box A/B measurements (profiles/r02_batched_ab.txt).  Same
# results either way.
BATCHED = os.environ.get("SCAN_BATCHED", "1") != "0"
# SCAN_CAT_IN_PLACE=0: the CKA discriminators build their class-branch input with torch.cat (+ the contiguous() copy of
# its gradient slice in the GroupNorm backward) instead of normalising into place, for A/B.  Same values.
CAT_IN_PLACE = os.environ.get("SCAN_CAT_IN_PLACE", "1") != "0"
# SCAN_FPN_DIRECT=0: the FPN output convs write tensors of their own that are concatenated afterwards, and the top-down
# join is an up-sampling copy + add (torch), for A/B.  Same values.
FPN_DIRECT = os.environ.get("SCAN_FPN_DIRECT", "1") != "0"
# Side streams other parts may borrow for short independent chains (the post-processor's per-image NMS): engine.Trainer
# registers the ones it made.  Borrowing instead of creating matters: one more HIP stream in the process shifts the stream ->
# hardware-queue assignment of all the others (profiles/r03_head_out_split.txt).
SIDE_STREAMS = []


def borrow_side_streams(n):
    while len(SIDE_STREAMS) < n:
        SIDE_STREAMS.append(torch.cuda.Stream())
    return SIDE_STREAMS[:n]


# Stream the weight-gradient launches of flat-buffer parameters are queued on (engine.Trainer sets it; None = the stream
# of the backward pass).  A weight gradient is a leaf of the backward graph -- nothing but the optimizer step (and the
# gradient all-reduce) reads it -- so on a stream of its own it fills what the dependent chain leaves idle: the graph
# tier's tiny launches, the tails of launches that are 1.4 waves of workgroups long.
WGRAD_STREAM = None
# SCAN_HEAD_OUT_SPLIT=0: the middle head's output conv runs as ONE conv over cat(features, act maps) as in the reference,
# instead of feature share + act-map share (modeling/condgraph.py: _out_features), for A/B.  Same sum, other rounding order.
HEAD_OUT_SPLIT = os.environ.get("SCAN_HEAD_OUT_SPLIT", "1") != "0"
SPLIT_EPOCH = None
_split_cache = {}
_epoch_counter = [0]
_active_plan = None


class SplitPlan:
    """The conv weights one owner (an engine.Trainer) needs as bf16 planes in every iteration.  The first iteration
    splits them one launch per weight as they are met and records (parameter, geometry, plane buffers) here; every later
    begin_weight_epoch(plan) re-splits ALL of them with ONE launch (scan_weight_split_batched) into the same buffers
    -- ~120 launches of 3-8 us on the critical path become one.  A job holds only a weak reference to its parameter:
    it is dropped when the parameter dies, and a weight that stops matching (another model at the same address) misses
    on the shape test in _conv_split and replaces its job."""

    def __init__(self):
        self.jobs = {}     # (data_ptr, mode, csw, pieces) -> (weakref to the parameter, O, T, cs_w, mode, rows, csw, planes)
        self.table = None  # device int64 [n_jobs, SPLIT_JOB_WORDS]
        self.order = []
        self.blocks = 0
        self.dirty = True

    def add(self, key, param, O, T, cs_w, mode, rows, csw, planes):
        self.jobs[key] = (weakref.ref(param), O, T, cs_w, mode, rows, csw, planes)
        self.dirty = True

    def run(self):
        # a job also dies when the arithmetic it was recorded for is no longer the current one (ops.CONV_MODE switched from
        # three pieces to two or back: k[3] is the piece count) -- otherwise every epoch would keep re-splitting, and
        # holding, the planes of BOTH modes
        npc = split_pieces()
        dead = [k for k, j in self.jobs.items() if j[0]() is None or j[0]().data_ptr() != k[0] or (npc > 0 and k[3] != npc)]
        for k in dead:
            _split_cache.pop(k, None)
            del self.jobs[k]
        if dead:
            self.dirty = True
        if not self.jobs:
            return
        if self.dirty:
            rows, off = [], 0
            self.order = list(self.jobs.keys())
            for k in self.order:
                _, O, T, cs_w, mode, r, csw, planes = self.jobs[k]
                rows.append([k[0], planes[0].data_ptr(), planes[1].data_ptr(), O, T, cs_w, mode, r, csw, off,
                             planes[2].data_ptr() if len(planes) == 3 else 0])
                off += query("scan_weight_split_job_blocks", O, T, cs_w, mode, csw)
            dev = self.jobs[self.order[0]][7][0].device
            self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
            self.blocks = off
            self.dirty = False
        call("scan_weight_split_batched", _ptr(self.table), len(self.order), int(self.table.shape[1]), self.blocks, _stream())
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        for k in self.order:
            _split_cache[k] = (SPLIT_EPOCH, self.jobs[k][7], ev)


# GroupNorm workspaces (fp64 sums the kernels accumulate into with atomics) must start at zero.  Cleared one by one that is a
# memset launch per conv-with-sums and per GroupNorm backward, 60 per training step; instead they are slices of one buffer
# that begin_weight_epoch() clears with ONE memset (what the previous iteration used of it), and the library is told not to
# clear (scan_conv3x3_gn_acc_bf16x3; bit 1 of scan_groupnorm_relu_backward's accumulate).  Outside an epoch, or when the buffer
# runs out, the workspace is a fresh allocation the library call clears itself.
ZERO_POOL = os.environ.get("SCAN_ZERO_POOL", "1") != "0"
_zero_pool = {"buf": None, "off": 0, "active": False}
_ZERO_POOL_DOUBLES = 1 << 20


def _ws_f64(n, device):
    """n doubles on `device` for a kernel that accumulates into them -> (tensor, True if it is already zero: a slice of
    the pool, valid until the next begin_weight_epoch; False: uninitialised, the library call clears it)."""
    zp = _zero_pool
    n_al = (n + 31) // 32 * 32
    if ZERO_POOL and zp["active"] and zp["buf"] is not None and zp["buf"].device == device \
            and zp["off"] + n_al <= _ZERO_POOL_DOUBLES:
        out = zp["buf"][zp["off"]:zp["off"] + n]
        zp["off"] += n_al
        return out, True
    return torch.empty((n,), dtype=torch.float64, device=device), False


_loss_zero = {"buf": None, "off": 0}


def _zeros_f32(n, device):
    """n zeroed floats for a loss kernel's accumulators.  Inside a training iteration they are carved from ONE torch.zeros
    allocated per iteration (a NEW tensor every iteration, so loss scalars that are views of it stay valid after the next
    iteration starts) -- ten fill launches per step become one."""
    lz = _loss_zero
    n_al = (n + 15) // 16 * 16
    if ZERO_POOL and _zero_pool["active"] and lz["buf"] is not None and lz["buf"].device == device \
            and lz["off"] + n_al <= lz["buf"].numel():
        out = lz["buf"][lz["off"]:lz["off"] + n]
        lz["off"] += n_al
        return out
    return torch.zeros((n,), dtype=torch.float32, device=device)


def _reset_zero_pool(device):
    _loss_zero["buf"] = torch.zeros((1024,), dtype=torch.float32, device=device) \
        if ZERO_POOL and device is not None and device.type == "cuda" else None
    _loss_zero["off"] = 0
    zp = _zero_pool
    if not ZERO_POOL or device is None or device.type != "cuda":
        zp["active"] = False
        return
    if zp["buf"] is None or zp["buf"].device != device:
        zp["buf"] = torch.zeros((_ZERO_POOL_DOUBLES,), dtype=torch.float64, device=device)
    elif zp["off"] > 0:
        zp["buf"][:zp["off"]].zero_()
    zp["off"] = 0
    zp["active"] = True


# True only inside engine.inference(): plain nn.Parameters (not re-homed into a Trainer's flat buffers) may then use the plane
# cache too.  Anywhere else a cached plane of a Parameter could outlive its model (another model's weight at the same address
# and shape would be served the old planes), so the switch is off and such weights are split per call.
CACHE_PLAIN_PARAMS = False


def begin_weight_epoch(plan=None, device=None):
    """Start of a span in which parameters do not change (one training iteration): the bf16 planes split inside it are
    reused by every launch that reads the same weight (forward, data gradient, source / target passes).  With a
    SplitPlan the planes it knows are produced right here, in one launch."""
    global SPLIT_EPOCH, _active_plan
    _epoch_counter[0] += 1
    SPLIT_EPOCH = _epoch_counter[0]
    _split_cache.clear()
    _active_plan = plan if BATCHED else None
    _reset_zero_pool(device)
    if _active_plan is not None:
        _active_plan.run()


def invalidate_weight_planes():
    """Parameters are about to change or have changed in place (optimizer step, state_dict / checkpoint load): drop
    every cached plane and stop caching until the next begin_weight_epoch().  In-place updates keep data_ptr, so a
    stale entry would otherwise be served to e.g. inference() after training."""
    global SPLIT_EPOCH, _active_plan
    SPLIT_EPOCH = None
    _active_plan = None
    _zero_pool["active"] = False
    _split_cache.clear()
    if _EXT_INVALIDATE is not None:  # the compiled operators' own plane cache (scan_ops_ext.cpp), when that module is loaded
        _EXT_INVALIDATE()


_EXT_INVALIDATE = None  # set by scan_amd.layers when scan_ops._ops is imported


# GroupNorm sums produced by a conv epilogue, handed to the groupnorm_relu call that consumes that conv's output
_gn_sums = {}


def _conv_split(x, shape, wp, cout, rows_out, cs_src, mode, bias, relu, ns, name, flops, cache_key=None,
                mask=None, dst_shape=None, cmap=0, pool=False, gn_sums=False, param=None, out=None, pieces=None):
    """Forward / data gradient of a 3x3 / stride-1 or 1x1 conv on the bf16 matrix cores with split operands (pieces = 3:
    "bf16x6", 2: "bf16x3"; default: the current CONV_MODE).  wp: packed fp32 weights [O][T][Cs_w], T = 9 or 1 (1x1; dst_shape = output pyramid, cmap =
    scan_conv1x1_*'s map).  mode 0: forward (Nout = O); mode 1: dgrad (Nout = Cs_w)."""
    st = _stream()
    npc = pieces or split_pieces()
    sfx = "bf16x6" if npc == 3 else "bf16x3"
    O, T, cs_w = wp.shape
    # plane rows are zero-padded to whole 32-channel K chunks for the 3x3 kernels: the LDS-DMA weight path needs whole
    # chunks (a 264-channel input then takes it too); 1x1 planes keep the 8-element granule
    rnd = _round32 if T == 9 else _round8
    if mode == 0:
        rows, csw, nout = O, rnd(cs_w), O
    else:
        rows, csw, nout = cs_w, rnd(max(O, cs_src)), cs_w
    hit = None
    if cache_key is not None and SPLIT_EPOCH is not None:
        key = (cache_key, mode, csw, npc)
        hit = _split_cache.get(key)
        if hit is not None and (hit[0] != SPLIT_EPOCH or tuple(hit[1][0].shape) != (rows, T, csw)):
            hit = None
    if hit is not None:
        planes = hit[1]
        # planes written on another stream (e.g. the target pass on its side stream) must be complete
        torch.cuda.current_stream().wait_event(hit[2])
    else:
        planes = tuple(torch.empty((rows, T, csw), dtype=torch.bfloat16, device=x.device) for _ in range(npc))
        if npc == 3:
            call("scan_weight_split3", _ptr(wp), O, T, cs_w, mode, _ptr(planes[0]), _ptr(planes[1]), _ptr(planes[2]), csw, st)
        else:
            call("scan_weight_split", _ptr(wp), O, T, cs_w, mode, _ptr(planes[0]), _ptr(planes[1]), csw, st)
        if cache_key is not None and SPLIT_EPOCH is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            _split_cache[(cache_key, mode, csw, npc)] = (SPLIT_EPOCH, planes, ev)
            if _active_plan is not None and param is not None and wp.data_ptr() == cache_key == param.data_ptr():
                _active_plan.add((cache_key, mode, csw, npc), param, O, T, cs_w, mode, rows, csw, planes)
    wptrs = [_ptr(t) for t in planes]
    y = out if out is not None else (x.new_zeros if ns != nout else x.new_empty)((rows_out, ns))
    if kernel_timer.enabled:  # label the record with the template instance the launch takes (64 / 128 / 256 channels)
        if T == 9:
            inst = query("scan_conv3x3_%s_instance" % sfx, (dst_shape or shape).ref(), nout)
        elif npc == 3:
            inst = query("scan_conv1x1_bf16x6_instance", (dst_shape or shape).ref(), nout, csw)
        else:
            inst = 128 if nout > 64 else 64
        ev = kernel_timer.begin("%s_bn%d" % (name, inst), flops)
    else:
        ev = None
    if gn_sums:
        sums, cleared = _ws_f64(shape.n_levels * shape.n_images * 32 * 2, x.device)
        if npc == 3:
            call("scan_conv3x3_gn_bf16x6", _ptr(x), shape.ref(), cs_src, *wptrs, csw, _ptr(bias), _ptr(y), nout, ns,
                 _ptr(sums), 0 if cleared else 1, st)
        else:
            call("scan_conv3x3_gn_acc_bf16x3" if cleared else "scan_conv3x3_gn_bf16x3", _ptr(x), shape.ref(), cs_src, *wptrs, csw,
                 _ptr(bias), _ptr(y), nout, ns, _ptr(sums), st)
        _gn_sums.clear()  # at most one pending hand-over: conv and its GroupNorm are adjacent calls of one thread
        _gn_sums[y.data_ptr()] = sums
    elif pool:
        call("scan_conv3x3_pool2_" + sfx, _ptr(x), shape.ref(), cs_src, *wptrs, csw, _ptr(bias), _ptr(y),
             nout, ns, int(bool(relu)), st)
    elif T == 1:
        call("scan_conv1x1_" + sfx, _ptr(x), shape.ref(), cs_src, *wptrs, csw, _ptr(bias), _ptr(mask),
             _ptr(y), (dst_shape or shape).ref(), nout, ns, int(bool(relu)), cmap, st)
    else:
        fn = "scan_conv3x3_" + sfx
        rem = nout % 128
        if nout > 128 and 0 < rem <= 64 and cs_src >= 512 and rows_out >= 100000:
            # 128-wide output tiles plus a small remainder (data gradient of the 264-channel discriminator input at
            # P3, K = 1024): the remainder columns go through the 64-channel instance instead of a third, almost empty
            # 128-wide tile (2022 -> 1794 us).  With a short K loop or few rows the extra launch costs more than the
            # empty tile (265-channel head_out input: 614 -> 721 us), hence the size test.
            main = nout - rem

            def off(t, nbytes):
                return ctypes.c_void_p(t.data_ptr() + nbytes) if t is not None else ctypes.c_void_p(0)

            call(fn, _ptr(x), shape.ref(), cs_src, *wptrs, csw, _ptr(bias), _ptr(mask), _ptr(y), main, ns,
                 int(bool(relu)), st)
            call(fn, _ptr(x), shape.ref(), cs_src, *[off(t, main * T * csw * 2) for t in planes], csw, off(bias, main * 4),
                 off(mask, main * 4), off(y, main * 4), rem, ns, int(bool(relu)), st)
        else:
            call(fn, _ptr(x), shape.ref(), cs_src, *wptrs, csw, _ptr(bias), _ptr(mask), _ptr(y), nout, ns,
                 int(bool(relu)), st)
    kernel_timer.end(ev)
    return y


class _Conv2d(torch.autograd.Function):
    """conv (k in {1,3,5,7}, stride in {1,2}, pad k//2) + bias (+ ReLU) on a pyramid.

    relu = "deferred": the forward applies the ReLU, the backward does NOT mask dy -- the contract is that every
    consumer of y multiplies its dx by (y > 0) itself (mask_dx=True on a conv, relu_input=True on the max-pool),
    which those kernels do for free in their epilogue; it removes the separate relu-backward pass over dy."""

    @staticmethod
    def forward(ctx, x, weight, bias, shape, ksize, stride, relu, cout_s, mask_dx=False, pool=False, gn_sums=False,
                out=None):
        _chk(x, bias)
        if not weight.is_cuda:
            raise RuntimeError("scan_amd ops run only on the GPU (HIP); no CPU fallback")
        cs = x.shape[1]
        assert x.shape[0] == shape.rows, (x.shape, shape)
        cout = weight.shape[0]
        cout_s = cout_s or pad4(cout)
        wp = pack_weight(weight, cs)
        oshape = shape.conv_out(ksize, stride)
        if out is not None:  # write into a caller-provided row block (a slice of a pyramid buffer, see assemble_rows)
            if pool or not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
                            and tuple(out.shape) == (oshape.rows, cout_s) and cout_s == cout):
                raise RuntimeError("conv2d(out=...): needs a contiguous fp32 [%d, %d] GPU block with Cout == Cout_s"
                                   % (oshape.rows, cout_s))
        fast = split_pieces() > 0 and ((ksize == 3 and stride == 1) or ksize == 1)
        tag = CONV_MODE  # kernel_timer labels
        flops = 2.0 * oshape.rows * cout * ksize * ksize * weight.shape[1]
        # planes are cached per weight epoch for tensors whose storage outlives a call: flat-buffer parameters (a Trainer's)
        # and plain nn.Parameters (an inference-only model: engine.inference(static_weights=True) keeps the epoch open across
        # batches -- without this its 32 conv weights were re-split every batch).  A temporary (torch.cat of two heads'
        # weights, stacked discriminator weights) is never cached: the allocator hands its address to the next temporary
        ckey = weight.data_ptr() if (getattr(weight, "_scan_flat", False)
                                     or (CACHE_PLAIN_PARAMS and isinstance(weight, torch.nn.Parameter))) else None
        # first layer (3 input channels): dedicated K = taps x 4 forward kernel (the backward, if any, is generic)
        first = split_pieces() > 0 and cs == 4 and cout <= 64 and (ksize, stride) in ((3, 1), (7, 2)) \
            and shape.n_levels == 1
        if first:
            (h, w_), n = shape.sizes[0], shape.n_images
            y = out if out is not None else (x.new_zeros if cout_s != cout else x.new_empty)((oshape.rows, cout_s))
            ev = kernel_timer.begin("conv_smallcin_" + tag, flops)
            call("scan_conv_smallcin_" + tag, _ptr(x), n, h, w_, _ptr(wp), _ptr(bias), _ptr(y), cout, cout_s, ksize,
                 stride, int(bool(relu)), _stream())
            kernel_timer.end(ev)
        elif fast:
            if pool:  # conv + ReLU + 2x2 max-pool in one launch: forward only (frozen stages)
                if ksize != 3 or shape.n_levels != 1:
                    raise RuntimeError("conv2d(pool=True) needs a 3x3 conv on a single-level pyramid")
                (h, w_) = shape.sizes[0]
                y = _conv_split(x, shape, wp, cout, shape.n_images * (h // 2) * (w_ // 2), cs, 0, bias, relu, cout_s,
                                    "conv3x3_%s_fwd" % tag, flops, cache_key=ckey, pool=True, param=weight)
            else:
                y = _conv_split(x, shape, wp, cout, oshape.rows, cs, 0, bias, relu, cout_s,
                                    ("conv3x3_%s_fwd" if ksize == 3 else "conv1x1_%s_fwd") % tag, flops,
                                    cache_key=ckey, dst_shape=oshape, cmap=stride - 1, param=weight,
                                    gn_sums=gn_sums and ksize == 3 and cout == 256 and cout_s == 256 and not relu,
                                    out=out)
        else:
            y = out if out is not None else (x.new_zeros if cout_s != cout else x.new_empty)((oshape.rows, cout_s))
            ev = kernel_timer.begin("conv_igemm_fwd", flops)
            call("scan_conv2d_forward", _ptr(x), shape.ref(), cs, _ptr(wp), _ptr(bias), _ptr(y), oshape.ref(), cout,
                 cout_s, ksize, stride, int(bool(relu)), _stream())
            kernel_timer.end(ev)
        ctx.save_for_backward(x, weight, y if relu is True else None)
        ctx.cfg = (shape, oshape, ksize, stride, relu, cout_s, bias is not None, fast)
        ctx.conv_mode = CONV_MODE  # the backward runs on the kernels of the mode the forward ran in
        ctx.mask_dx = mask_dx
        ctx.ckey = ckey
        # parameters re-homed into a flat gradient buffer (engine.FlatGroup): accumulate straight into it
        ctx.wgrad_buf = ctx.bgrad_buf = None
        if getattr(weight, "_scan_flat", False) and weight.grad is not None and weight.shape[1] == cs \
                and weight.grad.permute(0, 2, 3, 1).is_contiguous():
            ctx.wgrad_buf = weight.grad
            if bias is not None and getattr(bias, "_scan_flat", False) and bias.grad is not None:
                ctx.bgrad_buf = bias.grad
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        shape, oshape, ksize, stride, relu, cout_s, has_bias, fast = ctx.cfg
        cs = x.shape[1]
        cout, cin = weight.shape[0], weight.shape[1]
        T = ksize * ksize
        st = _stream()
        sfx = ctx.conv_mode if ctx.conv_mode in SPLIT_MODES else "bf16x3"
        dy = dy.contiguous()
        if relu is True:
            g = torch.empty_like(dy)
            call("scan_relu_backward", _ptr(dy), _ptr(y), _ptr(g), dy.numel(), st)
            dy = g
        dx = dw = db = None
        mask = x if ctx.mask_dx else None  # x is a deferred-ReLU output: dx *= (x > 0) in the dgrad epilogue
        if ctx.needs_input_grad[0] and fast:
            dx = _conv_split(dy, oshape, pack_weight(weight, cs), cout, x.shape[0], cout_s, 1, None, False, cs,
                                 ("conv3x3_%s_dgrad" if ksize == 3 else "conv1x1_%s_dgrad") % sfx,
                                 2.0 * oshape.rows * cout * T * cin, cache_key=ctx.ckey, mask=mask, dst_shape=shape,
                                 cmap=0 if stride == 1 else 2, param=weight, pieces=SPLIT_MODES[sfx])
        elif ctx.needs_input_grad[0]:
            wp = pack_weight(weight, cs)
            wt = x.new_empty((cs, T, cout_s))
            call("scan_weight_transpose", _ptr(wp), cout, T, cs, _ptr(wt), cout_s, st)
            dx = torch.empty_like(x)
            ev = kernel_timer.begin("conv_igemm_dgrad", 2.0 * oshape.rows * cout * T * cin)
            call("scan_conv2d_dgrad", _ptr(dy), oshape.ref(), cout_s, _ptr(wt), _ptr(dx), shape.ref(), cs, cs, ksize,
                 stride, _ptr(mask), st)
            kernel_timer.end(ev)
        direct_w = ctx.wgrad_buf is not None
        direct_b = direct_w and ctx.bgrad_buf is not None
        db_done = False
        if ctx.needs_input_grad[1] and fast and ksize == 3:
            want_db = has_bias and ctx.needs_input_grad[2]
            side = WGRAD_STREAM if (direct_w and (direct_b or not want_db) and not kernel_timer.enabled) else None
            if side is not None:  # in place into the flat gradient buffers, nothing returned to autograd: off the chain
                side.wait_stream(torch.cuda.current_stream())
                x.record_stream(side)
                dy.record_stream(side)
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                ws = x.new_empty((query("scan_conv3x3_wgrad_%s_ws_floats" % sfx, shape.ref(), cs, cout),))
                dwp = ctx.wgrad_buf if direct_w else x.new_empty((cout, T, cs))
                ev = kernel_timer.begin("conv3x3_%s_wgrad" % sfx, 2.0 * oshape.rows * cout * T * cin)
                if want_db:
                    db = ctx.bgrad_buf if direct_b else x.new_empty((cout,))
                    db_done = True
                call("scan_conv3x3_wgrad_" + sfx, _ptr(x), shape.ref(), cs, _ptr(dy), cout, cout_s, _ptr(dwp),
                     _ptr(db) if want_db else None, int(direct_w), _ptr(ws), _stream())
                kernel_timer.end(ev)
            if want_db and direct_w and not direct_b:
                raise RuntimeError("conv2d: flat weight gradient without a flat bias gradient")
            dw = None if direct_w else unpack_weight_grad(dwp, weight)
            if direct_b:
                db = None
        elif ctx.needs_input_grad[1] and fast:  # 1x1, stride 1 or 2
            ws = x.new_empty((query("scan_conv1x1_wgrad_%s_ws_floats" % sfx, oshape.ref(), cs, cout),))
            dwp = ctx.wgrad_buf if direct_w else x.new_empty((cout, T, cs))
            ev = kernel_timer.begin("conv1x1_%s_wgrad" % sfx, 2.0 * oshape.rows * cout * T * cin)
            want_db = has_bias and ctx.needs_input_grad[2]
            if want_db:
                db = ctx.bgrad_buf if direct_b else x.new_empty((cout,))
                db_done = True
            call("scan_conv1x1_wgrad_" + sfx, _ptr(x), shape.ref(), cs, _ptr(dy), oshape.ref(), cout, cout_s, stride,
                 _ptr(dwp), _ptr(db) if want_db else None, int(direct_w), _ptr(ws), st)
            kernel_timer.end(ev)
            if want_db and direct_w and not direct_b:
                raise RuntimeError("conv2d: flat weight gradient without a flat bias gradient")
            dw = None if direct_w else unpack_weight_grad(dwp, weight)
            if direct_b:
                db = None
        elif ctx.needs_input_grad[1]:
            n = query("scan_conv2d_wgrad_ws_floats", oshape.ref(), cs, cout, ksize)
            ws = x.new_empty((n,))
            dwp = ctx.wgrad_buf if direct_w else x.new_empty((cout, T, cs))
            ev = kernel_timer.begin("conv_wgrad", 2.0 * oshape.rows * cout * T * cin)
            call("scan_conv2d_wgrad", _ptr(x), shape.ref(), cs, _ptr(dy), oshape.ref(), cout, cout_s, ksize, stride,
                 _ptr(dwp), int(direct_w), _ptr(ws), st)
            kernel_timer.end(ev)
            dw = None if direct_w else unpack_weight_grad(dwp, weight)
        if has_bias and ctx.needs_input_grad[2] and not db_done:
            M = dy.shape[0]
            ws = x.new_empty((query("scan_colsum_ws_floats", M, cout),))
            db = ctx.bgrad_buf if direct_b else x.new_empty((cout,))
            call("scan_colsum", _ptr(dy), M, cout, cout_s, _ptr(db), int(direct_b), _ptr(ws), st)
            if direct_b:
                db = None
        return dx, dw, db, None, None, None, None, None, None, None, None, None


def conv2d(x, weight, bias, shape, ksize=3, stride=1, relu=False, cout_s=None, mask_dx=False, pool=False,
           gn_sums=False, out=None):
    """Returns rows [M_out, Cout_s]; the output PyramidShape is shape.conv_out(ksize, stride).
    relu: False | True | "deferred" (see _Conv2d); mask_dx: x is a deferred-ReLU output; pool: fuse the following
    2x2 / stride-2 max-pool (forward-only, bf16x3 mode: frozen VGG stages) -- the rows returned are the pooled ones;
    gn_sums: the output feeds groupnorm_relu -- let the conv epilogue accumulate the GroupNorm sums (bf16x3 mode,
    256 channels; otherwise ignored and groupnorm_relu computes its own statistics)."""
    assert relu in (False, True, "deferred")
    if pool and torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad
                                             or (bias is not None and bias.requires_grad)):
        raise RuntimeError("conv2d(pool=True) is forward-only (frozen 3x3 conv on a single-level pyramid)")
    return _Conv2d.apply(x, weight, bias, shape, ksize, stride, relu, cout_s, mask_dx, pool, gn_sums, out)


def conv_pool_fusable(x, weight, bias, shape):
    """True when conv2d(pool=True) applies: bf16x3 mode, a 3x3 conv on a single-level pyramid with even sizes that is
    not the 3-channel first layer, and nothing on it needs a gradient."""
    (h, w_) = shape.sizes[0]
    needs = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad
                                         or (bias is not None and bias.requires_grad))
    return (split_pieces() > 0 and shape.n_levels == 1 and h % 2 == 0 and w_ % 2 == 0 and x.shape[1] != 4
            and tuple(weight.shape[2:]) == (3, 3) and not needs)


# ----------------------------------------------------------------------------- pyramid row split
class _SplitLevels(torch.autograd.Function):
    """rows [M, C] -> one view per level.  The backward concatenates the level gradients once instead of
    autograd's per-slice zero-fill + add of full-size tensors."""

    @staticmethod
    def forward(ctx, rows, shape):
        ctx.shape = shape
        ctx.cols = rows.shape[1]
        return tuple(rows[shape.row_off[l]:shape.row_off[l + 1]] for l in range(shape.n_levels))

    @staticmethod
    def backward(ctx, *grads):
        shape = ctx.shape
        ref = next(g for g in grads if g is not None)
        parts = [g if g is not None else ref.new_zeros((shape.row_off[l + 1] - shape.row_off[l], ctx.cols))
                 for l, g in enumerate(grads)]
        return torch.cat(parts, 0), None


def split_levels(rows, shape):
    return _SplitLevels.apply(rows, shape)


class _SplitLevelsGRL(torch.autograd.Function):
    """rows [M, C] -> one view per level, each behind a gradient-reversal layer of its own strength (reference
    discriminator/layer.py:6-33: forward identity, backward -lambda_l * g).  The backward scales every level's gradient
    STRAIGHT INTO its row range of one [M, C] matrix: no per-level scaled copy followed by a concatenation."""

    @staticmethod
    def forward(ctx, rows, shape, lams):
        _chk(rows)
        ctx.cfg = (shape, tuple(float(v) for v in lams), rows.shape[1])
        return tuple(rows[shape.row_off[l]:shape.row_off[l + 1]] for l in range(shape.n_levels))

    @staticmethod
    def backward(ctx, *grads):
        shape, lams, cols = ctx.cfg
        ref = next(g for g in grads if g is not None)
        out = ref.new_empty((shape.rows, cols))
        st = _stream()
        for l, g in enumerate(grads):
            part = out[shape.row_off[l]:shape.row_off[l + 1]]
            if g is None:
                part.zero_()
            else:
                g = g.contiguous()
                call("scan_scale", _ptr(g), -lams[l], _ptr(part), g.numel(), st)
        return out, None, None


def split_levels_grl(rows, shape, lams):
    """split_levels with grad_reverse(., lams[l]) applied to level l (the discriminators' own GRL is then skipped)."""
    return _SplitLevelsGRL.apply(rows, shape, lams)


class _TakeImages(torch.autograd.Function):
    """rows of images [i0, i1) of every level, as the pyramid of those images (level-major order is kept)."""

    @staticmethod
    def forward(ctx, rows, shape, i0, i1):
        ctx.cfg = (shape, i0, i1, rows.shape)
        parts = []
        for l, (h, w) in enumerate(shape.sizes):
            r0 = shape.row_off[l]
            parts.append(rows[r0 + i0 * h * w:r0 + i1 * h * w])
        return torch.cat(parts, 0)

    @staticmethod
    def backward(ctx, g):
        shape, i0, i1, full = ctx.cfg
        if g.is_cuda and g.dtype == torch.float32 and len(full) == 2:
            out = g.new_empty(full)  # one pass: the taken images' rows copied, the others zeroed
            call("scan_take_images_backward", _ptr(g.contiguous()), shape.ref(), i0, i1, full[1], _ptr(out), _stream())
            return out, None, None, None
        out = g.new_zeros(full)
        o = 0
        for l, (h, w) in enumerate(shape.sizes):
            r0, n = shape.row_off[l], (i1 - i0) * h * w
            out[r0 + i0 * h * w:r0 + i0 * h * w + n] = g[o:o + n]
            o += n
        return out, None, None, None


def take_images(rows, shape, i0, i1):
    """sub-batch [i0, i1) of a pyramid -> (rows, PyramidShape of i1 - i0 images)."""
    return _TakeImages.apply(rows, shape, i0, i1), PyramidShape(i1 - i0, shape.sizes)


# ----------------------------------------------------------------------------- 2x2 max pooling
class _MaxPool2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, shape, relu_input=False):
        _chk(x)
        (h, w), n = shape.sizes[0], shape.n_images
        c = x.shape[1]
        y = x.new_empty((n * (h // 2) * (w // 2), c))
        call("scan_maxpool2x2_forward", _ptr(x), n, h, w, c, _ptr(y), _stream())
        ctx.save_for_backward(x, y)
        ctx.dims = (n, h, w, c, int(relu_input))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        n, h, w, c, relu_input = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        call("scan_maxpool2x2_backward", _ptr(x), _ptr(y), _ptr(dy), n, h, w, c, _ptr(dx), relu_input, _stream())
        return dx, None, None


def maxpool2x2(x, shape, relu_input=False):
    """2x2/2 max-pool of a single-level pyramid; returns (rows, PyramidShape).  relu_input: x is a deferred-ReLU
    conv output (ops.conv2d relu="deferred"): the backward also applies that ReLU's mask."""
    assert shape.n_levels == 1
    (h, w) = shape.sizes[0]
    return _MaxPool2x2.apply(x, shape, relu_input), PyramidShape(shape.n_images, [(h // 2, w // 2)])


def maxpool3x3s2(x, shape):
    """F.max_pool2d(x, 3, 2, 1) of a single-level pyramid (ResNet stem, reference backbone/resnet.py:335).
    Forward only: the stem is frozen for FREEZE_CONV_BODY_AT >= 1."""
    assert shape.n_levels == 1
    _chk(x)
    if x.requires_grad:
        raise RuntimeError("maxpool3x3s2 has no backward: the ResNet stem must be frozen (FREEZE_CONV_BODY_AT >= 1)")
    (h, w), n = shape.sizes[0], shape.n_images
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = x.new_empty((n * ho * wo, x.shape[1]))
    call("scan_maxpool3x3s2_forward", _ptr(x), n, h, w, x.shape[1], _ptr(y), _stream())
    return y, PyramidShape(n, [(ho, wo)])


class _Upsample2xAdd(torch.autograd.Function):
    """lateral + nearest-2x-upsampled coarse map (FPN top-down join) in one pass; backward: the gradient itself for the
    lateral, its 2x2 window sums for the coarse map."""

    @staticmethod
    def forward(ctx, lat, coarse, shape_c):
        _chk(lat, coarse)
        (h, w), n = shape_c.sizes[0], shape_c.n_images
        C = lat.shape[1]
        assert coarse.shape == (n * h * w, C) and lat.shape[0] == 4 * n * h * w
        y = torch.empty_like(lat)
        call("scan_upsample2x_add", _ptr(lat), _ptr(coarse), n, h, w, C, _ptr(y), _stream())
        ctx.dims = (n, h, w, C)
        return y

    @staticmethod
    def backward(ctx, g):
        n, h, w, C = ctx.dims
        g = g.contiguous()
        d = g.new_empty((n * h * w, C))
        call("scan_downsample2x_sum", _ptr(g), n, h, w, C, _ptr(d), _stream())
        return g, d, None


def upsample2x_add(lat, coarse, shape_coarse):
    """lat rows of a single-level pyramid twice the size of shape_coarse (exactly: FPN levels of /32-padded frames)."""
    return _Upsample2xAdd.apply(lat, coarse, shape_coarse)


class _AssembleRows(torch.autograd.Function):
    """The row blocks ``parts`` already ARE consecutive row ranges of ``buf`` (convs launched with out=buf[r0:r1]):
    return buf as one tensor; backward hands each producer its rows of the gradient as a view.  Replaces
    torch.cat(parts, 0) -- a read + write of the whole pyramid -- by nothing."""

    @staticmethod
    def forward(ctx, buf, *parts):
        off, offs = 0, []
        for p in parts:
            if p.data_ptr() != buf.data_ptr() + off * buf.shape[1] * buf.element_size() or p.shape[1] != buf.shape[1] \
                    or not p.is_contiguous():
                raise RuntimeError("assemble_rows: parts must be consecutive row ranges of buf")
            offs.append((off, off + p.shape[0]))
            off += p.shape[0]
        if off != buf.shape[0]:
            raise RuntimeError("assemble_rows: parts do not cover buf")
        ctx.offs = offs
        return torch.empty(0, dtype=buf.dtype, device=buf.device).set_(buf.untyped_storage(), buf.storage_offset(),
                                                                      buf.shape, buf.stride())

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(g[a:b] for a, b in ctx.offs)


def assemble_rows(buf, parts):
    return _AssembleRows.apply(buf, *parts)


class _AddReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _chk(a, b)
        assert a.shape == b.shape
        y = torch.empty_like(a)
        call("scan_add_relu", _ptr(a), _ptr(b), _ptr(y), a.numel(), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        g = torch.empty_like(dy)
        call("scan_relu_backward", _ptr(dy), _ptr(y), _ptr(g), dy.numel(), _stream())
        return g, g


def add_relu(a, b):
    """max(a + b, 0): the residual join of a bottleneck block."""
    return _AddReLU.apply(a, b)


# ----------------------------------------------------------------------------- GroupNorm + ReLU
def _col_slice_ld(t, C):
    """row stride (floats) of t when it is a [M, C] column slice of a wider row-major fp32 matrix that the kernels can
    address in place (unit column stride, 16-byte aligned rows), else None"""
    if t.dim() == 2 and t.shape[1] == C and t.stride(1) == 1 and t.stride(0) >= C and t.stride(0) % 4 == 0 \
            and t.data_ptr() % 16 == 0:
        return t.stride(0)
    return None


class _GroupNormReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, shape, relu, eps, out_buf=None):
        _chk(x, gamma, beta)
        C = x.shape[1]
        st = _stream()
        stats = x.new_empty((shape.n_levels * shape.n_images * 32 * 2,))
        sums = _gn_sums.pop(x.data_ptr(), None)
        if out_buf is None:
            y, ldy = torch.empty_like(x), C
        else:  # normalise straight into the first C columns of a wider matrix (cat_into completes it)
            if not (out_buf.is_cuda and out_buf.dtype == torch.float32 and out_buf.is_contiguous()
                    and out_buf.shape[0] == x.shape[0] and out_buf.shape[1] >= C and out_buf.shape[1] % 4 == 0):
                raise RuntimeError("groupnorm_relu: out_buf must be a contiguous fp32 [M, >= C] GPU matrix with a row "
                                   "length that is a multiple of 4")
            y, ldy = out_buf[:, :C], out_buf.shape[1]
        if sums is not None and not BATCHED:
            call("scan_groupnorm_stats_from_sums", _ptr(sums), shape.ref(), C, 32, eps, _ptr(stats), st)
            call("scan_groupnorm_relu_forward_ld", _ptr(x), shape.ref(), C, 32, _ptr(stats), _ptr(gamma), _ptr(beta),
                 int(relu), _ptr(y), ldy, st)
        elif sums is not None:  # accumulated by the epilogue of the conv that produced x: one launch normalises and
            # leaves (mean, rstd) behind for the backward
            call("scan_groupnorm_relu_forward_from_sums_ld", _ptr(x), shape.ref(), C, 32, _ptr(sums), eps, _ptr(gamma),
                 _ptr(beta), int(relu), _ptr(y), ldy, _ptr(stats), st)
        else:
            nws = query("scan_groupnorm_ws_floats", shape.ref(), C, 32)
            ws = torch.empty((nws // 2 + 1,), dtype=torch.float64, device=x.device)
            call("scan_groupnorm_stats", _ptr(x), shape.ref(), C, 32, eps, _ptr(stats), _ptr(ws), st)
            call("scan_groupnorm_relu_forward_ld", _ptr(x), shape.ref(), C, 32, _ptr(stats), _ptr(gamma), _ptr(beta),
                 int(relu), _ptr(y), ldy, st)
        ctx.save_for_backward(x, beta, gamma, stats)  # the backward recomputes the ReLU mask from x: y is not kept
        ctx.cfg = (shape, relu)
        ctx.gbuf = ctx.bbuf = None
        if getattr(gamma, "_scan_flat", False) and getattr(beta, "_scan_flat", False) and gamma.grad is not None \
                and beta.grad is not None:
            ctx.gbuf, ctx.bbuf = gamma.grad, beta.grad
        return y

    @staticmethod
    def backward(ctx, dy):
        x, beta, gamma, stats = ctx.saved_tensors
        shape, relu = ctx.cfg
        C = x.shape[1]
        lddy = _col_slice_ld(dy, C)  # e.g. the first C columns of the class-branch conv's data gradient: read in place
        if lddy is None:
            dy, lddy = dy.contiguous(), C
        nws = query("scan_groupnorm_ws_floats", shape.ref(), C, 32)
        ws, cleared = _ws_f64(nws // 2 + 1, x.device)
        dx = torch.empty_like(x)
        direct = ctx.gbuf is not None
        dg = ctx.gbuf if direct else x.new_empty((C,))
        db = ctx.bbuf if direct else x.new_empty((C,))
        call("scan_groupnorm_relu_backward_ld", _ptr(x), _ptr(beta), _ptr(dy), lddy, shape.ref(), C, 32, _ptr(stats),
             _ptr(gamma), int(relu), _ptr(dx), _ptr(dg), _ptr(db), int(direct) | (2 if cleared else 0), _ptr(ws), _stream())
        if direct:
            return dx, None, None, None, None, None, None
        return dx, dg, db, None, None, None, None


def groupnorm_relu(x, gamma, beta, shape, relu=True, eps=1e-5, out_buf=None):
    """out_buf: a contiguous [M, C + e] matrix -- the result is written into (and returned as) its first C columns;
    cat_into() then adds the remaining columns without copying these."""
    return _GroupNormReLU.apply(x, gamma, beta, shape, relu, eps, out_buf)


class _CatInto(torch.autograd.Function):
    """cat([y, extra, zeros], 1) where y already IS the first C columns of ``buf`` (groupnorm_relu(out_buf=buf)): only
    the few extra columns are copied.  Backward: the two column slices of the gradient, as views -- the GroupNorm
    backward reads its slice in place (scan_groupnorm_relu_backward_ld)."""

    @staticmethod
    def forward(ctx, y, extra, buf):
        C, e = y.shape[1], extra.shape[1]
        if y.data_ptr() != buf.data_ptr() or y.stride(0) != buf.shape[1] or C + e > buf.shape[1]:
            raise RuntimeError("cat_into: y must be the leading columns of buf")
        if buf.is_cuda and extra.dtype == torch.float32 and extra.stride(1) == 1:
            # one coalesced pass (scan_copy_cols): the e columns + the zero tail (torch: a strided slice copy + a fill)
            call("scan_copy_cols", _ptr(extra), extra.stride(0), ctypes.c_void_p(buf.data_ptr() + 4 * C), buf.shape[1],
                 buf.shape[0], e, buf.shape[1] - C - e, _stream())
        else:
            buf[:, C:C + e] = extra
            if C + e < buf.shape[1]:
                buf[:, C + e:] = 0
        ctx.cols = (C, e)
        # a fresh tensor object over buf's storage: to autograd the result is not a view of an input
        return torch.empty(0, dtype=buf.dtype, device=buf.device).set_(buf.untyped_storage(), buf.storage_offset(),
                                                                      buf.shape, buf.stride())

    @staticmethod
    def backward(ctx, g):
        C, e = ctx.cols
        return g[:, :C], g[:, C:C + e], None


def cat_into(y, extra, buf):
    return _CatInto.apply(y, extra, buf)


class _PadCols(torch.autograd.Function):
    """[M, K] -> [M, cs] with zero columns behind (F.pad(x, (0, cs - K))) in one pass; backward: the first K columns."""

    @staticmethod
    def forward(ctx, x, cs):
        _chk(x)
        M, K = x.shape
        ctx.k = K
        y = x.new_empty((M, cs))
        call("scan_copy_cols", _ptr(x), K, _ptr(y), cs, M, K, cs - K, _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        d = g.new_empty((g.shape[0], ctx.k))
        call("scan_copy_cols", _ptr(g), g.shape[1], _ptr(d), ctx.k, g.shape[0], ctx.k, 0, _stream())
        return d, None


def pad_cols(x, cs):
    return x if x.shape[1] == cs else _PadCols.apply(x, int(cs))


# ----------------------------------------------------------------------------- dynamic conv + softmax
class _DynConvSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, kernels):
        _chk(feat, kernels)
        M, C = feat.shape
        K = kernels.shape[0]
        kernels = kernels.contiguous()
        logits = feat.new_empty((M, K))
        probs = feat.new_empty((M, K))
        call("scan_dynconv_softmax_forward", _ptr(feat), _ptr(kernels), M, C, K, _ptr(logits), _ptr(probs), _stream())
        ctx.save_for_backward(feat, kernels, probs)
        return logits, probs

    @staticmethod
    def backward(ctx, d_logits, d_probs):
        feat, kernels, probs = ctx.saved_tensors
        M, C = feat.shape
        K = kernels.shape[0]
        d_logits = d_logits.contiguous() if d_logits is not None else None
        d_probs = d_probs.contiguous() if d_probs is not None else None
        d_feat = torch.empty_like(feat)
        d_k = torch.empty_like(kernels)
        ws = feat.new_empty((query("scan_dynconv_ws_floats", M, C, K),))
        call("scan_dynconv_softmax_backward", _ptr(feat), _ptr(kernels), _ptr(probs), _ptr(d_logits), _ptr(d_probs), M,
             C, K, _ptr(d_feat), _ptr(d_k), _ptr(ws), _stream())
        return d_feat, d_k


def dynconv_softmax(feat, kernels):
    """feat [M,256], kernels [K,256] -> (logits [M,K], probs [M,K])."""
    return _DynConvSoftmax.apply(feat, kernels)


# ----------------------------------------------------------------------------- losses
class _SigmoidFocalSum(torch.autograd.Function):
    """sum of the element-wise sigmoid focal loss (what SigmoidFocalLoss.forward returns)."""

    @staticmethod
    def forward(ctx, logits, targets, gamma, alpha):
        _chk(logits, targets)
        if logits.dim() != 2:
            raise RuntimeError("logits must be [M, C]")
        if targets.dtype != torch.int32:
            raise RuntimeError("targets must be int32")
        M, C = logits.shape
        out = logits.new_zeros((1,))
        call("scan_sigmoid_focal_loss_forward", _ptr(logits), _ptr(targets), M, C, gamma, alpha, None, _ptr(out),
             _stream())
        ctx.save_for_backward(logits, targets)
        ctx.cfg = (gamma, alpha)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        logits, targets = ctx.saved_tensors
        gamma, alpha = ctx.cfg
        M, C = logits.shape
        d = torch.empty_like(logits)
        dl = g.expand(M, C).contiguous()
        call("scan_sigmoid_focal_loss_backward", _ptr(logits), _ptr(targets), _ptr(dl), 1.0, M, C, gamma, alpha, _ptr(d),
             _stream())
        return d, None, None, None


def sigmoid_focal_loss_sum(logits, targets, gamma, alpha):
    return _SigmoidFocalSum.apply(logits.contiguous(), targets, float(gamma), float(alpha))


class _SigmoidFocalElem(torch.autograd.Function):
    """element-wise losses [M,C]; the _C.sigmoid_focalloss_forward/backward pair."""

    @staticmethod
    def forward(ctx, logits, targets, gamma, alpha):
        _chk(logits, targets)
        M, C = logits.shape
        losses = torch.empty_like(logits)
        call("scan_sigmoid_focal_loss_forward", _ptr(logits), _ptr(targets), M, C, gamma, alpha, _ptr(losses), None,
             _stream())
        ctx.save_for_backward(logits, targets)
        ctx.cfg = (gamma, alpha)
        return losses

    @staticmethod
    def backward(ctx, d_loss):
        logits, targets = ctx.saved_tensors
        gamma, alpha = ctx.cfg
        M, C = logits.shape
        d_loss = d_loss.contiguous()
        d = torch.empty_like(logits)
        call("scan_sigmoid_focal_loss_backward", _ptr(logits), _ptr(targets), _ptr(d_loss), 1.0, M, C, gamma, alpha,
             _ptr(d), _stream())
        return d, None, None, None


class _IouLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight):
        _chk(pred, target, weight)
        P = pred.shape[0]
        out = pred.new_zeros((2,))
        call("scan_iou_loss_forward", _ptr(pred), _ptr(target), _ptr(weight), P, _ptr(out), _stream())
        ctx.save_for_backward(pred, target, weight, out)
        return out[0] / out[1]

    @staticmethod
    def backward(ctx, g):
        pred, target, weight, out = ctx.saved_tensors
        gn = (g / out[1]).reshape(1).contiguous()
        d = torch.empty_like(pred)
        call("scan_iou_loss_backward", _ptr(pred), _ptr(target), _ptr(weight), pred.shape[0], _ptr(gn), _ptr(d),
             _stream())
        return d, None, None


def iou_loss(pred, target, weight=None):
    """weighted mean when weight is given (and sums > 0), plain mean otherwise."""
    return _IouLoss.apply(pred.contiguous(), target.contiguous(), weight.contiguous() if weight is not None else None)


class _BceLogitsMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets):
        _chk(logits, targets)
        M = logits.numel()
        out = logits.new_zeros((2,))
        call("scan_bce_logits_forward", _ptr(logits), _ptr(targets), 0.0, None, 0, M, _ptr(out), _stream())
        ctx.save_for_backward(logits, targets)
        return out[0] / M

    @staticmethod
    def backward(ctx, g):
        logits, targets = ctx.saved_tensors
        M = logits.numel()
        gn = (g / M).reshape(1).contiguous()
        d = torch.empty_like(logits)
        call("scan_bce_logits_backward", _ptr(logits), _ptr(targets), 0.0, None, 0, M, _ptr(gn), _ptr(d), _stream())
        return d, None


def bce_with_logits_mean(logits, targets):
    return _BceLogitsMean.apply(logits.contiguous(), targets.contiguous())


class _CkaBce(torch.autograd.Function):
    """sum_c [ sum_m act[m,c+1] bce(logit[m,c], t) / sum_m act[m,c+1] ] / Cf -- one launch forward (the last block to
    finish forms the scalar), one launch backward (the kernel forms the per-class coefficients from the gradient of the
    scalar and the forward's sums)."""

    @staticmethod
    def forward(ctx, logits, act, target, cf):
        _chk(logits, act)
        M = logits.shape[0]
        assert logits.shape[1] == cf and act.shape[1] == cf + 1
        out = _zeros_f32(2 * cf + 2, logits.device)
        if M > 0:
            call("scan_cka_bce_forward_loss", _ptr(logits), _ptr(act), M, cf, target, _ptr(out), _stream())
        ctx.save_for_backward(logits, act, out)
        ctx.cfg = (target, cf)
        return out[2 * cf + 1] if M > 0 else out[2 * cf + 1] / 0.0  # no rows: the reference's 0 / 0

    @staticmethod
    def backward(ctx, g):
        logits, act, out = ctx.saved_tensors
        target, cf = ctx.cfg
        d = torch.empty_like(logits)
        if logits.shape[0] > 0:
            call("scan_cka_bce_backward_loss", _ptr(logits), _ptr(act), logits.shape[0], cf, target,
                 _ptr(g.reshape(1).contiguous()), _ptr(out), _ptr(d), _stream())
        return d, None, None, None


class _CkaBcePair(torch.autograd.Function):
    """the two domain losses of one level in the paired step -- rows [0, m) against label 1 (source), rows [m, M) against label 0
    (target), each with its own act-map normaliser (_CkaBce on the two halves) -- with ONE gradient buffer: the two backward
    launches write their halves of d_logits directly (split_rows2 + two _CkaBce nodes: two buffers + two copies per level)."""

    @staticmethod
    def forward(ctx, logits, act, m, cf):
        _chk(logits, act)
        M = logits.shape[0]
        assert logits.shape[1] == cf and act.shape[1] == cf + 1 and 0 < m < M
        outs = []
        for lo, hi, target in ((0, m, 1.0), (m, M, 0.0)):
            out = _zeros_f32(2 * cf + 2, logits.device)
            call("scan_cka_bce_forward_loss", _ptr(logits[lo:hi]), _ptr(act[lo:hi]), hi - lo, cf, target, _ptr(out), _stream())
            outs.append(out)
        ctx.save_for_backward(logits, act, outs[0], outs[1])
        ctx.cfg = (m, cf)
        return outs[0][2 * cf + 1], outs[1][2 * cf + 1]

    @staticmethod
    def backward(ctx, gs, gt):
        logits, act, out_s, out_t = ctx.saved_tensors
        m, cf = ctx.cfg
        M = logits.shape[0]
        d = torch.empty_like(logits)
        for lo, hi, target, g, out in ((0, m, 1.0, gs, out_s), (m, M, 0.0, gt, out_t)):
            if g is None:
                d[lo:hi].zero_()
            else:
                call("scan_cka_bce_backward_loss", _ptr(logits[lo:hi]), _ptr(act[lo:hi]), hi - lo, cf, target,
                     _ptr(g.reshape(1).contiguous()), _ptr(out), _ptr(d[lo:hi]), _stream())
        return d, None, None, None


def cka_bce_pair(logits, act_detached, m, cf):
    return _CkaBcePair.apply(logits.contiguous(), act_detached.contiguous(), int(m), int(cf))


class _SplitRows2(torch.autograd.Function):
    """(x[:m], x[m:]) as views; the backward writes the two gradients into ONE buffer (torch's two slice backwards each
    zero-fill a full-size tensor, copy their part and are then added: 2 fills + 2 copies + 1 add per use)."""

    @staticmethod
    def forward(ctx, x, m):
        ctx.m = m
        ctx.shape = x.shape
        return x[:m], x[m:]

    @staticmethod
    def backward(ctx, g1, g2):
        m = ctx.m
        g = (g1 if g1 is not None else g2).new_empty(ctx.shape)
        if g1 is not None:
            g[:m] = g1
        else:
            g[:m].zero_()
        if g2 is not None:
            g[m:] = g2
        else:
            g[m:].zero_()
        return g, None


def split_rows2(x, m):
    return _SplitRows2.apply(x, m)


def cka_bce(logits, act_detached, target, cf):
    return _CkaBce.apply(logits.contiguous(), act_detached.contiguous(), float(target), int(cf))


class _SoftmaxFocalMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, gamma):
        _chk(logits, labels)
        if labels.dtype != torch.int64:
            raise RuntimeError("labels must be int64")
        M, K = logits.shape
        out = logits.new_zeros((1,))
        call("scan_softmax_focal_forward", _ptr(logits), _ptr(labels), M, K, gamma, _ptr(out), _stream())
        ctx.save_for_backward(logits, labels)
        ctx.gamma = gamma
        return out[0] / M

    @staticmethod
    def backward(ctx, g):
        logits, labels = ctx.saved_tensors
        M, K = logits.shape
        d = torch.empty_like(logits)
        # d_scale must be a host float: one small sync-free path would pass it by pointer; g is a 0-dim tensor
        call("scan_softmax_focal_backward", _ptr(logits), _ptr(labels), M, K, ctx.gamma, 1.0 / M, _ptr(d), _stream())
        return d * g, None, None


def softmax_focal_loss_mean(logits, labels, gamma=2.0):
    return _SoftmaxFocalMean.apply(logits.contiguous(), labels.contiguous(), float(gamma))


class _GradReverse(torch.autograd.Function):
    """reference discriminator/layer.py:6-24: forward x.clone(), backward -lambda * g.  The forward here is a view of
    x -- same values, and nothing downstream writes into it in place -- which saves a read + write of every feature
    and act map handed to the five discriminators (375 MB per iteration at the bench size).  Any dense layout is taken as
    it is (row matrices; channels_last NCHW tensors on the drop-in surface): the backward is element-wise on the storage."""

    @staticmethod
    def forward(ctx, x, lam):
        if not x.is_cuda:
            raise RuntimeError("scan_amd ops run only on the GPU (HIP) -- got a %s tensor; no CPU fallback" % x.device)
        ctx.lam = lam
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if not (g.is_contiguous() or (g.dim() == 4 and g.is_contiguous(memory_format=torch.channels_last))):
            g = g.contiguous()
        d = torch.empty_like(g)  # preserve_format: the same dense strides as g
        call("scan_scale", _ptr(g), -ctx.lam, _ptr(d), g.numel(), _stream())
        return d, None


def grad_reverse(x, lam):
    return _GradReverse.apply(x, float(lam))


# ----------------------------------------------------------------------------- NMS
def _nms_impl(dets, scores, labels, thr, rule_ge):
    n = dets.shape[0]
    if n == 0:
        # reference csrc/nms.h:17-18: empty input -> empty CPU int64 tensor
        return torch.empty((0,), dtype=torch.int64, device="cpu")
    _chk(dets, scores, labels)
    if n > _lib.NMS_MAX:
        raise RuntimeError("nms: n=%d exceeds SCAN_NMS_MAX=%d" % (n, _lib.NMS_MAX))
    dets = dets.contiguous().float()
    scores = scores.contiguous().float()
    if labels is not None:
        labels = labels.contiguous().float()
    keep, cnt = _nms_launch(dets, scores, labels, thr, rule_ge)
    return keep[:int(cnt.item())]


def _nms_launch(dets, scores, labels, thr, rule_ge):
    """the launches of one NMS without the host read of the count: (keep buffer [n], count [1] int32)"""
    n = dets.shape[0]
    ws = torch.empty((query("scan_nms_ws_bytes", n) + 15) // 16 * 2, dtype=torch.float64, device=dets.device)
    keep = torch.empty((n,), dtype=torch.int64, device=dets.device)
    cnt = torch.empty((1,), dtype=torch.int32, device=dets.device)
    call("scan_nms", _ptr(dets), _ptr(scores), _ptr(labels), n, float(thr), int(rule_ge), _ptr(keep), _ptr(cnt),
         _ptr(ws), _stream())
    return keep, cnt


def nms_by_label_async(dets, scores, labels, thr):
    """ml_nms / per-class NMS launched on the current stream, nothing read back: returns finish() -> kept indices.  Lets a
    caller queue the NMS of several images (each a single-workgroup chain of kernels) on different streams before the
    first host read."""
    n = dets.shape[0]
    if n == 0:
        return lambda: torch.empty((0,), dtype=torch.int64, device="cpu")
    _chk(dets, scores, labels)
    if n > _lib.NMS_MAX:
        raise RuntimeError("nms: n=%d exceeds SCAN_NMS_MAX=%d" % (n, _lib.NMS_MAX))
    keep, cnt = _nms_launch(dets.contiguous().float(), scores.contiguous().float(), labels.contiguous().float(), thr, True)
    return lambda: keep[:int(cnt.item())]


def nms(dets, scores, thr, rule_ge=True):
    """kept original indices, ascending (reference csrc/cpu/nms_cpu.cpp:64, cuda/nms.cu:127-130)."""
    return _nms_impl(dets, scores, None, thr, rule_ge)


def nms_by_label(dets, scores, labels, thr):
    """greedy NMS per label in one launch, CPU '>=' rule (= boxlist_nms applied class by class,
    reference rpn/fcos/inference.py:160-176 + csrc/cpu/nms_cpu.cpp:60); kept original indices ascending."""
    return _nms_impl(dets, scores, labels.float(), thr, True)


def ml_nms(dets, scores, labels, thr):
    """label-aware NMS with the CUDA '>' rule (reference csrc/cuda/ml_nms.cu:13-24,62)."""
    return _nms_impl(dets, scores, labels, thr, False)


# ----------------------------------------------------------------------------- density clustering (target nodes)
def dbscan_in_cluster0(pts, eps, min_samples=5):
    """bool [n]: True where sklearn.cluster.DBSCAN(eps, min_samples).fit_predict(pts) would return label 0 -- all the
    target-node sampling needs (reference rpn/fcos/loss.py:397-423: noise -> 1, cluster 0 -> 0, selected = non-zero).
    Neighbour search, core test, breadth-first growth of cluster 0 and border assignment run on the device
    (scan_dbscan_*); the host only reads one flag per breadth-first level."""
    _chk(pts)
    n, d = pts.shape
    if n == 0:
        return torch.zeros((0,), dtype=torch.bool, device=pts.device)
    nbytes = query("scan_dbscan_ws_bytes", n)
    if nbytes < 0:
        raise RuntimeError("dbscan: n=%d exceeds SCAN_DBSCAN_MAX" % n)
    ws = torch.empty((nbytes // 8 + 1,), dtype=torch.float64, device=pts.device)
    info = torch.empty((2,), dtype=torch.int32, device=pts.device)
    st = _stream()
    call("scan_dbscan_prepare", _ptr(pts), n, d, float(eps), int(min_samples), _ptr(ws), _ptr(info), st)
    if int(info[0].item()) >= n:  # no core point: everything is noise
        return torch.zeros((n,), dtype=torch.bool, device=pts.device)
    changed = torch.zeros((1,), dtype=torch.int32, device=pts.device)
    parity = 0
    while True:
        call("scan_dbscan_bfs_step", n, _ptr(ws), parity, _ptr(changed), st)
        if int(changed.item()) == 0:
            break
        parity ^= 1
    out = torch.empty((n,), dtype=torch.uint8, device=pts.device)
    call("scan_dbscan_finish", n, _ptr(ws), _ptr(out), st)
    return out.bool()


# ----------------------------------------------------------------------------- CKA discriminator class branches
class _CkaStackedWeights(torch.autograd.Function):
    """The Cf per-class classifier branches (conv3x3 C+1 -> H, conv3x3 H -> 1; reference
    discriminator/fcos_head_discriminator_con.py:44-62) as the weights of two stacked convolutions -- one launch
    forward (scan_cka_stack_weights), one launch backward (scan_cka_unstack_grads, straight into the parameters' flat
    gradient buffers when they have them) instead of the ~40 stack / slice / mul / cat kernels and their autograd
    mirror images per discriminator."""

    @staticmethod
    def forward(ctx, C, H, cs1, *params):
        cf = len(params) // 4
        w0s, b0s, w2s, b2s = params[0::4], params[1::4], params[2::4], params[3::4]
        for t in params:  # strided (channels-last) weights are fine here: the kernel takes element strides
            if not t.is_cuda or t.dtype != torch.float32:
                raise RuntimeError("scan_amd ops run only on fp32 GPU (HIP) tensors -- got %s %s; no CPU fallback"
                                   % (t.device, t.dtype))
        w0, w2 = w0s[0], w2s[0]
        assert tuple(w0.shape) == (H, C + 1, 3, 3) and tuple(w2.shape) == (1, H, 3, 3), (w0.shape, w2.shape)
        for a, b in zip(w0s, w2s):
            if a.stride() != w0.stride() or b.stride() != w2.stride() or a.shape != w0.shape or b.shape != w2.shape:
                raise RuntimeError("cka_stacked_weights: the class branches must share one memory layout")
        if w0.stride(2) != 3 * w0.stride(3) or w2.stride(2) != 3 * w2.stride(3):
            raise RuntimeError("cka_stacked_weights: the 3x3 taps of a weight must be evenly strided")
        strides = (w0.stride(0), w0.stride(1), w0.stride(3), w2.stride(1), w2.stride(3))
        cs2 = cf * H
        dev = w0.device
        w1 = torch.empty((cf * H, 9, cs1), device=dev)
        b1 = torch.empty((cf * H,), device=dev)
        w2o = torch.empty((cf, 9, cs2), device=dev)
        b2 = torch.empty((cf,), device=dev)
        arr = (_lib.CkaBranch * cf)()
        for a, p0, q0, p2, q2 in zip(arr, w0s, b0s, w2s, b2s):
            a.w0, a.b0, a.w2, a.b2 = p0.data_ptr(), q0.data_ptr(), p2.data_ptr(), q2.data_ptr()
        call("scan_cka_stack_weights", arr, cf, C, H, *strides, cs1, cs2, _ptr(w1), _ptr(b1), _ptr(w2o), _ptr(b2),
             _stream())
        ctx.cfg = (cf, C, H, cs1, cs2, strides)
        ctx.params = params  # parameters (leaf tensors), needed for their .grad buffers; not graph intermediates
        # logical conv weights [O, Cin, 3, 3] stored channels-last: conv2d reads them in place
        return (w1.view(cf * H, 3, 3, cs1).permute(0, 3, 1, 2), b1, w2o.view(cf, 3, 3, cs2).permute(0, 3, 1, 2), b2)

    @staticmethod
    def backward(ctx, dw1, db1, dw2, db2):
        cf, C, H, cs1, cs2, strides = ctx.cfg
        params = ctx.params

        def ohwi(g):
            return None if g is None else g.permute(0, 2, 3, 1).contiguous()

        dw1, dw2 = ohwi(dw1), ohwi(dw2)
        db1 = None if db1 is None else db1.contiguous()
        db2 = None if db2 is None else db2.contiguous()
        direct = all(getattr(p, "_scan_flat", False) and p.grad is not None and p.grad.stride() == p.stride()
                     for p in params)
        tgt = [p.grad for p in params] if direct else [torch.zeros_like(p) for p in params]
        arr = (_lib.CkaBranch * cf)()
        for c, a in enumerate(arr):
            a.w0, a.b0, a.w2, a.b2 = (t.data_ptr() for t in tgt[4 * c:4 * c + 4])
        call("scan_cka_unstack_grads", arr, cf, C, H, *strides, cs1, cs2, _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2),
             1, _stream())
        return (None, None, None) + (tuple(None for _ in params) if direct else tuple(tgt))


def cka_stacked_weights(branches, C, H, cs1):
    """branches: the Cf (conv0, conv2) module pairs.  Returns (w1 [Cf*H, cs1, 3, 3], b1, w2 [Cf, Cf*H, 3, 3], b2) with
    cs1 = the (padded) channel count of cat(x, act[1:])."""
    params = []
    for c0, c2 in branches:
        params += [c0.weight, c0.bias, c2.weight, c2.bias]
    return _CkaStackedWeights.apply(int(C), int(H), int(cs1), *params)


class _GroupedConvTo1(torch.autograd.Function):
    """conv3x3 [M, G*128] -> [M, Ns >= G], one output channel per group of 128 (the class branches' second conv);
    w = the stacked weight [G, G*128, 3, 3] (channels-last) whose diagonal blocks are the per-class weights.
    mask_dx: x is a deferred-ReLU output (see _Conv2d) -- the forward then also leaves (x > 0) as a bit mask (1/32 of x)
    so that the backward does not read x a second time for it."""

    @staticmethod
    def forward(ctx, x, w, bias, shape, G, mask_dx):
        _chk(x, bias)
        gc = G * 128
        assert x.shape == (shape.rows, gc) and tuple(w.shape) == (G, gc, 3, 3) and w.permute(0, 2, 3, 1).is_contiguous()
        ns = pad4(G)
        y = x.new_empty((shape.rows, ns))
        ws = x.new_empty((query("scan_gconv3x3_to1_ws_floats", shape.ref(), G, 128),))
        bits = None
        mfma = query("scan_tune_get", b"gconv_mfma")  # read-only: no write to the launch-selection state
        if mfma == 1 and mask_dx and torch.is_grad_enabled() and x.requires_grad:
            bits = torch.empty((shape.rows * G * 4,), dtype=torch.int32, device=x.device)
            call("scan_gconv3x3_to1_forward_bits", _ptr(x), shape.ref(), G, 128, _ptr(w), _ptr(bias), _ptr(y), ns,
                 _ptr(ws), _ptr(bits), _stream())
        else:
            call("scan_gconv3x3_to1_forward", _ptr(x), shape.ref(), G, 128, _ptr(w), _ptr(bias), _ptr(y), ns, _ptr(ws),
                 _stream())
        ctx.save_for_backward(x, w, bits)
        ctx.cfg = (shape, G, ns, mask_dx, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, bits = ctx.saved_tensors
        shape, G, ns, mask_dx, has_bias = ctx.cfg
        dy = dy.contiguous()
        st = _stream()
        dx = dw = db = dwp = None
        if ctx.needs_input_grad[1]:
            dwp = x.new_zeros((G, 9, G * 128))  # off-diagonal blocks: exact zeros (they meet a zero in the adjoint)
            ws = x.new_empty((query("scan_gconv3x3_to1_ws_floats", shape.ref(), G, 128),))
            dw = dwp.view(G, 3, 3, G * 128).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
        if dx is not None and dwp is not None and bits is not None:
            call("scan_gconv3x3_to1_backward_bits", _ptr(x), _ptr(dy), ns, shape.ref(), G, 128, _ptr(w), _ptr(bits),
                 _ptr(dx), _ptr(dwp), 0, _ptr(ws), st)
        elif dx is not None and dwp is not None:
            call("scan_gconv3x3_to1_backward", _ptr(x), _ptr(dy), ns, shape.ref(), G, 128, _ptr(w), int(mask_dx), _ptr(dx),
                 _ptr(dwp), 0, _ptr(ws), st)
        elif dx is not None:
            call("scan_gconv3x3_to1_dgrad", _ptr(dy), ns, shape.ref(), G, 128, _ptr(w), _ptr(x if mask_dx else None),
                 _ptr(dx), st)
        elif dwp is not None:
            call("scan_gconv3x3_to1_wgrad", _ptr(x), _ptr(dy), ns, shape.ref(), G, 128, _ptr(dwp), 0, _ptr(ws), st)
        if has_bias and ctx.needs_input_grad[2]:
            M = dy.shape[0]
            cws = x.new_empty((query("scan_colsum_ws_floats", M, G),))
            db = x.new_empty((G,))
            call("scan_colsum", _ptr(dy), M, G, ns, _ptr(db), 0, _ptr(cws), st)
        return dx, dw, db, None, None, None


# SCAN_GROUPED_CLS=0: the class branches' second conv on the dense MFMA kernel (block-diagonal weight), for A/B
GROUPED_CLS = os.environ.get("SCAN_GROUPED_CLS", "1") != "0"


def gconv3x3_to1(x, w, bias, shape, G, mask_dx=False):
    """x [M, G*128] -> [M, pad4(G)]: y[:, g] = conv3x3(x[:, g*128:(g+1)*128], w[g, g*128:(g+1)*128]) + bias[g]."""
    return _GroupedConvTo1.apply(x, w, bias, shape, int(G), bool(mask_dx))


# ----------------------------------------------------------------------------- optimizer
# SCAN_COND_RNN=0: the conditioned-kernel generator (paradigm -> 2-layer tanh RNN -> (T, 1) conv) runs as the torch loop it is
# written as in modeling/condgraph.py (T x 2 x (2 linear + add + tanh), ~120 launches with its backward) instead of the
# 15 launches of csrc/condrnn.hip (one per link of the chain, spread over 8-32 workgroups), for A/B.  Same values up to the summation order.
COND_RNN_FUSED = os.environ.get("SCAN_COND_RNN", "1") != "0"


class _CondRnn(torch.autograd.Function):
    """proto [K, 256, T] (a buffer: no gradient), the ten parameters in scan_cond_rnn_forward's order -> kernels [K, 256]"""

    @staticmethod
    def forward(ctx, proto, *params):
        if not (proto.is_cuda and all(p.is_cuda and p.dtype == torch.float32 for p in params)):
            raise RuntimeError("scan_amd ops run only on the GPU (HIP); no CPU fallback")
        K, C, T = proto.shape
        proto = proto.permute(2, 0, 1).contiguous()  # [T, K, 256]: the RNN's sequence-first input
        ws_ = [p.detach().contiguous() for p in params]
        h0 = proto.new_empty((T, K, 512))
        h1 = torch.empty_like(h0)
        ker = proto.new_empty((K, 256))
        wptr = (ctypes.c_void_p * 10)(*[w.data_ptr() for w in ws_])
        call("scan_cond_rnn_forward", _ptr(proto), K, T, wptr, _ptr(h0), _ptr(h1), _ptr(ker), _stream())
        ctx.save_for_backward(proto, h0, h1, *ws_)
        return ker

    @staticmethod
    def backward(ctx, dker):
        proto, h0, h1, *ws_ = ctx.saved_tensors
        T, K, C = proto.shape
        dker = dker.contiguous()
        grads = [torch.empty_like(w) for w in ws_]
        wptr = (ctypes.c_void_p * 10)(*[w.data_ptr() for w in ws_])
        gptr = (ctypes.c_void_p * 10)(*[g.data_ptr() for g in grads])
        ws = proto.new_empty((query("scan_cond_rnn_ws_floats"),))
        call("scan_cond_rnn_backward", _ptr(proto), K, T, wptr, _ptr(h0), _ptr(h1), _ptr(dker), gptr, _ptr(ws), _stream())
        return (None, *grads)


def cond_rnn_supported(proto, params):
    K, C, T = proto.shape
    return (COND_RNN_FUSED and proto.is_cuda and K <= 9 and T <= 3 and C == 256 and tuple(params[0].shape) == (512, 256)
            and tuple(params[8].shape[:3]) == (256, 512, T))


def cond_rnn(proto, params):
    """reference condgraph.py:313-319 get_conded_weight as one launch (two in the backward); params: weight_ih_l0,
    weight_hh_l0, bias_ih_l0, bias_hh_l0, weight_ih_l1, weight_hh_l1, bias_ih_l1, bias_hh_l1, cond_nx1.weight, cond_nx1.bias"""
    return _CondRnn.apply(proto, *params)


def sgd_momentum_multi_(segments, momentum):
    """segments: list of (p, g, buf, lr, wd, first_step) flat fp32 tensors of equal length -- all updated by ONE launch
    (scan_sgd_momentum_multi), element arithmetic as sgd_momentum_."""
    segments = [sg for sg in segments if sg[0].numel() > 0]
    for k in range(0, len(segments), _lib.SGD_MAX_SEGMENTS):
        part = segments[k:k + _lib.SGD_MAX_SEGMENTS]
        arr = (_lib.SgdSegment * len(part))()
        for a, (p, g, buf, lr, wd, first) in zip(arr, part):
            _chk(p, g, buf)
            assert p.numel() == g.numel() == buf.numel()
            a.p, a.g, a.buf, a.n = p.data_ptr(), g.data_ptr(), buf.data_ptr(), p.numel()
            a.lr, a.wd, a.first_step, a.reserved = float(lr), float(wd), int(bool(first)), 0
        call("scan_sgd_momentum_multi", arr, len(part), float(momentum), _stream())


def sgd_momentum_(p, g, buf, lr, wd, momentum, first_step):
    _chk(p, g, buf)
    call("scan_sgd_momentum", _ptr(p), _ptr(g), _ptr(buf), p.numel(), float(lr), float(wd), float(momentum),
         int(first_step), _stream())
