"""SCAN "middle head" (GRAPHModule) on pyramid activations.

Mirrors the reference's fcos_core/modeling/rpn/fcos/condgraph.py (GRAPHHead :68-119,
GRAPHModule :122-655) and the source node sampling of rpn/fcos/loss.py:428-463 for the
configuration SCAN ships (C2F yaml: RNN paradigm, PROTO_ITER 3, COSINE_UPDATE_ON, PROTO_WITH_BG,
GLOBAL_GCN, softmaxFL act loss, CAT_ACT_MAP).  Parameter/buffer names equal the reference's
(prototype, head_in.middle_tower.N, head_out.middle_tower.0, proto_cls_hidden, proto_cls,
multihead_attn.*, cond_nx1, cond_rnn.*, cond_2).

Device split (north star): the conv towers, the semantic-conditioned dynamic convolution +
softmax and the activation-map focal loss are HIP kernels; graph-node aggregation (attention over
a few hundred nodes), the paradigm EMA and the RNN kernel generator are small torch-tier ops.
"""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from .. import ops
from ..layers import FocalLoss, MultiHeadAttention
from .backbone import conv_holder
from .fcos import make_tower, run_tower, source_node_index, target_plan


class PROTOTYPECounter:
    """reference condgraph.py:46-65."""

    def __init__(self, cycle=3, stop=False):
        self.cycle = cycle
        self.counter = -1
        self.stop = stop

    def __call__(self):
        if self.stop:
            if self.counter == self.cycle:
                return self.cycle
            self.counter += 1
            return self.counter
        self.counter += 1
        if self.counter == self.cycle:
            self.counter = 0
        return self.counter


class GRAPHHead(nn.Module):
    """reference condgraph.py:68-119: n x [conv3x3, (GN32), ReLU]; GN only for mode 'in'."""

    def __init__(self, in_channels, out_channel, num_convs, mode="in"):
        super().__init__()
        self.mode = mode
        self.num_convs = num_convs
        self.in_channels = in_channels
        if mode == "in":
            self.middle_tower = make_tower(num_convs, in_channels)
        else:
            layers = []
            for _ in range(num_convs):
                layers += [conv_holder(in_channels, out_channel, 3), nn.ReLU()]
            self.middle_tower = nn.Sequential(*layers)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, std=0.01)
                nn.init.constant_(m.bias, 0)

    def forward(self, rows, shape):
        if self.mode == "in":
            return run_tower(self.middle_tower, rows, shape, self.num_convs)
        for i in range(self.num_convs):
            conv = self.middle_tower[2 * i]
            rows = ops.conv2d(rows, conv.weight, conv.bias, shape, 3, 1, relu=True)
        return rows


class _RNNParams(nn.Module):
    """Parameter holder with nn.RNN(256, 512, 2, 'tanh') names (cond_rnn.weight_ih_l0, ...)."""

    def __init__(self, cin=256, hidden=512, layers=2):
        super().__init__()
        bound = hidden ** -0.5
        for l in range(layers):
            for name, shp in (("weight_ih", (hidden, cin if l == 0 else hidden)), ("weight_hh", (hidden, hidden)),
                              ("bias_ih", (hidden,)), ("bias_hh", (hidden,))):
                setattr(self, "%s_l%d" % (name, l), nn.Parameter(torch.empty(shp).uniform_(-bound, bound)))
        self.layers = layers
        self.hidden = hidden

    def forward(self, x):  # x [T, B, cin], seq-first, h0 = 0
        for l in range(self.layers):
            wih, whh = getattr(self, "weight_ih_l%d" % l), getattr(self, "weight_hh_l%d" % l)
            bih, bhh = getattr(self, "bias_ih_l%d" % l), getattr(self, "bias_hh_l%d" % l)
            h = x.new_zeros(x.shape[1], self.hidden)
            outs = []
            for t in range(x.shape[0]):
                h = torch.tanh(F.linear(x[t], wih, bih) + F.linear(h, whh, bhh))
                outs.append(h)
            x = torch.stack(outs, 0)
        return x


def sample_source_nodes(feats, labels, shape):
    """PrototypeComputation.__call__ source branch (reference loss.py:428-463): node features and labels, order
    [all neg, all pos] (fcos.source_node_index picks the rows)."""
    index, node_labels = source_node_index(labels, shape)
    return feats[index], node_labels


def sim_matrix(a, b, eps=1e-8):
    """reference condgraph.py:35-43."""
    a_n, b_n = a.norm(dim=1)[:, None], b.norm(dim=1)[:, None]
    return torch.mm(a / torch.clamp(a_n, min=eps), (b / torch.clamp(b_n, min=eps)).transpose(0, 1))


# SCAN_FUSED_PARADIGM=1: update_prototype_nx1_rnn as ONE launch (scan_paradigm_update) instead of ~25 one-row torch launches.  OFF by
# default: the kernel's reductions run in another order than torch's, the paradigm then differs from the torch spelling's by ~1e-7,
# and on the N = 1 cfg-5 fixture that perturbation moves ONE sampled element of a P7 discriminator bias gradient (231 rows behind a ReLU)
# from 1.3e-2 to 2.01e-2 of its scale -- over the 2e-2 bar of tests/test_gpu_model.py.  ~0.1 ms per step is not worth a new allowance.
FUSED_PARADIGM_UPDATE = os.environ.get("SCAN_FUSED_PARADIGM", "0") == "1"
DBSCAN_BACKEND = "device"  # "host": sklearn on the host cores, the reference's own call (kept for cross-checks)
# MEASUREMENT ONLY (bench.py --ft-positives, SURVEY.md 8d "second series"): a random-init model's act maps are nearly
# uniform, so EVERY (pixel, class) entry passes the 0.05 threshold and the clustering sees 100 % of the entries -- a
# trained model passes a small fraction.  A value p in (0, 1] keeps a fixed pseudo-random fraction p of the entries that
# pass the threshold as clustering candidates (seeded per level: the same entries every iteration).  None = the
# reference's rule, always used outside that bench series.
FT_CANDIDATE_FRACTION = None
_ft_masks = {}


def dbscan_positive_rows(feat_l, act_l, n_images, eps, thr):
    """PrototypeComputation.DBSCAN_batch_cpu (reference loss.py:397-423) on one level.  feat_l [N*HW, C], act_l
    [N*HW, K] rows.  Points are feat * act[c] at the (image, class, pixel) entries with act > thr, in that order;
    noise (-1) -> 1, cluster 0 -> 0; a pixel row is selected when any of its class entries is non-zero -- i.e. a point
    counts iff it is NOT in DBSCAN's cluster 0, which ops.dbscan_in_cluster0 decides on the device (the reference
    runs sklearn with n_jobs=-1 on the host: O(n^2), seconds per iteration once tens of thousands of pixels pass)."""
    K = act_l.shape[1]
    hw = feat_l.shape[0] // n_images
    act = act_l.detach().view(n_images, hw, K)[:, :, 1:].permute(0, 2, 1).contiguous()  # [N, CLS, HW]
    mask = act > thr
    if FT_CANDIDATE_FRACTION is None:
        _ft_masks.clear()
    else:
        key = (tuple(act.shape), str(act.device), float(FT_CANDIDATE_FRACTION))
        keep = _ft_masks.get(key)
        if keep is None:
            g = torch.Generator(device=act.device).manual_seed(977 + act.shape[2])
            keep = _ft_masks[key] = torch.rand(act.shape, generator=g, device=act.device) < FT_CANDIDATE_FRACTION
        mask = mask & keep
    idx = mask.nonzero()  # (n, cls, hw) lexicographic = the reference's (n, cls, h, w) order
    sel = torch.zeros_like(act)
    if idx.shape[0] > 0:
        pts = feat_l.detach().view(n_images, hw, -1)[idx[:, 0], idx[:, 2]] * act[mask][:, None]
        if bool(pts.bool().any()):
            if DBSCAN_BACKEND == "host":
                from sklearn import cluster
                y = cluster.DBSCAN(eps=eps, n_jobs=-1).fit_predict(pts.cpu().numpy())
                y[y < 0] = 1
                sel[mask] = torch.from_numpy(y.astype(np.float32)).to(sel.device)
            else:
                sel[mask] = (~ops.dbscan_in_cluster0(pts.contiguous(), eps, 5)).to(sel.dtype)
        else:
            sel[mask] = 1.0
    return sel.sum(1).bool().reshape(-1)  # [N*HW] in row order


def sample_target_nodes(feats, maps, shape, eps=3, thr=0.05):
    """PrototypeComputation.__call__ target branch, TARGET_SAMPLING_CFG 'dbscan' (reference loss.py:464-518):
    pseudo-label = argmax foreground act map; as many background rows as positives, linspace-subsampled."""
    pos_pts, pos_lab, neg_pts = [], [], []
    for l in range(shape.n_levels):
        r0, r1 = shape.row_off[l], shape.row_off[l + 1]
        f, m = feats[r0:r1], maps[r0:r1]
        conf = dbscan_positive_rows(f, m, shape.n_images, eps, thr)
        if bool(conf.any()):
            pi = torch.nonzero(conf).squeeze(1)
            ni = torch.nonzero(~conf).squeeze(1)
            pos_pts.append(f[pi])
            pos_lab.append(m[pi, 1:].argmax(dim=-1) + 1)
            idx = np.floor(np.linspace(0, ni.numel() - 2, pi.numel())).astype(np.int64)
            neg_pts.append(f[ni[torch.from_numpy(idx).to(ni.device)]])
    if not pos_pts:
        return None, None
    pos_pts, pos_lab, neg_pts = torch.cat(pos_pts, 0), torch.cat(pos_lab, 0), torch.cat(neg_pts, 0)
    return torch.cat([neg_pts, pos_pts], 0), torch.cat([pos_lab.new_zeros(neg_pts.shape[0]), pos_lab])


_shift_cache = {}
_SHIFT_CACHE_MAX = 64  # ragged batches: every padded size adds two entries; oldest first out


def _shift_put(key, value):
    if len(_shift_cache) >= _SHIFT_CACHE_MAX:
        _shift_cache.pop(next(iter(_shift_cache)))
    _shift_cache[key] = value
    return value


def _level_bounds(shape_src, device):
    """row offsets of levels 1 .. L-1 of the source-only pyramid (bucketize boundaries: row -> level)"""
    key = ("b", tuple(shape_src.sizes), shape_src.n_images, str(device))
    if key not in _shift_cache:
        return _shift_put(key, torch.tensor(list(shape_src.row_off[1:shape_src.n_levels]), dtype=torch.int64, device=device))
    return _shift_cache[key]


def _level_shift(shape_src, shape, device):
    """per level: (row offset in the joint pyramid) - (row offset in the source-only pyramid)"""
    key = ("s", tuple(shape.sizes), shape_src.n_images, shape.n_images, str(device))
    if key not in _shift_cache:
        return _shift_put(key, torch.tensor([shape.row_off[l] - shape_src.row_off[l] for l in range(shape.n_levels)],
                                            dtype=torch.int64, device=device))
    return _shift_cache[key]


class GRAPHModule(nn.Module):
    """model["middle_head"]."""

    def __init__(self, in_channels=256, num_classes=9, proto_iter=3, attn_dropout=0.1, transfer_cfg=("NODES", "ADJ"),
                 dbscan_eps=3, dbscan_thr=0.05):
        super().__init__()
        self.transfer_cfg = tuple(transfer_cfg)
        self.dbscan_eps, self.dbscan_thr = dbscan_eps, dbscan_thr
        self.lamda3 = self.lamda4 = 1.0
        self.num_classes_fg = num_classes - 1
        self.used_num_classes = num_classes  # PROTO_WITH_BG
        self.prototype_iter = proto_iter
        K = self.used_num_classes
        self.lamda1 = self.lamda2 = 1.0
        self.head_in = GRAPHHead(in_channels, in_channels, 2, mode="in")
        self.register_buffer("prototype", torch.randn(K, 256, proto_iter))
        self.head_out = GRAPHHead(in_channels + K, in_channels, 1, mode="out")
        self.cat_stride = ops.pad4(in_channels + K)
        self.out_stream = None  # engine.Trainer: the stream head_out's feature share runs on beside the graph tier
        self.act_loss_func = FocalLoss(K)
        self.proto_cls_hidden = nn.Linear(256, 512)
        self.proto_cls = nn.Linear(512, K)
        self.multihead_attn = MultiHeadAttention(256, 4, dropout=attn_dropout)
        self.cond_nx1 = nn.Conv2d(512, 256, kernel_size=(proto_iter, 1))
        self.cond_rnn = _RNNParams(256, 512, 2)
        self.counter_rnn = PROTOTYPECounter(proto_iter, stop=True)
        self.cond_2 = nn.Linear(512, 256)  # present in the reference state_dict, unused in RNN mode
        for m in (self.cond_2, self.proto_cls, self.proto_cls_hidden):
            nn.init.normal_(m.weight, std=0.01)
            nn.init.constant_(m.bias, 0)

    # ---- graph tier (torch ops on a few hundred nodes) ----
    def _forward_gcns(self, pts, labs):
        """reference condgraph.py:386-402 (GLOBAL_GCN)."""
        K = self.used_num_classes
        nodes = self.multihead_attn(pts.unsqueeze(0), pts.unsqueeze(0), pts.unsqueeze(0))[0].squeeze(0)
        onehot = F.one_hot(labs.long(), K).to(nodes.dtype)  # [n, K]
        cnt = onehot.sum(0)
        means = (onehot.t() @ nodes) / cnt.clamp(min=1)[:, None]
        proto_batch = torch.where((cnt > 0)[:, None], means, torch.zeros_like(means))
        logits = self.proto_cls(F.relu(self.proto_cls_hidden(nodes)))
        node_loss = self.lamda1 * F.cross_entropy(logits, labs.long())
        return node_loss, proto_batch

    @torch.no_grad()
    def update_prototype_nx1_rnn(self, proto_batch):
        """reference condgraph.py:586-606 with COSINE_UPDATE_ON."""
        it = self.counter_rnn()
        pb = proto_batch.detach()
        if dist.is_available() and dist.is_initialized():
            # data parallel: average the per-rank class means over the ranks that saw
            # the class, so the paradigm buffer stays identical on every rank (9 x 257 floats)
            ex = pb.sum(-1).bool().to(pb.dtype)[:, None]
            buf = torch.cat([pb * ex, ex], 1)
            dist.all_reduce(buf)
            pb = buf[:, :-1] / buf[:, -1:].clamp(min=1)
        # The reference indexes with the boolean mask `exist` (P[exist, :, t] = ...), which costs a host<->device
        # synchronisation per use (nonzero) in the middle of the forward pass -- the host then cannot enqueue ahead and
        # the GPU idles through the small-kernel tier that follows.  Row-wise arithmetic on ALL classes and a masked
        # select give the same rows: cosine_similarity works row by row, absent classes (pb = 0) keep their old value.
        P = self.prototype
        if P.is_cuda and P.dtype == torch.float32 and P.is_contiguous() and P.shape[1] <= 1024 and FUSED_PARADIGM_UPDATE:
            # the rest of this function as ONE launch (csrc/pointwise.hip: paradigm_update_kernel) instead of ~25 one-row torch
            # launches on the critical small-kernel stretch of the step
            ops.call("scan_paradigm_update", ops._ptr(P), ops._ptr(pb.contiguous().float()), P.shape[0], P.shape[1], P.shape[2],
                     int(it), ops._stream())
            return
        exist = pb.sum(-1).bool()[:, None]
        slot = it - 1 if it == self.prototype_iter else it
        cur = P[:, :, slot]
        m = F.cosine_similarity(cur, pb).unsqueeze(1)
        upd = torch.where(exist, cur * m + pb * (1 - m), cur)
        if it == self.prototype_iter:
            for i in range(it - 1):
                P[:, :, i] = P[:, :, i + 1]
        P[:, :, slot] = upd

    def get_conded_weight(self):
        """reference condgraph.py:313-319: paradigm [K,256,T] -> RNN -> Conv2d(512,256,(T,1)) -> [K,256]."""
        r = self.cond_rnn
        params = [getattr(r, "%s_l%d" % (n, l)) for l in range(r.layers)
                  for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")] + [self.cond_nx1.weight, self.cond_nx1.bias]
        if r.layers == 2 and ops.cond_rnn_supported(self.prototype, params):
            return ops.cond_rnn(self.prototype, params)  # the whole chain below as one launch (csrc/condrnn.hip)
        seq = self.cond_rnn(self.prototype.permute(2, 0, 1))  # [T, K, 512]
        # Conv2d(512, 256, (T,1)) over a [K,512,T,1] input is one linear map over (c, t)
        x = seq.permute(1, 2, 0).reshape(seq.shape[1], -1)  # [K, 512*T], (c, t) order
        w = self.cond_nx1.weight.reshape(self.cond_nx1.weight.shape[0], -1)  # [256, 512*T]
        return F.linear(x, w, self.cond_nx1.bias)

    # ---- HIP tier ----
    def _out_features(self, feats, shape, side=False):
        """head_out's share of the 256 feature channels (ops.HEAD_OUT_SPLIT): conv(cat(feats, maps)) = conv(feats; W[:, :256])
        + conv(maps; W[:, 256:]), and the first -- 97 % of the layer -- does not wait for the graph tier (node sampling,
        GCN, attention, paradigm RNN: a few hundred tiny launches with the GPU nearly idle).  With ``out_stream`` set (by
        engine.Trainer) and side=True (the paired step; in the three-phase schedule the passes already run beside each
        other and a third MFMA-dense stream cost 1.3 ms) it runs on that stream BESIDE the graph tier, and so do its data
        and weight gradients in the backward; the 265-channel input the kernels handled badly (a third, almost empty 128-wide channel tile in the data
        gradient, the old weight-gradient kernel) becomes 256 clean channels.  Returns (rows, stream they are produced on or
        None), or None when the split is off."""
        if not ops.HEAD_OUT_SPLIT or self.head_out.num_convs != 1:
            return None
        conv = self.head_out.middle_tower[0]
        C = feats.shape[1]
        if self.out_stream is None or not side:
            return ops.conv2d(feats, conv.weight[:, :C], conv.bias, shape, 3, 1), None
        self.out_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.out_stream):
            return ops.conv2d(feats, conv.weight[:, :C], conv.bias, shape, 3, 1), self.out_stream

    def _act_and_out(self, feats, shape, kernels, main=None):
        logits, maps = ops.dynconv_softmax(feats, kernels)
        if main is not None:  # relu(feature share + act-map share): same sum as the one conv over the concatenation
            main, main_stream = main
            conv = self.head_out.middle_tower[0]
            C, K = feats.shape[1], maps.shape[1]
            thin_in = ops.pad_cols(maps, ops.pad4(K))  # [M, 9] -> [M, 12], zero tail, one pass
            thin = ops.conv2d(thin_in, conv.weight[:, C:C + K], None, shape, 3, 1)
            if main_stream is not None:
                torch.cuda.current_stream().wait_stream(main_stream)
                # allocated from the side stream's pool, read here: tell the caching allocator, or a later allocation on
                # that stream which does not first wait for this one could be handed the block while it is still read
                main.record_stream(torch.cuda.current_stream())
            return logits, maps, ops.add_relu(main, thin)
        pad = self.cat_stride - feats.shape[1] - maps.shape[1]
        cat = torch.cat([feats, maps, feats.new_zeros(feats.shape[0], pad)], 1)
        return logits, maps, self.head_out(cat, shape)

    def get_transfer_loss(self, tg_prototype, tg_nodes, tg_labels):
        """Graph-guided semantic transfer (reference condgraph.py:457-498) for TRANSFER_CFG in {NODES, ADJ}."""
        sr = self.prototype.mean(dim=-1).detach()
        losses = []
        if "NODES" in self.transfer_cfg or "NODE" in self.transfer_cfg:
            # nn.KLDivLoss() default reduction: mean over all elements
            losses.append(F.kl_div(tg_nodes.softmax(-1).log(), sr[tg_labels.long()].softmax(-1), reduction="mean"))
        if "ADJ" in self.transfer_cfg:
            indx = tg_prototype.sum(dim=-1).bool()
            adj_sr = sim_matrix(sr[indx], sr[indx]).view(1, -1)
            adj_tg = sim_matrix(tg_prototype[indx], tg_prototype[indx]).view(1, -1)
            losses.append(F.cosine_embedding_loss(adj_sr, adj_tg, adj_sr.new_ones(1), margin=0.0))
        return sum(losses) if losses else None

    def _forward_train_target(self, feats, shape):
        """reference condgraph.py:500-534 (GCN_SELF_TRAINING False): act maps with the conditioned kernels (HIP),
        DBSCAN node sampling (host), graph aggregation + GST losses (torch tier)."""
        main = self._out_features(feats, shape)
        kernels = self.get_conded_weight()
        _, maps, out = self._act_and_out(feats, shape, kernels, main)
        pts, labs = sample_target_nodes(feats, maps, shape, self.dbscan_eps, self.dbscan_thr)
        if pts is not None and self.transfer_cfg and self.transfer_cfg[0] is not None:
            _, tg_proto = self._forward_gcns(pts, labs)
            tl = self.get_transfer_loss(tg_proto, pts, labs)
            if tl is not None:
                tl = self.lamda3 * tl
            return out, (None, tl), None, maps
        return out, None, None, maps

    def forward(self, rows, shape, targets=None, mode="source", forward_target=False):
        """-> feats [M,256], (node_loss, transfer_loss) or None, act_loss or None, act_maps [M,K]."""
        feats = self.head_in(rows, shape)
        if self.training and targets and mode == "source":
            main = self._out_features(feats, shape)
            plan = target_plan(shape, targets, rows.device)
            labels = plan.labels
            pts, labs = feats[plan.node_index], plan.node_labels
            node_loss, proto_batch = self._forward_gcns(pts, labs)
            self.update_prototype_nx1_rnn(proto_batch)
            kernels = self.get_conded_weight()
            logits, maps, out = self._act_and_out(feats, shape, kernels, main)
            act_loss = self.lamda2 * self.act_loss_func(logits, labels.long())
            return out, (node_loss, 0), act_loss, maps
        if self.training and mode == "target" and forward_target:
            return self._forward_train_target(feats, shape)
        main = self._out_features(feats, shape)
        kernels = self.get_conded_weight()
        _, maps, out = self._act_and_out(feats, shape, kernels, main)
        return out, None, None, maps


    def forward_pair(self, rows, shape, targets, n_src, forward_target=False):
        """Source and target frames in ONE pyramid (images [0, n_src) are the source batch): the towers, the dynamic
        conv and head_out run once over all frames; node sampling, the paradigm update and the act loss use the source
        rows only.  Same arithmetic as forward(mode='source') followed by forward(mode='target'): the target frames
        see the kernels generated from the paradigm the source frames have just updated (reference trainer.py:284-296
        then :346-352).  With forward_target the target rows additionally go through DBSCAN node sampling, the graph
        aggregation and the GST losses (_forward_train_target).
        -> out [M,256], node_loss, act_loss, act_maps [M,K], consistency loss or None."""
        feats = self.head_in(rows, shape)
        main = self._out_features(feats, shape, side=True)
        shape_src = ops.PyramidShape(n_src, shape.sizes)
        plan = target_plan(shape_src, targets, rows.device)
        # the sampled nodes are rows of the SOURCE images; in the joint pyramid a level holds its source images first, so a
        # source-pyramid row index only shifts by the difference of the level offsets -- gather them straight from `feats`
        # (round 5: was take_images(feats) = a 90 MB copy of the source rows forward, a 179 MB zero-fill + copy backward,
        # for a gather of ~2 k rows)
        node_rows = plan.node_index + _level_shift(shape_src, shape, rows.device)[
            torch.bucketize(plan.node_index, _level_bounds(shape_src, rows.device), right=True)]
        node_loss, proto_batch = self._forward_gcns(feats[node_rows], plan.node_labels)
        self.update_prototype_nx1_rnn(proto_batch)
        kernels = self.get_conded_weight()
        logits, maps, out = self._act_and_out(feats, shape, kernels, main)
        act_loss = self.lamda2 * self.act_loss_func(ops.take_images(logits, shape, 0, n_src)[0], plan.labels.long())
        consistency = None
        if forward_target:
            tgt, shape_t = ops.take_images(feats, shape, n_src, shape.n_images)
            tmaps, _ = ops.take_images(maps, shape, n_src, shape.n_images)
            pts, labs = sample_target_nodes(tgt, tmaps, shape_t, self.dbscan_eps, self.dbscan_thr)
            if pts is not None and self.transfer_cfg and self.transfer_cfg[0] is not None:
                _, tg_proto = self._forward_gcns(pts, labs)
                tl = self.get_transfer_loss(tg_proto, pts, labs)
                if tl is not None:
                    consistency = self.lamda3 * tl
        return out, node_loss, act_loss, maps, consistency


def build_condgraph(cfg=None, in_channels=256, num_classes=9, transfer_cfg=("NODES", "ADJ")):
    """reference rpn/rpn.py:215 build_middle_head(cfg, in_channels).  cfg: a config.settings dict (MODEL.MIDDLE_HEAD.*
    and MODEL.FCOS.NUM_CLASSES as the reference's GRAPHModule.__init__ reads them, condgraph.py:127-253) or None for
    the keyword arguments.  transfer_cfg = MODEL.MIDDLE_HEAD.TRANSFER_CFG: ('NODES', 'ADJ') in the C2F yaml, the
    default (None,) (reference config/defaults.py:694) in the Sim10k / KITTI yamls."""
    if cfg is None:
        return GRAPHModule(in_channels, num_classes, transfer_cfg=transfer_cfg)
    if cfg["num_convs_in"] != 2 or cfg["num_convs_out"] != 1:
        raise ValueError("MODEL.MIDDLE_HEAD.NUM_CONVS_IN/OUT other than 2/1 are not built")
    m = GRAPHModule(in_channels, cfg["num_classes"], proto_iter=cfg["proto_iter"], transfer_cfg=cfg["transfer_cfg"],
                    dbscan_eps=cfg["dbscan_eps"], dbscan_thr=cfg["dbscan_thr"])
    m.lamda1, m.lamda2 = cfg["gcn_loss_weight"], cfg["act_loss_weight"]  # condgraph.py:160-163
    m.lamda3, m.lamda4 = cfg["con_loss_weight"], cfg["gcn_loss_weight_tg"]
    return m
