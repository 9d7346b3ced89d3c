"""oracle/scan_ref.py -- torch-CPU restatement of SCAN's hot path.

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this file; scan_amd/ never does.

Plain fp32 torch ops on CPU (F.conv2d / group_norm / softmax / autograd), one
function per reference unit, each citing the reference file:line it follows
(paths relative to the reference checkout, fcos_core/...).  Parameters are
plain dicts keyed by the reference's state_dict names.

Pinning: every function here is checked against the imported reference modules
in the authoring container (oracle/make_golden.py --check) and against the
golden vectors that script wrote into tests/golden/ (tests/test_oracle.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

INF = 100000000  # rpn/fcos/loss.py:22
FPN_STRIDES = (8, 16, 32, 64, 128)  # config/defaults.py:339
SIZES_OF_INTEREST = ((-1, 64), (64, 128), (128, 256), (256, 512), (512, INF))  # loss.py:41-47
VGG_CONVS = ((0, 2), (5, 7), (10, 12, 14), (17, 19, 21), (24, 26, 28))  # mmdetection/vgg.py:52-57


def params(sd, requires_grad=True, frozen_prefixes=(), dtype=torch.float32):
    """dtype=torch.float64: the same iteration in double precision -- not the reference's arithmetic but the yardstick for its
    rounding error (oracle/make_golden.py gen_traj_yaml: how far the fp32 reference itself is from exact arithmetic)."""
    out = {}
    for k, v in sd.items():
        t = v.detach().clone().to(dtype) if v.is_floating_point() else v.detach().clone().float()
        bn_buffer = ".bn" in k or ".downsample.1." in k  # FrozenBatchNorm2d holds buffers, not parameters
        if requires_grad and t.is_floating_point() and k != "prototype" and not bn_buffer \
                and not k.startswith(tuple(frozen_prefixes)):
            t.requires_grad_(True)
        out[k] = t
    return out


# ----------------------------------------------------------------------------- backbone
def vgg_fpn_forward(p, x):
    """VGG16 body (backbone/mmdetection/vgg.py:154-169) + FPN (backbone/fpn.py:44-91)
    + LastLevelP6P7 on P5 (fpn.py:118-130, USE_C5 False).  Returns [P3..P7]."""
    outs = []
    for stage in VGG_CONVS:
        for idx in stage:
            x = F.relu(F.conv2d(x, p["body.features.%d.weight" % idx], p["body.features.%d.bias" % idx], padding=1))
        x = F.max_pool2d(x, 2, 2)
        outs.append(x)
    c3, c4, c5 = outs[2], outs[3], outs[4]

    def conv(name, t, stride=1, pad=0):
        return F.conv2d(t, p[name + ".weight"], p[name + ".bias"], stride=stride, padding=pad)

    inner5 = conv("fpn.fpn_inner5", c5)
    p5 = conv("fpn.fpn_layer5", inner5, pad=1)
    inner4 = conv("fpn.fpn_inner4", c4) + F.interpolate(inner5, scale_factor=2, mode="nearest")
    p4 = conv("fpn.fpn_layer4", inner4, pad=1)
    inner3 = conv("fpn.fpn_inner3", c3) + F.interpolate(inner4, scale_factor=2, mode="nearest")
    p3 = conv("fpn.fpn_layer3", inner3, pad=1)
    p6 = conv("fpn.top_blocks.p6", p5, stride=2, pad=1)
    p7 = conv("fpn.top_blocks.p7", F.relu(p6), stride=2, pad=1)
    return [p3, p4, p5, p6, p7]


def _frozen_bn(p, name, x):
    """FrozenBatchNorm2d.forward (layers/batch_norm.py:17-24): no eps."""
    scale = p[name + ".weight"] * p[name + ".running_var"].rsqrt()
    bias = p[name + ".bias"] - p[name + ".running_mean"] * scale
    return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


def resnet_fpn_forward(p, x):
    """R-50/101-FPN-RETINANET (backbone/backbone.py:94-117): StemWithFixedBatchNorm (resnet.py:316-336),
    BottleneckWithFixedBatchNorm blocks with the stride in the first 1x1 (resnet.py:228-314), FPN on C3..C5
    (in_channels_list [0, 512, 1024, 2048] -> fpn_inner/layer 2..4) + LastLevelP6P7 on P5.  Returns [P3..P7]."""
    x = F.relu(_frozen_bn(p, "body.stem.bn1", F.conv2d(x, p["body.stem.conv1.weight"], stride=2, padding=3)))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = []
    for i in range(1, 5):
        b = 0
        while "body.layer%d.%d.conv1.weight" % (i, b) in p:
            q = "body.layer%d.%d" % (i, b)
            stride = 2 if (i > 1 and b == 0) else 1
            out = F.relu(_frozen_bn(p, q + ".bn1", F.conv2d(x, p[q + ".conv1.weight"], stride=stride)))
            out = F.relu(_frozen_bn(p, q + ".bn2", F.conv2d(out, p[q + ".conv2.weight"], padding=1)))
            out = _frozen_bn(p, q + ".bn3", F.conv2d(out, p[q + ".conv3.weight"]))
            idt = x
            if q + ".downsample.0.weight" in p:
                idt = _frozen_bn(p, q + ".downsample.1", F.conv2d(x, p[q + ".downsample.0.weight"], stride=stride))
            x = F.relu(out + idt)
            b += 1
        outs.append(x)
    c3, c4, c5 = outs[1], outs[2], outs[3]

    def conv(name, t, stride=1, pad=0):
        return F.conv2d(t, p[name + ".weight"], p[name + ".bias"], stride=stride, padding=pad)

    inner5 = conv("fpn.fpn_inner4", c5)
    p5 = conv("fpn.fpn_layer4", inner5, pad=1)
    inner4 = conv("fpn.fpn_inner3", c4) + F.interpolate(inner5, scale_factor=2, mode="nearest")
    p4 = conv("fpn.fpn_layer3", inner4, pad=1)
    inner3 = conv("fpn.fpn_inner2", c3) + F.interpolate(inner4, scale_factor=2, mode="nearest")
    p3 = conv("fpn.fpn_layer2", inner3, pad=1)
    p6 = conv("fpn.top_blocks.p6", p5, stride=2, pad=1)
    p7 = conv("fpn.top_blocks.p7", F.relu(p6), stride=2, pad=1)
    return [p3, p4, p5, p6, p7]


def backbone_forward(p, x):
    return resnet_fpn_forward(p, x) if "body.stem.conv1.weight" in p else vgg_fpn_forward(p, x)


def tower(p, prefix, x, n, gn=True):
    """n x [conv3x3, GroupNorm(32), ReLU] (condgraph.py:86-106, fcos.py:25-49,
    fcos_head_discriminator_con.py:20-34); gn=False drops the norm (head_out)."""
    step = 3 if gn else 2
    for i in range(n):
        x = F.conv2d(x, p["%s.%d.weight" % (prefix, step * i)], p["%s.%d.bias" % (prefix, step * i)], padding=1)
        if gn:
            x = F.group_norm(x, 32, p["%s.%d.weight" % (prefix, step * i + 1)], p["%s.%d.bias" % (prefix, step * i + 1)])
        x = F.relu(x)
    return x


# ----------------------------------------------------------------------------- locations / targets
def compute_locations(feats, strides=FPN_STRIDES):
    """condgraph.py:631-655 / fcos.py:234-258: x-fastest grid + stride//2."""
    locs = []
    for f, s in zip(feats, strides):
        h, w = f.shape[-2:]
        xs = torch.arange(0, w * s, step=s, dtype=torch.float32)
        ys = torch.arange(0, h * s, step=s, dtype=torch.float32)
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        locs.append(torch.stack((xx.reshape(-1), yy.reshape(-1)), dim=1) + s // 2)
    return locs


def assign_targets(locations, targets):
    """loss.py:40-126 (identical copy at :262-343).  targets: list of
    (boxes [G,4], labels [G]).  Returns level-first labels [N*HW_l] (int64) and
    reg targets [N*HW_l, 4]."""
    npl = [len(l) for l in locations]
    soi = torch.cat([torch.tensor(SIZES_OF_INTEREST[l], dtype=torch.float32)[None].expand(n, -1)
                     for l, n in enumerate(npl)], 0)
    pts = torch.cat(locations, 0)
    xs, ys = pts[:, 0], pts[:, 1]
    labels, regs = [], []
    for boxes, lab in targets:
        boxes = boxes.float()
        area = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)  # BoxList.area, bounding_box.py:226-236
        l = xs[:, None] - boxes[:, 0][None]
        t = ys[:, None] - boxes[:, 1][None]
        r = boxes[:, 2][None] - xs[:, None]
        b = boxes[:, 3][None] - ys[:, None]
        reg = torch.stack([l, t, r, b], dim=2)
        inside = reg.min(dim=2)[0] > 0
        mx = reg.max(dim=2)[0]
        cared = (mx >= soi[:, [0]]) & (mx <= soi[:, [1]])
        a = area[None].repeat(len(pts), 1)
        a[inside == 0] = INF
        a[cared == 0] = INF
        amin, gi = a.min(dim=1)
        reg = reg[range(len(pts)), gi]
        lb = lab[gi].clone()
        lb[amin == INF] = 0
        labels.append(torch.split(lb, npl, 0))
        regs.append(torch.split(reg, npl, 0))
    lab_lf = [torch.cat([li[l] for li in labels], 0) for l in range(len(locations))]
    reg_lf = [torch.cat([ri[l] for ri in regs], 0) for l in range(len(locations))]
    return lab_lf, reg_lf


def centerness_targets(reg):
    """loss.py:128-133."""
    lr = reg[:, [0, 2]]
    tb = reg[:, [1, 3]]
    return torch.sqrt((lr.min(-1)[0] / lr.max(-1)[0]) * (tb.min(-1)[0] / tb.max(-1)[0]))


# ----------------------------------------------------------------------------- pointwise losses
def sigmoid_focal_loss(logits, targets, gamma=2.0, alpha=0.25):
    """SigmoidFocalLoss_cuda.cu:20-58 (the stable CUDA formula), element-wise."""
    C = logits.shape[1]
    cls = torch.arange(1, C + 1, dtype=targets.dtype)[None]
    t = targets[:, None]
    p = torch.sigmoid(logits)
    term1 = (1 - p) ** gamma * torch.log(p.clamp(min=torch.finfo(torch.float32).tiny))
    ge = (logits >= 0).float()
    term2 = p ** gamma * (-logits * ge - torch.log(1 + torch.exp(logits - 2 * logits * ge)))
    return -(t == cls).float() * term1 * alpha - ((t != cls) & (t >= 0)).float() * term2 * (1 - alpha)


def iou_loss(pred, target, weight=None):
    """layers/iou_loss.py:5-36."""
    ta = (target[:, 0] + target[:, 2]) * (target[:, 1] + target[:, 3])
    pa = (pred[:, 0] + pred[:, 2]) * (pred[:, 1] + pred[:, 3])
    wi = torch.min(pred[:, 0], target[:, 0]) + torch.min(pred[:, 2], target[:, 2])
    hi = torch.min(pred[:, 3], target[:, 3]) + torch.min(pred[:, 1], target[:, 1])
    ai = wi * hi
    au = ta + pa - ai
    losses = -torch.log((ai + 1.0) / (au + 1.0))
    if weight is not None and weight.sum() > 0:
        return (losses * weight).sum() / weight.sum()
    return losses.mean()


def softmax_focal_loss(logits, labels, gamma=2):
    """layers/sigmoid_focal_loss_wbg.py:38-64 (the FocalLoss actually bound,
    layers/__init__.py:24): alpha = 1, mean over rows."""
    P = logits.softmax(dim=1)
    probs = P.gather(1, labels.view(-1, 1))
    if (probs < 1e-15).sum() > 0:
        probs = probs.clamp(min=1e-15)
    return (-(1 - probs) ** gamma * probs.log()).mean()


# ----------------------------------------------------------------------------- middle head (condgraph)
def sample_source_nodes(feats, labels_lf):
    """PrototypeComputation.__call__ source branch, loss.py:428-463."""
    C = feats[0].shape[1]
    pos_pts, pos_lab, neg_pts = [], [], []
    for f, lab in zip(feats, labels_lf):
        flat = f.permute(0, 2, 3, 1).reshape(-1, C)
        lab = lab.reshape(-1)
        pi = lab > 0
        ni = lab == 0
        pos_pts.append(flat[pi])
        pos_lab.append(lab[pi])
        negs = flat[ni]
        n_pos, n_neg = int(pi.sum()), int(ni.sum())
        if n_pos > n_neg:
            neg_pts.append(negs)
        else:
            idx = list(np.floor(np.linspace(0, n_neg - 2, n_pos)).astype(int))
            neg_pts.append(negs[idx])
    pos_pts = torch.cat(pos_pts, 0)
    pos_lab = torch.cat(pos_lab, 0)
    neg_pts = torch.cat(neg_pts, 0)
    pts = torch.cat([neg_pts, pos_pts], 0)
    labs = torch.cat([pos_lab.new_zeros(neg_pts.shape[0]), pos_lab])
    return pts, labs


def multihead_attention(p, x, dropout_p=0.0):
    """layers/transformer.py:57-90 with key=value=query=x [1,n,256], 4 heads.
    Heads are a plain .view(4,-1,64) of the [1,n,256] tensor (no transpose);
    scale = (64 // 4) ** -0.5."""
    pre = "multihead_attn."
    k = F.linear(x, p[pre + "linear_k.weight"], p[pre + "linear_k.bias"]).view(4, -1, 64)
    v = F.linear(x, p[pre + "linear_v.weight"], p[pre + "linear_v.bias"]).view(4, -1, 64)
    q = F.linear(x, p[pre + "linear_q.weight"], p[pre + "linear_q.bias"]).view(4, -1, 64)
    att = torch.bmm(q, k.transpose(1, 2)) * ((64 // 4) ** -0.5)
    att = F.dropout(att.softmax(dim=2), dropout_p)
    ctx = torch.bmm(att, v).view(1, -1, 256)
    out = F.dropout(F.linear(ctx, p[pre + "linear_final.weight"], p[pre + "linear_final.bias"]), dropout_p)
    return F.layer_norm(x + out, (256,), p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"])


def forward_gcns(p, pts, labs, K=9, lam=1.0):
    """GRAPHModule._forward_gcns, GLOBAL_GCN branch (condgraph.py:386-402)."""
    nodes = multihead_attention(p, pts.unsqueeze(0)).squeeze()
    proto_batch = pts.new_zeros(K, 256)
    rows = []
    for i in range(K):
        idx = labs == i
        rows.append(nodes[idx].mean(dim=0) if idx.any() else proto_batch[i])
    proto_batch = torch.stack(rows, 0)
    h = F.relu(F.linear(nodes, p["proto_cls_hidden.weight"], p["proto_cls_hidden.bias"]))
    logits = F.linear(h, p["proto_cls.weight"], p["proto_cls.bias"])
    return lam * F.cross_entropy(logits, labs.long()), proto_batch


class PrototypeState:
    """prototype buffer [K,256,3] + PROTOTYPECounter(stop=True) (condgraph.py:46-65)."""

    def __init__(self, prototype, cycle=3):
        self.prototype = prototype.detach().clone()
        self.cycle = cycle
        self.counter = -1

    def tick(self):
        if self.counter == self.cycle:
            return self.cycle
        self.counter += 1
        return self.counter


def update_prototype(state, proto_batch):
    """update_prototype_nx1_rnn with COSINE_UPDATE_ON (condgraph.py:586-606)."""
    it = state.tick()
    pb = proto_batch.detach()
    exist = pb.sum(-1).bool()
    P = state.prototype
    if it == state.cycle:
        m = F.cosine_similarity(P[exist, :, it - 1], pb[exist]).unsqueeze(1)
        for i in range(it - 1):
            P[:, :, i] = P[:, :, i + 1]
        P[exist, :, it - 1] = P[exist, :, it - 1] * m + pb[exist] * (1 - m)
    else:
        m = F.cosine_similarity(P[exist, :, it], pb[exist]).unsqueeze(1)
        P[exist, :, it] = P[exist, :, it] * m + pb[exist] * (1 - m)


def conded_weight(p, prototype):
    """get_conded_weight, RNN branch (condgraph.py:313-319): prototype [K,256,T]
    -> seq-first tanh RNN(256->512, 2 layers) -> Conv2d(512,256,(T,1)) -> [K,256]."""
    x = prototype.permute(2, 0, 1)  # [T, K, 256]
    for layer in range(2):
        wih, whh = p["cond_rnn.weight_ih_l%d" % layer], p["cond_rnn.weight_hh_l%d" % layer]
        bih, bhh = p["cond_rnn.bias_ih_l%d" % layer], p["cond_rnn.bias_hh_l%d" % layer]
        h = x.new_zeros(x.shape[1], 512)
        outs = []
        for t in range(x.shape[0]):
            h = torch.tanh(F.linear(x[t], wih, bih) + F.linear(h, whh, bhh))
            outs.append(h)
        x = torch.stack(outs, 0)
    y = x.permute(1, 2, 0).unsqueeze(-1)  # [K, 512, T, 1]
    return F.conv2d(y, p["cond_nx1.weight"], p["cond_nx1.bias"]).squeeze()


def act_maps_from(feats, w):
    """dynamic_conv (condgraph.py:619-629) + softmax over K (condgraph.py:344-346)."""
    logits = [F.conv2d(f, w.view(w.shape[0], -1, 1, 1)) for f in feats]
    return logits, [l.softmax(dim=1) for l in logits]


def middle_head_source(p, state, feats, targets, K=9):
    """GRAPHModule.forward -> _forward_train_source (condgraph.py:423-445, 547-551)."""
    feats = [tower(p, "head_in.middle_tower", f, 2) for f in feats]
    locs = compute_locations(feats)
    labels_lf, _ = assign_targets(locs, targets)
    pts, labs = sample_source_nodes(feats, labels_lf)
    node_loss, proto_batch = forward_gcns(p, pts, labs, K)
    update_prototype(state, proto_batch)
    w = conded_weight(p, state.prototype)
    logits, maps = act_maps_from(feats, w)
    flat = torch.cat([l.permute(0, 2, 3, 1).reshape(-1, K) for l in logits], 0)
    lab = torch.cat([l.reshape(-1) for l in labels_lf], 0)
    act_loss = softmax_focal_loss(flat, lab.long())
    out = [tower(p, "head_out.middle_tower", torch.cat([f, m], 1), 1, gn=False) for f, m in zip(feats, maps)]
    return out, node_loss, act_loss, maps


def dbscan_mask(act_fg, feat, eps=3, thr=0.05):
    """PrototypeComputation.DBSCAN_batch_cpu (loss.py:397-423).  act_fg [N,CLS,H,W] (foreground act maps),
    feat [N,C,H,W].  Points = feat * act[c] at entries with act > thr, in (n, cls, h, w) order; sklearn DBSCAN;
    noise (-1) -> 1, cluster 0 -> 0; a pixel is selected if any class entry is non-zero."""
    from sklearn import cluster
    act_fg = act_fg.detach()
    feat = feat.detach()
    N, CLS, H, W = act_fg.shape
    fn = torch.cat([(feat * act_fg[:, i].unsqueeze(1)).unsqueeze(0) for i in range(CLS)], 0)
    fn = fn.permute(1, 0, 3, 4, 2).reshape(-1, feat.shape[1])
    mask = (act_fg > thr).reshape(-1)
    mask_float = mask.float()
    pos = fn[mask]
    if pos.bool().any():
        Y = cluster.DBSCAN(eps=eps, n_jobs=-1).fit_predict(pos.numpy())
        Y[Y < 0] = 1
        mask_float[mask] = torch.from_numpy(Y.astype(np.float32))
    Y = mask_float.reshape(N, CLS, H, W)
    return Y.permute(0, 2, 3, 1).reshape(-1, CLS).sum(-1).bool()


def sample_target_nodes(feats, maps, eps=3, thr=0.05):
    """PrototypeComputation.__call__ target branch, 'dbscan' (loss.py:464-518)."""
    pos_pts, pos_lab, neg_pts = [], [], []
    for f, m in zip(feats, maps):
        C, K = f.shape[1], m.shape[1]
        conf = dbscan_mask(m[:, 1:], f, eps, thr)
        if conf.any():
            act = m.permute(0, 2, 3, 1).reshape(-1, K)
            flat = f.permute(0, 2, 3, 1).reshape(-1, C)
            pos_pts.append(flat[conf])
            pos_lab.append(act[conf, 1:].argmax(dim=-1) + 1)
            negs = flat[~conf]
            idx = list(np.floor(np.linspace(0, int((~conf).sum()) - 2, int(conf.sum()))).astype(int))
            neg_pts.append(negs[idx])
    if not pos_pts:
        return None, None
    pos_pts, pos_lab, neg_pts = torch.cat(pos_pts, 0), torch.cat(pos_lab, 0), torch.cat(neg_pts, 0)
    return torch.cat([neg_pts, pos_pts], 0), torch.cat([pos_lab.new_zeros(neg_pts.shape[0]), pos_lab])


def sim_matrix(a, b, eps=1e-8):
    """condgraph.py:35-43."""
    a_n, b_n = a.norm(dim=1)[:, None], b.norm(dim=1)[:, None]
    return torch.mm(a / torch.clamp(a_n, min=eps), (b / torch.clamp(b_n, min=eps)).transpose(0, 1))


def transfer_loss(prototype, tg_prototype, tg_nodes, tg_labels):
    """get_transfer_loss with TRANSFER_CFG ('NODES', 'ADJ') (condgraph.py:457-498)."""
    sr = prototype.mean(dim=-1).detach()
    # nn.KLDivLoss() default reduction 'mean' = mean over all elements
    l_node = F.kl_div(tg_nodes.softmax(-1).log(), sr[tg_labels.long()].softmax(-1), reduction="mean")
    indx = tg_prototype.sum(dim=-1).bool()
    adj_sr = sim_matrix(sr[indx], sr[indx]).view(1, -1)
    adj_tg = sim_matrix(tg_prototype[indx], tg_prototype[indx]).view(1, -1)
    l_adj = F.cosine_embedding_loss(adj_sr, adj_tg, adj_sr.new_ones(1), margin=0.0)
    return l_node + l_adj


def middle_head_target(p, state, feats, K=9, eps=3, thr=0.05, lam3=1.0, transfer=True):
    """GRAPHModule._forward_train_target (condgraph.py:500-534), GCN_SELF_TRAINING False."""
    feats = [tower(p, "head_in.middle_tower", f, 2) for f in feats]
    w = conded_weight(p, state.prototype)
    _, maps = act_maps_from(feats, w)
    pts, labs = sample_target_nodes(feats, maps, eps, thr)
    out = [tower(p, "head_out.middle_tower", torch.cat([f, m], 1), 1, gn=False) for f, m in zip(feats, maps)]
    if pts is None or not transfer:  # TRANSFER_CFG (None,): condgraph.py:521 skips the GST branch
        return out, None, maps
    _, tg_proto = forward_gcns(p, pts, labs, K)
    return out, lam3 * transfer_loss(state.prototype, tg_proto, pts, labs), maps


def middle_head_plain(p, state, feats, K=9):
    """target pass with forward_target False / inference (condgraph.py:536-545):
    head_in -> act maps -> head_out, no losses."""
    feats = [tower(p, "head_in.middle_tower", f, 2) for f in feats]
    w = conded_weight(p, state.prototype)
    _, maps = act_maps_from(feats, w)
    out = [tower(p, "head_out.middle_tower", torch.cat([f, m], 1), 1, gn=False) for f, m in zip(feats, maps)]
    return out, maps


# ----------------------------------------------------------------------------- FCOS head + loss
def fcos_head(p, feats):
    """FCOSHead.forward with REG_CTR_ON (fcos.py:89-114)."""
    logits, reg, ctr = [], [], []
    for l, f in enumerate(feats):
        ct = tower(p, "head.cls_tower", f, 4)
        logits.append(F.conv2d(ct, p["head.cls_logits.weight"], p["head.cls_logits.bias"], padding=1))
        rt = tower(p, "head.bbox_tower", f, 4)
        ctr.append(F.conv2d(rt, p["head.centerness.weight"], p["head.centerness.bias"], padding=1))
        bp = F.conv2d(rt, p["head.bbox_pred.weight"], p["head.bbox_pred.bias"], padding=1)
        reg.append(torch.exp(bp * p["head.scales.%d.scale" % l]))
    return logits, reg, ctr


def fcos_loss(box_cls, box_reg, ctr, targets, gamma=2.0, alpha=0.25):
    """FCOSLossComputation.__call__ (loss.py:168-230)."""
    N, C = box_cls[0].shape[:2]
    locs = compute_locations(box_cls)
    labels, regs = assign_targets(locs, targets)
    cls_f = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, C) for t in box_cls], 0)
    reg_f = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, 4) for t in box_reg], 0)
    ctr_f = torch.cat([t.reshape(-1) for t in ctr], 0)
    lab_f = torch.cat([l.reshape(-1) for l in labels], 0)
    rt_f = torch.cat([r.reshape(-1, 4) for r in regs], 0)
    pos = torch.nonzero(lab_f > 0).squeeze(1)
    cls_loss = sigmoid_focal_loss(cls_f, lab_f.int(), gamma, alpha).sum() / (pos.numel() + N)
    reg_f, rt_f, ctr_f = reg_f[pos], rt_f[pos], ctr_f[pos]
    if pos.numel() > 0:
        ct = centerness_targets(rt_f)
        reg_loss = iou_loss(reg_f, rt_f, ct)
        ctr_loss = F.binary_cross_entropy_with_logits(ctr_f, ct)
    else:
        reg_loss = reg_f.sum()
        ctr_loss = ctr_f.sum()
    return cls_loss, reg_loss, ctr_loss


# ----------------------------------------------------------------------------- CKA discriminator
class _GRL(torch.autograd.Function):
    """discriminator/layer.py:6-24."""

    @staticmethod
    def forward(ctx, x, lam):
        ctx.lam = lam
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return -ctx.lam * g, None


def discriminator_loss(p, feat, act, domain_label, K=9, grl_lambda=0.02):
    """FCOSDiscriminator_con.forward, 'concat' fusion, GRL on both inputs
    (fcos_head_discriminator_con.py:88-126)."""
    feat = _GRL.apply(feat, grl_lambda)
    act = _GRL.apply(act, grl_lambda)
    x = tower(p, "dis_tower", feat, 4)
    ncls = K - 1
    loss = 0
    for c in range(ncls):
        a = act[:, c + 1].unsqueeze(1)
        h = F.relu(F.conv2d(torch.cat((x, a), 1), p["classifier_cls_%d.0.weight" % c],
                            p["classifier_cls_%d.0.bias" % c], padding=1))
        logit = F.conv2d(h, p["classifier_cls_%d.2.weight" % c], p["classifier_cls_%d.2.bias" % c], padding=1)
        tgt = torch.full_like(logit, domain_label)
        if ncls > 1:
            lc = F.binary_cross_entropy_with_logits(logit, tgt, weight=a.detach(), reduction="sum") / a.sum().detach()
        else:
            lc = F.binary_cross_entropy_with_logits(logit, tgt)
        loss = loss + lc / ncls
    return loss


def pad_images(images, size_divisible=32):
    """to_image_list(list, size_divisible) (structures/image_list.py:29-72): zero-pad bottom/right to the common
    size rounded up to a multiple of size_divisible; returns (batch [N,3,H,W], true sizes)."""
    if isinstance(images, torch.Tensor):
        return images, [tuple(images.shape[-2:])] * images.shape[0]
    h = max(i.shape[1] for i in images)
    w = max(i.shape[2] for i in images)
    h = -(-h // size_divisible) * size_divisible
    w = -(-w // size_divisible) * size_divisible
    out = torch.zeros(len(images), images[0].shape[0], h, w)
    for o, i in zip(out, images):
        o[:, :i.shape[1], :i.shape[2]] = i
    return out, [tuple(i.shape[-2:]) for i in images]


# ----------------------------------------------------------------------------- DA iteration
def da_iteration(P, state, images_s, targets_s, images_t, con_lambda=0.1, K=9, skip_dead_target_fcos=True,
                 forward_target=False, transfer=True):
    """Three-phase DA iteration (engine/trainer.py:266-385) with forward_target
    False: returns the loss dict (floats); gradients accumulate in P[...].grad.
    The target-pass FCOS head only yields the identically-zero 'zero' loss
    (fcos.py:215-220); it is skipped unless skip_dead_target_fcos=False."""
    out = {}
    images_s, _ = pad_images(images_s)
    images_t, _ = pad_images(images_t)
    feats = backbone_forward(P["backbone"], images_s)
    f_s, node_loss, act_loss, maps_s = middle_head_source(P["middle_head"], state, feats, targets_s, K)
    lg, rg, ct = fcos_head(P["fcos"], f_s)
    lc, lr, lctr = fcos_loss(lg, rg, ct, targets_s)
    gs = {"node_loss_gs": node_loss, "act_loss_gs": act_loss, "loss_cls_gs": lc, "loss_reg_gs": lr,
          "loss_centerness_gs": lctr}
    sum(gs.values()).backward(retain_graph=True)
    out.update({k: float(v.detach()) for k, v in gs.items()})
    ds = {}
    for i, lvl in reversed(list(enumerate(("P3", "P4", "P5", "P6", "P7")))):
        ds["loss_adv_%s_CON_ds" % lvl] = con_lambda * discriminator_loss(P["dis_%s_CON" % lvl], f_s[i], maps_s[i], 1.0, K)
    sum(ds.values()).backward()
    out.update({k: float(v.detach()) for k, v in ds.items()})
    feats = backbone_forward(P["backbone"], images_t)
    dt = {}
    if forward_target:
        f_t, cons, maps_t = middle_head_target(P["middle_head"], state, feats, K, transfer=transfer)
        if cons is not None:
            dt["consistency_loss_gt"] = cons
    else:
        f_t, maps_t = middle_head_plain(P["middle_head"], state, feats, K)
    if not skip_dead_target_fcos:
        lg, rg, ct = fcos_head(P["fcos"], f_t)
        dt["zero_gt"] = 0.0 * sum(0.0 * x.sum() for x in lg + rg + ct)
    for i, lvl in reversed(list(enumerate(("P3", "P4", "P5", "P6", "P7")))):
        dt["loss_adv_%s_CON_dt" % lvl] = con_lambda * discriminator_loss(P["dis_%s_CON" % lvl], f_t[i], maps_t[i], 0.0, K)
    sum(dt.values()).backward()
    out.update({k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in dt.items()})
    return out


def lr_factor(iteration, warmup_iters=1000, factor=1.0 / 3, steps=(60000, 80000), gamma=0.1, method="constant"):
    """WarmupMultiStepLR.get_lr / base_lr with last_epoch = iteration (solver/lr_scheduler.py:39-52)."""
    from bisect import bisect_right
    f = 1.0
    if iteration < warmup_iters:
        if method == "constant":
            f = factor
        else:
            alpha = float(iteration) / warmup_iters
            f = factor * (1 - alpha) + alpha
    return f * gamma ** bisect_right(list(steps), iteration)


def sgd_step(P, bufs, lr=0.0025, momentum=0.9, wd=1e-4, warmup=1.0 / 3, solver=None, iteration=0):
    """solver/build.py:7-43 + lr_scheduler.py:39-52: torch.optim.SGD(momentum, dampening 0) with one group per
    parameter -- weights lr / wd, biases lr * BIAS_LR_FACTOR / WEIGHT_DECAY_BIAS; parameters without a gradient are
    skipped (no decay, no buffer).  Default: the C2F yaml inside its constant warm-up (factor 1/3 for 1000
    iterations).  solver = {sub-model or 'dis': dict(lr, bias_lr_factor, wd, wd_bias, momentum, steps, gamma,
    warmup_iters, warmup_factor, warmup_method)} + iteration selects per-sub-model schedules."""
    with torch.no_grad():
        for mname, pd in P.items():
            if solver is not None:
                sv = solver["dis" if mname.startswith("dis_") else mname]
                f = lr_factor(iteration, sv["warmup_iters"], sv["warmup_factor"], sv["steps"], sv["gamma"],
                              sv["warmup_method"])
                lr_w, lr_b, wd_w, wd_b, mom = sv["lr"] * f, sv["lr"] * sv["bias_lr_factor"] * f, sv["wd"], sv["wd_bias"], \
                    sv["momentum"]
            else:
                lr_w, lr_b, wd_w, wd_b, mom = lr * warmup, 2 * lr * warmup, wd, 0.0, momentum
            for k, v in pd.items():
                if v.grad is None:
                    continue
                is_bias = "bias" in k
                g = v.grad + (wd_b if is_bias else wd_w) * v
                key = mname + "/" + k
                if key not in bufs:
                    bufs[key] = g.clone()
                else:
                    bufs[key].mul_(mom).add_(g)
                v.add_(bufs[key], alpha=-(lr_b if is_bias else lr_w))
                v.grad = None


# ----------------------------------------------------------------------------- inference
def postprocess(locs, box_cls, box_reg, ctr, image_sizes, nms_fn, mode="common", pre_nms_thresh=0.05,
                pre_nms_top_n=1000, nms_thresh=0.6, post_top_n=100, num_classes=9):
    """FCOSPostProcessor (rpn/fcos/inference.py:54-194).  box_cls must already
    be fused for 'precision'/'light' (fcos.py:162-169).  nms_fn(boxes, scores,
    thr) -> kept indices ascending.  Returns per image (boxes, scores, labels)."""
    per_img = [[] for _ in image_sizes]
    for loc, cls, reg, ct in zip(locs, box_cls, box_reg, ctr):
        N, C, H, W = cls.shape
        cls = cls.permute(0, 2, 3, 1).reshape(N, -1, C)
        if mode == "common":
            cls = cls.sigmoid()
        reg = reg.permute(0, 2, 3, 1).reshape(N, -1, 4)
        ct = ct.permute(0, 2, 3, 1).reshape(N, -1).sigmoid()
        cand = cls > pre_nms_thresh
        topn = cand.reshape(N, -1).sum(1).clamp(max=pre_nms_top_n)
        cls = cls * ct[:, :, None]
        for i in range(N):
            sc = cls[i][cand[i]]
            nz = cand[i].nonzero()
            bl, kl = nz[:, 0], nz[:, 1] + 1
            rg, lc = reg[i][bl], loc[bl]
            if cand[i].sum().item() > topn[i].item():
                sc, ti = sc.topk(int(topn[i]), sorted=False)
                kl, rg, lc = kl[ti], rg[ti], lc[ti]
            det = torch.stack([lc[:, 0] - rg[:, 0], lc[:, 1] - rg[:, 1], lc[:, 0] + rg[:, 2], lc[:, 1] + rg[:, 3]], 1)
            h, w = image_sizes[i]
            det[:, 0].clamp_(min=0, max=w - 1)
            det[:, 1].clamp_(min=0, max=h - 1)
            det[:, 2].clamp_(min=0, max=w - 1)
            det[:, 3].clamp_(min=0, max=h - 1)
            ws, hs = det[:, 2] - det[:, 0] + 1, det[:, 3] - det[:, 1] + 1
            keep = ((ws >= 0) & (hs >= 0)).nonzero().squeeze(1)
            per_img[i].append((det[keep], torch.sqrt(sc)[keep], kl[keep]))
    results = []
    for lv in per_img:
        boxes = torch.cat([x[0] for x in lv], 0)
        scores = torch.cat([x[1] for x in lv], 0)
        labels = torch.cat([x[2] for x in lv], 0)
        rb, rs, rl = [], [], []
        for j in range(1, num_classes):
            inds = (labels == j).nonzero().view(-1)
            bj, sj = boxes[inds].view(-1, 4), scores[inds]
            keep = nms_fn(bj, sj, nms_thresh)
            rb.append(bj[keep])
            rs.append(sj[keep])
            rl.append(torch.full((len(keep),), j, dtype=torch.int64))
        rb, rs, rl = torch.cat(rb), torch.cat(rs), torch.cat(rl)
        n = len(rs)
        if n > post_top_n > 0:
            th, _ = torch.kthvalue(rs, n - post_top_n + 1)
            k = torch.nonzero(rs >= th.item()).squeeze(1)
            rb, rs, rl = rb[k], rs[k], rl[k]
        results.append((rb, rs, rl))
    return results


def inference(P, state, images, nms_fn, mode="precision", K=9):
    """eval path: backbone -> _forward_inference (condgraph.py:536-545) -> FCOS
    head -> score fusion (fcos.py:162-169) -> post-processor."""
    with torch.no_grad():
        images, sizes = pad_images(images)
        feats = backbone_forward(P["backbone"], images)
        f, maps = middle_head_plain(P["middle_head"], state, feats, K)
        lg, rg, ct = fcos_head(P["fcos"], f)
        if mode == "light":
            lg = [m[:, 1:] for m in maps]
        elif mode == "precision":
            lg = [0.5 * l.sigmoid() + 0.5 * m[:, 1:] for l, m in zip(lg, maps)]
        locs = compute_locations(f)
        return postprocess(locs, lg, rg, ct, sizes, nms_fn, mode=mode, num_classes=K)
