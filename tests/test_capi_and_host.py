"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/scan_hip.h declares,
argument validation works without a device, and the host-side logic (label assignment, node sampling,
paradigm counter, LR schedule, state_dict names, flat parameter groups) matches the reference's vectors."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from scan_amd import _lib, engine, ops, synth
from scan_amd.modeling import condgraph, fcos

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "scan_hip.h")).read()
    declared = set(re.findall(r"\b(scan_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("scan_pyramid_t")
    L = _lib.lib()
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), "libscan_hip.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert L.scan_abi_version() == 1


def test_argument_validation_without_device():
    d = ops.PyramidShape(1, [(4, 4)])
    with pytest.raises(RuntimeError, match="Cin_s"):
        _lib.call("scan_conv2d_forward", None, d.ref(), 6, None, None, None, d.ref(), 8, 8, 3, 1, 0, None)
    with pytest.raises(RuntimeError, match="ksize"):
        _lib.call("scan_conv2d_forward", None, d.ref(), 8, None, None, None, d.ref(), 8, 8, 4, 1, 0, None)
    with pytest.raises(RuntimeError, match="SCAN_NMS_MAX"):
        _lib.call("scan_nms", None, None, None, _lib.NMS_MAX + 1, 0.5, 1, None, ctypes.c_void_p(8), None, None)
    with pytest.raises(RuntimeError, match="K in"):
        _lib.call("scan_dynconv_softmax_forward", None, None, 10, 256, 5, None, None, None)
    assert _lib.query("scan_nms_ws_bytes", _lib.NMS_MAX + 1) == -1
    assert _lib.query("scan_nms_ws_bytes", 100) > 0
    # beyond one panel the mask is n x ceil(n / 64) words (the reference's own size, csrc/cuda/nms.cu:95-100) + sort keys
    n = 20000
    assert n * ((n + 63) // 64) * 8 <= _lib.query("scan_nms_ws_bytes", n) <= 1.1 * n * ((n + 63) // 64) * 8 + (1 << 20)


def test_ops_refuse_cpu_tensors():
    with pytest.raises(RuntimeError, match="GPU"):
        ops.sigmoid_focal_loss_sum(torch.zeros(4, 8), torch.zeros(4, dtype=torch.int32), 2.0, 0.25)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.groupnorm_relu(torch.zeros(16, 256), torch.ones(256), torch.zeros(256), ops.PyramidShape(1, [(4, 4)]))
    from scan_amd import _C
    with pytest.raises(RuntimeError, match="CPU"):
        _C.ml_nms(torch.zeros(3, 4), torch.zeros(3), torch.zeros(3), 0.5)
    assert _C.nms(torch.zeros(0, 4), torch.zeros(0), 0.5).numel() == 0
    with pytest.raises(RuntimeError):
        _C.roi_align_forward()


def _need_pybind11():
    """the two compiled modules need pybind11 + g++ at build time (__graft_entry__.build_extension): without them
    scan_amd.layers serves the same operator surface through ctypes and the compiled-module tests do not apply"""
    try:
        import pybind11  # noqa: F401
    except ImportError:
        pytest.skip("pybind11 not importable: compiled extension modules are not built in this environment")


def _import_fcos_core_C():
    """the compiled module under the name the reference imports it by: ``from fcos_core import _C``"""
    import importlib
    import sys
    _need_pybind11()
    ext = os.path.join(ROOT, "scan_amd", "ext")
    if ext not in sys.path:
        sys.path.insert(0, ext)
    return importlib.import_module("fcos_core._C")


def test_compiled_fcos_core_C_module_loads_and_refuses_cpu_tensors():
    """scan_amd/csrc/fcos_core_C.cpp built by __graft_entry__.build(): importable as fcos_core._C, exports what the
    reference's csrc/vision.cpp:8-17 binds, and behaves like it off the GPU (no compute here: there is no GPU)."""
    _C = _import_fcos_core_C()
    for name in ("nms", "ml_nms", "sigmoid_focalloss_forward", "sigmoid_focalloss_backward", "roi_align_forward",
                 "roi_align_backward", "roi_pool_forward", "roi_pool_backward"):
        assert callable(getattr(_C, name)), name
    assert _C.scan_abi_version() == _lib.lib().scan_abi_version()
    k = _C.nms(torch.zeros(0, 4), torch.zeros(0), 0.5)  # csrc/nms.h:17-18
    assert k.numel() == 0 and k.dtype == torch.int64 and k.device.type == "cpu"
    with pytest.raises(RuntimeError, match="same type"):  # csrc/cpu/nms_cpu.cpp:11
        _C.nms(torch.zeros(3, 4), torch.zeros(3, dtype=torch.float64), 0.5)
    with pytest.raises(RuntimeError, match="CPU"):
        _C.ml_nms(torch.zeros(3, 4), torch.zeros(3), torch.zeros(3), 0.5)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        _C.sigmoid_focalloss_forward(torch.zeros(4, 8), torch.zeros(4, dtype=torch.int32), 8, 2.0, 0.25)
    with pytest.raises(RuntimeError):
        _C.roi_align_forward()
    from scan_amd import layers
    assert layers.C_BACKEND == "compiled" and layers.nms.__module__ is not None


def _both_C():
    from scan_amd import _C as ctypes_C
    return (("compiled", _import_fcos_core_C()), ("ctypes", ctypes_C))


def test_nms_on_cpu_tensors_runs_the_reference_unit_test(gold_dir):
    """The reference's only hot-path unit test, tests/test_nms.py:11-58,60-217 (Caffe2's UtilsNMSTest cases, recorded
    with their expected keep lists in tests/golden/nms_kat.json by oracle/make_golden.py while that unittest passed on the
    imported reference), runs on CPU tensors: the reference's ``_C.nms`` dispatches those to nms_cpu
    (csrc/nms.h:26, csrc/cpu/nms_cpu.cpp:5-75).  Same call, same answers, through the compiled module and its ctypes twin."""
    kat = json.load(open(os.path.join(gold_dir, "nms_kat.json")))
    assert len(kat["cases"]) == 6
    for name, _C in _both_C():
        for case in kat["cases"]:
            boxes, scores = torch.tensor(case["boxes"], dtype=torch.float32), torch.tensor(case["scores"], dtype=torch.float32)
            keep = _C.nms(boxes, scores, case["thresh"])
            assert keep.dtype == torch.int64 and keep.device.type == "cpu", name
            assert np.sort(keep.numpy()).tolist() == case["keep_sorted"], (name, case["thresh"])
            # AT_DISPATCH_FLOATING_TYPES (nms_cpu.cpp:71): double tensors take the same loop in double
            assert np.sort(_C.nms(boxes.double(), scores.double(), case["thresh"]).numpy()).tolist() == case["keep_sorted"]


def test_nms_on_cpu_tensors_bit_exact_against_oracle_both_tie_rules():
    """random boxes on a 4-pixel grid (exact IoU ties with the threshold, duplicate boxes) and tied scores: the host loop of
    the compiled module and of its ctypes twin against the C oracle's keep list (>=), and the '>' rule of
    csrc/cuda/nms.cu:60 through ``cuda_rule=True`` / SCAN_NMS_RULE=gt against the oracle's ml_nms with one label."""
    from oracle import coracle
    for n in (1, 2, 65, 700, 3000):
        rs = np.random.RandomState(n)
        # corners on a 4-pixel grid with x2 = 4 j - 1: widths + 1 are multiples of 4, so IoU hits 1/4 and 1/2 exactly
        xy = np.floor(rs.uniform(0, 200, (n, 2)) / 4.0) * 4.0
        boxes = np.concatenate([xy, xy + np.ceil(rs.uniform(1, 70, (n, 2)) / 4.0) * 4.0 - 1.0], 1).astype(np.float32)
        scores = (rs.randint(0, max(2, n // 3), n) / float(max(2, n // 3))).astype(np.float32)
        one = np.ones(n, np.float32)
        differ = 0
        for thr in (0.25, 0.5, 0.6):
            ge, gt = coracle.nms(boxes, scores, thr), coracle.ml_nms(boxes, scores, one, thr)
            differ += int(not np.array_equal(ge, gt))
            for name, _C in _both_C():
                if name == "ctypes" and n > 700:
                    continue  # the Python twin is a loop over kept boxes; the compiled one takes every size
                b, s = torch.from_numpy(boxes), torch.from_numpy(scores)
                assert np.array_equal(_C.nms(b, s, thr).numpy(), ge), (name, n, thr)
                assert np.array_equal(_C.nms(b, s, thr, cuda_rule=True).numpy(), gt), (name, n, thr)
                os.environ["SCAN_NMS_RULE"] = "gt"
                try:
                    assert np.array_equal(_C.nms(b, s, thr).numpy(), gt), (name, n, thr)
                finally:
                    del os.environ["SCAN_NMS_RULE"]
        if n >= 700:
            assert differ > 0, "the two tie rules never disagreed: the cases do not exercise the switch"


def test_compiled_scan_ops_module_loads_and_refuses_cpu_tensors():
    """scan_amd/ext/scan_ops/_ops (csrc/scan_ops_ext.cpp, built by __graft_entry__.build()): the conv / GroupNorm / dynamic
    conv operators with C++ autograd.  No compute here (no GPU): the module imports, is linked to this libscan_hip.so, exports
    the four operators with the documented keyword names, and fails loudly on CPU tensors (there is no fallback)."""
    _need_pybind11()
    from scan_amd import layers
    assert layers.OPS_BACKEND == "compiled"
    _ops = layers._ops
    assert _ops.scan_abi_version() == _lib.lib().scan_abi_version()
    for name in ("conv2d", "conv3x3_gn_relu", "group_norm_relu", "dynamic_conv_softmax"):
        assert callable(getattr(_ops, name)), name
    with pytest.raises(RuntimeError, match="GPU tensor"):
        _ops.conv2d(input=torch.zeros(1, 4, 8, 8), weight=torch.zeros(8, 4, 3, 3), bias=None, stride=1, relu=False)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        _ops.conv3x3_gn_relu(torch.zeros(1, 256, 8, 8), torch.zeros(256, 256, 3, 3), None, torch.ones(256), torch.zeros(256))
    with pytest.raises(RuntimeError, match="GPU tensor"):
        _ops.group_norm_relu(torch.zeros(1, 256, 8, 8), torch.ones(256), torch.zeros(256), eps=1e-5, relu=True)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        _ops.dynamic_conv_softmax(features=torch.zeros(1, 256, 8, 8), kernel_par=torch.zeros(9, 256))


def test_pyramid_shape():
    s = ops.PyramidShape(2, [(128, 256), (64, 128), (32, 64), (16, 32), (8, 16)])
    assert s.rows == 2 * 43648 and s.row_off[1] == 2 * 128 * 256
    assert s.conv_out(3, 1) == s
    assert ops.PyramidShape(2, [(32, 64)]).conv_out(3, 2).sizes == [(16, 32)]
    assert ops.PyramidShape(1, [(7, 9)]).conv_out(3, 2).sizes == [(4, 5)]
    assert s.desc.row_off[5] == s.rows and s.desc.n_levels == 5


def test_label_assignment_matches_reference(gold_dir):
    """FCOS location->GT assignment + level-major row order (reference loss.py:40-126)."""
    gold = json.load(open(os.path.join(gold_dir, "step_128x256.json")))
    g = np.load(os.path.join(gold_dir, "step_128x256.npz"))
    H, W, N = gold["H"], gold["W"], gold["N"]
    shape = ops.PyramidShape(N, [(H // s, W // s) for s in fcos.FPN_STRIDES])
    locs = fcos.compute_locations(shape, torch.device("cpu"))
    labels, reg = fcos.assign_targets(locs, synth.synth_targets(N, H, W, 8, 12, 4321))
    for l in range(5):
        assert np.array_equal(labels[shape.row_off[l]:shape.row_off[l + 1]].numpy(), g["label_map_%d" % l])
    assert (labels > 0).sum() > 0
    pos = reg[labels > 0]
    assert (pos.min(1)[0] > 0).all()
    ct = fcos.centerness_targets(pos)
    assert ((ct > 0) & (ct <= 1)).all()


def test_source_node_sampling_order(gold_dir):
    """neg = floor(linspace(0, n_neg-2, n_pos)) rows, order [neg..., pos...] (reference loss.py:443-458)."""
    g = np.load(os.path.join(gold_dir, "step_128x256.npz"))
    shape = ops.PyramidShape(2, [(16, 32), (8, 16), (4, 8), (2, 4), (1, 2)])
    labels = torch.from_numpy(np.concatenate([g["label_map_%d" % l] for l in range(5)]))
    feats = torch.arange(shape.rows, dtype=torch.float32)[:, None].repeat(1, 4)  # row index as feature
    pts, labs = condgraph.sample_source_nodes(feats, labels, shape)
    assert np.array_equal(labs.numpy(), g["node_labels"])
    n_pos = int((labels > 0).sum())
    assert (labs[-n_pos:] > 0).all() and (labs[:-n_pos] == 0).all()
    # positives appear in row order, level by level
    assert np.array_equal(pts[-n_pos:, 0].numpy(), torch.nonzero(labels > 0).squeeze(1).float().numpy())


def test_counter_and_schedule():
    c = condgraph.PROTOTYPECounter(3, stop=True)
    assert [c() for _ in range(7)] == [0, 1, 2, 3, 3, 3, 3]  # reference condgraph.py:52-59
    c = condgraph.PROTOTYPECounter(3)
    assert [c() for _ in range(7)] == [0, 1, 2, 0, 1, 2, 0]
    assert engine.warmup_factor(0) == pytest.approx(1 / 3) and engine.warmup_factor(999) == pytest.approx(1 / 3)
    assert engine.warmup_factor(1000) == 1.0 and engine.warmup_factor(60000) == pytest.approx(0.1)
    assert engine.warmup_factor(80000) == pytest.approx(0.01)


def test_state_dict_names_and_flat_groups():
    model = engine.build_model(9, device="cpu")
    sds = synth.all_state_dicts(9)
    for k, m in model.items():
        assert set(m.state_dict().keys()) == set(sds[k].keys()), k
    engine.load_state_dicts(model, sds)
    bb = model["backbone"]
    assert not bb.body.features[0].weight.requires_grad and not bb.body.features[7].weight.requires_grad
    assert bb.body.features[10].weight.requires_grad
    w = bb.body.features[10].weight
    assert w.permute(0, 2, 3, 1).is_contiguous()  # stored OHWI: what the HIP kernels read without a repack
    grp = engine.FlatGroup(model["fcos"], 0.0025)
    n = sum(p.numel() for p in model["fcos"].parameters())
    assert grp.flat_p.numel() == n and grp.n_w + grp.n_b == n
    p = model["fcos"].head.cls_logits.weight
    assert torch.equal(p.detach(), sds["fcos"]["head.cls_logits.weight"])  # values survive the re-homing
    assert p.data.data_ptr() >= grp.flat_p.data_ptr() and p.grad.data_ptr() >= grp.flat_g.data_ptr()
    p.grad.add_(1.0)
    assert grp.flat_g.sum().item() == p.numel()
    grp.zero_grad()
    assert p.grad.abs().sum().item() == 0


def test_prototype_update_matches_reference(gold_dir):
    """three EMA updates then the slide (reference condgraph.py:586-606) against the oracle restatement."""
    from oracle import scan_ref
    mh = condgraph.GRAPHModule(256, 9)
    sd = synth.middle_head_state_dict(9)
    mh.load_state_dict(sd)
    st = scan_ref.PrototypeState(sd["prototype"])
    g = torch.Generator().manual_seed(0)
    for it in range(5):
        pb = torch.randn(9, 256, generator=g)
        if it % 2:
            pb[3] = 0  # class absent in this batch
        mh.update_prototype_nx1_rnn(pb)
        scan_ref.update_prototype(st, pb)
        assert torch.allclose(mh.prototype, st.prototype, rtol=1e-6, atol=1e-7)
    with torch.no_grad():
        k = mh.get_conded_weight()
        kr = scan_ref.conded_weight(scan_ref.params(sd, requires_grad=False), st.prototype)
    assert torch.allclose(k, kr, rtol=1e-4, atol=1e-6)


def test_checkpoint_wire_format(tmp_path):
    """reference DetectronCheckpointer layout: one .pth with model_backbone / model_fcos / middle_head /
    model_dis_P*_CON state_dicts, a last_checkpoint tag file, suffix-matched loading (ImageNet VGG keys)."""
    from scan_amd import checkpoint
    model = engine.build_model(9, device="cpu")
    engine.load_procedural_weights(model)
    path = checkpoint.save(model, str(tmp_path), "model_final", iteration=123)
    assert checkpoint.get_checkpoint_file(str(tmp_path)) == path
    raw = torch.load(path)
    assert set(raw) == {"model_backbone", "model_fcos", "middle_head", "iteration"} | {"model_dis_P%d_CON" % i for i in range(3, 8)}
    assert "body.features.0.weight" in raw["model_backbone"] and "prototype" in raw["middle_head"]
    fresh = engine.build_model(9, device="cpu")
    rest = checkpoint.load(fresh, path, load_dis=False)
    assert rest["iteration"] == 123
    a, b = model["fcos"].state_dict(), fresh["fcos"].state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert torch.equal(model["middle_head"].prototype, fresh["middle_head"].prototype)
    assert not torch.equal(model["dis_P3_CON"].dis_tower[0].weight, fresh["dis_P3_CON"].dis_tower[0].weight)  # load_dis=False
    assert fresh["backbone"].body.features[10].weight.permute(0, 2, 3, 1).is_contiguous()  # re-homed channels-last
    # ImageNet-style VGG file: keys "features.N.*" fill "body.features.N.*"; DDP "module." prefix is stripped
    vgg = {"module." + k[len("body."):]: v + 1 for k, v in synth.backbone_state_dict().items() if k.startswith("body.")}
    vpath = os.path.join(str(tmp_path), "vgg16.pth")
    torch.save(vgg, vpath)
    checkpoint.load(fresh, vpath)
    assert torch.equal(fresh["backbone"].body.features[0].weight, synth.backbone_state_dict()["body.features.0.weight"] + 1)
    assert torch.equal(fresh["backbone"].fpn.fpn_inner3.weight, model["backbone"].fpn.fpn_inner3.weight)  # untouched


def test_to_image_list_padding():
    """structures/image_list.py:29-72: zero pad bottom/right to the common size rounded up to /32, keep true sizes."""
    from scan_amd.structures import ImageList, to_image_list
    a, b = torch.ones(3, 33, 50), 2 * torch.ones(3, 20, 70)
    il = to_image_list([a, b], 32)
    assert isinstance(il, ImageList) and tuple(il.tensors.shape) == (2, 3, 64, 96)
    assert il.image_sizes == [(33, 50), (20, 70)]
    assert float(il.tensors[0, :, :33, :50].min()) == 1.0 and float(il.tensors[0].sum()) == 3 * 33 * 50
    assert float(il.tensors[1].sum()) == 2 * 3 * 20 * 70
    assert to_image_list(il) is il
    t = torch.zeros(2, 3, 64, 64)
    assert to_image_list(t).image_sizes == [(64, 64)] * 2 and to_image_list(t).tensors is t
    assert tuple(to_image_list(torch.zeros(3, 40, 40), 32).tensors.shape) == (1, 3, 64, 64)
    with pytest.raises(TypeError):
        to_image_list(np.zeros((3, 4, 4)))


def test_target_plan_is_consistent_and_per_batch(gold_dir):
    """fcos.target_plan (the once-per-batch, ground-truth-only part of the source pass): same labels / node order as the
    stand-alone functions pinned above, centerness targets on the positives, cached for one batch only."""
    gold = json.load(open(os.path.join(gold_dir, "step_128x256.json")))
    g = np.load(os.path.join(gold_dir, "step_128x256.npz"))
    H, W, N = gold["H"], gold["W"], gold["N"]
    shape = ops.PyramidShape(N, [(H // s, W // s) for s in fcos.FPN_STRIDES])
    cpu = torch.device("cpu")
    tg = synth.synth_targets(N, H, W, 8, 12, 4321)
    fcos.reset_target_plan()
    plan = fcos.target_plan(shape, tg, cpu)
    for l in range(5):
        assert np.array_equal(plan.labels[shape.row_off[l]:shape.row_off[l + 1]].numpy(), g["label_map_%d" % l])
    assert np.array_equal(plan.node_labels.numpy(), g["node_labels"])
    assert plan.labels_i32.dtype == torch.int32 and torch.equal(plan.labels_i32.long(), plan.labels)
    assert torch.equal(plan.pos_inds, torch.nonzero(plan.labels > 0).squeeze(1)) and plan.n_pos == len(plan.pos_inds)
    assert torch.equal(plan.reg_pos, plan.reg_targets[plan.pos_inds])
    assert torch.allclose(plan.ctr_pos, fcos.centerness_targets(plan.reg_pos))
    idx, labs = fcos.source_node_index(plan.labels, shape)
    assert torch.equal(idx, plan.node_index) and torch.equal(labs, plan.node_labels)
    # node features are a plain gather with that index
    feats = torch.randn(shape.rows, 8)
    pts, _ = condgraph.sample_source_nodes(feats, plan.labels, shape)
    assert torch.equal(pts, feats[plan.node_index])
    # cache: same batch object -> same plan; another batch or a reset -> a new one
    assert fcos.target_plan(shape, tg, cpu) is plan
    assert fcos.target_plan(shape, synth.synth_targets(N, H, W, 8, 12, 999), cpu) is not plan
    fcos.reset_target_plan()
    assert fcos.target_plan(shape, tg, cpu) is not plan


def test_model_configs_build_on_cpu():
    """engine.CONFIGS: the shipped yamls' model variants have the reference's parameter names / shapes."""
    for name, cfg in engine.CONFIGS.items():
        body = cfg.get("conv_body", "VGG-16-FPN-RETINANET")
        model = engine.build_model(cfg["num_classes"], cfg["test_mode"], device="cpu", transfer_cfg=cfg["transfer_cfg"],
                                   conv_body=body)
        sds = synth.all_state_dicts(cfg["num_classes"], body)
        for k, m in model.items():
            missing, unexpected = m.load_state_dict(sds[k], strict=False)
            assert not unexpected and not [x for x in missing if "cond_2" not in x], (name, k, missing, unexpected)
        K = cfg["num_classes"]
        assert model["fcos"].head.cls_logits.weight.shape[0] == K - 1
        assert model["middle_head"].prototype.shape == (K, 256, 3)
        assert model["middle_head"].transfer_cfg == tuple(cfg["transfer_cfg"])
        assert model["fcos"].box_selector_test.mode == cfg["test_mode"]
        if body.startswith("R-"):
            frozen = [n for n, p in model["backbone"].named_parameters() if not p.requires_grad]
            assert "body.stem.conv1.weight" in frozen and "body.layer1.2.conv3.weight" in frozen
            assert all(not n.startswith(("body.layer2", "body.layer3", "body.layer4", "fpn.")) for n in frozen)


def test_take_images_and_split_levels_autograd():
    """ops.take_images (sub-batch of a pyramid, used by the paired step) and ops.split_levels: forward equals plain
    indexing, backward scatters / concatenates the gradients back into the full pyramid."""
    shape = ops.PyramidShape(4, [(4, 6), (2, 3), (1, 2)])
    g = torch.Generator().manual_seed(0)
    rows = torch.randn(shape.rows, 5, generator=g, requires_grad=True)
    sub, sshape = ops.take_images(rows, shape, 1, 3)
    assert sshape.n_images == 2 and sshape.sizes == shape.sizes and sub.shape[0] == sshape.rows
    idx = torch.cat([torch.arange(shape.row_off[l] + 1 * h * w, shape.row_off[l] + 3 * h * w)
                     for l, (h, w) in enumerate(shape.sizes)])
    assert torch.equal(sub, rows[idx])
    wgt = torch.randn(sub.shape, generator=g)
    (sub * wgt).sum().backward()
    ref = torch.zeros_like(rows)
    ref[idx] = wgt
    assert torch.equal(rows.grad, ref)
    rows.grad = None
    parts = ops.split_levels(rows, shape)
    assert [p.shape[0] for p in parts] == [shape.row_off[l + 1] - shape.row_off[l] for l in range(3)]
    (parts[0].sum() * 2.0 + parts[2].sum() * 3.0).backward()  # level 1 unused: its gradient is zero
    exp = torch.zeros_like(rows)
    exp[:shape.row_off[1]] = 2.0
    exp[shape.row_off[2]:] = 3.0
    assert torch.equal(rows.grad, exp)


def test_optimizer_state_round_trip():
    """FlatGroup.optimizer_state_dict: torch.optim.SGD layout with one group per trainable parameter in
    named_parameters order (reference solver/build.py:7-43), momentum buffers in the parameters' logical shapes."""
    model = engine.build_model(9, device="cpu")
    g = engine.FlatGroup(model["fcos"], 0.0025)
    assert g.optimizer_state_dict()["state"] == {}  # no step taken yet
    g.flat_m.copy_(torch.arange(g.flat_m.numel(), dtype=torch.float32))
    g.first = False
    sd = g.optimizer_state_dict()
    names = [n for n, p in model["fcos"].named_parameters() if p.requires_grad]
    assert len(sd["param_groups"]) == len(names) == len(sd["state"])
    for i, n in enumerate(names):
        p = dict(model["fcos"].named_parameters())[n]
        grp = sd["param_groups"][i]
        assert grp["params"] == [i] and grp["momentum"] == 0.9
        assert grp["lr"] == (0.005 if "bias" in n else 0.0025) and grp["weight_decay"] == (0.0 if "bias" in n else 1e-4)
        assert sd["state"][i]["momentum_buffer"].shape == p.shape
    model2 = engine.build_model(9, device="cpu")
    g2 = engine.FlatGroup(model2["fcos"], 0.0025)
    g2.load_optimizer_state_dict(sd)
    assert torch.equal(g2.flat_m, g.flat_m) and g2.first is False
    with pytest.raises(ValueError):
        engine.FlatGroup(model2["middle_head"], 0.0025).load_optimizer_state_dict(sd)


def test_reference_written_checkpoint_loads(gold_dir, tmp_path):
    """tests/golden/refckpt_c2f.pth.gz was written by the REFERENCE's DetectronCheckpointer.save
    (utils/checkpoint.py:141-301) for the full C2F model dict with per-tensor constant values, after three optimizer /
    scheduler steps (oracle/make_golden.py gen_ckpt).  It holds the eight model state_dicts plus optimizer_ /
    scheduler_ entries for the discriminators only and no iteration.  Loading it must fill every parameter and
    buffer, restore the discriminators' momentum and take the iteration from the schedulers."""
    import gzip
    import json
    import shutil
    from scan_amd import checkpoint, engine, synth
    path = str(tmp_path / "model_0000003.pth")
    with gzip.open(os.path.join(gold_dir, "refckpt_c2f.pth.gz"), "rb") as fi, open(path, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    man = json.load(open(os.path.join(gold_dir, "refckpt_c2f.manifest.json")))
    assert "iteration" not in man["top_level_keys"] and "optimizer_backbone" not in man["top_level_keys"]
    model = engine.build_model(9, device="cpu")
    # same keys, shapes and dtypes as the file the reference wrote
    mine = checkpoint.state_to_save(model)
    for key in (k for k in man["top_level_keys"] if k.startswith("model_") or k == "middle_head"):
        assert set(mine[key].keys()) == set(man[key].keys()), key
        for n, (shape, dtype, _) in man[key].items():
            assert list(mine[key][n].shape) == shape and str(mine[key][n].dtype) == dtype, (key, n)
    trainer = engine.Trainer(model)
    rest = trainer.load_checkpoint(path)
    assert not rest, list(rest)
    # expected values: the constant of every tensor, moved by three SGD steps with the constant gradient the generator
    # set (solver/build.py:7-43 groups, constant warm-up factor 1/3) -- replayed here on scalars
    def replay(p0, g, lr, wd, steps=3, mom=0.9):
        p, buf = torch.tensor(p0, dtype=torch.float32), None
        g = torch.tensor(g, dtype=torch.float32)
        for _ in range(steps):
            d = g + wd * p
            buf = d.clone() if buf is None else buf * mom + d
            p = p - lr * buf
        return float(p), float(buf)

    expect_m = {}
    for mk, m in model.items():
        sv = engine.CONFIGS["c2f"]["solver"]["dis" if mk.startswith("dis_") else mk]
        trainable = {n for n, q in m.named_parameters() if q.requires_grad}
        for k, v in m.state_dict().items():
            c = synth.const_of(mk + "/" + k)
            if k in trainable and not k.startswith("cond_2"):
                bias = "bias" in k
                c, expect_m[(mk, k)] = replay(c, synth.const_of("grad/" + mk + "/" + k),
                                              sv["lr"] * (sv["bias_lr_factor"] if bias else 1.0) * sv["warmup_factor"],
                                              sv["wd_bias"] if bias else sv["wd"])
            assert bool((v == v.reshape(-1)[0]).all()), (mk, k)
            assert abs(float(v.reshape(-1)[0]) - c) <= 1e-6 * max(abs(c), 1e-3), (mk, k, float(v.reshape(-1)[0]), c)
    assert trainer.iteration == 3  # scheduler_dis_*['last_epoch']
    # momentum of the discriminators comes back; the reference file carries no optimizer for the other sub-models
    g = trainer.groups["dis_P3_CON"]
    assert not g.first and trainer.groups["backbone"].first
    for name, p, m in g._logical():
        e = expect_m[("dis_P3_CON", name)]
        assert abs(float(m.reshape(-1)[0]) - e) <= 1e-6 * max(abs(e), 1e-3) and bool((m == m.reshape(-1)[0]).all()), name
    # weights-only load (the reference's own call, load_opt_sch=False) leaves optimizer and iteration alone
    tr2 = engine.Trainer(engine.build_model(9, device="cpu"))
    tr2.load_checkpoint(path, load_opt_sch=False)
    assert tr2.iteration == 0 and tr2.groups["dis_P3_CON"].first


def test_split_rows2_backward_equals_two_slices():
    """ops.split_rows2 (the discriminators' source / target halves of a level): same values and gradient as x[:m], x[m:],
    also when only one half is used."""
    import torch
    from scan_amd import ops
    x = torch.randn(7, 3, requires_grad=True)
    a, b = ops.split_rows2(x, 4)
    (a.sum() * 2 + (b ** 2).sum()).backward()
    y = x.detach().clone().requires_grad_(True)
    (y[:4].sum() * 2 + (y[4:] ** 2).sum()).backward()
    assert torch.equal(x.grad, y.grad)
    x.grad = None
    a, b = ops.split_rows2(x, 4)
    assert torch.equal(a, x[:4]) and torch.equal(b, x[4:])
    b.sum().backward()
    assert torch.equal(x.grad, torch.cat([torch.zeros(4, 3), torch.ones(3, 3)]))


def test_seeded_multi_root_backward_equals_backward_of_the_weighted_sum():
    """engine.Trainer._backward_terms (round 6): the loss terms as roots of ONE autograd pass, each seeded with its weight, leave
    bit-identical gradients to (sum_i w_i * t_i).backward() -- on a graph with shared trunks, repeated weights and a term without
    a gradient (the zero `consistency_loss_gt` of a rank that sampled no target node)."""
    import torch
    from scan_amd import engine
    trainer = engine.Trainer(engine.build_model(9, device="cpu"))
    torch.manual_seed(0)
    w1, w2, w3 = (torch.randn(16, 16, requires_grad=True) for _ in range(3))
    x = torch.randn(8, 16)

    def terms():
        h = torch.tanh(x @ w1)                      # shared trunk
        a, b = (h @ w2).square().mean(), (h @ w3).abs().mean()
        c = (torch.relu(h @ w2) @ w3).sum() * 1e-3  # touches w2 and w3 again
        return [(a, 1.0), (b, 0.1), (c, 0.1), (torch.zeros(()), 1.0), (h.mean(), 1.0)]

    sum(w * t for t, w in terms()).backward()
    ref = [p.grad.clone() for p in (w1, w2, w3)]
    for p in (w1, w2, w3):
        p.grad = None
    trainer._backward_terms(terms())
    for p, r in zip((w1, w2, w3), ref):
        assert torch.equal(p.grad, r)


def test_surface_adopt_and_flatten_order():
    """scan_amd.surface (host side, no GPU): adopt() re-classes nn.Conv2d / nn.GroupNorm of every sub-model as the scan_amd.layers
    drop-ins without touching parameters or state_dict keys, and _flatten() puts per-level NCHW maps into the pyramid's row order
    (level, image, y, x) -- the order the reference concatenates in before its losses (rpn/fcos/loss.py:191-202) and the engine's
    PyramidShape uses."""
    import torch
    from torch import nn
    from scan_amd import engine, layers, ops, surface
    model = engine.build_model(9, device="cpu")
    keys = {k: list(m.state_dict().keys()) for k, m in model.items()}
    ptrs = {k: [p.data_ptr() for p in m.parameters()] for k, m in model.items()}
    surface.adopt(model)
    for k, m in model.items():
        assert list(m.state_dict().keys()) == keys[k] and [p.data_ptr() for p in m.parameters()] == ptrs[k]
        for sub in m.modules():
            assert type(sub) is not nn.Conv2d and type(sub) is not nn.GroupNorm
    assert isinstance(model["fcos"].head.cls_tower[0], layers.Conv2d) and isinstance(model["fcos"].head.cls_tower[1], layers.GroupNorm)
    assert model["fcos"].head.cls_tower[1].fuse_relu is False
    n, c, sizes = 2, 3, [(4, 6), (2, 3), (1, 2)]
    shape = ops.PyramidShape(n, sizes)
    levels = [torch.arange(n * c * h * w, dtype=torch.float32).reshape(n, c, h, w) + 1000 * l for l, (h, w) in enumerate(sizes)]
    rows = surface._flatten(levels)
    assert rows.shape == (shape.rows, c)
    for l, (h, w) in enumerate(sizes):
        for i in range(n):
            for y in range(h):
                for x in range(w):
                    r = shape.row_off[l] + (i * h + y) * w + x
                    assert torch.equal(rows[r], levels[l][i, :, y, x])
