"""Input pipeline: the oracle restatement against the fixture the reference's own transforms
produced (tests/golden/pipeline.npz) and against PIL itself; the device kernels (scan_amd/data.py, csrc/imgproc.hip)
byte-exact on the resized uint8 image and bit-exact on the normalised, collated batch."""
import os

import numpy as np
import pytest
import torch

from oracle import pipeline_ref as pr
from scan_amd import config, synth

# must equal oracle/make_golden.py PIPE_CASES
CASES = [("up", [(97, 131), (80, 120)], 0.0, 160, 240), ("down_flip", [(150, 301), (128, 256)], 1.0, 100, 180),
         ("same", [(64, 96)], 0.0, 64, 333)]
CFG = config.load("c2f")


def _inputs(sizes):
    return [(synth.synth_u8_image(h, w, 777 + i), synth.synth_targets(1, h, w, 8, 5, 888 + i)[0]) for i, (h, w) in
            enumerate(sizes)]


def test_get_size_matches_reference_table(gold_dir):
    from scan_amd import data
    tab = np.load(os.path.join(gold_dir, "pipeline.npz"))["get_size_table"]
    assert len(tab) == 32
    for w, h, mn, mx, oh, ow in tab.tolist():
        assert pr.get_size((w, h), mn, mx) == (oh, ow)
        assert data.Resize(mn, mx).get_size((w, h)) == (oh, ow)


def test_coefficient_tables_product_equals_oracle():
    from scan_amd import data
    for a, b in [(131, 216), (301, 180), (1024, 800), (2048, 1600), (7, 3), (3, 7), (1, 5)]:
        bounds, kk = pr.coeffs(a, b)
        pb, pc, k = data.bilinear_tables(a, b)
        assert pb.tolist() == [list(x) for x in bounds] and pc.tolist() == kk and k == len(kk[0])
        assert all(sum(r) in range((1 << 22) - 8, (1 << 22) + 9) for r in kk)  # weights sum to 1.0 in fixed point


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_pipeline_vs_reference_fixture(gold_dir, case):
    name, sizes, flip, mn, mx = case
    g = np.load(os.path.join(gold_dir, "pipeline.npz"))
    tensors = []
    for i, (img, (boxes, labels)) in enumerate(_inputs(sizes)):
        h, w = img.shape[:2]
        oh, ow = pr.get_size((w, h), mn, mx)
        r = pr.resize(img, oh, ow)
        assert np.array_equal(r, g["%s_resized_%d" % (name, i)])  # byte-exact
        b = pr.resize_boxes(boxes.numpy(), (w, h), (ow, oh))
        if flip:
            r = r[:, ::-1]
            b = pr.hflip_boxes(b, ow)
        np.testing.assert_allclose(b, g["%s_boxes_%d" % (name, i)], rtol=0, atol=1e-4)
        assert tuple(g["%s_boxsize_%d" % (name, i)]) == (ow, oh)
        tensors.append(pr.to_tensor_normalize(r, CFG.INPUT.PIXEL_MEAN, CFG.INPUT.PIXEL_STD, CFG.INPUT.TO_BGR255))
    batch, sizes_out = pr.collate(tensors, CFG.DATALOADER.SIZE_DIVISIBILITY)
    assert batch.shape == g["%s_batch" % name].shape
    assert np.array_equal(batch, g["%s_batch" % name])  # bit-exact fp32
    assert [list(s) for s in sizes_out] == g["%s_image_sizes" % name].tolist()


def test_oracle_resize_vs_pil_random_sizes():
    Image = pytest.importorskip("PIL.Image")
    rs = np.random.RandomState(5)
    for _ in range(12):
        h, w, oh, ow = rs.randint(5, 90), rs.randint(5, 90), rs.randint(3, 120), rs.randint(3, 120)
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(pr.resize(img, oh, ow), ref), (h, w, oh, ow)


# ----------------------------------------------------------------------------- device kernels
@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_device_pipeline_matches_reference_fixture(device, gold_dir, case):
    """the reference's Compose([Resize, RandomHorizontalFlip, ToTensor, Normalize]) + BatchCollator(32) against
    scan_amd.data's classes of the same names: resized bytes equal, boxes 1e-4, collated batch bit-exact -- in the
    reference's NCHW layout and in the NHWC4 rows the first convolution reads."""
    from scan_amd import data
    name, sizes, flip, mn, mx = case
    g = np.load(os.path.join(gold_dir, "pipeline.npz"))
    tf = data.Compose([data.Resize(mn, mx), data.RandomHorizontalFlip(flip), data.ToTensor(),
                       data.Normalize(CFG.INPUT.PIXEL_MEAN, CFG.INPUT.PIXEL_STD, CFG.INPUT.TO_BGR255)])
    batch = []
    for i, (img, (boxes, labels)) in enumerate(_inputs(sizes)):
        u8 = data.U8Image.from_numpy(img, device)
        rz, _ = data.Resize(mn, mx)(u8, None)
        assert np.array_equal(rz.data.cpu().numpy(), g["%s_resized_%d" % (name, i)])
        nimg, (b, l) = tf(u8, (boxes.to(device), labels.to(device)))
        np.testing.assert_allclose(b.cpu().numpy(), g["%s_boxes_%d" % (name, i)], rtol=0, atol=1e-4)
        batch.append((nimg, (b, l), i))
    il, targets, ids = data.BatchCollator(CFG.DATALOADER.SIZE_DIVISIBILITY)(batch)
    ref = g["%s_batch" % name]
    assert tuple(il.tensors.shape) == ref.shape and list(ids) == list(range(len(sizes)))
    assert np.array_equal(il.tensors.cpu().numpy(), ref)
    assert [list(s) for s in il.image_sizes] == g["%s_image_sizes" % name].tolist()
    n, _, hp, wp = ref.shape
    rows = il.rows.cpu().numpy().reshape(n, hp, wp, 4)
    assert np.array_equal(rows[..., :3].transpose(0, 3, 1, 2), ref) and not rows[..., 3].any()
    # a single image's tensor (what the reference's transform returns before collation)
    t0 = batch[0][0].tensor().cpu().numpy()
    assert np.array_equal(t0, ref[0][:, :t0.shape[1], :t0.shape[2]])


@pytest.mark.gpu
def test_device_resize_full_size_vs_pil_and_properties(device):
    """BASELINE frame size: 1024x2048 -> 800x1600 (the yaml's MIN_SIZE_TEST) against PIL when importable, plus
    size-independent properties: same-size resize is the identity, a constant image stays constant, resize commutes
    with a horizontal flip (the filter is symmetric)."""
    from scan_amd import data
    img = synth.synth_u8_image(1024, 2048, 1)
    u8 = torch.from_numpy(img).to(device)
    out = data.resize_u8(u8, 800, 1600)
    try:
        from PIL import Image
        ref = np.asarray(Image.fromarray(img).resize((1600, 800), Image.BILINEAR))
        assert np.array_equal(out.cpu().numpy(), ref)
    except ImportError:
        pass
    assert torch.equal(data.resize_u8(u8, 1024, 2048), u8)
    const = torch.full((333, 500, 3), 77, dtype=torch.uint8, device=device)
    assert bool((data.resize_u8(const, 800, 1201) == 77).all())
    a = data.resize_u8(torch.flip(u8, dims=(1,)), 800, 1600)
    assert torch.equal(a, torch.flip(out, dims=(1,)))


@pytest.mark.gpu
def test_collated_rows_feed_the_detector(device):
    """the NHWC4 batch written by the collator gives the same detections as the NCHW tensor path."""
    from scan_amd import data, engine
    tf = data.build_transforms(config.load("c2f", ["INPUT.MIN_SIZE_TEST", 128, "INPUT.MAX_SIZE_TEST", 256]), is_train=False)
    batch = []
    for i in range(2):
        nimg, _ = tf(data.U8Image.from_numpy(synth.synth_u8_image(150 + 10 * i, 301, 40 + i), device), None)
        batch.append((nimg, None, i))
    il, _, _ = data.BatchCollator(32)(batch)
    model = engine.build_model(9, device=device)
    engine.load_state_dicts(model, synth.shifted_state_dicts(9))
    a = engine.inference(model, il)
    b = engine.inference(model, engine.ImageList(il.tensors, il.image_sizes))
    for (b1, s1, l1), (b2, s2, l2) in zip(a, b):
        assert torch.equal(b1, b2) and torch.equal(s1, s2) and torch.equal(l1, l2) and len(s1) > 0
