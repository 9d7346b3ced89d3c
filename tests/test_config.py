"""Config boundary (scan_amd/config.py) against the reference's merged cfg (tests/golden/cfg_*.json, written by
oracle/make_golden.py gen_cfg from config/defaults.py + configs/scan/*.yaml) and the LR schedule against the
learning rates the reference's WarmupMultiStepLR produced in the trajectory fixture."""
import json
import os

import pytest

from scan_amd import config


@pytest.mark.parametrize("name", ["c2f", "s2c", "k2c"])
def test_shipped_yaml_equals_reference_merged_cfg(gold_dir, name):
    gold = json.load(open(os.path.join(gold_dir, "cfg_%s.json" % name)))["cfg"]
    assert config.hot_path_view(config.load(name)) == gold


REF_YAML = {"c2f": "scan_vgg16_cityscapace_to_foggy.yaml", "s2c": "scan_vgg16_sim10k_to_cityscapes.yaml",
            "k2c": "scan_vgg16_kitti_to_cityscapes.yaml"}


@pytest.mark.parametrize("name", ["c2f", "s2c", "k2c"])
def test_reference_yaml_parses_to_the_same_cfg(gold_dir, name):
    """a user's reference yaml (incl. the mis-indented S2C file yacs itself rejects) loads through the same parser."""
    path = os.path.join("/root/reference/configs/scan", REF_YAML[name])
    if not os.path.exists(path):
        pytest.skip("reference checkout not present (GPU box)")
    gold = json.load(open(os.path.join(gold_dir, "cfg_%s.json" % name)))["cfg"]
    assert config.hot_path_view(config.load(path)) == gold


def test_merge_from_list_and_coercion():
    cfg = config.load("c2f", ["SOLVER.DIS.STEPS", "(3, 9)", "MODEL.MIDDLE_HEAD.TRANSFER_CFG", "(None,)",
                              "SOLVER.FCOS.BASE_LR", 1, "TEST.MODE", "light"])
    s = config.settings(cfg)
    assert s["solver"]["dis"]["steps"] == (3, 9) and s["transfer_cfg"] == (None,)
    assert s["solver"]["fcos"]["lr"] == 1.0 and isinstance(cfg.SOLVER.FCOS.BASE_LR, float)
    assert s["test_mode"] == "light"
    with pytest.raises(ValueError):
        config.settings(config.load("c2f", ["MODEL.MIDDLE_HEAD.USE_RNN", "GRU"]))
    with pytest.raises(ValueError):
        config.load("c2f", ["SOLVER.MAX_ITER"])


def test_engine_configs_come_from_the_yaml():
    from scan_amd import engine
    assert engine.CONFIGS["c2f"]["num_classes"] == 9 and engine.CONFIGS["c2f"]["test_mode"] == "precision"
    assert engine.CONFIGS["s2c"]["solver"]["backbone"]["steps"] == (60000, 70000)
    assert engine.CONFIGS["k2c"]["solver"]["dis"]["steps"] == (8000, 15000) and engine.CONFIGS["k2c"]["max_iter"] == 25000
    assert engine.CONFIGS["k2c_r50"]["conv_body"] == "R-50-FPN-RETINANET"
    assert engine.CONFIGS["s2c"]["transfer_cfg"] == (None,)


def test_lr_schedule_matches_reference_scheduler(gold_dir):
    """per sub-model learning rates of weights and biases at every iteration of the trajectory fixture: constant and
    linear warm-up, one and two milestones inside the run, per-group base lr (reference solver/build.py:7-84,
    solver/lr_scheduler.py:39-52)."""
    from scan_amd import engine
    gold = json.load(open(os.path.join(gold_dir, "traj_128x256.json")))
    opts = [tuple(x) if isinstance(x, list) else x for x in gold["opts"]]
    s = config.settings(config.load("c2f", opts))
    for it, lrs in enumerate(gold["lr"]):
        for k, (lw, lb) in lrs.items():
            sv = s["solver"]["dis" if k.startswith("dis_") else k]
            f = engine.warmup_factor(it, sv["warmup_iters"], sv["warmup_factor"], sv["steps"], sv["gamma"],
                                     sv["warmup_method"])
            assert abs(sv["lr"] * f - lw) <= 1e-12 + 1e-9 * lw, (it, k, sv["lr"] * f, lw)
            assert abs(sv["lr"] * sv["bias_lr_factor"] * f - lb) <= 1e-12 + 1e-9 * lb, (it, k)
    # the shipped yamls: decay milestones of each experiment
    for name, (s1, s2) in (("c2f", (60000, 80000)), ("s2c", (60000, 70000)), ("k2c", (8000, 15000))):
        sv = engine.CONFIGS[name]["solver"]["backbone"]
        fac = lambda i: engine.warmup_factor(i, sv["warmup_iters"], sv["warmup_factor"], sv["steps"], sv["gamma"],
                                             sv["warmup_method"])
        assert fac(0) == pytest.approx(1 / 3) and fac(999) == pytest.approx(1 / 3) and fac(1000) == 1.0
        assert fac(s1 - 1) == 1.0 and fac(s1) == pytest.approx(0.1) and fac(s2) == pytest.approx(0.01)
