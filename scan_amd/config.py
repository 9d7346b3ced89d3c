"""Configuration boundary of the hot path: the yaml keys the SCAN configs set and this path reads.

The reference drives everything from a yacs tree (fcos_core/config/defaults.py, 712 lines) merged with one of
configs/scan/*.yaml and trailing ``KEY VALUE`` pairs (tools/train_net_da.py:680-707).  Only a small part of that
tree reaches the hot path; this module holds exactly that part:

* ``DEFAULTS``      the reference's default value of every key the path reads (file:line cited per block),
* ``Cfg``           attribute-style nested dict with ``merge_from_file`` / ``merge_from_list`` (same coercion rules
                    as yacs for the value types these keys use: tuples stay tuples, ``"('NODES', 'ADJ')"`` strings
                    are literal-evaluated); keys outside DEFAULTS are kept but ignored, so a user's full reference
                    yaml (datasets, output dir, ...) loads unchanged,
* ``load(name)``    the three shipped experiment definitions (scan_amd/configs/*.yaml: c2f, s2c, k2c) and the
                    R-50 variant of K2C (BASELINE.json configs[3]),
* ``settings(cfg)`` the flat view engine.build_model / engine.Trainer / the post-processor consume.

tests/test_config.py checks ``hot_path_view`` of the shipped files against tests/golden/cfg_*.json, which
oracle/make_golden.py wrote from the reference's own merged cfg.
"""
import ast
import copy
import os
import re

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
CONFIG_DIR = os.path.join(_HERE, "configs")

_SOLVER_GROUP = {"BASE_LR": 0.005, "BIAS_LR_FACTOR": 2, "GAMMA": 0.1, "STEPS": (30000,), "WARMUP_FACTOR": 1.0 / 3,
                 "WARMUP_ITERS": 500, "WARMUP_METHOD": "linear"}  # defaults.py:537-565, 666-673

DEFAULTS = {
    "INPUT": {  # defaults.py:45-61
        "MIN_SIZE_TRAIN": (800,), "MIN_SIZE_RANGE_TRAIN": (-1, -1), "MAX_SIZE_TRAIN": 1333, "MIN_SIZE_TEST": 800,
        "MAX_SIZE_TEST": 1333, "PIXEL_MEAN": [102.9801, 115.9465, 122.7717], "PIXEL_STD": [1.0, 1.0, 1.0],
        "TO_BGR255": True,
    },
    "DATALOADER": {"SIZE_DIVISIBILITY": 0},  # defaults.py:85
    "MODEL": {
        "BACKBONE": {"CONV_BODY": "R-50-C4", "FREEZE_CONV_BODY_AT": 2},  # defaults.py:101-104
        "RESNETS": {"BACKBONE_OUT_CHANNELS": 1024},  # defaults.py:283 (256 * 4)
        "RETINANET": {"USE_C5": True},  # defaults.py:431
        "FCOS": {  # defaults.py:336-351, 663, 688-689
            "NUM_CLASSES": 81, "FPN_STRIDES": [8, 16, 32, 64, 128], "PRIOR_PROB": 0.01, "INFERENCE_TH": 0.05,
            "NMS_TH": 0.6, "PRE_NMS_TOP_N": 1000, "LOSS_ALPHA": 0.25, "LOSS_GAMMA": 2.0, "NUM_CONVS_REG": 4,
            "NUM_CONVS_CLS": 4, "REG_CTR_ON": False,
        },
        "ADV": {  # defaults.py:356-411, 589-616
            "USE_DIS_CON": False, "CON_DIS_LAMBDA": 0.1, "CON_WITH_GA": False, "CON_FUSUIN_CFG": "concat",
            "GRL_APPLIED_DOMAIN": "both", "PATCH_STRIDE": None,
            **{"USE_DIS_%s_CON" % l: False for l in ("P3", "P4", "P5", "P6", "P7")},
            **{"CON_NUM_SHARED_CONV_%s" % l: 4 for l in ("P3", "P4", "P5", "P6", "P7")},
            **{"GRL_WEIGHT_%s" % l: 0.1 for l in ("P3", "P4", "P5", "P6", "P7")},
        },
        "MIDDLE_HEAD": {  # defaults.py:619-712
            "CONDGRAPH_ON": False, "NUM_CONVS_IN": 1, "NUM_CONVS_OUT": 1, "CAT_ACT_MAP": True, "IN_NORM": "GN",
            "COSINE_UPDATE_ON": False, "PROTO_ITER": 1, "USE_RNN": None, "PROTO_WITH_BG": True,
            "COND_WITH_BIAS": False, "PROTO_CHANNEL": 256, "COND_HIDDEN_CHANNEL": 512, "TRANSFER_CFG": (None,),
            "GCN_SELF_TRAINING": False, "TARGET_SAMPLING_CFG": "score_threshold", "DBSCAN_EPS": 3, "DBSCAN_THR": 0.05,
            "ACT_LOSS": None, "ACT_LOSS_WEIGHT": 1.0, "GCN_LOSS_WEIGHT": 1.0, "CON_LOSS_WEIGHT": 1.0,
            "GCN_LOSS_WEIGHT_TG": 1.0, "GLOBAL_GCN": False,
        },
    },
    "TEST": {"MODE": "common", "DETECTIONS_PER_IMG": 100, "IMS_PER_BATCH": 4},  # defaults.py:576-578, 691
    "SOLVER": {  # defaults.py:513-565, 655-682
        "MAX_ITER": 40000, "MOMENTUM": 0.9, "WEIGHT_DECAY": 0.0005, "WEIGHT_DECAY_BIAS": 0, "IMS_PER_BATCH": 16,
        "CHECKPOINT_PERIOD": 2500, "INITIAL_AP50": 10, "VAL_ITER": 250, "VAL_TYPE": "AP50", "ADAPT_VAL_ON": True,
        "BACKBONE": dict(_SOLVER_GROUP), "FCOS": dict(_SOLVER_GROUP), "DIS": dict(_SOLVER_GROUP),
        "MIDDLE_HEAD": dict(_SOLVER_GROUP),
    },
}


class Cfg(dict):
    """Nested dict with attribute access (what the path needs of yacs.config.CfgNode)."""

    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = Cfg(v) if isinstance(v, dict) else copy.deepcopy(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return Cfg(self)

    @staticmethod
    def _coerce(new, old):
        """yacs semantics for the value types used here: strings that look like literals are evaluated
        ("('NODES', 'ADJ')" -> tuple), a list replacing a tuple becomes a tuple and vice versa, an int may
        replace a float."""
        if isinstance(new, str):
            try:
                new = ast.literal_eval(new)
            except (ValueError, SyntaxError):
                pass
        if isinstance(old, tuple) and isinstance(new, list):
            new = tuple(new)
        elif isinstance(old, list) and isinstance(new, tuple):
            new = list(new)
        elif isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
            new = float(new)
        return new

    def _merge(self, d):
        for k, v in d.items():
            if isinstance(v, dict):
                if not isinstance(self.get(k), Cfg):
                    self[k] = Cfg()
                self[k]._merge(v)
            else:
                self[k] = self._coerce(v, self.get(k))

    def merge_from_file(self, path):
        with open(path) as f:
            text = f.read()
        self._merge(parse_yaml(text))
        return self

    def merge_from_list(self, lst):
        """trailing ``KEY VALUE`` pairs of the reference CLI (tools/train_net_da.py:680-685,706)."""
        if len(lst) % 2:
            raise ValueError("merge_from_list needs KEY VALUE pairs")
        for key, val in zip(lst[0::2], lst[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if not isinstance(node.get(p), Cfg):
                    node[p] = Cfg()
                node = node[p]
            node[parts[-1]] = self._coerce(val, node.get(parts[-1]))
        return self


def parse_yaml(text):
    """yaml.safe_load, tolerant of the one malformed line the reference ships: configs/scan/
    scan_vgg16_sim10k_to_cityscapes.yaml:5 indents ``WEIGHT`` by three spaces inside a two-space mapping (not valid
    YAML).  A key line whose indent is one more than its predecessor's sibling level is pulled back."""
    try:
        return yaml.safe_load(text) or {}
    except yaml.YAMLError:
        lines = text.split("\n")
        fixed, prev = [], None
        for ln in lines:
            m = re.match(r"^( +)([A-Za-z_][A-Za-z0-9_]*):", ln)
            if m and prev is not None and len(m.group(1)) == prev + 1:
                ln = ln[1:]
                m = re.match(r"^( +)", ln)
            if m and ln.strip() and not ln.strip().startswith("#"):
                prev = len(m.group(1))
            elif ln.strip() and not ln.strip().startswith("#") and not ln.startswith(" "):
                prev = 0
            fixed.append(ln)
        return yaml.safe_load("\n".join(fixed)) or {}


def defaults():
    return Cfg(DEFAULTS)


SHIPPED = {"c2f": "c2f.yaml", "s2c": "s2c.yaml", "k2c": "k2c.yaml"}
# BASELINE.json configs[3]: the K2C experiment on the R-50-FPN-RETINANET body (the reference's own ResNet yamls,
# configs/epm/*R_101*, set these two keys)
R50_OVERRIDE = ["MODEL.BACKBONE.CONV_BODY", "R-50-FPN-RETINANET", "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 256]


def load(name_or_path, opts=()):
    """``c2f`` / ``s2c`` / ``k2c`` / ``k2c_r50`` (shipped) or a path to a reference-style yaml; ``opts`` = KEY VALUE list."""
    cfg = defaults()
    extra = []
    if name_or_path == "k2c_r50":
        name_or_path, extra = "k2c", list(R50_OVERRIDE)
    path = os.path.join(CONFIG_DIR, SHIPPED[name_or_path]) if name_or_path in SHIPPED else name_or_path
    cfg.merge_from_file(path)
    cfg.merge_from_list(extra + list(opts))
    return cfg


def hot_path_view(cfg):
    """the DEFAULTS-shaped part of a cfg as plain python (tuples as lists), for comparison with the golden JSON."""
    def pick(node, ref):
        out = {}
        for k, v in ref.items():
            if isinstance(v, dict):
                out[k] = pick(node.get(k, {}), v)
            else:
                x = node.get(k, v)
                out[k] = list(x) if isinstance(x, (tuple, list)) else x
        return out
    return pick(cfg, DEFAULTS)


_GROUP_OF = {"backbone": "BACKBONE", "fcos": "FCOS", "middle_head": "MIDDLE_HEAD"}  # dis_* -> DIS (solver/build.py:7-43)


def solver_group(cfg, sub_model):
    """SGD + WarmupMultiStepLR settings of one sub-model (reference solver/build.py:7-84; tools/train_net_da.py builds
    the discriminators' optimizers with name='discriminator')."""
    g = cfg.SOLVER[_GROUP_OF.get(sub_model, "DIS")]
    return dict(lr=float(g.BASE_LR), bias_lr_factor=float(g.BIAS_LR_FACTOR), gamma=float(g.GAMMA),
                steps=tuple(int(s) for s in g.STEPS), warmup_factor=float(g.WARMUP_FACTOR),
                warmup_iters=int(g.WARMUP_ITERS), warmup_method=str(g.WARMUP_METHOD),
                wd=float(cfg.SOLVER.WEIGHT_DECAY), wd_bias=float(cfg.SOLVER.WEIGHT_DECAY_BIAS),
                momentum=float(cfg.SOLVER.MOMENTUM))


LEVELS = ("P3", "P4", "P5", "P6", "P7")


def settings(cfg):
    """Flat view for engine.build_model / Trainer; raises on anything this path does not build."""
    M, A, H, F = cfg.MODEL, cfg.MODEL.ADV, cfg.MODEL.MIDDLE_HEAD, cfg.MODEL.FCOS
    if not (H.CONDGRAPH_ON and H.USE_RNN == "RNN" and H.COSINE_UPDATE_ON and H.PROTO_WITH_BG and H.GLOBAL_GCN
            and H.ACT_LOSS == "softmaxFL" and H.CAT_ACT_MAP and H.IN_NORM == "GN" and not H.COND_WITH_BIAS
            and not H.GCN_SELF_TRAINING and H.TARGET_SAMPLING_CFG == "dbscan"):
        raise ValueError("MODEL.MIDDLE_HEAD: only the condgraph setting of configs/scan/*.yaml is built "
                         "(RNN paradigm, cosine update, bg prototype, global GCN, softmaxFL, GN, dbscan sampling)")
    if not (A.USE_DIS_CON and A.CON_FUSUIN_CFG == "concat" and A.GRL_APPLIED_DOMAIN == "both" and not A.CON_WITH_GA
            and A.PATCH_STRIDE is None and all(A["USE_DIS_%s_CON" % l] for l in LEVELS)):
        raise ValueError("MODEL.ADV: only the CKA discriminators of configs/scan/*.yaml are built "
                         "(USE_DIS_CON on P3..P7, 'concat' fusion, GRL on both domains)")
    if not F.REG_CTR_ON or M.RETINANET.USE_C5:
        raise ValueError("MODEL.FCOS.REG_CTR_ON True and MODEL.RETINANET.USE_C5 False are what is built")
    return dict(
        num_classes=int(F.NUM_CLASSES), test_mode=str(cfg.TEST.MODE), transfer_cfg=tuple(H.TRANSFER_CFG),
        conv_body=str(M.BACKBONE.CONV_BODY), proto_iter=int(H.PROTO_ITER), num_convs_in=int(H.NUM_CONVS_IN),
        num_convs_out=int(H.NUM_CONVS_OUT), dbscan_eps=H.DBSCAN_EPS, dbscan_thr=float(H.DBSCAN_THR),
        act_loss_weight=float(H.ACT_LOSS_WEIGHT), gcn_loss_weight=float(H.GCN_LOSS_WEIGHT),
        con_loss_weight=float(H.CON_LOSS_WEIGHT), gcn_loss_weight_tg=float(H.GCN_LOSS_WEIGHT_TG),
        num_convs_cls=int(F.NUM_CONVS_CLS), num_convs_reg=int(F.NUM_CONVS_REG), prior_prob=float(F.PRIOR_PROB),
        loss_gamma=float(F.LOSS_GAMMA), loss_alpha=float(F.LOSS_ALPHA), fpn_strides=tuple(F.FPN_STRIDES),
        inference_th=float(F.INFERENCE_TH), pre_nms_top_n=int(F.PRE_NMS_TOP_N), nms_th=float(F.NMS_TH),
        detections_per_img=int(cfg.TEST.DETECTIONS_PER_IMG), test_ims_per_batch=int(cfg.TEST.IMS_PER_BATCH),
        con_dis_lambda=float(A.CON_DIS_LAMBDA),
        dis_num_convs={l: int(A["CON_NUM_SHARED_CONV_%s" % l]) for l in LEVELS},
        grl_weight={l: float(A["GRL_WEIGHT_%s" % l]) for l in LEVELS},
        size_divisibility=int(cfg.DATALOADER.SIZE_DIVISIBILITY), ims_per_batch=int(cfg.SOLVER.IMS_PER_BATCH),
        initial_ap50=float(cfg.SOLVER.INITIAL_AP50), max_iter=int(cfg.SOLVER.MAX_ITER),
        val_iter=int(cfg.SOLVER.VAL_ITER), val_type=str(cfg.SOLVER.VAL_TYPE), adapt_val_on=bool(cfg.SOLVER.ADAPT_VAL_ON),
        solver={k: solver_group(cfg, k) for k in ("backbone", "fcos", "middle_head", "dis")},
    )
