"""Checkpoint wire format of the reference's DetectronCheckpointer for the SCAN model dict
(reference fcos_core/utils/checkpoint.py:141-301 save, :417-492 load; suffix matching of
fcos_core/utils/model_serialization.py:10-80).

One ``.pth`` (torch.save of a dict) with the keys the reference writes:
``model_backbone``, ``model_fcos``, ``middle_head``, ``model_dis_P{3..7}_CON`` (+ ``iteration`` and any extra
keyword arguments), each a plain ``state_dict`` with the reference's parameter names, plus a ``last_checkpoint``
pointer file next to it.  Loading aligns names by longest matching suffix, so an ImageNet VGG16 file with keys
``features.N.weight`` fills ``body.features.N.weight`` and a DDP-saved file with ``module.`` prefixes loads too.
Weights are re-homed to channels-last (the layout the HIP kernels read) after loading.
"""
import os
from collections import OrderedDict

import torch

MODEL_KEYS = {"backbone": "model_backbone", "fcos": "model_fcos", "middle_head": "middle_head"}


def _ckpt_key(name):
    return MODEL_KEYS.get(name, "model_" + name)  # dis_P3_CON -> model_dis_P3_CON


def state_to_save(model, **extra):
    data = OrderedDict()
    for name, m in model.items():
        data[_ckpt_key(name)] = OrderedDict((k, v.detach().cpu().contiguous()) for k, v in m.state_dict().items())
    data.update(extra)
    return data


def save(model, save_dir, name, **extra):
    """reference Checkpointer.save: <save_dir>/<name>.pth + last_checkpoint tag file."""
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, "%s.pth" % name)
    torch.save(state_to_save(model, **extra), path)
    with open(os.path.join(save_dir, "last_checkpoint"), "w") as f:
        f.write(path)
    return path


def get_checkpoint_file(save_dir):
    try:
        with open(os.path.join(save_dir, "last_checkpoint")) as f:
            return f.read().strip()
    except IOError:
        return ""


def strip_prefix_if_present(state_dict, prefix="module."):
    keys = sorted(state_dict.keys())
    if not keys or not all(k.startswith(prefix) for k in keys):
        return state_dict
    return OrderedDict((k.replace(prefix, ""), v) for k, v in state_dict.items())


def align_and_update_state_dicts(model_state_dict, loaded_state_dict):
    """For every model key pick the loaded key that is its longest suffix (model_serialization.py:10-58)."""
    matched = {}
    loaded_keys = sorted(loaded_state_dict.keys())
    for key in sorted(model_state_dict.keys()):
        best = max((j for j in loaded_keys if key.endswith(j)), key=len, default=None)
        if best is not None:
            model_state_dict[key] = loaded_state_dict[best]
            matched[key] = best
    return matched


def load_state_dict(module, loaded_state_dict):
    sd = module.state_dict()
    matched = align_and_update_state_dicts(sd, strip_prefix_if_present(loaded_state_dict))
    module.load_state_dict(sd)
    for p in module.parameters():
        if p.dim() == 4 and not p.data.is_contiguous(memory_format=torch.channels_last):
            p.data = p.data.contiguous(memory_format=torch.channels_last)
    return matched


def load(model, path, load_dis=True):
    """reference DetectronCheckpointer.load(f, load_dis=...): returns the remaining entries (e.g. iteration).
    A file without the ``model_*`` keys is treated as a bare backbone state_dict (ImageNet VGG16)."""
    ckpt = torch.load(path, map_location="cpu")
    if "model_backbone" not in ckpt:
        sd = ckpt.get("state_dict", ckpt.get("model", ckpt))
        load_state_dict(model["backbone"], sd)
        return {}
    used = set()
    for name, m in model.items():
        if name.startswith("dis_") and not load_dis:
            continue
        key = _ckpt_key(name)
        if key in ckpt:
            load_state_dict(m, ckpt[key])
            used.add(key)
    return {k: v for k, v in ckpt.items() if k not in used}
