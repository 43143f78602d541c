"""Checkpoint discovery and state_dict extraction, compatible with the reference's
utils/commons/ckpt_utils.py:17-95: highest-step `model_ckpt_steps_*.ckpt` in a directory (or an explicit file),
`checkpoint['state_dict'][model_name]` or flat `model_name.`-prefixed keys, shape-tolerant load when
strict=False, and `assert False` when nothing is found and force=True."""
import glob
import os
import re

import torch


def get_all_ckpts(work_dir, steps=None):
    pat = f"{work_dir}/model_ckpt_steps_*.ckpt" if steps is None else f"{work_dir}/model_ckpt_steps_{steps}.ckpt"
    return sorted(glob.glob(pat), key=lambda x: -int(re.findall(r".*steps\_(\d+)\.ckpt", x)[0]))


def get_last_checkpoint(work_dir, steps=None):
    paths = get_all_ckpts(work_dir, steps)
    if paths:
        return torch.load(paths[0], map_location="cpu", weights_only=False), paths[0]
    return None, None


def extract_state_dict(checkpoint, model_name="model"):
    sd = checkpoint["state_dict"]
    if len([k for k in sd.keys() if "." in k]) > 0:
        return {k[len(model_name) + 1:]: v for k, v in sd.items() if k.startswith(f"{model_name}.")}
    if "." not in model_name:
        return sd[model_name]
    base, rest = model_name.split(".")[0], model_name[len(model_name.split(".")[0]) + 1:]
    return {k[len(rest) + 1:]: v for k, v in sd[base].items() if k.startswith(f"{rest}.")}


def load_ckpt(cur_model, ckpt_base_dir, model_name="model", force=True, strict=True):
    if os.path.isfile(ckpt_base_dir):
        base_dir, ckpt_path = os.path.dirname(ckpt_base_dir), ckpt_base_dir
        checkpoint = torch.load(ckpt_base_dir, map_location="cpu", weights_only=False)
    else:
        base_dir = ckpt_base_dir
        checkpoint, ckpt_path = get_last_checkpoint(ckpt_base_dir)
    if checkpoint is None:
        msg = f"| ckpt not found in {base_dir}."
        if force:
            assert False, msg
        print(msg)
        return
    sd = extract_state_dict(checkpoint, model_name)
    if not strict:
        cur = cur_model.state_dict()
        for key in [k for k, v in sd.items() if k in cur and cur[k].shape != v.shape]:
            print("| Unmatched keys: ", key, cur[key].shape, sd[key].shape)
            del sd[key]
    cur_model.load_state_dict(sd, strict=strict)
    print(f"| load '{model_name}' from '{ckpt_path}'.")


def load_ckpt_emformer(cur_model, ckpt_base_dir, model_name="model", force=True, strict=True):
    """utils/commons/ckpt_utils.py:67-95: like load_ckpt but the checkpoint's state_dict is flat."""
    if os.path.isfile(ckpt_base_dir):
        checkpoint, ckpt_path = torch.load(ckpt_base_dir, map_location="cpu", weights_only=False), ckpt_base_dir
    else:
        checkpoint, ckpt_path = get_last_checkpoint(ckpt_base_dir)
    if checkpoint is None:
        msg = f"| ckpt not found in {ckpt_base_dir}."
        if force:
            assert False, msg
        print(msg)
        return
    sd = checkpoint["state_dict"]
    if not strict:
        cur = cur_model.state_dict()
        for key in [k for k, v in sd.items() if k in cur and cur[k].shape != v.shape]:
            del sd[key]
    cur_model.load_state_dict(sd, strict=strict)
    print(f"| load '{model_name}' from '{ckpt_path}'.")
