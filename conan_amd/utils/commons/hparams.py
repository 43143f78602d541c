"""yaml hyper-parameter chains, compatible with the reference's utils/commons/hparams.py:25-131:
depth-first `base_config` inheritance with './' relative resolution (:51-72), `checkpoints/<exp>/config.yaml`
overriding unless reset (:76-88), `k=v,k2.k3=v` command-line overrides with type coercion (:93-106) and a
process-global `hparams` dict (:8, :121-123).  Inference-only: nothing is written to disk."""
import argparse
import ast
import os

import yaml

hparams = {}


def override_config(old_config, new_config):
    for k, v in new_config.items():
        if isinstance(v, dict) and k in old_config and isinstance(old_config[k], dict):
            override_config(old_config[k], v)
        else:
            old_config[k] = v


def _load_chain(config_fn, loaded, chain):
    if not os.path.exists(config_fn):
        return {}
    with open(config_fn) as f:
        hp = yaml.safe_load(f) or {}
    loaded.add(config_fn)
    out = {}
    if "base_config" in hp:
        if not isinstance(hp["base_config"], list):
            hp["base_config"] = [hp["base_config"]]     # kept as a list in the result, like hparams.py:60-61
        for c in hp["base_config"]:
            if c.startswith("."):
                c = os.path.normpath(f"{os.path.dirname(config_fn)}/{c}")
            if c not in loaded:
                override_config(out, _load_chain(c, loaded, chain))
    override_config(out, hp)
    chain.append(config_fn)
    return out


def _coerce(old, v):
    v = v.strip("'\" ")
    if v in ("True", "False") or isinstance(old, (bool, list, dict)):
        if isinstance(old, list):
            v = v.replace(" ", ",")
        return ast.literal_eval(v)
    return type(old)(v) if old is not None else v


def set_hparams(config="", exp_name="", hparams_str="", print_hparams=True, global_hparams=True, reset=False):
    if config == "" and exp_name == "":
        p = argparse.ArgumentParser(description="")
        p.add_argument("--config", type=str, default="")
        p.add_argument("--exp_name", type=str, default="")
        p.add_argument("-hp", "--hparams", type=str, default="")
        p.add_argument("--infer", action="store_true")
        p.add_argument("--validate", action="store_true")
        p.add_argument("--reset", action="store_true")
        p.add_argument("--remove", action="store_true")
        p.add_argument("--debug", action="store_true")
        args, _ = p.parse_known_args()
        config, exp_name, hparams_str, reset = args.config, args.exp_name, args.hparams, args.reset
        flags = dict(infer=args.infer, debug=args.debug, validate=args.validate)
    else:
        flags = dict(infer=False, debug=False, validate=False)
    assert config != "" or exp_name != ""
    if config != "":
        assert os.path.exists(config), config
    chain, saved = [], {}
    work_dir = f"checkpoints/{exp_name}" if exp_name != "" else ""
    if work_dir and os.path.exists(f"{work_dir}/config.yaml"):
        with open(f"{work_dir}/config.yaml") as f:
            saved.update(yaml.safe_load(f) or {})
    hp = {}
    if config != "":
        hp.update(_load_chain(config, set(), chain))
    if not reset:
        hp.update(saved)
    hp["work_dir"] = work_dir
    if hparams_str != "":
        for item in hparams_str.split(","):
            k, v = item.split("=")
            node = hp
            for k_ in k.split(".")[:-1]:
                node = node[k_]
            k = k.split(".")[-1]
            node[k] = _coerce(node.get(k), v)
    hp.update(flags)
    hp["exp_name"] = exp_name
    if global_hparams:
        hparams.clear()
        hparams.update(hp)
    if print_hparams and global_hparams:
        print("| Hparams chains: ", chain)
    return hp
