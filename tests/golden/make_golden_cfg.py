"""Golden fixture: the configuration dicts the reference's `configs/config.py::get_config` builds for the three experiment
types and a few categories (JSON-able part: everything except the torch device and the nested object / pointnet tables,
which are compared through the keys derived from them).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_cfg.py [--ref /root/reference]
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))

from tests.golden.make_golden import import_reference  # noqa: E402
from tests.golden.make_golden_train import ref_cfg  # noqa: E402

CASES = [("config_track.yml", "1", "obj_info_nocs.yml"), ("config_track.yml", "5", "obj_info_nocs.yml"),
         ("config_track.yml", "drawers", "obj_info_sapien.yml"), ("config_track.yml", "glasses", "obj_info_sapien.yml"),
         ("config_coordnet.yml", "3", "obj_info_nocs.yml"), ("config_rotnet.yml", "laptop", "obj_info_sapien.yml")]


def jsonable(cfg):
    out = {}
    for k, v in cfg.items():
        if k in ("device", "obj", "root_dset", "experiment_dir", "num_expr"):
            continue
        out[k] = v
    # dataset bookkeeping of the object tables (instance ids, splits, augmentation axes) is outside the path
    out["obj_info"] = {k: v for k, v in out["obj_info"].items()
                       if k not in ("bad_ins", "test_list", "train_list", "template", "exemplar", "augment_idx", "parts_map")}
    return json.loads(json.dumps(out, default=str))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    import_reference(args.ref)
    out = {}
    for config, cat, objcfg in CASES:
        out[f"{config}|{cat}|{objcfg}"] = jsonable(ref_cfg(config, cat, objcfg))
    with open(HERE / "cfg_reference.json", "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", HERE / "cfg_reference.json", len(out), "configs")


if __name__ == "__main__":
    main()
