"""Golden fixture G16: checkpoints WRITTEN BY THE REFERENCE's `Trainer.save` (network/trainer.py:196-210), read by this
repository's `Trainer.resume` (captra_amd/trainer.py; reference trainer.py:147-194: the CoordNet experiment's `net.*` keys land
under `npcs_net.*`, the RotationNet experiment's load as they are).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_checkpoint.py [--ref /root/reference]

The released checkpoints (README.md:236-254) are not in the container, so the reference's own two training experiments
(`canon_coord` / `rot`, config_coordnet.yml / config_rotnet.yml, bottle) are built with seeded weights
(tests/weights.make_state_dict, seeds below), saved by the reference's code into a scratch directory -- model, epoch, iteration
and the Adam state dict -- and resumed HERE, in the same process, by captra_amd's Trainer under the tracking config.  Written:
`ref_checkpoint.json` = the files' structure (top-level keys, epoch, iteration, optimizer keys), the ordered key list of each
`model` entry, and per tensor the sha256 of what captra_amd's model holds after `resume()` -- asserted equal, tensor by tensor,
to what the reference's model held when it saved.  The checkpoints themselves (2 x 16 MB) are not committed; the CPU test
re-creates the tensors from the same seeds, lays them out as the recorded files and checks `resume()` against the hashes.
"""
from __future__ import annotations

import argparse
import contextlib
import hashlib
import io
import json
import shutil
import sys
import tempfile
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))

from tests.golden.make_golden import import_reference  # noqa: E402
from tests.golden.make_golden_train import ref_cfg  # noqa: E402
from tests.weights import make_state_dict  # noqa: E402

COORD_SEED, ROT_SEED, COORD_EPOCH, ROT_EPOCH = 31, 32, 3, 7


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


def describe(path):
    ck = torch.load(path, map_location="cpu")
    return {"top_level_keys": sorted(ck), "epoch": int(ck["epoch"]), "iteration": int(ck["iteration"]),
            "optimizer_keys": sorted(ck["optimizer"]), "optimizer_param_groups": len(ck["optimizer"]["param_groups"]),
            "model_keys": list(ck["model"]), "model_dtypes": sorted({str(v.dtype) for v in ck["model"].values()})}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    import_reference(args.ref)
    from trainer import Trainer as RefTrainer
    tmp = Path(tempfile.mkdtemp(prefix="captra_ref_ckpt_"))
    held, files = {}, {}
    for tag, config, seed, epoch in (("coord", "config_coordnet.yml", COORD_SEED, COORD_EPOCH), ("rot", "config_rotnet.yml", ROT_SEED, ROT_EPOCH)):
        cfg = ref_cfg(config, "1", "obj_info_nocs.yml")
        cfg["experiment_dir"] = str(tmp / tag)
        with contextlib.redirect_stdout(io.StringIO()):
            tr = RefTrainer(cfg)
            shapes = {k: tuple(v.shape) for k, v in tr.model.state_dict().items()}
            tr.model.load_state_dict(make_state_dict(shapes, seed=seed))
            tr.epoch, tr.iteration = epoch, 100 * epoch
            tr.save()                                              # the reference's own writer
        files[tag] = Path(tr.ckpt_dir) / f"model_{epoch:04d}.pt"
        assert files[tag].exists(), files[tag]
        held[tag] = {k: v.clone() for k, v in tr.model.state_dict().items()}
    # ---- this repository's reader ---------------------------------------------------------------------------------------
    from captra_amd.configs import make_config
    from captra_amd.trainer import Trainer
    cfg = make_config("1", experiment_dir=str(tmp / "rot"), **{"coord_exp/dir": str(tmp / "coord")})
    with contextlib.redirect_stdout(io.StringIO()):
        mine = Trainer(cfg)
        got_epoch = mine.resume()
    assert got_epoch == ROT_EPOCH, got_epoch
    loaded = mine.model.state_dict()
    hashes, n_coord, n_rot = {}, 0, 0
    for k, v in loaded.items():
        if k.startswith("npcs_net."):
            src = held["coord"]["net" + k[len("npcs_net"):]]
            n_coord += 1
        else:
            src = held["rot"][k]
            n_rot += 1
        assert v.dtype == src.dtype and torch.equal(v.cpu(), src), k
        hashes[k] = sha(v)
    assert n_coord == len(held["coord"]) and n_rot == len(held["rot"]), (n_coord, len(held["coord"]), n_rot, len(held["rot"]))
    out = {"seeds": {"coord": COORD_SEED, "rot": ROT_SEED}, "files": {t: describe(p) for t, p in files.items()},
           "resume_verified_on_reference_written_files": True, "resume_epoch": got_epoch,
           "loaded_tensors": len(loaded), "loaded_sha256_16": hashes}
    with open(HERE / "ref_checkpoint.json", "w") as f:
        json.dump(out, f)
    shutil.rmtree(tmp, ignore_errors=True)
    print("wrote", HERE / "ref_checkpoint.json", len(hashes), "tensors;", "coord", n_coord, "rot", n_rot)


if __name__ == "__main__":
    main()
