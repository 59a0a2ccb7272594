"""Golden G16 (tests/golden/make_golden_checkpoint.py): `Trainer.resume` against checkpoints written by the REFERENCE's own
`Trainer.save` (network/trainer.py:196-210).  The generator resumed the reference-written files with this repository's Trainer
and recorded their structure plus a hash per loaded tensor; here the same tensors are re-created from the recorded seeds, laid
out exactly as the recorded files (top-level keys, `net.*` model keys in the reference's order, an Adam state dict), and
`resume()` must reproduce every hash -- the CoordNet experiment's `net.*` under `npcs_net.*` (trainer.py:159-169)."""
import hashlib
import json
from pathlib import Path

import torch

from tests.weights import make_state_dict

G = Path(__file__).resolve().parent / "golden"


def _sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


def test_resume_reproduces_the_reference_written_checkpoints(tmp_path):
    from captra_amd.configs import make_config
    from captra_amd.trainer import Trainer
    rec = json.loads((G / "ref_checkpoint.json").read_text())
    assert rec["resume_verified_on_reference_written_files"] and rec["loaded_tensors"] == 375
    cfg = make_config("1", experiment_dir=str(tmp_path / "rot"), **{"coord_exp/dir": str(tmp_path / "coord")})
    probe = Trainer(cfg)
    shapes = {k: tuple(v.shape) for k, v in probe.model.state_dict().items()}
    for tag, prefix in (("coord", "npcs_net."), ("rot", "net.")):
        info = rec["files"][tag]
        assert info["top_level_keys"] == ["epoch", "iteration", "model", "optimizer"] and info["model_dtypes"] in (["torch.float32"], ["torch.float32", "torch.int64"])
        # the experiment's own state dict: `net.*` keys in the reference's order, values from the recorded seed
        exp_shapes = {k: shapes[(prefix + k[len("net."):]) if tag == "coord" else k] for k in info["model_keys"]}
        sd = make_state_dict(exp_shapes, seed=rec["seeds"][tag])          # (values depend on the names, not on the order)
        sd = {k: sd[k] for k in info["model_keys"]}                       # the reference's own key order
        (tmp_path / tag / "ckpt").mkdir(parents=True)
        torch.save({"epoch": info["epoch"], "iteration": info["iteration"], "model": sd,
                    "optimizer": {"state": {}, "param_groups": [{} for _ in range(info["optimizer_param_groups"])]}},
                   tmp_path / tag / "ckpt" / f"model_{info['epoch']:04d}.pt")
    tr = Trainer(cfg)
    assert tr.resume() == rec["resume_epoch"]
    loaded = tr.model.state_dict()
    assert len(loaded) == rec["loaded_tensors"]
    bad = [k for k, v in loaded.items() if _sha(v) != rec["loaded_sha256_16"][k]]
    assert not bad, bad[:5]
