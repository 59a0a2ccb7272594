"""Training step against golden G12 = the reference's own `Trainer.update` on the same seeded batch
(tests/golden/make_golden_train.py).  CPU: the loss code alone, fed with the reference's network outputs.  GPU: the whole
step — training-mode forward over the HIP operators, losses, backward through captra_group_points_grad /
captra_three_interpolate_grad, Adam."""
from pathlib import Path

import numpy as np
import pytest
import torch

from captra_amd.configs import make_config
from captra_amd.trainer import Trainer
from tests import clouds
from tests.golden.make_golden_train import CASES, TORCH_SEED
from tests.weights import make_state_dict

G = np.load(Path(__file__).resolve().parent / "golden" / "g12_train.npz")
G64 = np.load(Path(__file__).resolve().parent / "golden" / "g12_train64.npz")   # the same step, reference run in float64


def _trainer(tag, device):
    _, ntype, config, cat, objcfg, kind, wseed = next(c for c in CASES if c[0] == tag)
    cfg = make_config(cat, objcfg, config=config)
    cfg["device"] = device
    trainer = Trainer(cfg)
    model = trainer.model
    model.load_state_dict(make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=wseed))
    data = clouds.make_trajectory(kind, 2, 2, seed=3)[1]
    return trainer, data


def _loss_keys(tag):
    return sorted(k.split("/", 2)[2] for k in G.files if k.startswith(f"{tag}/loss/"))


@pytest.mark.parametrize("tag", ["coord_bottle", "coord_camera", "rot_bottle"])
def test_losses_from_reference_outputs_cpu(tag):
    """prepare_data (same noise draws as the reference) + compute_loss on the reference's network outputs: every entry of
    the loss dict, including the pair-wise-match term whose point pairs come from the generator after the pose noise."""
    trainer, data = _trainer(tag, "cpu")
    model = trainer.model
    torch.manual_seed(TORCH_SEED)
    np.random.seed(TORCH_SEED)
    model.set_data(data)
    part = {k: torch.from_numpy(G[f"{tag}/part/{k}"]) for k in ("rotation", "scale", "translation")}
    if tag.startswith("coord"):
        model.prepare_data()
        model.pred_dict = {"seg": torch.from_numpy(G[f"{tag}/pred/seg"]), "nocs": torch.from_numpy(G[f"{tag}/pred/nocs"]), "part": part}
        model.compute_loss()
    else:
        model.prepare_data(model.raw_feed_dict)
        model.pred_dict = {"part": part, "point_rotation": torch.from_numpy(G[f"{tag}/pred/point_rotation"])}
        model.compute_loss(test_mode=False)
    assert sorted(model.loss_dict) == _loss_keys(tag)
    for k in _loss_keys(tag):
        np.testing.assert_allclose(float(model.loss_dict[k]), float(G[f"{tag}/loss/{k}"]), rtol=2e-5, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("tag", [c[0] for c in CASES])
def test_reference_fp32_gradient_noise_floor_cpu(tag):
    """The two reference runs of the same step (fp32, G12; float64, G12-64) against each other: losses agree to 5e-5, the
    head gradients to 1e-3, while the backbone gradients of the fp32 run sit 0.3-1.7 % off the float64 ones — the noise
    floor the GPU gradient test is written against."""
    for k in _loss_keys(tag):
        if "loss" in k:
            assert abs(float(G[f"{tag}/loss/{k}"]) - float(G64[f"{tag}/loss/{k}"])) < 5e-5, k
    errs = {}
    for k in G64.files:
        if k.startswith(f"{tag}/grad/"):
            errs[k.split("/", 2)[2]] = np.abs(G[k] - G64[k]).max() / np.abs(G64[k]).max()
    heads = [v for n, v in errs.items() if "_head." in n or ".pose_pred." in n]
    body = [v for n, v in errs.items() if not ("_head." in n or ".pose_pred." in n)]
    assert heads and max(heads) < 1e-3
    assert 3e-3 < max(body) < 2e-2, errs


def test_trainer_schedules_cpu():
    """step_epoch: StepLR halves the rate every lr_step_size epochs down to lr_clip, BatchNorm momentum follows its own
    decay (reference trainer.py:125-145); save / resume round-trips the optimiser state."""
    import tempfile
    cfg = make_config("1", config="config_coordnet.yml", experiment_dir=tempfile.mkdtemp())
    cfg["device"] = "cpu"
    t = Trainer(cfg)
    for _ in range(41):
        t.step_epoch()
    assert t.epoch == 41 and abs(t.lr - 0.001 * 0.25) < 1e-12 and abs(t.momentum - 0.1 * 0.25) < 1e-12
    assert all(abs(m.momentum - 0.025) < 1e-12 for m in t.model.modules() if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)))
    t.save()
    t2 = Trainer(cfg)
    assert t2.resume() == 41 and t2.optimizer.state_dict()["param_groups"][0]["lr"] == t.optimizer.state_dict()["param_groups"][0]["lr"]


@pytest.mark.gpu
@pytest.mark.parametrize("tag", [c[0] for c in CASES])
def test_update_step_vs_reference_gpu(device, tag):
    """One `Trainer.update` on the GPU against the reference's on its CPU path: losses, predicted part poses, gradient
    norm, the gradients of parameters spread from the first SA layer to the heads, the parameters after the Adam step and
    a BatchNorm running mean.  Tolerances: 2e-4 relative on losses.  Gradients are judged against the reference's backward
    run in FLOAT64 (G12-64, make_golden_train64.py): the head probes to 1e-3 of the tensor's largest entry; the backbone
    probes — where fp32 itself is 0.3-1.7 % away from the float64 gradient, as the reference's OWN fp32 run recorded in G12
    shows — to 4x the error of that fp32 reference run (never more than 2 %), i.e. the GPU backward is held to the fp32
    noise floor of this gradient, measured, not assumed."""
    trainer, data = _trainer(tag, device)
    model = trainer.model
    torch.manual_seed(TORCH_SEED)
    np.random.seed(TORCH_SEED)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    loss_dict = trainer.update(data)
    for k in _loss_keys(tag):
        # rdiff = acos of a trace, in degrees: ill-conditioned at the few-degree errors of a perturbed pose (1e-7 on the
        # trace is 2e-4 degrees at 2 degrees) — an absolute 5e-3 degrees there
        np.testing.assert_allclose(float(loss_dict[k].detach()) if torch.is_tensor(loss_dict[k]) else float(loss_dict[k]),
                                   float(G[f"{tag}/loss/{k}"]), rtol=2e-4, atol=5e-3 if "rdiff" in k else 2e-5, err_msg=k)
    for k in ("rotation", "scale", "translation"):
        got, ref = model.pred_dict["part"][k].detach().cpu().numpy(), G[f"{tag}/part/{k}"]
        if k == "rotation" and model.sym:
            # a symmetric object's rotation is its y axis (column 1); the x / z columns complete it to a frame through a
            # normalised cross product, which amplifies the 1e-5 of the pooled axis
            np.testing.assert_allclose(got[..., 1], ref[..., 1], atol=1e-4, err_msg="rotation (axis)")
            np.testing.assert_allclose(got, ref, atol=1e-3, err_msg="rotation (frame)")
        else:
            np.testing.assert_allclose(got, ref, atol=1e-4, err_msg=k)
    params = dict(model.named_parameters())
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params.values() if p.grad is not None)))
    np.testing.assert_allclose(gn, float(G[f"{tag}/grad_norm"]), rtol=2e-3)
    np.testing.assert_allclose(gn, float(G64[f"{tag}/grad_norm"]), rtol=2e-3)
    lr = float(G[f"{tag}/meta"][0])
    probes = [k.split("/", 2)[2] for k in G.files if k.startswith(f"{tag}/grad/")]
    assert len(probes) >= 6
    for n in probes:
        ref = G[f"{tag}/grad/{n}"]
        got = params[n].grad.cpu().numpy()
        assert np.abs(got - ref).max() <= 0.02 * np.abs(ref).max() + 1e-7, (n, np.abs(got - ref).max(), np.abs(ref).max())
        ref64 = G64[f"{tag}/grad/{n}"]
        top = np.abs(ref64).max()
        err_gpu, err_ref32 = np.abs(got - ref64).max() / top, np.abs(ref - ref64).max() / top
        bound = 1e-3 if ("_head." in n or ".pose_pred." in n) else min(max(1e-3, 4.0 * err_ref32), 0.02)
        assert err_gpu <= bound, (n, "gpu vs float64", err_gpu, "reference fp32 vs float64", err_ref32)
        # Adam's first step moves every weight by lr * sign-like(g): compare the step where the gradient is not ~0
        step_ref = G[f"{tag}/param/{n}"] - before[n].cpu().numpy()
        step_got = params[n].detach().cpu().numpy() - before[n].cpu().numpy()
        big = np.abs(ref) > 0.05 * np.abs(ref).max()
        np.testing.assert_allclose(step_got[big], step_ref[big], atol=0.05 * lr, err_msg=n)
    bn = [k for k in G.files if k.startswith(f"{tag}/buffer/")][0]
    np.testing.assert_allclose(model.state_dict()[bn.split("/", 2)[2]].cpu().numpy(), G[bn], rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_train_cli_one_epoch(device, tmp_path, monkeypatch):
    """python -m captra_amd.train: config from flags, one epoch over synthetic batches, averaged losses, checkpoint written
    and resumable (epoch, optimiser state)."""
    import sys
    from captra_amd import train
    exp = str(tmp_path / "exp")
    monkeypatch.setattr(sys, "argv", ["train", "--config", "config_coordnet.yml", "--obj_category", "1", "--experiment_dir", exp,
                                      "--batch_size", "2", "--total_epoch", "1", "--samples", "4"])
    train.main()
    ckpt = torch.load(f"{exp}/ckpt/model_0001.pt", map_location="cpu")
    assert ckpt["epoch"] == 1 and ckpt["iteration"] == 2 and ckpt["optimizer"]["state"]
    cfg = make_config("1", config="config_coordnet.yml", experiment_dir=exp)
    cfg["device"] = device
    assert Trainer(cfg).resume() == 1


@pytest.mark.gpu
@pytest.mark.parametrize("tag", [c[0] for c in CASES])
def test_trainer_test_pass_vs_reference_gpu(device, tag):
    """`Trainer.test(data)` of the training experiments (eval-mode networks = the fused inference kernels, predicted labels,
    the experiment's loss dict) against the reference's on the same seeded batch."""
    trainer, data = _trainer(tag, device)
    torch.manual_seed(TORCH_SEED + 1)
    np.random.seed(TORCH_SEED + 1)
    _, loss_dict = trainer.test(data)
    keys = sorted(k.split("/", 2)[2] for k in G.files if k.startswith(f"{tag}/test_loss/"))
    assert sorted(loss_dict) == keys
    for k in keys:
        got = float(loss_dict[k].detach()) if torch.is_tensor(loss_dict[k]) else float(loss_dict[k])
        ref = float(G[f"{tag}/test_loss/{k}"])
        if "deg" in k and "cm" in k:
            assert abs(got - ref) < 1e-6, k              # hit rates: the same poses fall on the same side of 5 deg / 5 cm
        else:
            np.testing.assert_allclose(got, ref, rtol=2e-4, atol=5e-3 if "rdiff" in k else 2e-5, err_msg=k)


@pytest.mark.parametrize("tag", [c[0] for c in CASES])
def test_training_model_state_dict_keys_and_shapes(tag):
    """Checkpoints of the training experiments are interchangeable with the reference's: same state-dict keys and tensor
    shapes (captured from the reference's Trainer(cfg).model)."""
    import json
    ref = json.load(open(Path(__file__).resolve().parent / "golden" / "state_dict_keys_train.json"))[tag]
    trainer, _ = _trainer(tag, "cpu")
    mine = {k: list(v.shape) for k, v in trainer.model.state_dict().items()}
    assert mine == ref
