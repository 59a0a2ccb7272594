"""cProfile of the EvalTrackModel loop's host side (where the 0.7 ms per step beyond the captured graph goes)."""
import cProfile
import pstats
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from captra_amd.configs import make_config  # noqa: E402
from captra_amd.trainer import Trainer  # noqa: E402
from captra_amd import synthetic as clouds  # noqa: E402
from captra_amd.synthetic import make_state_dict  # noqa: E402

dev = torch.device("cuda:0")
cfg = make_config("1", experiment_dir="/tmp/captra_prof", hipgraph=True)
cfg["device"] = dev
trainer = Trainer(cfg)
trainer.model.load_state_dict(make_state_dict({k: tuple(v.shape) for k, v in trainer.model.state_dict().items()}, seed=7))
data = clouds.make_trajectory("nocs", 32, 12, seed=0)
trainer.model.eval()
for _ in range(2):
    trainer.model.set_data(data)
    trainer.model.test(save=False, no_eval=True)
torch.cuda.synchronize()
trainer.model.set_data(data)
pr = cProfile.Profile()
pr.enable()
trainer.model.test(save=False, no_eval=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
