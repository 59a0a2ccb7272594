"""Worker of the multi-rank harness tests: `python -m captra_amd.track` semantics inside one rank of a world.

CPU flavour (tests/test_parallel_cpu.py): the product has no CPU path for the networks, so the rank runs the REAL harness
(captra_amd/track.py: sharding, FramePoseGather, result gathering), the REAL EvalTrackModel loop (forward, hooks, _save)
and a stand-in for the one method that needs the GPU -- `track_step`, replaced by deterministic host arithmetic on the
frame's cloud and the previous pose.  GPU flavour (tests/test_dist_gpu.py): nothing is replaced."""
import os
import sys

import torch


def install_cpu_track_step():
    from captra_amd import model as M

    class HostStepModel(M.EvalTrackModel):
        def _graph_usable(self, input):
            return False

        def track_step(self, input, npcs_input, last_pose):
            pts = input["points"].double()                                   # (B,3,N)
            B, _, N = pts.shape
            P = self.num_parts
            s = pts.mean(dim=(1, 2)).reshape(B, 1)                           # a per-trajectory number that depends on the data
            ang = 0.01 * (1.0 + s)                                           # small rotation about y, composed on the previous pose
            c, sn = torch.cos(ang), torch.sin(ang)
            z, o = torch.zeros_like(c), torch.ones_like(c)
            dR = torch.stack([torch.cat([c, z, sn], 1), torch.cat([z, o, z], 1), torch.cat([-sn, z, c], 1)], 1)   # (B,3,3)
            pose = {"rotation": torch.matmul(last_pose["rotation"].double(), dR.unsqueeze(1)).float(),
                    "translation": (last_pose["translation"].double() + 0.001 * pts[:, :, :P].transpose(1, 2).unsqueeze(-1)).float(),
                    "scale": (last_pose["scale"].double() * (1.0 + 0.01 * s)).float()}
            C = P + int(self.cfg["obj"]["extra_dims"])
            seg = torch.softmax(pts[:, :1].expand(B, C, N) * torch.arange(1, C + 1).reshape(1, C, 1), dim=1).float()
            nocs = (pts.repeat(1, P, 1) * 0.5).float()
            return {"seg": seg, "nocs": nocs, "points": input["points"]}, pose

    M.EvalTrackModel = HostStepModel
    import captra_amd.trainer as T
    T.EvalTrackModel = HostStepModel


def main():
    out_dir, world_tag = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    if os.environ.get("CAPTRA_TEST_HOST_STEP") == "1":
        install_cpu_track_step()
    from captra_amd import track
    torch.manual_seed(0)
    res = track.main(["--obj_category", "1", "--experiment_dir", os.path.join(out_dir, world_tag), "--batch_size", "2",
                      "--data", "synthetic", "--num_traj", os.environ.get("CAPTRA_TEST_NUM_TRAJ", "5"), "--num_frames", "4", "--random_init", "--save",
                      "--pose_perturb/r", "0", "--pose_perturb/t", "0", "--pose_perturb/s", "0"] + extra)
    if int(os.environ.get("RANK", "0")) == 0:
        import json
        with open(os.path.join(out_dir, f"{world_tag}_result.json"), "w") as f:
            json.dump(res, f)


if __name__ == "__main__":
    main()
