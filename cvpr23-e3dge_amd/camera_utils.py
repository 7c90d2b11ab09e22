"""Camera sampling for the renderer's inputs -- mirror of project/utils/camera_utils.py:8-155
(`generate_camera_params`), without the pytorch3d import the reference file drags in (:2).

(azim, elev) -> camera on the unit sphere looking at the origin -> c2w pose (B,3,4), focal = 0.5*res/tan(fov),
near/far = 1 -/+ dist_radius, and (optionally) the uv-space calibration matrices the local branch projects with."""
import numpy as np
import torch
from torch.nn import functional as F


def generate_camera_params(resolution, device, batch=1, locations=None, sweep=False, uniform=False,
                           azim_range=0.3, elev_range=0.15, fov_ang=6, dist_radius=0.12, return_calibs=False,
                           azim_mean=0., elev_mean=0.):
    if locations is not None:
        azim = locations[:, 0].reshape(-1, 1)
        elev = locations[:, 1].reshape(-1, 1)
        n = azim.shape[0]
    elif sweep:
        azim = (-azim_range + (2 * azim_range / 7) * torch.arange(8, device=device)).reshape(-1, 1).repeat(batch, 1)
        elev = (-elev_range + 2 * elev_range * torch.rand(batch, 1, device=device).repeat(1, 8).reshape(-1, 1))
        n = batch * 8
    else:
        if uniform:
            azim = -azim_range + 2 * azim_range * torch.rand(batch, 1, device=device)
            elev = -elev_range + 2 * elev_range * torch.rand(batch, 1, device=device)
        else:
            azim = azim_range * torch.randn(batch, 1, device=device)
            elev = elev_range * torch.randn(batch, 1, device=device)
        n = batch
    dist = torch.ones(n, 1, device=device)                      # cameras live on the unit sphere
    near, far = (dist - dist_radius).unsqueeze(-1), (dist + dist_radius).unsqueeze(-1)
    fov_angle = fov_ang * torch.ones(n, 1, device=device) * np.pi / 180
    focal = 0.5 * resolution / torch.tan(fov_angle).unsqueeze(-1)

    azim = azim_mean + azim
    elev = elev_mean + elev
    viewpoint = torch.cat([azim, elev], 1)

    camera_dir = torch.stack([torch.cos(elev) * torch.sin(azim), torch.sin(elev),
                              torch.cos(elev) * torch.cos(azim)], dim=1).reshape(-1, 3)
    camera_loc = dist * camera_dir
    up = torch.tensor([[0, 1, 0]]).float().to(device) * torch.ones_like(dist)
    z_axis = F.normalize(camera_dir, eps=1e-5)                  # -z points into the screen
    x_axis = F.normalize(torch.cross(up, z_axis, dim=1), eps=1e-5)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    is_close = torch.isclose(x_axis, torch.tensor(0.0, device=device), atol=5e-3).all(dim=1, keepdim=True)
    if is_close.any():
        x_axis = torch.where(is_close, F.normalize(torch.cross(y_axis, z_axis, dim=1), eps=1e-5), x_axis)
    w2c_R = torch.stack((x_axis, y_axis, z_axis), dim=1)
    T = camera_loc[:, :, None]
    poses = torch.cat((w2c_R.transpose(1, 2), T), -1)           # (B,3,4) camera-to-world

    if not return_calibs:
        return poses, focal, near, far, viewpoint

    extrinsics = torch.cat((w2c_R, -w2c_R @ T), dim=-1)         # (B,3,4) world-to-camera
    focal_mat = torch.zeros(n, 3, 2, device=device)
    focal_mat[:, 0, 0] = focal_mat[:, 1, 1] = focal[0].squeeze()
    uv_offset = torch.zeros(n, 3, 1, device=device)
    uv_offset[:, -1, -1] = 1.
    intrinsics = torch.cat([focal_mat / (resolution / 2), uv_offset], -1)
    calibs = intrinsics @ extrinsics
    homo = torch.cat([torch.zeros(n, 1, 3), torch.ones(n, 1, 1)], dim=-1).to(device)
    return dict(poses=poses, extrinsics=extrinsics, focal=focal, near=near, far=far, viewpoint=viewpoint,
                intrinsics=intrinsics, calibs=torch.cat([calibs, homo], -2), locations=locations,
                azim_range=azim_range, elev_range=elev_range)


def trajectory_locations(n_frames, azim_amp=0.45, device="cpu"):
    """The azimuth sweep of the novel-view demo (trainer.py:2349-2388): azim = 1.5*0.3*cos(pi t), elev 0."""
    t = torch.arange(n_frames, dtype=torch.float32, device=device) / max(n_frames - 1, 1)
    return torch.stack([azim_amp * torch.cos(math_pi() * t), torch.zeros_like(t)], 1)


def math_pi():
    return float(np.pi)
