"""CPU restatement of generate_camera_params for explicit (azim, elev) locations
(project/utils/camera_utils.py:23-35, 77-111).  TEST INFRASTRUCTURE -- see oracle/__init__.py."""
import math

import torch
from torch.nn import functional as F


def camera_from_locations(resolution, locations, fov_ang=6, dist_radius=0.12, dtype=torch.float32):
    loc = locations.to(dtype)
    azim, elev = loc[:, 0:1], loc[:, 1:2]
    n = azim.shape[0]
    dist = torch.ones(n, 1, dtype=dtype)
    near, far = (dist - dist_radius).unsqueeze(-1), (dist + dist_radius).unsqueeze(-1)
    focal = 0.5 * resolution / torch.tan(fov_ang * torch.ones(n, 1, dtype=dtype) * math.pi / 180).unsqueeze(-1)
    cam_dir = torch.stack([torch.cos(elev) * torch.sin(azim), torch.sin(elev), torch.cos(elev) * torch.cos(azim)], 1).reshape(-1, 3)
    up = torch.tensor([[0., 1., 0.]], dtype=dtype).expand(n, 3)
    z = F.normalize(cam_dir, eps=1e-5)
    x = F.normalize(torch.cross(up, z, dim=1), eps=1e-5)
    y = F.normalize(torch.cross(z, x, dim=1), eps=1e-5)
    close = torch.isclose(x, torch.tensor(0.0, dtype=dtype), atol=5e-3).all(dim=1, keepdim=True)
    x = torch.where(close, F.normalize(torch.cross(y, z, dim=1), eps=1e-5), x)
    R = torch.stack([x, y, z], dim=2)                              # columns = camera axes in the world frame
    poses = torch.cat([R, (dist * cam_dir)[:, :, None]], -1)
    return poses, focal, near, far
