"""Surface extraction, device half: mirrors project/utils/mesh_utils.py of the reference.

    align_volume(volume, near=0.88, far=1.12)        mesh_utils.py:17-44, one HIP kernel (e3dge_align_volume)
    frustum_tables(h, w, d, near, far, device)       the four linspace tables the kernel takes, cached per shape
    marching_cubes_mesh(aligned_sdf)                 the CPU step that follows it (volume_renderer.py:1733-1758): skimage +
                                                     trimesh, third-party and outside the path -- raises ImportError with that
                                                     message when they are not installed

The renderer calls align_volume for `return_mesh=True` (volume_renderer.py:1703-1731 of the reference) and returns the aligned
volume as 'aligned_sdf'; 'mesh' is filled only when the third-party packages are there."""
import torch
import torch.nn.functional as F

from . import _lib

_TABLES = {}


def frustum_tables(h, w, d, near, far, device):
    """xs (w), ys (h), zs (d) = linspace(-1, 1, n) and coef (d) = linspace(far / near, 1, d): torch.linspace on the host, so
    the values are the reference's own (:20-28)."""
    key = (int(h), int(w), int(d), float(near), float(far), str(device))
    t = _TABLES.get(key)
    if t is None:
        t = tuple(x.to(device) for x in (torch.linspace(-1, 1, w), torch.linspace(-1, 1, h), torch.linspace(-1, 1, d),
                                         torch.linspace(far / near, 1, d)))
        if len(_TABLES) > 16:
            _TABLES.clear()
        _TABLES[key] = t
    return t


def align_volume(volume, near=0.88, far=1.12):
    """(b, h, w, d, c) sampling volume along the camera frustum -> the same shape on the regular grid; voxels outside the
    frustum become 1.  CUDA tensors: e3dge_align_volume; CPU tensors: grid_sample, as the reference does."""
    if volume.dim() != 5:
        raise RuntimeError(f"align_volume expects (b, h, w, d, c), got {tuple(volume.shape)}")
    b, h, w, d, c = volume.shape
    xs, ys, zs, coef = frustum_tables(h, w, d, near, far, volume.device)
    if not volume.is_cuda:
        gx = (xs.view(1, 1, w) * coef.view(d, 1, 1)).expand(d, h, w)
        gy = (ys.view(1, h, 1) * coef.view(d, 1, 1)).expand(d, h, w)
        gz = zs.view(d, 1, 1).expand(d, h, w)
        grid = torch.stack([gx, gy, gz], -1).unsqueeze(0).expand(b, d, h, w, 3).to(volume.dtype)
        out = F.grid_sample(volume.permute(0, 4, 3, 1, 2), grid, padding_mode="border", align_corners=True)
        out = out.permute(0, 3, 4, 2, 1).contiguous()
        outside = ((grid < -1) | (grid > 1)).any(-1)[0].permute(1, 2, 0)                      # (h, w, d)
        return torch.where(outside.view(1, h, w, d, 1), torch.ones((), dtype=out.dtype), out)
    if volume.dtype != torch.float32:
        raise RuntimeError(f"align_volume: float32 expected on the GPU, got {volume.dtype}")
    vol = volume.detach().contiguous()
    out = torch.empty_like(vol)
    with torch.cuda.device(vol.device):
        rc = _lib.load().e3dge_align_volume(_lib.ptr(out), _lib.ptr(vol), _lib.ptr(xs), _lib.ptr(ys), _lib.ptr(zs),
                                            _lib.ptr(coef), b, h, w, d, c, _lib.stream_of(vol))
    _lib.check(rc, "e3dge_align_volume")
    return out


def marching_cubes_mesh(aligned_sdf):
    """The reference's _extract_mesh_with_marching_cubes (volume_renderer.py:1733-1758) on an aligned (1, h, w, d, 1) volume:
    (mesh, verts, faces).  skimage / trimesh are third-party CPU code outside the path."""
    try:
        from skimage.measure import marching_cubes
        import trimesh
    except ImportError as e:
        raise ImportError("marching cubes needs scikit-image and trimesh (CPU, third-party, outside the accelerated path); "
                          "pass 'aligned_sdf' to your own extractor") from e
    _, h, w, d, _ = aligned_sdf.shape
    vol = aligned_sdf[0, ..., 0].permute(1, 0, 2).cpu().numpy()             # (y, x, z) -> (x, y, z)
    verts, faces, _, _ = marching_cubes(vol, 0)
    for axis, n in enumerate((w, h, d)):
        verts[:, axis] = (verts[:, axis] / float(n) - 0.5) * 0.24            # back to the scene scale [-0.12, 0.12]
    verts[:, 2] *= -1
    verts[:, 1] *= -1
    return trimesh.Trimesh(verts, faces), verts, faces
