"""CPU restatement of the device half of the reference's surface extraction -- TEST INFRASTRUCTURE (only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this).

align_volume follows project/utils/mesh_utils.py:17-44: the sampling volume rendered along the camera frustum,
(b, h, w, d, c), is looked up at  (xs[x] coef[z], ys[y] coef[z], zs[z])  with
F.grid_sample(align_corners=True, padding_mode="border") -- written out here corner by corner in the order and
arithmetic of ATen's grid_sampler_3d (unnormalise ((g + 1) / 2) (n - 1), clamp, weights (x * y) * z, eight
multiply-then-add steps) -- and voxels whose sample point leaves [-1, 1]^3 are set to 1 (:40-42).
Pinned against the imported reference function by oracle/gen_golden_align.py (bit-exact, see the fixture's report)."""
import torch


def frustum_tables(h, w, d, near=0.88, far=1.12, dtype=torch.float32):
    """xs (w), ys (h), zs (d), coef (d) exactly as the reference builds them (torch.linspace, :20-28)."""
    return (torch.linspace(-1, 1, w, dtype=dtype), torch.linspace(-1, 1, h, dtype=dtype), torch.linspace(-1, 1, d, dtype=dtype),
            torch.linspace(far / near, 1, d, dtype=dtype))


def align_volume(volume, near=0.88, far=1.12):
    b, h, w, d, c = volume.shape
    dt = volume.dtype
    xs, ys, zs, coef = frustum_tables(h, w, d, near, far, dt)
    gx = (xs.view(1, w, 1) * coef.view(1, 1, d)).expand(h, w, d)
    gy = (ys.view(h, 1, 1) * coef.view(1, 1, d)).expand(h, w, d)
    gz = zs.view(1, 1, d).expand(h, w, d)
    outside = (gx < -1) | (gx > 1) | (gy < -1) | (gy > 1) | (gz < -1) | (gz > 1)

    def source_index(g, n):
        i = ((g + 1) / 2) * (n - 1)
        return torch.clamp(i, 0, n - 1)
    ix, iy, iz = source_index(gx, w), source_index(gy, h), source_index(gz, d)
    fx, fy, fz = torch.floor(ix), torch.floor(iy), torch.floor(iz)
    x0, y0, z0 = fx.long(), fy.long(), fz.long()
    wx = ((fx + 1) - ix, ix - fx)
    wy = ((fy + 1) - iy, iy - fy)
    wz = ((fz + 1) - iz, iz - fz)
    out = torch.zeros((b, h, w, d, c), dtype=dt)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                xx, yy, zz = x0 + dx, y0 + dy, z0 + dz
                ok = (xx < w) & (yy < h) & (zz < d)
                wgt = (wx[dx] * wy[dy]) * wz[dz]
                val = volume[:, yy.clamp(max=h - 1), xx.clamp(max=w - 1), zz.clamp(max=d - 1), :]      # (b, h, w, d, c)
                term = val * wgt.unsqueeze(-1)
                out = torch.where(ok.view(1, h, w, d, 1), out + term, out)
    return torch.where(outside.view(1, h, w, d, 1), torch.ones((), dtype=dt), out)
