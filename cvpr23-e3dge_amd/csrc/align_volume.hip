// Surface extraction, device half: re-sampling of the 128^3 SDF volume that the renderer produced along the
// camera frustum onto the regular grid marching cubes expects.
//
// Reference: align_volume (project/utils/mesh_utils.py:17-44), called from VolumeFeatureRenderer.render on
// render_rays_out['sdf'] (volume_renderer.py:1706).  volume (b, h, w, d, c); for the output voxel (y, x, z) the sample point is
//     gx = xs[x] * coef[z],  gy = ys[y] * coef[z],  gz = zs[z]       xs, ys, zs = linspace(-1, 1, n), coef = linspace(far/near, 1, d)
// looked up by F.grid_sample(align_corners=True, padding_mode="border", trilinear) with x -> the w axis, y -> the h axis,
// z -> the d axis; voxels whose (gx, gy, gz) leave [-1, 1] are set to 1 afterwards ("avoid marching cubes distortions").
// The four 1-D tables are built by the host with torch.linspace, so their values are the reference's to the bit; the
// un-normalisation ((g + 1) / 2) * (n - 1), the clamp and the corner weights follow ATen's grid_sampler in fp32.
//
// One thread per output voxel and channel group, z (the contiguous axis of the (b, h, w, d, c) layout) across lanes: the
// eight corners of neighbouring lanes are neighbouring addresses.  Bound: HBM -- 8 B per voxel algorithmic (one read, one
// write; the corner re-reads hit L2), 16.8 MB for a 128^3 single-channel volume.
#include "common.h"

namespace e3dge {

__global__ void __launch_bounds__(256)
align_volume_kernel(float* __restrict__ out, const float* __restrict__ vol, const float* __restrict__ xs,
                    const float* __restrict__ ys, const float* __restrict__ zs, const float* __restrict__ coef,
                    int h, int w, int d, int c, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    int64_t r = e;
    const int z = (int)(r % d); r /= d;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h);
    const int64_t bi = r / h;
    const float cf = coef[z];
    const float gx = __fmul_rn(xs[x], cf), gy = __fmul_rn(ys[y], cf), gz = zs[z];
    float* __restrict__ o = out + e * c;
    if (gx < -1.0f || gx > 1.0f || gy < -1.0f || gy > 1.0f || gz < -1.0f || gz > 1.0f) {
        for (int ch = 0; ch < c; ++ch) o[ch] = 1.0f;
        return;
    }
    // grid_sampler_unnormalize (align_corners) + clip_coordinates (border)
    float ix = __fmul_rn(__fdiv_rn(__fadd_rn(gx, 1.0f), 2.0f), (float)(w - 1));
    float iy = __fmul_rn(__fdiv_rn(__fadd_rn(gy, 1.0f), 2.0f), (float)(h - 1));
    float iz = __fmul_rn(__fdiv_rn(__fadd_rn(gz, 1.0f), 2.0f), (float)(d - 1));
    ix = fminf((float)(w - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(h - 1), fmaxf(iy, 0.0f));
    iz = fminf((float)(d - 1), fmaxf(iz, 0.0f));
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float tx1 = __fsub_rn(ix, fx), ty1 = __fsub_rn(iy, fy), tz1 = __fsub_rn(iz, fz);          // weight of the +1 corner
    const float tx0 = __fsub_rn(__fadd_rn(fx, 1.0f), ix), ty0 = __fsub_rn(__fadd_rn(fy, 1.0f), iy), tz0 = __fsub_rn(__fadd_rn(fz, 1.0f), iz);
    const float* __restrict__ vb = vol + bi * (int64_t)h * w * d * c;
    for (int ch = 0; ch < c; ++ch) {
        float acc = 0.0f;
        // ATen's order: the z0 plane (nw, ne, sw, se), then the z0 + 1 plane; corners outside the volume contribute nothing
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int xx = x0 + dx, yy = y0 + dy, zz = z0 + dz;
                    if (xx < w && yy < h && zz < d) {
                        const float wgt = __fmul_rn(__fmul_rn(dx ? tx1 : tx0, dy ? ty1 : ty0), dz ? tz1 : tz0);
                        acc = __fadd_rn(acc, __fmul_rn(vb[(((int64_t)yy * w + xx) * d + zz) * c + ch], wgt));
                    }
                }
        o[ch] = acc;
    }
}

}  // namespace e3dge

using namespace e3dge;

extern "C" int e3dge_align_volume(float* out, const float* volume, const float* xs, const float* ys, const float* zs,
                                  const float* coef, int batch, int height, int width, int depth, int channels,
                                  e3dge_stream_t stream) {
    E3DGE_REQUIRE(batch >= 0 && height >= 1 && width >= 1 && depth >= 1 && channels >= 1, "align_volume: bad sizes");
    if (batch == 0) return E3DGE_OK;
    E3DGE_REQUIRE(out && volume && xs && ys && zs && coef, "align_volume: null pointer");
    E3DGE_REQUIRE(out != volume, "align_volume: in-place is not supported (every voxel gathers eight others)");
    const int64_t total = (int64_t)batch * height * width * depth;
    const int64_t grid = (total + 255) / 256;
    E3DGE_REQUIRE(grid < ((int64_t)1 << 31), "align_volume: volume too large");
    align_volume_kernel<<<dim3((unsigned)grid), dim3(256), 0, as_stream(stream)>>>(out, volume, xs, ys, zs, coef, height, width, depth,
                                                                                   channels, total);
    return check_launch("align_volume");
}
