// Reference-view hit probability of query-view points: VolumeFeatureRenderer.query_hitting_probability_fixed_interval
// (project/utils/volume_renderer.py:1326-1495; compositing without the far-plane stop :826-837, :884; caller cycle_runner.py:139-158).
//
// For every point p of the query view: the reference camera's ray THROUGH p is sampled at the renderer's Sn fixed depths, the SDF
// network is queried there (e3dge_siren_points_fwd, sdf only), the samples are composited with `no_force_stop` (last interval =
// first interval, no background weight) and the per-sample weight (or visibility) is linearly interpolated at p's position along
// that ray.  Round 2 ran everything around the point query as ~20 eager torch kernels; here it is two launches:
//   e3dge_hitprob_points     p -> the Sn query points + (lo, hi, frac) of the interpolation        (bound: HBM, 12 Sn + 28 B per point)
//   e3dge_hitprob_composite  sdf (Sn per point) -> alpha, transmittance scan, lerp                 (bound: HBM, 4 Sn + 20 B per point)
// Arithmetic follows the reference's op order with explicitly rounded fp32 operations (no contraction).
#include "siren_common.h"

namespace e3dge {

// row-major (3, 4) matrix times (x, y, z, 1) / (x, y, z, 0), products summed left to right as the einsum's inner loop does
__device__ __forceinline__ float dot3(const float* m, float x, float y, float z) {
    return __fadd_rn(__fadd_rn(__fmul_rn(m[0], x), __fmul_rn(m[1], y)), __fmul_rn(m[2], z));
}

__global__ void __launch_bounds__(256)
hitprob_points_kernel(float* __restrict__ q, float* __restrict__ aux, const float* __restrict__ pts, const float* __restrict__ poses,
                      const float* __restrict__ extr, const float* __restrict__ near, const float* __restrict__ far,
                      const float* __restrict__ t_vals, int64_t rays, int s_pts, int sn, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;        // (b, ray, s)
    if (i >= total) return;
    const int64_t per_b = rays * s_pts;
    const int b = (int)(i / per_b);
    const int64_t ray = (i - (int64_t)b * per_b) / s_pts;
    const float* E = extr + (size_t)b * 12;
    const float* P = poses + (size_t)b * 12;
    const float px = pts[i * 3 + 0], py = pts[i * 3 + 1], pz = pts[i * 3 + 2];
    // the point in the reference camera's frame, its ray direction scaled like the mesh-grid directions (z = -1) (:1367-1379)
    const float rx = __fadd_rn(dot3(E + 0, px, py, pz), E[3]), ry = __fadd_rn(dot3(E + 4, px, py, pz), E[7]), rz = __fadd_rn(dot3(E + 8, px, py, pz), E[11]);
    const float nz = -rz;
    const float dx = __fdiv_rn(rx, nz), dy = __fdiv_rn(ry, nz), dz = __fdiv_rn(rz, nz);
    const float wx = dot3(P + 0, dx, dy, dz), wy = dot3(P + 4, dx, dy, dz), wz = dot3(P + 8, dx, dy, dz);     // world-space direction
    const float ox = P[3], oy = P[7], oz = P[11];
    const float nr = near[(size_t)b * rays + ray], fr = far[(size_t)b * rays + ray];
    const float z0 = __fadd_rn(__fmul_rn(nr, __fsub_rn(1.0f, t_vals[0])), __fmul_rn(fr, t_vals[0]));
    const float z1 = __fadd_rn(__fmul_rn(nr, __fsub_rn(1.0f, t_vals[1])), __fmul_rn(fr, t_vals[1]));
    const float nrm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(wx, wx), __fmul_rn(wy, wy)), __fmul_rn(wz, wz)));
    const float interval = __fmul_rn(__fsub_rn(z1, z0), nrm);
    float* __restrict__ qo = q + i * 3 * sn;
    float q0x = 0.f, q0y = 0.f, q0z = 0.f;
    for (int s = 0; s < sn; ++s) {
        const float t = t_vals[s];
        const float z = __fadd_rn(__fmul_rn(nr, __fsub_rn(1.0f, t)), __fmul_rn(fr, t));
        const float x_ = __fadd_rn(ox, __fmul_rn(wx, z)), y_ = __fadd_rn(oy, __fmul_rn(wy, z)), z_ = __fadd_rn(oz, __fmul_rn(wz, z));
        if (s == 0) { q0x = x_; q0y = y_; q0z = z_; }
        qo[s * 3 + 0] = x_; qo[s * 3 + 1] = y_; qo[s * 3 + 2] = z_;
    }
    const float ex = __fsub_rn(px, q0x), ey = __fsub_rn(py, q0y), ez = __fsub_rn(pz, q0z);
    const float dist = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez)));
    const float idx = __fadd_rn(__fdiv_rn(dist, interval), 1e-5f);
    const float top = (float)(sn - 1);
    const float lo = fminf(fmaxf(floorf(idx), 0.0f), top), hi = fminf(fmaxf(ceilf(idx), 0.0f), top);
    aux[i * 4 + 0] = lo; aux[i * 4 + 1] = hi; aux[i * 4 + 2] = __fsub_rn(idx, lo); aux[i * 4 + 3] = idx;
}

__global__ void __launch_bounds__(256)
hitprob_composite_kernel(float* __restrict__ out, const float* __restrict__ sdf, const float* __restrict__ aux,
                         const float* __restrict__ near, const float* __restrict__ far, const float* __restrict__ t_vals,
                         float beta, int visibility, int64_t rays, int s_pts, int sn, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t per_b = rays * s_pts;
    const int b = (int)(i / per_b);
    const int64_t ray = (i - (int64_t)b * per_b) / s_pts;
    const float nr = near[(size_t)b * rays + ray], fr = far[(size_t)b * rays + ray];
    const int lo = (int)aux[i * 4 + 0], hi = (int)aux[i * 4 + 1];
    const float w = aux[i * 4 + 2];
    const float* __restrict__ sd = sdf + i * sn;
    auto zval = [&](int s) { const float t = t_vals[s]; return __fadd_rn(__fmul_rn(nr, __fsub_rn(1.0f, t)), __fmul_rn(fr, t)); };
    const float d_first = __fsub_rn(zval(1), zval(0));
    float vis = 1.0f, f = 0.0f, c = 0.0f, zprev = zval(0);
    for (int s = 0; s < sn; ++s) {                 // front-to-back, the order of torch.cumprod
        float dist = d_first;                      // no_force_stop: the last interval repeats the first (:826-837)
        if (s + 1 < sn) { const float zn = zval(s + 1); dist = __fsub_rn(zn, zprev); zprev = zn; }
        const float sg = __fdiv_rn(sigmoid_f32(__fdiv_rn(-sd[s], beta)), beta);
        const float alpha = __fsub_rn(1.0f, expf(-__fmul_rn(sg, dist)));
        const float val = visibility ? vis : __fmul_rn(alpha, vis);
        if (s == lo) f = val;
        if (s == hi) c = val;
        vis = __fmul_rn(vis, __fadd_rn(__fsub_rn(1.0f, alpha), 1e-10f));
    }
    // torch.lerp: start + w (end - start) for w < 0.5, end - (end - start)(1 - w) otherwise
    const float diff = __fsub_rn(c, f);
    out[i] = w < 0.5f ? __fadd_rn(f, __fmul_rn(w, diff)) : __fsub_rn(c, __fmul_rn(diff, __fsub_rn(1.0f, w)));
}

}  // namespace e3dge

using namespace e3dge;

extern "C" int e3dge_hitprob_points(float* q, float* aux, const float* pts, const float* poses, const float* extrinsics, const float* near,
                                    const float* far, const float* t_vals, int batch, int64_t rays, int s_pts, int n_samples,
                                    e3dge_stream_t stream) {
    E3DGE_REQUIRE(batch >= 0 && rays >= 0 && s_pts >= 1 && n_samples >= 2, "hitprob_points: bad sizes");
    const int64_t total = (int64_t)batch * rays * s_pts;
    if (total == 0) return E3DGE_OK;
    E3DGE_REQUIRE(q && aux && pts && poses && extrinsics && near && far && t_vals, "hitprob_points: null pointer");
    E3DGE_REQUIRE((total + 255) / 256 < ((int64_t)1 << 31), "hitprob_points: too many points");
    hitprob_points_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(q, aux, pts, poses, extrinsics, near, far,
                                                                                                      t_vals, rays, s_pts, n_samples, total);
    return check_launch("hitprob_points");
}

extern "C" int e3dge_hitprob_composite(float* out, const float* sdf, const float* aux, const float* near, const float* far,
                                       const float* t_vals, float sigmoid_beta, int visibility, int batch, int64_t rays, int s_pts,
                                       int n_samples, e3dge_stream_t stream) {
    E3DGE_REQUIRE(batch >= 0 && rays >= 0 && s_pts >= 1 && n_samples >= 2 && sigmoid_beta > 0.0f, "hitprob_composite: bad sizes");
    const int64_t total = (int64_t)batch * rays * s_pts;
    if (total == 0) return E3DGE_OK;
    E3DGE_REQUIRE(out && sdf && aux && near && far && t_vals, "hitprob_composite: null pointer");
    E3DGE_REQUIRE((total + 255) / 256 < ((int64_t)1 << 31), "hitprob_composite: too many points");
    hitprob_composite_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(out, sdf, aux, near, far, t_vals,
                                                                                                         sigmoid_beta, visibility, rays, s_pts,
                                                                                                         n_samples, total);
    return check_launch("hitprob_composite");
}
