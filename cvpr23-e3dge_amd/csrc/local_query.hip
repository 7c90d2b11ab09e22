// Per-point query of the local branch's feature maps: perspective projection of world-space points into a
// view, bilinear gather of a feature map at the projected position, in-image mask, and the positional encoding of the
// points -- the inputs of the second renderer pass's texture head (project/trainers/E3DGE/e3dge_full_runner.py:185-317).
//
// Reference being replaced (per point, PyTorch ops):
//   HGPIFuNetGAN.query (vendor/pifu/lib/model/HGPIFuGANNet.py:85-151): xyz = perspective(points, calibs); y *= -1;
//       in_img = |x| <= 1 & |y| <= 1; feats = index(im_feat, xy)
//   perspective (vendor/pifu/lib/geometry.py:101-129): homo = trans + rot p;  z = -homo_z if the FIRST point of the FIRST
//       sample has homo_z < 0 else homo_z;  xy = homo_xy / z
//   index (geometry.py:64-80): grid_sample(feat, uv, bilinear, zeros padding, align_corners=False)
//   PosEncoding.forward (project/utils/misc_utils.py:148-185): [x, sin(2^k x), cos(2^k x)]_k, k = 0..n_freqs-1
//
// Layout: the feature map is CHANNEL-LAST (B, h, w, C) in HBM, so the four corners of a point are four contiguous C-float
// rows: one wave per point, 16 B per lane per corner, every byte of a fetched line is used.  (NCHW would make each of the
// 4*C corner reads of a point its own cache line.)  Outputs go straight into a slice [col_off, col_off + C) of a wider
// per-point row (leading dimension ld), so the concatenations of the reference (:255-262, :279) cost nothing.
// Bound: HBM / L2 gather, (4 C reads + C writes) * 4 B per point.
#include "common.h"

namespace e3dge {

typedef float lq_f4 __attribute__((ext_vector_type(4)));

struct LocalQueryK {
    const float* pts;      // (B, N, 3) world-space points
    const float* calibs;   // (B, 3, 4)
    const float* fmap;     // (B, h, w, C) channel-last, or null (projection / mask only)
    float* out;            // (B, N, ld): features written to [col_off, col_off + C)
    float* in_img;         // (B, N) or null: 1.0 inside the image plane, else 0.0 (may alias a column of `out` via mask_ld)
    float* proj;           // (B, N, 3) or null: projected (x, y flipped, depth)
    long long N;
    int B, C, h, w, ld, col_off, mask_ld, mask_off;
};

__global__ void __launch_bounds__(256) local_query_kernel(const LocalQueryK a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // sign convention of the whole batch: decided by the first point of the first sample (geometry.py:115-118)
    float zsign;
    {
        const float* c = a.calibs;
        const float* p = a.pts;
        const float hz = fmaf(c[8], p[0], fmaf(c[9], p[1], fmaf(c[10], p[2], c[11])));
        zsign = hz < 0.0f ? -1.0f : 1.0f;
    }
    const long long total = (long long)a.B * a.N;
    for (long long pt = (long long)blockIdx.x * 4 + wave; pt < total; pt += (long long)gridDim.x * 4) {
        const int b = (int)(pt / a.N);
        const float* c = a.calibs + (size_t)b * 12;
        const float* p = a.pts + (size_t)pt * 3;
        const float px = p[0], py = p[1], pz = p[2];
        const float hx = c[3] + (c[0] * px + c[1] * py + c[2] * pz);
        const float hy = c[7] + (c[4] * px + c[5] * py + c[6] * pz);
        const float hz = c[11] + (c[8] * px + c[9] * py + c[10] * pz);
        const float z = zsign * hz;
        const float x = hx / z, y = -(hy / z);                        // y flipped to grid_sample's convention (:109)
        const bool inside = x >= -1.0f && x <= 1.0f && y >= -1.0f && y <= 1.0f;
        if (lane == 0) {
            if (a.in_img) a.in_img[(size_t)pt * a.mask_ld + a.mask_off] = inside ? 1.0f : 0.0f;
            if (a.proj) { float* q = a.proj + (size_t)pt * 3; q[0] = x; q[1] = y; q[2] = z; }
        }
        if (!a.fmap) continue;
        // grid_sample, align_corners=False: pixel coordinate = ((g + 1) * size - 1) / 2
        const float fx = ((x + 1.0f) * (float)a.w - 1.0f) * 0.5f, fy = ((y + 1.0f) * (float)a.h - 1.0f) * 0.5f;
        const float x0f = floorf(fx), y0f = floorf(fy);
        const float tx = fx - x0f, ty = fy - y0f;
        // NaN / huge coordinates: every corner out of range -> zeros (as grid_sample's zeros padding gives)
        const bool finite = fx > -2.0f && fx < (float)a.w + 1.0f && fy > -2.0f && fy < (float)a.h + 1.0f;
        const int x0 = finite ? (int)x0f : -5, y0 = finite ? (int)y0f : -5;
        const float w00 = (1.0f - tx) * (1.0f - ty), w01 = tx * (1.0f - ty), w10 = (1.0f - tx) * ty, w11 = tx * ty;
        const bool vx0 = x0 >= 0 && x0 < a.w, vx1 = x0 + 1 >= 0 && x0 + 1 < a.w;
        const bool vy0 = y0 >= 0 && y0 < a.h, vy1 = y0 + 1 >= 0 && y0 + 1 < a.h;
        const float* base = a.fmap + (size_t)b * a.h * a.w * a.C;
        float* o = a.out + (size_t)pt * a.ld + a.col_off;
        for (int ch = lane * 4; ch < a.C; ch += 256) {
            lq_f4 acc = {0.f, 0.f, 0.f, 0.f};
            // same accumulation order as the native kernel: nw, ne, sw, se
            if (vy0 && vx0) acc += w00 * *reinterpret_cast<const lq_f4*>(base + ((size_t)y0 * a.w + x0) * a.C + ch);
            if (vy0 && vx1) acc += w01 * *reinterpret_cast<const lq_f4*>(base + ((size_t)y0 * a.w + x0 + 1) * a.C + ch);
            if (vy1 && vx0) acc += w10 * *reinterpret_cast<const lq_f4*>(base + ((size_t)(y0 + 1) * a.w + x0) * a.C + ch);
            if (vy1 && vx1) acc += w11 * *reinterpret_cast<const lq_f4*>(base + ((size_t)(y0 + 1) * a.w + x0 + 1) * a.C + ch);
            if ((a.ld & 3) == 0 && (a.col_off & 3) == 0) *reinterpret_cast<lq_f4*>(o + ch) = acc;
            else { o[ch] = acc[0]; o[ch + 1] = acc[1]; o[ch + 2] = acc[2]; o[ch + 3] = acc[3]; }
        }
    }
}

// out[m, col_off + ...] = [x(3), sin(f_0 x)(3), cos(f_0 x)(3), sin(f_1 x)(3), ...], f_k = 2^k
__global__ void __launch_bounds__(256)
pos_encoding_kernel(float* __restrict__ out, const float* __restrict__ pts, long long M, int n_freqs, int ld, int col_off) {
    const int width = 3 * (2 * n_freqs + 1);
    const long long total = M * width;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long m = e / width;
        const int j = (int)(e - m * width);
        const int grp = j / 3, ax = j - grp * 3;
        const float x = pts[m * 3 + ax];
        float v = x;
        if (grp > 0) {
            const float f = (float)(1 << ((grp - 1) >> 1));
            v = ((grp - 1) & 1) ? cosf(f * x) : sinf(f * x);
        }
        out[m * ld + col_off + j] = v;
    }
}

}  // namespace e3dge

using namespace e3dge;

extern "C" int e3dge_local_query(float* out, int ld, int col_off, float* in_img, int mask_ld, int mask_off, float* proj,
                                 const float* pts, const float* calibs, const float* fmap_nhwc, int batch, int64_t n_pts,
                                 int channels, int fh, int fw, e3dge_stream_t stream) {
    E3DGE_REQUIRE(batch >= 0 && n_pts >= 0, "local_query: bad sizes");
    if (batch == 0 || n_pts == 0) return E3DGE_OK;
    E3DGE_REQUIRE(pts && calibs, "local_query: null pointer");
    if (fmap_nhwc) {
        E3DGE_REQUIRE(out != nullptr && channels >= 4 && channels % 4 == 0 && fh >= 1 && fw >= 1, "local_query: feature map needs C %% 4 == 0");
        E3DGE_REQUIRE(ld >= col_off + channels && col_off >= 0, "local_query: output slice [%d, %d) outside ld=%d", col_off, col_off + channels, ld);
        E3DGE_REQUIRE((reinterpret_cast<uintptr_t>(fmap_nhwc) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "local_query: fmap / out must be 16-B aligned");
    }
    E3DGE_REQUIRE(in_img == nullptr || (mask_ld >= 1 && mask_off >= 0 && mask_off < mask_ld), "local_query: bad mask stride");
    LocalQueryK k{};
    k.pts = pts; k.calibs = calibs; k.fmap = fmap_nhwc; k.out = out; k.in_img = in_img; k.proj = proj; k.N = n_pts; k.B = batch;
    k.C = channels; k.h = fh; k.w = fw; k.ld = ld; k.col_off = col_off; k.mask_ld = mask_ld; k.mask_off = mask_off;
    int64_t blocks = ((int64_t)batch * n_pts + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    local_query_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(k);
    return check_launch("local_query");
}

extern "C" int e3dge_pos_encoding(float* out, int ld, int col_off, const float* pts, int64_t n_pts, int n_freqs,
                                  e3dge_stream_t stream) {
    E3DGE_REQUIRE(n_pts >= 0 && n_freqs >= 0 && n_freqs <= 16, "pos_encoding: bad sizes");
    if (n_pts == 0) return E3DGE_OK;
    E3DGE_REQUIRE(out && pts && col_off >= 0 && ld >= col_off + 3 * (2 * n_freqs + 1), "pos_encoding: output slice outside ld");
    const int64_t total = n_pts * 3 * (2 * n_freqs + 1);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    pos_encoding_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(out, pts, n_pts, n_freqs, ld, col_off);
    return check_launch("pos_encoding");
}
