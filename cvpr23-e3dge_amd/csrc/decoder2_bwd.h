// Data gradient of the packed decoder pipeline (round 5; included by decoder2.hip inside namespace e3dge): d image -> d features through
// Decoder.forward (project/models/stylesdf_model.py:742-797) with the generator frozen -- the backward that train_ae.py's stage-1 step
// takes for every sample (trainer.py:1017-1031 puts the pixel loss on pool_256(gen_imgs); trainer.py:728 calls loss.backward()).
// The reference differentiates ModulatedConv2d (:317-362: F.conv2d / F.conv_transpose2d on the modulated weights), Blur / Upsample
// (op/upfirdn2d.py:18-142: the adjoint is upfirdn2d with the flipped kernel and up <-> down), FusedLeakyReLU (op/fused_act.py:19-84:
// grad = 1 of fused_bias_act with the OUTPUT as reference) and ToRGB (:531-541) through autograd.  Here the chain is written out on
// the packed layout, top level first:
//
//     G2[top]   = lrelu'(act2[top]) sqrt 2 . (ToRGB_top^T d img)                                     pk_rgbt_mask_kernel
//     per level u = top .. 0 (resolution R, C channels; the level below has C' channels at R / 2):
//         d rgb[u-1] = Upsample^T d rgb[u]   (4x4 FIR, stride 2)                                     pk_drgb_down_kernel
//         G1[u]   = lrelu'(act1[u]) sqrt 2 . conv3x3(G2[u], flip(w'')^T)                              pkconv_s1_kernel<.., BWD = 1>
//         P[u]    = Blur^T G1[u], split into the four stride-2 phases  ((R/2 + 1)^2 each)             pk_dblur_kernel
//         G2[u-1] = lrelu'(act2[u-1]) sqrt 2 . (conv3x3_stride2(P[u], w''^T) + ToRGB_{u-1}^T d rgb[u-1])   pkconv_down_kernel
//     d features = conv3x3(G2[-1], flip(w''_conv1)^T)                                                 pkconv_s1_kernel<.., BWD = 2>
//
// Every gradient tensor between the kernels is packed exactly as the activations are (16-byte entries of eight f16 hi / lo halves,
// one-entry zero border, one scale exponent per tensor from an a-priori bound), so the convolutions are the forward's LDS-DMA + MFMA
// machinery on transposed weight images (pk_prepack_t_kernel + pk_weights_kernel with swap).  lrelu' comes from the sign of the hi
// half of the packed FORWARD activation (v > 0 <=> hi > 0 unless |v| < 2^-39 of the tensor's bound): no mask tensor is stored.
// Bounds: |conv^T g| <= max|g| max_row sum|w''|, and sum_tap |w''| <= 3 sqrt(sum_tap w''^2) = 3 |s| demod sqrt(wsq) (Cauchy-Schwarz on the
// nine taps, then on co) gives a deterministic operator norm from the column sums of the (co, ci) table the forward's demodulation uses (pk_bwd_bounds_kernel).

// ---- operator norms of the transposed images + ToRGB tables (one block per layer; also clears the call's amax block) ----------------
// max_{b, ci} sum_{co, tap} |w''| <= max_b [ 3 sqrt(sum_co demod^2) max_ci |s[ci]| sqrt(wcol[ci]) ],  wcol[ci] = sum_co wsq[co][ci]
// (Cauchy-Schwarz over the taps, then over co; wcol is a per-weight-update table, so a backward pays O(co + ci) per layer.  The first
// version summed demod[co] sqrt(wsq[co][ci]) over co per (b, ci): 233 us of dependent loads for 13 layers at 1024^2.)
struct PkBndConv { const float* style; const float* demod; const float* wcol; int co, ci; };
struct PkBndRgb { const float* wm; int ci; };
struct PkBndTab { PkBndConv conv[2 * E3DGE_DEC2_MAX_UP + 1]; PkBndRgb rgb[E3DGE_DEC2_MAX_UP + 1]; int n_conv, n_rgb, batch, n_zero; float* out; float* zero; };

__global__ void __launch_bounds__(256) pk_bwd_bounds_kernel(const PkBndTab tab) {
    __shared__ float red[2][4];
    const int layer = blockIdx.x, tid = threadIdx.x;
    for (int i = layer * 256 + tid; i < tab.n_zero; i += gridDim.x * 256) tab.zero[i] = 0.0f;
    auto block_reduce = [&](float v, bool is_max) {          // fixed order: deterministic
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(v, off, kWave); v = is_max ? fmaxf(v, o) : v + o; }
        __syncthreads();
        if ((tid & 63) == 0) red[0][tid >> 6] = v;
        __syncthreads();
        return is_max ? fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3])) : (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    };
    float m = 0.0f;
    if (layer < tab.n_conv) {
        const PkBndConv L = tab.conv[layer];
        for (int b = 0; b < tab.batch; ++b) {
            float d2 = 0.0f, sm = 0.0f;
            for (int co = tid; co < L.co; co += 256) { const float d = L.demod[(size_t)b * L.co + co]; d2 = fmaf(d, d, d2); }
            for (int ci = tid; ci < L.ci; ci += 256) sm = fmaxf(sm, fabsf(L.style[(size_t)b * L.ci + ci]) * sqrtf(L.wcol[ci]));
            d2 = block_reduce(d2, false);
            sm = block_reduce(sm, true);
            m = fmaxf(m, 3.0f * sqrtf(d2) * sm);
        }
    } else {
        const PkBndRgb L = tab.rgb[layer - tab.n_conv];
        for (int b = 0; b < tab.batch; ++b)
            for (int ci = tid; ci < L.ci; ci += 256) {
                const float* w = L.wm + (size_t)b * 3 * L.ci + ci;
                m = fmaxf(m, fabsf(w[0]) + fabsf(w[L.ci]) + fabsf(w[2 * L.ci]));
            }
        m = block_reduce(m, true);
    }
    if (tid == 0) tab.out[layer] = m * 1.0001f;
}

// ---- Upsample^T of a 3-channel image: d skip[iy][ix] = sum_{m,n} d out[2 iy - 1 + m][2 ix - 1 + n] fir[m][n]  (the adjoint of
// upfirdn2d(skip, fir, up = 2, pad = (2, 1)) as pk_torgb_kernel / the fused ToRGB epilogues apply it) + max |.| ------------------------
__global__ void __launch_bounds__(256)
pk_drgb_down_kernel(float* __restrict__ y, float* __restrict__ y_amax, const float* __restrict__ x, const float* __restrict__ fir, int planes, int r) {
    __shared__ float red[4];
    const int h = r >> 1, hw = h * h;
    const int p = blockIdx.x * 256 + threadIdx.x, pl = blockIdx.y;
    float v = 0.0f;
    if (p < hw && pl < planes) {
        const int iy = p / h, ix = p - iy * h;
        const float* xp = x + (int64_t)pl * r * r;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int oy = 2 * iy - 1 + m;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int ox = 2 * ix - 1 + n;
                if (oy >= 0 && oy < r && ox >= 0 && ox < r) v = fmaf(xp[(int64_t)oy * r + ox], fir[m * 4 + n], v);
            }
        }
        y[(int64_t)pl * hw + p] = v;
    }
    float m = fabsf(v);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomic_max_nonneg(y_amax + ((int)(blockIdx.x + blockIdx.y * gridDim.x) & (kAmaxSlots - 1)) * kAmaxStride, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

// ---- top of the chain: G2 = lrelu'(act) sqrt 2 . (ToRGB^T d img), one thread per (pixel, group of eight channels) ---------------------
struct PkRgbtK {
    const float* d_img; const float* wm; const unsigned char* act; unsigned char* y;
    const float* d_amax; const float* rgb_l1; int* out_meta; float* out_amax;
    float act_scale, slope; int B, C, R;
};
constexpr int kRgbtPix = 1;     // (2 and 4 pixels per thread with all loads first measured slower: 50 vs 44 us at 1024^2)
__global__ void __launch_bounds__(256) pk_rgbt_mask_kernel(const PkRgbtK a) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63;
    const float bound = a.act_scale * amax_read(a.d_amax, lane) * a.rgb_l1[0] * 1.001f;
    const unsigned eb = scale_exponent(bound);
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) a.out_meta[0] = (int)eb;
    const float gmul = a.act_scale * pow2_bits(268u - eb);
    // kRgbtPix pixels per thread, every load of the thread first
    const int g = blockIdx.y, b = blockIdx.z, G = a.C >> 3, R = a.R;
    const int64_t hw = (int64_t)R * R, plane = (int64_t)(R + 2) * (R + 2);
    float m = 0.0f;
    u32x4 ah[kRgbtPix];
    float d[kRgbtPix][3];
    int64_t e[kRgbtPix];
    bool in[kRgbtPix];
#pragma unroll
    for (int u = 0; u < kRgbtPix; ++u) {
        const int p = ((int)blockIdx.x * kRgbtPix + u) * 256 + threadIdx.x;
        in[u] = p < R * R;
        const int pc = in[u] ? p : 0;
        const int y = pc / R, x = pc - y * R;
        e[u] = ((int64_t)(b * G + g) * 2) * plane + (int64_t)(y + 1) * (R + 2) + x + 1;
        ah[u] = reinterpret_cast<const u32x4*>(a.act)[e[u]];
#pragma unroll
        for (int c = 0; c < 3; ++c) d[u][c] = a.d_img[((int64_t)b * 3 + c) * hw + pc];
    }
    const float* __restrict__ w = a.wm + (size_t)b * 3 * a.C + 8 * g;              // (block-uniform: scalar loads)
#pragma unroll
    for (int u = 0; u < kRgbtPix; ++u) {
        if (!in[u]) continue;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = w[j] * d[u][0];
            t = fmaf(w[a.C + j], d[u][1], t);
            t = fmaf(w[2 * a.C + j], d[u][2], t);
            const unsigned hb = (ah[u][j >> 1] >> (16 * (j & 1))) & 0xffffu;
            t = (hb - 1u < 0x7fffu) ? t : t * a.slope;
            v[j] = t * gmul;
            m = fmaxf(m, fabsf(v[j]));
        }
        u32x4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) SPLIT2_TO(v[2 * q], v[2 * q + 1], hi[q], lo[q]);
        reinterpret_cast<u32x4*>(a.y)[e[u]] = hi;
        reinterpret_cast<u32x4*>(a.y)[e[u] + plane] = lo;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
    if (lane == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomic_max_nonneg(a.out_amax + ((int)(blockIdx.x + 7 * blockIdx.y) & (kAmaxSlots - 1)) * kAmaxStride,
                          fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * pow2_bits(eb - 14u));
}

// ---- Blur^T + stride-2 phase split -------------------------------------------------------------------------------------------------------
// Forward: pre = upfirdn2d(T, K, pad = (1, 1)), T (2h+1)^2 -> (2h)^2:  pre[y][x] = sum_{m,n} T[y + 2 - m][x + 2 - n] K[m][n]  (the op
// correlates with the flipped kernel).  Adjoint: d T[Y][X] = sum_{m,n} g[Y + m - 2][X + n - 2] K[m][n], g = 0 outside [0, 2h)^2.
// Output: P (B, C/8, 2, 4, h+2, h+2) entries, P[..][2 py + px][i][j] = d T[2i + py][2j + px] (0 beyond 2h): the stride-2 convolution
// that follows reads tap (ky, kx) from phase (ky & 1, kx & 1) at offset (ky >> 1, kx >> 1) -- unit-stride patches for its LDS-DMA.
// A block owns 8 x 32 phase positions of one channel group; a thread all four phases of one position (coalesced 16-byte stores into
// four planes) from the 5 x 5 neighbourhood in LDS (fp32, even / odd columns and channel halves in separate planes: lane-linear reads).
struct PkDblurK {
    const unsigned char* g; const int* in_meta; const float* in_amax; const float* fir;
    unsigned char* p; int* out_meta; float* out_amax;
    int B, C, R, tiles_x, tiles_y;
};
constexpr int kDbTI = 8, kDbTJ = 32, kDbRows = 2 * kDbTI + 3, kDbCols = 2 * kDbTJ + 3, kDbCP = 36;
__global__ void __launch_bounds__(256) pk_dblur_kernel(const PkDblurK a) {
    __shared__ f32x4 sm[kDbRows][2][2][kDbCP];           // [row][column parity][channel half][column / 2]
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int R = a.R, h = R >> 1, G = a.C >> 3;
    int t = blockIdx.x;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y; t /= a.tiles_y;
    const int g = t % G, b = t / G;
    const int i0 = ty * kDbTI, j0 = tx * kDbTJ;
    float ksum = 0.0f, K[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { K[i] = a.fir[i]; ksum += fabsf(K[i]); }
    const float bound = amax_read(a.in_amax, lane) * ksum * 1.001f;
    const unsigned eb = scale_exponent(bound);
    if (blockIdx.x == 0 && tid == 0) a.out_meta[0] = (int)eb;
    const float inv_in = pow2_bits((unsigned)a.in_meta[0] - 14u);              // 2^(eb_in - 141)
    const float sc = pow2_bits(268u - eb);
    {
        const int64_t plane = (int64_t)(R + 2) * (R + 2);
        const u32x4* __restrict__ gp = reinterpret_cast<const u32x4*>(a.g) + ((int64_t)(b * G + g) * 2) * plane;
        // all of a thread's loads first, from clamped addresses (round 6: a load per trip of a loop that also unpacks and writes LDS kept
        // ~24 KB per CU in flight -- the kernel streamed at 3.8 TB/s where the library's element-wise kernels reach 5-7)
        constexpr int NIT = (kDbRows * kDbCols + 255) / 256;
        u32x4 hi[NIT], lo[NIT];
        bool in[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it;
            const int r = idx / kDbCols, cx = idx - r * kDbCols;
            const int Y = 2 * i0 - 2 + r, X = 2 * j0 - 2 + cx;
            in[it] = idx < kDbRows * kDbCols && Y >= 0 && Y < R && X >= 0 && X < R;
            const int64_t e = in[it] ? (int64_t)(Y + 1) * (R + 2) + X + 1 : 0;
            hi[it] = gp[e]; lo[it] = gp[e + plane];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it;
            if (idx >= kDbRows * kDbCols) break;
            const int r = idx / kDbCols, cx = idx - r * kDbCols;
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (in[it]) {
                v0[0] = (f16lo(hi[it][0]) + f16lo(lo[it][0])) * inv_in; v0[1] = (f16hi(hi[it][0]) + f16hi(lo[it][0])) * inv_in;
                v0[2] = (f16lo(hi[it][1]) + f16lo(lo[it][1])) * inv_in; v0[3] = (f16hi(hi[it][1]) + f16hi(lo[it][1])) * inv_in;
                v1[0] = (f16lo(hi[it][2]) + f16lo(lo[it][2])) * inv_in; v1[1] = (f16hi(hi[it][2]) + f16hi(lo[it][2])) * inv_in;
                v1[2] = (f16lo(hi[it][3]) + f16lo(lo[it][3])) * inv_in; v1[3] = (f16hi(hi[it][3]) + f16hi(lo[it][3])) * inv_in;
            }
            sm[r][cx & 1][0][cx >> 1] = v0;
            sm[r][cx & 1][1][cx >> 1] = v1;
        }
    }
    __syncthreads();
    const int ti = tid >> 5, tj = tid & 31;
    f32x4 acc[2][4];                                     // [channel half][phase]
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) acc[ch][ph] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            f32x4 row[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) row[c] = sm[2 * ti + r][c & 1][ch][tj + (c >> 1)];
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int m = r - py;
                if (m < 0 || m > 3) continue;
#pragma unroll
                for (int px = 0; px < 2; ++px)
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const float k = K[m * 4 + n];
                        const f32x4 x = row[n + px];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[ch][2 * py + px][j] = fmaf(x[j], k, acc[ch][2 * py + px][j]);
                    }
            }
        }
    }
    const int i = i0 + ti, j = j0 + tj;
    float m = 0.0f;
    if (i <= h && j <= h) {
        const int64_t planep = (int64_t)(h + 2) * (h + 2);
        u32x4* __restrict__ pp = reinterpret_cast<u32x4*>(a.p) + ((int64_t)(b * G + g) * 8) * planep + (int64_t)i * (h + 2) + j;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const bool in = (2 * i + (ph >> 1) <= 2 * h) && (2 * j + (ph & 1) <= 2 * h);
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                v[q] = in ? acc[q >> 2][ph][q & 3] * sc : 0.0f;
                m = fmaxf(m, fabsf(v[q]));
            }
            u32x4 hi, lo;
#pragma unroll
            for (int q = 0; q < 4; ++q) SPLIT2_TO(v[2 * q], v[2 * q + 1], hi[q], lo[q]);
            pp[(int64_t)ph * planep] = hi;
            pp[(int64_t)(4 + ph) * planep] = lo;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
    if (lane == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0)
        atomic_max_nonneg(a.out_amax + ((int)blockIdx.x & (kAmaxSlots - 1)) * kAmaxStride, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * pow2_bits(eb - 14u));
}

// ---- stride-2 3x3 convolution of the phase planes: the data gradient of conv_transpose2d(stride 2) --------------------------------------
// d x[ci][i][j] = sum_{co, ky, kx} w''[co][ci][ky][kx] d T[co][2i + ky][2j + kx] = sum_taps w'' P[ky & 1, kx & 1][co][i + (ky >> 1)][j + (kx >> 1)].
// The input is four times the output's pixels, so a tile's patch is sixteen planes ((k-half, hi | lo) x four phases): a workgroup of
// four waves owns 4 x 32 output pixels x 32 NCT channels (one row per wave), 16 x 5 x 33 entries + NCT weight slabs per stage, two
// stages (158 KB with NCT = 2: one workgroup per CU; the K loop is LDS-DMA bound at ~84 KB per 1.7 k cycles of MFMAs).  Steps, barrier
// and DMA issue as in pkconv_s1_kernel; pieces are dealt out statically: slot sl of wave w is piece w + 4 sl -- patch pieces first
// (plane = w + 4 (sl % 4), 64-entry round = sl / 4: the round is a compile-time constant, the plane only enters scalar address
// arithmetic), then the weight pieces.  Epilogue = bwd_tile_epilogue (ToRGB^T of the level below + lrelu' + split + store).
template <int NCT>
__global__ void __launch_bounds__(256) pkconv_down_kernel(const PkConvK a) {
    constexpr int NW = 4, NT = 256, TH = 4, TW = 32, PH = TH + 1, PW = TW + 1, NPIX = PH * PW, NPP = (NPIX + 63) / 64;
    constexpr int NPL = 16, XPLANE = NPIX * 16, XST = NPL * XPLANE, WST = NCT * kPkSlab, STAGE = XST + WST;
    constexpr int NXS = (NPL / NW) * NPP, NWP = NCT * 18, NWS = (NWP + NW - 1) / NW, NSLOT = NXS + NWS, PPT = (NSLOT + E3DGE_PK_ISSUE_TAPS - 1) / E3DGE_PK_ISSUE_TAPS;
    static_assert(NPL % NW == 0, "planes are dealt out four per round");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pk[];
    float* const tab = reinterpret_cast<float*>(smem_pk + 2 * STAGE);        // [3][32 NCT]: (scale W) s of this workgroup's channels
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int my_tiles = (a.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nsteps = my_tiles * a.n_chunks;
    if (nsteps <= 0) return;
    const int HP = a.H + 2, WP = a.W + 2, G = a.Ci >> 3;
    const int64_t plane_b = (int64_t)HP * WP * 16;                          // bytes of one (g, hl, phase) plane of P

    const float oscale = pow2_bits((unsigned)a.in_meta[0] - 21u);
    const float rg = a.rgbt_d ? amax_read(a.rgbt_amax, lane) * a.rgbt_l1[0] : 0.0f;
    const float bound = a.act_scale * (amax_read(a.in_amax, lane) * a.bwd_wl1[0] * 1.002f + rg) * 1.001f;
    const unsigned eb_out = scale_exponent(bound);
    const float sc_out = pow2_bits(268u - eb_out);
    if (blockIdx.x == 0 && tid == 0) a.out_meta[0] = (int)eb_out;

    struct Pos { int k, c, b, cb, ty, tx; };
    auto tile_of = [&](Pos& p) {
        if (p.k >= my_tiles) return;
        int L = xcd_logical((int)blockIdx.x + p.k * (int)gridDim.x, a.n_tiles);
        p.cb = L % a.co_blocks; L /= a.co_blocks;
        p.tx = L % a.tiles_x; L /= a.tiles_x;
        p.ty = L % a.tiles_y; p.b = L / a.tiles_y;
    };
    auto advance = [&](Pos& p) { if (++p.c == a.n_chunks) { p.c = 0; ++p.k; tile_of(p); } };

    // static per-lane piece data: patch entry (row << 8 | column) of each 64-entry round, source offset of each weight slot
    uint32_t pk[NPP], wvo[NWS];
#pragma unroll
    for (int pp = 0; pp < NPP; ++pp) {
        const int e = pp * 64 + lane, prow = e / PW, pcol = e - prow * PW;
        pk[pp] = e < NPIX ? (uint32_t)(prow << 8 | pcol) : 0xffffffffu;
        asm volatile("" : "+v"(pk[pp]));
    }
#pragma unroll
    for (int j = 0; j < NWS; ++j) {
        const int i = wave + j * NW, ct = i / 18, pc = i - ct * 18;
        wvo[j] = (uint32_t)lane * 16u + (uint32_t)((ct * a.n_chunks * 18 + pc) * 1024);
        asm volatile("" : "+v"(wvo[j]));
    }
    struct Src { uint32_t wlo, whi, xlo, xhi; uint32_t vo[NPP]; };
    auto src_of = [&](const Pos& ps) {
        const uint64_t w = reinterpret_cast<uint64_t>(a.wimg + (int64_t)ps.b * a.wimg_bytes + ((int64_t)(ps.cb * NCT) * a.n_chunks + ps.c) * kPkSlab);
        const uint64_t x = reinterpret_cast<uint64_t>(a.x + ((int64_t)(ps.b * G + 2 * ps.c) * 8) * plane_b);
        Src sc;
        sc.wlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w); sc.whi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(w >> 32));
        sc.xlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x); sc.xhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(x >> 32));
        asm volatile("" : "+s"(sc.wlo), "+s"(sc.whi), "+s"(sc.xlo), "+s"(sc.xhi));
        const int gy0 = ps.ty * TH, gx0 = ps.tx * TW;
#pragma unroll
        for (int pp = 0; pp < NPP; ++pp) {
            // (clamped into the plane: tiles that overhang the image read entries whose results are never stored)
            const int gy = min(gy0 + (int)(pk[pp] >> 8), HP - 1), gx = min(gx0 + (int)(pk[pp] & 255u), WP - 1);
            sc.vo[pp] = pk[pp] == 0xffffffffu ? 0xffffffffu : (uint32_t)(gy * WP + gx) * 16u;
        }
        return sc;
    };
    const uint32_t plb = (uint32_t)plane_b;
    auto issue = [&](const Src& sc, uint32_t xl, int s_lo, int s_hi) {
        const void* wsrc = reinterpret_cast<const void*>((uint64_t)sc.whi << 32 | sc.wlo);
        const uint64_t xs = (uint64_t)sc.xhi << 32 | sc.xlo;
#pragma unroll
        for (int sl = s_lo; sl < s_hi; ++sl) {
            if (sl < NXS) {
                const int pp = sl / (NPL / NW), pl = wave + NW * (sl % (NPL / NW));
                const void* xsrc = reinterpret_cast<const void*>(xs + (uint64_t)pl * plb);
                const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(xl + (uint32_t)(pl * XPLANE + pp * 1024)));
                if (sc.vo[pp] != 0xffffffffu) glds16_saddr<0>(uniform_ptr(xsrc), sc.vo[pp], dst);
            } else {
                const int j = sl - NXS;
                if ((j + 1) * NW <= NWP || wave + j * NW < NWP)
                    glds16_saddr<0>(wsrc, wvo[j], xl + (uint32_t)(XST + (wave + j * NW) * 1024));
            }
        }
    };
    auto stage_lds = [&](int stage) {
        uint32_t xl = lds_u32(smem_pk) + (uint32_t)(stage * STAGE);
        asm volatile("" : "+s"(xl));
        return xl;
    };

    Pos p_cur{0, 0, 0, 0, 0, 0};
    tile_of(p_cur);
    Pos p_nx1 = p_cur; advance(p_nx1);
    issue(src_of(p_cur), stage_lds(0), 0, NSLOT);

    f32x16 acc[NCT];
    uint2 mwp[NCT][4];
    float drp[3] = {0.0f, 0.0f, 0.0f};
    float amax_l = 0.0f;
    int b_tab = -1, cb_tab = -1;
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        const bool has_next = step + 1 < nsteps, last_chunk = p_cur.c == a.n_chunks - 1;
        const Src src_nx = src_of(p_nx1);
        const uint32_t xl_nx = stage_lds(cur ^ 1);
        // the tile's sign words / d rgb values: requested one step before its last chunk, pinned behind that chunk's barrier (see bwd_mask_load)
        if (p_cur.c == a.n_chunks - 2) {
            const int oy = p_cur.ty * TH + wave, ox = p_cur.tx * TW + col;
            drp[0] = drp[1] = drp[2] = 0.0f;
            if (a.rgbt_d) {
                const float* dp = a.rgbt_d + (int64_t)p_cur.b * 3 * a.H * a.W + (unsigned)(min(oy, a.H - 1) * a.W + min(ox, a.W - 1));
#pragma unroll
                for (int c = 0; c < 3; ++c) drp[c] = dp[(int64_t)c * a.H * a.W];
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) bwd_mask_load(a, p_cur.b, p_cur.cb * NCT + ct, oy, ox, half, mwp[ct]);
        }
        if (last_chunk) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) asm volatile("" : "+v"(mwp[ct][g4].x), "+v"(mwp[ct][g4].y));
            asm volatile("" : "+v"(drp[0]), "+v"(drp[1]), "+v"(drp[2]));
        }
        if (p_cur.c == 0) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = zero16();
            // (every wave has finished the previous tile's epilogue before this step's barrier; readers are >= 1 barrier away: n_chunks >= 2)
            if (a.rgbt_d && (p_cur.b != b_tab || p_cur.cb != cb_tab)) {
                b_tab = p_cur.b; cb_tab = p_cur.cb;
                for (int i = tid; i < 3 * 32 * NCT; i += NT) {
                    const int c = i / (32 * NCT), ch = i - c * 32 * NCT;
                    tab[i] = a.rgb_wm[((size_t)b_tab * 3 + c) * a.Co + cb_tab * 32 * NCT + ch];
                }
            }
        }
        {
            const unsigned char* xb = smem_pk + cur * STAGE;
            const unsigned char* wb = xb + XST + lane * 16;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3, ph = (ky & 1) * 2 + (kx & 1);
                const int pix = (wave + (ky >> 1)) * PW + col + (kx >> 1);
                const u32x4 bh = *reinterpret_cast<const u32x4*>(xb + (size_t)((half * 2 + 0) * 4 + ph) * XPLANE + pix * 16);
                const u32x4 bl = *reinterpret_cast<const u32x4*>(xb + (size_t)((half * 2 + 1) * 4 + ph) * XPLANE + pix * 16);
                u32x4 ah[NCT], al[NCT];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    ah[ct] = *reinterpret_cast<const u32x4*>(wb + ct * kPkSlab + (tap * 2 + 0) * 1024);
                    al[ct] = *reinterpret_cast<const u32x4*>(wb + ct * kPkSlab + (tap * 2 + 1) * 1024);
                }
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = mfma16(ah[ct], bh, acc[ct]);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = mfma16(al[ct], bh, acc[ct]);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = mfma16(ah[ct], bl, acc[ct]);
                if (has_next && tap * PPT < NSLOT) issue(src_nx, xl_nx, tap * PPT, min((tap + 1) * PPT, NSLOT));
                if (tap % 3 == 2) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (last_chunk) {
            const int oy = p_cur.ty * TH + wave, ox = p_cur.tx * TW + col;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                bwd_tile_epilogue<false>(a, acc[ct], p_cur.b, p_cur.cb * NCT + ct, oy, ox, half, a.rgbt_d ? tab : nullptr, 32 * NCT,
                                         p_cur.cb * 32 * NCT, drp, mwp[ct], oscale, sc_out, amax_l);
        }
        p_cur = p_nx1;
        advance(p_nx1);
    }
    if (a.out_amax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax_l = fmaxf(amax_l, __shfl_xor(amax_l, off, kWave));
        if (lane == 0) atomic_max_nonneg(a.out_amax + (((int)blockIdx.x * NW + wave) & (kAmaxSlots - 1)) * kAmaxStride, amax_l / sc_out);
    }
}


// ---- d latent (round 5, optional): the gradient to the decoder's W+ latent WITHOUT a weight-gradient contraction ---------------------------
// For a modulated 3x3 layer  w'' = (scale W) s[ci] demod[co],  demod = rsqrt(sum_{ci,t} ((scale W) s)^2 + eps)  (stylesdf_model.py:317-326):
//     dL/ds[ci] = (1 / s[ci]) sum_p x[ci,p] dx[ci,p]  -  s[ci] sum_co demod[co]^2 wsq[co][ci] (sum_p g[co,p] y[co,p])
// with x the layer's input, dx ITS data gradient (this layer's contribution only, before the producer's lrelu'), g = dL/d(pre-activation) of the
// layer's output and y its convolution output before noise and bias (for the up-sampling layers after the blur: <Blur^T g, T> = <g, Blur T>).
// The first sum is sum_{co,t} dL/dw'' w'' regrouped by input channel, the second the same regrouped by output channel (the demodulation's own
// derivative): the (co, ci, 9) weight gradient itself -- a GEMM over all pixels -- is never formed.  ToRGB (no demodulation):
// dL/ds[ci] = (1 / s[ci]) sum_p act[ci,p] v[ci,p], v = ToRGB^T d rgb.  Everything is per-channel dot products of tensors the backward
// has left in its workspace: for every activation tensor A (packed forward) with G = dL/d(pre) (packed gradient), mask gain m and v:
//     dx = G / m - v,   S1 = sum_p A dx  (-> the 3x3 layer that CONSUMES A),   S2 = sum_p G y  (-> the layer that PRODUCED A),   S3 = sum_p A v  (-> A's ToRGB)
// (pk_dstyle_sums_kernel: one streaming pass over A and G, per-chunk partials; pk_dstyle_fold_kernel: the chunks, fixed order), then dL/ds per
// modulation row (pk_dstyle_kernel) and dL/dlatent = modulation^T dL/ds (pk_dlatent_kernel; EqualLinear :234-244).
struct PkDsSumsK {
    const unsigned char* act; const unsigned char* g; const int* act_meta; const int* g_meta;
    const float* noise; const float* noise_w; const float* bias;      // of the PRODUCING layer (noise (noise_batch, R, R) or null)
    const float* wm; const float* drgb;                                 // ToRGB table (B, 3, C) and d rgb (B, 3, R, R), or null
    float* part;                                                        // (B * C/8, n_chunks, 3, 8)
    float slope, act_scale; int C, R, noise_batch, n_chunks, chunk;
};
__global__ void __launch_bounds__(256) pk_dstyle_sums_kernel(const PkDsSumsK a) {
    __shared__ float red[4][24];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int chunk = blockIdx.x, g = blockIdx.y, b = blockIdx.z, G = a.C >> 3, R = a.R;
    const int64_t plane = (int64_t)(R + 2) * (R + 2), hw = (int64_t)R * R;
    const u32x4* __restrict__ ap = reinterpret_cast<const u32x4*>(a.act) + ((int64_t)(b * G + g) * 2) * plane;
    const u32x4* __restrict__ gp = reinterpret_cast<const u32x4*>(a.g) + ((int64_t)(b * G + g) * 2) * plane;
    const float inv_a = pow2_bits((unsigned)a.act_meta[0] - 14u), inv_g = pow2_bits((unsigned)a.g_meta[0] - 14u);
    const float nw = a.noise ? a.noise_w[0] : 0.0f;
    const float* __restrict__ nz = a.noise ? a.noise + (int64_t)(a.noise_batch > 1 ? b : 0) * hw : nullptr;
    float bs[8], w[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bs[j] = a.bias[8 * g + j];
#pragma unroll
        for (int c = 0; c < 3; ++c) w[c][j] = a.wm ? a.wm[((size_t)b * 3 + c) * a.C + 8 * g + j] : 0.0f;
    }
    const float i_pos = 1.0f / a.act_scale, i_neg = 1.0f / (a.act_scale * a.slope);          // (a division per element is ~10 instructions)
    float s1[8] = {}, s2[8] = {}, s3[8] = {};
    const int p_end = min((chunk + 1) * a.chunk, (int)hw);
#pragma unroll 4                                                             // (four pixels' sixteen 16-byte loads in flight per thread)
    for (int p = chunk * a.chunk + tid; p < p_end; p += 256) {
        const int y = p / R, x = p - y * R;
        const int64_t e = (int64_t)(y + 1) * (R + 2) + x + 1;
        const u32x4 ah = ap[e], al = ap[e + plane], gh = gp[e], gl = gp[e + plane];
        float d[3] = {0.f, 0.f, 0.f};
        if (a.drgb) {
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c] = a.drgb[((int64_t)b * 3 + c) * hw + p];
        }
        const float nzv = nz ? nw * nz[p] : 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned hb = (ah[j >> 1] >> (16 * (j & 1))) & 0xffffu;
            const float im = (hb - 1u < 0x7fffu) ? i_pos : i_neg;                       // the backward's own branch test
            const float av = ((j & 1) ? f16hi(ah[j >> 1]) + f16hi(al[j >> 1]) : f16lo(ah[j >> 1]) + f16lo(al[j >> 1])) * inv_a;
            const float gv = ((j & 1) ? f16hi(gh[j >> 1]) + f16hi(gl[j >> 1]) : f16lo(gh[j >> 1]) + f16lo(gl[j >> 1])) * inv_g;
            const float v = fmaf(w[2][j], d[2], fmaf(w[1][j], d[1], w[0][j] * d[0]));
            const float yv = av * im - nzv - bs[j];
            s1[j] = fmaf(av, gv * im - v, s1[j]);
            s2[j] = fmaf(gv, yv, s2[j]);
            s3[j] = fmaf(av, v, s3[j]);
        }
    }
    float vals[24];
#pragma unroll
    for (int j = 0; j < 8; ++j) { vals[j] = s1[j]; vals[8 + j] = s2[j]; vals[16 + j] = s3[j]; }
#pragma unroll
    for (int i = 0; i < 24; ++i) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vals[i] += __shfl_xor(vals[i], off, kWave);
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 24; ++i) red[wv][i] = vals[i];
    }
    __syncthreads();
    if (tid < 24) a.part[((int64_t)(b * G + g) * a.n_chunks + chunk) * 24 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// conv1's input is the fp32 feature map: S1[b][ci] = sum_p features d features, one block per (ci, b)
__global__ void __launch_bounds__(256) pk_dot_planes_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ dx, int C, int hw) {
    __shared__ float red[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const float* xp = x + ((int64_t)b * C + c) * hw;
    const float* dp = dx + ((int64_t)b * C + c) * hw;
    float s = 0.0f;
    for (int p = threadIdx.x; p < hw; p += 256) s = fmaf(xp[p], dp[p], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, kWave);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[(int64_t)b * C + c] = (red[0] + red[1]) + (red[2] + red[3]);
}

// The fold of the chunk partials, one block per (tensor, sample, channel group): sums[(b * G + g) * 24 + v] = sum over chunks, ten slices of the
// chunks summed by ten thread rows and folded in fixed order
struct PkDsFoldT { const float* part; float* sums; int groups, n_chunks, first_block; };
struct PkDsFoldK { PkDsFoldT t[2 * E3DGE_DEC2_MAX_UP + 2]; int n_tensors; };
__global__ void __launch_bounds__(256) pk_dstyle_fold_kernel(const PkDsFoldK a) {
    __shared__ float red[10][24];
    int ti = 0;
    while (ti + 1 < a.n_tensors && (int)blockIdx.x >= a.t[ti + 1].first_block) ++ti;
    const PkDsFoldT t = a.t[ti];
    const int bg = (int)blockIdx.x - t.first_block, tid = threadIdx.x;         // bg = b * groups + g
    if (tid < 240) {
        const int v = tid % 24, sl = tid / 24;
        const float* __restrict__ pp = t.part + (int64_t)bg * t.n_chunks * 24 + v;
        float acc = 0.0f;
        for (int c = sl; c < t.n_chunks; c += 10) acc += pp[(int64_t)c * 24];
        red[sl][v] = acc;
    }
    __syncthreads();
    if (tid < 24) {
        float acc = red[0][tid];
#pragma unroll
        for (int sl = 1; sl < 10; ++sl) acc += red[sl][tid];
        t.sums[(int64_t)bg * 24 + tid] = acc;
    }
}

// dL/ds of one modulation row (table rows of Decoder._mod_layers), one block per (row, sample):  ds[(b * n_rows + t) * 1024 + ci]
struct PkDsRow { const float* p1; const float* p2; int sum1; };               // folded sums of the consumed / produced tensor; sum1: 0 = S1, 2 = S3, 3 = p1 is a plain (b, ci) array
struct PkDsK { const E3dgeModLayer* tab; PkDsRow row[3 * E3DGE_DEC2_MAX_UP + 2]; int n_rows; float* ds; };
__global__ void __launch_bounds__(256) pk_dstyle_kernel(const PkDsK a) {
    __shared__ float r2[1024];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const E3dgeModLayer L = a.tab[t];
    const PkDsRow rw = a.row[t];
    const float* __restrict__ s = L.style_out + (size_t)b * L.ci;
    const bool conv = L.demod_out != nullptr;
    if (conv) {                                                             // r2[co] = demod^2 sum_p g y
        for (int co = tid; co < L.co; co += 256) {
            const float v = rw.p2[((int64_t)b * (L.co >> 3) + (co >> 3)) * 24 + 8 + (co & 7)];
            const float dm = L.demod_out[(size_t)b * L.co + co];
            r2[co] = dm * dm * v;
        }
    }
    __syncthreads();
    for (int ci = tid; ci < L.ci; ci += 256) {
        const float r1 = rw.sum1 == 3 ? rw.p1[(int64_t)b * L.ci + ci] : rw.p1[((int64_t)b * (L.ci >> 3) + (ci >> 3)) * 24 + 8 * rw.sum1 + (ci & 7)];
        const float sv = s[ci];
        float d = fabsf(sv) > 1e-30f ? r1 / sv : 0.0f;
        if (conv) {
            float tb[4] = {0.f, 0.f, 0.f, 0.f};                              // four independent chains (co is a multiple of 32), folded in fixed order
            for (int co = 0; co < L.co; co += 16) {                          // sixteen loads in flight: the loop is latency-, not bandwidth-bound
                float w[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) w[q] = L.wsq[(size_t)(co + q) * L.ci + ci];
#pragma unroll
                for (int q = 0; q < 16; ++q) tb[q & 3] = fmaf(r2[co + q], w[q], tb[q & 3]);
            }
            d -= sv * ((tb[0] + tb[1]) + (tb[2] + tb[3]));
        }
        a.ds[((int64_t)b * a.n_rows + t) * 1024 + ci] = d;
    }
}

// dL/dlatent[b][li][k] = sum over the table rows with latent_index == li of lin_scale sum_ci dL/ds[ci] mod_weight[ci][k]  (EqualLinear^T);
// one block per (latent row, sample, 64 columns k): four thread rows take a quarter of the input channels each, folded in fixed order
struct PkDlatK { const E3dgeModLayer* tab; const float* ds; int n_rows, n_latent, style_dim; float* d_latent; };
__global__ void __launch_bounds__(256) pk_dlatent_kernel(const PkDlatK a) {
    __shared__ float red[4][64];
    const int li = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, kk = tid & 63, q = tid >> 6;
    const int k = (int)blockIdx.z * 64 + kk;
    float acc = 0.0f;
    for (int t = 0; t < a.n_rows; ++t) {
        const E3dgeModLayer L = a.tab[t];
        if (L.latent_index != li) continue;                                // (block-uniform)
        const float* __restrict__ ds = a.ds + ((int64_t)b * a.n_rows + t) * 1024;
        const int per = (L.ci + 3) >> 2, c0 = q * per, c1 = min(L.ci, c0 + per);
        float v = 0.0f;
        if (k < a.style_dim) {
            int ci = c0;
            for (; ci + 8 <= c1; ci += 8) {                                // eight loads in flight (latency-bound otherwise); same summation order
                float w[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) w[q] = L.mod_weight[(size_t)(ci + q) * a.style_dim + k];
#pragma unroll
                for (int q = 0; q < 8; ++q) v = fmaf(ds[ci + q], w[q], v);
            }
            for (; ci < c1; ++ci) v = fmaf(ds[ci], L.mod_weight[(size_t)ci * a.style_dim + k], v);
        }
        acc = fmaf(v, L.lin_scale, acc);
    }
    red[q][kk] = acc;
    __syncthreads();
    if (q == 0 && k < a.style_dim) a.d_latent[((int64_t)b * a.n_latent + li) * a.style_dim + k] = (red[0][kk] + red[1][kk]) + (red[2][kk] + red[3][kk]);
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------------
template <int NCT>
static int launch_down(PkConvK k, hipStream_t st, const char* what) {
    constexpr int STAGE = 16 * (5 * 33) * 16 + NCT * kPkSlab, lds = 2 * STAGE + 3 * 32 * NCT * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    E3DGE_REQUIRE(k.Co % (32 * NCT) == 0 && k.n_chunks >= 2, "%s: Co=%d not a multiple of %d, or fewer than 32 input channels", what, k.Co, 32 * NCT);      // (only NCT = 2 is instantiated)
    E3DGE_REQUIRE(k.y && k.out_meta && k.mask_act && k.bwd_wl1 && k.in_amax && (!k.rgbt_d || (k.rgb_wm && k.rgbt_amax && k.rgbt_l1)), "%s: missing pointer", what);
    k.tiles_y = (k.H + 3) / 4;
    k.tiles_x = (k.W + 31) / 32;
    k.co_blocks = k.Co / (32 * NCT);
    const int64_t n_tiles = (int64_t)k.B * k.co_blocks * k.tiles_y * k.tiles_x;
    E3DGE_REQUIRE(n_tiles < ((int64_t)1 << 30), "%s: too many tiles", what);
    k.n_tiles = (int)n_tiles;
    auto fn = &pkconv_down_kernel<NCT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e));
    int grid = 256 * ((160 * 1024) / lds >= 2 ? 2 : 1);
    if (grid > k.n_tiles) grid = k.n_tiles;
    fn<<<dim3((unsigned)grid), dim3(256), lds, st>>>(k);
    return check_launch(what);
}

// stride-1 data-gradient convolution: the forward's tile shapes (conv_s1), the table of ToRGB^T in LDS when the level has one
template <int BWD>
static int conv_s1_bwd(PkConvK k, hipStream_t st) {
    int v = shape_override("E3DGE_DEC2_S1_BWD");
    const int64_t px = (int64_t)k.H * k.W;
    if (v < 0) {
        if (k.Co % 64 != 0) v = 3;
        else if (px <= 64 * 64) v = 0;
        else if (px <= 128 * 128) v = 1;
        else v = 2;
        if (v == 2 && BWD == 1 && 158208 + 16 * k.Co > 160 * 1024) v = 3;      // (the 64-channel 8 x 64 tile + table would not fit)
    }
    if (k.Co % 64 != 0 && v != 3) v = 3;
    switch (v) {
        case 0: return launch_s1<1, 1, 1, 2, 4, 1, 0, BWD>(k, st, "dec2 conv^T<64co,4x32>");
        case 1: return launch_s1<1, 1, 2, 2, 4, 1, 0, BWD>(k, st, "dec2 conv^T<64co,4x64>");
        case 2: return launch_s1<2, 1, 2, 1, 8, 1, 0, BWD>(k, st, "dec2 conv^T<64co,8x64>");
        default: return launch_s1<1, 1, 2, 1, 8, 1, 0, BWD>(k, st, "dec2 conv^T<32co,8x64>");
    }
}
