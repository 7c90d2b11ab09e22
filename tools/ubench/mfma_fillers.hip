// µbench: cycles per MFMA when N independent filler instructions (v_fma_f32 or ds_read_b128) sit between consecutive
// MFMAs; one wave per SIMD; f16 32x32x16 and f32 32x32x2; 1 or 2 accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int NACC, bool F16, int NFILL, bool LDSFILL>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int iters) {
    __shared__ f32x4 lds[1024];
    lds[threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f}; lds[threadIdx.x + 256] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    f32x16 acc[2];
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float av = threadIdx.x * 1e-3f, bv = 1.0f + threadIdx.x * 1e-4f;
    half8 ah, bh; for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)(av + j); bh[j] = (_Float16)(bv - j); }
    float f[8]; for (int j = 0; j < 8; ++j) f[j] = av + j;
    f32x4 lsum = {0, 0, 0, 0};
    const f32x4* lp = lds + (threadIdx.x & 63);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if (F16) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[u % NACC], 0, 0, 0);
            else     acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[u % NACC], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NFILL; ++n) {
                if (LDSFILL) { f32x4 v = lp[((u * NFILL + n) & 7) * 64]; asm volatile("" :: "v"(v)); }
                else f[n % 8] = __builtin_fmaf(f[n % 8], 1.0001f, 0.5f);     // 8 independent chains
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int j = 0; j < 8; ++j) s += f[j];
    out[blockIdx.x * 256 + threadIdx.x] = s + lsum[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC, bool F16, int NFILL, bool LDSFILL> void run() {
    static float* out = nullptr; static long long* cyc = nullptr;
    if (!out) { (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8); }
    const int iters = 32;
    k<NACC, F16, NFILL, LDSFILL><<<256, 256>>>(out, cyc, iters); (void)hipDeviceSynchronize();
    k<NACC, F16, NFILL, LDSFILL><<<256, 256>>>(out, cyc, iters); (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s %dacc fill=%d%s: %.1f cyc/MFMA\n", F16 ? "f16" : "f32", NACC, NFILL, LDSFILL ? "(ds_read_b128)" : "(v_fma)", c / (double)(iters * 32));
}
int main() {
    run<1, true, 0, false>(); run<1, true, 2, false>(); run<1, true, 4, false>(); run<1, true, 6, false>(); run<1, true, 8, false>();
    run<2, true, 0, false>(); run<2, true, 2, false>(); run<2, true, 4, false>(); run<2, true, 6, false>(); run<2, true, 8, false>();
    run<2, true, 1, true>(); run<2, true, 2, true>();
    run<1, false, 0, false>(); run<1, false, 4, false>(); run<1, false, 8, false>(); run<1, false, 14, false>();
    run<2, false, 4, false>(); run<2, false, 8, false>(); run<2, false, 14, false>();
    return 0;
}
