// Micro-benchmark: does v_mfma_f32_32x32x16_f16 / 16x16x32_f16 slow down when a source operand (A or B) sits in the AGPR half
// of the unified register file?  One wave per SIMD, two independent accumulators, register classes forced by inline asm.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_agpr.hip -o /tmp/mfma_agpr && /tmp/mfma_agpr
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define LOOP_BODY(ASM)                                                                                              \
    for (int i = 0; i < iters; ++i) {                                                                               \
        ASM(c0) ASM(c1) ASM(c0) ASM(c1) ASM(c0) ASM(c1) ASM(c0) ASM(c1)                                             \
    }

template <int MODE>
__global__ void __launch_bounds__(256) k32(float* out, int iters, unsigned seed) {
    u32x4 a = {seed + threadIdx.x, seed * 3u, 0x3c003c00u, 0x38003800u}, b = {0x3c003c00u, seed, 0x34003400u, seed + 7u};
    f32x16 c0 = {}, c1 = {};
    long long t0 = clock64();
    if (MODE == 0) {
#define M0(C) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(C) : "v"(a), "v"(b));
        LOOP_BODY(M0)
    } else if (MODE == 1) {
#define M1(C) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(C) : "v"(a), "a"(b));
        LOOP_BODY(M1)
    } else if (MODE == 2) {
#define M2(C) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(C) : "a"(a), "a"(b));
        LOOP_BODY(M2)
    } else if (MODE == 3) {
#define M3(C) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(C) : "v"(a), "a"(b));
        LOOP_BODY(M3)
    } else if (MODE == 4) {
#define M4(C) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(C) : "v"(a), "v"(b));
        LOOP_BODY(M4)
    } else if (MODE == 5) {
#define M5(C) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(C) : "a"(a), "v"(b));
        LOOP_BODY(M5)
    }
    long long t1 = clock64();
    f32x16 s = c0 + c1;
    float r = 0;
    for (int i = 0; i < 16; ++i) r += s[i];
    out[blockIdx.x * 256 + threadIdx.x] = r + (float)(t1 - t0) * 1e-30f;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + 256 * 1024)[0] = t1 - t0;
}

template <int MODE>
__global__ void __launch_bounds__(256) k16(float* out, int iters, unsigned seed) {
    u32x4 a = {seed + threadIdx.x, seed * 3u, 0x3c003c00u, 0x38003800u}, b = {0x3c003c00u, seed, 0x34003400u, seed + 7u};
    f32x4 c0 = {}, c1 = {};
    long long t0 = clock64();
    if (MODE == 0) {
#define N0(C) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(C) : "v"(a), "v"(b));
        LOOP_BODY(N0)
    } else if (MODE == 1) {
#define N1(C) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(C) : "v"(a), "a"(b));
        LOOP_BODY(N1)
    } else if (MODE == 2) {
#define N2(C) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(C) : "a"(a), "a"(b));
        LOOP_BODY(N2)
    } else if (MODE == 3) {
#define N3(C) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(C) : "v"(a), "a"(b));
        LOOP_BODY(N3)
    } else if (MODE == 4) {
#define N4(C) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(C) : "v"(a), "v"(b));
        LOOP_BODY(N4)
    } else if (MODE == 5) {
#define N5(C) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(C) : "a"(a), "v"(b));
        LOOP_BODY(N5)
    }
    long long t1 = clock64();
    f32x4 s = c0 + c1;
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + (float)(t1 - t0) * 1e-30f;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + 256 * 1024)[0] = t1 - t0;
}

template <class K>
static void run(const char* name, K kern, float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<256, 256>>>(d, iters, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<256, 256>>>(d, iters, 1u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long cyc = 0;
    hipMemcpy(&cyc, d + 256 * 1024, sizeof(cyc), hipMemcpyDeviceToHost);
    printf("%-44s %8.3f ms   %7.1f clock64 ticks / MFMA\n", name, ms, (double)cyc / (8.0 * iters));
}

int main() {
    float* d;
    hipMalloc(&d, (256 * 1024 + 16) * sizeof(float));
    const int iters = 20000;
    run("32x32x16  A=vgpr B=vgpr C=agpr", k32<0>, d, iters);
    run("32x32x16  A=vgpr B=agpr C=agpr", k32<1>, d, iters);
    run("32x32x16  A=agpr B=agpr C=agpr", k32<2>, d, iters);
    run("32x32x16  A=agpr B=vgpr C=agpr", k32<5>, d, iters);
    run("32x32x16  A=vgpr B=agpr C=vgpr", k32<3>, d, iters);
    run("32x32x16  A=vgpr B=vgpr C=vgpr", k32<4>, d, iters);
    run("16x16x32  A=vgpr B=vgpr C=agpr", k16<0>, d, iters);
    run("16x16x32  A=vgpr B=agpr C=agpr", k16<1>, d, iters);
    run("16x16x32  A=agpr B=agpr C=agpr", k16<2>, d, iters);
    run("16x16x32  A=agpr B=vgpr C=agpr", k16<5>, d, iters);
    run("16x16x32  A=vgpr B=agpr C=vgpr", k16<3>, d, iters);
    run("16x16x32  A=vgpr B=vgpr C=vgpr", k16<4>, d, iters);
    return 0;
}
