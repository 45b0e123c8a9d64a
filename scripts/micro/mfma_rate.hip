// Microbenchmark: issue rate of fp32 MFMA shapes with operands in registers / from LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, int iters, float a0, float b0)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k32(float* out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// 16x16x4 with operands re-read from LDS every step (1 A + 8 B per 8 MFMAs), like the conv consumer loop
__global__ void k16_lds(float* out, int iters)
{
    __shared__ float s[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = i * 1e-3f;
    __syncthreads();
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    const int lane = threadIdx.x & 63, base = (threadIdx.x >> 6) * 64;
    float a[2], b[2][8];
    a[0] = s[lane]; for (int f = 0; f < 8; ++f) b[0][f] = s[base + f * 624 % 4096 + lane];
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int o = ((it + q + 1) * 37) & 1023;
            a[(q + 1) & 1] = s[o + lane];
#pragma unroll
            for (int f = 0; f < 8; ++f) b[(q + 1) & 1][f] = s[o + f * 600 + lane];
#pragma unroll
            for (int f = 0; f < 8; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q & 1], b[q & 1][f], acc[f], 0, 0, 0);
        }
    }
    float r = 0; for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename F> float timeit(F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main()
{
    float* out; hipMalloc(&out, 1 << 24);
    const int iters = 20000;
    for (int wpb : {4, 8, 16}) {       // waves per block; 256 blocks of 1 per CU -> waves per SIMD = wpb/4
        const int threads = wpb * 64, blocks = 256;
        float ms = timeit([&] { hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 2.f); });
        double fl = 2.0 * 16 * 16 * 4 * 8 * (double)iters * wpb * blocks;
        printf("16x16x4 regs  %2d waves/CU: %7.3f ms  %6.1f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", wpb, ms, fl / ms / 1e9, ms * 1e-3 * 2.4e9 / (8.0 * iters * wpb / 4));
        ms = timeit([&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 2.f); });
        fl = 2.0 * 32 * 32 * 2 * 4 * (double)iters * wpb * blocks;
        printf("32x32x2 regs  %2d waves/CU: %7.3f ms  %6.1f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", wpb, ms, fl / ms / 1e9, ms * 1e-3 * 2.4e9 / (4.0 * iters * wpb / 4));
        ms = timeit([&] { hipLaunchKernelGGL(k16_lds, dim3(blocks), dim3(threads), 0, 0, out, iters); });
        fl = 2.0 * 16 * 16 * 4 * 8 * (double)iters * wpb * blocks;
        printf("16x16x4 LDS   %2d waves/CU: %7.3f ms  %6.1f TFLOP/s\n", wpb, ms, fl / ms / 1e9);
    }
    return 0;
}
