// Microbenchmark: fp32 GEMM on the bf16 matrix cores by a three-way operand split ("bf16x6").
//   a = a_h + a_m + a_l (three bf16 pieces, exact), a*b ~= a_h b_h + a_h b_m + a_m b_h + a_m b_m + a_h b_l + a_l b_h
//   (dropped: m l + l m + l l <= 2^-23 |a b| with round-to-nearest pieces — what the kernels use — 2^-21 with truncated pieces)
// on v_mfma_f32_16x16x32_bf16 (16 cycles per instruction per SIMD, 16x the fp32 MFMA rate): six instructions replace eight
// v_mfma_f32_16x16x4_f32 (256 cycles) -> 2.67x the fp32 matrix peak at fp32 accuracy.
// Part 1: operand layout + accuracy against fp64 (fp32 MFMA, x6 with RNE pieces, x6 with truncated pieces, x3).
// Part 2: issue rate of the x6 stream with 0 / 1 / 2 VALU fillers per MFMA, one and two waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/bf16x6.hip -o /tmp/bf16x6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// split two floats into packed bf16 pieces (low half = first element)
__device__ __forceinline__ void split_rne(float a0, float a1, unsigned& h, unsigned& m, unsigned& l)
{
    auto rne = [](float f) { const unsigned u = __float_as_uint(f); return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u; };
    const unsigned h0 = rne(a0), h1 = rne(a1);
    const float r0 = a0 - __uint_as_float(h0), r1 = a1 - __uint_as_float(h1);
    const unsigned m0 = rne(r0), m1 = rne(r1);
    const float s0 = r0 - __uint_as_float(m0), s1 = r1 - __uint_as_float(m1);
    const unsigned l0 = rne(s0), l1 = rne(s1);
    h = (h0 >> 16) | h1; m = (m0 >> 16) | m1; l = (l0 >> 16) | l1;
}
__device__ __forceinline__ void split_trunc(float a0, float a1, unsigned& h, unsigned& m, unsigned& l)
{
    const unsigned h0 = __float_as_uint(a0) & 0xffff0000u, h1 = __float_as_uint(a1) & 0xffff0000u;
    const float r0 = a0 - __uint_as_float(h0), r1 = a1 - __uint_as_float(h1);
    const unsigned m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
    const float s0 = r0 - __uint_as_float(m0), s1 = r1 - __uint_as_float(m1);
    h = __builtin_amdgcn_perm(h1, h0, 0x07060302u); m = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

// D[16][16] = A[16][K] * B[K][16], one wave.  mode 0: fp32 MFMA; 1: x6 RNE; 2: x6 truncated; 3: x3 (RNE, two pieces)
__global__ void gemm_check(const float* A, const float* B, float* D, int K, int mode)
{
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (mode == 0) {
        for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[n * K + k0 + g], B[(k0 + g) * 16 + n], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 32) {
            u32x4 ah, am, al, bh, bm, bl;
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + 8 * g + 2 * j;      // lane (g, n): k = 8 g .. 8 g + 7, element pair j
                unsigned h, m, l;
                if (mode == 2) split_trunc(A[n * K + k], A[n * K + k + 1], h, m, l); else split_rne(A[n * K + k], A[n * K + k + 1], h, m, l);
                ah[j] = h; am[j] = m; al[j] = l;
                if (mode == 2) split_trunc(B[k * 16 + n], B[(k + 1) * 16 + n], h, m, l); else split_rne(B[k * 16 + n], B[(k + 1) * 16 + n], h, m, l);
                bh[j] = h; bm[j] = m; bl[j] = l;
            }
            if (mode != 3) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(al), as_bf(bh), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(ah), as_bf(bl), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(am), as_bf(bm), acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(am), as_bf(bh), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(ah), as_bf(bm), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(ah), as_bf(bh), acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + n] = acc[r];      // row m = 4 (lane >> 4) + r, column n = lane & 15
}

// rate: NACC accumulators, per MFMA `FILL` alignbit fillers on independent registers
template <int NACC, int FILL>
__global__ void rate(float* out, int iters, unsigned seed)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    u32x4 a = {seed + threadIdx.x, seed * 3 + threadIdx.x, seed * 5, seed * 7}, b = {seed * 11 + threadIdx.x, seed * 13, seed * 17, seed * 19};
    a &= 0x3f803f80u; b &= 0x3f803f80u;
    unsigned f0 = threadIdx.x, f1 = seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a), as_bf(b), acc[i], 0, 0, 0);
            if (FILL >= 1) f0 = __builtin_amdgcn_alignbit(f0, f1, 16);
            if (FILL >= 2) f1 = __builtin_amdgcn_alignbit(f1, f0, 16);
            if (FILL >= 3) f0 = __builtin_amdgcn_alignbit(f0, f1, 15);
            if (FILL >= 4) f1 = __builtin_amdgcn_alignbit(f1, f0, 17);
        }
    }
    float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(f0 ^ f1);
}

// legacy K = 16 shape (4 bf16 per lane): candidate for the 4-channel remainder groups
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void rate16(float* out, int iters, unsigned seed)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    s16x4 a = {(short)0x3f80, (short)(0x3f80 + (threadIdx.x & 3)), (short)0x3f80, (short)0x3f81}, b = {(short)0x3f80, (short)0x3f82, (short)(0x3f80 + (seed & 1)), (short)0x3f80};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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
    for (int K : {32, 1024, 4096}) {
        std::vector<float> A(16 * K), B(K * 16), D(256);
        srand(1234 + K);
        auto rnd = [] { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0) * expf((float)(rand() % 9 - 4)); };      // magnitudes over e^-4 .. e^4
        for (auto& v : A) v = rnd();
        for (auto& v : B) v = rnd();
        float *dA, *dB, *dD; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        std::vector<double> ref(256), mag(256);
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { double s = 0, a = 0; for (int k = 0; k < K; ++k) { s += (double)A[m * K + k] * B[k * 16 + n]; a += fabs((double)A[m * K + k] * B[k * 16 + n]); } ref[m * 16 + n] = s; mag[m * 16 + n] = a; }
        const char* names[] = {"fp32 MFMA 16x16x4", "bf16x6 RNE pieces", "bf16x6 truncated", "bf16x3 (two RNE pieces)"};
        for (int mode = 0; mode < 4; ++mode) {
            hipLaunchKernelGGL(gemm_check, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, mode);
            hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
            double worst = 0, mean = 0;
            for (int i = 0; i < 256; ++i) { const double e = fabs(D[i] - ref[i]) / mag[i]; worst = fmax(worst, e); mean += e / 256; }
            printf("K=%5d  %-26s max |err| / sum|a b| = %.3e   mean %.3e\n", K, names[mode], worst, mean);
        }
        hipFree(dA); hipFree(dB); hipFree(dD);
    }
    float* out; hipMalloc(&out, 1 << 24);
    const int iters = 4000;
    for (int wpb : {4, 8}) {
        const int threads = wpb * 64, blocks = 256;
        auto report = [&](const char* name, float ms, int nacc) {
            const double fl = 2.0 * 16 * 16 * 32 * nacc * (double)iters * wpb * blocks;
            printf("%-28s %d waves/CU: %7.3f ms  %7.1f TFLOP/s bf16 = %6.1f fp32-equivalent (x6)  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", name, wpb, ms, fl / ms / 1e9, fl / ms / 1e9 / 6,
                   ms * 1e-3 * 2.4e9 / ((double)nacc * iters * wpb / 4));
        };
        report("16x16x32 bf16, 21 acc", timeit([&] { hipLaunchKernelGGL((rate<21, 0>), dim3(blocks), dim3(threads), 0, 0, out, iters, 7u); }), 21);
        report("  + 1 alignbit per MFMA", timeit([&] { hipLaunchKernelGGL((rate<21, 1>), dim3(blocks), dim3(threads), 0, 0, out, iters, 7u); }), 21);
        report("  + 2 alignbit per MFMA", timeit([&] { hipLaunchKernelGGL((rate<21, 2>), dim3(blocks), dim3(threads), 0, 0, out, iters, 7u); }), 21);
        { const float ms = timeit([&] { hipLaunchKernelGGL((rate16<21>), dim3(blocks), dim3(threads), 0, 0, out, iters, 7u); });
          printf("16x16x16 bf16_1k, 21 acc     %d waves/CU: %7.3f ms  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", wpb, ms, ms * 1e-3 * 2.4e9 / (21.0 * iters * wpb / 4)); }
        report("  + 4 alignbit per MFMA", timeit([&] { hipLaunchKernelGGL((rate<21, 4>), dim3(blocks), dim3(threads), 0, 0, out, iters, 7u); }), 21);
    }
    return 0;
}
