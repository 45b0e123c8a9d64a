// Microbenchmark: how fast can ONE CU issue v_mfma_f32_16x16x4_f32 when the B operand of every MFMA comes from LDS, with the
// reads software-pipelined one group ahead (sched_barrier keeps them there), and what do co-resident "producer" waves doing
// VALU + ds_write cost.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_feed mfma_feed.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: operands in registers.  MODE 1: NB b-reads + 1 a-read per group of NB MFMAs, prefetched one group ahead.
// MODE 2: like 1 but only every 3rd MFMA has a fresh operand (NB/3 reads per group).
// PROD: waves >= CW act as producers: PV VALU ops + one ds_write_b128 per iteration, no MFMA.
template <int MODE, int NB, int PV>
__global__ __launch_bounds__(1024) void kfeed(float* out, int iters, int cw)
{
    extern __shared__ __align__(16) float s[];
    const int nlds = 8192;
    for (int i = threadIdx.x; i < nlds; i += blockDim.x) s[i] = i * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv >= cw) {                                   // producer stand-in
        float4 v = make_float4(lane, 1.f, 2.f, 3.f);
        float* dst = s + nlds + (wv - cw) * 256 + lane * 4;
        __builtin_amdgcn_s_setprio(2);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < PV; ++q) { v.x = fmaf(v.x, 1.0001f, v.y); v.y = v.y > 0.f ? v.y : v.y * 0.2f; v.z = fmaf(v.z, v.x, 0.5f); v.w = fmaf(v.w, 0.999f, v.z); }
            *reinterpret_cast<float4*>(dst) = v;
            __builtin_amdgcn_sched_barrier(0);
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = v.x + v.y + v.z + v.w;
        return;
    }
    f32x4 acc[NB];
    for (int i = 0; i < NB; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    const float* base = s + wv * 64 + lane;
    float a[2], b[2][NB];
    constexpr int NR = MODE == 2 ? (NB + 2) / 3 : NB;
    auto load = [&](int it, float& aa, float (&bb)[NB]) {
        const float* p = base + ((it * 37) & 1023);
        aa = p[0];
#pragma unroll
        for (int f = 0; f < NR; ++f) bb[f] = p[(f + 1) * 400];
    };
    auto mm = [&](float aa, const float (&bb)[NB]) {
#pragma unroll
        for (int f = 0; f < NB; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, bb[MODE == 2 ? f / 3 : f], acc[f], 0, 0, 0);
    };
    if (MODE == 0) {
        float aa = lane, bb[NB]; for (int f = 0; f < NB; ++f) bb[f] = lane + f;
        for (int it = 0; it < iters; ++it) { mm(aa, bb); __builtin_amdgcn_sched_barrier(0); }
    } else {
        load(0, a[0], b[0]);
        for (int it = 0; it < iters; it += 2) {
            load(it + 1, a[1], b[1]);
            __builtin_amdgcn_sched_barrier(0);
            mm(a[0], b[0]);
            __builtin_amdgcn_sched_barrier(0);
            load(it + 2, a[0], b[0]);
            __builtin_amdgcn_sched_barrier(0);
            mm(a[1], b[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0; for (int i = 0; i < NB; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename F> float timeit(F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
template <int MODE, int NB, int PV> void run(const char* name, int cw, int pw, float* out)
{
    const int iters = 4000, blocks = 256;
    const size_t lds = (8192 + 16 * 256) * sizeof(float);
    float ms = timeit([&] { hipLaunchKernelGGL((kfeed<MODE, NB, PV>), dim3(blocks), dim3((cw + pw) * 64), lds, 0, out, iters, cw); });
    const double fl = 2.0 * 16 * 16 * 4 * NB * (double)iters * cw * blocks;
    printf("%-34s consumers %2d producers %2d (PV %2d): %7.3f ms  %6.1f TFLOP/s  %4.1f%% of 157.3\n", name, cw, pw, PV, ms, fl / ms / 1e9, fl / ms / 1e9 / 1.573);
}
int main()
{
    float* out; hipMalloc(&out, 1 << 24);
    for (int cw : {4, 8, 12}) {
        run<0, 9, 0>("regs", cw, 0, out);
        run<1, 9, 0>("LDS 10 reads / 9 MFMA, pipelined", cw, 0, out);
        run<2, 9, 0>("LDS 4 reads / 9 MFMA, pipelined", cw, 0, out);
        run<1, 27, 0>("LDS 28 reads / 27 MFMA, pipelined", cw, 0, out);
    }
    for (int cw : {4, 8}) {
        run<1, 9, 8>("LDS 10/9 + producers", cw, 4, out);
        run<1, 9, 32>("LDS 10/9 + producers", cw, 4, out);
        run<1, 9, 8>("LDS 10/9 + producers", cw, 8, out);
        run<0, 9, 8>("regs + producers", cw, 4, out);
        run<0, 9, 32>("regs + producers", cw, 4, out);
    }
    return 0;
}
