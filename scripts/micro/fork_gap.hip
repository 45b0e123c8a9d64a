// What does forking work onto a second stream cost the FIRST stream?  Chain of N (kernel A on main; kernel S on side behind A) pairs:
//   mode 0: no fork (S not launched)                          -> the bare A chain
//   mode 1: hipEventRecord(e, main) + hipStreamWaitEvent(side, e) + S on side     (what mfvi_backward does per layer)
//   mode 2: A launched with hipExtLaunchKernelGGL(..., stopEvent = e) + hipStreamWaitEvent(side, e) + S   (the event rides on A's own packet)
// build: hipcc --offload-arch=gfx950 -O3 -o fork_gap fork_gap.hip ; run: ./fork_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void spin(float* p, int iters)
{
    float v = p[threadIdx.x & 63];
    for (int i = 0; i < iters; ++i) v = __builtin_fmaf(v, 1.0000001f, 1e-9f);
    if (v == 12345.f) p[0] = v;
}
int main()
{
    float* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    hipStream_t m, s; hipStreamCreate(&m);
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi); hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo);
    const int N = 200;
    std::vector<hipEvent_t> ev(N); for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEvent_t t0, t1, join; hipEventCreate(&t0); hipEventCreate(&t1); hipEventCreateWithFlags(&join, hipEventDisableTiming);
    const int blocks = 256, itA = 20000, itS = 4000;       // A ~ 20 us on 256 CUs, S shorter
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(t0, m);
            for (int i = 0; i < N; ++i) {
                if (mode == 2) hipExtLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, m, nullptr, ev[i], 0, d, itA);
                else hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, m, d, itA);
                if (mode == 1) hipEventRecord(ev[i], m);
                if (mode >= 1) { hipStreamWaitEvent(s, ev[i], 0); hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d + 1024, itS); }
            }
            if (mode >= 1) { hipEventRecord(join, s); hipStreamWaitEvent(m, join, 0); }
            hipEventRecord(t1, m);
            hipEventSynchronize(t1);
            float ms; hipEventElapsedTime(&ms, t0, t1);
            printf("mode %d rep %d: %.3f ms for %d pairs = %.2f us per A (+fork)\n", mode, rep, ms, N, ms * 1e3f / N);
        }
    return 0;
}
