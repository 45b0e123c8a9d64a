// Shared device/host definitions of libmfvi_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

// A kernel launch that carries a pending "stop" event on its own dispatch packet (hipExtLaunchKernelGGL): mfvi_backward arms the event in
// front of the last launch on the caller's stream before it forks work onto the side stream, so the fork costs that stream no marker
// packet of its own (scripts/micro/fork_gap.hip: hipEventRecord + hipStreamWaitEvent add 5.1 us to the recording stream per fork, the
// event on the kernel's packet 1.8 us).  Not armed (the normal case, and every launch outside mfvi_backward): a plain launch.
extern thread_local hipEvent_t mfvi_tl_stop_event;
// kernel family of the conv launch in progress, for mfvi_plan_last_kernel: the plan sets 1 (fp32 MFMA kernels) in front of an MFMA-path
// launcher, the row-phase launchers (conv_rp.hip) overwrite it with 2, the bf16x6 launchers (conv_x6.hip, conv_bww_x6.hip) with 3
extern thread_local int mfvi_tl_family;
template <typename F, typename... Args>
inline void mfvi_launch(F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args)
{
    if (mfvi_tl_stop_event) {
        hipEvent_t e = mfvi_tl_stop_event; mfvi_tl_stop_event = nullptr;
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, st, nullptr, e, 0, args...);
    } else hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
}

#define MFVI_MAX_C 256           // max channels of any activation tensor handled by the kernels

// ------------------------------------------------------------------------------------------------
// RNG spec v1.  Written with explicit fmaf / separate mul so that the bits match the
// CPU oracle's independent restatement (built with -ffp-contract=off).
// ------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)

__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                       uint32_t k0, uint32_t k1, uint32_t r[4])
{
#pragma unroll
    for (int i = 0; i < 10; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
#else
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
#endif
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

__device__ __forceinline__ float spec_logf(float u)
{
    const uint32_t b = __float_as_uint(u);
    int e = (int)((b >> 23) & 0xffu) - 127;
    float m = __uint_as_float((b & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f;
    const float z = f * f;
    float y = 7.0376836292E-2f;
    y = __builtin_fmaf(y, f, -1.1514610310E-1f);
    y = __builtin_fmaf(y, f, 1.1676998740E-1f);
    y = __builtin_fmaf(y, f, -1.2420140846E-1f);
    y = __builtin_fmaf(y, f, 1.4249322787E-1f);
    y = __builtin_fmaf(y, f, -1.6668057665E-1f);
    y = __builtin_fmaf(y, f, 2.0000714765E-1f);
    y = __builtin_fmaf(y, f, -2.4999993993E-1f);
    y = __builtin_fmaf(y, f, 3.3333331174E-1f);
    y = y * f;
    y = y * z;
    const float fe = (float)e;
    y = __builtin_fmaf(-2.12194440e-4f, fe, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = f + y;
    r = __builtin_fmaf(0.693359375f, fe, r);
    return r;
}

__device__ __forceinline__ void spec_boxmuller(uint32_t a, uint32_t b, float& z0, float& z1)
{
    const float u1 = ((float)(a >> 9) + 0.5f) * 1.1920928955078125e-07f;
    const float rad = __builtin_sqrtf(-2.0f * spec_logf(u1));   // correctly rounded (HIP default); __fsqrt_rn is the native approximation
    const uint32_t t = b >> 8;
    const uint32_t n = (t + 0x200000u) >> 22;
    const int32_t d = (int32_t)t - (int32_t)(n << 22);
    const float p = ((float)d * 1.1920928955078125e-07f) * 3.14159265358979323846f;
    const float z = p * p;
    float s = -1.9515295891E-4f;
    s = __builtin_fmaf(s, z, 8.3321608736E-3f);
    s = __builtin_fmaf(s, z, -1.6666654611E-1f);
    s = s * z;
    s = __builtin_fmaf(s, p, p);
    float c = 2.443315711809948E-005f;
    c = __builtin_fmaf(c, z, -1.388731625493765E-003f);
    c = __builtin_fmaf(c, z, 4.166664568298827E-002f);
    c = c * z;
    c = c * z;
    c = __builtin_fmaf(-0.5f, z, c);
    c = c + 1.0f;
    const uint32_t q = n & 3u;
    const float cs = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
    const float sn = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    z0 = rad * cs;
    z1 = rad * sn;
}

enum { DOMAIN_EPS = 0, DOMAIN_INPUT = 1, DOMAIN_INIT = 2, DOMAIN_UNIFORM = 3, DOMAIN_SGLD = 4, DOMAIN_DROPOUT = 5, DOMAIN_ROUND = 6, DOMAIN_LRT = 7 };

struct RngKey {              // everything but the block index
    uint32_t k0, k1;         // seed lo/hi
    uint32_t stream;         // (domain << 24) | stream id
    uint32_t sample, step;
    // Device-resident step counter (mfvi_plan_set_step_source, mfvi_perturb_input_dev): when set, `step` is an OFFSET and the counter word
    // the kernel uses is step + *step_dev, read on the device at run time — so that a captured HIP graph of one iteration replays with
    // a fresh counter every time (the reference's loop index i, bayesian_optimization.py:1360).  nullptr: `step` is the counter itself.
    const int32_t* step_dev;
};
// first statement of every kernel that takes a key: resolve the device-resident part of the step counter (one scalar load)
__device__ __forceinline__ RngKey key_now(RngKey k)
{
    if (k.step_dev) { k.step += (uint32_t)*k.step_dev; k.step_dev = nullptr; }
    return k;
}

// 4 standard normals of Philox block `blk` (elements 4*blk .. 4*blk+3 of the stream)
__device__ __forceinline__ void spec_normal4(const RngKey& key, uint32_t blk, float z[4])
{
    uint32_t r[4];
    philox4x32_10(blk, key.stream, key.sample, key.step, key.k0, key.k1, r);
    spec_boxmuller(r[0], r[1], z[0], z[1]);
    spec_boxmuller(r[2], r[3], z[2], z[3]);
}

#pragma clang fp contract(fast)

// bfloat16 parameter storage (mu / rho of BASELINE configs[4]): the upper 16 bits of the float32 pattern
typedef uint16_t bf16_t;
__host__ __device__ __forceinline__ float bf16_to_f32(bf16_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float((uint32_t)b << 16);
#else
    union { uint32_t u; float f; } c; c.u = (uint32_t)b << 16; return c.f;
#endif
}
__device__ __forceinline__ float4 bf16x4_to_f32(const bf16_t* p)      // 8-byte aligned quad
{
    const uint2 q = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u));
}
// round to nearest even (NaN-free parameters)
__device__ __forceinline__ bf16_t f32_to_bf16_rne(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
// stochastic rounding: round the magnitude up with probability (discarded bits) / 2^16, `r16` uniform in [0, 2^16); unbiased, so
// Adam updates far below one bf16 ulp (lr 1e-3 against ulp(rho = -3) = 1.6e-2) still move the parameter in expectation
__device__ __forceinline__ bf16_t f32_to_bf16_sr(float f, uint32_t r16)
{
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7f800000u) == 0x7f800000u) return (bf16_t)(u >> 16);       // inf / NaN pass through
    return (bf16_t)((u + (r16 & 0xffffu)) >> 16);
}

// bf16x6 kernels (conv_bww_x6.hip, conv_x6.hip): two fp32 values -> three packed bf16 pairs (low half = first value), a = h + m + l
// exactly (round-to-nearest-even pieces of 8 significand bits: what is left after two steps has at most 8 bits), |m| <= 2^-8 |a|,
// |l| <= 2^-16 |a|, so the three products the kernels drop (m l, l m, l l) stay below 2^-23 |a b|.  v_cvt_pk_bf16_f32 rounds and packs a
// pair in one instruction: 11 VALU operations per pair, the same as a truncating split (which would leave 2^-21).
typedef __bf16 mfvi_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mfvi_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_bf16x3(float a0, float a1, unsigned& h, unsigned& m, unsigned& l)
{
    const mfvi_f32x2 a = {a0, a1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(a, mfvi_bf16x2));
    const mfvi_f32x2 r = {a0 - __uint_as_float(h << 16), a1 - __uint_as_float(h & 0xffff0000u)};
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, mfvi_bf16x2));
    const mfvi_f32x2 t = {r.x - __uint_as_float(m << 16), r.y - __uint_as_float(m & 0xffff0000u)};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(t, mfvi_bf16x2));
}

// torch.nn.functional.softplus(beta=1, threshold=20)
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
// softplus for the conv kernels' hot weight-sampling loops: hardware exp/log (v_exp_f32 / v_log_f32), with the
// log1p series where 1 + e would lose e's low bits.  Relative error < 3e-6 (tests/test_gpu_parity.py).
__device__ __forceinline__ float softplus_fast(float x)
{
    if (x > 20.f) return x;
    const float e = __expf(x);
    if (e < 0.06f) return e * (1.f - e * (0.5f - e * (0.33333333f - e * (0.25f - e * 0.2f))));
    return __logf(1.f + e);
}

// ------------------------------------------------------------------------------------------------
// Tensor views
// ------------------------------------------------------------------------------------------------
// Read side of an activation tensor: raw data + the deferred BatchNorm(train, N=1)/LeakyReLU.
struct TView {
    const float* data;       // sample 0
    long long sstride;       // floats between samples (0: shared by all samples)
    int C, H, W;
    const double* stats;     // [n_samples][C][2] = sum, sum of squares; nullptr: no BN
    const float* gamma;      // gamma[C], beta[C] follows at gamma + C
    float eps, slope;
    int act;
    const float* drop;       // [n_samples][C] Dropout2d factors (0 or 1/(1-p)) applied between the raw data and the BN; nullptr: none
};

// Gradient wrt a raw tensor y, formed on load from ga = dL/d(BN output) (or dL/dy when no BN):
//   dy = gamma*rstd * (ga - mean(ga) - xhat * mean(ga*xhat))
struct GView {
    const float* ga; long long gstride;
    const float* y;  long long ystride;
    int C, H, W;
    const double* stats;     // forward sums of y (nullptr: no BN, dy = ga)
    const double* bsums;     // [n_samples][C][2] = sum ga, sum ga*xhat
    const float* gamma;
    float eps;
    const float* drop;       // as TView::drop
};

// per-channel constants of a view, computed once per block into LDS/registers
struct ChanFwd { float mean, scale, beta, rstd; };     // v = (y - mean) * scale + beta
struct ChanBwd { float mean, rstd, c1, c2, c3; };      // dy = c1 * (ga - c2 - xhat*c3), xhat = (y-mean)*rstd

__device__ __forceinline__ ChanFwd chan_fwd(const TView& v, int k, int c)
{
    ChanFwd r;
    if (v.stats == nullptr) { r.mean = 0.f; r.scale = 1.f; r.beta = 0.f; r.rstd = 1.f; return r; }
    const double n = (double)v.H * (double)v.W;
    const double* s = v.stats + ((long long)k * v.C + c) * 2;
    const double m = s[0] / n;
    double var = s[1] / n - m * m; if (var < 0) var = 0;
    // Dropout2d factor d between y and the BN: BN(d*y) = (y - m) * (d*rstd_d) * gamma + beta with rstd_d = 1/sqrt(d^2 var + eps)
    const double d = v.drop ? (double)v.drop[(long long)k * v.C + c] : 1.0;
    const double rstd = d / sqrt(d * d * var + (double)v.eps);
    r.mean = (float)m; r.scale = (float)(rstd * (double)v.gamma[c]); r.beta = v.gamma[v.C + c]; r.rstd = (float)rstd;
    return r;
}
// act: bit 0 = LeakyReLU(slope) after the BN; bit 1 (MFVI_ACT_SQUARE) = square the result — the x**2 operand of the local-reparameterisation
// layers' variance convolution (BayTorch/modules/reparam_layers.py:69), set by the plan on the view it hands to that convolution only
#define MFVI_ACT_SQUARE 2
__device__ __forceinline__ float apply_fwd(const ChanFwd& c, float y, int act, float slope)
{
    float v = __builtin_fmaf(y - c.mean, c.scale, c.beta);
    const float se = (act & 1) ? slope : 1.f;      // uniform: LeakyReLU(slope), or the identity written as slope 1
    v = v > 0.f ? v : v * se;
    if (act & MFVI_ACT_SQUARE) v = v * v;
    return v;
}
// Four elements at once on the packed fp32 instructions (v_pk_add / v_pk_fma / v_pk_mul: two lanes' worth per instruction), bit-identical to
// apply_fwd for 0 <= slope <= 1 (LeakyReLU as max(v, slope * v); plan creation rejects other slopes).  The staging waves of the MFMA
// kernels share their SIMD's issue port with the matrix instructions, so every VALU instruction saved there is matrix time gained:
// 10 instead of 20 per float4.
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void apply_fwd4(const ChanFwd& c, float (&e)[4], int act, float slope)
{
    const float se = (act & 1) ? slope : 1.f;
    const f2v m = {c.mean, c.mean}, sc = {c.scale, c.scale}, be = {c.beta, c.beta}, s2 = {se, se};
    f2v a = {e[0], e[1]}, b = {e[2], e[3]};
    a = __builtin_elementwise_fma(a - m, sc, be); b = __builtin_elementwise_fma(b - m, sc, be);
    const f2v a2 = a * s2, b2 = b * s2;
    e[0] = __builtin_fmaxf(a.x, a2.x); e[1] = __builtin_fmaxf(a.y, a2.y); e[2] = __builtin_fmaxf(b.x, b2.x); e[3] = __builtin_fmaxf(b.y, b2.y);
    if (act & MFVI_ACT_SQUARE) { e[0] *= e[0]; e[1] *= e[1]; e[2] *= e[2]; e[3] *= e[3]; }
}
__device__ __forceinline__ ChanBwd chan_bwd(const GView& g, int k, int c)
{
    ChanBwd r;
    if (g.stats == nullptr) { r.mean = 0.f; r.rstd = 0.f; r.c1 = 1.f; r.c2 = 0.f; r.c3 = 0.f; return r; }
    const double n = (double)g.H * (double)g.W;
    const double* s = g.stats + ((long long)k * g.C + c) * 2;
    const double* b = g.bsums + ((long long)k * g.C + c) * 2;
    const double m = s[0] / n;
    double var = s[1] / n - m * m; if (var < 0) var = 0;
    const double d = g.drop ? (double)g.drop[(long long)k * g.C + c] : 1.0;
    const double rstd = d / sqrt(d * d * var + (double)g.eps);          // xhat = (y - m) * rstd, dy = d * gamma * rstd_d * (...)
    r.mean = (float)m; r.rstd = (float)rstd; r.c1 = (float)(rstd * (double)g.gamma[c]);
    r.c2 = (float)(b[0] / n); r.c3 = (float)(b[1] / n);
    return r;
}
__device__ __forceinline__ float apply_bwd(const ChanBwd& c, float ga, float y)
{
    const float xhat = (y - c.mean) * c.rstd;
    return c.c1 * __builtin_fmaf(-xhat, c.c3, ga - c.c2);
}
// packed form, same operations per element (5 packed instructions per pair instead of 5 per element)
__device__ __forceinline__ void apply_bwd4(const ChanBwd& c, float (&e)[4], const float (&y)[4])
{
    const f2v m = {c.mean, c.mean}, rs = {c.rstd, c.rstd}, c1 = {c.c1, c.c1}, c2 = {c.c2, c.c2}, nc3 = {-c.c3, -c.c3};
    f2v ga0 = {e[0], e[1]}, ga1 = {e[2], e[3]}, y0 = {y[0], y[1]}, y1 = {y[2], y[3]};
    const f2v xh0 = (y0 - m) * rs, xh1 = (y1 - m) * rs;
    const f2v r0 = c1 * __builtin_elementwise_fma(xh0, nc3, ga0 - c2), r1 = c1 * __builtin_elementwise_fma(xh1, nc3, ga1 - c2);
    e[0] = r0.x; e[1] = r0.y; e[2] = r1.x; e[3] = r1.y;
}

__device__ __forceinline__ int reflect_idx(int i, int n)
{
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    return i;
}

// Barrier that publishes LDS writes but does NOT drain outstanding global loads: __syncthreads() makes hipcc emit
// s_waitcnt vmcnt(0) first, which would serialise every register-prefetch behind the barrier (cdna_hip_programming.md §5).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// Block reductions (256-thread blocks, wave = 64)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the block; result valid in thread 0.  `red` is LDS scratch of >= (blockDim.x/64) doubles.
__device__ __forceinline__ double block_sum_d(double v, double* red)
{
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

// ------------------------------------------------------------------------------------------------
// Host-side launch plumbing shared by the .hip files
// ------------------------------------------------------------------------------------------------
// Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.  Decode a 1-D launch of nx*ny*nz
// blocks so that (a) the ny blocks that read the same input tile (output-channel tiles / channel-pair tiles of one spatial
// unit) run back to back on ONE XCD and (b) each XCD owns a contiguous band of (sample, spatial) units, so halo rows are
// re-read from its L2, not from HBM.  Falls back to the plain order when nx*nz is not a multiple of 8.
__device__ __forceinline__ void xcd_decode(int L, int nx, int ny, int nz, int& bx, int& by, int& bz)
{
    const int units = nx * nz;
    if ((units & 7) == 0) {
        const int xcd = L & 7, slot = L >> 3;
        by = slot % ny;
        const int u = xcd * (units >> 3) + slot / ny;
        bx = u % nx; bz = u / nx;
    } else { bx = L % nx; by = (L / nx) % ny; bz = L / (nx * ny); }
}

struct ConvGeom {
    int Cin, Cout, H, W, Ho, Wo, ks, stride;
    long long w_off, b_off;    // into mu / rho
    int layer_id;
    int tune[3];               // MFMA tiling chosen by mfvi_plan_autotune for forward / backward-data / backward-weight (0 = heuristic)
};

struct OutDesc {               // raw output tensor of a forward op
    float* data; long long sstride;
    double* stats;             // [n_samples][C][2] to accumulate, or nullptr
};

void set_error(const char* fmt, ...);

int launch_conv_fwd(const TView& in, const ConvGeom& g, const float* mu, const float* rho, RngKey key, int sample_weights,
                    OutDesc out, int n_samples, hipStream_t st);
int launch_conv_bwd_data(const GView& gy, const ConvGeom& g, const float* mu, const float* rho, RngKey key, int sample_weights,
                         float* dxp, long long dxp_sstride, int n_samples, hipStream_t st);
int launch_conv_bwd_weight(const TView& in, const GView& gy, const ConvGeom& g, const float* rho, RngKey key, int sample_weights,
                           float* dmu, float* drho, int n_samples, hipStream_t st);
// MFMA variants (conv_mfma.hip): return -2 when the shape is not served and the generic kernel must run.
// w: weights of sample 0, sample k at w + k*wstride (plan.hip: the buffer filled by launch_sample_weights, or mu with stride 0)
int launch_conv_fwd_mfma(const TView& in, const ConvGeom& g, const float* w, long long wstride, OutDesc out, int n_samples, hipStream_t st);
// fuse (optional, 1x1 layers whose input has no other consumer): the epilogue does the fold of that input tensor itself — multiplies by
// LeakyReLU'(view(x)), accumulates the BN-backward sums of x (bsums, nullptr when x carries no BatchNorm) and writes ga directly; no
// padded-gradient scratch, no finalize_dx launch.
struct FoldFuse { TView x; float* ga = nullptr; long long ga_sstride = 0; double* bsums = nullptr; };
int launch_conv_bwd_data_mfma(const GView& gy, const ConvGeom& g, const float* w, long long wstride, float* dxp, long long dxp_sstride,
                              int n_samples, hipStream_t st, const FoldFuse* fuse = nullptr);
// Row-phase kernels for 3x3 stride-1 layers on maps whose width is a multiple of 64 (conv_rp.hip).  tune = mf | r << 8 | rem << 12 | T << 16
// (output fragments per block, rows per wave, 4 extra channels on the 4x4x1 instruction, tiles per block); -2: shape not served, -3: tiling
// not valid for the shape.  Backward-data always runs the fold of the input tensor in its epilogue (fuse.ga required).
#define MFVI_TUNE_RP (1 << 24)
int launch_conv_fwd_rp(const TView& in, const ConvGeom& g, const float* w, long long wstride, OutDesc out, int tune, int n_samples, hipStream_t st);
int launch_conv_bwd_data_rp(const GView& gy, const ConvGeom& g, const float* w, long long wstride, int tune, int n_samples, hipStream_t st, const FoldFuse& fuse);
int rp_default_tune(const ConvGeom& g, int mode, int n_samples);      // 0: not served / disabled (MFVI_RP=0)
// Forward of the 3x3 stride-1 layers with 32 n (+ 4) input channels on maps whose width is a multiple of 64, on the bf16 matrix cores with
// three-way split operands (conv_x6.hip).  tune = mf | sr << 8 (output fragments per block, output rows per block).  Needs a scratch region
// of x6_fwd_scratch_floats() floats for the split weight pieces: the plan hands it over in mfvi_tl_x6w around the launch (nullptr: -2).
#define MFVI_TUNE_X6 (1 << 25)
// small-map forward (conv_small.hip): the whole reduction of a (sample, 16 output channels, 64 pixels) block in LDS, one stage; tune bit 26
#define MFVI_TUNE_SM (1 << 26)
// "in-kernel eps" (tune bit 27, any of the three passes): the layer runs on the generic kernels of conv_fwd.hip / conv_bwd_*.hip, which draw eps, form
// w = mu + softplus(rho) * eps and convolve in ONE launch (BayTorch/modules/module.py:82-85 + reparam_layers.py:28-37, as BASELINE's north_star
// words it) — no sampled-weight slab, eps bit-identical by the RNG spec.  A candidate of the autotuner for layers of at most MFVI_INKERNEL_MAX_W
// weights (every 1x1 skip convolution, the 16 -> 16 layers): kept where it is at least as fast as the matrix-core kernel reading the slab.
#define MFVI_TUNE_GENERIC (1 << 27)
#define MFVI_INKERNEL_MAX_W 2560
int launch_conv_fwd_small(const TView& in, const ConvGeom& g, const float* w, long long wstride, OutDesc out, int n_samples, hipStream_t st);
int launch_conv_bwd_data_small(const GView& gy, const ConvGeom& g, const float* w, long long wstride, int n_samples, hipStream_t st, const FoldFuse& fuse);
// Streaming forward of the narrow 1x1 layers (conv_1x1.hip, conv1_stream_kernel): Cin in {4, 8, 12, 16, 32, 64}, Cout <= 16 (<= 32 for Cin 16 / 32), H*W a multiple of 64;
// nothing through LDS, the pixel operand straight from global memory into the matrix instruction.  Tune bit 28; -2: shape not served
#define MFVI_TUNE_ST (1 << 28)
int launch_conv1_fwd_stream(const TView& in, const ConvGeom& g, const float* w, long long wstride, OutDesc out, int n_samples, hipStream_t st);
// One-stage 1x1 kernels (conv_1x1.hip): 16 | Cin, Cout <= 128, H*W a multiple of 64; same tune bit (MFVI_TUNE_SM) on a 1x1 layer; -2: shape not served
int launch_conv1_fwd_small(const TView& in, const ConvGeom& g, const float* w, long long wstride, OutDesc out, int n_samples, hipStream_t st);
int launch_conv1_bwd_data_small(const GView& gy, const ConvGeom& g, const float* w, long long wstride, int n_samples, hipStream_t st, const FoldFuse& fuse);
extern thread_local float* mfvi_tl_x6w;
extern thread_local bool mfvi_tl_x6w_ready;      // the pieces of this pass are already in the scratch (launch_x6_split_all)
long long x6_fwd_scratch_floats(const ConvGeom& g, int n_samples);
// one launch for all bf16x6 layers of a pass: dst_off = the layer's scratch offset (floats) in the arena, first_block = running block count
struct X6SplitEntry { long long w_off, dst_off; int Cin, Cout, COp, ncg, rem, units, first_block, pad; };
bool x6_split_entry(const ConvGeom& g, long long dst_off, X6SplitEntry* e);      // false: shape not served
int launch_x6_split_all(const X6SplitEntry* table_dev, int n_entries, int n_blocks, const float* w, long long wstride, int n_k, float* arena, hipStream_t st);
int launch_conv_fwd_x6(const TView& in, const ConvGeom& g, const float* w, long long wstride, OutDesc out, int tune, int n_samples, hipStream_t st);
// Backward-data WITH the fold of the 3x3 stride-1 layers with 16 / 32 / 64 output channels on maps a multiple of 64 wide, on the bf16
// matrix cores (conv_bwd_x6.hip).  tune = T | sr << 8 (strips per block, output rows per strip: 8 / 4 / 2 for 16 / 32 / 64 output channels).
// Scratch for the split weight pieces: x6_bwd_scratch_floats() floats, handed over in mfvi_tl_x6bw around the launch (nullptr: -2).
extern thread_local float* mfvi_tl_x6bw;
extern thread_local bool mfvi_tl_x6bw_ready;
long long x6_bwd_scratch_floats(const ConvGeom& g, int n_samples);
struct X6BSplitEntry { long long w_off, dst_off; int CI, CO, NF, NG, k16, units, first_block, rem_units; };      // rem_units: the (ky, c) operand of the last 4 input channels (conv_bwd_x6s.hip), behind the regular units
bool x6b_split_entry(const ConvGeom& g, long long dst_off, X6BSplitEntry* e);
int launch_x6b_split_all(const X6BSplitEntry* table_dev, int n_entries, int n_blocks, const float* w, long long wstride, int n_k, float* arena, hipStream_t st);
// strip-resident form for 32 (+4) -> 16 layers (conv_bwd_x6s.hip; bit 16 of the bf16x6 backward-data tiling); -3: shape not served
bool x6s_shape_ok(const ConvGeom& g);
int launch_conv_bwd_data_x6s(const GView& gy, const ConvGeom& g, const unsigned* wsp, long long wsp_stride_u4, int rem_off_u4, int T, int n_samples,
                             hipStream_t st, const FoldFuse& fuse);
int launch_conv_bwd_data_x6(const GView& gy, const ConvGeom& g, const float* w, long long wstride, int tune, int n_samples, hipStream_t st, const FoldFuse& fuse);
// One launch per pass: W[k][j] = mu[j] + softplus(rho[j]) * eps_k[j] for the weights and biases of every layer in the table.
struct SampleEntry { long long w_off, b_off; int n_w, n_b, layer_id, first_block; };
// bf16: mu / rho point to bf16_t arrays; sample = 0 writes W = mu (RTLayer's eval branch) — callers then launch it for ONE sample
int launch_sample_weights(const SampleEntry* table_dev, int n_entries, int n_blocks, const void* mu, const void* rho, RngKey key,
                          int n_samples, float* wsamp, long long wstride, hipStream_t st, int bf16 = 0, int sample = 1, double* zero = nullptr, long long n_zero = 0);      // zero: n_zero doubles cleared by the same launch
// bf16 -> float32 expansion (the generic fp32 kernels of shapes the MFMA path does not serve read float32 mu / rho)
int launch_expand_bf16(const void* src, long long n, float* dst, hipStream_t st);
constexpr int SAMPLE_QUADS = 256;       // weight quads per block of the sampling kernel
// The MFMA backward-weight kernel writes per-(pixel strip, sample) partial sums of dW (and of the bias gradient) with plain
// stores: part.base[(strip * n_samples + k) * part.stride + j], j < n_w weights then n_b biases; launch_grad_finalize reduces
// them, multiplies by eps * sigmoid(rho) per sample and accumulates into dmu / drho — no atomics, deterministic.
struct BwwPart { float* base; long long stride; int max_strips; };
int launch_conv_bwd_weight_mfma(const TView& in, const GView& gy, const ConvGeom& g, BwwPart part, int* strips_used, int n_samples,
                                hipStream_t st);
// 3x3 stride-1 layers on maps whose width is a multiple of 64, on the bf16 matrix cores with three-way split operands (conv_bww_x6.hip):
// cof = 16-channel output fragments per block (1 / 2), target = block-count target; same slabs as launch_conv_bwd_weight_mfma
int launch_conv_bwd_weight_x6(const TView& in, const GView& gy, const ConvGeom& g, BwwPart part, int* strips_used, int cof, int target,
                              int n_samples, hipStream_t st);
struct GradFinEntry { long long w_off, b_off, part_off, stride; int n_w, n_b, strips, layer_id, first_block, pad; };
// wsamp (optional): the sampled-weight slab of this pass, sample k at wsamp + k*wstride; then eps_k*softplus(rho) is read as W_k - mu
struct BnGradEntry { long long bsums_off; long long bn_off; int C; int hw; };      // hw = H * W of the normalised tensor
// BatchNorm parameter gradients of ONE table entry from the accumulated BN-backward sums: d gamma = sum ga * xhat, d beta = sum ga
// (a block's worth of work: bn_param_grads_kernel, or an extra block of grad_finalize_kernel)
__device__ __forceinline__ void bn_param_grads_entry(const BnGradEntry e, const double* __restrict__ bsums_base, int n_samples, float* __restrict__ dbn)
{
    for (int c = threadIdx.x; c < e.C; c += blockDim.x) {
        double sb = 0, sg = 0;
        for (int k = 0; k < n_samples; ++k) {
            const double* s = bsums_base + e.bsums_off + ((long long)k * e.C + c) * 2;
            sb += s[0]; sg += s[1];
        }
        dbn[e.bn_off + c] += (float)sg;            // d gamma = sum ga * xhat
        dbn[e.bn_off + e.C + c] += (float)sb;      // d beta  = sum ga
    }
}
// bn_table / n_bn / bsums_base / dbn (optional): the BatchNorm parameter gradients as n_bn extra blocks of the same launch
int launch_grad_finalize(const GradFinEntry* table_dev, int n_entries, int n_blocks, const float* part_base, const void* rho, RngKey key,
                         int sample_weights, int n_samples, float* dmu, float* drho, const float* wsamp, long long wstride, const void* mu,
                         hipStream_t st, int bf16 = 0, const BnGradEntry* bn_table = nullptr, int n_bn = 0, const double* bsums_base = nullptr, float* dbn = nullptr);
constexpr int GRAD_FIN_QUADS = 64;      // weight quads per block of the finalize kernel
// mul2v: the source is the gradient wrt view(X)**2 (variance convolution of an LRT layer): it enters with the factor 2 * view(X)
struct FoldSrc { const float* d; long long sstride; int pad; int mul2v; };
constexpr int MAX_FOLD_SRC = 4;      // two consumers, each an RT (one source) or an LRT (two sources) convolution
struct FoldSrcs { FoldSrc s[MAX_FOLD_SRC]; int n; };
// ga_X = act'(X) * fold(sum of sources); accumulates BN-backward sums of X.
int launch_finalize_dx(const FoldSrc* srcs, int n_src, const TView& x, float* ga, long long ga_sstride, double* bsums,
                       int n_samples, hipStream_t st);
int launch_finalize_dx_inline1x1(const FoldSrc& s0, const GView& g1, const float* w1, long long w1_sstride, int cs1, const TView& x, float* ga,
                                 long long ga_sstride, double* bsums, int n_samples, hipStream_t st);
// ---- local reparameterisation (LRTLayer.forward, BayTorch/modules/reparam_layers.py:59-72) around two ordinary convolutions ----
// sig2[j] = softplus(rho[j])^2 for j in [0, n): the weights / bias of the variance convolution
int launch_lrt_sigma2(const float* rho, long long n, float* sig2, hipStream_t st);
// y = a + sqrt(1e-16 + s2) * eps, eps = N(0,1) of RNG domain LRT, stream layer_id, sample k0 + k, element index over [C][HW];
// accumulates the statistics of the BatchNorm that follows y
int launch_lrt_combine(const float* a, const float* s2, long long sstride, int C, long long HW, RngKey key, int layer_id, OutDesc y,
                       int n_samples, hipStream_t st);
// ds2 = dy * eps / (2 sqrt(1e-16 + s2)), dy formed from gy (BN-backward on load)
int launch_lrt_ds2(const GView& gy, const float* s2, long long sstride, RngKey key, int layer_id, float* ds2, int n_samples, hipStream_t st);
// drho[j] += dsig2[j] * 2 softplus(rho[j]) sigmoid(rho[j])
int launch_lrt_drho(const float* dsig2, const float* rho, long long n, float* drho, hipStream_t st);
int launch_concat_up_fwd(const TView* a, const TView& b, OutDesc out, int nearest, int n_samples, hipStream_t st);
int launch_concat_up_bwd(const GView& gc, const TView* a, float* ga_a, long long ga_a_sstride, double* bsums_a,
                         const TView& b, float* ga_b, long long ga_b_sstride, double* bsums_b, int nearest, int n_samples, hipStream_t st);
// Dropout2d factors of one forward: arena[e.drop_off + k*C + c] for every entry, sample k, channel c (RNG domain 5, stream layer_id)
struct DropEntry { long long drop_off; int C, layer_id; float p; int pad; };
int launch_dropout_masks(const DropEntry* table_dev, int n_entries, RngKey key, int n_samples, float* arena, hipStream_t st);
// nn.BatchNorm2d's running statistics after n_samples training forwards (batch sums in fstats), and their use in eval mode
int launch_bn_update_running(const BnGradEntry* table_dev, int n_entries, int max_c, const double* fstats_base, int n_samples, float momentum,
                             float* running, hipStream_t st);
int launch_bn_eval_fill(const BnGradEntry* table_dev, int n_entries, int max_c, double* fstats_base, int n_samples, const float* running, hipStream_t st);
int launch_bn_param_grads(const BnGradEntry* table_dev, int n_entries, int max_c, const double* bsums_base, int n_samples,
                          float* dbn, hipStream_t st);
