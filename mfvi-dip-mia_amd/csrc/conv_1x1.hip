// conv_1x1.hip — 1x1 layers with 32 ... 128 channels on the small and middle maps (round 4): one stage, everything in flight.
//
// Reference op: BayTorch/modules/reparam_layers.py:26-37 with a 1x1 filter (ReflectionPad2d(0), models/common.py:100-135) — the `up` 1x1
// convolutions of skip() (models/skip.py:110-119): 128 -> 128 @16^2 / 32^2, 64 -> 64 @64^2, 32 -> 32 @128^2 at the BASELINE configs.
//
// On those shapes the staged kernel (conv_mfma.hip, 32-channel stages) is a chain of 2-4 stages of [global load -> LDS -> barrier -> a few
// MFMAs] for a GEMM of half a GFLOP: 17-28 us per launch where the matrix work is 2 us (profiles/r04_kernel_table_full.txt).  Here, as in
// conv_small.hip, a block's WHOLE reduction is in LDS: a block = (sample, 64 consecutive pixels, ALL output channels),
//   pixels  [RED / 16 groups][64 pixels][16]      (deferred BN + LeakyReLU resp. BN-backward applied on load)
//   weights [RED / 16 groups][MOUT channels][16]  (W_k of the sample from the slab; MODE 1: transposed)
// every global load is issued before anything waits, there is ONE barrier in front of the matrix phase, and the 8 waves split the output
// fragments (wave = fragment w % NFR, pixel part w / NFR).  The 16 floats of an entry are ordered [channel mod 4][k-step] so that one
// ds_read_b128 is a lane's operand of four consecutive v_mfma_f32_16x16x4_f32 k-steps.
//
// MODE 0 forward: bias, raw output, BN statistics (16-lane DPP sums, one fp64 atomic per (block, channel, moment)).
// MODE 1 backward-data with the fold of the input tensor (as conv_mfma.hip's fused 1x1 path): the reduction runs over the layer's OUTPUT
//        channels (dy formed from ga / y on load), the epilogue multiplies by LeakyReLU'(view(x)), accumulates the BN-backward sums of x and writes ga.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct C1Bwd { float qc, c1, k3, pad; };      // dy = ga * c1 + (y * qc + k3)

struct C1Args {
    TView xin; GView gin; ConvGeom g; const float* w; long long wstride; OutDesc out;
    float* fga; long long fga_sstride; double* fbsums;
    int nx, nz;        // 64-pixel tiles per sample, samples
};

constexpr int C1_MAXC = 128;

__device__ __forceinline__ float c1_row_sum16(float v)      // sum over the 16 lanes of a DPP row, total in lane 15
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}

template <int MODE>
__global__ __launch_bounds__(512) void conv1_sm_kernel(C1Args A)
{
    extern __shared__ __align__(16) float s_dyn[];          // pixels | weights
    __shared__ ChanFwd s_ch[C1_MAXC];                       // MODE 0: view of the input; MODE 1: view of the tensor being folded (the block's MOUT channels)
    __shared__ C1Bwd s_chb[MODE == 1 ? C1_MAXC : 1];
    __shared__ float s_bias[C1_MAXC];
    __shared__ float s_red[4][C1_MAXC][2];                  // [pixel part][channel][moment]

    const ConvGeom& g = A.g;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    int bx, by, k;
    xcd_decode(blockIdx.x, A.nx, 1, A.nz, bx, by, k);
    const int HW = g.H * g.W, p0 = bx * 64;
    const int RED = MODE == 0 ? g.Cin : g.Cout, MOUT = MODE == 0 ? g.Cout : g.Cin;
    const int NG = RED >> 4, NFR = MOUT >> 4;               // reduction groups of 16, output fragments (2, 4 or 8)
    float* __restrict__ s_x = s_dyn;
    float* __restrict__ s_w = s_dyn + NG * 64 * 16;
    const float* __restrict__ wk = A.w + (long long)k * A.wstride + g.w_off;

    // ---- every global load of the block, issued before anything waits ----
    // pixels: thread = (pixel tid & 63, part tid >> 6): residue cb = part & 3, groups G = part >> 2, + 2, ...: a wave's load is 64 consecutive pixels of one channel
    const float* __restrict__ xk = MODE == 0 ? A.xin.data + (long long)k * A.xin.sstride : A.gin.ga + (long long)k * A.gin.gstride;
    const float* __restrict__ yk = (MODE == 1 && A.gin.stats) ? A.gin.y + (long long)k * A.gin.ystride : nullptr;
    const int px = tid & 63, cb = wv & 3, gh = wv >> 2;
    constexpr int NGH = C1_MAXC / 16 / 2;                   // groups per thread (at most)
    float xr[NGH][4], yr[MODE == 1 ? NGH : 1][4];
#pragma unroll
    for (int j = 0; j < NGH; ++j) {
        const int G = gh + 2 * j;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = min(16 * G + cb + 4 * i, RED - 1);
            const bool ok = G < NG;
            xr[j][i] = ok ? xk[(long long)c * HW + p0 + px] : 0.f;
            if (MODE == 1) yr[j][i] = (ok && yk) ? yk[(long long)c * HW + p0 + px] : 0.f;
        }
    }
    // MODE 1: raw values of the tensor being folded at this lane's outputs (wave = fragment fr, pixel part ph)
    const int fr = wv % NFR, ph = wv / NFR, PF = NFR >> 1;   // PF pixel fragments per wave (8 / NFR parts of 64 pixels)
    float xq[4][4];
    const bool fold_sums = MODE == 1 && A.fbsums != nullptr, fold_x = MODE == 1 && (A.fbsums != nullptr || (A.xin.act & 1));
    if (fold_x) {
        const float* __restrict__ xin = A.xin.data + (long long)k * A.xin.sstride + p0 + l15;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int q = 0; q < 4; ++q) xq[f][q] = f < PF ? xin[(long long)(16 * fr + 4 * l4 + q) * HW + 16 * (ph * PF + f)] : 0.f;
    }
    // weights: item = (group G, output channel m): the 16 reduction channels 16 G .. 16 G + 15 of row m
    //   MODE 0: W[m][16 G ..] is contiguous (four float4);  MODE 1: W[16 G + j][m], j = 0..15 (consecutive lanes = consecutive m: coalesced)
    const int NWI = NG * MOUT;                              // <= 1024
    float wt[2][16];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int it = min(tid + 512 * j, NWI - 1), G = it / MOUT, m = it - G * MOUT;
        if (MODE == 0) {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(wk + (long long)m * g.Cin + 16 * G);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float4 v = src[q]; wt[j][4 * q] = v.x; wt[j][4 * q + 1] = v.y; wt[j][4 * q + 2] = v.z; wt[j][4 * q + 3] = v.w; }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) wt[j][q] = wk[(long long)(16 * G + q) * g.Cin + m];
        }
    }
    // per-channel constants
    if (MODE == 0) {
        for (int c = tid; c < RED; c += 512) s_ch[c] = chan_fwd(A.xin, k, c);
        for (int c = tid; c < MOUT; c += 512) s_bias[c] = g.b_off >= 0 ? wk[g.b_off - g.w_off + c] : 0.f;
    } else {
        for (int c = tid; c < RED; c += 512) { const ChanBwd b = chan_bwd(A.gin, k, c); C1Bwd r; r.qc = -b.c1 * b.c3 * b.rstd; r.c1 = b.c1; r.k3 = __builtin_fmaf(-b.mean, r.qc, -b.c1 * b.c2); r.pad = 0.f; s_chb[c] = r; }
        if (fold_x) for (int c = tid; c < MOUT; c += 512) s_ch[c] = chan_fwd(A.xin, k, c);
    }
    for (int i = tid; i < 4 * C1_MAXC * 2; i += 512) (&s_red[0][0][0])[i] = 0.f;
    __syncthreads();
    const int act = A.xin.act; const float slope = A.xin.slope;
#pragma unroll
    for (int j = 0; j < NGH; ++j) {
        const int G = gh + 2 * j;
        if (G < NG) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 16 * G + cb + 4 * i;
                if (MODE == 0) v[i] = apply_fwd(s_ch[c], xr[j][i], act, slope);
                else { const C1Bwd kb = s_chb[c]; v[i] = yk ? __builtin_fmaf(xr[j][i], kb.c1, __builtin_fmaf(yr[MODE == 1 ? j : 0][i], kb.qc, kb.k3)) : xr[j][i]; }
            }
            *reinterpret_cast<float4*>(&s_x[((G * 64 + px) << 4) + (cb << 2)]) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int it = tid + 512 * j, G = it / MOUT, m = it - G * MOUT;
        if (it < NWI) {
            float* __restrict__ d = s_w + ((G * MOUT + m) << 4);
            // entry order [channel mod 4][k-step]: float4 r = channels r, r + 4, r + 8, r + 12 of the group
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<float4*>(d + 4 * r) = make_float4(wt[j][r], wt[j][r + 4], wt[j][r + 8], wt[j][r + 12]);
        }
    }
    __syncthreads();

    // ---- matrix phase: wave = (output fragment fr, pixel part ph), PF pixel fragments; D[m = channel 16 fr + 4 l4 + q][n = pixel 16 pf + l15] ----
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int G = 0; G < NG; ++G) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(s_w + ((G * MOUT + 16 * fr + l15) << 4) + (l4 << 2));
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (f < PF) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(s_x + ((G * 64 + 16 * (ph * PF + f) + l15) << 4) + (l4 << 2));
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i], b4[i], acc[f], 0, 0, 0);
            }
        }
    }
    // ---- epilogue ----
    const bool do_stats = MODE == 0 ? A.out.stats != nullptr : fold_sums;
    float* __restrict__ yout = (MODE == 0 ? A.out.data + (long long)k * A.out.sstride : A.fga + (long long)k * A.fga_sstride) + p0 + l15;
    const int xact = A.xin.act; const float xslope = A.xin.slope;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ch = 16 * fr + 4 * l4 + q;
        float a_ = 0.f, b_ = 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (f < PF) {
                float v = acc[f][q];
                if (MODE == 0) { v += s_bias[ch]; a_ += v; b_ = __builtin_fmaf(v, v, b_); }
                else if (fold_x) {
                    const ChanFwd cf = s_ch[ch];
                    const float ym = xq[f][q] - cf.mean;
                    if (xact & 1) { const float vv = __builtin_fmaf(ym, cf.scale, cf.beta); v *= (vv > 0.f) ? 1.f : xslope; }
                    a_ += v; b_ = __builtin_fmaf(v, ym * cf.rstd, b_);
                }
                yout[(long long)ch * HW + 16 * (ph * PF + f)] = v;
            }
        }
        if (do_stats) {
            a_ = c1_row_sum16(a_); b_ = c1_row_sum16(b_);
            if (l15 == 15) { s_red[ph][ch][0] = a_; s_red[ph][ch][1] = b_; }      // (wave, 16-lane group) owns (part, channel): plain stores
        }
    }
    if (do_stats) {
        __syncthreads();
        for (int i = tid; i < MOUT * 2; i += 512) {
            const int ch = i >> 1, which = i & 1;
            const float v = (s_red[0][ch][which] + s_red[1][ch][which]) + (s_red[2][ch][which] + s_red[3][ch][which]);
            atomicAdd((MODE == 0 ? A.out.stats : A.fbsums) + ((long long)k * MOUT + ch) * 2 + which, (double)v);
        }
    }
}

bool c1_shape_ok(const ConvGeom& g)
{
    if (g.ks != 1 || g.stride != 1 || (g.w_off & 3)) return false;
    if ((g.Cin & 15) || (g.Cout & 15) || g.Cin > C1_MAXC || g.Cout > C1_MAXC) return false;
    if (((long long)g.H * g.W) & 63) return false;
    return (long long)max(g.Cin, g.Cout) * g.H * g.W < (1LL << 30);
}

int c1_launch(int mode, C1Args& A, int n_samples, hipStream_t st)
{
    const ConvGeom& g = A.g;
    const int RED = mode == 0 ? g.Cin : g.Cout, MOUT = mode == 0 ? g.Cout : g.Cin;
    const int nfr = MOUT >> 4;
    if (nfr != 2 && nfr != 4 && nfr != 8) return -2;
    if (RED * MOUT / 16 > 1024) return -2;                 // weight items of two per thread
    A.nx = (g.H * g.W) >> 6; A.nz = n_samples;
    const size_t lds_bytes = sizeof(float) * ((size_t)(RED >> 4) * 64 * 16 + (size_t)(RED >> 4) * MOUT * 16);
    if (lds_bytes > 144 * 1024) return -2;
    static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_sm_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_sm_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    if (attr0 != hipSuccess || attr1 != hipSuccess) return (int)(attr0 != hipSuccess ? attr0 : attr1);
    mfvi_tl_family = 4;
    if (mode == 0) mfvi_launch(conv1_sm_kernel<0>, dim3(A.nx * A.nz), dim3(512), lds_bytes, st, A);
    else mfvi_launch(conv1_sm_kernel<1>, dim3(A.nx * A.nz), dim3(512), lds_bytes, st, A);
    return (int)hipGetLastError();
}

// ---- streaming forward for the narrow 1x1 layers of the top scales (round 4): 16 -> 16 / 4 / 2 @256^2 ----
// These launches are memory streams (67 MB in, 67 MB out for 16 -> 16 @256^2, K = 16) that the staged kernel runs at 2.9-3.7 TB/s: a tile goes
// global -> registers -> LDS -> barrier -> matrix -> store, a few tiles in flight per CU.  Here nothing goes through LDS: a wave owns groups
// of 64 consecutive pixels; lane (k = lane >> 4, n = lane & 15) loads the float4 x[4 s + k][p0 + 4 n ..] straight into the B operands of four
// v_mfma_f32_16x16x4_f32 (deferred BN + LeakyReLU in the register), the A operand W[co = n][4 s + k] stays in registers, the results get
// their bias and leave as float4 of consecutive pixels (256-byte channel segments both ways); UNR groups' loads are in flight at once.  BN statistics: per-lane partials, one
// 16-lane DPP reduction at the end, LDS across the block's waves, one fp64 atomic per (block, channel, moment).
constexpr int C1S_MAXNS = 16;      // reduction channels / 4

__device__ __forceinline__ float c1s_row_sum16(float v) { return c1_row_sum16(v); }

template <int NS, int UNR, int NF>      // reduction channels / 4, groups in flight per wave, output fragments of 16 channels (1 or 2)
__global__ __launch_bounds__(256) void conv1_stream_kernel(C1Args A)
{
    __shared__ ChanFwd s_ch[4 * C1S_MAXNS];
    __shared__ float s_part[4][16 * NF][2];
    const ConvGeom& g = A.g;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int k = blockIdx.y;
    const int HW = g.H * g.W, Cin = g.Cin, Cout = g.Cout;
    const float* __restrict__ wk = A.w + (long long)k * A.wstride + g.w_off;
    const float* __restrict__ xk = A.xin.data + (long long)k * A.xin.sstride;
    float* __restrict__ yk = A.out.data + (long long)k * A.out.sstride;
    for (int c = tid; c < Cin; c += 256) s_ch[c] = chan_fwd(A.xin, k, c);
    // A operand: W[co = 16 f + l15][ci = 4 s + l4]; bias of this lane's four output channels per fragment
    float a[NF][NS], bias[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int s = 0; s < NS; ++s) a[f][s] = 16 * f + l15 < Cout ? wk[(long long)(16 * f + l15) * Cin + 4 * s + l4] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[f][r] = (g.b_off >= 0 && 16 * f + 4 * l4 + r < Cout) ? wk[g.b_off - g.w_off + 16 * f + 4 * l4 + r] : 0.f;
    }
    __syncthreads();
    float ksc[NS], ksh[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { const ChanFwd f = s_ch[4 * s + l4]; ksc[s] = f.scale; ksh[s] = __builtin_fmaf(-f.mean, f.scale, f.beta); }
    const bool lrelu = (A.xin.act & 1) != 0; const float slope = A.xin.slope;
    // a group = 64 consecutive pixels: lane (k, n) loads the float4 x[4 s + k][p0 + 4 n .. 4 n + 3] (16 lanes = 256 contiguous bytes of a channel),
    // component j of it is the B operand of the j-th of four matrix instructions, whose result register r is y[4 k + r][p0 + 4 n + j]:
    // the four results of a channel are again a float4 of consecutive pixels
    const int n_groups = HW >> 6;
    const int gstride = gridDim.x * 4;
    float s1[NF][4], s2[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[f][r] = 0.f; s2[f][r] = 0.f; }
    const bool do_stats = A.out.stats != nullptr;
    for (int g0 = blockIdx.x * 4 + wv; g0 < n_groups; g0 += gstride * UNR) {
        float4 xv[UNR][NS];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int gi = min(g0 + u * gstride, n_groups - 1);      // (clamped: every load unconditional; groups past the end are not stored)
#pragma unroll
            for (int s = 0; s < NS; ++s) xv[u][s] = *reinterpret_cast<const float4*>(xk + (long long)(4 * s + l4) * HW + 64 * gi + 4 * l15);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int gi = g0 + u * gstride;
            f32x4 acc[NF][4];
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float v[4] = {xv[u][s].x, xv[u][s].y, xv[u][s].z, xv[u][s].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = __builtin_fmaf(v[j], ksc[s], ksh[s]);
                    if (lrelu) v[j] = __builtin_fmaxf(v[j], v[j] * slope);
#pragma unroll
                    for (int f = 0; f < NF; ++f) acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f][s], v[j], acc[f][j], 0, 0, 0);
                }
            }
            if (gi >= n_groups) continue;
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y0 = acc[f][0][r] + bias[f][r], y1 = acc[f][1][r] + bias[f][r], y2 = acc[f][2][r] + bias[f][r], y3 = acc[f][3][r] + bias[f][r];
                    if (16 * f + 4 * l4 + r < Cout) *reinterpret_cast<float4*>(yk + (long long)(16 * f + 4 * l4 + r) * HW + 64 * gi + 4 * l15) = make_float4(y0, y1, y2, y3);
                    s1[f][r] += (y0 + y1) + (y2 + y3); s2[f][r] = __builtin_fmaf(y0, y0, __builtin_fmaf(y1, y1, __builtin_fmaf(y2, y2, __builtin_fmaf(y3, y3, s2[f][r]))));
                }
        }
    }
    if (do_stats) {
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t1 = c1s_row_sum16(s1[f][r]), t2 = c1s_row_sum16(s2[f][r]);
                if (l15 == 15) { s_part[wv][16 * f + 4 * l4 + r][0] = t1; s_part[wv][16 * f + 4 * l4 + r][1] = t2; }
            }
        __syncthreads();
        if (tid < 2 * Cout) {
            const int c = tid >> 1, which = tid & 1;
            const float v = (s_part[0][c][which] + s_part[1][c][which]) + (s_part[2][c][which] + s_part[3][c][which]);
            atomicAdd(A.out.stats + ((long long)k * Cout + c) * 2 + which, (double)v);
        }
    }
}

}  // namespace

// streaming forward of a narrow 1x1 layer (tune bit 28): Cin a multiple of 4 up to 64, Cout <= 16, H*W a multiple of 16.  -2: shape not served
int launch_conv1_fwd_stream(const TView& in, const ConvGeom& g, const float* w, long long wstride, OutDesc out, int n_samples, hipStream_t st)
{
    if (g.ks != 1 || g.stride != 1 || (g.Cin & 3) || g.Cin > 4 * C1S_MAXNS || g.Cout > 32 || (((long long)g.H * g.W) & 63) || (in.act & MFVI_ACT_SQUARE)) return -2;
    if ((in.sstride & 3) || (out.sstride & 3) || (((uintptr_t)in.data | (uintptr_t)out.data) & 15)) return -2;      // float4 rows
    if ((long long)max(g.Cin, g.Cout) * g.H * g.W >= (1LL << 31)) return -2;
    C1Args A{};
    A.xin = in; A.g = g; A.w = w; A.wstride = wstride; A.out = out;
    const int n_groups = (g.H * g.W) >> 6;
    const int ns = g.Cin >> 2;
    if (ns != 1 && ns != 2 && ns != 3 && ns != 4 && ns != 8 && ns != 16) return -2;      // (instantiated reduction depths: 4 ... 16, 32, 64 channels)
    // blocks: enough to fill the chip four times over at most, each wave with at least one full unrolled batch where the map allows
    const int nb = max(1, min((n_groups + 7) / 8, (256 * 8 + n_samples - 1) / n_samples));
    mfvi_tl_family = 6;
    const dim3 grid(nb, n_samples);
#define C1S_GO(NS_, U_, NF_) mfvi_launch((conv1_stream_kernel<NS_, U_, NF_>), grid, dim3(256), 0, st, A)
    if (g.Cout > 16) {      // two output fragments: the 16 / 32 -> 32 layers
        if (ns == 4) C1S_GO(4, 2, 2); else if (ns == 8) C1S_GO(8, 1, 2); else return -2;
    }
    else if (ns <= 4) { if (ns == 4) C1S_GO(4, 2, 1); else if (ns == 3) C1S_GO(3, 2, 1); else if (ns == 2) C1S_GO(2, 4, 1); else C1S_GO(1, 4, 1); }
    else if (ns == 8) C1S_GO(8, 1, 1);
    else C1S_GO(16, 1, 1);
#undef C1S_GO
    return (int)hipGetLastError();
}

// -2: shape not served
int launch_conv1_fwd_small(const TView& in, const ConvGeom& g, const float* w, long long wstride, OutDesc out, int n_samples, hipStream_t st)
{
    if (!c1_shape_ok(g) || (in.act & MFVI_ACT_SQUARE)) return -2;
    if ((g.Cin & 3) || (uintptr_t)(w + g.w_off) & 15) return -2;      // float4 weight rows
    C1Args A{};
    A.xin = in; A.g = g; A.w = w; A.wstride = wstride; A.out = out;
    return c1_launch(0, A, n_samples, st);
}

int launch_conv1_bwd_data_small(const GView& gy, const ConvGeom& g, const float* w, long long wstride, int n_samples, hipStream_t st, const FoldFuse& fuse)
{
    if (!c1_shape_ok(g) || !fuse.ga) return -2;
    C1Args A{};
    A.xin = fuse.x; A.gin = gy; A.g = g; A.w = w; A.wstride = wstride;
    A.fga = fuse.ga; A.fga_sstride = fuse.ga_sstride; A.fbsums = fuse.bsums;
    return c1_launch(1, A, n_samples, st);
}
