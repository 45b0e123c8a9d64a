// K1 — fused reparameterised convolution, forward.
//
// Replaces, per Conv2dRT layer of the reference (BayTorch/modules/reparam_layers.py:26-37,
// BayTorch/modules/module.py:82-85, models/common.py:100-135):
//   softplus(rho), randn_like, mul, add (weight and bias)  -> generated in-kernel from (mu, rho, Philox eps)
//   ReflectionPad2d(k//2)                                  -> folded into the tile loader's index math
//   the preceding BatchNorm2d(train, N=1) + LeakyReLU(0.2) -> applied on load (TView)
//   F.conv2d(x, w, b, stride)                              -> LDS-tiled direct convolution
//   statistics of the following BatchNorm2d                -> per-channel sum / sum^2 epilogue (fp64 atomics)
//
// This file holds the generic fp32 VALU path (any Cin/Cout, 1x1 and 3x3, stride 1/2).
#include "common.h"

namespace {

template <int KS, int STRIDE>
struct FwdCfg {
    static constexpr int TW = 32;
    static constexpr int PPT = (STRIDE == 1) ? 2 : 1;          // output rows per thread: ty, ty+8
    static constexpr int TH = 8 * PPT;
    static constexpr int CT = 16;                               // output channels per block
    static constexpr int CC = 8;                                // input channels per LDS stage
    static constexpr int P = KS / 2;
    static constexpr int IN_TH = (TH - 1) * STRIDE + KS;
    static constexpr int IN_TW = (TW - 1) * STRIDE + KS;
    static constexpr int IN_TWP = IN_TW | 1;                    // odd row pitch
};

template <int KS, int STRIDE>
__global__ __launch_bounds__(256) void conv_fwd_kernel(TView in, ConvGeom g, const float* __restrict__ mu,
                                                       const float* __restrict__ rho, RngKey key, int sample_weights,
                                                       OutDesc out, int tiles_x)
{
    key = key_now(key);
    using Cfg = FwdCfg<KS, STRIDE>;
    constexpr int TW = Cfg::TW, PPT = Cfg::PPT, TH = Cfg::TH, CT = Cfg::CT, CC = Cfg::CC, P = Cfg::P;
    constexpr int IN_TH = Cfg::IN_TH, IN_TW = Cfg::IN_TW, IN_TWP = Cfg::IN_TWP, KK = KS * KS;

    __shared__ float s_in[CC][IN_TH][IN_TWP];
    __shared__ __align__(16) float s_w[CC][KK][CT];
    __shared__ ChanFwd s_ch[MFVI_MAX_C];
    __shared__ float s_bias[CT];
    __shared__ double s_red[4][CT][2];

    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    const int k = blockIdx.z;
    const int co0 = blockIdx.y * CT;
    const int ox0 = (blockIdx.x % tiles_x) * TW, oy0 = (blockIdx.x / tiles_x) * TH;
    const int H = g.H, W = g.W, Cin = g.Cin, Cout = g.Cout;

    RngKey kw = key; kw.sample += (uint32_t)k; kw.stream = ((uint32_t)DOMAIN_EPS << 24) | (uint32_t)(2 * g.layer_id);
    RngKey kb = kw; kb.stream += 1u;

    for (int c = t; c < Cin; c += 256) s_ch[c] = chan_fwd(in, k, c);
    if (t < CT) {
        const int co = co0 + t;
        float b = 0.f;
        if (co < Cout && g.b_off >= 0) {
            b = mu[g.b_off + co];
            if (sample_weights) {
                float z[4]; spec_normal4(kb, (uint32_t)(co >> 2), z);
                b += softplus_f(rho[g.b_off + co]) * z[co & 3];
            }
        }
        s_bias[t] = b;
    }

    float acc[PPT][CT];
#pragma unroll
    for (int p = 0; p < PPT; ++p)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[p][c] = 0.f;

    const float* __restrict__ xin = in.data + (long long)k * in.sstride;
    const long long HW = (long long)H * W;

    for (int ci0 = 0; ci0 < Cin; ci0 += CC) {
        __syncthreads();
        const int cc = min(CC, Cin - ci0);
        // ---- stage the input tile: reflection pad + deferred BN/LeakyReLU ----
        for (int idx = t; idx < cc * IN_TH * IN_TW; idx += 256) {
            const int c = idx / (IN_TH * IN_TW), r = idx - c * (IN_TH * IN_TW);
            const int iy = r / IN_TW, ix = r - iy * IN_TW;
            int gy = reflect_idx(oy0 * STRIDE + iy - P, H), gx = reflect_idx(ox0 * STRIDE + ix - P, W);
            gy = min(max(gy, 0), H - 1); gx = min(max(gx, 0), W - 1);          // tile overhang (masked at the store)
            const float v = xin[(long long)(ci0 + c) * HW + (long long)gy * W + gx];
            s_in[c][iy][ix] = apply_fwd(s_ch[ci0 + c], v, in.act, in.slope);
        }
        // ---- sample the weight slab w[co0..co0+CT)[ci0..ci0+cc)[KK] = mu + softplus(rho) * eps ----
        {
            const int len = cc * KK;
            const int G = (len >> 2) + 2;                                       // Philox blocks touching one row
            for (int idx = t; idx < CT * G; idx += 256) {
                const int co_l = idx / G, gi = idx - co_l * G;
                const int co = co0 + co_l;
                if (co < Cout) {
                    const long long j0 = ((long long)co * Cin + ci0) * KK;
                    const long long blk = (j0 >> 2) + gi, jb = blk << 2;
                    if (jb < j0 + len) {
                        float z[4] = {0.f, 0.f, 0.f, 0.f};
                        if (sample_weights) spec_normal4(kw, (uint32_t)blk, z);
#pragma unroll
                        for (int l = 0; l < 4; ++l) {
                            const long long j = jb + l;
                            if (j >= j0 && j < j0 + len) {
                                const int rel = (int)(j - j0), c = rel / KK, tap = rel - c * KK;
                                float w = mu[g.w_off + j];
                                if (sample_weights) w += softplus_f(rho[g.w_off + j]) * z[l];
                                s_w[c][tap][co_l] = w;
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        const int rel = gi * 4 + l;
                        if (rel < len) { const int c = rel / KK, tap = rel - c * KK; s_w[c][tap][co_l] = 0.f; }
                    }
                }
            }
        }
        __syncthreads();
        // ---- accumulate ----
        for (int c = 0; c < cc; ++c) {
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const float4* wp = reinterpret_cast<const float4*>(&s_w[c][ky * KS + kx][0]);
                    const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
                    const float wv[CT] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w,
                                          w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
#pragma unroll
                    for (int p = 0; p < PPT; ++p) {
                        const float xv = s_in[c][(ty + 8 * p) * STRIDE + ky][tx * STRIDE + kx];
#pragma unroll
                        for (int q = 0; q < CT; ++q) acc[p][q] = __builtin_fmaf(xv, wv[q], acc[p][q]);
                    }
                }
        }
    }

    // ---- epilogue: bias, store raw output, BN statistics of the output ----
    // statistics in double: E[y^2] - E[y]^2 must survive channels whose mean dominates their spread
    double sum[CT], sq[CT];
#pragma unroll
    for (int q = 0; q < CT; ++q) { sum[q] = 0.0; sq[q] = 0.0; }
    float* __restrict__ yout = out.data + (long long)k * out.sstride;
    const long long HWo = (long long)g.Ho * g.Wo;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        const int oy = oy0 + ty + 8 * p, ox = ox0 + tx;
        if (oy < g.Ho && ox < g.Wo) {
#pragma unroll
            for (int q = 0; q < CT; ++q) {
                if (co0 + q < Cout) {
                    const float v = acc[p][q] + s_bias[q];
                    yout[(long long)(co0 + q) * HWo + (long long)oy * g.Wo + ox] = v;
                    sum[q] += (double)v; sq[q] += (double)v * (double)v;
                }
            }
        }
    }
    if (out.stats != nullptr) {
        const int lane = t & 63, wv = t >> 6;
#pragma unroll
        for (int q = 0; q < CT; ++q) {
            const double a = wave_sum_d(sum[q]), b = wave_sum_d(sq[q]);
            if (lane == 0) { s_red[wv][q][0] = a; s_red[wv][q][1] = b; }
        }
        __syncthreads();
        if (t < CT * 2) {
            const int q = t >> 1, which = t & 1;
            if (co0 + q < Cout) {
                const double v = s_red[0][q][which] + s_red[1][q][which] + s_red[2][q][which] + s_red[3][q][which];
                atomicAdd(out.stats + ((long long)k * Cout + co0 + q) * 2 + which, v);
            }
        }
    }
}

}  // namespace

int launch_conv_fwd(const TView& in, const ConvGeom& g, const float* mu, const float* rho, RngKey key, int sample_weights,
                    OutDesc out, int n_samples, hipStream_t st)
{
    if (g.Cin > MFVI_MAX_C) { set_error("conv_fwd: Cin %d > %d", g.Cin, MFVI_MAX_C); return -1; }
#define LAUNCH(KS_, S_)                                                                                              \
    {                                                                                                                \
        using Cfg = FwdCfg<KS_, S_>;                                                                                 \
        const int tiles_x = (g.Wo + Cfg::TW - 1) / Cfg::TW, tiles_y = (g.Ho + Cfg::TH - 1) / Cfg::TH;                \
        dim3 grid(tiles_x * tiles_y, (g.Cout + Cfg::CT - 1) / Cfg::CT, n_samples);                                   \
        hipLaunchKernelGGL((conv_fwd_kernel<KS_, S_>), grid, dim3(256), 0, st, in, g, mu, rho, key, sample_weights, \
                           out, tiles_x);                                                                            \
    }
    if (g.ks == 3 && g.stride == 1) LAUNCH(3, 1)
    else if (g.ks == 3 && g.stride == 2) LAUNCH(3, 2)
    else if (g.ks == 1 && g.stride == 1) LAUNCH(1, 1)
    else if (g.ks == 5 && g.stride == 1) LAUNCH(5, 1)
    else if (g.ks == 5 && g.stride == 2) LAUNCH(5, 2)
    else { set_error("conv_fwd: unsupported ksize %d stride %d", g.ks, g.stride); return -1; }
#undef LAUNCH
    return (int)hipGetLastError();
}
