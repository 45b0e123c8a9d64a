// K2b — fused reparameterised convolution, backward wrt (mu, rho) (generic fp32 VALU path).
//
// Autograd of w = mu + softplus(rho)*eps followed by conv2d (BayTorch/modules/module.py:82-85,
// reparam_layers.py:28-37):   dW[co][ci][tap] = sum_pix dy[co][pix] * xpad[ci][S*pix + tap]
//   d mu  += dW                                (summed over the MC samples of this call)
//   d rho += dW * eps * sigmoid(rho)           (eps re-derived from the counter RNG, never stored)
// and the same for the bias with db[co] = sum_pix dy[co][pix].
// Both operands are formed on load: xpad = reflection pad of LeakyReLU(BN(x_raw)) (TView), dy = BN-backward
// of ga (GView).
#include "common.h"

namespace {

template <int KS, int STRIDE>
struct BwwCfg {
    static constexpr int TW = 32;
    static constexpr int TH = (STRIDE == 1) ? 8 : 4;
    static constexpr int COT = 16, CIT = 16;
    static constexpr int IN_TH = (TH - 1) * STRIDE + KS;
    static constexpr int IN_TW = (TW - 1) * STRIDE + KS;
    static constexpr int X_PLANE = (IN_TH * IN_TW) | 1;     // odd plane pitch: 16 channels hit 16 banks
    static constexpr int G_PLANE = (TH * TW) | 1;
    static constexpr int ROW = CIT * KS * KS;               // dW elements per output channel row of this block
    static constexpr int STAGE = COT * G_PLANE + CIT * X_PLANE;
    static constexpr int EPI = 2 * COT * ROW;
    static constexpr int LDS_FLOATS = STAGE > EPI ? STAGE : EPI;
};

template <int KS, int STRIDE>
__global__ __launch_bounds__(256) void conv_bwd_weight_kernel(TView in, GView gy, ConvGeom g,
                                                              const float* __restrict__ rho, RngKey key,
                                                              int sample_weights, float* __restrict__ dmu,
                                                              float* __restrict__ drho, int tiles_x, int n_tiles,
                                                              int tiles_per_block, int ci_tiles)
{
    key = key_now(key);
    using Cfg = BwwCfg<KS, STRIDE>;
    constexpr int TW = Cfg::TW, TH = Cfg::TH, COT = Cfg::COT, CIT = Cfg::CIT, KK = KS * KS, P = KS / 2;
    constexpr int IN_TH = Cfg::IN_TH, IN_TW = Cfg::IN_TW, X_PLANE = Cfg::X_PLANE, G_PLANE = Cfg::G_PLANE, ROW = Cfg::ROW;

    __shared__ __align__(16) float lds[Cfg::LDS_FLOATS];
    __shared__ ChanFwd s_chx[CIT];
    __shared__ ChanBwd s_chg[COT];
    float* s_g = lds;                       // [COT][G_PLANE]
    float* s_x = lds + COT * G_PLANE;       // [CIT][X_PLANE]

    const int t = threadIdx.x, ci_l = t & 15, co_l = t >> 4;
    const int k = blockIdx.z;
    const int co0 = (blockIdx.y / ci_tiles) * COT, ci0 = (blockIdx.y % ci_tiles) * CIT;
    const int Cin = g.Cin, Cout = g.Cout, H = g.H, W = g.W, Ho = g.Ho, Wo = g.Wo;
    const bool do_bias = (ci0 == 0) && (g.b_off >= 0);

    if (t < CIT) { s_chx[t] = chan_fwd(in, k, min(ci0 + t, Cin - 1)); }
    if (t >= 64 && t < 64 + COT) { s_chg[t - 64] = chan_bwd(gy, k, min(co0 + t - 64, Cout - 1)); }

    float acc[KK];
#pragma unroll
    for (int q = 0; q < KK; ++q) acc[q] = 0.f;
    float bsum = 0.f;

    const float* __restrict__ xin = in.data + (long long)k * in.sstride;
    const float* __restrict__ gap = gy.ga + (long long)k * gy.gstride;
    const float* __restrict__ yp = gy.y ? gy.y + (long long)k * gy.ystride : nullptr;
    const long long HW = (long long)H * W, HWo = (long long)Ho * Wo;

    const int tile_begin = blockIdx.x * tiles_per_block, tile_end = min(n_tiles, tile_begin + tiles_per_block);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int ox0 = (tile % tiles_x) * TW, oy0 = (tile / tiles_x) * TH;
        __syncthreads();
        for (int idx = t; idx < COT * TH * TW; idx += 256) {
            const int c = idx / (TH * TW), r = idx - c * (TH * TW);
            const int oy = oy0 + r / TW, ox = ox0 + (r % TW), co = co0 + c;
            float v = 0.f;
            if (co < Cout && oy < Ho && ox < Wo) {
                const long long off = (long long)co * HWo + (long long)oy * Wo + ox;
                const float ga = gap[off];
                v = yp ? apply_bwd(s_chg[c], ga, yp[off]) : ga;
            }
            s_g[c * G_PLANE + r] = v;
        }
        for (int idx = t; idx < CIT * IN_TH * IN_TW; idx += 256) {
            const int c = idx / (IN_TH * IN_TW), r = idx - c * (IN_TH * IN_TW);
            const int iy = r / IN_TW, ix = r - iy * IN_TW, ci = ci0 + c;
            float v = 0.f;
            if (ci < Cin) {
                int gyy = reflect_idx(oy0 * STRIDE + iy - P, H), gxx = reflect_idx(ox0 * STRIDE + ix - P, W);
                gyy = min(max(gyy, 0), H - 1); gxx = min(max(gxx, 0), W - 1);      // overhang rows meet dy == 0
                v = apply_fwd(s_chx[c], xin[(long long)ci * HW + (long long)gyy * W + gxx], in.act, in.slope);
            }
            s_x[c * X_PLANE + r] = v;
        }
        __syncthreads();
        const float* gp = s_g + co_l * G_PLANE;
        const float* xp = s_x + ci_l * X_PLANE;
        for (int oy = 0; oy < TH; ++oy) {
#pragma unroll 4
            for (int ox = 0; ox < TW; ++ox) {
                const float gv = gp[oy * TW + ox];
                bsum += gv;
#pragma unroll
                for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx)
                        acc[ky * KS + kx] = __builtin_fmaf(gv, xp[(oy * STRIDE + ky) * IN_TW + ox * STRIDE + kx], acc[ky * KS + kx]);
            }
        }
    }

    // ---- epilogue: transpose through LDS so that the atomics are contiguous per wave ----
    __syncthreads();
    float* s_dw = lds;                  // [COT][ROW]
    float* s_dr = lds + COT * ROW;      // [COT][ROW]
#pragma unroll
    for (int q = 0; q < KK; ++q) s_dw[co_l * ROW + ci_l * KK + q] = acc[q];
    __syncthreads();
    const int cit = min(CIT, Cin - ci0);
    const int len = cit * KK;
    RngKey kw = key; kw.sample += (uint32_t)k; kw.stream = ((uint32_t)DOMAIN_EPS << 24) | (uint32_t)(2 * g.layer_id);
    if (sample_weights) {
        const int G = (len >> 2) + 2;
        for (int idx = t; idx < COT * G; idx += 256) {
            const int r = idx / G, gi = idx - r * G, co = co0 + r;
            if (co >= Cout) continue;
            const long long j0 = ((long long)co * Cin + ci0) * KK;
            const long long blk = (j0 >> 2) + gi, jb = blk << 2;
            if (jb >= j0 + len) continue;
            float z[4]; spec_normal4(kw, (uint32_t)blk, z);
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const long long j = jb + l;
                if (j >= j0 && j < j0 + len) {
                    const int rel = (int)(j - j0);
                    s_dr[r * ROW + rel] = s_dw[r * ROW + rel] * z[l] * sigmoid_f(rho[g.w_off + j]);
                }
            }
        }
        __syncthreads();
    }
    for (int idx = t; idx < COT * len; idx += 256) {
        const int r = idx / len, rel = idx - r * len, co = co0 + r;
        if (co >= Cout) continue;
        const long long j = ((long long)co * Cin + ci0) * KK + rel;
        atomicAdd(dmu + g.w_off + j, s_dw[r * ROW + rel]);
        if (sample_weights) atomicAdd(drho + g.w_off + j, s_dr[r * ROW + rel]);
    }
    if (do_bias && ci_l == 0) {
        const int co = co0 + co_l;
        if (co < Cout) {
            atomicAdd(dmu + g.b_off + co, bsum);
            if (sample_weights) {
                RngKey kb = kw; kb.stream += 1u;
                float z[4]; spec_normal4(kb, (uint32_t)(co >> 2), z);
                atomicAdd(drho + g.b_off + co, bsum * z[co & 3] * sigmoid_f(rho[g.b_off + co]));
            }
        }
    }
}

}  // namespace

int launch_conv_bwd_weight(const TView& in, const GView& gy, const ConvGeom& g, const float* rho, RngKey key, int sample_weights,
                           float* dmu, float* drho, int n_samples, hipStream_t st)
{
#define LAUNCH(KS_, S_)                                                                                                   \
    {                                                                                                                     \
        using Cfg = BwwCfg<KS_, S_>;                                                                                      \
        const int tiles_x = (g.Wo + Cfg::TW - 1) / Cfg::TW, tiles_y = (g.Ho + Cfg::TH - 1) / Cfg::TH;                     \
        const int n_tiles = tiles_x * tiles_y;                                                                            \
        const int co_tiles = (g.Cout + Cfg::COT - 1) / Cfg::COT, ci_tiles = (g.Cin + Cfg::CIT - 1) / Cfg::CIT;            \
        /* enough blocks to fill the chip, but long strips to amortise the atomic epilogue */                             \
        int strips = (2048 + co_tiles * ci_tiles * n_samples - 1) / (co_tiles * ci_tiles * n_samples);                    \
        strips = strips < 1 ? 1 : (strips > n_tiles ? n_tiles : strips);                                                  \
        const int tpb = (n_tiles + strips - 1) / strips;                                                                  \
        strips = (n_tiles + tpb - 1) / tpb;                                                                               \
        dim3 grid(strips, co_tiles * ci_tiles, n_samples);                                                                \
        hipLaunchKernelGGL((conv_bwd_weight_kernel<KS_, S_>), grid, dim3(256), 0, st, in, gy, g, rho, key, sample_weights, \
                           dmu, drho, tiles_x, n_tiles, tpb, ci_tiles);                                                   \
    }
    if (g.ks == 3 && g.stride == 1) LAUNCH(3, 1)
    else if (g.ks == 3 && g.stride == 2) LAUNCH(3, 2)
    else if (g.ks == 1 && g.stride == 1) LAUNCH(1, 1)
    else if (g.ks == 5 && g.stride == 1) LAUNCH(5, 1)
    else if (g.ks == 5 && g.stride == 2) LAUNCH(5, 2)
    else { set_error("conv_bwd_weight: unsupported ksize %d stride %d", g.ks, g.stride); return -1; }
#undef LAUNCH
    return (int)hipGetLastError();
}
