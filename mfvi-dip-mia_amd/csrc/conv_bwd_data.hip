// K2a — fused reparameterised convolution, backward wrt the input (generic fp32 VALU path).
//
// Autograd of ReflectionPad2d + F.conv2d (BayTorch/modules/reparam_layers.py:37, models/common.py:118-123):
// this kernel produces the gradient wrt the PADDED input, d_xp[Cin][H+2p][W+2p] (a zero-padded full
// correlation of dy with the re-sampled weights; eps is re-derived from the counter RNG, w is never stored);
// the reflection fold, LeakyReLU' and the BN-backward sums are applied by finalize_dx (finalize.hip).
// dy itself is formed on load from (ga, y, BN sums) — GView — so BN-backward never materialises.
#include "common.h"

namespace {

template <int KS, int STRIDE>
struct BwdCfg {
    static constexpr int TW = 32, TH = 8;        // padded-input pixels per block (one per thread)
    static constexpr int CIT = 16;               // input channels per block
    static constexpr int COC = 8;                // output channels per LDS stage
    static constexpr int GT_H = (STRIDE == 1) ? TH + KS - 1 : TH / 2 + (KS + 1) / 2;
    static constexpr int GT_W = (STRIDE == 1) ? TW + KS - 1 : TW / 2 + (KS + 1) / 2;
    static constexpr int GT_WP = GT_W | 1;
};

__device__ __forceinline__ int floor_div2(int a) { return a >> 1; }   // arithmetic shift = floor for negatives

template <int KS, int STRIDE>
__global__ __launch_bounds__(256) void conv_bwd_data_kernel(GView gy, ConvGeom g, const float* __restrict__ mu,
                                                            const float* __restrict__ rho, RngKey key,
                                                            int sample_weights, float* __restrict__ dxp,
                                                            long long dxp_sstride, int tiles_x)
{
    key = key_now(key);
    using Cfg = BwdCfg<KS, STRIDE>;
    constexpr int TW = Cfg::TW, TH = Cfg::TH, CIT = Cfg::CIT, COC = Cfg::COC, KK = KS * KS, P = KS / 2;
    constexpr int GT_H = Cfg::GT_H, GT_W = Cfg::GT_W, GT_WP = Cfg::GT_WP;

    __shared__ float s_g[COC][GT_H][GT_WP];
    __shared__ __align__(16) float s_w[COC][KK][CIT];
    __shared__ ChanBwd s_ch[MFVI_MAX_C];

    const int t = threadIdx.x, lx = t & 31, ly = t >> 5;
    const int k = blockIdx.z;
    const int ci0 = blockIdx.y * CIT;
    const int pc0 = (blockIdx.x % tiles_x) * TW, pr0 = (blockIdx.x / tiles_x) * TH;
    const int Cin = g.Cin, Cout = g.Cout, Ho = g.Ho, Wo = g.Wo;
    const int Hp = g.H + 2 * P, Wp = g.W + 2 * P;
    const int cit = min(CIT, Cin - ci0);

    RngKey kw = key; kw.sample += (uint32_t)k; kw.stream = ((uint32_t)DOMAIN_EPS << 24) | (uint32_t)(2 * g.layer_id);

    for (int c = t; c < Cout; c += 256) s_ch[c] = chan_bwd(gy, k, c);

    // origin of the staged dy tile in output coordinates
    const int base_r = (STRIDE == 1) ? pr0 - (KS - 1) : floor_div2(pr0 - (KS - 1));
    const int base_c = (STRIDE == 1) ? pc0 - (KS - 1) : floor_div2(pc0 - (KS - 1));

    float acc[CIT];
#pragma unroll
    for (int q = 0; q < CIT; ++q) acc[q] = 0.f;

    const float* __restrict__ gap = gy.ga + (long long)k * gy.gstride;
    const float* __restrict__ yp = gy.y ? gy.y + (long long)k * gy.ystride : nullptr;
    const long long HWo = (long long)Ho * Wo;

    for (int co0 = 0; co0 < Cout; co0 += COC) {
        __syncthreads();
        // ---- stage dy (BN-backward formed on load), zero outside the output ----
        for (int idx = t; idx < COC * GT_H * GT_W; idx += 256) {
            const int c = idx / (GT_H * GT_W), r = idx - c * (GT_H * GT_W);
            const int iy = r / GT_W, ix = r - iy * GT_W;
            const int orow = base_r + iy, ocol = base_c + ix, co = co0 + c;
            float v = 0.f;
            if (co < Cout && orow >= 0 && orow < Ho && ocol >= 0 && ocol < Wo) {
                const long long off = (long long)co * HWo + (long long)orow * Wo + ocol;
                const float ga = gap[off];
                v = yp ? apply_bwd(s_ch[co], ga, yp[off]) : ga;
            }
            s_g[c][iy][ix] = v;
        }
        // ---- re-sample the weight slab w[co0..+COC)[ci0..+cit)[KK] ----
        {
            const int len = cit * KK;
            const int G = (len >> 2) + 2;
            for (int idx = t; idx < COC * G; idx += 256) {
                const int co_c = idx / G, gi = idx - co_c * G;
                const int co = co0 + co_c;
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
                                const int rel = (int)(j - j0), ci_l = rel / KK, tap = rel - ci_l * KK;
                                float w = mu[g.w_off + j];
                                if (sample_weights) w += softplus_f(rho[g.w_off + j]) * z[l];
                                s_w[co_c][tap][ci_l] = w;
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        const int rel = gi * 4 + l;
                        if (rel < len) { const int ci_l = rel / KK, tap = rel - ci_l * KK; s_w[co_c][tap][ci_l] = 0.f; }
                    }
                }
            }
            // lanes ci_l >= cit are never stored, but keep them finite
            if (cit < CIT)
                for (int idx = t; idx < COC * KK * (CIT - cit); idx += 256) {
                    const int ci_l = cit + idx % (CIT - cit), r = idx / (CIT - cit);
                    s_w[r / KK][r % KK][ci_l] = 0.f;
                }
        }
        __syncthreads();
        // ---- accumulate: d_xp[ci][pr][pc] += w[co][ci][ky][kx] * dy[co][(pr-ky)/S][(pc-kx)/S] ----
        const int pr = pr0 + ly, pc = pc0 + lx;
        for (int c = 0; c < COC; ++c) {
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    float gv;
                    if (STRIDE == 1) {
                        gv = s_g[c][ly + (KS - 1) - ky][lx + (KS - 1) - kx];
                    } else {
                        const int ry = pr - ky, rx = pc - kx;
                        const bool ok = ((ry | rx) & 1) == 0;
                        const int iy = floor_div2(ry) - base_r, ix = floor_div2(rx) - base_c;
                        gv = ok ? s_g[c][iy][ix] : 0.f;
                    }
                    const float4* wp = reinterpret_cast<const float4*>(&s_w[c][ky * KS + kx][0]);
                    const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
                    const float wv[CIT] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w,
                                           w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
#pragma unroll
                    for (int q = 0; q < CIT; ++q) acc[q] = __builtin_fmaf(gv, wv[q], acc[q]);
                }
        }
    }

    const int pr = pr0 + ly, pc = pc0 + lx;
    if (pr < Hp && pc < Wp) {
        float* __restrict__ o = dxp + (long long)k * dxp_sstride + (long long)ci0 * Hp * Wp + (long long)pr * Wp + pc;
#pragma unroll
        for (int q = 0; q < CIT; ++q)
            if (q < cit) o[(long long)q * Hp * Wp] = acc[q];
    }
}

}  // namespace

int launch_conv_bwd_data(const GView& gy, const ConvGeom& g, const float* mu, const float* rho, RngKey key, int sample_weights,
                         float* dxp, long long dxp_sstride, int n_samples, hipStream_t st)
{
    if (g.Cout > MFVI_MAX_C) { set_error("conv_bwd_data: Cout %d > %d", g.Cout, MFVI_MAX_C); return -1; }
    const int P = g.ks / 2, Hp = g.H + 2 * P, Wp = g.W + 2 * P;
#define LAUNCH(KS_, S_)                                                                                                  \
    {                                                                                                                    \
        using Cfg = BwdCfg<KS_, S_>;                                                                                     \
        const int tiles_x = (Wp + Cfg::TW - 1) / Cfg::TW, tiles_y = (Hp + Cfg::TH - 1) / Cfg::TH;                        \
        dim3 grid(tiles_x * tiles_y, (g.Cin + Cfg::CIT - 1) / Cfg::CIT, n_samples);                                      \
        hipLaunchKernelGGL((conv_bwd_data_kernel<KS_, S_>), grid, dim3(256), 0, st, gy, g, mu, rho, key, sample_weights, \
                           dxp, dxp_sstride, tiles_x);                                                                   \
    }
    if (g.ks == 3 && g.stride == 1) LAUNCH(3, 1)
    else if (g.ks == 3 && g.stride == 2) LAUNCH(3, 2)
    else if (g.ks == 1 && g.stride == 1) LAUNCH(1, 1)
    else if (g.ks == 5 && g.stride == 1) LAUNCH(5, 1)
    else if (g.ks == 5 && g.stride == 2) LAUNCH(5, 2)
    else { set_error("conv_bwd_data: unsupported ksize %d stride %d", g.ks, g.stride); return -1; }
#undef LAUNCH
    return (int)hipGetLastError();
}
