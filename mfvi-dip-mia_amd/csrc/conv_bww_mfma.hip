// K2b on the matrix cores — gradient of the fused reparameterised convolution wrt (mu, rho).
//
//   dW[co][ci][tap] = sum_pix dy[co][pix] * xpad[ci][S*pix + tap]      (autograd of reparam_layers.py:37)
// as 9 (or 1) independent GEMMs on v_mfma_f32_16x16x4_f32 whose K dimension is the PIXEL index:
//   D_tap[m = co][n = ci] += A[m][k] * B_tap[k][n],   k = 4 consecutive output pixels of one row,
//   A = dy (BN-backward formed on load), B_tap = reflection-padded LeakyReLU(BN(x)) shifted by the tap.
// A block owns one (16 cout) x (16 cin) weight tile and a strip of pixel tiles; its 4 waves split each tile's rows,
// accumulate 9 fragments each in registers and are summed through LDS at the end.  Epilogue (as the generic path):
//   d mu += dW,  d rho += dW * eps * sigmoid(rho)  with eps re-derived from the counter RNG, contiguous atomics.
// A tenth MFMA against a constant-one B fragment yields the bias gradient sum_pix dy for free.
// LDS planes are pitched == 2 (mod 32) floats so the 16 channels x 2 pixels of a half-wave read hit 32 banks.
#include "common.h"
#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int pitch2(int n) { return ((n + 29) / 32) * 32 + 2; }        // smallest p >= n with p % 32 == 2
static_assert(pitch2(256) == 258 && pitch2(340) == 354 && pitch2(2) == 2 && pitch2(3) == 34, "pitch2");

template <int KS, int STRIDE>
struct WCfg {
    static constexpr int TW = 32;
    static constexpr int TH = (STRIDE == 1) ? 8 : 4;
    static constexpr int KK = KS * KS;
    static constexpr int IN_TH = (TH - 1) * STRIDE + KS;
    static constexpr int IN_TW = (TW - 1) * STRIDE + KS;
    static constexpr int GPLANE = pitch2(TH * TW);
    static constexpr int XPLANE = pitch2(IN_TH * IN_TW);
    static constexpr int ROW = 16 * KK;
    static constexpr int STAGE = 16 * GPLANE + 16 * XPLANE;
    static constexpr int EPI = 2 * 16 * ROW + 16;
    static constexpr int LDS_FLOATS = STAGE > EPI ? STAGE : EPI;
    static constexpr int NG = (16 * TH * TW) / 256;                       // dy elements per thread per tile
    static constexpr int NX = (16 * IN_TH * IN_TW + 255) / 256;           // x elements per thread per tile
};

template <int KS, int STRIDE>
__global__ __launch_bounds__(256) void conv_bww_mfma_kernel(TView in, GView gy, ConvGeom g, const float* __restrict__ rho,
                                                            RngKey key, int sample_weights, float* __restrict__ dmu,
                                                            float* __restrict__ drho, int tiles_x, int n_tiles,
                                                            int tiles_per_block, int ci_tiles, int dbg)
{
    using Cfg = WCfg<KS, STRIDE>;
    constexpr int TW = Cfg::TW, TH = Cfg::TH, KK = Cfg::KK, P = KS / 2, IN_TH = Cfg::IN_TH, IN_TW = Cfg::IN_TW;
    constexpr int GPLANE = Cfg::GPLANE, XPLANE = Cfg::XPLANE, ROW = Cfg::ROW, NG = Cfg::NG, NX = Cfg::NX;

    __shared__ __align__(16) float lds[Cfg::LDS_FLOATS];
    __shared__ ChanFwd s_chx[16];
    __shared__ ChanBwd s_chg[16];
    float* s_g = lds;                    // [16][GPLANE]
    float* s_x = lds + 16 * GPLANE;      // [16][XPLANE]

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int k = blockIdx.z;
    const int co0 = (blockIdx.y / ci_tiles) * 16, ci0 = (blockIdx.y % ci_tiles) * 16;
    const int Cin = g.Cin, Cout = g.Cout, H = g.H, W = g.W, Ho = g.Ho, Wo = g.Wo;
    const bool do_bias = (ci0 == 0) && (g.b_off >= 0);
    const int cot = min(16, Cout - co0), cit = min(16, Cin - ci0);

    if (t < 16) s_chx[t] = chan_fwd(in, k, min(ci0 + t, Cin - 1));
    if (t >= 64 && t < 80) s_chg[t - 64] = chan_bwd(gy, k, min(co0 + t - 64, Cout - 1));

    f32x4 acc[KK], accb = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < KK; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* __restrict__ xin = in.data + (long long)k * in.sstride;
    const float* __restrict__ gap = gy.ga + (long long)k * gy.gstride;
    const float* __restrict__ yp = gy.y ? gy.y + (long long)k * gy.ystride : nullptr;
    const int HW = H * W, HWo = Ho * Wo;

    // Staging slots are fixed per thread: ONE output pixel of the dy tile for 16/CPP channels, NPOS positions of the
    // input tile for all 16 channels — so a tile costs a handful of index computations, not one per element.
    constexpr int PIX = TH * TW, CPP = 256 / PIX, NGC = 16 / CPP;          // CPP channels are staged per pass of 256 threads
    constexpr int NPOS = (IN_TH * IN_TW + 255) / 256;
    static_assert(256 % PIX == 0 && 16 % CPP == 0, "tile shape");
    const int gpix = t % PIX, gsub = t / PIX;
    float gr[NGC], yr[NGC], xr[NPOS][16];
    int goff = -1, xoff[NPOS];

    auto prefetch = [&](int tile) {
        const int ox0 = (tile % tiles_x) * TW, oy0 = (tile / tiles_x) * TH;
        const int yy = oy0 + gpix / TW, xx = ox0 + (gpix % TW);
        goff = (yy < Ho && xx < Wo) ? yy * Wo + xx : -1;
#pragma unroll
        for (int j = 0; j < NGC; ++j) {
            const int c = j * CPP + gsub;
            const bool ok = goff >= 0 && c < cot;
            const int off = (co0 + c) * HWo + goff;
            gr[j] = ok ? gap[off] : 0.f;
            yr[j] = (ok && yp) ? yp[off] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NPOS; ++q) {
            const int p = t + 256 * q;
            if (p < IN_TH * IN_TW) {
                const int iy = p / IN_TW, ix = p - iy * IN_TW;
                int gyy = reflect_idx(oy0 * STRIDE + iy - P, H), gxx = reflect_idx(ox0 * STRIDE + ix - P, W);
                gyy = min(max(gyy, 0), H - 1); gxx = min(max(gxx, 0), W - 1);         // overhang meets dy == 0
                xoff[q] = gyy * W + gxx;
            } else xoff[q] = -1;
#pragma unroll
            for (int c = 0; c < 16; ++c) xr[q][c] = (xoff[q] >= 0 && c < cit) ? xin[(ci0 + c) * HW + xoff[q]] : 0.f;
        }
    };

    const int tile_begin = blockIdx.x * tiles_per_block, tile_end = min(n_tiles, tile_begin + tiles_per_block);
    if (tile_begin < tile_end) prefetch(tile_begin);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NGC; ++j) {
            const int c = j * CPP + gsub;
            float v = 0.f;
            if (goff >= 0 && c < cot) v = yp ? apply_bwd(s_chg[c], gr[j], yr[j]) : gr[j];
            s_g[c * GPLANE + gpix] = v;
        }
#pragma unroll
        for (int q = 0; q < NPOS; ++q) {
            const int p = t + 256 * q;
            if (p < IN_TH * IN_TW) {
#pragma unroll
                for (int c = 0; c < 16; ++c) s_x[c * XPLANE + p] = c < cit ? apply_fwd(s_chx[c], xr[q][c], in.act, in.slope) : 0.f;
            }
        }
        __syncthreads();
        if (tile + 1 < tile_end) prefetch(tile + 1);
        // ---- MFMA over this wave's rows: k-steps of 4 consecutive pixels ----
        for (int row = wv; row < TH; row += 4) {
#pragma unroll 2
            for (int c4 = 0; c4 < TW; c4 += 4) {
                const float a = s_g[l15 * GPLANE + row * TW + c4 + l4];
                const float* xb = s_x + l15 * XPLANE + (row * STRIDE) * IN_TW + (c4 + l4) * STRIDE;
#pragma unroll
                for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx)
                        acc[ky * KS + kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xb[ky * IN_TW + kx], acc[ky * KS + kx], 0, 0, 0);
                if (do_bias) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(a, 1.0f, accb, 0, 0, 0);
            }
        }
    }

    // ---- sum the 4 waves through LDS.  D layout: column n (ci) = lane & 15, row m (co) = (lane >> 4) * 4 + reg ----
    __syncthreads();
    float* s_dw = lds;                  // [16][ROW]   dW tile, element (co, ci*KK + tap)
    float* s_dr = lds + 16 * ROW;       // [16][ROW]   dW * eps * sigmoid(rho)
    float* s_db = lds + 2 * 16 * ROW;   // [16]
    for (int i = t; i < 16 * ROW; i += 256) s_dw[i] = 0.f;
    if (t < 16) s_db[t] = 0.f;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < KK; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&s_dw[(l4 * 4 + r) * ROW + l15 * KK + q], acc[q][r]);
    if (do_bias && l15 == 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&s_db[l4 * 4 + r], accb[r]);
    __syncthreads();

    const int len = cit * KK;
    RngKey kw = key; kw.sample += (uint32_t)k; kw.stream = ((uint32_t)DOMAIN_EPS << 24) | (uint32_t)(2 * g.layer_id);
    if (sample_weights) {
        const int G = (len >> 2) + 2;
        for (int idx = t; idx < 16 * G; idx += 256) {
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
    for (int idx = t; idx < 16 * len && !(dbg & 64); idx += 256) {
        const int r = idx / len, rel = idx - r * len, co = co0 + r;
        if (co >= Cout) continue;
        const long long j = ((long long)co * Cin + ci0) * KK + rel;
        atomicAdd(dmu + g.w_off + j, s_dw[r * ROW + rel]);
        if (sample_weights) atomicAdd(drho + g.w_off + j, s_dr[r * ROW + rel]);
    }
    if (do_bias && t < cot) {
        const int co = co0 + t;
        const float bsum = s_db[t];
        atomicAdd(dmu + g.b_off + co, bsum);
        if (sample_weights) {
            RngKey kb = kw; kb.stream += 1u;
            float z[4]; spec_normal4(kb, (uint32_t)(co >> 2), z);
            atomicAdd(drho + g.b_off + co, bsum * z[co & 3] * sigmoid_f(rho[g.b_off + co]));
        }
    }
}

}  // namespace

int launch_conv_bwd_weight_mfma(const TView& in, const GView& gy, const ConvGeom& g, const float* rho, RngKey key, int sample_weights,
                                float* dmu, float* drho, int n_samples, hipStream_t st)
{
    static const int dbg = [] { const char* e = getenv("MFVI_DBG"); return e ? atoi(e) : 0; }();
#define LAUNCH(KS_, S_)                                                                                                        \
    {                                                                                                                          \
        using Cfg = WCfg<KS_, S_>;                                                                                             \
        const int tiles_x = (g.Wo + Cfg::TW - 1) / Cfg::TW, tiles_y = (g.Ho + Cfg::TH - 1) / Cfg::TH;                          \
        const int n_tiles = tiles_x * tiles_y;                                                                                 \
        const int co_tiles = (g.Cout + 15) / 16, ci_tiles = (g.Cin + 15) / 16;                                                 \
        const long long pairs = (long long)co_tiles * ci_tiles * n_samples;                                                    \
        int strips = (int)((1536 + pairs - 1) / pairs);                                                                        \
        strips = strips < 1 ? 1 : (strips > n_tiles ? n_tiles : strips);                                                       \
        const int tpb = (n_tiles + strips - 1) / strips;                                                                       \
        strips = (n_tiles + tpb - 1) / tpb;                                                                                    \
        dim3 grid(strips, co_tiles * ci_tiles, n_samples);                                                                     \
        hipLaunchKernelGGL((conv_bww_mfma_kernel<KS_, S_>), grid, dim3(256), 0, st, in, gy, g, rho, key, sample_weights, dmu,   \
                           drho, tiles_x, n_tiles, tpb, ci_tiles, dbg);                                                             \
        return (int)hipGetLastError();                                                                                         \
    }
    if (g.ks == 3 && g.stride == 1) LAUNCH(3, 1)
    if (g.ks == 3 && g.stride == 2) LAUNCH(3, 2)
    if (g.ks == 1 && g.stride == 1) LAUNCH(1, 1)
#undef LAUNCH
    return -2;
}
