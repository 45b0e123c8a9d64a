// K1/K2a on the matrix cores — fused reparameterised convolution as an implicit GEMM on
// v_mfma_f32_16x16x4_f32 (fp32 in / fp32 accumulate: bit-for-bit an fp32 fma chain, so parity with the
// fp32 reference is unchanged; MI355X_MICROARCH.md "Matrix cores").
//
//   D[m][n] += A[m][k] * B[k][n]      m = output channel (16 per fragment), n = 16 consecutive pixels of one row,
//                                      k = 4 consecutive reduction channels of one filter tap
//   MODE 0 (forward, reparam_layers.py:26-37):   m = cout, k = (cin, tap), B = reflection-padded view(x)
//   MODE 1 (backward-data, stride 1):            m = cin,  k = (cout, flipped tap), B = zero-padded dy (formed on
//                                                 load from ga / y / BN sums), output = gradient wrt the PADDED input
//
// Per 256-thread block: a 32 x TH pixel tile x (16*MF) output channels.  Per reduction chunk of CC channels the block
// stages the activation tile (deferred BN + LeakyReLU, reflection in the index math) and SAMPLES the weight slab
// w = mu + softplus(rho)*eps into LDS (eps from Philox, never stored); each of the 4 waves then owns TH/4 rows =
// NF = TH/2 pixel fragments and issues MF*NF MFMAs per (tap, 4-channel) step from conflict-free ds_read_b32:
//   activations  s_x[k][row][col], plane pitch == 16 (mod 32) floats  -> lanes 0-15 / 16-31 of a half-wave hit disjoint banks
//   weights      s_w[tap][k][m],   row pitch   == 16 (mod 32) floats
// Epilogue (MODE 0): + sampled bias, raw store, per-channel sum / sum^2 of the BatchNorm that follows (fp64 atomics).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int pitch16(int n) { return ((n + 15) / 32) * 32 + 16; }      // smallest p >= n with p % 32 == 16

template <int KS, int STRIDE, int MF, int TH>
struct MCfg {
    static constexpr int TW = 32;
    static constexpr int CT = 16 * MF;
    static constexpr int CC = 8;                                   // reduction channels per stage
    static constexpr int NF = TH / 2;                              // pixel fragments per wave (TH/4 rows x 2 halves)
    static constexpr int KK = KS * KS;
    static constexpr int IN_TH = (TH - 1) * STRIDE + KS;
    static constexpr int IN_TW = (TW - 1) * STRIDE + KS;
    static constexpr int PITCH = IN_TW;
    static constexpr int PLANE = pitch16(IN_TH * PITCH);           // == 16 (mod 32)
    static constexpr int CTP = pitch16(CT);                        // == 16 (mod 32)
    static constexpr int X_FLOATS = CC * PLANE;
    static constexpr int W_FLOATS = KK * CC * CTP;
};

static_assert(pitch16(16) == 16 && pitch16(32) == 48 && pitch16(64) == 80 && pitch16(340) % 32 == 16, "pitch16");

template <int KS, int STRIDE, int MF, int TH, int MODE>
__global__ __launch_bounds__(256) void conv_mfma_kernel(TView xin, GView gin, ConvGeom g, const float* __restrict__ mu,
                                                        const float* __restrict__ rho, RngKey key, int sample_weights,
                                                        OutDesc out, float* __restrict__ dxp, long long dxp_sstride,
                                                        int tiles_x)
{
    using Cfg = MCfg<KS, STRIDE, MF, TH>;
    constexpr int TW = Cfg::TW, CT = Cfg::CT, CC = Cfg::CC, NF = Cfg::NF, KK = Cfg::KK, P = KS / 2;
    constexpr int IN_TH = Cfg::IN_TH, IN_TW = Cfg::IN_TW, PITCH = Cfg::PITCH, PLANE = Cfg::PLANE, CTP = Cfg::CTP;
    static_assert(MODE == 0 || STRIDE == 1, "backward-data on the MFMA path is stride 1 only");

    __shared__ float s_x[Cfg::X_FLOATS];
    __shared__ float s_w[Cfg::W_FLOATS];
    __shared__ float s_chf[MFVI_MAX_C * 4];          // ChanFwd (MODE 0) / ChanBwd c1..c3,mean,rstd packed (MODE 1 uses s_chb)
    __shared__ ChanBwd s_chb[MODE == 1 ? MFVI_MAX_C : 1];
    __shared__ float s_bias[CT];
    __shared__ double s_red[4][CT][2];

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int k = blockIdx.z;
    const int m0 = blockIdx.y * CT;                                  // first output channel of the block
    const int px0 = (blockIdx.x % tiles_x) * TW, py0 = (blockIdx.x / tiles_x) * TH;

    // MODE 0: reduce over cin, outputs = cout.   MODE 1: reduce over cout, outputs = cin.
    const int RED = MODE == 0 ? g.Cin : g.Cout;                      // reduction channels
    const int MOUT = MODE == 0 ? g.Cout : g.Cin;                     // output channels
    const int mt = min(CT, MOUT - m0);

    RngKey kw = key; kw.sample += (uint32_t)k; kw.stream = ((uint32_t)DOMAIN_EPS << 24) | (uint32_t)(2 * g.layer_id);

    ChanFwd* s_ch = reinterpret_cast<ChanFwd*>(s_chf);
    if (MODE == 0) {
        for (int c = t; c < g.Cin; c += 256) s_ch[c] = chan_fwd(xin, k, c);
        if (t < CT) {
            const int co = m0 + t; float b = 0.f;
            if (co < g.Cout && g.b_off >= 0) {
                b = mu[g.b_off + co];
                if (sample_weights) {
                    RngKey kb = kw; kb.stream += 1u;
                    float z[4]; spec_normal4(kb, (uint32_t)(co >> 2), z);
                    b += softplus_f(rho[g.b_off + co]) * z[co & 3];
                }
            }
            s_bias[t] = b;
        }
    } else {
        for (int c = t; c < g.Cout; c += 256) s_chb[c] = chan_bwd(gin, k, c);
    }

    f32x4 acc[MF][NF];
#pragma unroll
    for (int a = 0; a < MF; ++a)
#pragma unroll
        for (int b = 0; b < NF; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // source tile geometry
    const int H = g.H, W = g.W;
    const float* __restrict__ xsrc = MODE == 0 ? xin.data + (long long)k * xin.sstride : gin.ga + (long long)k * gin.gstride;
    const float* __restrict__ ysrc = (MODE == 1 && gin.y) ? gin.y + (long long)k * gin.ystride : nullptr;
    const int SH = MODE == 0 ? H : g.Ho, SW = MODE == 0 ? W : g.Wo;          // source plane size
    const int SHW = SH * SW;
    const int sy0 = MODE == 0 ? py0 * STRIDE - P : py0 - (KS - 1);          // tile origin in source coordinates
    const int sx0 = MODE == 0 ? px0 * STRIDE - P : px0 - (KS - 1);

    // Each thread owns NPOS fixed positions of the staged tile; their global / LDS offsets never change.
    constexpr int NPOS = (IN_TH * IN_TW + 255) / 256;
    int goff[NPOS], loff[NPOS];
#pragma unroll
    for (int j = 0; j < NPOS; ++j) {
        const int p = t + 256 * j;
        const int iy = p / IN_TW, ix = p - iy * IN_TW;
        loff[j] = p < IN_TH * IN_TW ? iy * PITCH + ix : -1;
        int gy = sy0 + iy, gx = sx0 + ix;
        if (MODE == 0) {
            gy = reflect_idx(gy, H); gx = reflect_idx(gx, W);
            gy = min(max(gy, 0), H - 1); gx = min(max(gx, 0), W - 1);       // tile overhang: masked at the store
            goff[j] = gy * W + gx;
        } else {
            goff[j] = (gy >= 0 && gy < SH && gx >= 0 && gx < SW) ? gy * SW + gx : -1;   // zero padding
        }
        if (loff[j] < 0) goff[j] = -1;
    }
    float xr[NPOS][CC], yr[MODE == 1 ? NPOS : 1][MODE == 1 ? CC : 1];

    auto prefetch = [&](int c0) {            // global loads of chunk c0 into registers; nothing waits on them here
        const int cc = min(CC, RED - c0);
#pragma unroll
        for (int j = 0; j < NPOS; ++j)
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const bool ok = goff[j] >= 0 && c < cc;
                const long long off = (long long)(c0 + c) * SHW + goff[j];
                xr[j][c] = ok ? xsrc[off] : 0.f;
                if (MODE == 1) yr[j][c] = (ok && ysrc) ? ysrc[off] : 0.f;
            }
    };

    // this wave's fragments: rows wv*(TH/4) .. +TH/4-1, two 16-pixel halves each
    int boff[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int row = wv * (TH / 4) + (f >> 1), col = (f & 1) * 16 + l15;
        boff[f] = l4 * PLANE + row * STRIDE * PITCH + col * STRIDE;
    }
    const int aoff = l4 * CTP + l15;
    const bool w_aligned = (g.w_off & 3) == 0;

    prefetch(0);
    for (int c0 = 0; c0 < RED; c0 += CC) {
        __syncthreads();                       // every wave is done reading the previous chunk
        const int cc = min(CC, RED - c0), cc4 = (cc + 3) & ~3;
        // ---- registers -> LDS: deferred BN/LeakyReLU (MODE 0) or BN-backward (MODE 1) ----
#pragma unroll
        for (int j = 0; j < NPOS; ++j) {
            if (loff[j] < 0) continue;
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                if (c >= cc4) break;
                float v = 0.f;
                if (c < cc) {
                    if (MODE == 0) v = apply_fwd(s_ch[c0 + c], xr[j][c], xin.act, xin.slope);
                    else v = goff[j] < 0 ? 0.f : (ysrc ? apply_bwd(s_chb[c0 + c], xr[j][c], yr[j][c]) : xr[j][c]);
                }
                s_x[c * PLANE + loff[j]] = v;
            }
        }
        // ---- sample the weight slab into s_w[tap][kk][m] ----
        {
            // MODE 0: rows = output channel m, row range ((m0+m)*Cin + c0)*KK + [0, cc*KK),   element -> (kk, tap)
            // MODE 1: rows = reduction channel kk, range ((c0+kk)*Cin + m0)*KK + [0, mt*KK),  element -> (m, flipped tap)
            const int rows = MODE == 0 ? CT : cc4;
            const int valid_rows = MODE == 0 ? mt : cc;
            const int len = (MODE == 0 ? cc : mt) * KK, G = (len >> 2) + 2;
            for (int idx = t; idx < rows * G; idx += 256) {
                const int row = idx / G, gi = idx - row * G;
                if (row < valid_rows) {
                    const long long j0 = MODE == 0 ? ((long long)(m0 + row) * g.Cin + c0) * KK : ((long long)(c0 + row) * g.Cin + m0) * KK;
                    const long long blk = (j0 >> 2) + gi, jb = blk << 2;
                    if (jb < j0 + len) {
                        float mv[4], rv[4];
                        if (w_aligned && jb >= j0 && jb + 4 <= j0 + len) {
                            const float4 a = *reinterpret_cast<const float4*>(mu + g.w_off + jb);
                            mv[0] = a.x; mv[1] = a.y; mv[2] = a.z; mv[3] = a.w;
                            if (sample_weights) { const float4 b = *reinterpret_cast<const float4*>(rho + g.w_off + jb); rv[0] = b.x; rv[1] = b.y; rv[2] = b.z; rv[3] = b.w; }
                        } else {
#pragma unroll
                            for (int l = 0; l < 4; ++l) {
                                const long long j = jb + l; const bool in = j >= j0 && j < j0 + len;
                                mv[l] = in ? mu[g.w_off + j] : 0.f; rv[l] = (in && sample_weights) ? rho[g.w_off + j] : 0.f;
                            }
                        }
                        float z[4] = {0.f, 0.f, 0.f, 0.f};
                        if (sample_weights) spec_normal4(kw, (uint32_t)blk, z);
#pragma unroll
                        for (int l = 0; l < 4; ++l) {
                            const long long j = jb + l;
                            if (j >= j0 && j < j0 + len) {
                                const int rel = (int)(j - j0), q = rel / KK, tap = rel - q * KK;
                                const float w = sample_weights ? __builtin_fmaf(softplus_fast(rv[l]), z[l], mv[l]) : mv[l];
                                if (MODE == 0) s_w[(tap * CC + q) * CTP + row] = w;
                                else s_w[((KK - 1 - tap) * CC + row) * CTP + q] = w;
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        const int rel = gi * 4 + l;
                        if (rel < len) {
                            const int q = rel / KK, tap = rel - q * KK;
                            if (MODE == 0) s_w[(tap * CC + q) * CTP + row] = 0.f; else s_w[(tap * CC + row) * CTP + q] = 0.f;
                        }
                    }
                }
            }
            if (MODE == 0 && cc4 > cc)       // pad the last 4-channel step with zero weights
                for (int idx = t; idx < (cc4 - cc) * KK * CT; idx += 256) {
                    const int m = idx % CT, r = idx / CT, kk = cc + r % (cc4 - cc), tap = r / (cc4 - cc);
                    s_w[(tap * CC + kk) * CTP + m] = 0.f;
                }
            if (MODE == 1 && mt < CT)        // output channels beyond the tensor: keep the (unstored) accumulators finite
                for (int idx = t; idx < KK * cc4 * (CT - mt); idx += 256) {
                    const int m = mt + idx % (CT - mt), r = idx / (CT - mt), kk = r % cc4, tap = r / cc4;
                    s_w[(tap * CC + kk) * CTP + m] = 0.f;
                }
        }
        __syncthreads();
        if (c0 + CC < RED) prefetch(c0 + CC);          // next chunk's loads fly while the matrix cores work
        // ---- MFMA ----
        const int steps = cc4 >> 2;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int tap = ky * KS + kx;
                for (int s = 0; s < steps; ++s) {
                    float a[MF], b[NF];
#pragma unroll
                    for (int i = 0; i < MF; ++i) a[i] = s_w[(tap * CC + s * 4) * CTP + aoff + i * 16];
#pragma unroll
                    for (int f = 0; f < NF; ++f) b[f] = s_x[s * 4 * PLANE + boff[f] + ky * PITCH + kx];
#pragma unroll
                    for (int i = 0; i < MF; ++i)
#pragma unroll
                        for (int f = 0; f < NF; ++f)
                            acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[f], acc[i][f], 0, 0, 0);
                }
            }
    }

    // ---- epilogue ----  D layout: column (pixel) = lane & 15, row (channel) = (lane >> 4) * 4 + reg
    if (MODE == 0) {
        float* __restrict__ yout = out.data + (long long)k * out.sstride;
        const long long HWo = (long long)g.Ho * g.Wo;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            double sum[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int oy = py0 + wv * (TH / 4) + (f >> 1), ox = px0 + (f & 1) * 16 + l15;
                if (oy < g.Ho && ox < g.Wo) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ml = i * 16 + l4 * 4 + r;
                        if (ml < mt) {
                            const float v = acc[i][f][r] + s_bias[ml];
                            yout[(long long)(m0 + ml) * HWo + (long long)oy * g.Wo + ox] = v;
                            sum[r] += (double)v; sq[r] += (double)v * (double)v;
                        }
                    }
                }
            }
            if (out.stats != nullptr) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double a = sum[r], b = sq[r];
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                    if (l15 == 0) { s_red[wv][i * 16 + l4 * 4 + r][0] = a; s_red[wv][i * 16 + l4 * 4 + r][1] = b; }
                }
            }
        }
        if (out.stats != nullptr) {
            __syncthreads();
            if (t < CT * 2) {
                const int q = t >> 1, which = t & 1;
                if (q < mt)
                    atomicAdd(out.stats + ((long long)k * g.Cout + m0 + q) * 2 + which,
                              s_red[0][q][which] + s_red[1][q][which] + s_red[2][q][which] + s_red[3][q][which]);
            }
        }
    } else {
        const int Hp = g.H + 2 * P, Wp = g.W + 2 * P;
        float* __restrict__ o = dxp + (long long)k * dxp_sstride;
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int pr = py0 + wv * (TH / 4) + (f >> 1), pc = px0 + (f & 1) * 16 + l15;
                if (pr < Hp && pc < Wp) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ml = i * 16 + l4 * 4 + r;
                        if (ml < mt) o[(long long)(m0 + ml) * Hp * Wp + (long long)pr * Wp + pc] = acc[i][f][r];
                    }
                }
            }
    }
}

template <int KS, int STRIDE, int MODE>
int launch_variant(const TView& xin, const GView& gin, const ConvGeom& g, const float* mu, const float* rho, RngKey key,
                   int sample_weights, OutDesc out, float* dxp, long long dxp_sstride, int n_samples, hipStream_t st)
{
    const int P = g.ks / 2;
    const int OH = MODE == 0 ? g.Ho : g.H + 2 * P, OW = MODE == 0 ? g.Wo : g.W + 2 * P;    // output pixel domain
    const int MOUT = MODE == 0 ? g.Cout : g.Cin;
#define GO(MF_, TH_)                                                                                                       \
    {                                                                                                                      \
        const int tiles_x = (OW + 31) / 32, tiles_y = (OH + TH_ - 1) / TH_;                                                \
        dim3 grid(tiles_x * tiles_y, (MOUT + 16 * MF_ - 1) / (16 * MF_), n_samples);                                       \
        hipLaunchKernelGGL((conv_mfma_kernel<KS, STRIDE, MF_, TH_, MODE>), grid, dim3(256), 0, st, xin, gin, g, mu, rho, key, \
                           sample_weights, out, dxp, dxp_sstride, tiles_x);                                                \
        return (int)hipGetLastError();                                                                                     \
    }
    // tall tiles amortise the in-kernel weight sampling; short ones keep small images from wasting lanes
    // Pick the largest tile that still gives the chip >= ~3 blocks per CU: big tiles amortise the in-kernel weight
    // sampling (each sampled weight is reused by every pixel of the tile), small ones keep 256 CUs busy.
    const auto blocks = [&](int mf, int th) { return (long long)((OW + 31) / 32) * ((OH + th - 1) / th) * ((MOUT + 16 * mf - 1) / (16 * mf)) * n_samples; };
    const int mf_max = MOUT <= 16 ? 1 : (MOUT <= 32 || MOUT % 64 != 0) ? 2 : 4;
    const long long want = 768;
    if constexpr (STRIDE == 1) {
        if (OH >= 16) {
            if (mf_max == 4 && blocks(4, 16) >= want) GO(4, 16)
            if (mf_max >= 2 && blocks(2, 16) >= want) GO(2, 16)
            if (blocks(1, 16) >= want) GO(1, 16)
        }
    }
    if (mf_max == 4 && blocks(4, 8) >= want) GO(4, 8)
    if (mf_max >= 2 && blocks(2, 8) >= want) GO(2, 8)
    GO(1, 8)
#undef GO
}

}  // namespace

// Returns -2 when the shape is not served by the MFMA path (caller falls back to the generic kernels).
int launch_conv_fwd_mfma(const TView& in, const ConvGeom& g, const float* mu, const float* rho, RngKey key, int sample_weights,
                         OutDesc out, int n_samples, hipStream_t st)
{
    if (g.Cin > MFVI_MAX_C) return -2;
    GView none{};
    if (g.ks == 3 && g.stride == 1) return launch_variant<3, 1, 0>(in, none, g, mu, rho, key, sample_weights, out, nullptr, 0, n_samples, st);
    if (g.ks == 3 && g.stride == 2) return launch_variant<3, 2, 0>(in, none, g, mu, rho, key, sample_weights, out, nullptr, 0, n_samples, st);
    if (g.ks == 1 && g.stride == 1) return launch_variant<1, 1, 0>(in, none, g, mu, rho, key, sample_weights, out, nullptr, 0, n_samples, st);
    return -2;
}

int launch_conv_bwd_data_mfma(const GView& gy, const ConvGeom& g, const float* mu, const float* rho, RngKey key, int sample_weights,
                              float* dxp, long long dxp_sstride, int n_samples, hipStream_t st)
{
    if (g.Cout > MFVI_MAX_C || g.stride != 1) return -2;
    TView none{}; OutDesc od{};
    if (g.ks == 3) return launch_variant<3, 1, 1>(none, gy, g, mu, rho, key, sample_weights, od, dxp, dxp_sstride, n_samples, st);
    if (g.ks == 1) return launch_variant<1, 1, 1>(none, gy, g, mu, rho, key, sample_weights, od, dxp, dxp_sstride, n_samples, st);
    return -2;
}
