// K2b on the matrix cores — gradient of the fused reparameterised convolution wrt (mu, rho).
//
//   dW[co][ci][tap] = sum_pix dy[co][pix] * xpad[ci][S*pix + tap]      (autograd of reparam_layers.py:37)
// as 9 (or 1) independent GEMMs on v_mfma_f32_16x16x4_f32 whose K dimension is the PIXEL index:
//   D_tap[m = co][n = ci] += A[m][k] * B_tap[k][n],   k = 4 consecutive output pixels of one row,
//   A = dy (BN-backward formed on load), B_tap = reflection-padded LeakyReLU(BN(x)) shifted by the tap.
// A block owns one (16 cout) x (16 cin) weight tile and a strip of pixel tiles; its 4 waves split each tile's rows,
// accumulate 9 fragments each in registers and are summed through LDS at the end.  The block's partial dW goes to a
// per-(strip, sample) slab with plain stores; grad_finalize (losses.hip) reduces the slabs and forms
//   d mu += dW,  d rho += dW * eps * sigmoid(rho)  with eps re-derived from the counter RNG (no atomics, deterministic).
// A tenth MFMA against a constant-one B fragment yields the bias gradient sum_pix dy for free.
// LDS planes are pitched == 2 (mod 32) floats so the 16 channels x 2 pixels of a half-wave read hit 32 banks.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2a __attribute__((ext_vector_type(2)));
// LDS planes are pitched == 2 (mod 32) floats: rows are 8-byte aligned, so a staged float4 goes out as two ds_write_b64
__device__ __forceinline__ void lds_store4(float* p, float a, float b, float c, float d)
{
    f2a lo, hi; lo.x = a; lo.y = b; hi.x = c; hi.y = d;
    *reinterpret_cast<f2a*>(p) = lo; *reinterpret_cast<f2a*>(p + 2) = hi;
}

constexpr int pitch2(int n) { return ((n + 29) / 32) * 32 + 2; }        // smallest p >= n with p % 32 == 2
static_assert(pitch2(256) == 258 && pitch2(340) == 354 && pitch2(2) == 2 && pitch2(3) == 34, "pitch2");

template <int KS, int STRIDE, int NB, int NT>
struct WCfg {
    static constexpr int TW = 32;
    static constexpr int TH = (STRIDE == 1) ? 8 : 4;
    static constexpr int KK = KS * KS;
    static constexpr int CIB = 16 * NB;                                   // input channels per block
    static constexpr int IN_TH = (TH - 1) * STRIDE + KS;
    static constexpr int IN_TW = (TW - 1) * STRIDE + KS;
    // The input window is stored with rows starting at an ALIGNED column (4 left of the tile when there is a halo), so the
    // specialised producers can stage it with float4 loads / ds_write_b128; the first column a tap needs sits at XOFF.
    static constexpr int HALO4 = KS > 1 ? 4 : 0;
    static constexpr int XOFF = HALO4 ? HALO4 - KS / 2 : 0;
    static constexpr int WV = (XOFF + IN_TW + 3) / 4 * 4;                  // 40 (3x3), 68 (3x3 stride 2), 32 (1x1)
    static constexpr int GPLANE = pitch2(TH * TW);
    static constexpr int XPLANE = pitch2(IN_TH * WV);
    static constexpr int ROW = CIB * KK;
    static constexpr int STAGE = 16 * GPLANE + CIB * XPLANE;
    static constexpr int ROWP = ROW + 2;                                   // 4 * ROWP == 8 (mod 32): the 4 row groups of a wave spread over banks
    static constexpr int EPI = 4 * 16 * ROWP + 64;                        // 4 wave regions of the cross-wave reduction + bias partials
    static constexpr int LDS_FLOATS = STAGE > EPI ? STAGE : EPI;
};

// NB = 16-channel input tiles per block (the dy tile is staged once for all of them), NT = threads per block.
// SPEC (NT == 512): waves 4-7 are producers (global loads -> deferred BN/LeakyReLU or BN-backward -> LDS, next tile
// prefetched into registers while the matrix cores run), waves 0-3 consumers that only read LDS and issue MFMAs.
template <int KS, int STRIDE, int NB, int NT, bool SPEC>
__global__ __launch_bounds__(NT) void conv_bww_mfma_kernel(TView in, GView gy, ConvGeom g, float* __restrict__ part,
                                                           long long part_stride, int tiles_x, int n_tiles,
                                                           int tiles_per_block, int ci_groups, int nx, int ny, int nz)
{
    using Cfg = WCfg<KS, STRIDE, NB, NT>;
    constexpr int TW = Cfg::TW, TH = Cfg::TH, KK = Cfg::KK, P = KS / 2, IN_TH = Cfg::IN_TH, IN_TW = Cfg::IN_TW;
    constexpr int GPLANE = Cfg::GPLANE, XPLANE = Cfg::XPLANE, ROW = Cfg::ROW, CIB = Cfg::CIB, WV = Cfg::WV, XOFF = Cfg::XOFF;
    constexpr int NS = SPEC ? NT / 2 : NT;            // staging threads
    constexpr int NW = (SPEC ? NT / 2 : NT) / 64;     // MFMA waves
    static_assert(!SPEC || NT == 512, "specialised variant is built for 8 waves");

    extern __shared__ __align__(16) float lds[];       // Cfg::LDS_FLOATS
    __shared__ ChanFwd s_chx[CIB];
    __shared__ ChanBwd s_chg[16];
    float* s_g = lds;                    // [16][GPLANE]
    float* s_x = lds + 16 * GPLANE;      // [CIB][XPLANE]

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, l15 = lane & 15, l4 = lane >> 4;
    const bool producer = SPEC && t >= NS;
    const int ts = SPEC ? (t & (NS - 1)) : t;         // staging slot of this thread
    int bx, by, k;
    xcd_decode(blockIdx.x, nx, ny, nz, bx, by, k);
    const int co0 = (by / ci_groups) * 16, ci0 = (by % ci_groups) * CIB;
    const int Cin = g.Cin, Cout = g.Cout, H = g.H, W = g.W, Ho = g.Ho, Wo = g.Wo;
    const bool do_bias = (ci0 == 0) && (g.b_off >= 0);
    const int cot = min(16, Cout - co0), cit = min(CIB, Cin - ci0);
    // 4-channel remainder group on the 4x4x1 matrix instruction: full-width tiles of the specialised / 4-wave variants only (the others keep
    // the padded fragment); decided per tile below, the epilogue follows `x4_used`
    const bool x4_blk = KS <= 3 && NB == 1 && NW == 4 && ci0 > 0 && cit <= 4 && Wo % TW == 0;

    if (t < CIB) s_chx[t] = chan_fwd(in, k, min(ci0 + t, Cin - 1));
    if (t >= 64 && t < 80) s_chg[t - 64] = chan_bwd(gy, k, min(co0 + t - 64, Cout - 1));

    f32x4 acc[NB][KK], accb = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < KK; ++q) acc[b][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* __restrict__ xin = in.data + (long long)k * in.sstride;
    const float* __restrict__ gap = gy.ga + (long long)k * gy.gstride;
    const float* __restrict__ yp = gy.y ? gy.y + (long long)k * gy.ystride : nullptr;
    const int HW = H * W, HWo = Ho * Wo;

    // Staging slots are fixed per thread: ONE output pixel of the dy tile for 16/CPP channels, NPOS positions of the
    // input tile for all CIB channels — so a tile costs a handful of index computations, not one per element.  The next
    // tile is prefetched into registers while the matrix cores work on the current one.
    constexpr int PIX = TH * TW, CPP = NS / PIX, NGC = 16 / CPP;           // CPP channels are staged per pass of NS threads
    constexpr int NPOS = (IN_TH * IN_TW + NS - 1) / NS;
    static_assert(NS % PIX == 0 && 16 % CPP == 0, "tile shape");
    const int gpix = ts % PIX, gsub = ts / PIX;
    float gr[NGC], yr[NGC], xr[NPOS][CIB];
    int goff = -1, xoff[NPOS];

    auto prefetch = [&](int tile) {
        const int ox0 = (tile % tiles_x) * TW, oy0 = (tile / tiles_x) * TH;
        const int yy = oy0 + gpix / TW, xx = ox0 + (gpix % TW);
        goff = (yy < Ho && xx < Wo) ? yy * Wo + xx : -1;
        // branch-free: every load uses a valid (clamped) address; invalid pixels / channels are zeroed at the LDS store
        const int gsafe = max(goff, 0);
#pragma unroll
        for (int j = 0; j < NGC; ++j) {
            const int off = (co0 + min(j * CPP + gsub, cot - 1)) * HWo + gsafe;
            gr[j] = gap[off];
            yr[j] = yp ? yp[off] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NPOS; ++q) {
            const int p = ts + NS * q;
            xoff[q] = -1;
            if ((ts & ~63) + NS * q >= IN_TH * IN_TW) continue;          // no lane of this wave owns a position in slot q
            int xsafe = 0;
            if (p < IN_TH * IN_TW) {
                const int iy = p / IN_TW, ix = p - iy * IN_TW;
                int gyy = reflect_idx(oy0 * STRIDE + iy - P, H), gxx = reflect_idx(ox0 * STRIDE + ix - P, W);
                gyy = min(max(gyy, 0), H - 1); gxx = min(max(gxx, 0), W - 1);         // overhang meets dy == 0
                xoff[q] = xsafe = gyy * W + gxx;
            }
            const float* __restrict__ px = xin + (long long)ci0 * HW + xsafe;
#pragma unroll
            for (int c = 0; c < CIB; ++c) xr[q][c] = px[(long long)min(c, cit - 1) * HW];
        }
    };

    auto stage = [&]() {                 // registers -> LDS with the deferred transforms
#pragma unroll
        for (int j = 0; j < NGC; ++j) {
            const int c = j * CPP + gsub;
            float v = 0.f;
            if (goff >= 0 && c < cot) v = yp ? apply_bwd(s_chg[c], gr[j], yr[j]) : gr[j];
            s_g[c * GPLANE + gpix] = v;
        }
#pragma unroll
        for (int q = 0; q < NPOS; ++q) {
            const int p = ts + NS * q;
            if (p < IN_TH * IN_TW) {
                const int iy = p / IN_TW, lp = iy * WV + (p - iy * IN_TW) + XOFF;
#pragma unroll
                for (int c = 0; c < CIB; ++c) s_x[c * XPLANE + lp] = c < cit ? apply_fwd(s_chx[c], xr[q][c], in.act, in.slope) : 0.f;
            }
        }
    };
    // k-steps of 4 consecutive pixels of one row, dealt round-robin to the MFMA waves.  The operands of k-step i+1 are read
    // from LDS into a second register set while the matrix core works through the NB*KK MFMAs of k-step i.
    // Only the k-steps that touch pixels inside the map are issued (dy is zero outside): on the 8x8 / 16x16 maps at the bottom
    // of the hour-glass that is 1/4 / 1/2 of the 8 x 32 tile.
    auto mfma_tile = [&](int tile) {
        constexpr int NOP = NB * KK;
        const int vr = min(TH, Ho - (tile / tiles_x) * TH), vc4 = (min(TW, Wo - (tile % tiles_x) * TW) + 3) >> 2;
        const int NKS = vr * vc4;
        // Full-width tiles with 4 MFMA waves (the hot case): wave wv owns pixel columns 4wv..4wv+3 and 4wv+16..+19 of every row, so the
        // operands of its two k-steps per row sit at fixed offsets from two pointers that advance by one row — no address arithmetic per
        // k-step — and the loop is software-pipelined at instruction level: LDS read q of k-step i+1, then MFMA q of k-step i, pinned in
        // that order (every operand is requested a full k-step = 9*NB MFMAs before its use).  Written as "all reads of i+1, fence, all
        // MFMAs of i" (the generic path below) the ~30 non-matrix instructions of a k-step issue back to back while the matrix pipe
        // drains (in-order issue): 40 % of the pipe's rate inside the MFMA phase.
        auto k_loop = [&](auto bias_c, auto fullw_c, auto x4_c) {
            constexpr bool BIAS = decltype(bias_c)::value, FULLW = decltype(fullw_c)::value, X4 = decltype(x4_c)::value;
            float a[2], bq[2][NOP];
            auto load = [&](int ks, float& aa, float (&bb)[NOP]) {
                const int row = ks / vc4, c4 = (ks - row * vc4) * 4;
                aa = s_g[l15 * GPLANE + row * TW + c4 + l4];
                const float* xb = s_x + l15 * XPLANE + (row * STRIDE) * WV + (c4 + l4) * STRIDE + XOFF;
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) bb[b * KK + ky * KS + kx] = xb[b * 16 * XPLANE + ky * WV + kx];
            };
            auto fma_all = [&](float aa, const float (&bb)[NOP]) {
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int q = 0; q < KK; ++q) acc[b][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, bb[b * KK + q], acc[b][q], 0, 0, 0);
                if constexpr (BIAS) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, 1.0f, accb, 0, 0, 0);
            };
            if constexpr (FULLW && NW == 4) {
                // MFMAs of the k-step in (aa, bb) with the reads of the k-step at (gp, xp) into (an, bn) woven in
                auto step = [&](float aa, const float (&bb)[NOP], const float* gp, const float* xp, float& an, float (&bn)[NOP]) {
#pragma unroll
                    for (int b = 0; b < NB; ++b)
#pragma unroll
                        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                            for (int kx = 0; kx < KS; ++kx) {
                                const int q = b * KK + ky * KS + kx;
                                bn[q] = xp[b * 16 * XPLANE + ky * WV + kx];
                                if (q == 0) an = gp[0];
                                if constexpr (X4) acc[b][ky * KS + kx] = __builtin_amdgcn_mfma_f32_4x4x1f32(aa, bb[q], acc[b][ky * KS + kx], 0, 0, 0);
                                else acc[b][ky * KS + kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, bb[q], acc[b][ky * KS + kx], 0, 0, 0);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                    if constexpr (BIAS) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, 1.0f, accb, 0, 0, 0);
                };
                const float* ga = s_g + l15 * GPLANE + wv * 4 + l4;
                // X4: the block's input-channel group holds only the layer's last 4 channels (36 / 68 / 132 = 16n + 4): 16 independent 4x4 outer
                // products per instruction (v_mfma_f32_4x4x1: block b = lane >> 2 pairs dy[co = 4*(b & 3) + i][pixel slice b >> 2], the lanes of
                // the ordinary A fragment, with x[pixel][channel lane & 3]) — 8 cycles per tap instead of a 32-cycle fragment with 12 empty columns
                const float* xa = s_x + (X4 ? (l15 & 3) : l15) * XPLANE + (wv * 4 + l4) * STRIDE + XOFF;
                a[0] = ga[0];
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) bq[0][b * KK + ky * KS + kx] = xa[b * 16 * XPLANE + ky * WV + kx];
                for (int r = 0; r < vr; ++r) {
                    step(a[0], bq[0], ga + 16, xa + 16 * STRIDE, a[1], bq[1]);
                    if (r + 1 < vr) { ga += TW; xa += STRIDE * WV; }          // last row: re-read it (values unused)
                    step(a[1], bq[1], ga, xa, a[0], bq[0]);
                }
                return;
            }
            int ks = wv;
            if (ks >= NKS) return;
#ifndef MFVI_FENCE
#define MFVI_FENCE 1
#endif
            load(ks, a[0], bq[0]);
            for (; ks + NW < NKS; ks += 2 * NW) {
                load(ks + NW, a[1], bq[1]);
                if (MFVI_FENCE) __builtin_amdgcn_sched_barrier(0);
                fma_all(a[0], bq[0]);
                if (MFVI_FENCE) __builtin_amdgcn_sched_barrier(0);
                if (ks + 2 * NW < NKS) load(ks + 2 * NW, a[0], bq[0]);
                if (MFVI_FENCE) __builtin_amdgcn_sched_barrier(0);
                fma_all(a[1], bq[1]);
                if (MFVI_FENCE) __builtin_amdgcn_sched_barrier(0);
            }
            if (ks < NKS) fma_all(a[0], bq[0]);
        };
        constexpr std::true_type yes{}; constexpr std::false_type no{};
        if (vc4 == TW / 4) { if (x4_blk) k_loop(no, yes, yes); else if (do_bias) k_loop(yes, yes, no); else k_loop(no, yes, no); }
        else { if (do_bias) k_loop(yes, no, no); else k_loop(no, no, no); }
    };

    const int tile_begin = bx * tiles_per_block, tile_end = min(n_tiles, tile_begin + tiles_per_block);
    if constexpr (SPEC) {
        if (producer) {
            // Producer wave pw stages input channels [pw*CPW, (pw+1)*CPW) and gradient channels [pw*4, pw*4+4): the channel is
            // wave-uniform, so its BN constants sit in registers and every element costs fma + select + one LDS store.
            constexpr int CPW = CIB / 4;                                  // input channels per producer wave
            // aligned float4 items: input window row iy, float4 column v (NV4 per row); dy tile row, float4 column (TW/4 per row).
            // A float4 is wholly inside or outside the image (launcher: W, Wo multiples of 4); reflection needs one element of
            // an outside float4 (column -1 <- x[1], column W <- x[W-2]), taken from the neighbouring inside one.
            constexpr int NV4 = WV / 4, NXI = IN_TH * NV4, NPX = (NXI + 63) / 64;      // input items / passes of 64 lanes
            constexpr int NGI = PIX / 4, NPG = (NGI + 63) / 64;                        // dy items / passes
            const int pw = wv - NW;
            __builtin_amdgcn_s_setprio(2);                                // younger half of the workgroup: do not starve behind the MFMA stream
            float4 pxr[CPW][NPX], pgr[4][NPG], pyr[4][NPG];
            int pxo[NPX], pgo[NPG];                                       // element offsets; input: low 2 bits = 1 left / 2 right reflected
            auto pfetch = [&](int tile) {
                const int ox0 = (tile % tiles_x) * TW, oy0 = (tile / tiles_x) * TH;
#pragma unroll
                for (int j = 0; j < NPG; ++j) {
                    const int q = min(lane + 64 * j, NGI - 1), yy = oy0 + q / (TW / 4), xx = ox0 + (q % (TW / 4)) * 4;
                    pgo[j] = (yy < Ho && xx < Wo) ? yy * Wo + xx : -1;
                    const int gsafe = max(pgo[j], 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int off = (co0 + min(pw * 4 + i, cot - 1)) * HWo + gsafe;
                        pgr[i][j] = *reinterpret_cast<const float4*>(gap + off);
                        pyr[i][j] = yp ? *reinterpret_cast<const float4*>(yp + off) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                const int ax0 = ox0 * STRIDE - Cfg::HALO4;
#pragma unroll
                for (int j = 0; j < NPX; ++j) {
                    const int q = min(lane + 64 * j, NXI - 1), iy = q / NV4, v = q - iy * NV4;
                    int gyy = reflect_idx(oy0 * STRIDE + iy - P, H); gyy = min(max(gyy, 0), H - 1);      // overhang meets dy == 0
                    int gx = ax0 + 4 * v, flag = 0;
                    if (gx < 0) { flag = 1; gx = 0; } else if (gx >= W) { flag = gx == W ? 2 : 0; gx = W - 4; }
                    pxo[j] = (gyy * W + gx) | flag;
                    if (pw * CPW >= cit) continue;          // this wave's input channels lie beyond the layer (4-channel remainder group): nothing to stage
#pragma unroll
                    for (int i = 0; i < CPW; ++i)
                        pxr[i][j] = *reinterpret_cast<const float4*>(xin + (long long)(ci0 + min(pw * CPW + i, cit - 1)) * HW + (pxo[j] & ~3));
                }
            };
            auto pstage = [&]() {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = pw * 4 + i;
                    const ChanBwd cgk = s_chg[c];                     // wave-uniform: one LDS read per channel and tile
#pragma unroll
                    for (int j = 0; j < NPG; ++j) {
                        const int q = lane + 64 * j;
                        float e[4] = {pgr[i][j].x, pgr[i][j].y, pgr[i][j].z, pgr[i][j].w};
                        if (yp) {
                            const float yy[4] = {pyr[i][j].x, pyr[i][j].y, pyr[i][j].z, pyr[i][j].w};
                            apply_bwd4(cgk, e, yy);
                        }
                        if (pgo[j] < 0 || c >= cot) { e[0] = 0.f; e[1] = 0.f; e[2] = 0.f; e[3] = 0.f; }
                        if (64 * (j + 1) <= NGI || q < NGI) lds_store4(s_g + c * GPLANE + 4 * q, e[0], e[1], e[2], e[3]);
                    }
                }
#pragma unroll
                for (int i = 0; i < CPW; ++i) {
                    const int c = pw * CPW + i;
                    if (x4_blk && pw * CPW >= cit) break;          // the 4x4x1 path reads channels 0..3 only
                    const ChanFwd cx = s_chx[c];
#pragma unroll
                    for (int j = 0; j < NPX; ++j) {
                        const int q = lane + 64 * j, flag = pxo[j] & 3;
                        float e[4] = {pxr[i][j].x, pxr[i][j].y, pxr[i][j].z, pxr[i][j].w};
                        apply_fwd4(cx, e, in.act, in.slope);
                        if (c >= cit) { e[0] = 0.f; e[1] = 0.f; e[2] = 0.f; e[3] = 0.f; }
                        if (flag == 1) e[3] = e[1];                       // column -1 <- x[1]
                        else if (flag == 2) e[0] = e[2];                  // column W  <- x[W-2]
                        if (64 * (j + 1) <= NXI || q < NXI) lds_store4(s_x + c * XPLANE + 4 * q, e[0], e[1], e[2], e[3]);
                    }
                }
            };
            if (tile_begin < tile_end) pfetch(tile_begin);               // requested before the channel tables exist: only the staging transform needs them
            __syncthreads();                                             // (S0) channel tables visible
            for (int tile = tile_begin; tile < tile_end; ++tile) {
                if (tile > tile_begin) __syncthreads();                  // (B1) consumers are done with the previous tile
                pstage();
                __syncthreads();                                         // (B2) tile published
                if (tile + 1 < tile_end) pfetch(tile + 1);
            }
        } else {
            __syncthreads();                                             // (S0)
            for (int tile = tile_begin; tile < tile_end; ++tile) {
                if (tile > tile_begin) __syncthreads();                  // (B1)
                __syncthreads();                                         // (B2)
                mfma_tile(tile);
            }
        }
    } else {
        if (tile_begin < tile_end) prefetch(tile_begin);
        for (int tile = tile_begin; tile < tile_end; ++tile) {
            __syncthreads();
            stage();
            __syncthreads();
            if (tile + 1 < tile_end) prefetch(tile + 1);
            mfma_tile(tile);
        }
    }

    // ---- sum the MFMA waves through LDS with plain stores (LDS float atomics cost ~700 cycles per wave instruction here):
    //      wave w writes its fragments to region w & 3; with 8 MFMA waves, waves 4-7 then add theirs onto waves 0-3's (same
    //      lane -> address map, so no conflicts); the 4 regions are summed on the way to global memory.
    //      D layout: column n (ci) = lane & 15, row m (co) = (lane >> 4) * 4 + reg ----
    constexpr int ROWP = Cfg::ROWP;
    __syncthreads();
    float* s_ep = lds;                      // [4][16][ROWP]   dW fragments per wave region, element (co, ci*KK + tap)
    float* s_db = lds + 4 * 16 * ROWP;      // [4][16]         bias-gradient partials
    const int rw = wv & 3;
    auto put = [&](bool add) {
        if (x4_blk) {       // 4x4x1 accumulators: lane 4b + j, register i = dW[co = 4*(b & 3) + i][ci = j] of pixel slice b >> 2: add the four slices first
#pragma unroll
            for (int q = 0; q < KK; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[0][q][r]; v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                    if (lane < 16) s_ep[(rw * 16 + 4 * (lane >> 2) + r) * ROWP + (lane & 3) * KK + q] = v;
                }
            return;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < KK; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* d = &s_ep[(rw * 16 + l4 * 4 + r) * ROWP + (b * 16 + l15) * KK + q];
                    *d = add ? *d + acc[b][q][r] : acc[b][q][r];
                }
        if (do_bias && l15 == 0)
#pragma unroll
            for (int r = 0; r < 4; ++r) { float* d = &s_db[rw * 16 + l4 * 4 + r]; *d = add ? *d + accb[r] : accb[r]; }
    };
    if (!producer && wv < 4) put(false);
    if (NW == 8) { __syncthreads(); if (wv >= 4) put(true); }
    __syncthreads();

    // ---- this block's partial sums -> its slab (contiguous rows; every (strip, sample) slab is covered exactly once) ----
    const int len = cit * KK;
    float* __restrict__ o = part + ((long long)bx * nz + k) * part_stride;
    for (int idx = t; idx < 16 * len; idx += NT) {
        const int r = idx / len, rel = idx - r * len, co = co0 + r;
        if (co < Cout) {
            const float* e = s_ep + r * ROWP + rel;
            o[((long long)co * Cin + ci0) * KK + rel] = (e[0] + e[16 * ROWP]) + (e[2 * 16 * ROWP] + e[3 * 16 * ROWP]);
        }
    }
    if (do_bias && t < cot) o[(long long)Cout * Cin * KK + co0 + t] = (s_db[t] + s_db[16 + t]) + (s_db[32 + t] + s_db[48 + t]);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Fragment-split variant for the 3x3 stride-1 layers on full-width tiles (w field 10).  A block owns 16 output channels and a group
// of up to 36 input channels (two 16-channel tiles + the 4-channel remainder of the 16n + 4 concat layers), so dy — the operand whose
// BN-backward staging the other variants repeat once per 16 input channels — is staged ONCE per 32-36 of them.  What makes that fit
// the register file: the (input tile, tap) accumulator fragments (up to 18 + 9 of the 4x4x1 kind) are dealt to the 4 consumer waves
// (wave w owns fragments w, w + 4, ...), each wave sweeps ALL k-steps of the tile for its own <= 7 fragments, and there is no
// cross-wave reduction at the end.  Same tile, staging and slab layout as the specialised variant above.
struct SCfg {
    static constexpr int TW = 32, TH = 8, KK = 9, CIB = 36;
    static constexpr int IN_TH = TH + 2, IN_TW = TW + 2, HALO4 = 4, XOFF = 3;
    static constexpr int WV = (XOFF + IN_TW + 3) / 4 * 4;                 // 40
    static constexpr int GPLANE = pitch2(TH * TW), XPLANE = pitch2(IN_TH * WV);
    // + 64: the software pipeline requests the first k-step of the row BELOW the tile's last one (values unused); for the last plane that
    // row starts up to 22 floats past the planes — kept inside the block's own allocation
    static constexpr int STAGE = 16 * GPLANE + CIB * XPLANE + 64;
    static constexpr int ROW = CIB * KK, ROWP = ROW + 2;
    static constexpr int EPI = 16 * ROWP + 16;
    static constexpr int LDS_FLOATS = STAGE > EPI ? STAGE : EPI;
};

__global__ __launch_bounds__(512, 4) void conv_bww_split_kernel(TView in, GView gy, ConvGeom g, float* __restrict__ part, long long part_stride,
                                                                int tiles_x, int n_tiles, int tiles_per_block, int ci_groups, int nx, int ny, int nz)
{
    using Cfg = SCfg;
    constexpr int TW = Cfg::TW, TH = Cfg::TH, KK = Cfg::KK, P = 1, IN_TH = Cfg::IN_TH, CIB = Cfg::CIB;
    constexpr int GPLANE = Cfg::GPLANE, XPLANE = Cfg::XPLANE, WV = Cfg::WV, XOFF = Cfg::XOFF, ROWP = Cfg::ROWP;
    extern __shared__ __align__(16) float lds[];
    __shared__ ChanFwd s_chx[CIB];
    __shared__ ChanBwd s_chg[16];
    float* s_g = lds;                    // [16][GPLANE]
    float* s_x = lds + 16 * GPLANE;      // [CIB][XPLANE]

    const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, l4 = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);      // wave-uniform for the compiler too: the per-wave dispatch below must be scalar branches
    const bool producer = wv >= 4;
    int bx, by, k;
    xcd_decode(blockIdx.x, nx, ny, nz, bx, by, k);
    const int co0 = (by / ci_groups) * 16, ci0 = (by % ci_groups) * 32;
    const int Cin = g.Cin, Cout = g.Cout, H = g.H, W = g.W, Ho = g.Ho, Wo = g.Wo;
    const bool do_bias = (ci0 == 0) && (g.b_off >= 0);
    const int cot = min(16, Cout - co0), cit = min(CIB, Cin - ci0);      // launcher: cit in {16, 20, 32, 36}
    const int nfull = cit >> 4; const bool x4 = (cit & 15) != 0;

    if (t < CIB) s_chx[t] = chan_fwd(in, k, min(ci0 + t, Cin - 1));
    if (t >= 64 && t < 80) s_chg[t - 64] = chan_bwd(gy, k, min(co0 + t - 64, Cout - 1));

    const float* __restrict__ xin = in.data + (long long)k * in.sstride;
    const float* __restrict__ gap = gy.ga + (long long)k * gy.gstride;
    const float* __restrict__ yp = gy.y ? gy.y + (long long)k * gy.ystride : nullptr;
    const int HW = H * W, HWo = Ho * Wo;
    const int tile_begin = bx * tiles_per_block, tile_end = min(n_tiles, tile_begin + tiles_per_block);

    constexpr int MAXF = 7;
    float* s_ep = lds;                      // [16][ROWP]   (epilogue, over the tile)
    float* s_db = lds + 16 * ROWP;          // [16]

    if (producer) {
        // producer wave pw stages input channels [9 pw, 9 pw + 9) as the specialised variant does (72 registers of prefetched tile); the
        // gradient tile is staged by the CONSUMER waves, which have the registers to spare and idle between (B1) and (B2) anyway
        constexpr int CPW = CIB / 4;
        constexpr int NV4 = WV / 4, NXI = IN_TH * NV4, NPX = (NXI + 63) / 64;
        const int pw = wv - 4;
        __builtin_amdgcn_s_setprio(2);
        float4 pxr[CPW][NPX];
        int pxo[NPX];
        auto pfetch = [&](int tile) {
            const int ox0 = (tile % tiles_x) * TW, oy0 = (tile / tiles_x) * TH;
            const int ax0 = ox0 - Cfg::HALO4;
#pragma unroll
            for (int j = 0; j < NPX; ++j) {
                const int q = min(lane + 64 * j, NXI - 1), iy = q / NV4, v = q - iy * NV4;
                int gyy = reflect_idx(oy0 + iy - P, H); gyy = min(max(gyy, 0), H - 1);
                int gx = ax0 + 4 * v, flag = 0;
                if (gx < 0) { flag = 1; gx = 0; } else if (gx >= W) { flag = gx == W ? 2 : 0; gx = W - 4; }
                pxo[j] = (gyy * W + gx) | flag;
#pragma unroll
                for (int i = 0; i < CPW; ++i)
                    pxr[i][j] = *reinterpret_cast<const float4*>(xin + (long long)(ci0 + min(pw * CPW + i, cit - 1)) * HW + (pxo[j] & ~3));
            }
        };
        auto pstage = [&]() {
#pragma unroll
            for (int i = 0; i < CPW; ++i) {
                const int c = pw * CPW + i;
                if (c >= cit) break;
                const ChanFwd cx = s_chx[c];
#pragma unroll
                for (int j = 0; j < NPX; ++j) {
                    const int q = lane + 64 * j, flag = pxo[j] & 3;
                    float e[4] = {pxr[i][j].x, pxr[i][j].y, pxr[i][j].z, pxr[i][j].w};
                    apply_fwd4(cx, e, in.act, in.slope);
                    if (flag == 1) e[3] = e[1];                       // column -1 <- x[1]
                    else if (flag == 2) e[0] = e[2];                  // column W  <- x[W-2]
                    if (64 * (j + 1) <= NXI || q < NXI) lds_store4(s_x + c * XPLANE + 4 * q, e[0], e[1], e[2], e[3]);
                }
            }
        };
        if (tile_begin < tile_end) pfetch(tile_begin);
        __syncthreads();                                             // (S0) channel tables visible
        for (int tile = tile_begin; tile < tile_end; ++tile) {
            if (tile > tile_begin) __syncthreads();                  // (B1) consumers are done with the previous tile
            pstage();
            __syncthreads();                                         // (B2) tile published
            if (tile + 1 < tile_end) pfetch(tile + 1);
        }
        __syncthreads();                                             // (E) consumers are done with the last tile
    } else {
        // (the accumulators live in this branch only: the producers need their registers for the prefetched tile)
        f32x4 acc[MAXF], accb;
        // consumer wave wv prefetches / stages gradient channels [4 wv, 4 wv + 4): one aligned float4 of dy and of y per lane and channel
        constexpr int PIX = TH * TW, NGI = PIX / 4;
        static_assert(NGI == 64, "one float4 per lane");
        float4 pgr[4], pyr[4];
        int pgo = -1;
        auto cfetch = [&](int tile) {
            const int ox0 = (tile % tiles_x) * TW, oy0 = (tile / tiles_x) * TH;
            const int yy = oy0 + lane / (TW / 4), xx = ox0 + (lane % (TW / 4)) * 4;
            pgo = (yy < Ho && xx < Wo) ? yy * Wo + xx : -1;
            const int gsafe = max(pgo, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = (co0 + min(wv * 4 + i, cot - 1)) * HWo + gsafe;
                pgr[i] = *reinterpret_cast<const float4*>(gap + off);
                pyr[i] = yp ? *reinterpret_cast<const float4*>(yp + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto cstage = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = wv * 4 + i;
                const ChanBwd cgk = s_chg[c];
                float e[4] = {pgr[i].x, pgr[i].y, pgr[i].z, pgr[i].w};
                if (yp) {
                    const float yy[4] = {pyr[i].x, pyr[i].y, pyr[i].z, pyr[i].w};
                    apply_bwd4(cgk, e, yy);
                }
                if (pgo < 0 || c >= cot) { e[0] = 0.f; e[1] = 0.f; e[2] = 0.f; e[3] = 0.f; }
                lds_store4(s_g + c * GPLANE + 4 * lane, e[0], e[1], e[2], e[3]);
            }
        };
        // ---- consumer wave W of 4, NF full input tiles, X4 = remainder group present: fragments f = W + 4j < 9 NF + 9 X4;
        //      f < 9 NF: 16x16x4 on input tile f / 9, tap f % 9;  else the 4x4x1 instruction on channels 16 NF .. 16 NF + 3, tap f - 9 NF.
        //      Software pipeline as in the specialised variant: the LDS reads of k-step i + 1 are woven between the MFMAs of k-step i.
        auto sweep = [&](auto w_c, auto nf_c, auto x4_c, auto bias_c) {
            constexpr int WI = decltype(w_c)::value, NF = decltype(nf_c)::value;
            constexpr bool X4 = decltype(x4_c)::value, BIAS = decltype(bias_c)::value;
            constexpr int NFRAG = 9 * NF + (X4 ? 9 : 0);
            constexpr int NJ = (NFRAG - WI + 3) / 4;                     // fragments of this wave
            static_assert(NJ >= 1 && NJ <= MAXF, "fragment count");
            const float* xb16 = s_x + l15 * XPLANE + l4 + XOFF;                          // full fragments: plane = channel l15 of the tile
            const float* xb4 = s_x + (NF * 16 + (l15 & 3)) * XPLANE + l4 + XOFF;         // 4x4x1 fragments: plane = remainder channel lane & 3
            const float* ga = s_g + l15 * GPLANE + l4;
            auto boff = [](int j) constexpr { const int f = WI + 4 * j; const bool full = f < 9 * NF; const int tp = full ? f % 9 : f - 9 * NF;
                                               return (full ? (f / 9) * 16 * XPLANE : 0) + (tp / 3) * WV + (tp % 3); };
            auto isfull = [](int j) constexpr { return WI + 4 * j < 9 * NF; };
            float a[2], b[2][NJ];
            // (first request, table barrier and accumulator reset sit INSIDE the per-wave variant: placed in front of the dispatch, the
            //  loop-invariant addresses the compiler hoists out of all 16 variants were live together with them: 146 registers)
            if (tile_begin < tile_end) cfetch(tile_begin);
            __syncthreads();                                             // (S0)
            accb = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < MAXF; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // reads of the k-step at (gp, xp16 / xp4) into (an, bn), woven into the MFMAs of (aa, bb)
            auto step = [&](float aa, const float (&bb)[NJ], const float* gp, const float* xp16, const float* xp4, float& an, float (&bn)[NJ]) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    bn[j] = isfull(j) ? xp16[boff(j)] : xp4[boff(j)];
                    if (j == 0) an = gp[0];
                    if (isfull(j)) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, bb[j], acc[j], 0, 0, 0);
                    else acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(aa, bb[j], acc[j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (BIAS) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, 1.0f, accb, 0, 0, 0);
            };
            for (int tile = tile_begin; tile < tile_end; ++tile) {
                if (tile > tile_begin) __syncthreads();                  // (B1)
                cstage();
                __syncthreads();                                         // (B2)
                const int vr = min(TH, Ho - (tile / tiles_x) * TH);
                const float* gp = ga; const float* xp16 = xb16; const float* xp4 = xb4;
                a[0] = gp[0];
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[0][j] = isfull(j) ? xp16[boff(j)] : xp4[boff(j)];
                auto row = [&]() {
                    // 8 k-steps of the row, every operand at a compile-time offset from the three row pointers; the last one requests the first
                    // k-step of the next row (below the tile's last row that is a row of the next plane, or zeros past the allocation: unused)
#pragma unroll
                    for (int c = 0; c < 8; c += 2) {
                        step(a[0], b[0], gp + 4 * (c + 1), xp16 + 4 * (c + 1), xp4 + 4 * (c + 1), a[1], b[1]);
                        if (c + 2 < 8) step(a[1], b[1], gp + 4 * (c + 2), xp16 + 4 * (c + 2), xp4 + 4 * (c + 2), a[0], b[0]);
                        else step(a[1], b[1], gp + TW, xp16 + WV, xp4 + WV, a[0], b[0]);
                    }
                    gp += TW; xp16 += WV; xp4 += WV;
                };
                for (int r = 0; r + 1 < vr; ++r) row();
                // the next tile's gradient is requested before the LAST row only: its 32 registers are then live for 1/8 of the sweep instead
                // of all of it (held across the whole sweep they did not fit beside the accumulators: 57 spilled registers)
                if (tile + 1 < tile_end) cfetch(tile + 1);
                __builtin_amdgcn_sched_barrier(0);
                row();
            }
        };
        constexpr std::true_type yes{}; constexpr std::false_type no{};
        constexpr std::integral_constant<int, 0> w0{}; constexpr std::integral_constant<int, 1> w1{}; constexpr std::integral_constant<int, 2> w2{};
        constexpr std::integral_constant<int, 3> w3{}; constexpr std::integral_constant<int, 1> n1{}; constexpr std::integral_constant<int, 2> n2{};
        // the bias fragment rides with wave 3 (the lightest fragment list)
        auto by_wave = [&](auto nf_c, auto x4_c) {
            if (wv == 0) sweep(w0, nf_c, x4_c, no);
            else if (wv == 1) sweep(w1, nf_c, x4_c, no);
            else if (wv == 2) sweep(w2, nf_c, x4_c, no);
            else if (do_bias) sweep(w3, nf_c, x4_c, yes);
            else sweep(w3, nf_c, x4_c, no);
        };
        if (nfull == 2) { if (x4) by_wave(n2, yes); else by_wave(n2, no); }
        else { if (x4) by_wave(n1, yes); else by_wave(n1, no); }
        // ---- every fragment is complete in its wave: transpose through LDS into slab rows (co, ci * 9 + tap), then contiguous stores
        __syncthreads();                                             // (E)
        const int nfrag = 9 * nfull + (x4 ? 9 : 0);
#pragma unroll
        for (int j = 0; j < MAXF; ++j) {
            const int f = wv + 4 * j;
            if (f >= nfrag) break;
            if (f < 9 * nfull) {
                const int gi = f / 9, tp = f - gi * 9;
#pragma unroll
                for (int r = 0; r < 4; ++r) s_ep[(l4 * 4 + r) * ROWP + (gi * 16 + l15) * KK + tp] = acc[j][r];
            } else {
                const int tp = f - 9 * nfull;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[j][r]; v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                    if (lane < 16) s_ep[(4 * (lane >> 2) + r) * ROWP + (nfull * 16 + (lane & 3)) * KK + tp] = v;
                }
            }
        }
        if (do_bias && wv == 3 && l15 == 0)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_db[l4 * 4 + r] = accb[r];
    }
    __syncthreads();
    const int len = cit * KK;
    float* __restrict__ o = part + ((long long)bx * nz + k) * part_stride;
    for (int idx = t; idx < 16 * len; idx += 512) {
        const int r = idx / len, rel = idx - r * len, co = co0 + r;
        if (co < Cout) o[((long long)co * Cin + ci0) * KK + rel] = s_ep[r * ROWP + rel];
    }
    if (do_bias && t < cot) o[(long long)Cout * Cin * KK + co0 + t] = s_db[t];
}

int env_tune_w()
{
    static const int t = [] { int nb = 0, w = 0, tb = 0; const char* e = getenv("MFVI_TUNE_W"); if (e) sscanf(e, "%d,%d,%d", &nb, &w, &tb); return nb > 0 ? (nb | w << 8 | tb << 16) : 0; }();
    return t;
}

}  // namespace

// Tiling (ConvGeom::tune[2], MFVI_TUNE_W=nb,w,target/256): nb | w << 8 | (target blocks / 256) << 16 with w = 4 (4 waves),
// 8 (8 waves), 9 (8 waves, producer/consumer specialised), 10 (fragment-split variant, nb = 2: 3x3 stride 1, full-width tiles) or
// 11 (bf16x6 kernel of conv_bww_x6.hip, nb = 16-channel output fragments per block: 3x3 stride 1 on maps whose width is a multiple of 64); 0 = heuristic.
int launch_conv_bwd_weight_mfma(const TView& in, const GView& gy, const ConvGeom& g, BwwPart part, int* strips_used, int n_samples,
                                hipStream_t st)
{
    if (!part.base || part.max_strips < 1 || (g.Cin & 3) || (g.w_off & 3)) return -2;
    if (g.tune[2] & MFVI_TUNE_GENERIC) return -2;                           // in-kernel eps: the generic kernel accumulates d mu / d rho itself
    int cfg = g.tune[2] ? g.tune[2] : env_tune_w();
    const bool forced = cfg != 0;
    if (!cfg) {
        // heuristic: the bf16x6 kernel where it serves the shape and measured ahead of the fp32 ones (3x3 stride 1, >= 32 input channels, maps a
        // multiple of 32 wide: profiles/r03_x6_layers.txt); otherwise stage dy once for up to 48 input channels when the layer has them, 8 waves
        // when the tile is big
        static const bool x6_on = [] { const char* e = getenv("MFVI_X6"); return !(e && e[0] == '0'); }();
        if (x6_on && g.ks == 3 && g.stride == 1 && !(g.W & 31) && !(g.H & 1) && g.H >= 4 && g.Cin >= 32 && ((g.Cin & 15) == 0 || (g.Cin & 15) == 4)) {
            const int rc = launch_conv_bwd_weight_x6(in, gy, g, part, strips_used, g.Cout >= 32 ? 2 : 1, 256, n_samples, st);
            if (rc != -2 && rc != -3) return rc;
        }
        const int nb = g.ks == 5 ? 1 : (g.Cin > 32 ? 3 : (g.Cin > 16 ? 2 : 1));
        cfg = nb | ((nb >= 2 || g.ks == 5 ? 9 : 4) << 8) | ((nb >= 2 ? 1 : 6) << 16);
    }
    // aligned float4 staging of the specialised variant: image rows, sample strides and base pointers multiples of 4 floats
    const bool vec_ok = !(g.W & 3) && g.W >= 4 && !(g.Wo & 3) && !(in.sstride & 3) && !((uintptr_t)in.data & 15) && !(gy.gstride & 3) &&
                        !((uintptr_t)gy.ga & 15) && (!gy.y || (!(gy.ystride & 3) && !((uintptr_t)gy.y & 15)));
    if (!forced && !vec_ok) cfg = (cfg & ~0xff00) | (4 << 8);      // heuristic falls back to the 4-wave variant
    const int nb = cfg & 255, wfield = (cfg >> 8) & 255, target = ((cfg >> 16) & 255) * 256;
    const bool spec = wfield == 9;
    if (wfield == 11) return launch_conv_bwd_weight_x6(in, gy, g, part, strips_used, nb, target, n_samples, st);      // bf16x6 kernel (conv_bww_x6.hip), nb = output fragments per block
    if (wfield == 10) {
        // fragment-split variant: 3x3 stride 1, full-width tiles, aligned float4 staging, input-channel groups of 32 with a 4-channel remainder
        // riding in the last one (Cin = 16n or 16n + 4)
        if (g.ks != 3 || g.stride != 1 || !vec_ok || (g.Wo & 31) || nb != 2 || target < 256) return -3;
        const int rem = g.Cin & 15;
        if ((rem != 0 && rem != 4) || g.Cin < 16) return -3;
        const int ci_groups = (g.Cin % 32 == 4 || g.Cin % 32 == 0) ? g.Cin / 32 : g.Cin / 32 + 1;       // 36 -> 1, 68 -> 2, 132 -> 4, 48 -> 2 (32 + 16), 52 -> 2 (32 + 20)
        constexpr size_t lds_bytes = sizeof(float) * SCfg::LDS_FLOATS;
        const int tiles_x = g.Wo / SCfg::TW, tiles_y = (g.Ho + SCfg::TH - 1) / SCfg::TH, n_tiles = tiles_x * tiles_y;
        const int co_tiles = (g.Cout + 15) / 16;
        const long long pairs = (long long)co_tiles * ci_groups * n_samples;
        int strips = (int)((target + pairs - 1) / pairs);
        strips = strips < 1 ? 1 : (strips > n_tiles ? n_tiles : strips);
        if (strips > part.max_strips) strips = part.max_strips;
        const int tpb = (n_tiles + strips - 1) / strips;
        strips = (n_tiles + tpb - 1) / tpb;
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bww_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (attr != hipSuccess) return (int)attr;
        mfvi_launch(conv_bww_split_kernel, dim3(strips * co_tiles * ci_groups * n_samples), dim3(512), lds_bytes, st, in, gy, g, part.base,
                           part.stride, tiles_x, n_tiles, tpb, ci_groups, strips, co_tiles * ci_groups, n_samples);
        if (strips_used) *strips_used = strips;
        return (int)hipGetLastError();
    }
    const int nt = spec ? 512 : wfield * 64;
#define LAUNCH(KS_, S_, NB_, NT_, SP_)                                                                                            \
    {                                                                                                                          \
        using Cfg = WCfg<KS_, S_, NB_, NT_>;                                                                                   \
        constexpr size_t lds_bytes = sizeof(float) * Cfg::LDS_FLOATS;                                                          \
        if (lds_bytes > 150 * 1024) return -3;                                                                                 \
        if (SP_ && !vec_ok) return -3;                                         /* specialised producers stage aligned float4 */ \
        const int tiles_x = (g.Wo + Cfg::TW - 1) / Cfg::TW, tiles_y = (g.Ho + Cfg::TH - 1) / Cfg::TH;                          \
        const int n_tiles = tiles_x * tiles_y;                                                                                 \
        const int co_tiles = (g.Cout + 15) / 16, ci_groups = (g.Cin + Cfg::CIB - 1) / Cfg::CIB;                                \
        if (forced && NB_ > 1 && ci_groups == 1 && 16 * (NB_ - 1) >= g.Cin) return -3;   /* an input tile would be empty */           \
        const long long pairs = (long long)co_tiles * ci_groups * n_samples;                                                   \
        int strips = (int)((target + pairs - 1) / pairs);                                                                      \
        strips = strips < 1 ? 1 : (strips > n_tiles ? n_tiles : strips);                                                       \
        if (strips > part.max_strips) strips = part.max_strips;                                                                \
        const int tpb = (n_tiles + strips - 1) / strips;                                                                       \
        strips = (n_tiles + tpb - 1) / tpb;                                                                                    \
        auto kern = conv_bww_mfma_kernel<KS_, S_, NB_, NT_, SP_>;                                                                  \
        if (lds_bytes > 64 * 1024) {                                                                                           \
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
            if (attr != hipSuccess) return (int)attr;                                                                          \
        }                                                                                                                      \
        mfvi_launch(kern, dim3(strips * co_tiles * ci_groups * n_samples), dim3(NT_), lds_bytes, st, in, gy, g, part.base, \
                           part.stride, tiles_x, n_tiles, tpb, ci_groups, strips, co_tiles * ci_groups, n_samples);            \
        if (strips_used) *strips_used = strips;                                                                                \
        return (int)hipGetLastError();                                                                                         \
    }
#define LAUNCH_NB(KS_, S_)                                                                                                     \
    {                                                                                                                          \
        if (nt == 256) { if (nb == 1) LAUNCH(KS_, S_, 1, 256, false) if (nb == 2) LAUNCH(KS_, S_, 2, 256, false) if (nb == 3) LAUNCH(KS_, S_, 3, 256, false) } \
        if (nt == 512 && !spec) { if (nb == 1) LAUNCH(KS_, S_, 1, 512, false) if (nb == 2) LAUNCH(KS_, S_, 2, 512, false) if (nb == 3) LAUNCH(KS_, S_, 3, 512, false) } \
        if (nt == 512 && spec) { if (nb == 1) LAUNCH(KS_, S_, 1, 512, true) if (nb == 2) LAUNCH(KS_, S_, 2, 512, true) if (nb == 3) LAUNCH(KS_, S_, 3, 512, true) } \
        return -3;                                                                                                             \
    }
    if (target < 256) return -3;
    if (g.ks == 3 && g.stride == 1) LAUNCH_NB(3, 1)
    if (g.ks == 3 && g.stride == 2) LAUNCH_NB(3, 2)
    if (g.ks == 1 && g.stride == 1) LAUNCH_NB(1, 1)
    // 5x5 layers (inpainting nets): 25 accumulator fragments per 16-channel input tile, so one input tile per block
    if (g.ks == 5 && nb == 1) {
        if (g.stride == 1) { if (nt == 256) LAUNCH(5, 1, 1, 256, false) if (nt == 512 && !spec) LAUNCH(5, 1, 1, 512, false) if (nt == 512 && spec) LAUNCH(5, 1, 1, 512, true) }
        if (g.stride == 2) { if (nt == 256) LAUNCH(5, 2, 1, 256, false) if (nt == 512 && !spec) LAUNCH(5, 2, 1, 512, false) if (nt == 512 && spec) LAUNCH(5, 2, 1, 512, true) }
        return -3;
    }
    if (g.ks == 5) return -3;
#undef LAUNCH_NB
#undef LAUNCH
    return -2;
}
