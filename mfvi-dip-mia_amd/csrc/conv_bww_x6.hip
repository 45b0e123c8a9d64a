// K2b for the 3x3 stride-1 layers on maps >= 64 wide, on the BF16 matrix cores at fp32 accuracy ("bf16x6", round 3).
//
//   dW[co][ci][ky][kx] = sum_pix dy[co][r][c] * xpad[ci][r + ky][c + kx]      (autograd of BayTorch/modules/reparam_layers.py:37 behind the
//   ReflectionPad2d(1) of models/common.py:100-135), the same GEMM as conv_bww_mfma.hip with the PIXEL index as reduction dimension.
//
// gfx950 runs v_mfma_f32_16x16x4_f32 at 1/16 of the rate of v_mfma_f32_16x16x32_bf16.  Every fp32 operand is the exact sum of three bf16
// pieces, a = a_h + a_m + a_l (8 significand bits each, formed by the staging waves with v_cvt_pk_bf16_f32 and two subtractions), and
//   a * b = a_h b_h + a_h b_m + a_m b_h + a_m b_m + a_h b_l + a_l b_h  (+ terms below 2^-23 |a b|, dropped)
// is six bf16 matrix instructions with fp32 accumulation in place of eight fp32 ones: 96 instead of 256 matrix cycles for the same
// 16 x 16 x 32 block of multiply-adds.  scripts/micro/bf16x6.hip (profiles/r03_bf16x6_micro.txt): error against fp64 3.7e-7 sum|a b|
// at K = 4096 (the fp32 instruction: 3.5e-7), 370-400 TFLOP/s fp32-equivalent, two VALU fillers per matrix instruction for free.
//
// Layout.  K = 32 consecutive pixels of one image row: lane (g = lane >> 4, n = lane & 15) supplies pixels 8 g .. 8 g + 7 of channel n, one
// ds_read_b128 from a bf16 row in LDS.  M = 16 output channels (A = dy pieces), N = 16 input channels (B = x pieces).  The three kx taps of
// an x row are the same eight pixels shifted by one element: the aligned read plus the dword before and after it, and 4 + 4 v_alignbit_b32.
// The three ky taps pair padded x row i with dy rows i, i - 1, i - 2.  So one (x row, 32-pixel segment, 16 x 16 channel pair) costs
// 9 taps x 6 = 54 matrix instructions from 9 + 9 LDS reads and 24 VALU instructions.  The 4 extra channels of the 16 n + 4 concat layers
// ride in one more fragment whose 16 columns are (kx slot, channel): per-lane shift, 3 x 6 instructions instead of 54.
// A seventh-of-a-percent extra: dy pieces against a constant-one fragment give the bias gradient.
//
// Block (512 threads, one per CU: 105-145 KB of LDS): a 64-pixel-wide band x a strip of padded rows x (16 COF output channels) x (up to 36
// input channels).  A stage = two padded x rows: waves 4-7 stage them (global float4 -> deferred BN / LeakyReLU resp. BN-backward -> three
// bf16 pieces -> ds_write_b64) one stage ahead into a double buffer, and the two new dy rows into a six-row ring; waves 0-3 issue the
// matrix instructions: wave w owns output fragment w % COF and the (row, segment) units w / COF of the stage, accumulates ALL of the
// block's (input fragment, tap) pairs for them, and the 4 / COF pixel slices are summed through LDS at the end.  One raw s_barrier per stage.
// The block's partial dW goes to its (band x strip, sample) slab with plain stores, as the other backward-weight kernels' (grad_finalize).
#include "common.h"
#include <type_traits>
#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma_bf(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// four floats -> three packed bf16 quads (h, m, l), a = h + m + l exactly (common.h, split_pair_bf16x3)
__device__ __forceinline__ void split4(const float (&e)[4], u32x2& h, u32x2& m, u32x2& l)
{
#ifdef X6_DBG_NOSPLIT
    h.x = __float_as_uint(e[0]); h.y = __float_as_uint(e[1]); m.x = __float_as_uint(e[2]); m.y = __float_as_uint(e[3]); l = h; return;
#endif
    unsigned h0, m0, l0, h1, m1, l1;
    split_pair_bf16x3(e[0], e[1], h0, m0, l0);
    split_pair_bf16x3(e[2], e[3], h1, m1, l1);
    h = (u32x2){h0, h1}; m = (u32x2){m0, m1}; l = (u32x2){l0, l1};
}

// BW = band width in pixels: 64 (stage = 2 padded rows x 2 segments of 32 pixels) or 32 (maps 32 wide: stage = 4 padded rows x 1 segment)
template <int BW>
struct X6Cfg {
    static constexpr int CIB = 36;                       // input channels per block: two fragments + the 4-channel remainder
    static constexpr int R = 128 / BW;                   // padded x rows per stage (4 units of 32 pixels)
    static constexpr int QX = BW / 4 + 2, QD = BW / 4;   // staged quads per x row (one halo quad each side) / dy row
    static constexpr int XROW = (BW + 16) * 2;           // bytes of one (channel, row, piece): image column c0 + t at element t + 8
    static constexpr int XCH = R * 3 * XROW + 32;        // 992 / 1184: (pitch / 16) mod 16 in {2, 6, 10, 14}, see the static_assert
    static constexpr int XBUF = CIB * XCH;
    static constexpr int DROW = BW * 2;
    static constexpr int DSLOT = 3 * DROW;
    static constexpr int NSLOT = 2 * R + 2;              // dy ring: rows i - 2 .. i + R - 1 in use, R being written
    static constexpr int DCH = NSLOT * DSLOT + 32;       // 2336 / 1952
    static constexpr int ROWP = CIB * 9 + 2;             // epilogue: floats per output channel row
    // A ds_read_b128 is served in four groups of 16 lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 — i.e. eight
    // channels of pixel chunk g with the OTHER eight channels of chunk g + 1 (16 bytes further).  With a channel pitch of 16 * m bytes the
    // group is conflict-free for m mod 16 in {2, 6, 10, 14}; the odd pitches a contiguous-lane grouping would ask for cost 4-20 extra
    // cycles per read here (SQ_LDS_BANK_CONFLICT was 49 % of SQ_LDS_IDX_ACTIVE with m = 13).
    static_assert((XCH / 16) % 4 == 2 && (DCH / 16) % 4 == 2 && XCH % 16 == 0 && DCH % 16 == 0, "plane pitches");
};

struct X6Args {
    TView in; GView gy; ConvGeom g;
    float* part; long long part_stride;
    int bands, rps;                                      // bands per row; padded rows per strip (a multiple of the stage's rows)
    int ci_groups, nx, ny, nz;
};

struct X6Bwd { float mean, qc, c1, k2; };                // dy = (y - mean) * qc + (ga * c1 + k2)   (conv_rp.hip's RpBwd)

template <int COF, int BW>
__global__ __launch_bounds__(512, 2) void conv_bww_x6_kernel(X6Args A)
{
    using C = X6Cfg<BW>;
    constexpr int CIB = C::CIB, XCH = C::XCH, XBUF = C::XBUF, DCH = C::DCH, DSLOT = C::DSLOT, ROWP = C::ROWP;
    constexpr int R = C::R, QX = C::QX, QD = C::QD, NSLOT = C::NSLOT, XI = R * QX;
    constexpr int COB = 16 * COF, NSL = 4 / COF;         // output channels per block, pixel slices
    extern __shared__ __align__(16) char lds[];          // [2][CIB][XCH] x pieces | [COB][DCH] dy ring; the epilogue's [NSL][COB][ROWP] floats over both
    __shared__ ChanFwd s_chx[CIB];
    __shared__ X6Bwd s_chg[COB];
    __shared__ float s_db[NSL][COB];
    char* const s_x = lds;
    char* const s_dy = lds + 2 * XBUF;

    const ConvGeom& g = A.g;
    const int tid = threadIdx.x;
    const bool producer = tid >= 256;
    const int t = tid & 255, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    int bx, by, k;
    xcd_decode(blockIdx.x, A.nx, A.ny, A.nz, bx, by, k);
    const int co0 = (by / A.ci_groups) * COB, ci0 = (by % A.ci_groups) * 32;
    const int Cin = g.Cin, Cout = g.Cout, H = g.H, W = g.W, HW = H * W;
    const int cot = min(COB, Cout - co0), cit = min(CIB, Cin - ci0);       // launcher: cit in {16, 20, 32, 36}
    const int nfull = cit >> 4; const bool x4 = (cit & 15) != 0;
    const bool do_bias = (ci0 == 0) && (g.b_off >= 0);
    const int band = bx % A.bands, strip = bx / A.bands;
    const int c0 = band * BW;
    // padded x rows of this block; rows past H + 1 (R = 4: H + 2 is not a multiple of it) meet dy rows >= H only, which are staged as zeros
    const int i0 = strip * A.rps, i1 = min((H + 2 + R - 1) / R * R, i0 + A.rps);
    const int nst = (i1 - i0) / R;                                          // stages (launcher: every strip holds >= 1)

    const float* __restrict__ xin = A.in.data + (long long)k * A.in.sstride;
    const float* __restrict__ gap = A.gy.ga + (long long)k * A.gy.gstride;
    const float* __restrict__ yp = (A.gy.stats && A.gy.y) ? A.gy.y + (long long)k * A.gy.ystride : nullptr;

    if (!producer) {
        // channel tables (channels beyond the tensor: constants that make the staged value zero)
        for (int c = t; c < CIB; c += 256) {
            ChanFwd f = chan_fwd(A.in, k, min(ci0 + c, Cin - 1));
            if (c >= cit) { f.scale = 0.f; f.beta = 0.f; }
            s_chx[c] = f;
        }
        for (int c = t; c < COB; c += 256) {
            const ChanBwd b = chan_bwd(A.gy, k, min(co0 + c, Cout - 1));
            X6Bwd r; r.mean = b.mean; r.qc = -b.c1 * b.c3 * b.rstd; r.c1 = b.c1; r.k2 = -b.c1 * b.c2;
            if (c >= cot) { r.qc = 0.f; r.c1 = 0.f; r.k2 = 0.f; }
            s_chg[c] = r;
        }
    }

    if (producer) {
        // ======================= staging waves =======================
#ifndef X6_PRODPRIO
#define X6_PRODPRIO 2
#endif
        __builtin_amdgcn_s_setprio(X6_PRODPRIO);
        // x items: (channel c, row r of the stage, quad v = 0 .. QX - 1 at image columns c0 - 4 + 4 v), XI per channel; fixed per thread
        constexpr int NXJ = (CIB * XI + 255) / 256;       // 6
        constexpr int NGJ = (COB * 32) / 256;             // 2 COF: (channel, row, quad v = 0 .. QD - 1 at columns c0 + 4 v), R QD = 32 per channel
        int xgo[NXJ]; int xlo[NXJ]; unsigned xr = 0, xfl = 0, xok = 0;      // xr / gr: two bits per item = its row of the stage
        const int n_xi = cit * XI;
#pragma unroll
        for (int j = 0; j < NXJ; ++j) {
            const int idx = t + 256 * j, q = min(idx, n_xi - 1);
            const int c = q / XI, rem = q - c * XI, r = rem / QX, v = rem - r * QX;
            int gx = c0 - 4 + 4 * v; unsigned fl = 0;
            if (gx < 0) { fl = 1; gx = 0; } else if (gx >= W) { fl = 2; gx = W - 4; }
            xgo[j] = (ci0 + c) * HW + gx;
            xlo[j] = c * XCH + r * (3 * C::XROW) + 8 + 8 * v;
            xr |= (unsigned)r << (2 * j); xfl |= fl << (2 * j); if (idx < n_xi) xok |= 1u << j;
        }
        int ggo[NGJ]; int glo[NGJ]; unsigned gr = 0;
#pragma unroll
        for (int j = 0; j < NGJ; ++j) {
            const int idx = t + 256 * j, c = idx >> 5, r = (idx & 31) / QD, v = idx % QD;
            ggo[j] = (co0 + min(c, cot - 1)) * HW + c0 + 4 * v;
            glo[j] = c * DCH + 8 * v;
            gr |= (unsigned)r << (2 * j);
        }
        // two register sets: the loads of stage s + 2 are in flight while stage s + 1 is transformed and stored (one stage of prefetch left the
        // loop waiting on HBM latency: 37 KB in flight per CU)
        struct Regs { float4 xv[NXJ], gv[NGJ], yv[NGJ]; };
        Regs R0, R1;
        auto pick = [&](const int (&ro)[R], unsigned r) { int v = ro[0]; if (R > 1) v = r == 1 ? ro[1] : v; if (R > 2) { v = r == 2 ? ro[2 % R] : v; v = r == 3 ? ro[3 % R] : v; } return v; };
        auto fetch_x = [&](Regs& Rg, int i) {              // padded rows i .. i + R - 1 = image rows reflect(i - 1 ..)
            int ro[R];
#pragma unroll
            for (int r = 0; r < R; ++r) ro[r] = reflect_idx(i - 1 + r, H) * W;
#pragma unroll
            for (int j = 0; j < NXJ; ++j) {
#if defined(X6_DBG_NOLOAD) || defined(X6_DBG_NOPROD)
                Rg.xv[j] = make_float4(1.f, 2.f, 3.f, 4.f + ro[0]); continue;
#endif
                Rg.xv[j] = *reinterpret_cast<const float4*>(xin + xgo[j] + pick(ro, (xr >> (2 * j)) & 3u));
            }
        };
        auto fetch_dy = [&](Regs& Rg, int r0) {            // dy rows r0 .. r0 + R - 1 (rows outside the image: any valid address, zeroed at the store)
            int ro[R];
#pragma unroll
            for (int r = 0; r < R; ++r) ro[r] = min(max(r0 + r, 0), H - 1) * W;
#pragma unroll
            for (int j = 0; j < NGJ; ++j) {
                const int off = ggo[j] + pick(ro, (gr >> (2 * j)) & 3u);
#if defined(X6_DBG_NOLOAD) || defined(X6_DBG_NOPROD)
                Rg.gv[j] = make_float4(1.f, 2.f, 3.f, 4.f + off); Rg.yv[j] = Rg.gv[j]; continue;
#endif
                Rg.gv[j] = *reinterpret_cast<const float4*>(gap + off);
                if (yp) Rg.yv[j] = *reinterpret_cast<const float4*>(yp + off);
            }
        };
        // every fetch is issued unconditionally (stages past the strip re-read its last rows and are never stored): the compiler can then
        // count the loads in flight and wait for the older register set only
        const int ilast = i0 + R * (nst - 1);
        auto fetch = [&](Regs& Rg, int s) { const int i = min(i0 + R * s, ilast); fetch_dy(Rg, i); fetch_x(Rg, i); };
        Regs RP;
        fetch_dy(RP, i0 - R); fetch(R0, 0); fetch(R1, 1);
        __syncthreads();                                   // (S0) channel tables visible
        float xm[NXJ], xs[NXJ], xb[NXJ];
#pragma unroll
        for (int j = 0; j < NXJ; ++j) { const int c = min(t + 256 * j, n_xi - 1) / XI; const ChanFwd f = s_chx[c]; xm[j] = f.mean; xs[j] = f.scale; xb[j] = f.beta; }
        X6Bwd gk[NGJ];
#pragma unroll
        for (int j = 0; j < NGJ; ++j) gk[j] = s_chg[(t + 256 * j) >> 5];
        const bool xlrelu = (A.in.act & 1) != 0; const float xslope = A.in.slope;
        auto store_x = [&](const Regs& Rg, char* __restrict__ dst) {
#ifdef X6_DBG_NOPROD
            return;
#endif
#pragma unroll
            for (int j = 0; j < NXJ; ++j) {
                float e[4] = {Rg.xv[j].x, Rg.xv[j].y, Rg.xv[j].z, Rg.xv[j].w};
#pragma unroll
                for (int l = 0; l < 4; ++l) { float v = __builtin_fmaf(e[l] - xm[j], xs[j], xb[j]); if (xlrelu) v = __builtin_fmaxf(v, v * xslope); e[l] = v; }
                const unsigned fl = (xfl >> (2 * j)) & 3u;
                e[3] = fl == 1 ? e[1] : e[3];              // column -1 <- x[1]
                e[0] = fl == 2 ? e[2] : e[0];              // column W  <- x[W-2]
                u32x2 h, m, l; split4(e, h, m, l);
                if ((xok >> j) & 1u) {
                    char* d = dst + xlo[j];
                    *reinterpret_cast<u32x2*>(d) = h; *reinterpret_cast<u32x2*>(d + C::XROW) = m; *reinterpret_cast<u32x2*>(d + 2 * C::XROW) = l;
                }
            }
        };
        auto store_dy = [&](const Regs& Rg, int r0) {      // rows r0 .. r0 + R - 1 -> ring slots (row + 2) % NSLOT
            int so[R]; bool any_inv = false; unsigned invm = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) { so[r] = ((r0 + r + 2 + 2 * NSLOT) % NSLOT) * DSLOT; const bool iv = r0 + r < 0 || r0 + r >= H; any_inv |= iv; invm |= (iv ? 1u : 0u) << r; }
#ifdef X6_DBG_NOPROD
            return;
#endif
#pragma unroll
            for (int j = 0; j < NGJ; ++j) {
                float e[4] = {Rg.gv[j].x, Rg.gv[j].y, Rg.gv[j].z, Rg.gv[j].w};
                if (yp) {
                    const float yy[4] = {Rg.yv[j].x, Rg.yv[j].y, Rg.yv[j].z, Rg.yv[j].w};
#pragma unroll
                    for (int l = 0; l < 4; ++l) e[l] = __builtin_fmaf(yy[l] - gk[j].mean, gk[j].qc, __builtin_fmaf(e[l], gk[j].c1, gk[j].k2));
                } else {
#pragma unroll
                    for (int l = 0; l < 4; ++l) e[l] *= gk[j].c1;      // no BatchNorm behind the layer: dy = ga (c1 = 1; 0 for channels beyond the tensor)
                }
                const unsigned rj = (gr >> (2 * j)) & 3u;
                if (any_inv) { if ((invm >> rj) & 1u) { e[0] = 0.f; e[1] = 0.f; e[2] = 0.f; e[3] = 0.f; } }
                u32x2 h, m, l; split4(e, h, m, l);
                char* d = s_dy + glo[j] + pick(so, rj);
                *reinterpret_cast<u32x2*>(d) = h; *reinterpret_cast<u32x2*>(d + C::DROW) = m; *reinterpret_cast<u32x2*>(d + 2 * C::DROW) = l;
            }
        };
        auto store = [&](const Regs& Rg, int s) { store_dy(Rg, i0 + R * s); store_x(Rg, s_x + (s & 1) * XBUF); };
        // prologue: the dy rows above the strip (i0 - 2, i0 - 1 are read), then stage 0 (dy rows and padded x rows i0 .. i0 + R - 1)
        store_dy(RP, i0 - R); store(R0, 0);
        fetch(R0, 2);
        lds_barrier();                                     // (A) stage 0 published
        for (int st = 0; st < nst; st += 2) {              // in flight at the top: R1 = stage st + 1 (older), R0 = stage st + 2
            if (st + 1 < nst) store(R1, st + 1);
            fetch(R1, st + 3);
            lds_barrier();
            if (st + 1 >= nst) break;
            if (st + 2 < nst) store(R0, st + 2);
            fetch(R0, st + 4);
            lds_barrier();
        }
    } else {
        // ======================= matrix waves =======================
        const int cf = COF == 1 ? 0 : (wv & 1), sl = COF == 1 ? wv : (wv >> 1);      // output fragment, pixel slice
        const char* const dyb = s_dy + (cf * 16 + l15) * DCH + l4 * 16;
        const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#ifdef X6_CONSPRIO
        __builtin_amdgcn_s_setprio(X6_CONSPRIO);
#endif
        __syncthreads();                                   // (S0)
        lds_barrier();                                     // (A)
        auto consume = [&](auto nf_c, auto x4_c, auto bias_c) {
            constexpr int NF = decltype(nf_c)::value; constexpr bool X4 = decltype(x4_c)::value, BIAS = decltype(bias_c)::value;
            f32x4 acc[NF][9], accx[X4 ? 3 : 1], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int q = 0; q < 9; ++q) acc[f][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < (X4 ? 3 : 1); ++q) accx[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int kxs = l15 >> 2;                                       // remainder fragment: column = (kx slot, channel l15 & 3)
            const char* const xfb = s_x + l15 * XCH + 16 + l4 * 16;         // full fragments: + f * 16 * XCH
            const char* const xrb = s_x + (NF * 16 + (l15 & 3)) * XCH + 16 + l4 * 16;
            // one unit: padded x row (row r of the stage), 32-pixel segment s; bs = ring slot of dy row i + r
            // A stage's work for this wave is a list of groups (x fragment f, x piece pb): prep = one ds_read_b128 + two ds_read_b32 and the
            // v_alignbit shifts that make the three kx operands; run = the matrix instructions of that x piece against the dy pieces it
            // meets (h: l, m, h; m: m, h; l: h -> 27 / 18 / 9, remainder fragment 9 / 6 / 3).  One wave per SIMD issues matrix instructions,
            // in order: whatever it waits for, the matrix pipe waits too.  So the list is software-pipelined by hand — the reads of group
            // g + 1 are issued in front of the first half of group g's matrix instructions, its shifts sit between the two halves — and
            // pinned with sched_barrier (left to the compiler, every read and shift of an iteration was hoisted in front of its first
            // matrix instruction: 4360 cycles per stage for 2064 of matrix work).
            constexpr int G = 3 * NF + (X4 ? 3 : 0);
            const unsigned sh = (kxs & 1) ? 0u : 16u;                       // remainder fragment, kx slot 1 (and the unused slot 3): the aligned read
            auto issue = [&](int gi, int off_x, int seg, unsigned (&r)[6]) {
                const int f = gi / 3, pb = gi - 3 * f;
                const char* xp = (f < NF ? xfb + f * 16 * XCH : xrb) + off_x + pb * C::XROW + seg * 64;
                const u32x4 d = *reinterpret_cast<const u32x4*>(xp);
                r[0] = *reinterpret_cast<const unsigned*>(xp - 4); r[1] = d.x; r[2] = d.y; r[3] = d.z; r[4] = d.w; r[5] = *reinterpret_cast<const unsigned*>(xp + 16);
            };
            auto prep = [&](int gi, const unsigned (&r)[6], u32x4 (&B)[3]) {
                if (gi / 3 < NF) {
                    B[0] = (u32x4){__builtin_amdgcn_alignbit(r[1], r[0], 16), __builtin_amdgcn_alignbit(r[2], r[1], 16), __builtin_amdgcn_alignbit(r[3], r[2], 16), __builtin_amdgcn_alignbit(r[4], r[3], 16)};
                    B[1] = (u32x4){r[1], r[2], r[3], r[4]};
                    B[2] = (u32x4){__builtin_amdgcn_alignbit(r[2], r[1], 16), __builtin_amdgcn_alignbit(r[3], r[2], 16), __builtin_amdgcn_alignbit(r[4], r[3], 16), __builtin_amdgcn_alignbit(r[5], r[4], 16)};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned lo = kxs == 0 ? r[j] : r[j + 1], hi = kxs == 2 ? r[j + 2] : r[j + 1];
                        B[0][j] = __builtin_amdgcn_alignbit(hi, lo, sh);
                    }
                }
            };
            auto run = [&](int gi, const u32x4 (&Aop)[3][3], const u32x4 (&B)[3], int half) {
                const int f = gi / 3, pb = gi - 3 * f;
                const int n = (3 - pb) * (f < NF ? 9 : 3), mid = (n + 1) / 2;
                int idx = 0;
#pragma unroll
                for (int pa = 2 - pb; pa >= 0; --pa)                        // pieces of dy that meet x piece pb
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        if (f < NF) {
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) { if ((idx < mid) == (half == 0)) acc[f][ky * 3 + kx] = mfma_bf(Aop[ky][pa], B[kx], acc[f][ky * 3 + kx]); ++idx; }
                        } else { if ((idx < mid) == (half == 0)) accx[ky] = mfma_bf(Aop[ky][pa], B[0], accx[ky]); ++idx; }
                    }
            };
            auto loadA = [&](u32x4 (&Aop)[3][3], int seg, int bs) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    int slot = bs - ky; slot += slot < 0 ? NSLOT : 0;
#pragma unroll
                    for (int p = 0; p < 3; ++p) Aop[ky][p] = *reinterpret_cast<const u32x4*>(dyb + slot * DSLOT + p * C::DROW + seg * 64);
                }
            };
            // NU units (BW = 64: one padded row, segments seg0 .. ; BW = 32: consecutive rows) as one pipelined list of NU * G groups
            auto units = [&](auto nu_c, int off_x0, int seg0, int bs0) {
                constexpr int NU = decltype(nu_c)::value;
                auto u_off = [&](int u) { return BW == 64 ? off_x0 : off_x0 + u * (3 * C::XROW); };
                auto u_seg = [&](int u) { return BW == 64 ? seg0 + u : seg0; };
                auto u_bs = [&](int u) { int b = BW == 64 ? bs0 : bs0 + u; b -= b >= NSLOT ? NSLOT : 0; return b; };
#ifdef X6_DBG_NOMFMA
                return;
#endif
                u32x4 Aop[NU][3][3], B[2][3]; unsigned raw[2][6];
                loadA(Aop[0], u_seg(0), u_bs(0));
                issue(0, u_off(0), u_seg(0), raw[0]);
                prep(0, raw[0], B[0]);
#pragma unroll
                for (int q = 0; q < NU * G; ++q) {
                    const int u = q / G, gi = q - u * G;
                    const int qn = q + 1, un = qn / G, gn = qn - un * G;
                    __builtin_amdgcn_sched_barrier(0);
                    if (qn < NU * G) issue(gn, u_off(un), u_seg(un), raw[qn & 1]);
                    if (NU > 1 && gi == G - 1 && u + 1 < NU) loadA(Aop[(u + 1) % NU], u_seg(u + 1), u_bs(u + 1));      // the next unit's dy pieces, under this unit's last group
                    __builtin_amdgcn_sched_barrier(0);
                    if (BIAS && gi == 0) {
#pragma unroll
                        for (int p = 2; p >= 0; --p) accb = mfma_bf(Aop[u][0][p], ones, accb);
                    }
                    run(gi, Aop[u], B[q & 1], 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (qn < NU * G) prep(gn, raw[qn & 1], B[qn & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    run(gi, Aop[u], B[q & 1], 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            constexpr std::integral_constant<int, 1> one_unit{}; constexpr std::integral_constant<int, 2> two_units{};
            int bs = (i0 + 2) % NSLOT;                                      // ring slot of dy row i0
            for (int st = 0; st < nst; ++st) {
                const int xb_off = (st & 1) * XBUF;
                if constexpr (COF == 1) {
                    const int r = BW == 64 ? sl >> 1 : sl, s = BW == 64 ? sl & 1 : 0;
                    int b = bs + r; b -= b >= NSLOT ? NSLOT : 0;
                    units(one_unit, xb_off + r * (3 * C::XROW), s, b);
                } else {
                    const int r = BW == 64 ? sl : 2 * sl;
                    int b = bs + r; b -= b >= NSLOT ? NSLOT : 0;
                    units(two_units, xb_off + r * (3 * C::XROW), 0, b);
                }
                bs += R; bs -= bs >= NSLOT ? NSLOT : 0;
                lds_barrier();
            }
            // ---- pixel slices -> LDS (over the staging buffers: every wave is past its last read) ----
            __syncthreads();                                                // (E1)
            float* s_ep = reinterpret_cast<float*>(lds);
            float* row = s_ep + (sl * COB + cf * 16 + l4 * 4) * ROWP;
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int q = 0; q < 9; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) row[r * ROWP + (f * 16 + l15) * 9 + q] = acc[f][q][r];
            if constexpr (X4) {
                if (kxs < 3)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int r = 0; r < 4; ++r) row[r * ROWP + (NF * 16 + (l15 & 3)) * 9 + ky * 3 + kxs] = accx[ky][r];
            }
            if (BIAS && l15 == 0)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_db[sl][cf * 16 + l4 * 4 + r] = accb[r];
        };
        constexpr std::true_type yes{}; constexpr std::false_type no{};
        constexpr std::integral_constant<int, 1> n1{}; constexpr std::integral_constant<int, 2> n2{};
        if (nfull == 2) {
            if (x4) { if (do_bias) consume(n2, yes, yes); else consume(n2, yes, no); }
            else { if (do_bias) consume(n2, no, yes); else consume(n2, no, no); }
        } else {
            if (x4) { if (do_bias) consume(n1, yes, yes); else consume(n1, yes, no); }
            else { if (do_bias) consume(n1, no, yes); else consume(n1, no, no); }
        }
    }
    if (producer) __syncthreads();                                          // (E1)
    __syncthreads();                                                        // (E2) slices in LDS
    const float* s_ep = reinterpret_cast<const float*>(lds);
    const int len = cit * 9;
    float* __restrict__ o = A.part + ((long long)bx * A.nz + k) * A.part_stride;
    for (int idx = tid; idx < COB * len; idx += 512) {
        const int r = idx / len, rel = idx - r * len, co = co0 + r;
        if (co < Cout) {
            const float* e = s_ep + r * ROWP + rel;
            float v = e[0];
#pragma unroll
            for (int s = 1; s < NSL; ++s) v += e[s * COB * ROWP];
            o[((long long)co * Cin + ci0) * 9 + rel] = v;
        }
    }
    if (do_bias && tid < cot) {
        float v = s_db[0][tid];
#pragma unroll
        for (int s = 1; s < NSL; ++s) v += s_db[s][tid];
        o[(long long)Cout * Cin * 9 + co0 + tid] = v;
    }
}

template <int COF, int BW>
int launch_x6k(X6Args& A, dim3 grid, hipStream_t st)
{
    using C = X6Cfg<BW>;
    constexpr size_t stage_bytes = 2 * (size_t)C::XBUF + (size_t)16 * COF * C::DCH, epi_bytes = sizeof(float) * (size_t)(4 / COF) * 16 * COF * C::ROWP;
    constexpr size_t lds_bytes = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
    static_assert(lds_bytes <= 150 * 1024, "LDS budget");
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bww_x6_kernel<COF, BW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (attr != hipSuccess) return (int)attr;
    mfvi_tl_family = 3;
    mfvi_launch((conv_bww_x6_kernel<COF, BW>), grid, dim3(512), lds_bytes, st, A);
    return (int)hipGetLastError();
}

}  // namespace

// tune: cof | 11 << 8 | (target blocks / 256) << 16.  Returns -2 when the shape is not served, -3 when the tiling is not valid for it.
int launch_conv_bwd_weight_x6(const TView& in, const GView& gy, const ConvGeom& g, BwwPart part, int* strips_used, int cof, int target,
                              int n_samples, hipStream_t st)
{
    if (g.ks != 3 || g.stride != 1 || (g.W & 31) || (g.H & 1) || g.H < 4 || g.Cin < 16) return -2;
    const int bw = (g.W & 63) ? 32 : 64, R = 128 / bw;
    const int rem = g.Cin & 15;
    if (rem != 0 && rem != 4) return -2;
    if ((in.sstride & 3) || ((uintptr_t)in.data & 15) || (gy.gstride & 3) || ((uintptr_t)gy.ga & 15) || (gy.y && ((gy.ystride & 3) || ((uintptr_t)gy.y & 15)))) return -2;
    if (in.act & MFVI_ACT_SQUARE) return -2;
    if ((long long)max(g.Cin, g.Cout) * g.H * g.W >= (1LL << 31)) return -2;
    if (cof != 1 && cof != 2) return -3;
    if (target < 256) return -3;
    const int cob = 16 * cof;
    const int ci_groups = (g.Cin % 32 == 4 || g.Cin % 32 == 0) ? g.Cin / 32 : g.Cin / 32 + 1;       // 36 -> 1, 68 -> 2, 132 -> 4, 48 -> 2 (32 + 16), 52 -> 2 (32 + 20)
    const int co_tiles = (g.Cout + cob - 1) / cob;
    const int bands = g.W / bw, pr2 = (g.H + 2 + R - 1) / R;               // stages over the whole map
    const long long pairs = (long long)co_tiles * ci_groups * n_samples * bands;
    int strips = (int)((target + pairs - 1) / pairs);
    strips = strips < 1 ? 1 : (strips > pr2 ? pr2 : strips);
    if (strips * bands > part.max_strips) strips = part.max_strips / bands;
    if (strips < 1) return -3;
    const int spb = (pr2 + strips - 1) / strips;                            // stages per strip
    strips = (pr2 + spb - 1) / spb;
    X6Args A{};
    A.in = in; A.gy = gy; A.g = g; A.part = part.base; A.part_stride = part.stride;
    A.bands = bands; A.rps = R * spb; A.ci_groups = ci_groups;
    A.nx = strips * bands; A.ny = co_tiles * ci_groups; A.nz = n_samples;
    const dim3 grid(A.nx * A.ny * A.nz);
    int rc;
    if (bw == 64) rc = cof == 1 ? launch_x6k<1, 64>(A, grid, st) : launch_x6k<2, 64>(A, grid, st);
    else rc = cof == 1 ? launch_x6k<1, 32>(A, grid, st) : launch_x6k<2, 32>(A, grid, st);
    if (rc == 0 && strips_used) *strips_used = strips * bands;
    return rc;
}
