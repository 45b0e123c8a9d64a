// K2a'' — the 32 (+4) -> 16 form of the bf16x6 backward-data kernel with the fold (conv_bwd_x6.hip): ONE pass per strip, weights resident in
// registers for the whole block.  Reference op as there: autograd of BayTorch/modules/reparam_layers.py:37 behind ReflectionPad2d(1)
// (models/common.py:100-135), then LeakyReLU' and the BatchNorm-backward sums of the layer's input (models/common.py:77-97) — the
// 36 -> 16 @256^2 layer at the top of skip()'s up path (models/skip.py:100-119), BASELINE.json's dominant kernel.
//
// What the pass structure of conv_bwd_x6.hip costs on this shape (profiles/NOTES.md R4.10): a strip of the 36 -> 16 layer is three passes
// (fragments of 16, 16 and 4-padded-to-16 input channels), each with its weight copy global -> LDS -> registers, two block barriers, a
// tile dump and 90 window reads per matrix wave; timing-only builds put the 4-channel fragment at a quarter of its cost for 4 %, one pass
// fewer for 13 %, one pass per strip for 30 %.  Here:
//   * a matrix wave = (full fragment f in {0, 1}, pixel half): it holds W_f (72 registers) and the 4-channel operand (24) from the block's
//     first instruction to its last — no weight traffic after the prologue, no LDS for weights;
//   * a strip is ONE sweep of the window rows: every (window row, kx) operand triple of the wave's two 16-pixel fragments meets the three tap
//     rows of three output rows (18 matrix instructions per 6 window reads); then a second, short sweep for the last 4 input channels of ONE
//     of the two pixel fragments: N = (tap row ky, channel c) — the three tap rows of a window row in ONE instruction triple, 12 of 16 columns
//     used — whose result columns are shifted onto their output rows with two DPP adds per register (3 matrix instructions per (row, kx)
//     instead of 9);
//   * two block barriers per strip: (D) the matrix waves are done with the window and the staging waves with the previous tiles, then the
//     tile dump (matrix waves) and the window's SR new rows (staging waves) go to LDS side by side, (E) both are published;
//   * the staging waves' strip is one software pipeline on the in-order vmcnt: dy rows of the next strip in two batches of three raw
//     register sets, between them the folds of the previous strip's two 16-channel tiles and of the 4-channel tile, the raw x of the fold
//     rolling through eight float4 registers (item j of the next tile is requested when item j of the current one has been consumed).
// Arithmetic, window layout, reflection adjoint, fold and BN-backward sums as in conv_bwd_x6.hip (three bf16 pieces per operand, K = 32 =
// [piece a | piece b] of the 16 output channels, three instructions per product-sum).
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_bf(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void split8(const float (&e)[8], u32x4& h, u32x4& m, u32x4& l)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) { unsigned hh, mm, ll; split_pair_bf16x3(e[2 * i], e[2 * i + 1], hh, mm, ll); h[i] = hh; m[i] = mm; l[i] = ll; }
}
// wave-uniform base + 32-bit BYTE offset per lane (conv_bwd_x6.hip: no 64-bit per-lane addresses in registers)
__device__ __forceinline__ float ldg_f(const float* base, unsigned boff) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + boff); }
__device__ __forceinline__ float4 ldg_f4(const float* base, unsigned boff) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + boff); }
__device__ __forceinline__ void stg_f4(float* base, unsigned boff, float4 v) { *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + boff) = v; }
template <int CTRL> __device__ __forceinline__ float dpp_mov0(float v)      // lanes whose source is outside the 16-lane row read 0
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum16(float v)      // total of a 16-lane row in its lane 15
{
    v += dpp_mov0<0x111>(v); v += dpp_mov0<0x112>(v); v += dpp_mov0<0x114>(v); v += dpp_mov0<0x118>(v);      // row_shr:1, 2, 4, 8
    return v;
}

struct BwdC { float qc, c1, k3, pad; };                   // dy = ga * c1 + (y * qc + k3)

#ifdef X6S_PROF
// dev build: per-phase s_memtime sums of matrix wave 0 / staging wave 0 of every block (scripts/dev/bwdx6s_prof.py)
__device__ unsigned long long g_x6s_prof[24];
__device__ __forceinline__ unsigned long long xs_now() { unsigned long long t; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define XS_T(v) const unsigned long long v = xs_now()
#define XS_ACC(slot, d) prof[slot] += (d)
#else
#define XS_T(v)
#define XS_ACC(slot, d)
#endif

constexpr int SR = 8, NOCTT = 2, PLANE = 80 * 16, PIECE = NOCTT * PLANE, ROWB = 3 * PIECE, NR = SR + 2, RING = NR * ROWB;
constexpr int OPB = SR * 256 + 16;                       // channel pitch of a finished tile [ch][SR][64] floats
constexpr int NT = 32;                                   // channels of the two full tiles

struct X6SArgs {
    GView gin; TView xin; ConvGeom g;
    const unsigned* wsp; long long wsp_stride_u4; int rem_off_u4;      // split pieces: per-sample stride and offset of the 4-channel units, in 16-byte units
    float* fga; long long fga_sstride; double* fbsums;
    int bands, strips, tpb, nx, nz;
};

template <bool REM>
__global__ __launch_bounds__(512, 2) void conv_bwd_x6s_kernel(X6SArgs A)
{
    extern __shared__ __align__(16) char lds[];             // ring [NR][3][2][80][16] | tiles [32][OPB] | 4-channel tiles [2][4][OPB] | tables
    char* const s_out = lds + RING;
    char* const s_rem = s_out + NT * OPB;
    const ConvGeom& g = A.g;
    const int CI = g.Cin, H = g.H, W = g.W, HW = H * W;
    constexpr int CO = 16, NFS = 48;
    BwdC* const s_chb = reinterpret_cast<BwdC*>(s_rem + 2 * 4 * OPB);               // [16]  (the 4-channel tile is double-buffered: its rows go to LDS as they complete)
    ChanFwd* const s_ch = reinterpret_cast<ChanFwd*>(s_chb + CO);                   // [48]
    float* const s_sum = reinterpret_cast<float*>(s_ch + NFS);                      // [32][2] | 4-channel tile: [4 channels][4 row groups][2]
    float* const s_sumr = s_sum + NT * 2;

    const int tid = threadIdx.x;
    const bool producer = tid >= 256;
    const int t = tid & 255, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
#ifdef X6S_PROF
    unsigned long long prof[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    XS_T(t_entry);
#endif
    int bx, by, k;
    xcd_decode(blockIdx.x, A.nx, 1, A.nz, bx, by, k);
    const int band = bx % A.bands, strip0 = (bx / A.bands) * A.tpb;
    const int n_strip = min(A.tpb, A.strips - strip0);
    const int c0 = band * 64;
    const bool fuse_sums = A.fbsums != nullptr;
    const u32x4* __restrict__ wsp = reinterpret_cast<const u32x4*>(A.wsp) + (long long)k * A.wsp_stride_u4;

    // ---- staging helpers (all eight waves stage the first window; afterwards only waves 4-7 use them); as conv_bwd_x6.hip ----
    const float* __restrict__ gsrc = A.gin.ga + (long long)k * A.gin.gstride;
#ifdef X6S_DBG_NOY      // timing only
    const float* __restrict__ ysrc = nullptr;
#define X6S_YSEL 1
#else
    // no BatchNorm behind the layer (never the case in skip()): the y loads still run, on ga, against qc = 0 — every load of the staging waves is
    // unconditional, so that the compiler can count the memory operations in flight (see the staging waves' comment)
    const float* __restrict__ ysrc = A.gin.stats ? A.gin.y + (long long)k * A.gin.ystride : gsrc;
#endif
#ifdef X6S_YSEL
#define YLD(o) 0.f
#else
#define YLD(o) ldg_f(ysrc, (o))
#endif
    const bool lb = c0 == 0, rb = c0 + 64 == W;
    // the SR new rows of a strip are 16 tasks (row j, octet q) of 64 pixels, four per staging wave (task i of wave w: number w + 4 i), plus one
    // task of the four special pixel slots of this wave's four (row, octet) pairs on lanes 0..15 (pair = lane >> 2, kind = lane & 3)
    const int pair = lane >> 2, kind = lane & 3;
    u32x4 pc[5][3];
    int jrow[5], qoct[5];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int tn = wv + 4 * i; jrow[i] = tn / NOCTT; qoct[i] = tn % NOCTT; }
    { const int tn = wv + 4 * min(pair, 3); jrow[4] = tn / NOCTT; qoct[4] = tn % NOCTT; }
    const bool sp_on = lane < 16 && (kind == 0 ? !lb : kind == 1 ? !rb : kind == 2 ? lb : rb);
    const int colA = kind == 0 ? max(c0 - 1, 0) : kind == 1 ? min(c0 + 64, W - 1) : kind == 2 ? 2 : W - 3;
    const int colB = kind == 2 ? 0 : W - 1;
    const bool two = kind >= 2;
    const int sslot = kind == 0 ? 0 : kind == 1 ? 65 : kind == 2 ? 66 : 67;
    const unsigned uHW = (unsigned)HW;
    auto fetch_to = [&](int i, float (&ga)[8], float (&yy)[8], int Rb) {
        const int R = min(max(Rb + jrow[i], 0), H - 1);
        const unsigned off = 4u * ((unsigned)(8 * qoct[i]) * uHW + (unsigned)(R * W + c0) + (unsigned)lane);
#pragma unroll
        for (int j = 0; j < 8; ++j) { ga[j] = ldg_f(gsrc, off + 4u * (unsigned)j * uHW); yy[j] = YLD(off + 4u * (unsigned)j * uHW); }
    };
    auto fetch_sp_to = [&](float (&ga)[8], float (&ya)[8], float (&gb)[8], float (&yb)[8], int Rb) {
        const int R = min(max(Rb + jrow[4], 0), H - 1);
        const unsigned base = (unsigned)(8 * qoct[4]) * uHW + (unsigned)(R * W);
        const unsigned oa = 4u * (base + (unsigned)colA), ob = 4u * (base + (unsigned)colB);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ga[j] = ldg_f(gsrc, oa + 4u * (unsigned)j * uHW); ya[j] = YLD(oa + 4u * (unsigned)j * uHW);
            gb[j] = ldg_f(gsrc, ob + 4u * (unsigned)j * uHW); yb[j] = YLD(ob + 4u * (unsigned)j * uHW);
        }
    };
    auto finish_from = [&](int i, const float (&ga)[8], const float (&yy)[8], int Rb) {
        const int R = Rb + jrow[i];
        float e[8];
        int qi = __builtin_amdgcn_readfirstlane(8 * qoct[i]); asm volatile("" : "+s"(qi));      // opaque: the table reads are not hoisted over the strip loop
#pragma unroll
        for (int j = 0; j < 8; ++j) { const BwdC b = s_chb[qi + j]; e[j] = __builtin_fmaf(ga[j], b.c1, __builtin_fmaf(yy[j], b.qc, b.k3)); }
        if (R < 0 || R >= H) {
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = 0.f;
        }
        split8(e, pc[i][0], pc[i][1], pc[i][2]);
    };
    auto finish_sp_from = [&](const float (&ga)[8], const float (&ya)[8], const float (&gb)[8], const float (&yb)[8], int Rb) {
        const int R = Rb + jrow[4];
        float e[8];
        int qi = 8 * qoct[4]; asm volatile("" : "+v"(qi));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const BwdC b = s_chb[qi + j];
            const float a = __builtin_fmaf(ga[j], b.c1, __builtin_fmaf(ya[j], b.qc, b.k3));
            const float c = __builtin_fmaf(gb[j], b.c1, __builtin_fmaf(yb[j], b.qc, b.k3));
            e[j] = two ? a + c : a;
        }
        if (!sp_on || R < 0 || R >= H) {
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = 0.f;
        }
        split8(e, pc[4][0], pc[4][1], pc[4][2]);
    };
    // the five tasks' pieces into the ring rows of image rows Rb .. Rb + nvalid - 1 (ring slot of image row R: (R + 1) mod NR)
    auto write = [&](int Rb, int nvalid) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (jrow[i] >= nvalid) continue;
            if (i == 4 && lane >= 16) continue;
            const int slot = (Rb + jrow[i] + 1 + NR) % NR;
            char* d = lds + slot * ROWB + qoct[i] * PLANE + (i == 4 ? sslot : 1 + lane) * 16;
            *reinterpret_cast<u32x4*>(d) = pc[i][0]; *reinterpret_cast<u32x4*>(d + PIECE) = pc[i][1]; *reinterpret_cast<u32x4*>(d + 2 * PIECE) = pc[i][2];
        }
    };

    // Prologue: the first strip's window (image rows r0 - 1 .. r0 + SR) by ALL eight waves — waves 4-7 its first SR rows, waves 0-3 the last
    // two — every global load of it (and the matrix waves' weight operands) issued before anything waits, the channel tables included
    const int pro_rd = producer ? 0 : 1;
    const int pro_Rb = strip0 * SR - 1 + pro_rd * SR, pro_nvalid = pro_rd == 0 ? SR : 2;
    float pg[6][8], py[6][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) if (pro_rd == 0 || jrow[i] < 2) fetch_to(i, pg[i], py[i], pro_Rb);
    fetch_sp_to(pg[4], py[4], pg[5], py[5], pro_Rb);
    // matrix wave = (fragment f, pixel half): its operands for the whole block
    const int mf = wv & 1, half = wv >> 1;
    u32x4 Wr[9][2], Wm[3][2];
    if (!producer) {
        const u32x4* __restrict__ wb = wsp + mf * (9 * 2 * 64) + lane;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int v = 0; v < 2; ++v) Wr[tp][v] = wb[(tp * 2 + v) * 64];
        if constexpr (REM) {
            const u32x4* __restrict__ wr = wsp + A.rem_off_u4 + lane;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int v = 0; v < 2; ++v) Wm[kx][v] = wr[(kx * 2 + v) * 64];
        }
        for (int c = t; c < CO; c += 256) { const ChanBwd b = chan_bwd(A.gin, k, c); BwdC r; r.qc = -b.c1 * b.c3 * b.rstd; r.c1 = b.c1; r.k3 = __builtin_fmaf(-b.mean, r.qc, -b.c1 * b.c2); r.pad = 0.f; s_chb[c] = r; }
        for (int c = t; c < NFS; c += 256) { ChanFwd f; if (fuse_sums) f = chan_fwd(A.xin, k, min(c, CI - 1)); else { f.mean = 0.f; f.scale = 1.f; f.beta = 0.f; f.rstd = 1.f; } s_ch[c] = f; }
        for (int i = t; i < NT * 2 + 32; i += 256) s_sum[i] = 0.f;
    }
    __syncthreads();                                        // (S0) channel tables visible
#pragma unroll
    for (int i = 0; i < 4; ++i) if (pro_rd == 0 || jrow[i] < 2) finish_from(i, pg[i], py[i], pro_Rb);
    finish_sp_from(pg[4], py[4], pg[5], py[5], pro_Rb);
    write(pro_Rb, pro_nvalid);

    if (producer) {
        // ======================= staging waves =======================
        // Everything below is STRAIGHT-LINE per (fold variant, first / middle / last strip): the compiler's s_waitcnt insertion counts the
        // younger memory operations of the path it is on, and any run-time branch around a load or store ("if (more)", "if (pend)", the
        // activation variant) makes it wait for vmcnt(0) at the next consumer — the fold of a strip then waited for the dy rows requested
        // right in front of it, one exposed memory round trip per strip (timing-only build without the raw-x loads: 113 -> 87 us).
        const bool xact = (A.xin.act & 1) != 0; const float xslope = A.xin.slope;
        const float* __restrict__ xq = A.xin.data + (long long)k * A.xin.sstride;
        float* __restrict__ gout = A.fga + (long long)k * A.fga_sstride;
        const int fchl = t >> 4, fv = t & 15;               // full tiles: thread = (channel of the fragment, float4 column), SR items = tile rows
        const int rg = lane >> 4;                           // 4-channel tile: wave = channel 32 + wv, lane = (row group, float4 column), rows rg and rg + 4
        auto run = [&](auto sums_c, auto act_c) {
            constexpr bool SUMS = decltype(sums_c)::value;
            float ga_[3][8], y_[3][8];
            float4 xpre[SR] = {};                           // raw x of the fold, rolling: item j of the next tile is requested when item j of the current one is consumed
            // byte offset of (channel ch, image row r0, this thread's float4 column)
            auto goff = [&](int ch, int r0) { return 4u * ((unsigned)ch * uHW + (unsigned)(r0 * W + c0 + 4 * fv)); };
            auto xissue_full = [&](int f, int r0, int j) {
#ifndef X6S_DBG_NOX
                if constexpr (SUMS) xpre[j] = ldg_f4(xq, goff(16 * f + fchl, r0) + 4u * (unsigned)(j * W));
#endif
            };
            auto xissue_rem = [&](int r0, int j) {
#ifndef X6S_DBG_NOX
                if constexpr (SUMS) xpre[j] = ldg_f4(xq, goff(32 + wv, r0) + 4u * (unsigned)((rg + 4 * j) * W));
#endif
            };
            auto fold_item = [&](float4 d4, float4 x4, const ChanFwd& cf, float& fs, float& fx) {
                float dd[4] = {d4.x, d4.y, d4.z, d4.w};
                if constexpr (SUMS) {
                    const float yy[4] = {x4.x, x4.y, x4.z, x4.w};
                    float ymv[4];
#pragma unroll
                    for (int l = 0; l < 4; ++l) ymv[l] = yy[l] - cf.mean;
                    if (xact) {     // wave-uniform branch around vector arithmetic ONLY (no memory operation inside: the waitcnt bookkeeping stays exact)
#pragma unroll
                        for (int l = 0; l < 4; ++l) { const float vv = __builtin_fmaf(ymv[l], cf.scale, cf.beta); dd[l] *= (vv > 0.f) ? 1.f : xslope; }
                    }
#pragma unroll
                    for (int l = 0; l < 4; ++l) { fs += dd[l]; fx = __builtin_fmaf(dd[l], ymv[l], fx); }      // (views without an activation — the concat tensors of the skip() nets — three operations per element)
                }
                return make_float4(dd[0], dd[1], dd[2], dd[3]);
            };
            // fold of full tile f of the strip at image row r0; NEXT: 1 = request tile f + 1's items behind each consumed one, 2 = the 4-channel tile's two
            auto fold_full = [&](int r0, int f, auto next_c) {
                constexpr int NEXT = decltype(next_c)::value;
                r0 = __builtin_amdgcn_readfirstlane(r0); asm volatile("" : "+s"(r0));
                const int ch = 16 * f + fchl;
                const unsigned off = goff(ch, r0);
                const char* so = s_out + ch * OPB + fv * 16;
                const ChanFwd cf = s_ch[ch];
                float fs = 0.f, fx = 0.f;
#pragma unroll
                for (int j0 = 0; j0 < SR; j0 += 4) {
                    float4 d4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) d4[j] = *reinterpret_cast<const float4*>(so + (j0 + j) * 256);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int j = j0 + jj;
                        const float4 o = fold_item(d4[jj], xpre[j], cf, fs, fx);
#ifdef X6S_DBG_NOFOLDST  // timing only
                        if (o.x == 1.2345f)
#endif
                        stg_f4(gout, off + 4u * (unsigned)(j * W), o);
                        if constexpr (NEXT == 1) xissue_full(f + 1, r0, j);
                        if constexpr (REM && NEXT == 2) { if (j < 2) xissue_rem(r0, j); }
                    }
                }
                if constexpr (SUMS) {
                    fs = row_sum16(fs); fx = row_sum16(fx);
                    if (fv == 15) { float* sp = s_sum + ch * 2; sp[0] += fs; sp[1] += fx; }      // one owner per channel: no atomics
                }
            };
            auto fold_rem = [&](int r0, int buf) {
                r0 = __builtin_amdgcn_readfirstlane(r0); asm volatile("" : "+s"(r0));
                const int ch = 32 + wv;
                const unsigned off = goff(ch, r0);
                const char* so = s_rem + (buf * 4 + wv) * OPB + fv * 16;
                const ChanFwd cf = s_ch[ch];
                float fs = 0.f, fx = 0.f;
                float4 d4[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) d4[j] = *reinterpret_cast<const float4*>(so + (rg + 4 * j) * 256);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 o = fold_item(d4[j], xpre[j], cf, fs, fx);
#ifdef X6S_DBG_NOFOLDST
                    if (o.x == 1.2345f)
#endif
                    stg_f4(gout, off + 4u * (unsigned)((rg + 4 * j) * W), o);
                }
                if constexpr (SUMS) {
                    fs = row_sum16(fs); fx = row_sum16(fx);
                    if (fv == 15) { float* sp = s_sumr + (wv * 4 + rg) * 2; sp[0] += fs; sp[1] += fx; }
                }
            };
            using N0 = std::integral_constant<int, 0>; using N1 = std::integral_constant<int, 1>; using N2 = std::integral_constant<int, 2>;
            // one strip of the staging waves.  PEND: the previous strip's tiles wait in s_out / s_rem (its first tile's raw x is on its way);
            // MORE: a next strip exists (its SR new window rows are staged).  Requests and consumers alternate so that nothing waits for a
            // load younger than the one it needs (vmcnt retires in order)
            auto period = [&](int ts, auto pend_c, auto more_c) {
                constexpr bool PEND = decltype(pend_c)::value, MORE = decltype(more_c)::value;
                const int r0 = (strip0 + ts) * SR, pr0 = r0 - SR;
                int Rb = r0 + SR + 1;                       // first NEW image row of the next strip's window
                Rb = __builtin_amdgcn_readfirstlane(Rb); asm volatile("" : "+s"(Rb));
                XS_T(s0);
                if constexpr (MORE) { fetch_to(0, ga_[0], y_[0], Rb); fetch_to(1, ga_[1], y_[1], Rb); fetch_to(2, ga_[2], y_[2], Rb); }
                XS_T(s1); XS_ACC(11, s1 - s0);
                if constexpr (PEND) fold_full(pr0, 0, N1{});
                XS_T(s2); XS_ACC(12, s2 - s1);
                if constexpr (MORE) {
                    finish_from(0, ga_[0], y_[0], Rb); finish_from(1, ga_[1], y_[1], Rb); finish_from(2, ga_[2], y_[2], Rb);
                    fetch_to(3, ga_[0], y_[0], Rb); fetch_sp_to(ga_[1], y_[1], ga_[2], y_[2], Rb);
                }
                XS_T(s3); XS_ACC(13, s3 - s2);
                if constexpr (PEND) { if constexpr (REM) fold_full(pr0, 1, N2{}); else fold_full(pr0, 1, N0{}); }
                XS_T(s4); XS_ACC(14, s4 - s3);
                if constexpr (MORE) { finish_from(3, ga_[0], y_[0], Rb); finish_sp_from(ga_[1], y_[1], ga_[2], y_[2], Rb); }
                XS_T(s5); XS_ACC(15, s5 - s4);
                if constexpr (REM && PEND) fold_rem(pr0, (ts + 1) & 1);
                XS_T(s6); XS_ACC(16, s6 - s5);
                // raw x of the first tile of THIS strip: consumed at the head of the next period
#pragma unroll
                for (int j = 0; j < SR; ++j) xissue_full(0, r0, j);
                XS_T(s7); XS_ACC(17, s7 - s6);
                lds_barrier();                              // (D) the matrix waves are done with the window; the previous tiles are folded
                XS_T(s8); XS_ACC(18, s8 - s7);
                if constexpr (MORE) write(Rb, SR);
                XS_T(s9); XS_ACC(19, s9 - s8);
                lds_barrier();                              // (E) tiles of this strip and the next window published
                XS_T(s10); XS_ACC(20, s10 - s9);
            };
            XS_T(sp0); XS_ACC(10, sp0 - t_entry);
            lds_barrier();                                  // (B1) window of strip 0 published
            if (n_strip == 1) period(0, std::false_type{}, std::false_type{});
            else {
                period(0, std::false_type{}, std::true_type{});
#pragma unroll 1
                for (int ts = 1; ts + 1 < n_strip; ++ts) period(ts, std::true_type{}, std::true_type{});
                period(n_strip - 1, std::true_type{}, std::false_type{});
            }
            XS_T(sf0);
            { const int pr0 = (strip0 + n_strip - 1) * SR;
              fold_full(pr0, 0, N1{});
              if constexpr (REM) { fold_full(pr0, 1, N2{}); fold_rem(pr0, (n_strip + 1) & 1); } else fold_full(pr0, 1, N0{}); }
#ifdef X6S_PROF
            { XS_T(sf1); XS_ACC(21, sf1 - sf0); if (t == 0) for (int i = 10; i < 24; ++i) atomicAdd(&g_x6s_prof[i], prof[i]); }
#endif
        };
        // ONE variant: the raw x is loaded and the sums are formed even when the input carries no BatchNorm (never on this path in skip():
        // identity constants, the sums are not written out); LeakyReLU' is a wave-uniform branch around vector arithmetic only
        run(std::true_type{}, std::true_type{});
    } else {
        // ======================= matrix waves: wave = (fragment mf, pixel half) =======================
        const bool rbd = c0 + 64 == W;
        const int pfA = 2 * half, pfB = pfA + 1, pfR = pfA + mf;
        int axA[3], axB[3];                                 // byte offsets of this lane's pixel operand inside a ring row, per kx ([x_h | x_h]; + PIECE: [x_m | x_m])
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            int sa = 16 * pfA + l15 + 2 - kx, sb = 16 * pfB + l15 + 2 - kx;
            if (lb && pfA == 0 && l15 == 1 && kx == 0) sa = 66;
            if (rbd && pfB == 3 && l15 == 14 && kx == 2) sb = 67;
            axA[kx] = (l4 & 1) * PLANE + sa * 16; axB[kx] = (l4 & 1) * PLANE + sb * 16;
        }
        const int lsel = l4 >= 2 ? 2 * PIECE : 0;           // [x_h | x_l]: the upper k-octets read the low piece
        char* const soA = s_out + (16 * mf + l15) * OPB + (16 * pfA + 4 * l4) * 4;      // register r of acc[o] = pixel 16 pf + 4 l4 + r of channel l15, tile row o
        char* const soB = soA + 64;
        char* const sor = s_rem + (l15 & 3) * OPB + (16 * pfR + 4 * l4) * 4;
        constexpr int QX[3] = {2, 1, 0}, QW[3] = {1, 0, 0};      // [x_h|x_l].[w_l|w_h], [x_m|x_m].[w_h|w_m], [x_h|x_h].[w_h|w_m]: small terms first
        lds_barrier();                                      // (B1)
        XS_T(mp0); XS_ACC(0, mp0 - t_entry);
#pragma unroll 1
        for (int ts = 0; ts < n_strip; ++ts) {
            XS_T(m0);
            const int r0 = (strip0 + ts) * SR;
            const bool first = r0 == 0, last = r0 + SR == H;
            int ro[SR + 2];                                 // ring row offsets of the window rows (image rows r0 - 1 + ii)
            { const int b0 = r0 % NR;
#pragma unroll
              for (int ii = 0; ii < SR + 2; ++ii) { int s = b0 + ii; s = s >= NR ? s - NR : s; ro[ii] = s * ROWB; } }
            {
                f32x4 accA[SR], accB[SR];
#pragma unroll
                for (int o = 0; o < SR; ++o) { accA[o] = (f32x4){0.f, 0.f, 0.f, 0.f}; accB[o] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                constexpr int NGRP = (SR + 2) * 3;
                u32x4 XA[2][3], XB[2][3];
                auto issue = [&](int gi, u32x4 (&xa)[3], u32x4 (&xb)[3]) {
                    const int ii = gi / 3, kx = gi - 3 * ii;
                    const char* pa = lds + ro[ii] + axA[kx]; const char* pb = lds + ro[ii] + axB[kx];
                    xa[0] = *reinterpret_cast<const u32x4*>(pa); xa[1] = *reinterpret_cast<const u32x4*>(pa + PIECE); xa[2] = *reinterpret_cast<const u32x4*>(pa + lsel);
                    xb[0] = *reinterpret_cast<const u32x4*>(pb); xb[1] = *reinterpret_cast<const u32x4*>(pb + PIECE); xb[2] = *reinterpret_cast<const u32x4*>(pb + lsel);
                };
                issue(0, XA[0], XB[0]);
#pragma unroll
                for (int gi = 0; gi < NGRP; ++gi) {
                    const int ii = gi / 3, kx = gi - 3 * ii;
                    __builtin_amdgcn_sched_barrier(0);
                    if (gi + 1 < NGRP) issue(gi + 1, XA[(gi + 1) & 1], XB[(gi + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    const u32x4 (&xa)[3] = XA[gi & 1]; const u32x4 (&xb)[3] = XB[gi & 1];
                    // image row r0 - 1 + ii meets output row o = ii + ky - 2 through tap row ky; consecutive matrix instructions go to different accumulators
#pragma unroll
                    for (int q = 0; q < 3; ++q)
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) {
                            const int o = ii + ky - 2;
                            if (o >= 0 && o < SR) {
                                accA[o] = mfma_bf(xa[QX[q]], Wr[ky * 3 + kx][QW[q]], accA[o]);
                                accB[o] = mfma_bf(xb[QX[q]], Wr[ky * 3 + kx][QW[q]], accB[o]);
                            }
                        }
                    if (ii == 1 && first) {                 // image row 0 -> padded row -1 -> row 1 (tap row 0)
#pragma unroll
                        for (int q = 0; q < 3; ++q) { accA[1] = mfma_bf(xa[QX[q]], Wr[kx][QW[q]], accA[1]); accB[1] = mfma_bf(xb[QX[q]], Wr[kx][QW[q]], accB[1]); }
                    }
                    if (ii == SR && last) {                 // image row H-1 -> padded row H -> row H-2 (tap row 2)
#pragma unroll
                        for (int q = 0; q < 3; ++q) { accA[SR - 2] = mfma_bf(xa[QX[q]], Wr[6 + kx][QW[q]], accA[SR - 2]); accB[SR - 2] = mfma_bf(xb[QX[q]], Wr[6 + kx][QW[q]], accB[SR - 2]); }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                XS_T(m1); XS_ACC(1, m1 - m0);
                if constexpr (!REM) lds_barrier();          // (D)
                else {
                    // The last 4 input channels of pixel fragment pfR: N = (ky, c).  Column group ky of window row ii belongs to output row ii + ky - 2:
                    // the result is shifted down the 16-lane rows by 4 ky lanes (row_shl) onto lanes 0..3 and added there.  An output row is complete
                    // when window row o + 2 has passed and goes to this strip's half of the double-buffered 4-channel tile at once (three rows live);
                    // the full tiles wait in their registers for barrier (D).
                    f32x4 racc[SR];
#pragma unroll
                    for (int o = 0; o < SR; ++o) racc[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    int axR[3];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) axR[kx] = mf ? axB[kx] : axA[kx];
                    char* const sorb = sor + (ts & 1) * 4 * OPB;
#pragma unroll
                    for (int i2 = 0; i2 < SR + 2; i2 += 2) {      // two window rows at a time: two independent accumulators
                        f32x4 ap[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            u32x4 x0[3], x1[3];
                            const char* p0 = lds + ro[i2] + axR[kx]; const char* p1 = lds + ro[i2 + 1] + axR[kx];
                            x0[0] = *reinterpret_cast<const u32x4*>(p0); x0[1] = *reinterpret_cast<const u32x4*>(p0 + PIECE); x0[2] = *reinterpret_cast<const u32x4*>(p0 + lsel);
                            x1[0] = *reinterpret_cast<const u32x4*>(p1); x1[1] = *reinterpret_cast<const u32x4*>(p1 + PIECE); x1[2] = *reinterpret_cast<const u32x4*>(p1 + lsel);
#pragma unroll
                            for (int q = 0; q < 3; ++q) { ap[0] = mfma_bf(x0[QX[q]], Wm[kx][QW[q]], ap[0]); ap[1] = mfma_bf(x1[QX[q]], Wm[kx][QW[q]], ap[1]); }
                        }
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const int ii = i2 + h2;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float v = ap[h2][r];
                                if (ii - 2 >= 0 && ii - 2 < SR) racc[ii - 2][r] += v;                          // ky = 0: lanes 0..3 as they are
                                if (ii - 1 >= 0 && ii - 1 < SR) racc[ii - 1][r] += dpp_mov0<0x104>(v);         // ky = 1: row_shl:4
                                if (ii >= 0 && ii < SR) racc[ii][r] += dpp_mov0<0x108>(v);                     // ky = 2: row_shl:8
                                if (ii == 1 && first) racc[1][r] += v;                                         // image row 0, tap row 0 -> row 1 as well
                                if (ii == SR && last) racc[SR - 2][r] += dpp_mov0<0x108>(v);                   // image row H-1, tap row 2 -> row H-2 as well
                            }
                            if (ii - 2 >= 0 && ii - 2 < SR && l15 < 4) *reinterpret_cast<f32x4*>(sorb + (ii - 2) * 256) = racc[ii - 2];
                        }
                    }
                    XS_T(m2); XS_ACC(2, m2 - m1);
                    lds_barrier();                          // (D)
                    XS_T(m2b); XS_ACC(3, m2b - m2);
                }
                XS_T(m3);
#pragma unroll
                for (int o = 0; o < SR; ++o) { *reinterpret_cast<f32x4*>(soA + o * 256) = accA[o]; *reinterpret_cast<f32x4*>(soB + o * 256) = accB[o]; }
                XS_T(m4); XS_ACC(4, m4 - m3);
            }
            XS_T(m4b);
            lds_barrier();                                  // (E)
            XS_T(m5); XS_ACC(5, m5 - m4b); XS_ACC(6, 1);
        }
#ifdef X6S_PROF
        if (t == 0) { XS_T(mz); prof[7] = mz - t_entry; prof[8] = 1; for (int i = 0; i < 10; ++i) atomicAdd(&g_x6s_prof[i], prof[i]); }
#endif
    }
    if (fuse_sums) {
        __syncthreads();                                    // (Z)
        for (int i = tid; i < CI * 2; i += 512) {
            const int q = i >> 1, which = i & 1;
            float v;
            if (q < NT) v = s_sum[q * 2 + which];
            else { const float* sp = s_sumr + (q - NT) * 8 + which; v = (sp[0] + sp[2]) + (sp[4] + sp[6]); }
            if (which) v *= s_ch[q].rstd;
            atomicAdd(A.fbsums + ((long long)k * CI + q) * 2 + which, (double)v);
        }
    }
}

}  // namespace

bool x6s_shape_ok(const ConvGeom& g)
{
    return g.ks == 3 && g.stride == 1 && !(g.W & 63) && g.H >= 8 && !(g.H & 7) && g.Cout == 16 && (g.Cin == 32 || g.Cin == 36) && !(g.w_off & 3);
}

// tune: strips per block.  The caller (launch_conv_bwd_data_x6) has checked the fold's pointers and split this layer's weights.
int launch_conv_bwd_data_x6s(const GView& gy, const ConvGeom& g, const unsigned* wsp, long long wsp_stride_u4, int rem_off_u4, int T, int n_samples,
                             hipStream_t st, const FoldFuse& fuse)
{
    if (!x6s_shape_ok(g)) return -3;
    X6SArgs A{};
    A.gin = gy; A.xin = fuse.x; A.g = g;
    A.wsp = wsp; A.wsp_stride_u4 = wsp_stride_u4; A.rem_off_u4 = rem_off_u4;
    A.fga = fuse.ga; A.fga_sstride = fuse.ga_sstride; A.fbsums = fuse.bsums;
    A.bands = g.W / 64; A.strips = g.H / SR; A.tpb = max(1, T);
    A.nx = A.bands * ((A.strips + A.tpb - 1) / A.tpb); A.nz = n_samples;
    const size_t lds_bytes = (size_t)RING + (NT + 8) * OPB + sizeof(BwdC) * 16 + sizeof(ChanFwd) * 48 + sizeof(float) * (NT * 2 + 32);
    static const hipError_t a0 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bwd_x6s_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    static const hipError_t a1 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bwd_x6s_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (a0 != hipSuccess || a1 != hipSuccess) return (int)(a0 != hipSuccess ? a0 : a1);
    if (lds_bytes > 160 * 1024) return -3;
    mfvi_tl_family = 3;
    if (g.Cin == 36) mfvi_launch(conv_bwd_x6s_kernel<true>, dim3(A.nx * A.nz), dim3(512), lds_bytes, st, A);
    else mfvi_launch(conv_bwd_x6s_kernel<false>, dim3(A.nx * A.nz), dim3(512), lds_bytes, st, A);
    return (int)hipGetLastError();
}

#ifdef X6S_PROF
extern "C" int mfvi_debug_x6s_prof(unsigned long long* out24, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_x6s_prof), sizeof(unsigned long long) * 24);
    if (e == hipSuccess && reset) { unsigned long long z[24] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_x6s_prof), z, sizeof(z)); }
    return (int)e;
}
#endif
