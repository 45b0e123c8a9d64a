// K1 / K2a' for the 3x3 stride-1 layers on maps >= 64 wide — "row-phase" operand layout (round 3).
//
// Same implicit GEMM on v_mfma_f32_16x16x4_f32 as conv_mfma.hip (fp32 in / fp32 accumulate, BayTorch/modules/reparam_layers.py:26-37 with
// the ReflectionPad2d(1) of models/common.py:100-135 in the index math), but the pixel operand is laid out so that ONE wide LDS read
// feeds twelve matrix instructions instead of one:
//
//   a pixel fragment's 16 columns (n = lane & 15) are NOT 16 consecutive pixels; lane n stands for the pixel QUAD 4n .. 4n+3 of a 64-pixel
//   row segment, and fragment j (the "phase", j = 0..3) holds pixel 4n + j.  Lane (k = lane >> 4, n) reads the six consecutive floats
//   win[4n .. 4n+5] of reduction channel k's window row (ds_read_b128 + ds_read_b64, window column 0 = image column x0 - 1); the operand of
//   (phase j, tap kx) is element j + kx of those six.  So 2 LDS instructions feed 4 phases x 3 taps = 12 MFMAs (conv_mfma.hip: 12 reads),
//   the weights of a (4-channel, ky) step are one ds_read_b128 per output fragment (layout [k][m][ky][4]), and a wave that owns R rows of
//   a 64-wide tile issues 36 * R * MF MFMAs per 4-channel step from 2 (R + 2) + 3 MF LDS reads.
//
//   The accumulator layout that falls out of it is store-friendly: register q of (fragment f, row r, phase j) is channel 16 f + 4 (lane >> 4) + q
//   at pixel 4 n + j, so the four phases of one register index ARE a float4 of one output row — the epilogue stores aligned float4 rows (16
//   lanes = 256 contiguous bytes per channel) straight from the accumulators; no LDS transpose, no hand-over to the producer waves.
//
// MODE 0 forward:       m = cout, k = cin; window = reflection-padded view(x) (deferred BN + LeakyReLU applied by the staging waves)
// MODE 1 backward-data: m = cin,  k = cout, flipped taps; window = zero-padded dy (BN-backward formed on load); the gradient is formed on the
//         UN-padded input domain with the adjoint of the reflection padding folded in, and the epilogue is the fold of the input tensor:
//         LeakyReLU'(view(x)), BN-backward sums, ga written once (what finalize_dx did as a separate launch in round 1).
//         Rows:    dx[1] = w'(0) * dy[0] + w'(1) * dy[1] + w'(2) * (dy[2] + dy[0])   (padded row -1 folds onto row 1), likewise
//                  dx[H-2] = w'(0) * (dy[H-3] + dy[H-1]) + ...: the staging wave of a channel adds the two rows in LDS into a spare window
//                  row and the one wave that owns image row 1 (H-2) reads that row in place of dy[2] (dy[H-3]) — a different LDS row
//                  offset, no extra MFMA, no branch in the matrix stream (rows 1 / H-2 are the last / first row of their wave for R <= 2).
//         Columns: one fma on the pixel operand of image columns 1 and W-2 (masks are zero elsewhere).
//   REM: the 4 + 16n channels of the skip() concats — the block that owns the last fragments carries the 4 extra output channels on
//         v_mfma_f32_4x4x1_16B_f32 against the pixel fragments already in registers (as conv_mfma.hip's REM).
// Block: 512 threads; waves 0-3 issue MFMAs (wave w owns rows w R .. w R + R - 1 of a (4 R) x 64 tile), waves 4-7 stage the next 4-channel
// window (producer wave p = channel p of the chunk: global float4 -> transform -> ds_write_b128) and the next weight chunk; one barrier per
// chunk.  Several tiles per block run through the same pipeline.
#include "common.h"
#include <type_traits>
#include <cstdio>
#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));      // 4 floats at dword alignment (window quads start at column x0 - 1)

struct RpArgs {
    TView xin; GView gin; ConvGeom g;
    const float* w; long long wstride;     // weights of sample k at w + k*wstride
    OutDesc out;                           // MODE 0
    float* fga; long long fga_sstride; double* fbsums;      // MODE 1: gradient wrt the BN output of the input tensor, its BN-backward sums (or nullptr)
    int tiles_x, n_tiles, tiles_per_block, interleave;      // interleave: block b takes tiles b, b + nx, b + 2 nx, ... (see launch_rp)
    int nx, ny, nz;                        // logical grid (tile groups, output-channel tiles, samples), launched 1-D
};

#ifdef RP_PROF
// dev build: per-phase s_memtime sums of consumer wave 0 / producer wave 0 of every block (scripts/dev/rp_prof.py)
__device__ unsigned long long g_rp_prof[16];
__device__ __forceinline__ unsigned long long rp_now() { unsigned long long t; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define RP_T(v) const unsigned long long v = rp_now()
#define RP_ACC(slot, d) prof[slot] += (d)
#else
#define RP_T(v)
#define RP_ACC(slot, d)
#endif

// BN-backward on load in three operations per element: dy = (y - mean) * qc + (ga * c1 + k2), qc = -c1 c3 rstd, k2 = -c1 c2
// (chan_bwd's  c1 * (ga - c2 - xhat * c3)  with the products of the channel constants formed once per block)
struct RpBwd { float mean, qc, c1, k2; };

// Narrow maps (WSH > 0: maps exactly 32 or 16 wide = 2^WSH float4 columns, WSH = 3 / 2): the 64 "pixels" of a strip are NSUB = 2 / 4 image rows
// side by side — lanes n = 0..7 the quads of image row y, lanes 8..15 those of image row y + TH (32 wide; 16 wide: four groups of four
// lanes, TH rows apart; the tile is NSUB * TH image rows tall) — so the vertical taps are +-1 window row for every group; a window row is
// NSUB x [W + 2 columns + 2 pad] = 18 / 20 quads, and a lane's six floats start at 4 (n + (n >> WSH)).  Every group carries both image borders.
template <int R, int MODE, int WSH = 0>
struct RpCfg {
    static constexpr bool W32 = WSH > 0;                    // (name kept from the first, 32-wide version: "narrow map")
    static constexpr int QPR = WSH > 0 ? (1 << WSH) : 16;   // float4 columns per image row inside the strip
    static constexpr int NSUB = 16 / QPR;                   // image rows side by side
    static constexpr int TH = 4 * R, WROWS = TH + 2, NQ = WSH > 0 ? NSUB * (QPR + 1) : 17, PITCH = 4 * NQ;       // 66 window columns -> 17 quads
    // MODE 1: two extra rows per channel plane hold the row part of the reflection adjoint (S1 = dy[2] + dy[0] for image row 1,
    // S2 = dy[H-3] + dy[H-1] for image row H-2; see the kernel's header)
    static constexpr int PROWS = WROWS + (MODE == 1 ? 2 : 0);
    static constexpr int PLANE = (PROWS * PITCH + 63) / 64 * 64;                    // == 0 (mod 64): the 16-lane groups of a ds_read_b128 hit 64 distinct banks
    static constexpr int NITEM = WROWS * NQ, NV = (NITEM + 63) / 64;
};

// MINW: minimum waves per SIMD the register allocation must allow (4 = 128 VGPRs = two 512-thread blocks per CU)
// KS: 4-channel k-steps per stage (one barrier per stage: a stage boundary costs ~800 cycles of matrix time, see NOTES)
template <int MODE, int MF, int R, bool REM, int KS, int MINW, int WSH = 0>
__global__ __launch_bounds__(512, MINW) void conv_rp_kernel(RpArgs A)
{
    static_assert(!REM || MODE == 1, "remainder channels: backward-data only");
    static_assert(MODE == 0 || R <= 2, "backward-data: image rows 1 / H-2 must be the last / first row of their wave");
    using Cfg = RpCfg<R, MODE, WSH>;
    constexpr bool W32 = WSH > 0; constexpr int QPR = Cfg::QPR, NSUB = Cfg::NSUB;
    constexpr int TH = Cfg::TH, PITCH = Cfg::PITCH, PLANE = Cfg::PLANE, NITEM = Cfg::NITEM, NV = Cfg::NV;
    constexpr int CT = 16 * MF, CTX = CT + (REM ? 4 : 0);
    constexpr int WFR = 4 * 16 * 12;                           // floats of one output fragment's weight chunk: [k 4][m 16][ky 3][4]
    constexpr int WCH1 = MF * WFR + (REM ? 4 * 4 * 12 : 0);    // + the 4 extra channels [k 4][m 4][ky 3][4]
    constexpr int WCH = KS * WCH1;                             // one stage

    extern __shared__ __align__(16) float s_dyn[];             // [2][WCH] weight chunks | ChanFwd[Cin] | ChanBwd[Cout]
    __shared__ __align__(16) float s_x[2][KS * 4 * PLANE];
    __shared__ float s_bias[CT];
    __shared__ double s_red[MODE == 0 ? 4 : 1][MODE == 0 ? CTX : 1][2];
    constexpr int OP = TH * 64;                                 // channel pitch of s_out
    __shared__ __align__(16) float s_out[MODE == 1 ? CTX * OP : 4];      // MODE 1: a finished tile on its way from the consumers to the fold

#ifdef RP_PROF
    unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    RP_T(t_entry);
#endif
    const ConvGeom& g = A.g;
    const int tid = threadIdx.x;
    const bool producer = tid >= 256;
    const int t = tid & 255, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    int bx, by, k;
    xcd_decode(blockIdx.x, A.nx, A.ny, A.nz, bx, by, k);
    const int m0 = by * CT;
    const int H = g.H, W = g.W, HW = H * W;
    const int RED = MODE == 0 ? g.Cin : g.Cout;
    const int MOUT = MODE == 0 ? g.Cout : g.Cin;
    const int mt = min(CT, MOUT - m0);
    const bool rem_blk = REM && by == A.ny - 1;
    const int mtx = mt + (rem_blk ? 4 : 0);
    const int n_chunks = RED / (4 * KS);                        // stages per tile (launcher: RED % (4 KS) == 0)

    const float* __restrict__ wk = A.w + (long long)k * A.wstride;
    float* __restrict__ s_w = s_dyn;
    ChanFwd* __restrict__ s_ch = reinterpret_cast<ChanFwd*>(s_dyn + 2 * WCH);
    RpBwd* __restrict__ s_chb = reinterpret_cast<RpBwd*>(s_ch + ((g.Cin + 3) & ~3));
    const bool fuse_sums = MODE == 1 && A.fbsums != nullptr;

    // per-channel constants, by the consumer waves (the producers go straight to their first global loads)
    if (!producer) {
        if (MODE == 0) {
            for (int c = tid; c < g.Cin; c += 256) s_ch[c] = chan_fwd(A.xin, k, c);
            if (tid < CT) { const int co = m0 + tid; s_bias[tid] = (co < g.Cout && g.b_off >= 0) ? wk[g.b_off + co] : 0.f; }
        } else {
            for (int c = tid; c < g.Cout; c += 256) { const ChanBwd b = chan_bwd(A.gin, k, c); RpBwd r; r.mean = b.mean; r.qc = -b.c1 * b.c3 * b.rstd; r.c1 = b.c1; r.k2 = -b.c1 * b.c2; s_chb[c] = r; }
            if (fuse_sums) for (int c = tid; c < g.Cin; c += 256) s_ch[c] = chan_fwd(A.xin, k, c);
        }
    }

    // the block's tiles by ordinal j: TILE(j).  Interleaved (default): b, b + nx, ... — at every moment the blocks of an XCD band work on
    // ADJACENT tiles, whose halo rows and the cache lines the 66-pixel window rows straddle are then L2 hits (as with one tile per block);
    // with a contiguous run of T tiles per block the neighbours ran T tiles apart in time and every window came from HBM again
    // (36->16 @256^2, T = 8: 503 MB of reads per launch against 288 MB at T = 1; profiles/r03_rp_traffic_tilings.txt).
    const int tstep = A.interleave ? A.nx : 1, tfirst = A.interleave ? bx : bx * A.tiles_per_block;
    const int n_my = A.interleave ? (A.n_tiles - bx + A.nx - 1) / A.nx : min(A.tiles_per_block, A.n_tiles - tfirst);
    if (n_my <= 0) return;
    auto TILE = [&](int j) { return tfirst + j * tstep; };
    const int n_iters = n_my * n_chunks;

    if (producer) {
        // ======================= producer waves =======================
#ifndef RP_NOPRIO
        __builtin_amdgcn_s_setprio(2);
#endif
        const int pw = wv;                                       // channel of the chunk this wave stages
        const float* __restrict__ xsrc = MODE == 0 ? A.xin.data + (long long)k * A.xin.sstride : A.gin.ga + (long long)k * A.gin.gstride;
        const float* __restrict__ ysrc = (MODE == 1 && A.gin.stats) ? A.gin.y + (long long)k * A.gin.ystride : nullptr;
        const bool xlrelu = (A.xin.act & 1) != 0; const float xslope = A.xin.slope;
        // The staging loop is kept to a minimum of instructions: its VALU / LDS-write issue competes with the consumers' matrix stream on the
        // same SIMDs (first version of this kernel: ~600 instructions per 5-quad stage, 0.64 of the matrix peak against 0.76 with the staging
        // switched off).  Item q = lane + 64 j of the channel's window = quad v of window row iy: goff = element offset of the float4 to load
        // (always a valid address; the channel base is wave-uniform and goes in as the scalar base of the load), flag (2 bits per item) =
        // 1 left image border (loaded x[0..3], wanted columns -1..2), 2 right border (loaded x[W-4..W-1], wanted W-1..W+2), 3 = a row
        // outside the image (MODE 1: zeros); anyf = items of this WAVE with a flagged lane (scalar: interior tiles skip the fix-up code).
        unsigned goff[NV]; unsigned flags = 0, anyf = 0;
        bool sp1 = false, sp2 = false;      // MODE 1: the tile of the stage in registers holds image row 1 / H-2
        float4 xv[KS][NV], yv[MODE == 1 ? KS : 1][MODE == 1 ? NV : 1];
        auto set_tile = [&](int tile) {
            const int px0 = W32 ? 0 : (tile % A.tiles_x) * 64, py0 = W32 ? tile * NSUB * TH : (tile / A.tiles_x) * TH;
            flags = 0; anyf = 0;
            sp1 = MODE == 1 && py0 == 0; sp2 = MODE == 1 && py0 + (W32 ? NSUB : 1) * TH == H;      // (narrow maps: image row H - 2 sits in the last group of the last tile)
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int q = min(lane + 64 * j, NITEM - 1), iy = q / Cfg::NQ, v = q - iy * Cfg::NQ;
                int gy = py0 - 1 + iy, gx = px0 - 1 + 4 * v; unsigned flag = 0;
                if (W32) { const int sub = v / (QPR + 1); gy += sub * TH; gx = -1 + 4 * (v - (QPR + 1) * sub); }      // group sub: image rows sub * TH further down
                if (MODE == 0) { gy = reflect_idx(gy, H); gy = min(max(gy, 0), H - 1); }
                else if (gy < 0 || gy >= H) { flag = 3; gy = 0; }
                if (gx < 0) { if (flag == 0) flag = 1; gx = 0; }
                else if (gx + 3 >= W) { if (flag == 0) flag = 2; gx = W - 4; }
                goff[j] = (unsigned)(gy * W + gx);
                flags |= flag << (2 * j);
                if (__builtin_amdgcn_ballot_w64(flag != 0) != 0) anyf |= 1u << j;
            }
        };
        auto prefetch = [&](int c0) {
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                const float* __restrict__ xc = xsrc + (long long)(c0 + 4 * s_ + pw) * HW;       // wave-uniform
                const float* __restrict__ yc = ysrc ? ysrc + (long long)(c0 + 4 * s_ + pw) * HW : xc;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
#if defined(RP_DBG_NOLOAD) || defined(RP_DBG_NOPROD)
                    xv[s_][j] = make_float4(1.f, 2.f, 3.f, 4.f); if (MODE == 1) yv[s_][j] = xv[s_][j]; continue;
#endif
                    const f4u a = *reinterpret_cast<const f4u*>(xc + goff[j]);
                    xv[s_][j] = make_float4(a.x, a.y, a.z, a.w);
                    if (MODE == 1) { if (ysrc) { const f4u b = *reinterpret_cast<const f4u*>(yc + goff[j]); yv[s_][j] = make_float4(b.x, b.y, b.z, b.w); } }
                }
            }
        };
        auto store = [&](int c0, float* __restrict__ dst) {
#ifdef RP_DBG_NOPROD
            return;
#endif
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                const int ch = c0 + 4 * s_ + pw;
                ChanFwd kf; RpBwd kb;
                if (MODE == 0) kf = s_ch[ch]; else kb = s_chb[ch];
                float* __restrict__ d = dst + (4 * s_ + pw) * PLANE + 4 * lane;
    #pragma unroll
                for (int j = 0; j < NV; ++j) {
                    float e[4] = {xv[s_][j].x, xv[s_][j].y, xv[s_][j].z, xv[s_][j].w};
                    // plain fp32 instructions: beside a matrix stream the packed forms (v_pk_fma_f32 ...) cost more issue time than the two
                    // scalar ones they replace (MI355X_MICROARCH.md, 'price of one filler beside MFMAs'; A/B on the three big layers: 2-4 %)
    #ifndef RP_DBG_RAWCOPY
                    if (MODE == 0) {
    #pragma unroll
                        for (int l = 0; l < 4; ++l) { float v = __builtin_fmaf(e[l] - kf.mean, kf.scale, kf.beta); if (xlrelu) v = __builtin_fmaxf(v, v * xslope); e[l] = v; }
                    } else if (ysrc) {
                        const float yy[4] = {yv[s_][j].x, yv[s_][j].y, yv[s_][j].z, yv[s_][j].w};
    #pragma unroll
                        for (int l = 0; l < 4; ++l) e[l] = __builtin_fmaf(yy[l] - kb.mean, kb.qc, __builtin_fmaf(e[l], kb.c1, kb.k2));
                    }
    #endif
                    if (anyf & (1u << j)) {       // wave-uniform: some lane of this item sits on an image border
                        const unsigned flag = (flags >> (2 * j)) & 3u;
                        if (MODE == 0) {
                            if (flag == 1) { const float e0 = e[0]; e[0] = e[1]; e[3] = e[2]; e[2] = e[1]; e[1] = e0; }      // columns -1..2 <- x[1], x[0], x[1], x[2]
                            else if (flag == 2) { e[0] = e[3]; e[1] = e[2]; }                                                 // columns W-1, W <- x[W-1], x[W-2]
                        } else {
                            if (flag == 1) { e[3] = e[2]; e[2] = e[1]; e[1] = e[0]; e[0] = 0.f; }                             // columns -1..2 <- 0, dy[0], dy[1], dy[2]
                            else if (flag == 2) { e[0] = e[3]; e[1] = 0.f; e[2] = 0.f; e[3] = 0.f; }                          // columns W-1.. <- dy[W-1], 0, 0, 0
                            else if (flag == 3) { e[0] = 0.f; e[1] = 0.f; e[2] = 0.f; e[3] = 0.f; }
                        }
                    }
    #ifdef RP_DBG_NOXSTORE
                    if (e[0] == 1.2345f)
    #endif
                    if (64 * (j + 1) <= NITEM || lane + 64 * j < NITEM) *reinterpret_cast<float4*>(d + 256 * j) = make_float4(e[0], e[1], e[2], e[3]);
                }
                if constexpr (MODE == 1) {
                    // row part of the reflection adjoint (header): window row iy = image row py0 - 1 + iy.  The wave's own ds_writes above are
                    // ordered before these reads (one wave's LDS operations complete in order).
                    if (sp1 && lane < Cfg::NQ) {        // tile row 0: S1 = dy[2] + dy[0] = window rows 3 + 1
                        const float4 u = *reinterpret_cast<const float4*>(d + 3 * PITCH), v = *reinterpret_cast<const float4*>(d + 1 * PITCH);
                        *reinterpret_cast<float4*>(d + Cfg::WROWS * PITCH) = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
                    }
                    if (sp2 && lane < Cfg::NQ) {        // last tile row: S2 = dy[H-3] + dy[H-1] = window rows TH - 2, TH
                        const float4 u = *reinterpret_cast<const float4*>(d + (TH - 2) * PITCH), v = *reinterpret_cast<const float4*>(d + TH * PITCH);
                        *reinterpret_cast<float4*>(d + (Cfg::WROWS + 1) * PITCH) = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
                    }
                }
            }
        };
        // weight chunk of reduction channels c0..c0+3: global -> registers (one stage ahead) -> LDS [f][k][m][ky][4] (+ extra channels [k][m 4][ky][4]).
        // One item = one DESTINATION quad (k, m, ky): its three taps are three consecutive floats of the global row [m][c][9] (MODE 1:
        // [c][m][9], reversed: flipped taps), fetched as one dword-aligned float4 — floats 0..3 of the row for its first tap row, 3..6 for
        // the second, 5..8 (taps in .y .z .w) for the third, so nothing outside the row is ever read — and stored with one ds_write_b128.  Source offset and LDS offset are fixed per thread.  (First version: aligned float4 of the source
        // scattered with four exec-masked ds_write_b32 each — 9 % of the kernel's time on the 132 -> 64 layer.)
        constexpr int WITEMS = CTX * 12;                          // quads per chunk
        constexpr int WNR = (WITEMS + 255) / 256;
        const float* __restrict__ wl = wk + g.w_off;
        f4u wreg[KS][WNR];
        int wsrc[WNR];               // float offset at chunk 0, -1: zeros (output channel beyond the tensor)
        short wdst_o[WNR];           // LDS float offset, -1: no such item
        bool wsh[WNR];               // third tap row of the source: the taps sit in .y .z .w
#pragma unroll
        for (int j = 0; j < WNR; ++j) {
            const int idx = t + 256 * j;
            wsrc[j] = -1; wdst_o[j] = -1; wsh[j] = false;
            if (idx < WITEMS) {
                const int ky = idx % 3, kk = (idx / 3) & 3, m = idx / 12;
                const int sr = MODE == 0 ? ky : 2 - ky;           // tap row of the source
                wsh[j] = sr == 2;
                wdst_o[j] = (short)(m < CT ? (((m >> 4) * 4 + kk) * 16 + (m & 15)) * 12 + ky * 4 : MF * WFR + (kk * 4 + (m - CT)) * 12 + ky * 4);
                if (m < mtx) wsrc[j] = (MODE == 0 ? ((m0 + m) * g.Cin + kk) * 9 : (kk * g.Cin + m0 + m) * 9) + (sr == 2 ? 5 : 3 * sr);
            }
        }
        const int wstep = MODE == 0 ? 36 : g.Cin * 36;            // floats per 4 reduction channels
        auto wfetch = [&](int c0) {
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                const float* __restrict__ wc = wl + (long long)((c0 >> 2) + s_) * wstep;
#pragma unroll
                for (int j = 0; j < WNR; ++j) {
                    wreg[s_][j] = (f4u){0.f, 0.f, 0.f, 0.f};
                    if (wsrc[j] >= 0) wreg[s_][j] = *reinterpret_cast<const f4u*>(wc + wsrc[j]);
                }
            }
        };
        auto wstore = [&](float* __restrict__ wdst) {
#ifdef RP_DBG_NOWSTORE
            return;
#endif
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                for (int j = 0; j < WNR; ++j)
                    if (256 * (j + 1) <= WITEMS || wdst_o[j] >= 0) {
                        const f4u v = wreg[s_][j];
                        const float t0 = wsh[j] ? v.y : v.x, t1 = wsh[j] ? v.z : v.y, t2 = wsh[j] ? v.w : v.z;
                        *reinterpret_cast<float4*>(wdst + s_ * WCH1 + wdst_o[j]) = MODE == 0 ? make_float4(t0, t1, t2, 0.f) : make_float4(t2, t1, t0, 0.f);
                    }
        };

        // ---- MODE 1: the fold of a dumped tile by the staging waves (idle for most of a stage otherwise).  Item idx = t + 256 j of
        //      [channel][tile row][float4 column]: the channel of (wave, j) is wave-uniform (TH * 16 >= 64 items per channel), so its BN
        //      constants are scalar and its BN-backward partial sums live in two registers per j for the whole block.  FP parts (items
        //      j == part mod FP), one per stage 0 .. FP-1 of the NEXT tile (FP <= n_chunks - 1: all parts are done before the consumers'
        //      next dump, which follows the MFMAs of that tile's last stage); a part's raw-x loads fly while the stage is staged.
        constexpr int NIT = MODE == 1 ? (CTX * TH * 16) / 256 : 1;
        static_assert(MODE == 0 || (CTX * TH * 16) % 256 == 0, "fold items");
        constexpr int FP = 3;                                    // parts when a tile has >= 4 stages; 2 or 3 stages: the whole fold rides on stage 0 (part -1)
        float fsum[NIT], fxs[NIT]; float4 fxr[MODE == 1 ? (NIT + FP - 1) / FP : 1];
#pragma unroll
        for (int j = 0; j < NIT; ++j) { fsum[j] = 0.f; fxs[j] = 0.f; }
        auto fold_fetch = [&](int tile, auto part_c) {
            constexpr int part = decltype(part_c)::value;
            if constexpr (MODE == 1) {
                if (!fuse_sums) return;
                const int px0 = W32 ? 0 : (tile % A.tiles_x) * 64, py0 = W32 ? tile * NSUB * TH : (tile / A.tiles_x) * TH;
                const float* __restrict__ xq = A.xin.data + (long long)k * A.xin.sstride + (long long)m0 * HW + py0 * W + px0;
                int tl = t; asm volatile("" : "+v"(tl));       // the items' index arithmetic is recomputed here, not hoisted over the stage loop (27 registers)
#pragma unroll
                for (int j = part; j < NIT; j += FP) {
                    const int idx = tl + 256 * j, ch = idx / (TH * 16), rw = (idx >> 4) % TH, v = idx & 15;
                    const int po = W32 ? (rw + (v >> WSH) * TH) * W + 4 * (v & (QPR - 1)) : rw * W + 4 * v;      // (narrow maps: float4 column v = group v >> WSH)
                    fxr[j / FP] = ch < mtx ? *reinterpret_cast<const float4*>(xq + ch * HW + po) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        auto fold_do = [&](int tile, auto part_c) {
            constexpr int part = decltype(part_c)::value;
            if constexpr (MODE == 1) {
                const int px0 = W32 ? 0 : (tile % A.tiles_x) * 64, py0 = W32 ? tile * NSUB * TH : (tile / A.tiles_x) * TH;
                float* __restrict__ o = A.fga + (long long)k * A.fga_sstride + (long long)m0 * HW + py0 * W + px0;
                const int xact = A.xin.act; const float xslope = A.xin.slope;
                int tl = t; asm volatile("" : "+v"(tl));
#pragma unroll
                for (int j = part; j < NIT; j += FP) {
                    const int idx = tl + 256 * j, ch = idx / (TH * 16), rw = (idx >> 4) % TH, v = idx & 15;
                    if (ch < mtx) {
                        const float4 d4 = *reinterpret_cast<const float4*>(&s_out[ch * OP + rw * 64 + 4 * v]);
                        float dd[4] = {d4.x, d4.y, d4.z, d4.w};
                        if (fuse_sums) {
                            // sums of ga and ga * (x - mean); the factor rstd of x-hat is applied once per block.  Views without an
                            // activation (the concat tensors: BatchNorm only) skip the LeakyReLU' part: 3 operations per element.
                            const float4 y4_ = fxr[j / FP]; const float yy[4] = {y4_.x, y4_.y, y4_.z, y4_.w};
                            const ChanFwd cf = s_ch[m0 + ch];
                            if (xact) {
#pragma unroll
                                for (int l = 0; l < 4; ++l) {
                                    const float ym = yy[l] - cf.mean;
                                    const float vv = __builtin_fmaf(ym, cf.scale, cf.beta);
                                    dd[l] *= (vv > 0.f) ? 1.f : xslope;
                                    fsum[j] += dd[l]; fxs[j] = __builtin_fmaf(dd[l], ym, fxs[j]);
                                }
                            } else {
#pragma unroll
                                for (int l = 0; l < 4; ++l) { fsum[j] += dd[l]; fxs[j] = __builtin_fmaf(dd[l], yy[l] - cf.mean, fxs[j]); }
                            }
                        }
                        *reinterpret_cast<float4*>(o + ch * HW + (W32 ? (rw + (v >> WSH) * TH) * W + 4 * (v & (QPR - 1)) : rw * W + 4 * v)) = make_float4(dd[0], dd[1], dd[2], dd[3]);
                    }
                }
            }
        };

        // running (tile, chunk) of the stage being fetched: two stages ahead of the consumers
        int ftile = 0, fc = 0;                            // (ordinal of the tile being fetched)
        auto advance = [&]() { fc += 4 * KS; if (fc >= RED) { fc = 0; ++ftile; return true; } return false; };
        set_tile(TILE(ftile)); prefetch(fc); wfetch(fc);
        __syncthreads();                                  // (S0) channel constants / bias visible
        store(fc, s_x[0]); wstore(s_w);
        int sc = 0;                                       // chunk base of the stage held in registers (stored next)
        if (n_iters > 1) { if (advance()) set_tile(TILE(ftile)); sc = fc; prefetch(fc); wfetch(fc); }
        lds_barrier();                                    // (A) chunk 0 published
        RP_T(p_loop0); RP_ACC(8, p_loop0 - t_entry);
        int fci = 0, fdt = -1;                            // stage index inside the consumers' current tile; ordinal of the tile whose dump is being folded
        for (int it = 0; it < n_iters; ++it) {
            RP_T(p0);
            // a part's raw-x loads are issued at the head of its stage and fly while the stage is staged.  (Issuing them one stage ahead —
            // they do not depend on the dump — was built and measured: 0 ... -5 %, NOTES.)
            constexpr std::integral_constant<int, 0> p0c{}; constexpr std::integral_constant<int, 1> p1c{}; constexpr std::integral_constant<int, 2> p2c{};
            const bool fold_now = MODE == 1 && fdt >= 0 && fci < FP;
            const int fdtile = TILE(max(fdt, 0));
            if (fold_now) { if (fci == 0) fold_fetch(fdtile, p0c); else if (fci == 1) fold_fetch(fdtile, p1c); else fold_fetch(fdtile, p2c); }
            if (it + 1 < n_iters) {
#ifdef RP_PROF
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                RP_T(p1); RP_ACC(9, p1 - p0);                                   // waiting for the prefetched loads
                store(sc, s_x[(it + 1) & 1]); wstore(s_w + ((it + 1) & 1) * WCH);
                RP_T(p2); RP_ACC(10, p2 - p1);                                  // transform + LDS writes
                if (it + 2 < n_iters) { if (advance()) set_tile(TILE(ftile)); sc = fc; prefetch(fc); wfetch(fc); }
                RP_T(p3); RP_ACC(11, p3 - p2);                                  // issuing the next loads
            }
            if (fold_now) { if (fci == 0) fold_do(fdtile, p0c); else if (fci == 1) fold_do(fdtile, p1c); else fold_do(fdtile, p2c); }
            RP_T(p4);
            lds_barrier();
            RP_T(p5); RP_ACC(12, p5 - p4);                                      // barrier wait
            if (++fci == n_chunks) { fci = 0; ++fdt; }
        }
        if constexpr (MODE == 1) {      // the block's last tile, then this thread's BN-backward partials (channel of (wave, j): wave-uniform)
            constexpr std::integral_constant<int, 0> p0c{}; constexpr std::integral_constant<int, 1> p1c{}; constexpr std::integral_constant<int, 2> p2c{};
            const int tlast = TILE(n_my - 1);
            fold_fetch(tlast, p0c); fold_do(tlast, p0c); fold_fetch(tlast, p1c); fold_do(tlast, p1c); fold_fetch(tlast, p2c); fold_do(tlast, p2c);
            if (fuse_sums) {
#pragma unroll
                for (int j = 0; j < NIT; ++j) {
                    const int ch = (t + 256 * j) / (TH * 16);
                    const float a_ = wave_sum(fsum[j]), b_ = wave_sum(fxs[j]) * s_ch[m0 + min(ch, mtx - 1)].rstd;
                    if (lane == 0 && ch < mtx) {
                        double* dst = A.fbsums + ((long long)k * g.Cin + m0 + ch) * 2;
                        atomicAdd(dst, (double)a_); atomicAdd(dst + 1, (double)b_);
                    }
                }
            }
        }
        RP_T(p_end); RP_ACC(13, p_end - p_loop0);
#ifdef RP_PROF
        if (t == 0) for (int i = 8; i < 14; ++i) atomicAdd(&g_rp_prof[i], prof[i]);
#endif
        if (MODE == 0 && A.out.stats != nullptr) __syncthreads();        // (Z)
    } else {
        // ======================= consumer waves =======================
        const bool do_stats = MODE == 0 && A.out.stats != nullptr;
        if (do_stats) for (int q = lane; q < CTX; q += 64) { s_red[wv][q][0] = 0.0; s_red[wv][q][1] = 0.0; }
        const int xb = l4 * PLANE + (wv * R) * PITCH + 4 * (l15 + (W32 ? l15 >> WSH : 0));      // this lane's six-float window read, row 0 of the wave
        const int wb = (l4 * 16 + l15) * 12;                             // weights of (k = l4, m = l15)
        const int wxb = MF * WFR + (l4 * 4 + (lane & 3)) * 12;           // REM: A operand of the 4x4x1 instruction = w[extra channel lane & 3][k = l4]
        __syncthreads();                                  // (S0)
        lds_barrier();                                    // (A)
        // the whole tile loop exists twice in a REM kernel: with the 4 extra channels (the block that owns the layer's last fragments) and
        // without them — one wave-uniform branch per block instead of conditions in the matrix stream
        auto consume = [&](auto remb_c) {
            constexpr bool REMB = decltype(remb_c)::value;
            f32x4 acc[MF][R][4];
            f32x4 accx[REMB ? R : 1][REMB ? 4 : 1];
            RP_T(c_loop0); RP_ACC(0, c_loop0 - t_entry);
            for (int it = 0; it < n_iters; ++it) {
                RP_T(c0);
                const int tile = TILE(it / n_chunks), ci = it % n_chunks;
                const int px0 = W32 ? 0 : (tile % A.tiles_x) * 64, py0 = W32 ? tile * NSUB * TH : (tile / A.tiles_x) * TH;
                const int row0 = py0 + wv * R;                               // first image row of this wave
                if (ci == 0) {
    #pragma unroll
                    for (int f = 0; f < MF; ++f)
    #pragma unroll
                        for (int r = 0; r < R; ++r)
    #pragma unroll
                            for (int j = 0; j < 4; ++j) acc[f][r][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if constexpr (REMB) {
    #pragma unroll
                        for (int r = 0; r < R; ++r)
    #pragma unroll
                            for (int j = 0; j < 4; ++j) accx[r][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                }
                const float* __restrict__ sx0 = s_x[it & 1] + xb;
                const float* __restrict__ sw0 = s_w + (it & 1) * WCH;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                const float* __restrict__ sx = sx0 + ks * 4 * PLANE;
                const float* __restrict__ sw = sw0 + ks * WCH1;
                // One stage = one straight-line block of MFMAs: no condition inside it (first version: one branch per 4x4x1 instruction cut the
                // matrix stream into basic blocks of two or three MFMAs with an s_waitcnt each).
                // window rows of the wave: R + 2 rows x 6 floats (+ MODE 1: the two column-patched operands, see the header)
                float b[R + 2][MODE == 1 ? 8 : 6];
    #pragma unroll
                for (int rr = 0; rr < R + 2; ++rr) {
                    int ro = rr * PITCH;
                    if constexpr (MODE == 1) {          // the spare window rows S1 / S2 stand in for dy[2] / dy[H-3] (header): wave-uniform; narrow maps: for the lanes of the first / last group
                        if constexpr (W32) {
                            if (rr == R + 1 && row0 + R - 1 == 1 && l15 < QPR) ro = (Cfg::WROWS - wv * R) * PITCH;
                            if (rr == 0 && row0 + (NSUB - 1) * TH == H - 2 && l15 >= 16 - QPR) ro = (Cfg::WROWS + 1 - wv * R) * PITCH;
                        } else {
                            if (rr == R + 1 && row0 + R - 1 == 1) ro = (Cfg::WROWS - wv * R) * PITCH;
                            if (rr == 0 && row0 == H - 2) ro = (Cfg::WROWS + 1 - wv * R) * PITCH;
                        }
                    }
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(sx + ro);
                    const f32x2 hi = *reinterpret_cast<const f32x2*>(sx + ro + 4);
                    b[rr][0] = lo.x; b[rr][1] = lo.y; b[rr][2] = lo.z; b[rr][3] = lo.w; b[rr][4] = hi.x; b[rr][5] = hi.y;
                }
                if constexpr (MODE == 1) {
                    // column part of the reflection adjoint: image column 1 (phase 1 of lane 0 in the leftmost tile) takes tap kx = 2 from
                    // win[3] + win[1]; column W-2 (phase 2 of lane 15 in the rightmost tile) takes tap kx = 0 from win[2] + win[4]
                    const float ml = (W32 ? (l15 & (QPR - 1)) == 0 : (px0 == 0 && l15 == 0)) ? 1.f : 0.f, mr = (W32 ? (l15 & (QPR - 1)) == QPR - 1 : (px0 + 64 == W && l15 == 15)) ? 1.f : 0.f;
    #pragma unroll
                    for (int rr = 0; rr < R + 2; ++rr) { b[rr][6] = __builtin_fmaf(ml, b[rr][1], b[rr][3]); b[rr][7] = __builtin_fmaf(mr, b[rr][4], b[rr][2]); }
                }
                auto bop = [&](int rr, int j, int kx) -> float {
                    if constexpr (MODE == 1) { if (j == 1 && kx == 2) return b[rr][6]; if (j == 2 && kx == 0) return b[rr][7]; }
                    return b[rr][j + kx];
                };
    #pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    f32x4 a[MF]; f32x4 ax;
    #pragma unroll
                    for (int f = 0; f < MF; ++f) a[f] = *reinterpret_cast<const f32x4*>(sw + f * WFR + wb + ky * 4);
                    if constexpr (REMB) ax = *reinterpret_cast<const f32x4*>(sw + wxb + ky * 4);
    #pragma unroll
                    for (int r = 0; r < R; ++r)
    #pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
    #pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float bv = bop(r + ky, j, kx);
    #pragma unroll
                                for (int f = 0; f < MF; ++f) acc[f][r][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f][kx], bv, acc[f][r][j], 0, 0, 0);
                                if constexpr (REMB) accx[r][j] = __builtin_amdgcn_mfma_f32_4x4x1f32(ax[kx], bv, accx[r][j], 0, 0, 0);
                            }
                }
                }

                RP_T(c1); RP_ACC(1, c1 - c0);                                        // MFMA phase (issue time: the last MFMAs still run)
    #ifdef RP_DBG_NOEPI
                if (ci == n_chunks - 1 && A.tiles_x < 0) {
    #else
                if (ci == n_chunks - 1) {
    #endif
                    // ---- epilogue: register q of (f, r, phase 0..3) = channel m0 + 16 f + 4 l4 + q, row row0 + r, pixels px0 + 4 l15 .. +3 ----
                    if constexpr (MODE == 0) {
                        float* __restrict__ yout = A.out.data + (long long)k * A.out.sstride + (long long)m0 * HW;
    #pragma unroll
                        for (int f = 0; f < MF; ++f) {
                            float fs[4] = {0.f, 0.f, 0.f, 0.f}, fq[4] = {0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                            for (int r = 0; r < R; ++r)
    #pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const int ml = f * 16 + l4 * 4 + q;
                                    const float bi = s_bias[ml];
                                    const float v0 = acc[f][r][0][q] + bi, v1 = acc[f][r][1][q] + bi, v2 = acc[f][r][2][q] + bi, v3 = acc[f][r][3][q] + bi;
                                    if (ml < mt) {
                                        if constexpr (W32) *reinterpret_cast<float4*>(yout + ml * HW + (row0 + r + (l15 >> WSH) * TH) * W + 4 * (l15 & (QPR - 1))) = make_float4(v0, v1, v2, v3);
                                        else *reinterpret_cast<float4*>(yout + ml * HW + (row0 + r) * W + px0 + 4 * l15) = make_float4(v0, v1, v2, v3);
                                        fs[q] += (v0 + v1) + (v2 + v3);
                                        fq[q] = __builtin_fmaf(v0, v0, __builtin_fmaf(v1, v1, __builtin_fmaf(v2, v2, __builtin_fmaf(v3, v3, fq[q]))));
                                    }
                                }
                            if (do_stats) {
    #pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    float a_ = fs[q], b_ = fq[q];
    #pragma unroll
                                    for (int o = 8; o > 0; o >>= 1) { a_ += __shfl_xor(a_, o, 64); b_ += __shfl_xor(b_, o, 64); }
                                    if (l15 == 0) { s_red[wv][f * 16 + l4 * 4 + q][0] += (double)a_; s_red[wv][f * 16 + l4 * 4 + q][1] += (double)b_; }
                                }
                            }
                        }
                    } else {
                        // hand the tile to the staging waves: accumulators -> s_out[channel][tile row][64 pixels] as float4 (one ds_write_b128 per
                        // register index); they fold and store it during the next tile's stages while this wave goes on with its MFMAs
#pragma unroll
                        for (int f = 0; f < MF; ++f)
#pragma unroll
                            for (int r = 0; r < R; ++r)
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    *reinterpret_cast<float4*>(&s_out[(f * 16 + l4 * 4 + q) * OP + (wv * R + r) * 64 + 4 * l15]) =
                                        make_float4(acc[f][r][0][q], acc[f][r][1][q], acc[f][r][2][q], acc[f][r][3][q]);
                        if constexpr (REMB) {
                            // 4x4x1 accumulators: register q of (r, phase j) = extra channel q at pixel 4 l15 + j, this lane's reduction-channel
                            // slice (k = l4): add the four slices; lane group l4 == q' then holds channel q'
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                float dd[4];
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    float mine = 0.f;
#pragma unroll
                                    for (int q = 0; q < 4; ++q) {
                                        float v = accx[r][j][q]; v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                                        if (l4 == q) mine = v;
                                    }
                                    dd[j] = mine;
                                }
                                *reinterpret_cast<float4*>(&s_out[(CT + l4) * OP + (wv * R + r) * 64 + 4 * l15]) = make_float4(dd[0], dd[1], dd[2], dd[3]);
                            }
                        }
                    }
                }
                RP_T(c2); RP_ACC(2, c2 - c1);                                        // epilogue
                lds_barrier();
                RP_T(c3); RP_ACC(3, c3 - c2);                                        // barrier wait
            }
            RP_T(c_end); RP_ACC(4, c_end - c_loop0);
    #ifdef RP_PROF
            if (t == 0) { for (int i = 0; i < 5; ++i) atomicAdd(&g_rp_prof[i], prof[i]); atomicAdd(&g_rp_prof[5], 1ull); atomicAdd(&g_rp_prof[6], (unsigned long long)n_iters); }
    #endif
        };
        if constexpr (REM) { if (rem_blk) consume(std::true_type{}); else consume(std::false_type{}); }
        else consume(std::false_type{});
        if (do_stats) {
            __syncthreads();                              // (Z)
            if (t < CTX * 2) {
                const int q = t >> 1, which = t & 1;
                if (q < mtx)
                    atomicAdd((MODE == 0 ? A.out.stats + ((long long)k * g.Cout + m0 + q) * 2 : A.fbsums + ((long long)k * g.Cin + m0 + q) * 2) + which,
                              s_red[0][q][which] + s_red[1][q][which] + s_red[2][q][which] + s_red[3][q][which]);
            }
        }
    }
}

template <int MODE, int MF, int R, bool REM, int KS, int WSH = 0>
int launch_rp(RpArgs& A, int T, int n_samples, hipStream_t st)
{
    using Cfg = RpCfg<R, MODE, WSH>;
    constexpr bool W32 = WSH > 0; constexpr int QPR = Cfg::QPR, NSUB = Cfg::NSUB;
    const ConvGeom& g = A.g;
    const int MOUT = MODE == 0 ? g.Cout : g.Cin;
    constexpr int CT = 16 * MF;
    if (g.H % (W32 ? NSUB * Cfg::TH : Cfg::TH)) return -3;
    const int RED = MODE == 0 ? g.Cin : g.Cout;
    if (RED % (4 * KS)) return -3;
    if (MODE == 1 && RED / (4 * KS) < 4) return -3;       // the fold of a tile rides on stages 0..2 of the next one
    if (REM && !((MOUT & 15) == 4 && (MOUT - 4) % CT == 0)) return -3;
    A.tiles_x = W32 ? 1 : g.W / 64;
    A.n_tiles = W32 ? g.H / (NSUB * Cfg::TH) : A.tiles_x * (g.H / Cfg::TH);
    A.tiles_per_block = T;
    { static const int il = [] { const char* e = getenv("MFVI_RP_INTERLEAVE"); return !(e && e[0] == '0'); }(); A.interleave = il; }
    A.nx = (A.n_tiles + T - 1) / T; A.ny = REM ? (MOUT - 4) / CT : (MOUT + CT - 1) / CT; A.nz = n_samples;
    constexpr int WCH = KS * (MF * 4 * 16 * 12 + (REM ? 4 * 4 * 12 : 0));
    const size_t dyn = sizeof(float) * 2 * WCH + sizeof(ChanFwd) * (size_t)((g.Cin + 3) & ~3) + sizeof(RpBwd) * (size_t)((g.Cout + 3) & ~3);
    constexpr int MINW = 4;
    mfvi_tl_family = 2;
    mfvi_launch((conv_rp_kernel<MODE, MF, R, REM, KS, MINW, WSH>), dim3(A.nx * A.ny * A.nz), dim3(512), dyn, st, A);
    return (int)hipGetLastError();
}

// tune code: mf | r << 8 | rem << 12 | ks << 13 | T << 16 | MFVI_TUNE_RP   (ks = k-steps per stage: 0 / 1 -> 1, 2)
template <int MODE>
int dispatch_rp(RpArgs& A, int tune, int n_samples, hipStream_t st)
{
    const int mf = tune & 255, r = (tune >> 8) & 15, rem = (tune >> 12) & 1, ks = max(1, (tune >> 13) & 7), T = max(1, (tune >> 16) & 255);
#define RP_GO2(MF_, R_, KS_) if (mf == MF_ && r == R_ && ks == KS_) { if constexpr (MODE == 1) { if (rem) return launch_rp<MODE, MF_, R_, true, KS_>(A, T, n_samples, st); } if (rem) return -3; return launch_rp<MODE, MF_, R_, false, KS_>(A, T, n_samples, st); }
#define RP_GO(MF_, R_) RP_GO2(MF_, R_, 1)      /* two k-steps per stage (KS = 2) built and measured: no gain, register spills in backward-data; not instantiated */
    if (A.g.W == 32) {      // maps 32 wide: two image rows per 64-pixel strip
        if constexpr (MODE == 0) {
#define RP_GO32(MF_, R_) if (mf == MF_ && r == R_ && ks == 1 && !rem) return launch_rp<0, MF_, R_, false, 1, 3>(A, T, n_samples, st);
            RP_GO32(1, 1) RP_GO32(1, 2) RP_GO32(2, 1) RP_GO32(2, 2) RP_GO32(4, 1) RP_GO32(1, 4)
#undef RP_GO32
        } else {
#define RP_GO32(MF_, R_) if (mf == MF_ && r == R_ && ks == 1) { if (rem) return launch_rp<1, MF_, R_, true, 1, 3>(A, T, n_samples, st); return launch_rp<1, MF_, R_, false, 1, 3>(A, T, n_samples, st); }
            RP_GO32(1, 1) RP_GO32(1, 2) RP_GO32(2, 1)
#undef RP_GO32
        }
        return -3;
    }
    if (A.g.W == 16) {      // maps 16 wide: four image rows per strip (a 16 x 16 map is ONE tile of 4-row groups)
        if constexpr (MODE == 0) {
#define RP_GO16(MF_, R_) if (mf == MF_ && r == R_ && ks == 1 && !rem) return launch_rp<0, MF_, R_, false, 1, 2>(A, T, n_samples, st);
            RP_GO16(1, 1) RP_GO16(2, 1) RP_GO16(4, 1) RP_GO16(1, 2) RP_GO16(2, 2)
#undef RP_GO16
            // one 16 x 16 map = one tile per block and 32 stages of 4 channels: a latency chain of ~1.2 us per stage (barrier, LDS round
            // trip, load latency) with nothing else on the CU to hide it -> 2 or 4 k-steps per stage
#define RP_GO16K(MF_, KS_) if (mf == MF_ && r == 1 && ks == KS_ && !rem) return launch_rp<0, MF_, 1, false, KS_, 2>(A, T, n_samples, st);
            RP_GO16K(1, 2) RP_GO16K(1, 4) RP_GO16K(2, 2) RP_GO16K(2, 4)
#undef RP_GO16K
        } else {
#define RP_GO16K(MF_, KS_) if (mf == MF_ && r == 1 && ks == KS_) { if (rem) return launch_rp<1, MF_, 1, true, KS_, 2>(A, T, n_samples, st); return launch_rp<1, MF_, 1, false, KS_, 2>(A, T, n_samples, st); }
            RP_GO16K(1, 2) RP_GO16K(1, 4) RP_GO16K(2, 2)
#undef RP_GO16K
#define RP_GO16(MF_, R_) if (mf == MF_ && r == R_ && ks == 1) { if (rem) return launch_rp<1, MF_, R_, true, 1, 2>(A, T, n_samples, st); return launch_rp<1, MF_, R_, false, 1, 2>(A, T, n_samples, st); }
            RP_GO16(1, 1) RP_GO16(2, 1) RP_GO16(1, 2)
#undef RP_GO16
        }
        return -3;
    }
    RP_GO(1, 1) RP_GO(1, 2) RP_GO(2, 1)
    if constexpr (MODE == 0) { RP_GO(2, 2) RP_GO2(4, 1, 1) RP_GO2(1, 4, 1) }       // backward-data: rows 1 / H-2 must be the last / first row of their wave (R <= 2); its out tile keeps (2, 2) / (4, 1) at one block per CU
#undef RP_GO
#undef RP_GO2
    return -3;
}

}  // namespace

// Returns -2 when the shape is not served by the row-phase kernels, -3 when the tiling is not valid for it.
int launch_conv_fwd_rp(const TView& in, const ConvGeom& g, const float* w, long long wstride, OutDesc out, int tune, int n_samples, hipStream_t st)
{
    if (g.ks != 3 || g.stride != 1 || ((g.W & 63) && g.W != 32 && g.W != 16) || (g.H & 3) || (g.Cin & 3) || (g.w_off & 3) || g.Cin > MFVI_MAX_C) return -2;
    if ((in.sstride & 3) || ((uintptr_t)in.data & 15) || (out.sstride & 3) || ((uintptr_t)out.data & 15)) return -2;
    if (in.act & MFVI_ACT_SQUARE) return -2;                                          // variance convolution of the LRT layers: round-2 kernels
    if ((long long)max(g.Cin, g.Cout) * g.H * g.W >= (1LL << 29)) return -2;          // 32-bit element offsets per sample, two flag bits
    RpArgs A{};
    A.xin = in; A.g = g; A.w = w; A.wstride = wstride; A.out = out;
    return dispatch_rp<0>(A, tune & ~(1 << 12), n_samples, st);      // the remainder bit only concerns backward-data
}

int launch_conv_bwd_data_rp(const GView& gy, const ConvGeom& g, const float* w, long long wstride, int tune, int n_samples, hipStream_t st, const FoldFuse& fuse)
{
    if (g.ks != 3 || g.stride != 1 || ((g.W & 63) && g.W != 32 && g.W != 16) || (g.H & 3) || g.H < 4 || (g.Cin & 3) || (g.Cout & 3) || (g.w_off & 3) || g.Cout > MFVI_MAX_C || g.Cin > MFVI_MAX_C) return -2;
    if (!fuse.ga || (fuse.ga_sstride & 3) || ((uintptr_t)fuse.ga & 15) || g.Cout < 16) return -2;      // the fold of a tile rides on stages 0..2 of the next one
    if ((gy.gstride & 3) || ((uintptr_t)gy.ga & 15) || (gy.stats && ((gy.ystride & 3) || ((uintptr_t)gy.y & 15)))) return -2;
    if (fuse.bsums && ((fuse.x.sstride & 3) || ((uintptr_t)fuse.x.data & 15))) return -2;
    if ((long long)max(g.Cin, g.Cout) * g.H * g.W >= (1LL << 29)) return -2;
    RpArgs A{};
    A.xin = fuse.x; A.gin = gy; A.g = g; A.w = w; A.wstride = wstride;
    A.fga = fuse.ga; A.fga_sstride = fuse.ga_sstride; A.fbsums = fuse.bsums;
    return dispatch_rp<1>(A, tune, n_samples, st);
}

// Heuristic tiling when the plan holds none (mfvi_plan_autotune times the candidates on the real tensors).  MFVI_RP=0 keeps the round-2 kernels
// (A/B runs and the cross-check tests); MFVI_TUNE_RP=mf,r,T[,rem] forces one tiling for experiments.
int rp_default_tune(const ConvGeom& g, int mode, int n_samples)
{
    static const int on = [] { const char* e = getenv("MFVI_RP"); return !(e && e[0] == '0'); }();
    if (!on || g.ks != 3 || g.stride != 1 || ((g.W & 63) && !(g.W == 32 && (g.H & 7) == 0) && !(g.W == 16 && (g.H & 15) == 0)) || (g.H & 3)) return 0;
    static const int forced = [] { int mf = 0, r = 0, T = 1, rem = 0, ks = 1; const char* e = getenv("MFVI_TUNE_RP"); if (e) sscanf(e, "%d,%d,%d,%d,%d", &mf, &r, &T, &rem, &ks); return mf > 0 ? (mf | r << 8 | (rem & 1) << 12 | (ks & 7) << 13 | T << 16) : 0; }();
    if (forced) return forced | MFVI_TUNE_RP;
    const int MOUT = mode == 0 ? g.Cout : g.Cin;
    const int rem = (mode == 1 && (MOUT & 15) == 4) ? 1 : 0;
    const int mo = MOUT - 4 * rem;
    if (mo < 16) return 0;
    if (mode == 1 && g.Cout < 16) return 0;
    const int mf = (mo % 32 == 0) ? 2 : 1;
    const long long units = (long long)max(1, g.W / 64) * (g.H / (g.W == 32 ? 8 : g.W == 16 ? 16 : 4)) * ((mo + 16 * mf - 1) / (16 * mf)) * n_samples;      // blocks with 4-row tiles, one per block
    // measured on the three big layers (profiles/r03_rp_layers.txt): forward — tall tiles (fewer stage barriers per MFMA) while the grid
    // still fills the chip twice; backward-data — 4-row tiles (the out tile of the fold is LDS) and several tiles per block (the last
    // tile's fold and the block prologue are exposed once per block)
    int r = 1, T = 1;
    // rows of a tile: 4 r image rows on the wide maps, NSUB * 4 r on the narrow ones (two / four image rows side by side): launch_rp's divisibility rule
    const int rows1 = g.W == 32 ? 8 : g.W == 16 ? 16 : 4;
    if (mode == 0) {
        if (mf == 1 && g.H % (4 * rows1) == 0 && units / 4 >= 1024) r = 4;
        else if (g.H % (2 * rows1) == 0 && units / 2 >= 512) r = 2;
    } else {
        while (T < 8 && units / (2 * T) >= 512) T *= 2;
    }
    return mf | r << 8 | rem << 12 | T << 16 | MFVI_TUNE_RP;
}

#ifdef RP_PROF
extern "C" int mfvi_debug_rp_prof(unsigned long long* out16, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_rp_prof), sizeof(unsigned long long) * 16);
    if (e == hipSuccess && reset) { unsigned long long z[16] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_rp_prof), z, sizeof(z)); }
    return (int)e;
}
#endif
