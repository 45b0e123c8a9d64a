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
// Per 512-thread block: 32 x TH pixel tiles x (16*MF) output channels.  The block copies its slab of the
// weights of sample k (w = mu + softplus(rho)*eps, drawn once per pass by sample_weights_kernel) into LDS —
//   WS = true  (weight-stationary): the whole slab [taps][all reduction channels][16*MF] once, then it walks several
//               pixel tiles, so every sampled weight is reused by thousands of pixels (high-resolution layers);
//   WS = false: one CC-channel chunk at a time inside the reduction loop (slabs too big for LDS / few tiles).
// Activation tiles stream through LDS in CC-channel chunks (deferred BN + LeakyReLU applied on the way, reflection in
// the index math); the next chunk's global loads are issued into registers before the MFMA loop of the current one.
// Each of the 4 waves owns TH/4 rows = NF = TH/2 pixel fragments and issues MF*NF MFMAs per (tap, 4-channel) step from
// conflict-free ds_read_b32:
//   activations  s_x[k][row][col], plane pitch == 16 (mod 32) floats  -> lanes 0-15 / 16-31 of a half-wave hit disjoint banks
//   weights      s_w[tap][k][m],   row pitch   == 16 (mod 32) floats
// Epilogue (MODE 0): + sampled bias, raw store, per-channel sum / sum^2 of the BatchNorm that follows (fp64 atomics).
#include "common.h"
#include <type_traits>
#include <cstdio>
#include <cstdlib>

namespace {

#ifndef PRODUCER_PRIO
#define PRODUCER_PRIO 2
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2a __attribute__((ext_vector_type(2)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));      // 4 floats at dword alignment (padded-gradient rows)
constexpr int EPP = 36;                 // row pitch of the per-wave output transpose buffer: 16 channels x 32 pixels (+4)

constexpr int pitch16(int n) { return ((n + 15) / 32) * 32 + 16; }      // smallest p >= n with p % 32 == 16

template <int KS, int STRIDE, int MF, int TH, bool BIGC = false, bool REM = false>
struct MCfg {
    static constexpr int TW = 32;
    static constexpr int CT = 16 * MF;
    static constexpr int CTX = CT + (REM ? 4 : 0);                // REM: the block that holds the layer's last 4 output channels carries them too
    // Reduction channels per activation stage.  A stage costs one barrier and (with the one-stage register prefetch) about one
    // global-load latency; a 1x1 convolution has so little matrix work per 8 channels (2 k-steps) that its loop ran at
    // latency x Cin/8 (26 us for 128 -> 4 channels on a 16x16 map).  Its windows are small (no halo), so with BIGC it stages 32
    // (16 for 16-row tiles) channels at a time: a quarter of the iterations, four times the bytes in flight per lane.  The 70 KB
    // of LDS halve the blocks per CU, which costs the bandwidth-bound 16/32-channel layers at 128^2 / 256^2 more than it saves:
    // BIGC is used from 64 reduction channels up.  The same holds for 3x3 layers on the 8x8 ... 32x32 maps at the bottom of the
    // hour-glass (FLAT tiles of 2 or 4 rows, 64-132 channels = 8-17 stages of 18 k-steps): they stage 32 channels with BIGC too.
#ifndef MFVI_CC1
#define MFVI_CC1 32
#endif
    // 3x3 stride-1 layers stage 4 channels at a time: with 48 KB of LDS three blocks share a CU instead of two (more MFMA streams
    // to interleave), and the extra barriers cost nothing measurable (4.10 -> 4.07 ms per iteration).
#ifndef MFVI_CC0
#define MFVI_CC0 4
#endif
    // (5x5 layers of the inpainting nets: 4 channels too — their windows carry a 4-row / 4-column halo, 25 taps per channel step)
    static constexpr int CC = !BIGC ? (((KS == 3 && STRIDE == 1) || KS == 5) ? MFVI_CC0 : 8) : (KS == 1 ? (TH >= 16 ? MFVI_CC1 / 2 : MFVI_CC1) : 32);
    static constexpr int NF = TH / 2;                              // pixel fragments per wave (TH/4 rows x 2 halves)
    static constexpr int KK = KS * KS;
    static constexpr int IN_TH = (TH - 1) * STRIDE + KS;
    static constexpr int IN_TW = (TW - 1) * STRIDE + KS;
    // Rectangular tiles stage the window as ALIGNED float4 columns: the LDS row starts 4 columns left of the tile when there is a
    // halo (the tile starts at a multiple of 32 columns, images are a multiple of 4 wide), so a row is WV floats and the first
    // column a tap needs sits at XOFF (+1 less for backward-data, whose halo is KS-1 instead of KS/2).
    static constexpr int HALO4 = KS > 1 ? 4 : 0;
    static constexpr int WV = ((HALO4 ? HALO4 - KS / 2 : 0) + IN_TW + 3) / 4 * 4;   // 40 (3x3, 5x5), 68 (3x3 stride 2), 72 (5x5 stride 2), 32 (1x1)
    static constexpr int PITCH = WV;
    static constexpr int PLANE = pitch16(IN_TH * PITCH);           // == 16 (mod 32)
    static constexpr int CTP = pitch16(CTX);                       // == 16 (mod 32)
    static constexpr int X_FLOATS = CC * PLANE;
};

static_assert(MCfg<3, 1, 1, 16>::WV == 40 && MCfg<3, 2, 1, 8>::WV == 68 && MCfg<1, 1, 1, 16>::WV == 32 && MCfg<3, 1, 1, 16>::PLANE == 720, "window");
static_assert(pitch16(16) == 16 && pitch16(32) == 48 && pitch16(64) == 80 && pitch16(340) % 32 == 16, "pitch16");

struct MfmaArgs {
    TView xin; GView gin; ConvGeom g;
    const float* w; long long wstride;     // weights of sample k at w + k*wstride (drawn by sample_weights_kernel; stride 0 = mu)
    OutDesc out; float* dxp; long long dxp_sstride;
    // backward-data of a 1x1 layer whose input has no other consumer: the fold (LeakyReLU', BN-backward sums of the input tensor) runs in
    // the epilogue and the gradient goes straight to ga (xin = view of that input tensor); fga == nullptr: plain padded-gradient output
    float* fga; long long fga_sstride; double* fbsums;
    int tiles_x, n_tiles, tiles_per_block;
    int nx, ny, nz;                        // logical grid (tile groups, output-channel tiles, samples), launched 1-D
    int vec_out;                           // forward: output rows / strides / pointer allow aligned float4 stores
    int ow, rt, wpitch, nwin;              // FLAT tiles: output-domain width, rows per tile, window pitch, window positions
};

// FLAT: for narrow output domains (<= ~66 wide) a tile is `rt` FULL rows of the domain, its TH*32 pixels dealt to the MFMA
// fragments in row-major order, instead of a TH x 32 rectangle: a 34-wide padded-gradient domain then fills 99% of the
// fragments (2 x 32-pixel tile columns fill 53%).  The staged window keeps the same LDS plane with a run-time pitch.
// PH (MODE 1, stride-2 layers, rectangular tiles): phase decomposition of the transposed convolution.  The gradient wrt the padded input
// at (r, c) only receives taps with (r + ky) and (c + kx) even; a pixel fragment therefore takes 16 pixels of ONE column parity
// (columns px0 + 2*l15 + parity) of one row, and issues the 1 / 2 / 2 / 4 of the 9 taps its (row parity, column parity) class can see,
// reading the UN-stuffed dy window ((TH/2 + 2) rows x 24 columns) — a quarter of the MFMAs and of the staged window of the
// zero-stuffed formulation, same per-element accumulation order (the skipped products are exact zeros).
// FF (MODE 1, 3x3 stride-1 layers whose input tensor has no other consumer): the gradient is formed on the UN-padded input domain and the
// fold runs in the epilogue.  The adjoint of ReflectionPad2d(1) sends padded row 0 to input row 1 and padded row H+1 to row H-2 (columns
// alike); in the tap sum  dx[i][j] = sum_{ky,kx} w'[ky][kx] * dy[i-1+ky][j-1+kx]  that is
//     row i == 1:    B(ky=2, kx) += B(ky=0, kx)         row i == H-2:  B(ky=0, kx) += B(ky=2, kx)
//     col j == 1:    B(ky, kx=2) += B(ky, kx=0)         col j == W-2:  B(ky, kx=0) += B(ky, kx=2)       (corners: both, 4 terms)
// on the pixel operand B of the lanes concerned — a few extra LDS reads in the border tiles, no extra MFMA, and neither the (H+2)x(W+2)
// padded-gradient scratch (written and re-read: 2 x 100 MB per pass of up_9) nor the finalize_dx launch exist any more.
// REM (with FF, rectangular 8-row tiles): the skip() concats have 4 + 32/64/128 channels, so backward-data's output channels come as
// 36 / 68 / 132 = whole 16-channel fragments + 4.  Instead of padding the 4 to a fifth / ninth ... fragment (33 / 18 / 9 % of the MFMAs of
// those layers), the block that owns the last fragments also carries the 4 extra channels on v_mfma_f32_4x4x1_16B_f32: 16 independent
// 4x4 outer products per instruction — block b = lane >> 2 takes the B operand's (k = b >> 2, pixels 4*(b & 3)..+3), i.e. exactly the
// lanes of the 16x16x4 pixel fragment already in registers, against A = w[extra channel lane & 3][k = lane >> 4] — 8 cycles instead of
// a 32-cycle fragment.  Each lane accumulates its own k-slice; the four slices are added across lanes (l15 + 16*l4) in the epilogue.
template <int KS, int STRIDE, int MF, int TH, int MODE, bool WS, bool FLAT, bool BIGC = false, bool PH = false, bool FF = false, bool REM = false>
__global__ __launch_bounds__(512, (BIGC ? (MF * TH <= 32 ? 2 : 1) : (MF * TH <= 16 ? 4 : (MF * TH <= 32 ? 2 : 1)))) void conv_mfma_kernel(MfmaArgs A)
{
    static_assert(!PH || (MODE == 1 && KS == 3 && !FLAT && !BIGC), "phase decomposition: 3x3 backward-data on rectangular tiles");
    static_assert(!FF || (MODE == 1 && KS == 3 && STRIDE == 1 && !PH), "fused fold: 3x3 stride-1 backward-data");
    static_assert(!REM || (FF && !FLAT && !BIGC && TH == 8), "remainder channels: fused-fold backward-data on rectangular 8-row tiles");
    using Cfg = MCfg<KS, STRIDE, MF, TH, BIGC, REM>;
    constexpr int TW = Cfg::TW, CT = Cfg::CT, CTX = Cfg::CTX, CC = Cfg::CC, NF = Cfg::NF, KK = Cfg::KK, P = KS / 2;
    constexpr int IN_TH = Cfg::IN_TH, IN_TW = Cfg::IN_TW, PITCH = Cfg::PITCH, PLANE = Cfg::PLANE, CTP = Cfg::CTP;
    constexpr int WCHUNK = KK * CC * CTP;               // floats of one weight chunk (non-WS double buffer)
    // PEPI (fused fold on rectangular tiles): the PRODUCER waves run the fold.  The consumers only dump a finished tile's accumulators into
    // s_out and go on with the next tile's MFMAs; the producers — idle for ~40 % of a tile otherwise — read it back row-wise during the next
    // tile's stages (raw x from global, LeakyReLU', BN-backward sums, coalesced float4 stores of ga).  The hand-over rides on the stage
    // barriers that exist anyway: the dump precedes the barrier that ends the tile's last stage, the three parts of the fold run in
    // the next tile's stages 0 .. 2, i.e. before the barrier that precedes the next dump (n_chunks >= 4: launcher).
    constexpr bool PEPI = FF && !FLAT;
    constexpr int OP = TH * 32 + 4;                      // channel pitch of s_out: 4 * OP == 16 (mod 32), the four 4-channel row groups of a dump land in distinct banks
    static_assert(MODE == 0 || STRIDE == 1, "backward-data always stages a stride-1 window");
    // MODE 1 of a stride-2 layer (g.stride == 2): the transposed convolution is the same full correlation over the ZERO-STUFFED
    // gradient G[r][c] = dy[r/2][c/2] (r, c even), formed on the fly; 3/4 of the MACs multiply zeros, but on the matrix cores.

    // Wave specialisation: waves 0-3 (consumers) only issue MFMAs; waves 4-7 (producers) stream the next activation
    // chunk (global loads -> deferred BN/LeakyReLU or BN-backward -> LDS) and sample the next weight chunk into the
    // other half of a double buffer.  One barrier per chunk; VALU/VMEM work hides under the matrix pipe.
    extern __shared__ __align__(16) float s_w[];          // WS: [KK][REDP][CTP]; else 2 x [KK][CC][CTP]
    __shared__ __align__(16) float s_x[2][Cfg::X_FLOATS];
    __shared__ float s_bias[CT];
    __shared__ double s_red[4][CTX][2];
    __shared__ __align__(16) float s_ep[(FLAT || PEPI) ? 1 : 4][(FLAT || PEPI) ? 1 : 16][(FLAT || PEPI) ? 4 : EPP];    // rectangular tiles: epilogue transpose, one slab per consumer wave
    __shared__ __align__(16) float s_out[PEPI ? CTX * OP : 4];

    const ConvGeom& g = A.g;
    const int tid = threadIdx.x;
    const bool producer = tid >= 256;
    const int t = tid & 255, lane = t & 63, wv = t >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    int bx, by, k;
    xcd_decode(blockIdx.x, A.nx, A.ny, A.nz, bx, by, k);
    const int m0 = by * CT;                                          // first output channel of the block

    // MODE 0: reduce over cin, outputs = cout.   MODE 1: reduce over cout, outputs = cin.
    const int RED = MODE == 0 ? g.Cin : g.Cout;
    const int MOUT = MODE == 0 ? g.Cout : g.Cin;
    const int mt = min(CT, MOUT - m0);
    const bool rem_blk = REM && by == A.ny - 1;                      // this block also owns the layer's last 4 output channels (m0 + CT ..)
    const int mtx = mt + (rem_blk ? 4 : 0);
    const int REDP = WS ? ((RED + 3) & ~3) : CC;                     // reduction-channel pitch of s_w
    const int n_chunks = (RED + CC - 1) / CC;

    const float* __restrict__ wk = A.w + (long long)k * A.wstride;
    // per-channel constants of the input view (ChanFwd[Cin]) and of the gradient view (ChanBwd[Cout]) live in dynamic LDS behind the weights,
    // sized by the layer (static MFVI_MAX_C-sized tables cost 9 KB per block — the difference between one and two blocks per CU for the
    // fused backward-data kernel with its out tile)
    ChanFwd* __restrict__ s_ch = reinterpret_cast<ChanFwd*>(s_w + (WS ? KK * REDP * CTP : 2 * WCHUNK));
    ChanBwd* __restrict__ s_chb = reinterpret_cast<ChanBwd*>(s_ch + ((g.Cin + 3) & ~3));

    // built by the consumer waves, which have nothing else to do before the first stage is published; the producers go straight to their
    // first global loads (the tables' statistics loads + fp64 arithmetic were ~1.5 us in front of every block's first request)
    if (!producer) {
        if (MODE == 0) {
            for (int c = tid; c < g.Cin; c += 256) s_ch[c] = chan_fwd(A.xin, k, c);
            if (tid < CT) {
                const int co = m0 + tid;
                const float b = (co < g.Cout && g.b_off >= 0) ? wk[g.b_off + co] : 0.f;
                s_bias[tid] = b;
            }
        } else {
            for (int c = tid; c < g.Cout; c += 256) s_chb[c] = chan_bwd(A.gin, k, c);
            if constexpr (KS == 1 || FF) { if (A.fga) for (int c = tid; c < g.Cin; c += 256) s_ch[c] = chan_fwd(A.xin, k, c); }
        }
    }
    const bool fuse = MODE == 1 && (KS == 1 || FF) && A.fga != nullptr;
    const bool fuse_sums = fuse && A.fbsums != nullptr;

    // Copy the weight slab of reduction channels [c0, c0+cc) of this sample into wdst[tap][kbase + kk][m] with `nthr` threads.
    //   MODE 0: rows = output channel m, global range ((m0+m)*Cin + c0)*KK + [0, cc*KK),   element -> (kk, tap)
    //   MODE 1: rows = reduction channel kk, range ((c0+kk)*Cin + m0)*KK + [0, mt*KK),     element -> (m, flipped tap)
    // The launcher guarantees Cin % 4 == 0 and w_off % 4 == 0, so every row range is a whole number of aligned float4.
    // Two halves: slab_fetch requests up to NR float4 per thread (items th + j*nthr, j0 <= j < j0 + NR) into registers — all loads of a
    // batch are in flight together — and slab_commit scatters them into LDS.  (One dependent load per loop trip cost a block one memory
    // latency per trip: 9 trips for the weight-stationary slab of the 36->16 layer, and one EXPOSED latency per reduction stage in the
    // chunked mode, which is what the 8x8 / 16x16 layers at the bottom of the hour-glass spent their time on.)
    const float4* __restrict__ w4 = reinterpret_cast<const float4*>(wk + g.w_off);
    auto slab_fetch = [&](auto nr_c, int c0, int cc, int cc4, int th, int nthr, int j0, float4* __restrict__ r) {
        constexpr int NR = decltype(nr_c)::value;
        const int rows = MODE == 0 ? CT : cc4;
        const int valid_rows = MODE == 0 ? mt : cc;
        const int G = ((MODE == 0 ? cc : mtx) * KK) >> 2;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int idx = th + (j0 + j) * nthr;
            const int row = idx / G, gi = idx - row * G;
            r[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < rows * G && row < valid_rows)
                r[j] = w4[(MODE == 0 ? (((m0 + row) * g.Cin + c0) * KK) : (((c0 + row) * g.Cin + m0) * KK)) / 4 + gi];
        }
    };
    auto slab_commit = [&](auto nr_c, int cc, int cc4, int kbase, float* __restrict__ wdst, int th, int nthr, int j0, const float4* __restrict__ r) {
        constexpr int NR = decltype(nr_c)::value;
        const int rows = MODE == 0 ? CT : cc4;
        const int G = ((MODE == 0 ? cc : mtx) * KK) >> 2;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int idx = th + (j0 + j) * nthr;
            if (idx >= rows * G) continue;
            const int row = idx / G, gi = idx - row * G;
            const float w[4] = {r[j].x, r[j].y, r[j].z, r[j].w};
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int rel = gi * 4 + l, q = rel / KK, tap = rel - q * KK;
                if (MODE == 0) wdst[(tap * REDP + kbase + q) * CTP + row] = w[l];
                else wdst[((KK - 1 - tap) * REDP + kbase + row) * CTP + q] = w[l];
            }
        }
    };
    auto slab_pad = [&](int cc, int cc4, int kbase, float* __restrict__ wdst, int th, int nthr) {
        if (MODE == 0 && cc4 > cc)       // pad the last 4-channel step with zero weights
            for (int idx = th; idx < (cc4 - cc) * KK * CT; idx += nthr) {
                const int m = idx % CT, r = idx / CT, kk = cc + r % (cc4 - cc), tap = r / (cc4 - cc);
                wdst[(tap * REDP + kbase + kk) * CTP + m] = 0.f;
            }
        if (MODE == 1 && mt < CT)        // output channels beyond the tensor: keep the (unstored) accumulators finite
            for (int idx = th; idx < KK * cc4 * (CT - mt); idx += nthr) {
                const int m = mt + idx % (CT - mt), r = idx / (CT - mt), kk = r % cc4, tap = r / cc4;
                wdst[(tap * REDP + kbase + kk) * CTP + m] = 0.f;
            }
    };
    // whole range in batches of 8 float4 per thread (weight-stationary slab; chunks too big for the register pipeline)
    auto load_slab = [&](int c0, int cc, int cc4, int kbase, float* __restrict__ wdst, int th, int nthr) {
        const int rows = MODE == 0 ? CT : cc4;
        const int G = ((MODE == 0 ? cc : mtx) * KK) >> 2;
        const int trips = (rows * G + nthr - 1) / nthr;
        constexpr std::integral_constant<int, 8> eight{};
        for (int j0 = 0; j0 < trips; j0 += 8) {
            float4 r[8];
            slab_fetch(eight, c0, cc, cc4, th, nthr, j0, r);
            slab_commit(eight, cc, cc4, kbase, wdst, th, nthr, j0, r);
        }
        slab_pad(cc, cc4, kbase, wdst, th, nthr);
    };

    const int tile_begin = bx * A.tiles_per_block, tile_end = min(A.n_tiles, tile_begin + A.tiles_per_block);
    if (tile_begin >= tile_end) return;
    const int n_iters = (tile_end - tile_begin) * n_chunks;

    if (WS)       // the whole slab, once, by all 8 waves
        load_slab(0, RED, (RED + 3) & ~3, 0, s_w, tid, 512);

    const int H = g.H, W = g.W;
    const int SH = MODE == 0 ? H : g.Ho, SW = MODE == 0 ? W : g.Wo;
    const int SHW = SH * SW;

    if (producer) {
        // ======================= producer waves =======================
        // The second-dispatched half of a 512-thread workgroup loses VALU arbitration to the older (consumer) wave of its SIMD
        // (priority, then age): raise the producers once so their short VALU bursts are not starved by the MFMA stream.
        __builtin_amdgcn_s_setprio(PRODUCER_PRIO);
        const float* __restrict__ xsrc = MODE == 0 ? A.xin.data + (long long)k * A.xin.sstride : A.gin.ga + (long long)k * A.gin.gstride;
        const float* __restrict__ ysrc = (MODE == 1 && A.gin.y) ? A.gin.y + (long long)k * A.gin.ystride : nullptr;
        const int xact = A.xin.act; const float xslope = A.xin.slope;
        // Producer wave pw stages channels {pw*CPW .. pw*CPW+CPW-1} of each 8-channel chunk at positions lane + 64*j of the
        // tile window: the channel is wave-uniform (its BN constants are read once per chunk, not per element) and the LDS
        // address of position p is p itself (PITCH == IN_TW), so an element costs a transform and one ds_write.
        constexpr int CPW = CC / 4;
        constexpr int NPOSW = IN_TH * PITCH;                     // capacity of one LDS channel plane
        constexpr int NPX = (NPOSW + 63) / 64;
        const int pw = wv;
        const int nwin = A.nwin;                                 // positions actually staged (rows x run-time pitch <= NPOSW)
        int goff[FLAT ? NPX : 1];                                           // global offset of position lane + 64*j; -1 = stage a zero
        auto set_tile_s = [&](int tile) {
            const int px0 = 0, py0 = tile * A.rt;
            const int sy0 = MODE == 0 ? py0 * STRIDE - P : py0 - (FF ? P : KS - 1);      // FF: un-padded output domain, halo P on each side
            const int sx0 = MODE == 0 ? px0 * STRIDE - P : px0 - (FF ? P : KS - 1);
#pragma unroll
            for (int j = 0; j < NPX; ++j) {
                const int p = min(lane + 64 * j, nwin - 1), iy = p / A.wpitch, ix = p - iy * A.wpitch;
                int gy = sy0 + iy, gx = sx0 + ix;
                if (MODE == 0) {
                    gy = reflect_idx(gy, H); gx = reflect_idx(gx, W);
                    gy = min(max(gy, 0), H - 1); gx = min(max(gx, 0), W - 1);       // tile overhang: masked at the store
                    goff[j] = gy * W + gx;
                } else {
                    if (g.stride == 2) {
                        const bool ok = gy >= 0 && gx >= 0 && !((gy | gx) & 1) && (gy >> 1) < SH && (gx >> 1) < SW;
                        goff[j] = ok ? (gy >> 1) * SW + (gx >> 1) : -1;                           // zero-stuffed source
                    } else
                        goff[j] = (gy >= 0 && gy < SH && gx >= 0 && gx < SW) ? gy * SW + gx : -1;   // zero padding
                }
            }
        };
        float xr[FLAT ? CPW : 1][FLAT ? NPX : 1], yr[(FLAT && MODE == 1) ? CPW : 1][(FLAT && MODE == 1) ? NPX : 1];
        // Branch-free: every load uses a valid (clamped) address; invalid positions / channels are zeroed at the LDS store.
        auto prefetch_s = [&](int c0) {
#pragma unroll
            for (int i = 0; i < CPW; ++i) {
                const long long cb = (long long)min(c0 + pw * CPW + i, RED - 1) * SHW;
#pragma unroll
                for (int j = 0; j < NPX; ++j) {
                    const long long off = cb + max(goff[j], 0);
                    xr[i][j] = xsrc[off];
                    if (MODE == 1) yr[i][j] = ysrc ? ysrc[off] : 0.f;
                }
            }
        };
        auto store_s = [&](int c0, float* __restrict__ dst) {      // registers -> LDS with the deferred transform
            const int cc = min(CC, RED - c0), cc4 = (cc + 3) & ~3;
#pragma unroll
            for (int i = 0; i < CPW; ++i) {
                const int cl = pw * CPW + i;
                if (cl >= cc4) continue;                          // wave-uniform: channel beyond the padded chunk
                const bool live = cl < cc;
                const int ch = min(c0 + cl, RED - 1);
                ChanFwd kf; ChanBwd kb;
                if (MODE == 0) kf = s_ch[ch]; else kb = s_chb[ch];
#pragma unroll
                for (int j = 0; j < NPX; ++j) {
                    const int p = lane + 64 * j;
                    float v;
                    if (MODE == 0) v = apply_fwd(kf, xr[i][j], xact, xslope);
                    else v = ysrc ? apply_bwd(kb, xr[i][j], yr[i][j]) : xr[i][j];
                    if (!live || (MODE == 1 && goff[j] < 0)) v = 0.f;
                    if (p < nwin) dst[cl * PLANE + p] = v;
                }
            }
        };
        // ---- rectangular tiles: aligned float4 staging ----
        // Item q = lane + 64*j of a channel is float4 column v of window row iy (NV4 columns per row).  A float4 is wholly
        // inside or wholly outside the image (tile origins and image widths are multiples of 4); reflection only ever needs
        // one element of an outside float4 (column -1 <- x[1], column W <- x[W-2]), taken from the neighbouring inside one.
        constexpr int NV4 = PH ? 6 : PITCH / 4, NITEM = (PH ? TH / 2 + 2 : IN_TH) * NV4, NV = (NITEM + 63) / 64;
        int voff[FLAT ? 1 : NV];          // global element offset of the float4 to load (always valid), with flags in the low 2 bits:
                                          // 1 = left-reflected (keep .y as column 3), 2 = right-reflected (.z as column 0), 3 = stage zeros
        float4 xv[FLAT ? 1 : CPW][FLAT ? 1 : NV], yv[(!FLAT && MODE == 1) ? CPW : 1][(!FLAT && MODE == 1) ? NV : 1];
        auto set_tile_v = [&](int tile) {
            const int px0 = (tile % A.tiles_x) * TW, py0 = (tile / A.tiles_x) * TH;
            const int sy0 = PH ? py0 / 2 - 1 : (MODE == 0 ? py0 * STRIDE - P : py0 - (FF ? P : KS - 1));
            const int ax0 = (PH ? px0 / 2 : (MODE == 0 ? px0 * STRIDE : px0)) - Cfg::HALO4;            // aligned first column of the LDS row
            const bool stuffed = !PH && MODE == 1 && g.stride == 2;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int q = min(lane + 64 * j, NITEM - 1), iy = q / NV4, v = q - iy * NV4;
                int gy = sy0 + iy, gx = ax0 + 4 * v, flag = 0;
                if (MODE == 0) {
                    gy = reflect_idx(gy, H); gy = min(max(gy, 0), H - 1);               // tile overhang: masked at the store
                    if (gx < 0) { flag = 1; gx = 0; } else if (gx >= W) { flag = gx == W ? 2 : 3; gx = W - 4; }
                    voff[j] = (gy * W + gx) | flag;                                     // gx % 4 == 0, W % 4 == 0: low bits are free
                } else if (stuffed) {
                    // zero-stuffed gradient G[r][c] = dy[r/2][c/2] (r, c even): window columns gx..gx+3 <- (s[gx/2], 0, s[gx/2+1], 0)
                    const bool ok = gy >= 0 && !(gy & 1) && (gy >> 1) < SH && gx >= 0 && (gx >> 1) < SW;      // SW even: the pair is inside together
                    voff[j] = ok ? (((gy >> 1) * SW + (gx >> 1)) << 2) : 3;            // 8-byte aligned pair; offset kept in bits 2..
                } else {
                    const bool ok = gy >= 0 && gy < SH && gx >= 0 && gx < SW;
                    voff[j] = ok ? (gy * SW + gx) : 3;
                }
            }
        };
        auto prefetch_v = [&](int c0) {
            const bool stuffed = !PH && MODE == 1 && g.stride == 2;
#pragma unroll
            for (int i = 0; i < CPW; ++i) {
                const long long cb = (long long)min(c0 + pw * CPW + i, RED - 1) * SHW;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    if (stuffed) {
                        const long long off = cb + (voff[j] >> 2);
                        const f2a a = *reinterpret_cast<const f2a*>(xsrc + off);
                        xv[i][j] = make_float4(a.x, 0.f, a.y, 0.f);
                        if (MODE == 1) { if (ysrc) { const f2a b = *reinterpret_cast<const f2a*>(ysrc + off); yv[i][j] = make_float4(b.x, 0.f, b.y, 0.f); } else yv[i][j] = make_float4(0.f, 0.f, 0.f, 0.f); }
                    } else {
                        const long long off = cb + (voff[j] & ~3);
                        xv[i][j] = *reinterpret_cast<const float4*>(xsrc + off);
                        if (MODE == 1) yv[i][j] = ysrc ? *reinterpret_cast<const float4*>(ysrc + off) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        };
        auto store_v = [&](int c0, float* __restrict__ dst) {    // registers -> LDS with the deferred transform
            const int cc = min(CC, RED - c0), cc4 = (cc + 3) & ~3;
            const bool stuffed = !PH && MODE == 1 && g.stride == 2;
#pragma unroll
            for (int i = 0; i < CPW; ++i) {
                const int cl = pw * CPW + i;
                if (cl >= cc4) continue;                          // wave-uniform: channel beyond the padded chunk
                const bool live = cl < cc;
                const int ch = min(c0 + cl, RED - 1);
                ChanFwd kf; ChanBwd kb;
                if (MODE == 0) kf = s_ch[ch]; else kb = s_chb[ch];
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int q = lane + 64 * j;
                    const int flag = voff[j] & 3;
                    float e[4] = {xv[i][j].x, xv[i][j].y, xv[i][j].z, xv[i][j].w};
                    if (MODE == 0) {
                        apply_fwd4(kf, e, xact, xslope);
                        if (flag == 1) { e[3] = e[1]; }                      // column -1 <- x[1]   (columns -4..-2 are never read)
                        else if (flag == 2) { e[0] = e[2]; }                 // column W  <- x[W-2] (columns W+1.. only feed masked outputs)
                    } else {
                        if (ysrc) {
                            const float yy[4] = {yv[i][j].x, yv[i][j].y, yv[i][j].z, yv[i][j].w};
                            apply_bwd4(kb, e, yy);
                        }
                        if (stuffed) { e[1] = 0.f; e[3] = 0.f; }
                    }
                    if (!live || (MODE == 1 && (stuffed ? voff[j] == 3 : flag == 3))) { e[0] = 0.f; e[1] = 0.f; e[2] = 0.f; e[3] = 0.f; }
                    if (64 * (j + 1) <= NITEM || q < NITEM) *reinterpret_cast<float4*>(dst + cl * PLANE + 4 * q) = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
        };
        auto set_tile = [&](int tile) { if constexpr (FLAT) set_tile_s(tile); else set_tile_v(tile); };
        auto prefetch = [&](int c0) { if constexpr (FLAT) prefetch_s(c0); else prefetch_v(c0); };
        auto store = [&](int c0, float* __restrict__ dst) { if constexpr (FLAT) store_s(c0, dst); else store_v(c0, dst); };
        auto chunk_of = [&](int it, int& tile, int& c0) { tile = tile_begin + it / n_chunks; c0 = (it % n_chunks) * CC; };

        // chunked weights (!WS): the slab of a stage travels global -> registers -> LDS one stage ahead like the activations when it is
        // at most SLAB_NR float4 per producer thread; bigger chunks (many output fragments x 32-channel stages) load in place, batched
        constexpr int SLAB_ITEMS = (CTX / 4) * CC * KK;                      // float4 of one full chunk
        constexpr int SLAB_NR = (SLAB_ITEMS + 255) / 256;
        constexpr bool SLAB_PIPE = !WS && SLAB_NR <= 12;
        constexpr std::integral_constant<int, SLAB_PIPE ? SLAB_NR : 1> snr{};
        float4 wreg[SLAB_PIPE ? SLAB_NR : 1];
        auto wfetch = [&](int c0) { if constexpr (SLAB_PIPE) { const int cc = min(CC, RED - c0); slab_fetch(snr, c0, cc, (cc + 3) & ~3, t, 256, 0, wreg); } };
        auto wstore = [&](int c0, float* __restrict__ wdst) {
            if constexpr (!WS) {
                const int cc = min(CC, RED - c0), cc4 = (cc + 3) & ~3;
                if constexpr (SLAB_PIPE) { slab_commit(snr, cc, cc4, 0, wdst, t, 256, 0, wreg); slab_pad(cc, cc4, 0, wdst, t, 256); }
                else load_slab(c0, cc, cc4, 0, wdst, t, 256);
            }
        };
        // ---- PEPI: the fold of a dumped tile, item idx = t + 256*j of [channel][row][float4 column]; the channel of (wave, j) is wave-uniform.
        //      Three parts (items j == part mod 3), one per stage 0..2 of the next tile (n_chunks >= 4: launcher), so a part's raw-x loads
        //      are few registers and fly while the stage is staged.
        constexpr int NIT = PEPI ? (CTX * TH * 8 + 255) / 256 : 1, NIT3 = (NIT + 2) / 3;
        float fsum[NIT], fxs[NIT]; float4 fxr[NIT3];
#pragma unroll
        for (int j = 0; j < NIT; ++j) { fsum[j] = 0.f; fxs[j] = 0.f; }
        auto fold_fetch = [&](int tile, auto part_c) {       // request the raw x of the part's items
            constexpr int PART = decltype(part_c)::value;
            if constexpr (PEPI) {
                if (!fuse_sums) return;
                const int px0 = (tile % A.tiles_x) * TW, py0 = (tile / A.tiles_x) * TH;
                const float* __restrict__ xq = A.xin.data + (long long)k * A.xin.sstride + (long long)m0 * H * W;
#pragma unroll
                for (int j = PART; j < NIT; j += 3) {
                    const int idx = t + 256 * j, ch = idx / (TH * 8), pr = py0 + ((idx >> 3) % TH), pc = px0 + 4 * (idx & 7);
                    fxr[j / 3] = (ch < mtx && pr < H && pc < W) ? *reinterpret_cast<const float4*>(xq + ch * H * W + pr * W + pc) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        auto fold_do = [&](int tile, auto part_c) {
            constexpr int PART = decltype(part_c)::value;
            if constexpr (PEPI) {
                const int px0 = (tile % A.tiles_x) * TW, py0 = (tile / A.tiles_x) * TH;
                float* __restrict__ o = A.fga + (long long)k * A.fga_sstride + (long long)m0 * H * W;
#pragma unroll
                for (int j = PART; j < NIT; j += 3) {
                    const int idx = t + 256 * j, ch = idx / (TH * 8), row = (idx >> 3) % TH, pr = py0 + row, pc = px0 + 4 * (idx & 7);
                    if (ch < mtx && pr < H && pc < W) {
                        const float4 v = *reinterpret_cast<const float4*>(&s_out[ch * OP + row * 32 + 4 * (idx & 7)]);
                        float dd[4] = {v.x, v.y, v.z, v.w};
                        if (fuse_sums) {
                            const float yy[4] = {fxr[j / 3].x, fxr[j / 3].y, fxr[j / 3].z, fxr[j / 3].w};
                            const ChanFwd cf = s_ch[m0 + ch];
#pragma unroll
                            for (int l = 0; l < 4; ++l) {
                                const float vv = __builtin_fmaf(yy[l] - cf.mean, cf.scale, cf.beta);
                                if (A.xin.act && !(vv > 0.f)) dd[l] *= A.xin.slope;
                                fsum[j] += dd[l]; fxs[j] = __builtin_fmaf(dd[l], (yy[l] - cf.mean) * cf.rstd, fxs[j]);
                            }
                        }
                        *reinterpret_cast<float4*>(o + ch * H * W + pr * W + pc) = make_float4(dd[0], dd[1], dd[2], dd[3]);
                    }
                }
            }
        };
        constexpr std::integral_constant<int, 0> p0{}; constexpr std::integral_constant<int, 1> p1{}; constexpr std::integral_constant<int, 2> p2{};
        int ptile, pc0;
        chunk_of(0, ptile, pc0);
        set_tile(ptile); prefetch(pc0); wfetch(pc0);
        __syncthreads();                                  // (S0) channel constants / bias / WS slab visible
        store(pc0, s_x[0]);
        wstore(pc0, s_w);
        if (n_iters > 1) { int nt, nc; chunk_of(1, nt, nc); if (nt != ptile) { set_tile(nt); ptile = nt; } prefetch(nc); wfetch(nc); }
        lds_barrier();                                    // (A) chunk 0 published
        for (int it = 0; it < n_iters; ++it) {
            // PEPI: stage ci of a tile carries part ci of the PREVIOUS tile's fold (parts 0 .. n_chunks - 2)
            const int fci = it % n_chunks, ftile = tile_begin + it / n_chunks - 1;
            const bool fold_now = PEPI && it >= n_chunks && fci < 3;
            if (fold_now) { if (fci == 0) fold_fetch(ftile, p0); else if (fci == 1) fold_fetch(ftile, p1); else fold_fetch(ftile, p2); }
            if (it + 1 < n_iters) {
                int nt, nc; chunk_of(it + 1, nt, nc);
                store(nc, s_x[(it + 1) & 1]);
                wstore(nc, s_w + ((it + 1) & 1) * WCHUNK);
                if (it + 2 < n_iters) { int n2, c2; chunk_of(it + 2, n2, c2); if (n2 != ptile) { set_tile(n2); ptile = n2; } prefetch(c2); wfetch(c2); }
            }
            if (fold_now) { if (fci == 0) fold_do(ftile, p0); else if (fci == 1) fold_do(ftile, p1); else fold_do(ftile, p2); }
            lds_barrier();
        }
        if constexpr (PEPI) {       // the block's last tile, then this thread's BN-backward partials: the channel of (wave, j) is wave-uniform
            fold_fetch(tile_end - 1, p0); fold_do(tile_end - 1, p0); fold_fetch(tile_end - 1, p1); fold_do(tile_end - 1, p1); fold_fetch(tile_end - 1, p2); fold_do(tile_end - 1, p2);
            if (fuse_sums) {
#pragma unroll
                for (int j = 0; j < NIT; ++j) {
                    float a = fsum[j], b = fxs[j];
#pragma unroll
                    for (int o2 = 32; o2 > 0; o2 >>= 1) { a += __shfl_xor(a, o2, 64); b += __shfl_xor(b, o2, 64); }
                    const int ch = (t + 256 * j) / (TH * 8);          // same for all lanes of the wave (64 | TH * 8)
                    if (lane == 0 && ch < mtx) {
                        double* dst = A.fbsums + ((long long)k * g.Cin + m0 + ch) * 2;
                        atomicAdd(dst, (double)a); atomicAdd(dst + 1, (double)b);
                    }
                }
            }
        }
        if ((MODE == 0 && A.out.stats != nullptr) || (fuse_sums && !PEPI)) __syncthreads();        // (Z) consumers publish their BN partial sums
    } else {
        // ======================= consumer waves =======================
        int boff[FLAT ? 1 : NF], boffk[FLAT ? NF : 1][FLAT ? KS : 1], frc[FLAT ? NF : 1];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if (FLAT) {      // pixel q of the tile (row-major over rt full rows of the domain) -> (r, c); fixed for the whole kernel
                const int q = (wv * NF + f) * 16 + l15, r = q / A.ow, c = q - r * A.ow;
                const bool ok = r < A.rt;
                frc[f] = ok ? (r << 16 | c) : -1;
                const int base = l4 * PLANE + (ok ? r * STRIDE * A.wpitch + c * STRIDE : 0);
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) boffk[f][ky] = base + ky * A.wpitch;
            } else {
                const int row = wv * (TH / 4) + (f >> 1), col = (f & 1) * 16 + l15;
                if constexpr (PH) boff[f] = l4 * PLANE + (wv * (TH / 8)) * 24 + l15 + 3;      // window row (R + r + ky)/2, R = wv*TH/4 even; column l15 + (parity + kx)/2 + 3
                else boff[f] = l4 * PLANE + row * STRIDE * PITCH + col * STRIDE + (Cfg::HALO4 ? Cfg::HALO4 - ((MODE == 0 || FF) ? P : KS - 1) : 0);
            }
        }
        const int aoff = l4 * CTP + l15;
        const int wtap = REDP * CTP;
        f32x4 acc[MF][NF];
        f32x4 accx[REM ? NF : 1];                          // REM: the 4 extra channels, this lane's k-slice
        const int xoff = CT + (l15 & 3) - l15;             // from aoff to the extra channels' weights of reduction channel l4
        const bool do_stats = (MODE == 0 && A.out.stats != nullptr) || (fuse_sums && !PEPI);
        // BN statistics of this wave's outputs: float partial sums per tile, folded into the wave's fp64 slots in LDS
        if (do_stats) for (int q = lane; q < CTX; q += 64) { s_red[wv][q][0] = 0.0; s_red[wv][q][1] = 0.0; }
        __syncthreads();                                  // (S0)
        lds_barrier();                                    // (A)
        for (int it = 0; it < n_iters; ++it) {
            const int tile = tile_begin + it / n_chunks, ci = it % n_chunks, c0 = ci * CC;
            const int cc = min(CC, RED - c0), cc4 = (cc + 3) & ~3;
            // FF: origin of the tile in the un-padded input domain and whether it holds a reflected row (1, H-2) / column (1, W-2)
            const int tpx0 = FLAT ? 0 : (tile % A.tiles_x) * TW, tpy0 = FLAT ? tile * A.rt : (tile / A.tiles_x) * TH;
            const int trows = FLAT ? A.rt : TH;
            const bool spr = FF && ((tpy0 <= 1 && 1 < tpy0 + trows) || (tpy0 <= g.H - 2 && g.H - 2 < tpy0 + trows));
            const bool spc = FF && (FLAT || tpx0 == 0 || (tpx0 <= g.W - 2 && g.W - 2 < tpx0 + TW));
            if (ci == 0) {
#pragma unroll
                for (int a = 0; a < MF; ++a)
#pragma unroll
                    for (int b = 0; b < NF; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (REM) {
#pragma unroll
                    for (int b = 0; b < NF; ++b) accx[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            // Fused fold on rectangular tiles: the epilogue needs the raw input tensor x (LeakyReLU' and the x-hat of the BN-backward sums)
            // at the tile's output pixels.  All its float4 are requested together — before the MFMAs of the tile's last chunk when the
            // registers allow (YPRE_EARLY), else at the head of the epilogue — instead of one dependent global load per 16-channel row pair
            // between the epilogue's wave barriers (one exposed HBM latency per row pair: 0.199 -> 0.276 ms on up_9).
#ifndef MFVI_YPRE_EARLY_MAX
#define MFVI_YPRE_EARLY_MAX 12
#endif
            constexpr bool YPRE = MODE == 1 && !FLAT && !PEPI && (KS == 1 || FF);
            constexpr bool YPRE_EARLY = YPRE && MF * NF <= MFVI_YPRE_EARLY_MAX;
            float4 ypre[YPRE ? MF : 1][YPRE ? NF / 2 : 1][2];
            auto load_ypre = [&]() {
                if constexpr (YPRE) {
                    const int Hq = FF ? g.H : g.H + 2 * P, Wq = FF ? g.W : g.W + 2 * P, HWq = Hq * Wq;
                    const float* __restrict__ xq = A.xin.data + (long long)k * A.xin.sstride + (long long)m0 * HWq;
                    const int ech_ = lane >> 3, ev4_ = lane & 7;
#pragma unroll
                    for (int i = 0; i < MF; ++i)
#pragma unroll
                        for (int rp = 0; rp < NF / 2; ++rp)
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const int ch = ech_ + 8 * u, pr = tpy0 + wv * (TH / 4) + rp, pc = tpx0 + 4 * ev4_;
                                const bool ok = i * 16 + ch < mt && pr < Hq && pc < Wq;
                                ypre[i][rp][u] = ok ? *reinterpret_cast<const float4*>(xq + (i * 16 + ch) * HWq + pr * Wq + pc) : make_float4(0.f, 0.f, 0.f, 0.f);
                            }
                }
            };
            if constexpr (YPRE_EARLY) { if (fuse_sums && ci == n_chunks - 1) load_ypre(); }
            // ---- MFMA: fully unrolled k-steps (tap x 4-channel step), operand fragments double-buffered in registers
            //      so the LDS reads of step q+1 are in flight while the matrix core runs step q ----
            const float* __restrict__ sx = s_x[it & 1];
            const float* __restrict__ wq = (WS ? s_w + c0 * CTP : s_w + (it & 1) * WCHUNK) + aoff;
            auto run_sp = [&](auto steps_c, auto spr_c, auto spc_c, int sbase) {          // STEPS 4-channel steps of the chunk, starting at step sbase
                constexpr int STEPS = decltype(steps_c)::value, NQ = KK * STEPS;
                constexpr bool SPR = decltype(spr_c)::value, SPC = decltype(spc_c)::value;
                constexpr int SG = 2;
                // Rectangular tiles: a fragment is 16 pixels of ONE row, so the row part of the reflection adjoint moves to the weight operand:
                //   row 1:   A'(ky=0) = A(ky=0) + A(ky=2)      row H-2:   A'(ky=2) = A(ky=2) + A(ky=0)
                // (the same products, summed on the A side; the corner term comes out of A' x B').  One extra A read per border-row k-step
                // instead of up to two extra B reads per fragment; the column part stays on B.
                constexpr bool AROW = FF && SPR && !FLAT;
                // The k-steps of a stage are ONE straight-line block, software-pipelined at instruction level and pinned with scheduling
                // barriers: after every MFMA of k-step q, one LDS read of k-step q+1 (into the other register set), so every operand is
                // requested a whole k-step (MF*NF MFMAs) before its first use and the wave never issues more than one non-matrix instruction
                // between two MFMAs.  Left to the machine scheduler the reads sank next to their uses (a dozen `s_waitcnt lgkmcnt(0)` per
                // stage) and the address / wait instructions issued back to back while the matrix pipe drained — harmless with three
                // consumer waves per SIMD, but 40 % of the matrix rate with one (backward-data at one block per CU; the same change took the
                // backward-weight kernel from 240 to 198 us).  Reads that patch an operand (reflection adjoint on B) land in their own
                // registers and are folded in by `finish` at the head of the k-step that uses them.
                constexpr bool BCOL = FF && SPC, BROW = FF && SPR && FLAT;
                float a[SG][MF], b[SG][NF], ax[SG], ao[SG][AROW ? MF : 1], axo[SG], e1[SG][BCOL ? NF : 1], e0[SG][BROW ? NF : 1], e2[SG][(BCOL && BROW) ? NF : 1];
                auto bidx = [&](int f, int st_, int ky, int kx) -> int {
                    if constexpr (PH) return (sbase + st_) * 4 * PLANE + boff[f] + (((f >> 1) + ky) >> 1) * 24 + (((f & 1) + kx) >> 1);
                    else if constexpr (FLAT) return (sbase + st_) * 4 * PLANE + boffk[f][ky] + kx;
                    else return (sbase + st_) * 4 * PLANE + boff[f] + ky * PITCH + kx;
                };
                // issue reads number [jlo, jhi) of k-step q into register set s; returns how many reads the k-step has
                auto load_range = [&](int q, int s_, int jlo, int jhi) -> int {
                    const int tap = q / STEPS, st_ = q % STEPS, ky = tap / KS, kx = tap % KS;
                    int j = 0;
                    auto take = [&]() { const bool t_ = j >= jlo && j < jhi; ++j; return t_; };
#pragma unroll
                    for (int i = 0; i < MF; ++i) if (take()) a[s_][i] = wq[tap * wtap + (sbase + st_) * 4 * CTP + i * 16];
                    if constexpr (REM) { if (take()) ax[s_] = wq[tap * wtap + (sbase + st_) * 4 * CTP + xoff]; }
                    if constexpr (AROW) {
                        if (ky != 1) {
                            const int tapo = (2 - ky) * KS + kx;
#pragma unroll
                            for (int i = 0; i < MF; ++i) if (take()) ao[s_][i] = wq[tapo * wtap + (sbase + st_) * 4 * CTP + i * 16];
                            if constexpr (REM) { if (take()) axo[s_] = wq[tapo * wtap + (sbase + st_) * 4 * CTP + xoff]; }
                        }
                    }
#pragma unroll
                    for (int f = 0; f < NF; ++f)
                        if (!PH || ((((f >> 1) + ky) & 1) == 0 && (((f & 1) + kx) & 1) == 0)) { if (take()) b[s_][f] = sx[bidx(f, st_, ky, kx)]; }
                    if constexpr (BCOL) {
                        if (kx != 1) {
#pragma unroll
                            for (int f = 0; f < NF; ++f) if (take()) e1[s_][f] = sx[bidx(f, st_, ky, 2 - kx)];
                        }
                    }
                    if constexpr (BROW) {
                        if (ky != 1) {
#pragma unroll
                            for (int f = 0; f < NF; ++f) if (take()) e0[s_][f] = sx[bidx(f, st_, 2 - ky, kx)];
                        }
                    }
                    if constexpr (BCOL && BROW) {
                        if (ky != 1 && kx != 1) {
#pragma unroll
                            for (int f = 0; f < NF; ++f) if (take()) e2[s_][f] = sx[bidx(f, st_, 2 - ky, 2 - kx)];
                        }
                    }
                    return j;
                };
                // adjoint of the reflection padding on the B operand (see the kernel's header); flags are 1.0 on the lanes concerned
                auto finish = [&](int q, int s_) {
                    if constexpr (BCOL || BROW) {
                        const int tap = q / STEPS, ky = tap / KS, kx = tap % KS;
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
                            int rowf, colf;
                            if constexpr (FLAT) { rowf = tpy0 + (frc[f] >> 16); colf = frc[f] & 0xffff; }
                            else { rowf = tpy0 + wv * (TH / 4) + (f >> 1); colf = tpx0 + (f & 1) * 16 + l15; }
                            const float fr = (BROW && ky != 1 && rowf == (ky == 2 ? 1 : g.H - 2)) ? 1.f : 0.f;
                            const float fc = (BCOL && kx != 1 && colf == (kx == 2 ? 1 : g.W - 2)) ? 1.f : 0.f;
                            float e = 0.f;
                            if constexpr (BROW) { if (ky != 1) e = __builtin_fmaf(fr, e0[s_][f], e); }
                            if constexpr (BCOL) { if (kx != 1) e = __builtin_fmaf(fc, e1[s_][f], e); }
                            if constexpr (BCOL && BROW) { if (ky != 1 && kx != 1) e = __builtin_fmaf(fr * fc, e2[s_][f], e); }
                            b[s_][f] += e;
                        }
                    }
                };
                load_range(0, 0, 0, 1 << 20);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int s_ = q % SG, sn = (q + 1) % SG;
                    const int n_next = q + 1 < NQ ? load_range(q + 1, sn, 0, 0) : 0;
                    int m = 0;
                    finish(q, s_);
                    const int tapq = q / STEPS, kyq = tapq / KS, kxq = tapq % KS;
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        // fragment f's row takes the other border tap's weights too (wave-uniform flag)
                        float frw = 0.f;
                        if constexpr (AROW) { if (kyq != 1) frw = (tpy0 + wv * (TH / 4) + (f >> 1) == (kyq == 0 ? 1 : g.H - 2)) ? 1.f : 0.f; }
#pragma unroll
                        for (int i = 0; i < MF; ++i)
                            if (!PH || ((((f >> 1) + kyq) & 1) == 0 && (((f & 1) + kxq) & 1) == 0)) {
                                float av = a[s_][i];
                                if constexpr (AROW) { if (kyq != 1) av = __builtin_fmaf(frw, ao[s_][i], av); }
                                if (m < n_next) load_range(q + 1, sn, m, m + 1);
                                ++m;
                                acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[s_][f], acc[i][f], 0, 0, 0);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        if constexpr (REM) {
                            float av = ax[s_];
                            if constexpr (AROW) { if (kyq != 1) av = __builtin_fmaf(frw, axo[s_], av); }
                            if (m < n_next) load_range(q + 1, sn, m, m + 1);
                            ++m;
                            if (rem_blk) accx[f] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, b[s_][f], accx[f], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (m < n_next) load_range(q + 1, sn, m, 1 << 20);
                }
            };
            auto run = [&](auto steps_c, int sbase) {          // border tiles take the variant with the reflected rows / columns they hold
                constexpr std::false_type no{}; constexpr std::true_type yes{};
                if constexpr (FF) {
                    // rectangular tiles: ONE block of code with the border patches always woven in (flags are zero in interior tiles; the four
                    // separately unrolled variants cost ~60 VGPRs of live state between them and the second block per CU with it)
                    if constexpr (!FLAT) run_sp(steps_c, yes, yes, sbase);
                    else if (spr && spc) run_sp(steps_c, yes, yes, sbase);
                    else if (spr) run_sp(steps_c, yes, no, sbase);
                    else if (spc) run_sp(steps_c, no, yes, sbase);
                    else run_sp(steps_c, no, no, sbase);
                } else run_sp(steps_c, no, no, sbase);
            };
            if constexpr (CC == 8) {
                if (cc4 == 8) run(std::integral_constant<int, 2>{}, 0); else run(std::integral_constant<int, 1>{}, 0);
            } else if constexpr (KS == 1) {      // big stages of a 1x1 layer: one tap, so any grouping keeps the channel order of the sum
                int sb = 0, left = cc4 >> 2;
                if (left == CC / 4) { run(std::integral_constant<int, CC / 4>{}, 0); left = 0; }
                for (; left >= 4; left -= 4, sb += 4) run(std::integral_constant<int, 4>{}, sb);
                for (; left >= 1; left -= 1, sb += 1) run(std::integral_constant<int, 1>{}, sb);
            } else {      // 3x3: MFVI_CC0 channels at a time, taps inside — every variant (4-channel stages, big stages) accumulates in this order
                int sb = 0, left = cc4 >> 2;
                if constexpr (MFVI_CC0 >= 8) { for (; left >= 2; left -= 2, sb += 2) run(std::integral_constant<int, 2>{}, sb); }
                for (; left >= 1; left -= 1, sb += 1) run(std::integral_constant<int, 1>{}, sb);
            }

            if (ci == n_chunks - 1) {
              if constexpr (PEPI) {
                // ---- hand the tile to the producers: accumulators -> s_out[channel][row][col] (D layout: column (pixel) = lane & 15,
                //      row (channel) = (lane >> 4) * 4 + reg); they fold and store it while this wave runs the next tile ----
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            s_out[(i * 16 + l4 * 4 + r) * OP + (wv * (TH / 4) + (f >> 1)) * 32 + (f & 1) * 16 + l15] = acc[i][f][r];
                if constexpr (REM) {
                    if (rem_blk) {      // the 4 extra channels: add the four k-slices (lanes l15 + 16 * l4) first
#pragma unroll
                        for (int f = 0; f < NF; ++f)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float v = accx[f][r]; v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                                if (l4 == 0) s_out[(CT + r) * OP + (wv * (TH / 4) + (f >> 1)) * 32 + (f & 1) * 16 + l15] = v;
                            }
                    }
                }
              } else {
                // ---- epilogue ----  D layout: column (pixel) = lane & 15, row (channel) = (lane >> 4) * 4 + reg.
                // 32-bit element offsets from a wave-uniform base keep the address math out of the VGPR budget.
                const int px0 = FLAT ? 0 : (tile % A.tiles_x) * TW, py0 = FLAT ? tile * A.rt : (tile / A.tiles_x) * TH;
                // Rectangular tiles go through a per-wave LDS transpose so every global store instruction writes whole 128-byte
                // rows (8 lanes x float4 per channel row): the direct D-layout stores are 64-byte partial lines and cost ~550
                // cycles per instruction here (17 k cycles per tile).
                bool done = false;
                if constexpr (!FLAT) {
                    float (*ep)[EPP] = s_ep[wv];
                    const int ech = lane >> 3, ev4 = lane & 7;                  // read-back item: channel ech (+8), float4 column ev4
                    if (MODE == 0 && A.vec_out) {
                        const int HWo = g.Ho * g.Wo;
                        float* __restrict__ yout = A.out.data + (long long)k * A.out.sstride + (long long)m0 * HWo;
                        float fs[MF][4], fq[MF][4];
#pragma unroll
                        for (int i = 0; i < MF; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) { fs[i][r] = 0.f; fq[i][r] = 0.f; }
#pragma unroll
                        for (int i = 0; i < MF; ++i)
#pragma unroll
                            for (int rp = 0; rp < NF / 2; ++rp) {
                                const int oy = py0 + wv * (TH / 4) + rp;
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    const bool inside = oy < g.Ho && px0 + h * 16 + l15 < g.Wo;
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        const int ml = i * 16 + l4 * 4 + r;
                                        const float v = acc[i][2 * rp + h][r] + s_bias[ml];
                                        ep[l4 * 4 + r][h * 16 + l15] = v;
                                        if (inside && ml < mt) { fs[i][r] += v; fq[i][r] = __builtin_fmaf(v, v, fq[i][r]); }
                                    }
                                }
                                __builtin_amdgcn_wave_barrier();
#pragma unroll
                                for (int u = 0; u < 2; ++u) {
                                    const int ch = ech + 8 * u, ox = px0 + 4 * ev4;
                                    if (i * 16 + ch < mt && oy < g.Ho && ox < g.Wo)
                                        *reinterpret_cast<float4*>(yout + (i * 16 + ch) * HWo + oy * g.Wo + ox) = *reinterpret_cast<const float4*>(&ep[ch][4 * ev4]);
                                }
                                __builtin_amdgcn_wave_barrier();
                            }
                        if (do_stats) {
#pragma unroll
                            for (int i = 0; i < MF; ++i)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    float a = fs[i][r], b = fq[i][r];
#pragma unroll
                                    for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                                    if (l15 == 0) { s_red[wv][i * 16 + l4 * 4 + r][0] += (double)a; s_red[wv][i * 16 + l4 * 4 + r][1] += (double)b; }
                                }
                        }
                        done = true;
                    }
                    if (MODE == 1) {
                        const int Hp = FF ? g.H : g.H + 2 * P, Wp = FF ? g.W : g.W + 2 * P, HWp = Hp * Wp;
                        float* __restrict__ o = fuse ? A.fga + (long long)k * A.fga_sstride + (long long)m0 * HWp
                                                     : A.dxp + (long long)k * A.dxp_sstride + (long long)m0 * HWp;
                        const float* __restrict__ xraw = fuse_sums ? A.xin.data + (long long)k * A.xin.sstride + (long long)m0 * HWp : nullptr;
                        float fsg[MF][2], fsx[MF][2];
#pragma unroll
                        for (int i = 0; i < MF; ++i) { fsg[i][0] = 0.f; fsg[i][1] = 0.f; fsx[i][0] = 0.f; fsx[i][1] = 0.f; }
                        if constexpr (YPRE && !YPRE_EARLY) { if (fuse_sums) load_ypre(); }
#pragma unroll
                        for (int i = 0; i < MF; ++i)
#pragma unroll
                            for (int rp = 0; rp < NF / 2; ++rp) {
                                const int pr = py0 + wv * (TH / 4) + rp;
#pragma unroll
                                for (int h = 0; h < 2; ++h)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) ep[l4 * 4 + r][PH ? 2 * l15 + h : h * 16 + l15] = acc[i][2 * rp + h][r];
                                __builtin_amdgcn_wave_barrier();
#pragma unroll
                                for (int u = 0; u < 2; ++u) {
                                    const int ch = ech + 8 * u, pc = px0 + 4 * ev4;
                                    if ((KS == 1 || FF) && fuse) {
                                        // 1x1 / un-padded 3x3 domain, single consumer: (pr, pc) IS the input pixel -> LeakyReLU' and the BN-backward sums here
                                        if (i * 16 + ch < mt && pr < Hp && pc < Wp) {
                                            const int ofs = (i * 16 + ch) * HWp + pr * Wp + pc;           // Wp == W, a multiple of 4: aligned float4
                                            const float4 v = *reinterpret_cast<const float4*>(&ep[ch][4 * ev4]);
                                            float dd[4] = {v.x, v.y, v.z, v.w};
                                            if (fuse_sums) {
                                                const float4 y4 = ypre[YPRE ? i : 0][YPRE ? rp : 0][u];
                                                const float yy[4] = {y4.x, y4.y, y4.z, y4.w};
                                                const ChanFwd cf = s_ch[m0 + i * 16 + ch];
#pragma unroll
                                                for (int l = 0; l < 4; ++l) {
                                                    const float vv = __builtin_fmaf(yy[l] - cf.mean, cf.scale, cf.beta);
                                                    if (A.xin.act && !(vv > 0.f)) dd[l] *= A.xin.slope;
                                                    fsg[i][u] += dd[l]; fsx[i][u] = __builtin_fmaf(dd[l], (yy[l] - cf.mean) * cf.rstd, fsx[i][u]);
                                                }
                                            }
                                            *reinterpret_cast<float4*>(o + ofs) = make_float4(dd[0], dd[1], dd[2], dd[3]);
                                        }
                                    } else
                                    if (i * 16 + ch < mt && pr < Hp && pc < Wp) {
                                        float* dst = o + (i * 16 + ch) * HWp + pr * Wp + pc;
                                        const float4 v = *reinterpret_cast<const float4*>(&ep[ch][4 * ev4]);
                                        if (pc + 3 < Wp) { f4u w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; *reinterpret_cast<f4u*>(dst) = w; }
                                        else { dst[0] = v.x; if (pc + 1 < Wp) dst[1] = v.y; if (pc + 2 < Wp) dst[2] = v.z; }
                                    }
                                }
                                __builtin_amdgcn_wave_barrier();
                            }
                        if (fuse_sums) {        // the 8 lanes that share a channel (ech) -> one partial per channel and wave, into its fp64 slot
#pragma unroll
                            for (int i = 0; i < MF; ++i)
#pragma unroll
                                for (int u = 0; u < 2; ++u) {
                                    float a = fsg[i][u], b = fsx[i][u];
#pragma unroll
                                    for (int o2 = 4; o2 > 0; o2 >>= 1) { a += __shfl_xor(a, o2, 64); b += __shfl_xor(b, o2, 64); }
                                    if (ev4 == 0) { s_red[wv][i * 16 + ech + 8 * u][0] += (double)a; s_red[wv][i * 16 + ech + 8 * u][1] += (double)b; }
                                }
                        }
                        done = true;
                    }
                }
                if (!done) {
                if (MODE == 0) {
                    const int HWo = g.Ho * g.Wo;
                    float* __restrict__ yout = A.out.data + (long long)k * A.out.sstride + (long long)m0 * HWo;
                    const int rowbase = l4 * 4 * HWo;
                    float fs[MF][4], fq[MF][4];
#pragma unroll
                    for (int i = 0; i < MF; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { fs[i][r] = 0.f; fq[i][r] = 0.f; }
#pragma unroll
                    for (int i = 0; i < MF; ++i) {
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
                            const int oy = FLAT ? py0 + (frc[f] >> 16) : py0 + wv * (TH / 4) + (f >> 1);
                            const int ox = FLAT ? (frc[f] & 0xffff) : px0 + (f & 1) * 16 + l15;
                            if ((!FLAT || frc[f] >= 0) && oy < g.Ho && ox < g.Wo) {
                                const int pofs = rowbase + oy * g.Wo + ox;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int ml = i * 16 + l4 * 4 + r;
                                    if (ml < mt) {
                                        const float v = acc[i][f][r] + s_bias[ml];
                                        yout[(i * 16 + r) * HWo + pofs] = v;
                                        fs[i][r] += v; fq[i][r] = __builtin_fmaf(v, v, fq[i][r]);
                                    }
                                }
                            }
                        }
                    }
                    if (do_stats) {
#pragma unroll
                        for (int i = 0; i < MF; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float a = fs[i][r], b = fq[i][r];
#pragma unroll
                                for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                                if (l15 == 0) { s_red[wv][i * 16 + l4 * 4 + r][0] += (double)a; s_red[wv][i * 16 + l4 * 4 + r][1] += (double)b; }
                            }
                    }
                } else if constexpr (FF) {
                    // FLAT tiles of the fused fold: per-element LeakyReLU' and BN-backward partial sums, gradient straight to ga
                    const int Hp = g.H, Wp = g.W, HWp = Hp * Wp;
                    float* __restrict__ o = A.fga + (long long)k * A.fga_sstride + (long long)m0 * HWp;
                    const float* __restrict__ xraw = fuse_sums ? A.xin.data + (long long)k * A.xin.sstride + (long long)m0 * HWp : nullptr;
                    const int rowbase = l4 * 4 * HWp;
                    float fs[MF][4], fq[MF][4];
#pragma unroll
                    for (int i = 0; i < MF; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { fs[i][r] = 0.f; fq[i][r] = 0.f; }
#pragma unroll
                    for (int i = 0; i < MF; ++i)
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
                            const int pr = py0 + (frc[f] >> 16), pc = frc[f] & 0xffff;
                            if (frc[f] >= 0 && pr < Hp && pc < Wp) {
                                const int pofs = rowbase + pr * Wp + pc;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int ml = i * 16 + l4 * 4 + r;
                                    if (ml < mt) {
                                        const int ofs = (i * 16 + r) * HWp + pofs;
                                        float d = acc[i][f][r];
                                        if (fuse_sums) {
                                            const float y = xraw[ofs];
                                            const ChanFwd cf = s_ch[m0 + ml];
                                            const float vv = __builtin_fmaf(y - cf.mean, cf.scale, cf.beta);
                                            if (A.xin.act && !(vv > 0.f)) d *= A.xin.slope;
                                            fs[i][r] += d; fq[i][r] = __builtin_fmaf(d, (y - cf.mean) * cf.rstd, fq[i][r]);
                                        }
                                        o[ofs] = d;
                                    }
                                }
                            }
                        }
                    if (fuse_sums) {
#pragma unroll
                        for (int i = 0; i < MF; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float a = fs[i][r], b = fq[i][r];
#pragma unroll
                                for (int o2 = 8; o2 > 0; o2 >>= 1) { a += __shfl_xor(a, o2, 64); b += __shfl_xor(b, o2, 64); }
                                if (l15 == 0) { s_red[wv][i * 16 + l4 * 4 + r][0] += (double)a; s_red[wv][i * 16 + l4 * 4 + r][1] += (double)b; }
                            }
                    }
                } else {
                    const int Hp = g.H + 2 * P, Wp = g.W + 2 * P, HWp = Hp * Wp;
                    float* __restrict__ o = A.dxp + (long long)k * A.dxp_sstride + (long long)m0 * HWp;
                    const int rowbase = l4 * 4 * HWp;
#pragma unroll
                    for (int i = 0; i < MF; ++i)
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
                            const int pr = FLAT ? py0 + (frc[f] >> 16) : py0 + wv * (TH / 4) + (f >> 1);
                            const int pc = FLAT ? (frc[f] & 0xffff) : px0 + (f & 1) * 16 + l15;
                            if ((!FLAT || frc[f] >= 0) && pr < Hp && pc < Wp) {
                                const int pofs = rowbase + pr * Wp + pc;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int ml = i * 16 + l4 * 4 + r;
                                    if (ml < mt) o[(i * 16 + r) * HWp + pofs] = acc[i][f][r];
                                }
                            }
                        }
                }
                            }
              }
            }
            lds_barrier();
        }
        if (do_stats) {
            // One fp64 atomic per (channel, moment) per BLOCK: same-address float atomics serialise at the memory side
            // (~0.2 us each), so per-tile or per-wave atomics would dominate the kernel.
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

#ifndef MFVI_KERNEL_ONLY      // scripts/dev/regs_conv_mfma.sh: explicit instantiations of a few kernels for a quick register / spill report
// MFVI_PHASE=0: stride-2 backward-data on rectangular tiles runs the zero-stuffed formulation (A/B and parity cross-checks)
bool phase_on()
{
    static const bool on = [] { const char* e = getenv("MFVI_PHASE"); return !(e && e[0] == '0'); }();
    return on;
}

// Tiling override: MFVI_TUNE=mf,th,T (experiments) or the per-op choice made by mfvi_plan_autotune (ConvGeom::tune).
int env_tune()
{
    static const int t = [] { int mf = 0, th = 0, T = 0; const char* e = getenv("MFVI_TUNE"); if (e) sscanf(e, "%d,%d,%d", &mf, &th, &T); return mf > 0 ? (mf | th << 8 | T << 16) : 0; }();
    return t;
}

template <int KS, int STRIDE, int MODE>
int launch_variant(const TView& xin, const GView& gin, const ConvGeom& g, const float* w, long long wstride, OutDesc out, float* dxp,
                   long long dxp_sstride, int n_samples, hipStream_t st, FoldFuse fuse = FoldFuse{})
{
    const int P = g.ks / 2, KK = KS * KS;
    constexpr bool CAN_FF = MODE == 1 && KS == 3 && STRIDE == 1;
    const bool ff = CAN_FF && g.stride == 1 && fuse.ga != nullptr;                           // 3x3 stride-1 backward-data with the fold in its epilogue
    const int OH = MODE == 0 ? g.Ho : (ff ? g.H : g.H + 2 * P), OW = MODE == 0 ? g.Wo : (ff ? g.W : g.W + 2 * P);    // output pixel domain
    const int MOUT = MODE == 0 ? g.Cout : g.Cin, RED = MODE == 0 ? g.Cin : g.Cout;
    const int RED4 = (RED + 3) & ~3;
    MfmaArgs A{xin, gin, g, w, wstride, out, dxp, dxp_sstride, fuse.ga, fuse.ga_sstride, fuse.bsums, 0, 0, 1};
    A.vec_out = MODE == 0 && (g.Wo & 3) == 0 && (out.sstride & 3) == 0 && ((uintptr_t)out.data & 15) == 0;

    // Pick the largest tile that still gives the chip enough blocks: big tiles amortise the in-kernel weight sampling
    // (each sampled weight is reused by every pixel of the tile), small ones keep 256 CUs busy.  When the whole slab of a
    // block fits in LDS and there are tiles to spare, go weight-stationary and give each block several tiles.
    const auto blocks = [&](int mf, int th) { return (long long)((OW + 31) / 32) * ((OH + th - 1) / th) * ((MOUT + 16 * mf - 1) / (16 * mf)) * n_samples; };
    // Output-channel fragments per block: padded fragments cost matrix-core time, every extra channel tile re-reads the
    // whole activation operand -> minimise frags + tiles/2.
    const auto cost = [&](int mf) { const int tiles = (MOUT + 16 * mf - 1) / (16 * mf); return 2 * tiles * mf + tiles; };
    int order[4] = {4, 3, 2, 1};
    for (int i = 1; i < 4; ++i) for (int j = i; j > 0 && cost(order[j]) < cost(order[j - 1]); --j) { const int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp; }
    const long long want = 768;
    int forced_T = 0;
    const size_t chan_bytes = sizeof(ChanFwd) * (size_t)((g.Cin + 3) & ~3) + sizeof(ChanBwd) * (size_t)((g.Cout + 3) & ~3);   // dynamic LDS behind the weights
    bool want_rem = false;      // tune bit 64 of the tile-height field: carry the layer's last 4 output channels on the 4x4x1 matrix instruction
#define GO_(MF_, TH_, FL_)                                                                                                 \
    {                                                                                                                      \
        using Cfg = MCfg<KS, STRIDE, MF_, TH_>;                                                                            \
        A.tiles_x = (OW + 31) / 32;                                                                                        \
        A.n_tiles = A.tiles_x * ((OH + TH_ - 1) / TH_);                                                                    \
        if (ff && !(FL_) && RED <= 3 * Cfg::CC) return -3;   /* producer-side fold: three parts in stages 0..2 of the next tile, dump at the last stage */ \
        if ((FL_) && ff && (OW & 7)) return -3;       /* fused fold on FLAT tiles: rows are whole half-fragments (8x8 maps: two rows per fragment) */ \
        if (FL_) {                                                                                                         \
            A.ow = OW; A.wpitch = (OW - 1) * STRIDE + KS;                                                                  \
            int rt = (TH_ * 32) / OW;                                                                                      \
            const int cap = (Cfg::IN_TH * Cfg::PITCH) / A.wpitch;            /* window rows that fit the LDS plane */      \
            if ((rt - 1) * STRIDE + KS > cap) rt = (cap - KS) / STRIDE + 1;                                                \
            if (rt > OH) rt = OH;                                                                                          \
            if (rt < 1 || OW > 0xffff) return -3;                                                                          \
            A.rt = rt; A.nwin = ((rt - 1) * STRIDE + KS) * A.wpitch;                                                       \
            A.n_tiles = (OH + rt - 1) / rt;                                                                                \
        }                                                                                                                  \
        constexpr bool CAN_REM = CAN_FF && !(FL_) && TH_ == 8;                                                            \
        if (want_rem && !(CAN_REM && ff && (MOUT & 15) == 4 && (MOUT - 4) % (16 * MF_) == 0)) return -3;                  \
        const bool rem = want_rem;                                                                                         \
        const int my = rem ? (MOUT - 4) / (16 * MF_) : (MOUT + 16 * MF_ - 1) / (16 * MF_);                                 \
        constexpr bool CAN_BIG = (KS == 1 && !(FL_)) || (KS == 3 && STRIDE == 1 && (FL_) && TH_ <= 4);   /* 3x3: the small maps at the bottom of the hour-glass, both passes */ \
        constexpr bool CAN_PH = MODE == 1 && KS == 3 && !(FL_);                                                           \
        const bool ph = CAN_PH && g.stride == 2 && phase_on() && (g.Wo & 3) == 0 && (gin.gstride & 3) == 0 && (!gin.y || (gin.ystride & 3) == 0); \
        using BigCfg = MCfg<KS, STRIDE, MF_, TH_, CAN_BIG>;                                                                \
        bool big = CAN_BIG && RED >= 64;                                                                                   \
        using RemCfg = MCfg<KS, STRIDE, MF_, TH_, false, CAN_REM>;                                                         \
        const size_t ws_bytes = sizeof(float) * (size_t)KK * RED4 * (rem ? RemCfg::CTP : Cfg::CTP);                        \
        const size_t ck_big = 2 * sizeof(float) * (size_t)KK * BigCfg::CC * Cfg::CTP;                                      \
        const size_t lds_big = 2 * sizeof(float) * (size_t)BigCfg::X_FLOATS + 24 * 1024;   /* static LDS of the big-stage variant */ \
        size_t ck_bytes = 2 * sizeof(float) * (size_t)KK * Cfg::CC * (rem ? RemCfg::CTP : Cfg::CTP);                       \
        const long long nb = (long long)A.n_tiles * my * n_samples;                                                        \
        int T = (int)(nb / 512); T = T < 1 ? 1 : (T > 8 ? 8 : T);                                                          \
        if (forced_T > 0) T = forced_T;                                                                                    \
        if (T >= 2 && ws_bytes <= 40 * 1024) {                                                                             \
            if (lds_big + ws_bytes > 150 * 1024) big = false;                                                              \
            A.tiles_per_block = T;                                                                                         \
            A.nx = (A.n_tiles + T - 1) / T; A.ny = my; A.nz = n_samples;                                                   \
            if (rem) mfvi_launch((conv_mfma_kernel<KS, STRIDE, MF_, TH_, MODE, true, FL_, false, false, CAN_REM, CAN_REM>), dim3(A.nx * A.ny * A.nz), dim3(512), ws_bytes + chan_bytes, st, A); \
            else if (ff && big) mfvi_launch((conv_mfma_kernel<KS, STRIDE, MF_, TH_, MODE, true, FL_, CAN_BIG, false, CAN_FF && CAN_BIG>), dim3(A.nx * A.ny * A.nz), dim3(512), ws_bytes + chan_bytes, st, A); \
            else if (ff) mfvi_launch((conv_mfma_kernel<KS, STRIDE, MF_, TH_, MODE, true, FL_, false, false, CAN_FF>), dim3(A.nx * A.ny * A.nz), dim3(512), ws_bytes + chan_bytes, st, A); \
            else if (big) mfvi_launch((conv_mfma_kernel<KS, STRIDE, MF_, TH_, MODE, true, FL_, CAN_BIG>), dim3(A.nx * A.ny * A.nz), dim3(512), ws_bytes + chan_bytes, st, A); \
            else if (ph) mfvi_launch((conv_mfma_kernel<KS, STRIDE, MF_, TH_, MODE, true, FL_, false, CAN_PH>), dim3(A.nx * A.ny * A.nz), dim3(512), ws_bytes + chan_bytes, st, A); \
            else mfvi_launch((conv_mfma_kernel<KS, STRIDE, MF_, TH_, MODE, true, FL_>), dim3(A.nx * A.ny * A.nz), dim3(512), ws_bytes + chan_bytes, st, A); \
        } else {                                                                                                           \
            /* chunked weights: one tile per block, except for the fused fold on rectangular tiles, whose producer-side fold of tile i   \
               overlaps the stages of tile i+1 (the weights of a stage are re-read from L2 per tile) */                                   \
            const bool multi = ff && !(FL_);                                                                               \
            if (forced_T > 1 && !multi) return -3;                                                                         \
            if (lds_big + ck_big > 150 * 1024) big = false;                                                                \
            if (big) ck_bytes = ck_big;                                                                                    \
            A.tiles_per_block = multi ? T : 1;                                                                             \
            A.nx = (A.n_tiles + A.tiles_per_block - 1) / A.tiles_per_block; A.ny = my; A.nz = n_samples;                   \
            if (rem) mfvi_launch((conv_mfma_kernel<KS, STRIDE, MF_, TH_, MODE, false, FL_, false, false, CAN_REM, CAN_REM>), dim3(A.nx * A.ny * A.nz), dim3(512), ck_bytes + chan_bytes, st, A); \
            else if (ff && big) mfvi_launch((conv_mfma_kernel<KS, STRIDE, MF_, TH_, MODE, false, FL_, CAN_BIG, false, CAN_FF && CAN_BIG>), dim3(A.nx * A.ny * A.nz), dim3(512), ck_bytes + chan_bytes, st, A); \
            else if (ff) mfvi_launch((conv_mfma_kernel<KS, STRIDE, MF_, TH_, MODE, false, FL_, false, false, CAN_FF>), dim3(A.nx * A.ny * A.nz), dim3(512), ck_bytes + chan_bytes, st, A); \
            else if (big) mfvi_launch((conv_mfma_kernel<KS, STRIDE, MF_, TH_, MODE, false, FL_, CAN_BIG>), dim3(A.nx * A.ny * A.nz), dim3(512), ck_bytes + chan_bytes, st, A); \
            else if (ph) mfvi_launch((conv_mfma_kernel<KS, STRIDE, MF_, TH_, MODE, false, FL_, false, CAN_PH>), dim3(A.nx * A.ny * A.nz), dim3(512), ck_bytes + chan_bytes, st, A); \
            else mfvi_launch((conv_mfma_kernel<KS, STRIDE, MF_, TH_, MODE, false, FL_>), dim3(A.nx * A.ny * A.nz), dim3(512), ck_bytes + chan_bytes, st, A); \
        }                                                                                                                  \
        return (int)hipGetLastError();                                                                                     \
    }
#define GO(MF_, TH_) GO_(MF_, TH_, false)
#define GO_MF(mf_, TH_) { if ((mf_) == 1) GO(1, TH_) if ((mf_) == 2) GO(2, TH_) if ((mf_) == 3) GO(3, TH_) }
#define GO_MF_FLAT(mf_, TH_) { if ((mf_) == 1) GO_(1, TH_, true) if ((mf_) == 2) GO_(2, TH_, true) if ((mf_) == 3) GO_(3, TH_, true) }
    const int forced = g.tune[MODE] ? g.tune[MODE] : env_tune();
    if (forced) {              // explicit tiling (mf | th << 8 | T << 16); -3 = not a valid tiling for this shape
        const int mf = forced & 255;
        int th = (forced >> 8) & 255;
        forced_T = (forced >> 16) & 255;
        if (th & 64) { want_rem = true; th &= ~64; if (th != 8) return -3; }
        if (th & 128) {        // FLAT tiles: 3x3 stride-1 kernels on domains up to 130 wide
            if constexpr (KS >= 3 && STRIDE == 2) {      // stride-2 forward on small outputs (MODE 1 always runs the stride-1 kernel)
                if (OW > 32) return -3;
                if ((th & 127) == 8) { if (mf == 1) GO_(1, 8, true) if (mf == 2) GO_(2, 8, true) if (mf == 4) GO_(4, 8, true) }
                if ((th & 127) == 4) { if (mf == 1) GO_(1, 4, true) if (mf == 2) GO_(2, 4, true) if (mf == 4) GO_(4, 4, true) }
                if ((th & 127) == 2) { if (mf == 1) GO_(1, 2, true) if (mf == 2) GO_(2, 2, true) if (mf == 4) GO_(4, 2, true) }
            }
            if constexpr (KS >= 3 && STRIDE == 1) {
                if (OW > 132) return -3;
                if constexpr (KS == 3) { if ((th & 127) == 16) { if (OH * OW >= 256) GO_MF_FLAT(mf, 16) return -3; } }
                if ((th & 127) == 8) { if constexpr (KS == 3) GO_MF_FLAT(mf, 8) else { if (mf == 1) GO_(1, 8, true) if (mf == 2) GO_(2, 8, true) } if (mf == 4) GO_(4, 8, true) }
                // 128- and 64-pixel tiles for the 8x8 / 10x10 maps at the bottom of the hour-glass (a 256-pixel tile is 25-39% full there)
                if ((th & 127) == 4) { if (mf == 1) GO_(1, 4, true) if (mf == 2) GO_(2, 4, true) if (mf == 4) GO_(4, 4, true) }
                if ((th & 127) == 2) { if (mf == 1) GO_(1, 2, true) if (mf == 2) GO_(2, 2, true) if (mf == 4) GO_(4, 2, true) }
            }
            return -3;
        }
        if (th == 16) { if constexpr (STRIDE == 1 && KS != 5) { if (OH >= 16) GO_MF(mf, 16) } return -3; }
        if (th == 8) { if constexpr (KS == 5) { if (mf == 1) GO(1, 8) if (mf == 2) GO(2, 8) } else GO_MF(mf, 8) if (mf == 4) GO(4, 8) }
        return -3;
    }
    if constexpr (KS == 5) {       // 25 taps per channel step: the variants are kept to 8-row tiles with 1, 2 or 4 output fragments
        for (int mf : {4, 2}) if (blocks(mf, 8) >= want && MOUT >= 16 * mf) { if (mf == 4) GO(4, 8) GO(2, 8) }
        GO(1, 8)
    } else {
        if constexpr (STRIDE == 1) {
            if (OH >= 16)
                for (int i = 0; i < 4; ++i) if (order[i] <= 3 && !(MODE == 0 && order[i] == 3) && blocks(order[i], 16) >= want) GO_MF(order[i], 16)
        }
        for (int i = 0; i < 4; ++i) if (blocks(order[i], 8) >= want) { GO_MF(order[i], 8) if (order[i] == 4) GO(4, 8) }
        GO(1, 8)
    }
#undef GO_MF_FLAT
#undef GO_MF
#undef GO
#undef GO_
}

#else
}  // namespace
#endif
#ifndef MFVI_KERNEL_ONLY
}  // namespace

// Returns -2 when the shape is not served by the MFMA path (caller falls back to the generic kernels).
int launch_conv_fwd_mfma(const TView& in, const ConvGeom& g, const float* w, long long wstride, OutDesc out, int n_samples, hipStream_t st)
{
    if (g.Cin > MFVI_MAX_C || (g.Cin & 3) || (g.w_off & 3)) return -2;      // Philox blocks must tile every weight row
    if (g.tune[0] & MFVI_TUNE_GENERIC) return -2;                           // in-kernel eps: the generic kernel draws and convolves in one launch
    if ((long long)g.Cout * g.Ho * g.Wo >= (1LL << 31)) return -2;          // the epilogue uses 32-bit element offsets per sample
    // aligned float4 staging: image rows, sample strides and the base pointer must be multiples of 4 floats
    if ((g.W & 3) || g.W < 4 || (in.sstride & 3) || ((uintptr_t)in.data & 15)) return -2;
    {   // row-phase kernels (conv_rp.hip): an explicit tiling of the plan / autotuner, or the default for the shapes they serve
        int tn = g.tune[0] ? g.tune[0] : (env_tune() ? 0 : rp_default_tune(g, 0, n_samples));
        if (tn & MFVI_TUNE_ST) {      // streaming forward of a narrow 1x1 layer (conv_1x1.hip): only as an explicit tiling of the plan / autotuner
            const int rc = launch_conv1_fwd_stream(in, g, w, wstride, out, n_samples, st);
            return rc == -2 ? -3 : rc;
        }
        if (tn & MFVI_TUNE_SM) {      // small-map forward (conv_small.hip): only as an explicit tiling of the plan / autotuner
            const int rc = g.ks == 1 ? launch_conv1_fwd_small(in, g, w, wstride, out, n_samples, st) : launch_conv_fwd_small(in, g, w, wstride, out, n_samples, st);
            return rc == -2 ? -3 : rc;
        }
        if (tn & MFVI_TUNE_X6) {      // bf16x6 forward (conv_x6.hip): only as an explicit tiling of the plan / autotuner
            const int rc = launch_conv_fwd_x6(in, g, w, wstride, out, tn & (MFVI_TUNE_X6 - 1), n_samples, st);
            if (rc != -2) return rc;
            // no scratch for the weight pieces in this call (w = mu of the eval branch, sample_weights = 0: the plan hands the scratch over
            // only behind a weight draw): the layer's fp32 default, not the generic kernels
            tn = env_tune() ? 0 : rp_default_tune(g, 0, n_samples);
        }
        if (tn & MFVI_TUNE_RP) {
            const int rc = launch_conv_fwd_rp(in, g, w, wstride, out, tn & (MFVI_TUNE_RP - 1), n_samples, st);
            // an explicit tiling of the plan answers for itself (-3); a heuristic one the shape does not admit falls through to the round-2 tiles
            if (g.tune[0] & MFVI_TUNE_RP) return rc == -2 ? -3 : rc;
            if (rc != -2 && rc != -3) return rc;
        }
    }
    GView none{};
    if (g.ks == 3 && g.stride == 1) return launch_variant<3, 1, 0>(in, none, g, w, wstride, out, nullptr, 0, n_samples, st);
    if (g.ks == 3 && g.stride == 2) return launch_variant<3, 2, 0>(in, none, g, w, wstride, out, nullptr, 0, n_samples, st);
    if (g.ks == 1 && g.stride == 1) return launch_variant<1, 1, 0>(in, none, g, w, wstride, out, nullptr, 0, n_samples, st);
    if (g.ks == 5 && g.stride == 1) return launch_variant<5, 1, 0>(in, none, g, w, wstride, out, nullptr, 0, n_samples, st);
    if (g.ks == 5 && g.stride == 2) return launch_variant<5, 2, 0>(in, none, g, w, wstride, out, nullptr, 0, n_samples, st);
    return -2;
}

int launch_conv_bwd_data_mfma(const GView& gy, const ConvGeom& g, const float* w, long long wstride, float* dxp, long long dxp_sstride,
                              int n_samples, hipStream_t st, const FoldFuse* fuse)
{
    if (g.Cout > MFVI_MAX_C || (g.stride != 1 && !(g.stride == 2 && g.ks >= 3)) || (g.Cin & 3) || (g.w_off & 3)) return -2;
    if (g.tune[1] & MFVI_TUNE_GENERIC) return -2;
    if ((long long)g.Cin * (g.H + 4) * (g.W + 4) >= (1LL << 31)) return -2;   // 32-bit element offsets per sample
    // aligned float4 (stride 2: float2) staging of the gradient and of the conv output it is normalised with
    const int wa = g.stride == 2 ? 1 : 3;
    if ((g.Wo & wa) || g.Wo < (wa + 1) || (gy.gstride & wa) || ((uintptr_t)gy.ga & 15) || (gy.y && ((gy.ystride & wa) || ((uintptr_t)gy.y & 15)))) return -2;
    TView none{}; OutDesc od{};
    if (fuse) {       // the fold runs in the epilogue: aligned float4 rows of the input tensor and of its gradient
        if (!(g.ks == 1 || (g.ks == 3 && g.stride == 1)) || !fuse->ga || (g.W & 3) || (fuse->ga_sstride & 3) || ((uintptr_t)fuse->ga & 15)) return -2;
        if (fuse->bsums && ((fuse->x.sstride & 3) || ((uintptr_t)fuse->x.data & 15))) return -2;
        if (g.ks == 3 && (g.H < 4 || g.W < 4)) return -2;      // rows 1 and H-2 (columns 1 and W-2) must be distinct, interior lines
        if (g.ks == 3) {
            int tn = g.tune[1] ? g.tune[1] : (env_tune() ? 0 : rp_default_tune(g, 1, n_samples));
            if (tn & MFVI_TUNE_X6) {      // bf16x6 backward-data with the fold (conv_bwd_x6.hip): only as an explicit tiling of the plan / autotuner
                const int rc = launch_conv_bwd_data_x6(gy, g, w, wstride, tn & (MFVI_TUNE_X6 - 1), n_samples, st, *fuse);
                if (rc != -2) return rc;
                // no scratch for the weight pieces in this call (w = mu without a weight draw): the layer's fp32 default
                tn = env_tune() ? 0 : rp_default_tune(g, 1, n_samples);
                if (tn & MFVI_TUNE_RP) { const int r2 = launch_conv_bwd_data_rp(gy, g, w, wstride, tn & (MFVI_TUNE_RP - 1), n_samples, st, *fuse); if (r2 != -2 && r2 != -3) return r2; }
                return launch_variant<3, 1, 1>(fuse->x, gy, g, w, wstride, od, nullptr, 0, n_samples, st, *fuse);
            }
            if (tn & MFVI_TUNE_SM) {      // small-map kernel (conv_small.hip): only as an explicit tiling of the plan / autotuner
                const int rc = launch_conv_bwd_data_small(gy, g, w, wstride, n_samples, st, *fuse);
                return rc == -2 ? -3 : rc;
            }
            if (tn & MFVI_TUNE_RP) {
                const int rc = launch_conv_bwd_data_rp(gy, g, w, wstride, tn & (MFVI_TUNE_RP - 1), n_samples, st, *fuse);
                if (g.tune[1]) return rc == -2 ? -3 : rc;
                if (rc != -2 && rc != -3) return rc;      // (heuristic tiling not valid for this shape: the round-2 tiles below)
            }
        }
        if (g.ks == 3) return launch_variant<3, 1, 1>(fuse->x, gy, g, w, wstride, od, nullptr, 0, n_samples, st, *fuse);
        if (g.tune[1] & MFVI_TUNE_SM) {      // one-stage 1x1 kernel (conv_1x1.hip): only as an explicit tiling of the plan / autotuner
            const int rc = launch_conv1_bwd_data_small(gy, g, w, wstride, n_samples, st, *fuse);
            return rc == -2 ? -3 : rc;
        }
        return launch_variant<1, 1, 1>(fuse->x, gy, g, w, wstride, od, nullptr, 0, n_samples, st, *fuse);
    }
    if (g.ks == 3) return launch_variant<3, 1, 1>(none, gy, g, w, wstride, od, dxp, dxp_sstride, n_samples, st);
    if (g.ks == 1) return launch_variant<1, 1, 1>(none, gy, g, w, wstride, od, dxp, dxp_sstride, n_samples, st);
    if (g.ks == 5) return launch_variant<5, 1, 1>(none, gy, g, w, wstride, od, dxp, dxp_sstride, n_samples, st);
    return -2;
}
#endif
