// K2a' for the 3x3 stride-1 layers on maps a multiple of 64 wide: backward-data WITH the fold of the input tensor, on the BF16 matrix cores
// at fp32 accuracy ("bf16x6", round 4).  Autograd of BayTorch/modules/reparam_layers.py:37 behind the ReflectionPad2d(1) of
// models/common.py:100-135, followed by LeakyReLU' and the BatchNorm-backward sums of the layer's input (models/common.py:77-97).
//
// Arithmetic as conv_x6.hip / conv_bww_x6.hip: every fp32 operand is the exact sum of three round-to-nearest bf16 pieces and six
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation stand for one fp32 product-sum (dropped terms <= 2^-23 |a b|).  Here
//     dx[ci][y][x] = sum_{co, ky, kx} W[co][ci][ky][kx] * dy[co][y + 1 - ky][x + 1 - kx]        (dy zero outside the image)
//   M = 16 consecutive pixels of a row, N = 16 INPUT channels ci of the convolution (a "fragment" f), K = 32 OUTPUT channels co of one tap.
//   16 output channels (the 36 -> 16 layer): K = 32 is [piece a | piece b] of the same 16 channels, so THREE instructions
//   ([x_h|x_l].[w_l|w_h], [x_m|x_m].[w_h|w_m], [x_h|x_h].[w_h|w_m]) carry the six products and no K slot is empty.
//
// What the round-3 design note asked for: the staged tensor is staged ONCE per block.  A block owns a 64-pixel
// band and walks T strips of SR output rows down it; the strip's dy window ((SR + 2) rows x 66 pixels x all Cout channels, BN-backward
// formed on load, as three bf16 pieces in channel-octet planes) is resident in LDS while the four matrix waves (wave = 16-pixel fragment)
// run one PASS per (input-channel fragment f, 32-channel reduction group): the pass's weight pieces (pre-split per sample by
// x6b_split_kernel: [f][group][tap][piece][n][k-octet]) go LDS -> registers at its head (108 VGPRs; 72 in the 16-channel form), then the
// window rows stream by: an input row meets the three tap rows of three output rows, the strip's accumulators stay in registers (4 per
// output row).  At the end of a fragment's last pass the SAME waves fold straight from the accumulators: the raw x of their 4 x SR pixels
// of channel 16 f + (lane & 15) was requested at the head of the pass, LeakyReLU'(BN(x)), the BN-backward sums of x (two floats per
// (wave, channel) in LDS, one fp64 atomic per (block, channel, moment) at the end) and ga written once as float4.
// The four staging waves prefetch the NEXT strip's SR new rows during the strip's passes (global dwords -> BN-backward -> three pieces,
// kept in 60 registers) and write them over the SR oldest ring rows at the strip boundary (one short exposed phase per strip); they
// also copy the next pass's weight pieces global -> registers -> LDS.
//
// Reflection adjoint (the gradient is formed on the UN-padded domain):
//   rows:    padded row -1 folds onto image row 1: image row 0 with tap row ky = 0 accumulates into output row 1 as well (18 extra matrix
//            instructions per fragment in a band's first strip); likewise image row H-1 with ky = 2 into row H-2 in the last strip.
//   columns: pixel 1 takes tap kx = 0 from dy[2] + dy[0], pixel W-2 takes kx = 2 from dy[W-3] + dy[W-1]: the staging waves write these
//            two sums (formed in fp32 like conv_rp.hip's column fix-up, then split) into two spare pixel slots of the row planes and the
//            one lane concerned reads its operand there — no extra matrix instruction, no branch.
#include "common.h"
#include <type_traits>
#include <cstdlib>

thread_local float* mfvi_tl_x6bw = nullptr;       // split weight pieces of the op being launched (plan.hip); nullptr: kernel not available
thread_local bool mfvi_tl_x6bw_ready = false;     // the pieces of this pass are already there (launch_x6b_split_all ran behind the weight draw)

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

// BN-backward on load in three operations per element (as conv_rp.hip): dy = (y - mean) * qc + (ga * c1 + k2)
struct BwdC { float mean, qc, c1, k2; };

template <bool K16, int SR, int NG>
struct X6BCfg {
    static constexpr int NOCT = K16 ? 2 : 4;               // channel octets per reduction group
    static constexpr int NOCTT = NOCT * NG;                 // octets of the staged tensor
    static constexpr int PLANE = 80 * 16;                   // one (piece, octet) row plane: slots 0..65 = image columns c0-1 .. c0+64, 66 / 67 = the column-adjoint sums; == 0 (mod 256 B): a ds_read_b128 lane group hits 64 distinct banks
    static constexpr int PIECE = NOCTT * PLANE;
    static constexpr int ROWB = 3 * PIECE;
    static constexpr int NR = SR + 2;                       // ring = exactly the strip's window
    static constexpr int RING = NR * ROWB;
    static constexpr int NV = K16 ? 2 : 3;                  // weight operand variants per tap
    static constexpr int WB = 9 * NV * 1024;                // weight pieces of one pass
    static constexpr int NWU = (WB / 16 + 255) / 256;       // 16-byte units per staging thread
};

struct X6BArgs {
    GView gin; TView xin; ConvGeom g;
    const unsigned* wsp; long long wsp_stride_u4;           // split pieces, per-sample stride in 16-byte units (0: one copy for all samples)
    float* fga; long long fga_sstride; double* fbsums;
    int NF, bands, strips, tpb, nx, nz;
};

// ---- weight pieces: thread = (f, group, tap, n, k-octet) of sample k -> NV 16-byte operand units ----
// destination (16-byte units): (((f * NG + grp) * 9 + tap) * NV + v) * 64 + n * 4 + g4
__device__ __forceinline__ void x6b_split_body(const X6BSplitEntry& E, int u, int k, const float* w, long long wstride, float* arena)
{
    if (u >= E.units) return;
    const int g4 = u & 3, n = (u >> 2) & 15, r = u >> 6, tap = r % 9, fg = r / 9, grp = fg % E.NG, f = fg / E.NG;
    const float* __restrict__ ww = w + (long long)k * wstride + E.w_off;
    const int ci = 16 * f + n, co0 = 32 * grp + 8 * (E.k16 ? (g4 & 1) : g4);
    float e8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e8[j] = ci < E.CI ? ww[((long long)(co0 + j) * E.CI + ci) * 9 + tap] : 0.f;
    u32x4 h, m, l; split8(e8, h, m, l);
    const int NV = E.k16 ? 2 : 3;
    u32x4* d = reinterpret_cast<u32x4*>(arena + E.dst_off) + (long long)k * ((long long)E.units * NV) + ((long long)r * NV) * 64 + n * 4 + g4;
    if (E.k16) { d[0] = g4 < 2 ? h : m; d[64] = g4 < 2 ? l : h; }      // [w_h | w_m], [w_l | w_h]
    else { d[0] = h; d[64] = m; d[128] = l; }
}
// every bf16x6 backward-data layer of a plan in ONE launch behind the weight draw
__global__ void x6b_split_kernel(const X6BSplitEntry* __restrict__ table, int n_entries, const float* w, long long wstride, float* arena)
{
    int e = 0;
    while (e + 1 < n_entries && (int)blockIdx.x >= table[e + 1].first_block) ++e;
    const X6BSplitEntry E = table[e];
    x6b_split_body(E, ((int)blockIdx.x - E.first_block) * blockDim.x + threadIdx.x, blockIdx.y, w, wstride, arena);
}
// one layer by itself (autotuning; a launch outside a pass-wide split): the entry travels as a kernel argument
__global__ void x6b_split_one_kernel(X6BSplitEntry E, const float* w, long long wstride, float* arena)
{
    x6b_split_body(E, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y, w, wstride, arena);
}

template <bool K16, int SR, int NG>
__global__ __launch_bounds__(512, 2) void conv_bwd_x6_kernel(X6BArgs A)
{
    using C = X6BCfg<K16, SR, NG>;
    constexpr int NOCT = C::NOCT, NOCTT = C::NOCTT, PLANE = C::PLANE, PIECE = C::PIECE, ROWB = C::ROWB, NR = C::NR, NV = C::NV;
    extern __shared__ __align__(16) char lds[];             // ring [NR][3][NOCTT][80][16] | weight pieces [9][NV][16][4][16] | tables
    char* const s_w = lds + C::RING;
    const ConvGeom& g = A.g;
    const int CI = g.Cin, CO = g.Cout, H = g.H, W = g.W, HW = H * W;
    const int NF = A.NF, NFS = NF * 16;
    BwdC* const s_chb = reinterpret_cast<BwdC*>(s_w + C::WB);                       // [CO]
    ChanFwd* const s_ch = reinterpret_cast<ChanFwd*>(s_chb + CO);                   // [NFS]
    float* const s_sum = reinterpret_cast<float*>(s_ch + NFS);                      // [4][NFS][2]

    const int tid = threadIdx.x;
    const bool producer = tid >= 256;
    const int t = tid & 255, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    int bx, by, k;
    xcd_decode(blockIdx.x, A.nx, 1, A.nz, bx, by, k);
    const int band = bx % A.bands, strip0 = (bx / A.bands) * A.tpb;
    const int n_strip = min(A.tpb, A.strips - strip0);
    const int c0 = band * 64;
    const int n_pps = NF * NG;                              // passes per strip
    const int n_pass = n_strip * n_pps;
    const bool fuse_sums = A.fbsums != nullptr;
    const unsigned* __restrict__ wsp = A.wsp + (long long)k * A.wsp_stride_u4 * 4;

    if (!producer) {
        for (int c = t; c < CO; c += 256) { const ChanBwd b = chan_bwd(A.gin, k, c); BwdC r; r.mean = b.mean; r.qc = -b.c1 * b.c3 * b.rstd; r.c1 = b.c1; r.k2 = -b.c1 * b.c2; s_chb[c] = r; }
        for (int c = t; c < NFS; c += 256) { ChanFwd f; if (fuse_sums) f = chan_fwd(A.xin, k, min(c, CI - 1)); else { f.mean = 0.f; f.scale = 1.f; f.beta = 0.f; f.rstd = 1.f; } s_ch[c] = f; }
        for (int i = t; i < 4 * NFS * 2; i += 256) s_sum[i] = 0.f;
    }

#ifdef X6B_DBG_NOPROD
    if (false) {
#else
    if (producer) {
#endif
        // ======================= staging waves =======================
        __builtin_amdgcn_s_setprio(1);
        const float* __restrict__ gsrc = A.gin.ga + (long long)k * A.gin.gstride;
        const float* __restrict__ ysrc = A.gin.stats ? A.gin.y + (long long)k * A.gin.ystride : nullptr;
        const bool lb = c0 == 0, rb = c0 + 64 == W;
        // The SR new rows of a strip are 16 tasks (row j, octet q) of 64 pixels — SR * NOCTT == 16 for every instantiation — four per
        // staging wave (task i of wave w: number w + 4 i), plus one task of the four special pixel slots of this wave's four (row, octet)
        // pairs on lanes 0..15 (pair = lane >> 2, kind = lane & 3: left halo, right halo, left sum dy[2] + dy[0], right sum dy[W-3] + dy[W-1]).
        static_assert(SR * NOCTT == 16, "task geometry");
        const int pair = lane >> 2, kind = lane & 3;
        u32x4 pc[5][3];                                     // finished pieces of the five tasks, waiting for the strip boundary
        float ga_[2][8], y_[2][8];                          // raw loads of two tasks in flight (the special-slot task uses both sets: column A, column B)
        int jrow[5], qoct[5];                               // (task 4: per lane)
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int tn = wv + 4 * i; jrow[i] = tn / NOCTT; qoct[i] = tn % NOCTT; }
        { const int tn = wv + 4 * min(pair, 3); jrow[4] = tn / NOCTT; qoct[4] = tn % NOCTT; }
        const bool sp_on = lane < 16 && (kind == 0 ? !lb : kind == 1 ? !rb : kind == 2 ? lb : rb);
        const int colA = kind == 0 ? max(c0 - 1, 0) : kind == 1 ? min(c0 + 64, W - 1) : kind == 2 ? 2 : W - 3;
        const int colB = kind == 2 ? 0 : W - 1;
        const bool two = kind >= 2;
        const int sslot = kind == 0 ? 0 : kind == 1 ? 65 : kind == 2 ? 66 : 67;
        // fetch: raw ga / y of task i for the rows starting at image row Rb (rows outside the image: clamped, zeroed in finish)
        // (32-bit element offsets from a wave-uniform base: with 64-bit per-lane addresses the compiler kept ~100 address registers live)
        const unsigned uHW = (unsigned)HW;
        auto fetch = [&](int i, int set, int Rb) {
            const int R = min(max(Rb + jrow[i], 0), H - 1);
            const unsigned off = (unsigned)(8 * qoct[i]) * uHW + (unsigned)(R * W + c0) + (unsigned)lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) { ga_[set][j] = gsrc[off + (unsigned)j * uHW]; y_[set][j] = ysrc ? ysrc[off + (unsigned)j * uHW] : 0.f; }
        };
        auto fetch_sp = [&](int Rb) {
            const int R = min(max(Rb + jrow[4], 0), H - 1);
            const unsigned base = (unsigned)(8 * qoct[4]) * uHW + (unsigned)(R * W);
            const unsigned oa = base + (unsigned)colA, ob = base + (unsigned)colB;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ga_[0][j] = gsrc[oa + (unsigned)j * uHW]; y_[0][j] = ysrc ? ysrc[oa + (unsigned)j * uHW] : 0.f;
                ga_[1][j] = gsrc[ob + (unsigned)j * uHW]; y_[1][j] = ysrc ? ysrc[ob + (unsigned)j * uHW] : 0.f;
            }
        };
        auto finish = [&](int i, int set, int Rb) {
            const int R = Rb + jrow[i];
            float e[8];
            int qi = __builtin_amdgcn_readfirstlane(8 * qoct[i]); asm volatile("" : "+s"(qi));      // opaque: the loop-invariant table reads are NOT hoisted over the strip loop (160 registers)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const BwdC b = s_chb[qi + j]; e[j] = __builtin_fmaf(y_[set][j] - b.mean, b.qc, __builtin_fmaf(ga_[set][j], b.c1, b.k2)); }
            if (R < 0 || R >= H) {
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = 0.f;
            }
            split8(e, pc[i][0], pc[i][1], pc[i][2]);
        };
        auto finish_sp = [&](int Rb) {
            const int R = Rb + jrow[4];
            float e[8];
            int qi = 8 * qoct[4]; asm volatile("" : "+v"(qi));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const BwdC b = s_chb[qi + j];
                const float a = __builtin_fmaf(y_[0][j] - b.mean, b.qc, __builtin_fmaf(ga_[0][j], b.c1, b.k2));
                const float c = __builtin_fmaf(y_[1][j] - b.mean, b.qc, __builtin_fmaf(ga_[1][j], b.c1, b.k2));
                e[j] = two ? a + c : a;
            }
            if (!sp_on || R < 0 || R >= H) {
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = 0.f;
            }
            split8(e, pc[4][0], pc[4][1], pc[4][2]);
        };
        // the staging of SR rows in four phases (loads of a phase are consumed by the next one)
        auto phase = [&](int ph, int Rb) {
            Rb = __builtin_amdgcn_readfirstlane(Rb); asm volatile("" : "+s"(Rb));      // opaque: the per-load offsets are recomputed where they are used, not hoisted out of the pass loop (they are invariant there: ~100 registers)
            if (ph == 0) { fetch(0, 0, Rb); fetch(1, 1, Rb); }
            else if (ph == 1) { finish(0, 0, Rb); finish(1, 1, Rb); fetch(2, 0, Rb); fetch(3, 1, Rb); }
            else if (ph == 2) { finish(2, 0, Rb); finish(3, 1, Rb); fetch_sp(Rb); }
            else finish_sp(Rb);
        };
        // write the five tasks' pieces into the ring rows of image rows Rb .. Rb + nvalid - 1 (ring slot of image row R: (R + 1) mod NR)
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
        // weight pieces of pass p (fragment p / NG % NF, group p % NG): WB contiguous bytes
        u32x4 wq[C::NWU];
        auto wfetch = [&](int p) {
            const int fg = p % n_pps;
            const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(wsp) + (long long)fg * (C::WB / 16);
#pragma unroll
            for (int j = 0; j < C::NWU; ++j) wq[j] = src[min(t + 256 * j, C::WB / 16 - 1)];
        };
        auto wstore = [&]() {
#pragma unroll
            for (int j = 0; j < C::NWU; ++j) if (t + 256 * j < C::WB / 16) *reinterpret_cast<u32x4*>(s_w + (t + 256 * j) * 16) = wq[j];
        };

        wfetch(0);
        __syncthreads();                                    // (S0) channel tables visible
        // prologue: the first strip's window, rows r0 - 1 .. r0 + SR, in two rounds of SR rows
        const int r00 = strip0 * SR;
#pragma unroll 1
        for (int rbase = 0; rbase < SR + 2; rbase += SR) {
            const int Rb = r00 - 1 + rbase, nvalid = min(SR, SR + 2 - rbase);
            phase(0, Rb); phase(1, Rb); phase(2, Rb); phase(3, Rb);
            write(Rb, nvalid);
        }
        wstore();
        lds_barrier();                                      // (B1) window of strip 0 and W(0) published
        int p = 0;
#pragma unroll 1
        for (int ts = 0; ts < n_strip; ++ts) {
            const bool more = ts + 1 < n_strip;
            const int Rb = (strip0 + ts + 1) * SR + 1;      // first NEW image row of the next strip's window
#pragma unroll 1
            for (int ps = 0; ps < n_pps; ++ps, ++p) {
                const bool wnext = p + 1 < n_pass;
                if (wnext) wfetch(p + 1);
                lds_barrier();                              // (B2) the matrix waves hold W(p) in registers
                if (wnext) wstore();
                if (more) {
                    // phase ph of the next strip's staging rides on pass min(ph, n_pps - 1) of this strip
                    const bool lastp = ps == n_pps - 1;
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph) if (ph == ps || (lastp && ph > ps)) phase(ph, Rb);
                }
                lds_barrier();                              // (B1) W(p + 1) published; the matrix waves are done with pass p
            }
            if (more) { write(Rb, SR); lds_barrier(); }     // (B3) next strip's window published
        }
        if (fuse_sums) __syncthreads();                     // (Z)
#ifdef X6B_DBG_NOMAT
    } else if (false) {
#else
    } else {
#endif
        // ======================= matrix waves: wave = 16-pixel fragment =======================
        const int pf = wv;
        const bool lb = c0 == 0, rbd = c0 + 64 == W;
        int ax[3], axl[3];                                  // byte offsets of this lane's pixel operand inside a ring row, per kx
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            int sl = 16 * pf + l15 + 2 - kx;
            if (lb && pf == 0 && l15 == 1 && kx == 0) sl = 66;
            if (rbd && pf == 3 && l15 == 14 && kx == 2) sl = 67;
            ax[kx] = (K16 ? (l4 & 1) : l4) * PLANE + sl * 16;
            axl[kx] = ax[kx] + (l4 >= 2 ? 2 * PIECE : 0);   // K16: [x_h | x_l]
        }
        const char* const swl = s_w + (l15 * 4 + l4) * 16;
        const int x0 = c0 + 16 * pf + 4 * l4;
        const bool xact = (A.xin.act & 1) != 0; const float xslope = A.xin.slope;
        const float* __restrict__ xq = fuse_sums ? A.xin.data + (long long)k * A.xin.sstride : nullptr;
        float* __restrict__ gout = A.fga + (long long)k * A.fga_sstride;
        __syncthreads();                                    // (S0)
        lds_barrier();                                      // (B1)
#pragma unroll 1
        for (int ts = 0; ts < n_strip; ++ts) {
            const int r0 = (strip0 + ts) * SR;
            const bool first = r0 == 0, last = r0 + SR == H;
            int ro[SR + 2];                                 // ring row offsets of the window rows (image rows r0 - 1 + ii)
            { const int b0 = r0 % NR;
#pragma unroll
              for (int ii = 0; ii < SR + 2; ++ii) { int s = b0 + ii; s = s >= NR ? s - NR : s; ro[ii] = s * ROWB; } }
#pragma unroll 1
            for (int f = 0; f < NF; ++f) {
                f32x4 acc[SR];
                float4 xf[SR];
                const int ch = 16 * f + l15;
                const bool chv = ch < CI;
#pragma unroll 1
                for (int grp = 0; grp < NG; ++grp) {
                    u32x4 Wr[9][NV];
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                        for (int v = 0; v < NV; ++v) Wr[tp][v] = *reinterpret_cast<const u32x4*>(swl + (tp * NV + v) * 1024);
                    lds_barrier();                          // (B2) W in registers
                    if (grp == 0) {
#pragma unroll
                        for (int o = 0; o < SR; ++o) acc[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (fuse_sums) {
#pragma unroll
                            for (int o = 0; o < SR; ++o) xf[o] = chv ? *reinterpret_cast<const float4*>(xq + (long long)ch * HW + (r0 + o) * W + x0) : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                    const int goff = grp * 4 * PLANE;
                    constexpr int NGRP = (SR + 2) * 3;
                    u32x4 X[2][3];
                    auto issue = [&](int gi, u32x4 (&x)[3]) {
                        const int ii = gi / 3, kx = gi - 3 * ii;
                        if constexpr (K16) {
                            const char* p = lds + ro[ii] + ax[kx];
                            x[0] = *reinterpret_cast<const u32x4*>(p);                           // [x_h | x_h]
                            x[1] = *reinterpret_cast<const u32x4*>(p + PIECE);                   // [x_m | x_m]
                            x[2] = *reinterpret_cast<const u32x4*>(lds + ro[ii] + axl[kx]);      // [x_h | x_l]
                        } else {
                            const char* p = lds + ro[ii] + ax[kx] + goff;
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) x[pc] = *reinterpret_cast<const u32x4*>(p + pc * PIECE);
                        }
                    };
                    auto mm = [&](int o, const u32x4 (&x)[3], const u32x4 (&w)[NV]) {
                        f32x4 a = acc[o];
                        if constexpr (K16) { a = mfma_bf(x[2], w[1], a); a = mfma_bf(x[1], w[0], a); a = mfma_bf(x[0], w[0], a); }
                        else {      // pieces (x, w): (l,h), (h,l), (m,m), (m,h), (h,m), (h,h) — small terms first
                            a = mfma_bf(x[2], w[0], a); a = mfma_bf(x[0], w[2], a); a = mfma_bf(x[1], w[1], a);
                            a = mfma_bf(x[1], w[0], a); a = mfma_bf(x[0], w[1], a); a = mfma_bf(x[0], w[0], a);
                        }
                        acc[o] = a;
                    };
                    issue(0, X[0]);
#pragma unroll
                    for (int gi = 0; gi < NGRP; ++gi) {
                        const int ii = gi / 3, kx = gi - 3 * ii;
                        __builtin_amdgcn_sched_barrier(0);
                        if (gi + 1 < NGRP) issue(gi + 1, X[(gi + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        // image row r0 - 1 + ii meets output row o = ii + ky - 2 through tap row ky
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) {
                            const int o = ii + ky - 2;
                            if (o >= 0 && o < SR) mm(o, X[gi & 1], Wr[ky * 3 + kx]);
                            else if (ii == 1 && ky == 0) { if (first) mm(1, X[gi & 1], Wr[ky * 3 + kx]); }                 // image row 0 -> padded row -1 -> row 1
                            else if (ii == SR && ky == 2) { if (last) mm(SR - 2, X[gi & 1], Wr[ky * 3 + kx]); }            // image row H-1 -> padded row H -> row H-2
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (grp == NG - 1) {
                        // ---- fold: register r of acc[o] = pixel x0 + r of channel ch, image row r0 + o ----
                        const ChanFwd cf = s_ch[ch];
                        float fs = 0.f, fx = 0.f;
#pragma unroll
                        for (int o = 0; o < SR; ++o) {
                            float dd[4] = {acc[o][0], acc[o][1], acc[o][2], acc[o][3]};
                            if (fuse_sums) {
                                const float yy[4] = {xf[o].x, xf[o].y, xf[o].z, xf[o].w};
#pragma unroll
                                for (int l = 0; l < 4; ++l) {
                                    const float ym = yy[l] - cf.mean;
                                    if (xact) { const float vv = __builtin_fmaf(ym, cf.scale, cf.beta); dd[l] *= (vv > 0.f) ? 1.f : xslope; }
                                    fs += dd[l]; fx = __builtin_fmaf(dd[l], ym, fx);
                                }
                            }
                            if (chv) *reinterpret_cast<float4*>(gout + (long long)ch * HW + (r0 + o) * W + x0) = make_float4(dd[0], dd[1], dd[2], dd[3]);
                        }
                        if (fuse_sums) {
                            fs += __shfl_xor(fs, 16, 64); fs += __shfl_xor(fs, 32, 64); fx += __shfl_xor(fx, 16, 64); fx += __shfl_xor(fx, 32, 64);
                            if (l4 == 0) { float* s = s_sum + (wv * NFS + ch) * 2; s[0] += fs; s[1] += fx; }      // wave-private slots: no atomics
                        }
                    }
                    lds_barrier();                          // (B1)
                }
            }
            if (ts + 1 < n_strip) lds_barrier();            // (B3)
        }
        if (fuse_sums) __syncthreads();                     // (Z)
    }
    if (fuse_sums) {
        for (int i = tid; i < NFS * 2; i += 512) {
            const int q = i >> 1, which = i & 1;
            if (q < CI) {
                float v = (s_sum[(0 * NFS + q) * 2 + which] + s_sum[(1 * NFS + q) * 2 + which]) + (s_sum[(2 * NFS + q) * 2 + which] + s_sum[(3 * NFS + q) * 2 + which]);
                if (which) v *= s_ch[q].rstd;
                atomicAdd(A.fbsums + ((long long)k * CI + q) * 2 + which, (double)v);
            }
        }
    }
}

template <bool K16, int SR, int NG>
int launch_one(X6BArgs& A, hipStream_t st)
{
    using C = X6BCfg<K16, SR, NG>;
    const size_t lds_bytes = (size_t)C::RING + C::WB + sizeof(BwdC) * A.g.Cout + sizeof(ChanFwd) * A.NF * 16 + sizeof(float) * 4 * A.NF * 16 * 2;
    if (lds_bytes > 160 * 1024) return -3;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bwd_x6_kernel<K16, SR, NG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return (int)attr;
    mfvi_tl_family = 3;
    mfvi_launch((conv_bwd_x6_kernel<K16, SR, NG>), dim3(A.nx * A.nz), dim3(512), lds_bytes, st, A);
    return (int)hipGetLastError();
}

bool x6b_shape_ok(const ConvGeom& g)
{
    return g.ks == 3 && g.stride == 1 && !(g.W & 63) && g.H >= 4 && (g.Cout == 16 || g.Cout == 32 || g.Cout == 64) && g.Cin >= 4 && g.Cin <= MFVI_MAX_C && !(g.w_off & 3);
}

}  // namespace

// floats of split weight pieces for n_samples samples (0: shape not served)
long long x6_bwd_scratch_floats(const ConvGeom& g, int n_samples)
{
    if (!x6b_shape_ok(g)) return 0;
    const int NF = (g.Cin + 15) / 16, NG = g.Cout == 64 ? 2 : 1, NV = g.Cout == 16 ? 2 : 3;
    return (long long)NF * NG * 9 * NV * 256 * n_samples;
}

bool x6b_split_entry(const ConvGeom& g, long long dst_off, X6BSplitEntry* e)
{
    if (!x6b_shape_ok(g)) return false;
    e->w_off = g.w_off; e->dst_off = dst_off; e->CI = g.Cin; e->CO = g.Cout; e->NF = (g.Cin + 15) / 16; e->NG = g.Cout == 64 ? 2 : 1;
    e->k16 = g.Cout == 16 ? 1 : 0; e->units = e->NF * e->NG * 9 * 64; e->first_block = 0; e->pad = 0;
    return true;
}

int launch_x6b_split_all(const X6BSplitEntry* table_dev, int n_entries, int n_blocks, const float* w, long long wstride, int n_k, float* arena, hipStream_t st)
{
    if (n_entries <= 0 || n_blocks <= 0) return 0;
    hipLaunchKernelGGL(x6b_split_kernel, dim3(n_blocks, n_k), dim3(256), 0, st, table_dev, n_entries, w, wstride, arena);
    return (int)hipGetLastError();
}

// tune: T | sr << 8 (strips per block, output rows per strip; MFVI_TUNE_X6 stripped by the caller).  -2: shape not served / no scratch, -3: tiling not valid.
int launch_conv_bwd_data_x6(const GView& gy, const ConvGeom& g, const float* w, long long wstride, int tune, int n_samples, hipStream_t st, const FoldFuse& fuse)
{
    float* scratch = mfvi_tl_x6bw;
    if (!scratch || !x6b_shape_ok(g)) return -2;
    if (!fuse.ga || (fuse.ga_sstride & 3) || ((uintptr_t)fuse.ga & 15)) return -2;
    if (fuse.bsums && ((fuse.x.sstride & 3) || ((uintptr_t)fuse.x.data & 15))) return -2;
    if ((long long)max(g.Cin, g.Cout) * g.H * g.W >= (1LL << 29)) return -2;      // 32-bit element offsets per sample
    const int T = max(1, tune & 255), sr = (tune >> 8) & 255;
    const int want_sr = g.Cout == 16 ? 8 : g.Cout == 32 ? 4 : 2;          // SR * octets == 16 (the staging waves' task geometry)
    if (sr != want_sr || (g.H % sr)) return -3;
    X6BSplitEntry E;
    if (!x6b_split_entry(g, 0, &E)) return -2;
    const int NV = E.k16 ? 2 : 3;
    const int n_k = wstride ? n_samples : 1;
    if (!mfvi_tl_x6bw_ready) {      // no pass-wide split ran: this layer's own launch (dst_off 0: `scratch` is the layer's region)
        E.dst_off = 0;
        hipLaunchKernelGGL(x6b_split_one_kernel, dim3((E.units + 255) / 256, n_k), dim3(256), 0, st, E, w, wstride, scratch);
    }
    X6BArgs A{};
    A.gin = gy; A.xin = fuse.x; A.g = g;
    A.wsp = reinterpret_cast<const unsigned*>(scratch); A.wsp_stride_u4 = wstride ? (long long)E.units * NV : 0;
    A.fga = fuse.ga; A.fga_sstride = fuse.ga_sstride; A.fbsums = fuse.bsums;
    A.NF = E.NF; A.bands = g.W / 64; A.strips = g.H / sr; A.tpb = T;
    A.nx = A.bands * ((A.strips + T - 1) / T); A.nz = n_samples;
    if (g.Cout == 16) return launch_one<true, 8, 1>(A, st);
    if (g.Cout == 32) return launch_one<false, 4, 1>(A, st);
    return launch_one<false, 2, 2>(A, st);
}
