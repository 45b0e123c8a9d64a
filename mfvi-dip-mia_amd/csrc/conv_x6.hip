// K1 for the 3x3 stride-1 layers on maps >= 64 wide, on the BF16 matrix cores at fp32 accuracy ("bf16x6", round 3).
//
// Same arithmetic idea as conv_bww_x6.hip (every fp32 operand = three bf16 pieces, six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block of
// multiply-adds instead of eight fp32 matrix instructions), here with the INPUT CHANNELS as reduction dimension:
//   y[co][p] = sum_{ci, ky, kx} W[co][ci][ky][kx] * xpad[ci][p + (ky, kx)]          (BayTorch/modules/reparam_layers.py:26-37 behind the
//   ReflectionPad2d(1) of models/common.py:100-135), one matrix instruction = 16 pixels x 16 output channels x 32 input channels of one tap.
//   M = 16 consecutive pixels of a row (lane (g, m): channels 8 g .. 8 g + 7 of pixel m, one ds_read_b128 from a channel-octet plane in LDS),
//   N = 16 output channels (lane (g, n): the weights of channels 8 g .. 8 g + 7 for output channel n and this tap, pre-split pieces in
//   REGISTERS: 9 taps x 3 pieces x 4 VGPRs per 32-channel group).
// The 4 extra channels of the 32 n + 4 concat layers are a fifth octet plane (4 real + 4 zero channels) whose K slots are the three kx taps.
//
// Row streaming: an input row meets the three tap rows of three OUTPUT rows, so the nine pixel operands read for it (3 kx x 3 pieces)
// feed 54 matrix instructions — the accumulators of all SR output rows of the block's strip stay in registers and the block walks
// over the strip's SR + 2 input rows once per 32-channel group.  Block (512 threads, one per CU): 64-pixel band x SR output rows x 16 MF
// output channels; waves 0-3 issue matrix instructions (MF = 1: wave = 16-pixel fragment; MF = 2: wave = (output fragment, fragment pair)),
// waves 4-7 stage two input rows per stage (global dwords -> deferred BN / LeakyReLU -> three bf16 pieces -> one ds_write_b128 per pixel and
// octet: lane = pixel, wave = octet) into a four-row ring; one raw s_barrier per stage.  Epilogue: bias, float4 stores straight from the
// accumulators (register q of a fragment = pixel 4 (lane >> 4) + q of channel lane & 15), BN sums as conv_rp.hip.
//
// The weight pieces come from x6_split_weights_kernel (one launch in front of the convolution): the sampled fp32 slab W_k of the layer ->
// [group][tap][piece][co][octet][8] bf16 (+ the remainder rows [ky][piece][co][kx slot][8]) in a scratch region of the plan's workspace.
#include "common.h"
#include <type_traits>
#include <cstdlib>

thread_local float* mfvi_tl_x6w = nullptr;      // scratch of the op being launched (plan.hip), nullptr: bf16x6 forward not available
thread_local bool mfvi_tl_x6w_ready = false;    // the scratch already holds this pass's pieces (launch_x6_split_all ran behind the weight draw)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_bf(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// eight floats -> three packed bf16 octets (h, m, l), a = h + m + l exactly (common.h, split_pair_bf16x3)
__device__ __forceinline__ void split8(const float (&e)[8], u32x4& h, u32x4& m, u32x4& l)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) { unsigned hh, mm, ll; split_pair_bf16x3(e[2 * i], e[2 * i + 1], hh, mm, ll); h[i] = hh; m[i] = mm; l[i] = ll; }
}

// ---- weight pieces: thread = one 16-byte unit (row, co, octet) of sample k -> three pieces ----
struct X6WArgs { const float* w; long long wstride; unsigned* dst; long long dstride_u4; int Cin, Cout, COp, ncg, rem; long long w_off; int units; };
__global__ void x6_split_weights_kernel(X6WArgs A)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (u >= A.units) return;
    const int oct = u & 3, co = (u >> 2) % A.COp, row = (u >> 2) / A.COp;
    const float* __restrict__ w = A.w + (long long)k * A.wstride + A.w_off;
    float e[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (co < A.Cout) {
        if (row < A.ncg * 9) {
            const int cg = row / 9, tap = row - cg * 9;
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = w[((long long)co * A.Cin + cg * 32 + oct * 8 + j) * 9 + tap];
        } else if (oct < 3) {       // remainder rows: K slot group = kx
            const int ky = row - A.ncg * 9;
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = w[((long long)co * A.Cin + A.ncg * 32 + j) * 9 + ky * 3 + oct];
        }
    }
    u32x4 h, m, l; split8(e, h, m, l);
    u32x4* d = reinterpret_cast<u32x4*>(A.dst) + (long long)k * A.dstride_u4 + ((long long)row * 3 * A.COp + co) * 4 + oct;
    d[0] = h; d[(long long)A.COp * 4] = m; d[(long long)A.COp * 8] = l;
}

// the same for every bf16x6 layer of a plan in ONE launch behind the weight draw (a launch per layer in front of its convolution put
// ~10 us of dependent launch latency on the forward pass's critical path per layer)
__global__ void x6_split_all_kernel(const X6SplitEntry* __restrict__ table, int n_entries, const float* w, long long wstride, float* arena)
{
    int e = 0;
    while (e + 1 < n_entries && (int)blockIdx.x >= table[e + 1].first_block) ++e;
    const X6SplitEntry E = table[e];
    X6WArgs A;
    A.w = w; A.wstride = wstride; A.dst = reinterpret_cast<unsigned*>(arena + E.dst_off); A.dstride_u4 = (long long)E.units * 3;
    A.Cin = E.Cin; A.Cout = E.Cout; A.COp = E.COp; A.ncg = E.ncg; A.rem = E.rem; A.w_off = E.w_off; A.units = E.units;
    const int u = ((int)blockIdx.x - E.first_block) * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (u >= A.units) return;
    const int oct = u & 3, co = (u >> 2) % A.COp, row = (u >> 2) / A.COp;
    const float* __restrict__ ww = A.w + (long long)k * A.wstride + A.w_off;
    float e8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (co < A.Cout) {
        if (row < A.ncg * 9) {
            const int cg = row / 9, tap = row - cg * 9;
#pragma unroll
            for (int j = 0; j < 8; ++j) e8[j] = ww[((long long)co * A.Cin + cg * 32 + oct * 8 + j) * 9 + tap];
        } else if (oct < 3) {
            const int ky = row - A.ncg * 9;
#pragma unroll
            for (int j = 0; j < 4; ++j) e8[j] = ww[((long long)co * A.Cin + A.ncg * 32 + j) * 9 + ky * 3 + oct];
        }
    }
    u32x4 h, m, l; split8(e8, h, m, l);
    u32x4* d = reinterpret_cast<u32x4*>(A.dst) + (long long)k * A.dstride_u4 + ((long long)row * 3 * A.COp + co) * 4 + oct;
    d[0] = h; d[(long long)A.COp * 4] = m; d[(long long)A.COp * 8] = l;
}

struct X6FCfg {
    static constexpr int NOCT = 5;                        // four octets of the 32-channel group + the remainder plane
    static constexpr int PLANE = 80 * 16;                 // bytes of one (piece, octet) row plane: 66 pixels of 16 bytes, == 0 (mod 256): a ds_read_b128 lane group (two octets x complementary pixels) hits 64 distinct banks
    static constexpr int SLOT = 3 * NOCT * PLANE;         // one ring row
    static constexpr int NSLOT = 4;
};

struct X6FArgs {
    TView in; ConvGeom g; OutDesc out;
    const float* w; long long wstride;                    // fp32 slab (bias)
    const unsigned* wsp; long long wsp_stride_u4;         // split pieces, per sample stride in 16-byte units
    int COp, ncg, rem, bands, strips, tpb, nx, ny, nz;    // tpb: consecutive strips of a band per block
};

// MRG (round 4, one output fragment per wave only): the remainder plane rides on the LAST group's pass — its three tap-row operands stay in
// registers beside the group's 27 (36 more), the staging waves write the plane in the same stages, a window row's matrix work is its
// 3 x 18 + 18 instructions in one sweep.  The 36 -> 16 layer is then ONE pass of five stages per strip instead of two (the second one
// re-ran the whole row pipeline, barriers and window staging included, for a quarter of the matrix work).
template <int MF, int SR, bool MRG>
__global__ __launch_bounds__(512, 2) void conv_fwd_x6_kernel(X6FArgs A)
{
    using C = X6FCfg;
    constexpr int PF = MF;                                // 16-pixel fragments per matrix wave
    constexpr int COB = 16 * MF;
    constexpr int PLANE = C::PLANE, SLOT = C::SLOT, NOCT = C::NOCT;
    extern __shared__ __align__(16) char lds[];           // [4][3][5][80][16] ring of input rows | [27][COB][4][16] weight pieces of the next pass
    char* const s_w = lds + C::NSLOT * SLOT;
    __shared__ ChanFwd s_ch[MFVI_MAX_C + 8];
    __shared__ float s_bias[COB];
    __shared__ double s_red[4][COB][2];

    const ConvGeom& g = A.g;
    const int tid = threadIdx.x;
    const bool producer = tid >= 256;
    const int t = tid & 255, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    int bx, by, k;
    xcd_decode(blockIdx.x, A.nx, A.ny, A.nz, bx, by, k);
    const int band = bx % A.bands, strip0 = (bx / A.bands) * A.tpb;
    const int n_strip = min(A.tpb, A.strips - strip0);    // strips of this block: weights (one group), tables and the staging pipeline carry over
    const int c0 = band * 64, co0 = by * COB;
    const int Cin = g.Cin, Cout = g.Cout, H = g.H, W = g.W, HW = H * W;
    const int ncg = A.ncg, rem = A.rem;
    static_assert(!MRG || MF == 1, "merged remainder: 144 weight registers, one output fragment per wave");
    const int n_stage1 = (ncg + (MRG ? 0 : rem)) * ((SR + 2) / 2);     // stages per strip: two input rows per stage; the remainder plane is a pass of its own (its weights take the place of a group's in the registers)
    const int n_stage = n_strip * n_stage1;
    const float* __restrict__ xin = A.in.data + (long long)k * A.in.sstride;

    if (!producer) {
        // channel tables (8 spare entries: the zero channels of the remainder octet)
        for (int c = t; c < Cin + 8; c += 256) {
            ChanFwd f = chan_fwd(A.in, k, min(c, Cin - 1));
            if (c >= Cin) { f.scale = 0.f; f.beta = 0.f; }
            s_ch[c] = f;
        }
        if (t < COB) { const int co = co0 + t; s_bias[t] = (co < Cout && g.b_off >= 0) ? A.w[(long long)k * A.wstride + g.b_off + co] : 0.f; }
    }

    if (producer) {
        // ======================= staging waves: wave = octet, lane = pixel =======================
        __builtin_amdgcn_s_setprio(2);
        const int pw = wv;
        // third round: lanes 0-3 the two halo columns of this wave's octet (2 rows x 2 sides), lanes 4-36 a quarter of the remainder plane (2 rows x 66 pixels)
        const bool c_halo = lane < 4, c_rem = rem && lane >= 4 && lane < 37;
        const bool c_on = c_halo || c_rem;
        int c_row, c_col, c_oct;
        if (c_halo) { c_row = lane >> 1; c_col = (lane & 1) ? 65 : 0; c_oct = pw; }
        else { const int id = min(pw * 33 + lane - 4, 131); c_row = id / 66; c_col = id - c_row * 66; c_oct = 4; }
        int gxc = c0 - 1 + c_col; gxc = reflect_idx(gxc, W);          // image column of the third-round pixel
        constexpr int NSET = 4;                             // register sets: the loads of stages s + 2 .. s + 4 are in flight while stage s + 1 is stored (two sets = 49 KB in flight per CU left ~30 % of the load latency exposed)
        float va[NSET][8], vb[NSET][8], vc[NSET][8];
        const int ilast = n_stage - 1;
        auto fetch = [&](int set, int q) {                           // stage q = group cg, input rows 2 s, 2 s + 1 of the strip window
#if defined(X6F_DBG_NOPROD) || defined(X6F_DBG_NOLOAD)
            for (int j = 0; j < 8; ++j) { va[set][j] = 1.f + q; vb[set][j] = 2.f; vc[set][j] = 3.f; } return;
#endif
            q = min(q, ilast);
            const int ts = q / n_stage1, r0 = (strip0 + ts) * SR; q -= ts * n_stage1;
            const int cg = q / ((SR + 2) / 2), s = q - cg * ((SR + 2) / 2);
            const int ra = reflect_idx(r0 - 1 + 2 * s, H) * W, rb = reflect_idx(r0 + 2 * s, H) * W;
            const int cgm = min(cg, ncg - 1);                        // (remainder pass: the main rounds re-read the last group, unused)
            // (float2 loads of a pixel pair per lane — half as many load instructions against the 6-bit vmcnt — were measured: 2-6 % slower)
            const float* __restrict__ p = xin + (long long)(cgm * 32 + pw * 8) * HW + c0 + lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) { va[set][j] = p[(long long)j * HW + ra]; vb[set][j] = p[(long long)j * HW + rb]; }
            const int cb = c_halo ? cgm * 32 + pw * 8 : ncg * 32;
            const float* __restrict__ pc = xin + (c_row ? rb : ra) + gxc;
#pragma unroll
            for (int j = 0; j < 8; ++j) vc[set][j] = pc[(long long)min(cb + j, Cin - 1) * HW];
        };
        const bool xlrelu = (A.in.act & 1) != 0; const float xslope = A.in.slope;
        // channel constants of the running pass in registers (read from the LDS table once per pass, not once per staged pixel): the
        // transform is v = x * scale + shift, shift = beta - mean * scale formed once per channel
        float kas[8], kah[8], kcs[8], kch[8];
        int k_cg = -1;
        auto load_consts = [&](int cg) {
            const int ca = min(cg, ncg - 1) * 32 + pw * 8, cc = (cg < ncg && c_halo) ? ca : (rem ? Cin - 4 : Cin);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const ChanFwd f = s_ch[ca + j]; kas[j] = f.scale; kah[j] = __builtin_fmaf(-f.mean, f.scale, f.beta);
                const ChanFwd g2 = s_ch[cc + j]; kcs[j] = g2.scale; kch[j] = __builtin_fmaf(-g2.mean, g2.scale, g2.beta);
            }
            k_cg = cg;
        };
        auto put = [&](const float (&v)[8], const float (&ks)[8], const float (&kh)[8], char* dst) {
            float e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { float x = __builtin_fmaf(v[j], ks[j], kh[j]); if (xlrelu) x = __builtin_fmaxf(x, x * xslope); e[j] = x; }
#ifdef X6F_DBG_NOSPLIT
            u32x4 h = {__float_as_uint(e[0]), __float_as_uint(e[1]), __float_as_uint(e[2]), __float_as_uint(e[3])}, m = {__float_as_uint(e[4]), __float_as_uint(e[5]), __float_as_uint(e[6]), __float_as_uint(e[7])}, l = h;
#else
            u32x4 h, m, l; split8(e, h, m, l);
#endif
            *reinterpret_cast<u32x4*>(dst) = h; *reinterpret_cast<u32x4*>(dst + NOCT * PLANE) = m; *reinterpret_cast<u32x4*>(dst + 2 * NOCT * PLANE) = l;
        };
        auto store = [&](int set, int q) {
#ifdef X6F_DBG_NOPROD
            return;
#endif
            const int cg = (q % n_stage1) / ((SR + 2) / 2);
            if (cg != k_cg) load_consts(cg);
            const int sa = (2 * q) & 3, sb = (2 * q + 1) & 3;                  // ring slots of the stage's two rows
            if (cg < ncg) {
                char* base = lds + pw * PLANE + (1 + lane) * 16;
                put(va[set], kas, kah, base + sa * SLOT);
                put(vb[set], kas, kah, base + sb * SLOT);
            }
            if (MRG ? (c_halo || (c_rem && cg == ncg - 1)) : (cg < ncg ? c_halo : c_rem)) put(vc[set], kcs, kch, lds + (c_row ? sb : sa) * SLOT + c_oct * PLANE + c_col * 16);
        };
        (void)c_on;
        // weight pieces of pass pp (strip pp / passes, group pp % passes): 27 (remainder: 9) chunks of COB x 64 bytes -> s_w, requested in
        // stage 1 of the pass before and stored in its stage 2 (the matrix waves took their copy at the head of stage 0)
        constexpr int NWC = MRG ? 36 : 27;                  // 16-byte-unit chunks of a pass's weight pieces (merged: the group's 27 + the remainder's 9)
        constexpr int NWU = (NWC * COB * 4 + 255) / 256;
        const int passes = ncg + (MRG ? 0 : rem), n_pass = n_strip * passes;
        const unsigned* __restrict__ wsp = A.wsp + (long long)k * A.wsp_stride_u4 * 4;
        u32x4 wq[NWU];
        auto wfetch = [&](int pp) {
            const int cg = pp % passes;
            const int nu = (MRG ? ((rem && cg == ncg - 1) ? 36 : 27) : (cg == ncg ? 9 : 27)) * COB * 4;
#pragma unroll
            for (int j = 0; j < NWU; ++j) {
                const int u = min(t + 256 * j, nu - 1), c = u / (COB * 4), within = u - c * (COB * 4);
                const int src = (MRG && c >= 27) ? ncg * 27 + (c - 27) : cg * 27 + c;      // (merged: chunks 27 .. 35 = the remainder's three tap rows x three pieces)
                wq[j] = *reinterpret_cast<const u32x4*>(wsp + ((long long)src * A.COp * 4 + co0 * 4 + within) * 4);
            }
        };
        auto wstore = [&]() {
#pragma unroll
            for (int j = 0; j < NWU; ++j) if (t + 256 * j < NWC * COB * 4) *reinterpret_cast<u32x4*>(s_w + (t + 256 * j) * 16) = wq[j];
        };
        constexpr int SPP = (SR + 2) / 2;                   // stages per pass
        wfetch(0);
        fetch(0, 0); fetch(1, 1); fetch(2, 2); fetch(3, 3);
        __syncthreads();                                   // (S0) channel tables visible
        wstore();
        store(0, 0);
        fetch(0, 4);
        lds_barrier();                                     // (A) stage 0 and the first pass's weights published
        auto wstep = [&](int q) {                          // behind the x store of stage q + 1
            const int s1 = (q + 1) % SPP, pp = (q + 1) / SPP;
            if (s1 == 1 && pp + 1 < n_pass) wfetch(pp + 1);
            if (s1 == 2 && pp + 1 < n_pass) wstore();
        };
        // in flight at the top of a round: sets 1, 2, 3, 0 = stages q + 1 .. q + 4 (oldest first); every fetch is unconditional
        // (stages past the block's last re-read it and are never stored), so the compiler counts the loads and waits for the oldest set only
        for (int q = 0; q < n_stage; q += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (q + u >= n_stage) break;
                if (q + u + 1 < n_stage) { store((u + 1) & 3, q + u + 1); wstep(q + u); }
                fetch((u + 1) & 3, q + u + 5);
                lds_barrier();
            }
        }
        if (A.out.stats != nullptr) __syncthreads();       // (Z)
    } else {
        // ======================= matrix waves =======================
        const int cf = MF == 1 ? 0 : (wv & 1);                                  // output fragment of this wave
        const int pf0 = MF == 1 ? wv : 2 * (wv >> 1);                           // first pixel fragment
        const bool do_stats = A.out.stats != nullptr;
        if (do_stats) for (int q = lane; q < COB; q += 64) { s_red[wv][q][0] = 0.0; s_red[wv][q][1] = 0.0; }
        f32x4 acc[SR][PF];
        const char* const xl = lds + l4 * PLANE + (pf0 * 16 + l15) * 16;         // + slot, piece plane block, kx
        const char* const xr = lds + 4 * PLANE + (pf0 * 16 + l15 + min(l4, 2)) * 16;
        __syncthreads();                                   // (S0)
        lds_barrier();                                     // (A)
        int q = 0;
        float fs = 0.f, fq = 0.f;
        const int ch = cf * 16 + l15, co = co0 + ch;
        const float bi = s_bias[ch];
        // Weight pieces of the running pass: 9 taps x 3 pieces of a 32-channel group (the remainder pass: its 3 tap rows in entries 0 .. 2),
        // read into registers at the head of the pass from the LDS copy the staging waves made during the previous pass (straight from
        // global memory the 27 loads left the matrix pipe idle for ~1.5 us per pass; reloading entries behind their last use inside the
        // pass was built too: 86 - 250 spilled registers, 30 - 40 % slower).
        const char* const swl = s_w + ((cf * 16 + l15) * 4 + l4) * 16;
        for (int ts = 0; ts < n_strip; ++ts) {
#pragma unroll
        for (int o = 0; o < SR; ++o)
#pragma unroll
            for (int f = 0; f < PF; ++f) acc[o][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int cg = 0; cg < ncg + (MRG ? 0 : rem); ++cg) {
            const bool is_rem = !MRG && cg == ncg, with_rem = MRG && rem && cg == ncg - 1;
            u32x4 Wr[9][3];
            u32x4 Wm[MRG ? 3 : 1][3];                       // merged: the remainder's tap rows
            auto rows = [&](auto rem_c, auto wr_c) {
                constexpr bool REM = decltype(rem_c)::value, WR = decltype(wr_c)::value;
                if constexpr (WR) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int p = 0; p < 3; ++p) Wm[ky][p] = *reinterpret_cast<const u32x4*>(swl + ((27 + ky * 3 + p)) * COB * 64);
                }
#pragma unroll
                for (int tp = 0; tp < (REM ? 3 : 9); ++tp)
#pragma unroll
                    for (int p = 0; p < 3; ++p) Wr[tp][p] = *reinterpret_cast<const u32x4*>(swl + (tp * 3 + p) * COB * 64);
                // groups of a row: (fragment f, kx) -> three pieces of the pixel operand, 18 matrix instructions against the three tap rows
                // (remainder pass: one group per fragment).  The reads of group j + 1 are issued in front of group j's matrix instructions,
                // those of the next row's first group in front of this row's last one when both rows belong to one stage.
                constexpr int GF = REM ? 1 : 3, GM = PF * GF, GR = GM + (WR ? PF : 0);      // groups of a row: the main ones, then (merged) the remainder's
                u32x4 X[2][3];
                auto issue = [&](int j, int slot, u32x4 (&x)[3]) {
                    const bool rg = WR && j >= GM;
                    const int f = rg ? j - GM : j / GF, kx = rg ? 0 : j - f * GF;
                    const char* p = ((REM || rg) ? xr : xl + kx * 16) + slot + f * 256;
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) x[pc] = *reinterpret_cast<const u32x4*>(p + pc * NOCT * PLANE);
                };
#pragma unroll
                for (int ii = 0; ii < SR + 2; ++ii) {
                    const int slot = ((2 * q + (ii & 1)) & 3) * SLOT;
                    const int par = (ii * GR) & 1;                              // register set of this row's first group
#ifndef X6F_DBG_NOMFMA
                    if (ii % 2 == 0) issue(0, slot, X[par]);                    // (odd rows: requested under the previous row)
#endif
#pragma unroll
                    for (int j = 0; j < GR; ++j) {
                        const bool rg = WR && j >= GM;
                        const int f = rg ? j - GM : j / GF, kx = rg ? 0 : j - f * GF;
#ifdef X6F_DBG_NOMFMA
                        continue;
#endif
                        __builtin_amdgcn_sched_barrier(0);
                        if (j + 1 < GR) issue(j + 1, slot, X[(par + j + 1) & 1]);
                        else if (ii % 2 == 0) issue(0, ((2 * q + 1) & 3) * SLOT, X[(par + GR) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        // pieces: (x, w) in {(l,h), (h,l), (m,m), (m,h), (h,m), (h,h)}
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) {
                            const int o = ii - ky;
                            if (o < 0 || o >= SR) continue;
                            const u32x4 (&Wk)[3] = rg ? Wm[WR ? ky : 0] : Wr[REM ? ky : ky * 3 + kx];
                            const u32x4 (&Xo)[3] = X[(par + j) & 1];
                            f32x4 a = acc[o][f];
                            a = mfma_bf(Xo[2], Wk[0], a); a = mfma_bf(Xo[0], Wk[2], a); a = mfma_bf(Xo[1], Wk[1], a);
                            a = mfma_bf(Xo[1], Wk[0], a); a = mfma_bf(Xo[0], Wk[1], a); a = mfma_bf(Xo[0], Wk[0], a);
                            acc[o][f] = a;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (ii & 1) { lds_barrier(); ++q; }
                }
            };
            if (is_rem) rows(std::true_type{}, std::false_type{});
            else if (with_rem) { if constexpr (MRG) rows(std::false_type{}, std::true_type{}); }
            else rows(std::false_type{}, std::false_type{});
        }
        // ---- epilogue: register r of (row o, fragment f) = pixel c0 + 16 (pf0 + f) + 4 l4 + r, channel co0 + 16 cf + l15 ----
        if (co < Cout) {
            float* __restrict__ yo = A.out.data + (long long)k * A.out.sstride + (long long)co * HW + (long long)(strip0 + ts) * SR * W + c0 + pf0 * 16 + 4 * l4;
#pragma unroll
            for (int o = 0; o < SR; ++o)
#pragma unroll
                for (int f = 0; f < PF; ++f) {
                    const float v0 = acc[o][f][0] + bi, v1 = acc[o][f][1] + bi, v2 = acc[o][f][2] + bi, v3 = acc[o][f][3] + bi;
                    *reinterpret_cast<float4*>(yo + o * W + f * 16) = make_float4(v0, v1, v2, v3);
                    fs += (v0 + v1) + (v2 + v3);
                    fq = __builtin_fmaf(v0, v0, __builtin_fmaf(v1, v1, __builtin_fmaf(v2, v2, __builtin_fmaf(v3, v3, fq))));
                }
        }
        }
        if (do_stats) {
            fs += __shfl_xor(fs, 16, 64); fs += __shfl_xor(fs, 32, 64); fq += __shfl_xor(fq, 16, 64); fq += __shfl_xor(fq, 32, 64);
            if (l4 == 0) { s_red[wv][ch][0] = (double)fs; s_red[wv][ch][1] = (double)fq; }
            __syncthreads();                               // (Z)
            if (t < COB * 2) {
                const int c = t >> 1, which = t & 1;
                // waves that own fragment cf = c >> 4 (MF = 1: all four; MF = 2: waves cf, cf + 2)
                double v = 0.0;
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2) if (MF == 1 || (w2 & 1) == (c >> 4)) v += s_red[w2][c][which];
                if (co0 + c < Cout) atomicAdd(A.out.stats + ((long long)k * Cout + co0 + c) * 2 + which, v);
            }
        }
    }
}

template <int MF, int SR, bool MRG = false>
int launch_one(X6FArgs& A, hipStream_t st)
{
    using C = X6FCfg;
    constexpr size_t lds_bytes = (size_t)C::NSLOT * C::SLOT + (size_t)(MRG ? 36 : 27) * 16 * MF * 64;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_x6_kernel<MF, SR, MRG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (attr != hipSuccess) return (int)attr;
    mfvi_tl_family = 3;
    mfvi_launch((conv_fwd_x6_kernel<MF, SR, MRG>), dim3(A.nx * A.ny * A.nz), dim3(512), lds_bytes, st, A);
    return (int)hipGetLastError();
}

}  // namespace

bool x6_split_entry(const ConvGeom& g, long long dst_off, X6SplitEntry* e)
{
    if (x6_fwd_scratch_floats(g, 1) == 0) return false;
    e->w_off = g.w_off; e->dst_off = dst_off; e->Cin = g.Cin; e->Cout = g.Cout; e->COp = (g.Cout + 31) / 32 * 32;
    e->ncg = g.Cin / 32; e->rem = (g.Cin & 31) ? 1 : 0; e->units = (e->ncg * 9 + (e->rem ? 3 : 0)) * e->COp * 4; e->first_block = 0; e->pad = 0;
    return true;
}

int launch_x6_split_all(const X6SplitEntry* table_dev, int n_entries, int n_blocks, const float* w, long long wstride, int n_k, float* arena, hipStream_t st)
{
    if (n_entries <= 0 || n_blocks <= 0) return 0;
    hipLaunchKernelGGL(x6_split_all_kernel, dim3(n_blocks, n_k), dim3(256), 0, st, table_dev, n_entries, w, wstride, arena);
    return (int)hipGetLastError();
}

// floats of split weight pieces for n_samples samples (0: shape not served)
long long x6_fwd_scratch_floats(const ConvGeom& g, int n_samples)
{
    if (g.ks != 3 || g.stride != 1 || (g.W & 63) || g.Cin < 32 || ((g.Cin & 31) != 0 && (g.Cin & 31) != 4)) return 0;
    const int ncg = g.Cin / 32, rem = g.Cin & 31 ? 1 : 0, COp = (g.Cout + 31) / 32 * 32;
    const long long units = (long long)(ncg * 9 + (rem ? 3 : 0)) * 3 * COp * 4;     // 16-byte units per sample
    return units * 4 * n_samples;
}

// tune: mf | sr << 8 | T << 16 (| MFVI_TUNE_X6 stripped by the caller; T = strips per block).  -2: shape not served / no scratch, -3: tiling not valid.
int launch_conv_fwd_x6(const TView& in, const ConvGeom& g, const float* w, long long wstride, OutDesc out, int tune, int n_samples, hipStream_t st)
{
    float* scratch = mfvi_tl_x6w;
    if (!scratch) return -2;
    if (g.ks != 3 || g.stride != 1 || (g.W & 63) || g.Cin < 32 || g.Cin > MFVI_MAX_C || (g.w_off & 3)) return -2;
    const int r32 = g.Cin & 31;
    if (r32 != 0 && r32 != 4) return -2;
    if (in.act & MFVI_ACT_SQUARE) return -2;
    if ((out.sstride & 3) || ((uintptr_t)out.data & 15)) return -2;
    if ((long long)max(g.Cin, g.Cout) * g.H * g.W >= (1LL << 31)) return -2;
    const int mf = tune & 255, sr = (tune >> 8) & 15, mrg = (tune >> 12) & 1;      // bit 12: the remainder plane rides on the last group's pass (mf = 1, Cin = 32 n + 4)
    if ((mf != 1 && mf != 2) || sr != 8 || (g.H % sr) || (mrg && (mf != 1 || r32 == 0))) return -3;
    const int ncg = g.Cin / 32, rem = r32 ? 1 : 0, COp = (g.Cout + 31) / 32 * 32;
    const int n_k = wstride ? n_samples : 1;
    const long long units = (long long)(ncg * 9 + (rem ? 3 : 0)) * COp * 4;          // threads of the split kernel per sample
    X6WArgs WA{};
    WA.w = w; WA.wstride = wstride; WA.dst = reinterpret_cast<unsigned*>(scratch); WA.dstride_u4 = units * 3;
    WA.Cin = g.Cin; WA.Cout = g.Cout; WA.COp = COp; WA.ncg = ncg; WA.rem = rem; WA.w_off = g.w_off; WA.units = (int)units;
    if (!mfvi_tl_x6w_ready) hipLaunchKernelGGL(x6_split_weights_kernel, dim3((unsigned)((units + 255) / 256), n_k), dim3(256), 0, st, WA);
    X6FArgs A{};
    A.in = in; A.g = g; A.out = out; A.w = w; A.wstride = wstride;
    A.wsp = reinterpret_cast<const unsigned*>(scratch); A.wsp_stride_u4 = wstride ? units * 3 : 0;
    A.COp = COp; A.ncg = ncg; A.rem = rem; A.bands = g.W / 64;
    const int T = max(1, (tune >> 16) & 255);
    A.strips = g.H / sr; A.tpb = T;
    A.nx = A.bands * ((A.strips + T - 1) / T); A.ny = (g.Cout + 16 * mf - 1) / (16 * mf); A.nz = n_samples;
    if (mf == 1 && mrg) return launch_one<1, 8, true>(A, st);
    if (mf == 1) return launch_one<1, 8>(A, st);
    return launch_one<2, 8>(A, st);
}
