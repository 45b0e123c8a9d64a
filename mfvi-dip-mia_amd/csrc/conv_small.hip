// conv_small.hip — 3x3 stride-1 forward and fused-fold backward-data on the small maps at the bottom of the hour-glass (8x8, 16x16):
// one stage, everything in flight.
//
// Reference op: BayTorch/modules/reparam_layers.py:26-37 behind models/common.py:100-135 (ReflectionPad2d(1) + Conv2d 3x3) — the layers
// `deeper` / `up` of the 8x8 and 16x16 scales of skip() (models/skip.py:60-110).
//
// Why a kernel of its own: on these maps the staged kernels (conv_mfma.hip FLAT tiles, conv_rp.hip narrow maps) run as a chain of 4 ... 33
// stages of [global load -> LDS -> barrier -> a few dozen MFMAs], one memory round trip per stage, with one block per CU and nothing beside it
// to hide the latency: 128 -> 128 on an 8x8 map takes 20-25 us for 3.8 us of matrix work per wave (NOTES R3.22).  Here a block owns one
// (sample, 16 output channels, 64 pixels = TR whole image rows) and its WHOLE reduction fits in LDS:
//   window  [Cin/16 groups][ (TR + 2) x (W + 2) pixels ][16]   (reflection pad in the index math, deferred BN + LeakyReLU applied on load)
//   weights [9 taps][Cin/16 groups][16 output channels][16]     (W_k of the sample, from the slab sample_weights_kernel wrote)
// so every global load of the block is issued before anything waits (one round trip), there is ONE barrier, and all 8 waves then run matrix
// instructions: wave = (pixel fragment of 16, half of the channel groups); the two halves of a fragment meet through LDS.  The 16 floats of a
// (group, pixel) / (tap, group, output channel) entry are ordered [channel mod 4][k-step] so that one ds_read_b128 gives a lane its operand of
// four consecutive k-steps: two reads per four v_mfma_f32_16x16x4_f32.
//
// MODE 1 (backward-data with the fold of the input tensor in its epilogue, as conv_rp.hip MODE 1): the reduction runs over the layer's OUTPUT
// channels, the window holds dy (BN-backward of the following BatchNorm applied on load, zeros outside the image), the weights go to LDS
// transposed and flipped.  The gradient is formed on the UN-padded domain; the adjoint of ReflectionPad2d(1) adds to image row 1 what the
// padded row -1 would have received (to row H-2: row H; columns alike): for those pixels the tap that reads row 2 reads a spare window row
// S1 = dy[2] + dy[0] instead (row H-3: S2 = dy[H-3] + dy[H-1]), the tap that reads column 2 a spare window column C1 = dy[.][2] + dy[.][0]
// (column W-3: C2), corners the four-fold sums — spare rows / columns are a second, small LDS pass over the staged window, and a lane picks
// its three row and three column offsets once.  Epilogue: LeakyReLU' of the input view, BN-backward sums, ga written once.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

struct SmBwd { float mean, qc, c1, k2; };      // dy = (y - mean) * qc + (ga * c1 + k2)   (conv_rp.hip RpBwd)

struct SmArgs {
    TView xin; ConvGeom g; const float* w; long long wstride; OutDesc out;
    GView gin; float* fga; long long fga_sstride; double* fbsums;      // MODE 1: dy view, gradient of the input tensor (written), its BN-backward sums
    int nx, ny, nz;       // tiles, output fragments, samples
    int tr, wsh;          // image rows per tile (tr * W == 64), log2(W)
};

constexpr int SM_MAXC = 144;      // reduction channels (LDS: 9 groups of window + weights = 143 KB at 16x16)
constexpr int SM_NXJ = 36;        // window items per thread: one window pixel x every 4th of 9 x 16 channels
constexpr int SM_NWJ = 2;         // weight items (9 taps x 4 channels of one (group, m, residue)) per thread: 9 x 64 / 512

template <int MODE>
__global__ __launch_bounds__(512) void conv_sm_kernel(SmArgs A)
{
    extern __shared__ __align__(16) float s_dyn[];      // window | weights
    __shared__ ChanFwd s_ch[MODE == 0 ? SM_MAXC : 16];      // MODE 1: the 16 channels of the input tensor this block folds
    __shared__ SmBwd s_chb[MODE == 1 ? SM_MAXC : 1];
    __shared__ float s_bias[16];
    __shared__ __align__(16) float s_comb[4][64][4];
    __shared__ double s_red[4][16][2];

    const ConvGeom& g = A.g;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    int bx, by, k;
    xcd_decode(blockIdx.x, A.nx, A.ny, A.nz, bx, by, k);
    const int m0 = by * 16, W = g.W, HW = g.H * W, WW = W + 2, WP = (A.tr + 2) * WW;
    const int RED = MODE == 0 ? g.Cin : g.Cout, MOUT = MODE == 0 ? g.Cout : g.Cin;      // reduction / output channels
    const int Cin = RED, NG = (RED + 15) >> 4;       // (the staging code below says Cin for "reduction channels")
    const int py0 = bx * A.tr;
    // LDS window: MODE 0 (TR + 2) x (W + 2); MODE 1 two spare columns C1, C2 and one (two when the map is ONE tile) spare rows S1 / S2
    const int LW = MODE == 0 ? WW : W + 4, S1R = A.tr + 2, S2R = A.nx == 1 ? A.tr + 3 : A.tr + 2;
    const int WPL = MODE == 0 ? WP : (S2R + 1) * LW;
    float* __restrict__ s_x = s_dyn;
    float* __restrict__ s_w = s_dyn + NG * WPL * 16;
    const float* __restrict__ wk = A.w + (long long)k * A.wstride;

    // ---- every global load of the block, issued before anything waits.  No division in the per-item index math (the first version
    //      spent ~15 us per block on i / WP, q / rowq, fl / 9 of ~90 items per thread): a thread owns ONE window pixel (128 lanes per
    //      channel row, WP <= 128 of them in use) and walks the channels cb, cb + 4, ...: group = j >> 2, k-step = j & 3 are compile-time ----
    const float* __restrict__ xk = MODE == 0 ? A.xin.data + (long long)k * A.xin.sstride : A.gin.ga + (long long)k * A.gin.gstride;
    const float* __restrict__ yk = (MODE == 1 && A.gin.stats) ? A.gin.y + (long long)k * A.gin.ystride : nullptr;
    // (lane order: 4 consecutive lanes = the 4 channel residues of one window pixel, so a wave's ds_write_b128 covers 1 KB of contiguous LDS;
    //  with the pixel as the fast lane index the 64-byte pixel stride put 16 lanes on the same banks)
    const int wp = tid >> 2, cb = tid & 3;
    const bool wp_ok = wp < WP;
    int xoff = 0, lwp = 0; bool inside = true;      // lwp: this window pixel's LDS index
    { const int wr = wp / WW, wc = wp - wr * WW; lwp = wr * LW + wc;
      if (MODE == 0) xoff = reflect_idx(py0 - 1 + wr, g.H) * W + reflect_idx(wc - 1, W);
      else { const int gy = py0 - 1 + wr, gx = wc - 1; inside = gy >= 0 && gy < g.H && gx >= 0 && gx < W; xoff = inside ? gy * W + gx : 0; } }
    float xr[SM_NXJ], yr[MODE == 1 ? SM_NXJ : 1];
#pragma unroll
    for (int j = 0; j < SM_NXJ; ++j) {
        const int c = cb + 4 * j;
        const bool ok = wp_ok && c < Cin && inside;
        xr[j] = ok ? xk[(long long)c * HW + xoff] : 0.f;
        if (MODE == 1) yr[j] = (ok && yk) ? yk[(long long)c * HW + xoff] : 0.f;
    }
    // MODE 1: raw values of the input tensor at this lane's four outputs (fold: LeakyReLU', x-hat of the BN-backward sums)
    float xq[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1 && wv < 4 && A.fbsums) {
        const float* __restrict__ xin = A.xin.data + (long long)k * A.xin.sstride + py0 * W + 16 * (wv & 3) + l15;
#pragma unroll
        for (int q = 0; q < 4; ++q) if (m0 + 4 * l4 + q < MOUT) xq[q] = xin[(long long)(m0 + 4 * l4 + q) * HW];
    }
    // weights: item = (group gw, output channel mw, residue lw) = the 9 taps of the 4 channels 16 gw + lw + 4 i: four 36-byte runs of the
    // global row (dword alignment), nine float4 {i = 0..3} in LDS; lanes in (mw, lw) order write 1 KB of contiguous LDS per tap
    const int NWI = NG * 64;
    float wt[SM_NWJ][4][9];
#pragma unroll
    for (int j = 0; j < SM_NWJ; ++j) {
        const int it = tid + 512 * j, gw = it >> 6, mw = (it >> 2) & 15, lw = it & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = min(16 * gw + lw + 4 * i, Cin - 1);
            // MODE 0: row (output m0 + mw, input c); MODE 1: row (output channel of the layer c = reduction, input channel m0 + mw = this block's output)
            const float* __restrict__ src = wk + g.w_off + (MODE == 0 ? ((long long)(m0 + mw) * g.Cin + c) : ((long long)c * g.Cin + min(m0 + mw, g.Cin - 1))) * 9;
            const bool ok = it < NWI && 16 * gw + lw + 4 * i < Cin && (MODE == 0 || m0 + mw < g.Cin);
            const f4u a = ok ? *reinterpret_cast<const f4u*>(src) : f4u{0.f, 0.f, 0.f, 0.f};
            const f4u b = ok ? *reinterpret_cast<const f4u*>(src + 4) : f4u{0.f, 0.f, 0.f, 0.f};
            const float c8 = ok ? src[8] : 0.f;
            wt[j][i][0] = a.x; wt[j][i][1] = a.y; wt[j][i][2] = a.z; wt[j][i][3] = a.w;
            wt[j][i][4] = b.x; wt[j][i][5] = b.y; wt[j][i][6] = b.z; wt[j][i][7] = b.w; wt[j][i][8] = c8;
        }
    }
    // per-channel constants of the deferred BN, bias (channels that pad the last group: zero window entries AND zero weights, written above)
    if (MODE == 0) {
        for (int c = tid; c < Cin; c += 512) s_ch[c] = chan_fwd(A.xin, k, c);
        if (tid < 16) s_bias[tid] = g.b_off >= 0 ? wk[g.b_off + m0 + tid] : 0.f;
    } else {
        for (int c = tid; c < Cin; c += 512) { const ChanBwd b = chan_bwd(A.gin, k, c); SmBwd r; r.mean = b.mean; r.qc = -b.c1 * b.c3 * b.rstd; r.c1 = b.c1; r.k2 = -b.c1 * b.c2; s_chb[c] = r; }
        if (tid < 16 && A.fbsums) s_ch[tid] = chan_fwd(A.xin, k, min(m0 + tid, MOUT - 1));
    }
    __syncthreads();
    const int act = A.xin.act; const float slope = A.xin.slope;
    if (wp_ok) {
#pragma unroll
        for (int G = 0; G < SM_NXJ / 4; ++G) {
            if (G < NG) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = cb + 16 * G + 4 * i;
                    if (MODE == 0) v[i] = c < Cin ? apply_fwd(s_ch[min(c, Cin - 1)], xr[4 * G + i], act, slope) : 0.f;
                    else {
                        const SmBwd kb = s_chb[min(c, Cin - 1)];
                        const float e = yk ? __builtin_fmaf(yr[MODE == 1 ? 4 * G + i : 0] - kb.mean, kb.qc, __builtin_fmaf(xr[4 * G + i], kb.c1, kb.k2)) : xr[4 * G + i];
                        v[i] = (c < Cin && inside) ? e : 0.f;
                    }
                }
                *reinterpret_cast<float4*>(&s_x[((G * WPL + lwp) << 4) + (cb << 2)]) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < SM_NWJ; ++j) {
        const int it = tid + 512 * j, gw = it >> 6, mw = (it >> 2) & 15, lw = it & 3;
        if (it < NWI) {
            float* __restrict__ d = s_w + (((gw << 4) + mw) << 4) + (lw << 2);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                *reinterpret_cast<float4*>(d + (((MODE == 0 ? tap : 8 - tap) * NG) << 8)) = make_float4(wt[j][0][tap], wt[j][1][tap], wt[j][2][tap], wt[j][3][tap]);
        }
    }
    __syncthreads();

    if constexpr (MODE == 1) {
        // ---- spare rows / columns of the reflection adjoint, from the staged window (regular entries only: no order inside the pass).
        //      Window row of image row i: i - py0 + 1; column of image column j: j + 1.  Items: (group, spare pixel, residue) as float4.
        const bool has1 = py0 == 0, has2 = py0 + A.tr == g.H;                   // this tile holds image row 1 / H-2 (both only when the map is one tile)
        const int TR = A.tr, NSP = 2 * WW + 2 * (TR + 4);                      // the two spare rows over the regular columns, then the two spare columns over all rows
        for (int it = tid; it < NG * NSP * 4; it += 512) {
            const int r4 = it & 3, sp = (it >> 2) % NSP, G = (it >> 2) / NSP;
            int dr, dc, ra, rb, ca, cb2; bool valid;                           // destination; source rows ra (+ rb), source columns ca (+ cb2); -1: none
            if (sp < 2 * WW) {
                const int second = sp >= WW;
                dc = sp - second * WW; ca = dc; cb2 = -1;
                valid = second ? has2 : has1; dr = second ? S2R : S1R; ra = second ? TR - 2 : 3; rb = second ? TR : 1;
            } else {
                const int q = sp - 2 * WW, second = q >= TR + 4, rr = q - second * (TR + 4);
                dc = W + 2 + second; ca = second ? W - 2 : 3; cb2 = second ? W : 1;
                if (rr < TR + 2) { valid = true; dr = rr; ra = rr; rb = -1; }
                else if (rr == TR + 2) { valid = has1; dr = S1R; ra = 3; rb = 1; }
                else { valid = has2; dr = S2R; ra = TR - 2; rb = TR; }
            }
            if (!valid) ra = -1;
            if (ra < 0) continue;
            const float* __restrict__ base = s_x + ((G * WPL) << 4) + (r4 << 2);
            float4 v = *reinterpret_cast<const float4*>(base + ((ra * LW + ca) << 4));
            auto add = [&](int rr, int cc) { const float4 u = *reinterpret_cast<const float4*>(base + ((rr * LW + cc) << 4)); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; };
            if (rb >= 0) add(rb, ca);
            if (cb2 >= 0) { add(ra, cb2); if (rb >= 0) add(rb, cb2); }
            *reinterpret_cast<float4*>(s_x + ((G * WPL + dr * LW + dc) << 4) + (r4 << 2)) = v;
        }
        __syncthreads();
    }

    // ---- matrix phase: wave = (pixel fragment f, half h of the channel groups) ----
    const int f = wv & 3, h = wv >> 2;
    const int p = 16 * f + l15, trow = p >> A.wsh, tcol = p & (W - 1);
    int ro[3] = {trow * LW, (trow + 1) * LW, (trow + 2) * LW}, co[3] = {tcol, tcol + 1, tcol + 2};
    if (MODE == 1) {
        const int r = py0 + trow;
        if (r == 1) ro[2] = S1R * LW;
        if (r == g.H - 2) ro[0] = S2R * LW;
        if (tcol == 1) co[2] = W + 2;
        if (tcol == W - 2) co[0] = W + 3;
    }
    const int gsplit = (NG + 1) >> 1, g0 = h ? gsplit : 0, g1 = h ? NG : gsplit;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int gi = g0; gi < g1; ++gi) {
        const float* __restrict__ xb = s_x + ((gi * WPL) << 4) + (l4 << 2);
        const float* __restrict__ wb_ = s_w + (((gi << 4) + l15) << 4) + (l4 << 2);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(wb_ + (((ky * 3 + kx) * NG) << 8));
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(xb + ((ro[ky] + co[kx]) << 4));
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i], b4[i], acc, 0, 0, 0);
            }
    }
    if (h) *reinterpret_cast<f32x4*>(&s_comb[f][lane][0]) = acc;
    __syncthreads();
    const bool do_stats = MODE == 0 ? A.out.stats != nullptr : A.fbsums != nullptr;
    if (!h) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(&s_comb[f][lane][0]);
        // (a tile is TR whole rows: pixel p of the tile is contiguous in memory)
        float* __restrict__ yout = (MODE == 0 ? A.out.data + (long long)k * A.out.sstride : A.fga + (long long)k * A.fga_sstride) + (long long)m0 * HW + py0 * W + p;
        const int xact = A.xin.act; const float xslope = A.xin.slope;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = 4 * l4 + q;
            float v = acc[q] + o[q], a_, b_;
            if (MODE == 0) { v += s_bias[ch]; a_ = v; b_ = v * v; }
            else {
                // the fold (conv_rp.hip fold_do): LeakyReLU' of the input view, sums of ga and ga * (x - mean) (x rstd below)
                a_ = 0.f; b_ = 0.f;
                if (do_stats) {
                    const ChanFwd cf = s_ch[ch];
                    const float ym = xq[q] - cf.mean;
                    if (xact) { const float vv = __builtin_fmaf(ym, cf.scale, cf.beta); v *= (vv > 0.f) ? 1.f : xslope; }
                    a_ = v; b_ = v * ym * cf.rstd;
                }
            }
            const bool chok = m0 + ch < MOUT;
            if (chok) yout[ch * HW] = v;
            if (do_stats) {
                if (!chok) { a_ = 0.f; b_ = 0.f; }
#pragma unroll
                for (int s2 = 8; s2 > 0; s2 >>= 1) { a_ += __shfl_xor(a_, s2, 64); b_ += __shfl_xor(b_, s2, 64); }
                if (l15 == 0) { s_red[f][ch][0] = (double)a_; s_red[f][ch][1] = (double)b_; }
            }
        }
    }
    if (do_stats) {
        __syncthreads();
        if (tid < 32) {
            const int ch = tid >> 1, which = tid & 1;
            if (m0 + ch < MOUT)
                atomicAdd((MODE == 0 ? A.out.stats : A.fbsums) + ((long long)k * MOUT + m0 + ch) * 2 + which,
                          s_red[0][ch][which] + s_red[1][ch][which] + s_red[2][ch][which] + s_red[3][ch][which]);
        }
    }
}

}  // namespace

// Return -2 when the shape is not served.
static int sm_launch(int mode, SmArgs& A, int red, int n_samples, hipStream_t st)
{
    const ConvGeom& g = A.g;
    const int tr = 64 / g.W;
    const int NG = (red + 15) >> 4, WP = (tr + 2) * (g.W + 2);
    A.nx = g.H / tr; A.ny = ((mode == 0 ? g.Cout : g.Cin) + 15) / 16; A.nz = n_samples; A.tr = tr; A.wsh = g.W == 8 ? 3 : 4;
    const int WPL = mode == 0 ? WP : (tr + 3 + (A.nx == 1 ? 1 : 0)) * (g.W + 4);
    if (WP > 128 || NG * 4 > SM_NXJ || NG * 64 > SM_NWJ * 512) return -2;
    const size_t lds_bytes = sizeof(float) * ((size_t)NG * WPL * 16 + (size_t)9 * NG * 256);
    if (lds_bytes > 152 * 1024) return -2;
    static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sm_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sm_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    if (attr0 != hipSuccess || attr1 != hipSuccess) return (int)(attr0 != hipSuccess ? attr0 : attr1);
    mfvi_tl_family = 4;
    if (mode == 0) mfvi_launch(conv_sm_kernel<0>, dim3(A.nx * A.ny * A.nz), dim3(512), lds_bytes, st, A);
    else mfvi_launch(conv_sm_kernel<1>, dim3(A.nx * A.ny * A.nz), dim3(512), lds_bytes, st, A);
    return (int)hipGetLastError();
}

int launch_conv_fwd_small(const TView& in, const ConvGeom& g, const float* w, long long wstride, OutDesc out, int n_samples, hipStream_t st)
{
    if (g.ks != 3 || g.stride != 1 || (g.W != 8 && g.W != 16) || (g.Cin & 3) || g.Cin > SM_MAXC || (g.Cout & 15) || (g.w_off & 3)) return -2;
    if (g.H % (64 / g.W) || g.H < 2) return -2;
    if ((long long)max(g.Cin, g.Cout) * g.H * g.W >= (1LL << 30)) return -2;
    SmArgs A{};
    A.xin = in; A.g = g; A.w = w; A.wstride = wstride; A.out = out;
    return sm_launch(0, A, g.Cin, n_samples, st);
}

// Backward-data with the fold of the input tensor in the epilogue (fuse.ga required, as launch_conv_bwd_data_rp).
int launch_conv_bwd_data_small(const GView& gy, const ConvGeom& g, const float* w, long long wstride, int n_samples, hipStream_t st, const FoldFuse& fuse)
{
    if (g.ks != 3 || g.stride != 1 || (g.W != 8 && g.W != 16) || (g.Cin & 3) || (g.Cout & 3) || g.Cout > SM_MAXC || (g.w_off & 3) || !fuse.ga) return -2;
    if (g.H % (64 / g.W) || g.H < 4) return -2;      // rows 1 and H-2 must be distinct interior rows
    if ((long long)max(g.Cin, g.Cout) * g.H * g.W >= (1LL << 30)) return -2;
    SmArgs A{};
    A.xin = fuse.x; A.gin = gy; A.g = g; A.w = w; A.wstride = wstride;
    A.fga = fuse.ga; A.fga_sstride = fuse.ga_sstride; A.fbsums = fuse.bsums;
    return sm_launch(1, A, g.Cout, n_samples, st);
}
