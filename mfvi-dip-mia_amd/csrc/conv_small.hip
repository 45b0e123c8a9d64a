// conv_small.hip — 3x3 stride-1 forward on the small maps at the bottom of the hour-glass (8x8, 16x16): one stage, everything in flight.
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
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

struct SmArgs {
    TView xin; ConvGeom g; const float* w; long long wstride; OutDesc out;
    int nx, ny, nz;       // tiles, output fragments, samples
    int tr, wsh;          // image rows per tile (tr * W == 64), log2(W)
};

constexpr int SM_MAXC = 144;      // reduction channels (LDS: 9 groups of window + weights = 143 KB at 16x16)
constexpr int SM_NXJ = 36;        // window items per thread: one window pixel x every 4th of 9 x 16 channels
constexpr int SM_NWJ = 2;         // weight items (9 taps x 4 channels of one (group, m, residue)) per thread: 9 x 64 / 512

__global__ __launch_bounds__(512) void conv_sm_fwd_kernel(SmArgs A)
{
    extern __shared__ __align__(16) float s_dyn[];      // window | weights
    __shared__ ChanFwd s_ch[SM_MAXC];
    __shared__ float s_bias[16];
    __shared__ __align__(16) float s_comb[4][64][4];
    __shared__ double s_red[4][16][2];

    const ConvGeom& g = A.g;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    int bx, by, k;
    xcd_decode(blockIdx.x, A.nx, A.ny, A.nz, bx, by, k);
    const int m0 = by * 16, W = g.W, HW = g.H * W, WW = W + 2, WP = (A.tr + 2) * WW;
    const int Cin = g.Cin, NG = (Cin + 15) >> 4;
    const int py0 = bx * A.tr;
    float* __restrict__ s_x = s_dyn;
    float* __restrict__ s_w = s_dyn + NG * WP * 16;
    const float* __restrict__ wk = A.w + (long long)k * A.wstride;

    // ---- every global load of the block, issued before anything waits.  No division in the per-item index math (the first version
    //      spent ~15 us per block on i / WP, q / rowq, fl / 9 of ~90 items per thread): a thread owns ONE window pixel (128 lanes per
    //      channel row, WP <= 128 of them in use) and walks the channels cb, cb + 4, ...: group = j >> 2, k-step = j & 3 are compile-time ----
    const float* __restrict__ xk = A.xin.data + (long long)k * A.xin.sstride;
    // (lane order: 4 consecutive lanes = the 4 channel residues of one window pixel, so a wave's ds_write_b128 covers 1 KB of contiguous LDS;
    //  with the pixel as the fast lane index the 64-byte pixel stride put 16 lanes on the same banks)
    const int wp = tid >> 2, cb = tid & 3;
    const bool wp_ok = wp < WP;
    int xoff = 0;
    { const int wr = wp / WW, wc = wp - wr * WW; xoff = reflect_idx(py0 - 1 + wr, g.H) * W + reflect_idx(wc - 1, W); }
    float xr[SM_NXJ];
#pragma unroll
    for (int j = 0; j < SM_NXJ; ++j) {
        const int c = cb + 4 * j;
        xr[j] = (wp_ok && c < Cin) ? xk[(long long)c * HW + xoff] : 0.f;
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
            const float* __restrict__ src = wk + g.w_off + ((long long)(m0 + mw) * Cin + c) * 9;
            const bool ok = it < NWI && 16 * gw + lw + 4 * i < Cin;
            const f4u a = ok ? *reinterpret_cast<const f4u*>(src) : f4u{0.f, 0.f, 0.f, 0.f};
            const f4u b = ok ? *reinterpret_cast<const f4u*>(src + 4) : f4u{0.f, 0.f, 0.f, 0.f};
            const float c8 = ok ? src[8] : 0.f;
            wt[j][i][0] = a.x; wt[j][i][1] = a.y; wt[j][i][2] = a.z; wt[j][i][3] = a.w;
            wt[j][i][4] = b.x; wt[j][i][5] = b.y; wt[j][i][6] = b.z; wt[j][i][7] = b.w; wt[j][i][8] = c8;
        }
    }
    // per-channel constants of the deferred BN, bias (channels that pad the last group: zero window entries AND zero weights, written above)
    for (int c = tid; c < Cin; c += 512) s_ch[c] = chan_fwd(A.xin, k, c);
    if (tid < 16) s_bias[tid] = g.b_off >= 0 ? wk[g.b_off + m0 + tid] : 0.f;
    __syncthreads();
    const int act = A.xin.act; const float slope = A.xin.slope;
    if (wp_ok) {
#pragma unroll
        for (int G = 0; G < SM_NXJ / 4; ++G) {
            if (G < NG) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int c = cb + 16 * G + 4 * i; v[i] = c < Cin ? apply_fwd(s_ch[min(c, Cin - 1)], xr[4 * G + i], act, slope) : 0.f; }
                *reinterpret_cast<float4*>(&s_x[((G * WP + wp) << 4) + (cb << 2)]) = make_float4(v[0], v[1], v[2], v[3]);
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
                *reinterpret_cast<float4*>(d + ((tap * NG) << 8)) = make_float4(wt[j][0][tap], wt[j][1][tap], wt[j][2][tap], wt[j][3][tap]);
        }
    }
    __syncthreads();

    // ---- matrix phase: wave = (pixel fragment f, half h of the channel groups) ----
    const int f = wv & 3, h = wv >> 2;
    const int p = 16 * f + l15, trow = p >> A.wsh, tcol = p & (W - 1);
    const int wb = trow * WW + tcol;
    const int gsplit = (NG + 1) >> 1, g0 = h ? gsplit : 0, g1 = h ? NG : gsplit;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int gi = g0; gi < g1; ++gi) {
        const float* __restrict__ xb = s_x + ((gi * WP + wb) << 4) + (l4 << 2);
        const float* __restrict__ wb_ = s_w + (((gi << 4) + l15) << 4) + (l4 << 2);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(wb_ + (((ky * 3 + kx) * NG) << 8));
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(xb + ((ky * WW + kx) << 4));
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i], b4[i], acc, 0, 0, 0);
            }
    }
    if (h) *reinterpret_cast<f32x4*>(&s_comb[f][lane][0]) = acc;
    __syncthreads();
    const bool do_stats = A.out.stats != nullptr;
    if (!h) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(&s_comb[f][lane][0]);
        float* __restrict__ yout = A.out.data + (long long)k * A.out.sstride + (long long)m0 * HW + py0 * W + p;      // (a tile is TR whole rows: pixel p of the tile is contiguous)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = 4 * l4 + q;
            const float v = acc[q] + o[q] + s_bias[ch];
            yout[ch * HW] = v;
            if (do_stats) {
                float a_ = v, b_ = v * v;
#pragma unroll
                for (int s = 8; s > 0; s >>= 1) { a_ += __shfl_xor(a_, s, 64); b_ += __shfl_xor(b_, s, 64); }
                if (l15 == 0) { s_red[f][ch][0] = (double)a_; s_red[f][ch][1] = (double)b_; }
            }
        }
    }
    if (do_stats) {
        __syncthreads();
        if (tid < 32) {
            const int ch = tid >> 1, which = tid & 1;
            atomicAdd(A.out.stats + ((long long)k * g.Cout + m0 + ch) * 2 + which, s_red[0][ch][which] + s_red[1][ch][which] + s_red[2][ch][which] + s_red[3][ch][which]);
        }
    }
}

}  // namespace

// Returns -2 when the shape is not served.
int launch_conv_fwd_small(const TView& in, const ConvGeom& g, const float* w, long long wstride, OutDesc out, int n_samples, hipStream_t st)
{
    if (g.ks != 3 || g.stride != 1 || (g.W != 8 && g.W != 16) || (g.Cin & 3) || g.Cin > SM_MAXC || (g.Cout & 15) || (g.w_off & 3)) return -2;
    const int tr = 64 / g.W;
    if (g.H % tr || g.H < 2) return -2;
    if ((long long)max(g.Cin, g.Cout) * g.H * g.W >= (1LL << 30)) return -2;
    const int NG = (g.Cin + 15) >> 4, WP = (tr + 2) * (g.W + 2);
    if (WP > 128 || NG * 4 > SM_NXJ || NG * 64 > SM_NWJ * 512) return -2;
    const size_t lds_bytes = sizeof(float) * ((size_t)NG * WP * 16 + (size_t)9 * NG * 256);
    if (lds_bytes > 150 * 1024) return -2;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sm_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (attr != hipSuccess) return (int)attr;
    SmArgs A{};
    A.xin = in; A.g = g; A.w = w; A.wstride = wstride; A.out = out;
    A.nx = g.H / tr; A.ny = g.Cout / 16; A.nz = n_samples; A.tr = tr; A.wsh = g.W == 8 ? 3 : 4;
    mfvi_tl_family = 4;
    mfvi_launch(conv_sm_fwd_kernel, dim3(A.nx * A.ny * A.nz), dim3(512), lds_bytes, st, A);
    return (int)hipGetLastError();
}
