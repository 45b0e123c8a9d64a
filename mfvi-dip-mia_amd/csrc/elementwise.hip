// K3/K4 — the non-convolution nodes of the skip() hour-glass, fused:
//   finalize_dx     : reflection-pad adjoint fold (models/common.py:118-121) + sum over the consumers of a tensor
//                     + LeakyReLU' (models/common.py:83) + BN-backward channel sums (models/common.py:96-97)
//   concat_up fwd   : Concat (models/common.py:23-43) of view(A) with bilinear x2 Upsample (models/skip.py:102)
//                     of view(B), written straight into the concat buffer + statistics of the BN that follows
//                     (models/skip.py:68)
//   concat_up bwd   : BN-backward of that BN on load, channel split, bilinear adjoint, LeakyReLU' of A and B,
//                     BN-backward sums of A and B
//   bn_param_grads  : d gamma / d beta from the accumulated sums
#include "common.h"
#include <cstdlib>

namespace {

constexpr int EW_ITEMS = 4;     // pixels per thread

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void finalize_dx_kernel(FoldSrcs S, TView x,
                                                          float* __restrict__ ga, long long ga_sstride,
                                                          double* __restrict__ bsums)
{
    const int n_src = S.n;
    __shared__ ChanFwd s_ch;
    __shared__ double s_red[8];
    const int t = threadIdx.x, k = blockIdx.z, c = blockIdx.y;
    const int H = x.H, W = x.W;
    const long long HW = (long long)H * W;
    if (t == 0) s_ch = chan_fwd(x, k, c);
    __syncthreads();
    const ChanFwd ch = s_ch;
    const bool has_bn = x.stats != nullptr;
    const float* __restrict__ yx = x.data + (long long)k * x.sstride + (long long)c * HW;
    float* __restrict__ gout = ga + (long long)k * ga_sstride + (long long)c * HW;
    double sg = 0.0, sgx = 0.0;
    for (int it = 0; it < EW_ITEMS; ++it) {
        const long long pix = ((long long)blockIdx.x * EW_ITEMS + it) * 256 + t;
        if (pix >= HW) break;
        const int r = (int)(pix / W), q = (int)(pix - (long long)r * W);
        float d = 0.f;
        // view(x) at this pixel: the factor of the sources that carry the gradient wrt view(x)**2
        const float vx = apply_fwd(ch, yx[pix], x.act & 1, x.slope);
        for (int s = 0; s < n_src; ++s) {
            const FoldSrc src = S.s[s];
            const int p = src.pad, Hp = H + 2 * p, Wp = W + 2 * p;
            const float* __restrict__ base = src.d + (long long)k * src.sstride + (long long)c * Hp * Wp;
            const float mul = src.mul2v ? 2.f * vx : 1.f;
            if (p == 0) { d += mul * base[(long long)r * Wp + q]; continue; }
            // padded rows that fold onto r under ReflectionPad2d(p): r+p always; p-r for 1 <= r <= p (top mirror);
            // p + 2(H-1) - r for H-1-p <= r <= H-2 (bottom mirror).  Columns alike.
            int rows[3], cols[3], nr = 0, nc = 0;
            rows[nr++] = r + p; if (r >= 1 && r <= p) rows[nr++] = p - r; if (r >= H - 1 - p && r <= H - 2) rows[nr++] = p + 2 * (H - 1) - r;
            cols[nc++] = q + p; if (q >= 1 && q <= p) cols[nc++] = p - q; if (q >= W - 1 - p && q <= W - 2) cols[nc++] = p + 2 * (W - 1) - q;
            float e = 0.f;
            for (int a = 0; a < nr; ++a)
                for (int b = 0; b < nc; ++b) e += base[(long long)rows[a] * Wp + cols[b]];
            d += mul * e;
        }
        if (has_bn) {
            const float yv = yx[pix];
            const float v = __builtin_fmaf(yv - ch.mean, ch.scale, ch.beta);
            if ((x.act & 1) && !(v > 0.f)) d *= x.slope;
            sg += (double)d; sgx += (double)d * (double)((yv - ch.mean) * ch.rstd);
        }
        gout[pix] = d;
    }
    if (has_bn) {
        const double a = block_sum_d(sg, s_red);
        const double b = block_sum_d(sgx, s_red);
        if (t == 0) {
            double* o = bsums + ((long long)k * x.C + c) * 2;
            atomicAdd(o, a); atomicAdd(o + 1, b);
        }
    }
}


typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));      // 4 floats at dword alignment (padded rows)
typedef float f2a __attribute__((ext_vector_type(2)));

constexpr int V_GROUPS = 4;     // float4 groups per thread in the vector kernels (4096 pixels per block)

// One row `prow` of a padded gradient plane folded horizontally onto pixels q..q+3 of the unpadded row (q % 4 == 0).
__device__ __forceinline__ float4 fold_row4(const float* __restrict__ base, int prow, int q, int W, int Wp, int pad)
{
    const float* __restrict__ p = base + (long long)prow * Wp + q + pad;
    const f4u v = *reinterpret_cast<const f4u*>(p);
    float4 r = make_float4(v.x, v.y, v.z, v.w);
    if (pad) {
        if (q == 0) r.y += base[(long long)prow * Wp];                     // pixel 1 <- padded column 0
        if (q + 4 == W) r.z += base[(long long)prow * Wp + W + 1];         // pixel W-2 <- padded column W+1
    }
    return r;
}

// finalize_dx for W % 4 == 0: float4 per lane, 4 groups per thread, loads issued before use.
__global__ __launch_bounds__(256) void finalize_dx_vec_kernel(FoldSrc s0, FoldSrc s1, int n_src, TView x,
                                                              float* __restrict__ ga, long long ga_sstride,
                                                              double* __restrict__ bsums)
{   // (the float4 kernel serves the RT layers: at most two plain sources; LRT sources take the scalar kernel)
    __shared__ ChanFwd s_ch;
    __shared__ double s_red[8];
    const int t = threadIdx.x, k = blockIdx.z, c = blockIdx.y;
    const int H = x.H, W = x.W, W4 = W >> 2;
    const long long HW = (long long)H * W;
    const int n_groups = H * W4;
    if (t == 0) s_ch = chan_fwd(x, k, c);
    const bool has_bn = x.stats != nullptr;
    const float* __restrict__ yx = x.data + (long long)k * x.sstride + (long long)c * HW;
    float* __restrict__ gout = ga + (long long)k * ga_sstride + (long long)c * HW;
    float4 d[V_GROUPS], yv[V_GROUPS];
#pragma unroll
    for (int it = 0; it < V_GROUPS; ++it) {
        const int gi = (blockIdx.x * V_GROUPS + it) * 256 + t;
        d[it] = make_float4(0.f, 0.f, 0.f, 0.f); yv[it] = d[it];
        if (gi >= n_groups) continue;
        const int r = gi / W4, q = (gi - r * W4) * 4;
        for (int s = 0; s < n_src; ++s) {
            const FoldSrc src = s == 0 ? s0 : s1;
            const int p = src.pad, Hp = H + 2 * p, Wp = W + 2 * p;
            const float* __restrict__ base = src.d + (long long)k * src.sstride + (long long)c * Hp * Wp;
            float4 v = fold_row4(base, r + p, q, W, Wp, p);
            if (p) {
                if (r == 1) { const float4 u = fold_row4(base, 0, q, W, Wp, p); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
                if (r == H - 2) { const float4 u = fold_row4(base, H + 1, q, W, Wp, p); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
            }
            d[it].x += v.x; d[it].y += v.y; d[it].z += v.z; d[it].w += v.w;
        }
        if (has_bn) yv[it] = *reinterpret_cast<const float4*>(yx + (long long)gi * 4);
    }
    __syncthreads();
    const ChanFwd ch = s_ch;
    double sg = 0.0, sgx = 0.0;
#pragma unroll
    for (int it = 0; it < V_GROUPS; ++it) {
        const int gi = (blockIdx.x * V_GROUPS + it) * 256 + t;
        if (gi >= n_groups) continue;
        float dd[4] = {d[it].x, d[it].y, d[it].z, d[it].w};
        if (has_bn) {
            const float yy[4] = {yv[it].x, yv[it].y, yv[it].z, yv[it].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = __builtin_fmaf(yy[j] - ch.mean, ch.scale, ch.beta);
                if ((x.act & 1) && !(v > 0.f)) dd[j] *= x.slope;
                sg += (double)dd[j]; sgx += (double)dd[j] * (double)((yy[j] - ch.mean) * ch.rstd);
            }
        }
        *reinterpret_cast<float4*>(gout + (long long)gi * 4) = make_float4(dd[0], dd[1], dd[2], dd[3]);
    }
    if (has_bn) {
        const double a = block_sum_d(sg, s_red);
        const double b = block_sum_d(sgx, s_red);
        if (t == 0) {
            double* o = bsums + ((long long)k * x.C + c) * 2;
            atomicAdd(o, a); atomicAdd(o + 1, b);
        }
    }
}

// finalize_dx with ONE padded source and the backward-data of a narrow 1x1 consumer formed in place (round 4): a tensor of
// the down path feeds the stride-2 3x3 convolution of its scale and the 4-channel 1x1 skip convolution (models/skip.py:60-66).  The skip
// branch's gradient wrt x is dx[c][p] = sum_j W_k[j][c] * dy_j[p] over CS <= 8 output channels — 4 multiply-adds per element — and used to be
// a launch of its own (8-12 us, a padded-gradient scratch written and read back).  Here dy_j is formed on load (BN-backward of the skip
// branch's BatchNorm) from the CS small planes every channel block of the grid re-reads through L2.  Two float4 groups per thread.
constexpr int V1_GROUPS = 2;
template <int V1_MAXC>      // 4 or 8: channels of the 1x1 consumer held in registers (cs1 <= V1_MAXC)
__global__ __launch_bounds__(256) void finalize_dx_vec1_kernel(FoldSrc s0, GView g1, const float* __restrict__ w1, long long w1_sstride, int cs1, TView x,
                                                               float* __restrict__ ga, long long ga_sstride, double* __restrict__ bsums)
{
    __shared__ ChanFwd s_ch;
    __shared__ ChanBwd s_cb[V1_MAXC];
    __shared__ float s_w1[V1_MAXC];
    __shared__ double s_red[8];
    const int t = threadIdx.x, k = blockIdx.z, c = blockIdx.y;
    const int H = x.H, W = x.W, W4 = W >> 2;
    const long long HW = (long long)H * W;
    const int n_groups = H * W4;
    if (t == 0) s_ch = chan_fwd(x, k, c);
    if (t >= 64 && t < 64 + cs1) { s_cb[t - 64] = chan_bwd(g1, k, t - 64); s_w1[t - 64] = w1[(long long)k * w1_sstride + (long long)(t - 64) * x.C + c]; }
    const bool has_bn = x.stats != nullptr, bn1 = g1.stats != nullptr;
    const float* __restrict__ yx = x.data + (long long)k * x.sstride + (long long)c * HW;
    float* __restrict__ gout = ga + (long long)k * ga_sstride + (long long)c * HW;
    const float* __restrict__ ga1 = g1.ga + (long long)k * g1.gstride;
    const float* __restrict__ y1 = bn1 ? g1.y + (long long)k * g1.ystride : ga1;      // (no BatchNorm behind the skip convolution: the loads still run, against qc = 0)
    const int p = s0.pad, Hp = H + 2 * p, Wp = W + 2 * p;
    const float* __restrict__ base = s0.d + (long long)k * s0.sstride + (long long)c * Hp * Wp;
    float4 d[V1_GROUPS], yv[V1_GROUPS], g4[V1_GROUPS][V1_MAXC], y4[V1_GROUPS][V1_MAXC];
#pragma unroll
    for (int it = 0; it < V1_GROUPS; ++it) {
        const int gi = min((int)(blockIdx.x * V1_GROUPS + it) * 256 + t, n_groups - 1);      // (clamped: every load unconditional; lanes past the end do not store)
        const int r = gi / W4, q = (gi - r * W4) * 4;
        float4 v = fold_row4(base, r + p, q, W, Wp, p);
        if (p) {
            if (r == 1) { const float4 u = fold_row4(base, 0, q, W, Wp, p); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
            if (r == H - 2) { const float4 u = fold_row4(base, H + 1, q, W, Wp, p); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
        }
        d[it] = v;
        yv[it] = *reinterpret_cast<const float4*>(yx + (long long)gi * 4);
#pragma unroll
        for (int j = 0; j < V1_MAXC; ++j) {
            const int jj = min(j, cs1 - 1);
            g4[it][j] = *reinterpret_cast<const float4*>(ga1 + (long long)jj * HW + (long long)gi * 4);
            y4[it][j] = *reinterpret_cast<const float4*>(y1 + (long long)jj * HW + (long long)gi * 4);
        }
    }
    __syncthreads();
    const ChanFwd ch = s_ch;
    double sg = 0.0, sgx = 0.0;
#pragma unroll
    for (int it = 0; it < V1_GROUPS; ++it) {
        const int gi = (blockIdx.x * V1_GROUPS + it) * 256 + t;
        float dd[4] = {d[it].x, d[it].y, d[it].z, d[it].w};
#pragma unroll
        for (int j = 0; j < V1_MAXC; ++j) {
            if (j < cs1) {
                const ChanBwd cb = s_cb[j]; const float wj = s_w1[j];
                dd[0] = __builtin_fmaf(wj, apply_bwd(cb, g4[it][j].x, y4[it][j].x), dd[0]); dd[1] = __builtin_fmaf(wj, apply_bwd(cb, g4[it][j].y, y4[it][j].y), dd[1]);
                dd[2] = __builtin_fmaf(wj, apply_bwd(cb, g4[it][j].z, y4[it][j].z), dd[2]); dd[3] = __builtin_fmaf(wj, apply_bwd(cb, g4[it][j].w, y4[it][j].w), dd[3]);
            }
        }
        if (gi >= n_groups) continue;
        if (has_bn) {
            const float yy[4] = {yv[it].x, yv[it].y, yv[it].z, yv[it].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = __builtin_fmaf(yy[j] - ch.mean, ch.scale, ch.beta);
                if ((x.act & 1) && !(v > 0.f)) dd[j] *= x.slope;
                sg += (double)dd[j]; sgx += (double)dd[j] * (double)((yy[j] - ch.mean) * ch.rstd);
            }
        }
        *reinterpret_cast<float4*>(gout + (long long)gi * 4) = make_float4(dd[0], dd[1], dd[2], dd[3]);
    }
    if (has_bn) {
        const double a = block_sum_d(sg, s_red);
        const double b = block_sum_d(sgx, s_red);
        if (t == 0) {
            double* o = bsums + ((long long)k * x.C + c) * 2;
            atomicAdd(o, a); atomicAdd(o + 1, b);
        }
    }
}

// bilinear x2, align_corners=False: src = (dst+0.5)/2-0.5 clamped at 0, i1 = min(i0+1, n-1)
__device__ __forceinline__ void up_coef(int d, int n, int& i0, int& i1, float& l1)
{
    float s = ((float)d + 0.5f) * 0.5f - 0.5f; s = s < 0.f ? 0.f : s;
    i0 = (int)s; i1 = min(i0 + 1, n - 1); l1 = s - (float)i0;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void concat_up_fwd_kernel(TView a, int has_a, TView b, OutDesc out, int H, int W, int nearest)
{
    __shared__ ChanFwd s_ch;
    __shared__ double s_red[8];
    const int t = threadIdx.x, k = blockIdx.z, c = blockIdx.y;
    const int Ca = has_a ? a.C : 0, Ct = Ca + b.C;
    const long long HW = (long long)H * W;
    const bool from_a = c < Ca;
    if (t == 0) s_ch = from_a ? chan_fwd(a, k, c) : chan_fwd(b, k, c - Ca);
    __syncthreads();
    const ChanFwd ch = s_ch;
    float* __restrict__ o = out.data + (long long)k * out.sstride + (long long)c * HW;
    double sum = 0.0, sq = 0.0;
    for (int it = 0; it < EW_ITEMS; ++it) {
        const long long pix = ((long long)blockIdx.x * EW_ITEMS + it) * 256 + t;
        if (pix >= HW) break;
        float v;
        if (from_a) {
            v = apply_fwd(ch, a.data[(long long)k * a.sstride + (long long)c * HW + pix], a.act, a.slope);
        } else {
            const int r = (int)(pix / W), q = (int)(pix - (long long)r * W);
            int y0, y1, x0, x1; float ly, lx;
            up_coef(r, b.H, y0, y1, ly); up_coef(q, b.W, x0, x1, lx);
            const float* __restrict__ p = b.data + (long long)k * b.sstride + (long long)(c - Ca) * b.H * b.W;
            if (nearest) { y0 = y1 = r >> 1; x0 = x1 = q >> 1; ly = 0.f; lx = 0.f; }      // mode='nearest': src = floor(dst / 2)
            const float v00 = apply_fwd(ch, p[y0 * b.W + x0], b.act, b.slope), v01 = apply_fwd(ch, p[y0 * b.W + x1], b.act, b.slope);
            const float v10 = apply_fwd(ch, p[y1 * b.W + x0], b.act, b.slope), v11 = apply_fwd(ch, p[y1 * b.W + x1], b.act, b.slope);
            v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        }
        o[pix] = v; sum += (double)v; sq += (double)v * (double)v;
    }
    if (out.stats != nullptr) {
        const double sa = block_sum_d(sum, s_red);
        const double sb = block_sum_d(sq, s_red);
        if (t == 0) { double* st = out.stats + ((long long)k * Ct + c) * 2; atomicAdd(st, sa); atomicAdd(st + 1, sb); }
    }
}


// concat_up forward for W % 4 == 0: every lane writes 4 consecutive pixels of one row.  The upsampled part reads the four
// low-res columns q/2-1 .. q/2+2 (clamped) of two low-res rows and blends with the same arithmetic as the scalar kernel.
__global__ __launch_bounds__(256) void concat_up_fwd_vec_kernel(TView a, int has_a, TView b, OutDesc out, int H, int W, int nearest)
{
    __shared__ ChanFwd s_ch;
    __shared__ double s_red[8];
    const int t = threadIdx.x, k = blockIdx.z, c = blockIdx.y;
    const int Ca = has_a ? a.C : 0, Ct = Ca + b.C;
    const long long HW = (long long)H * W;
    const int W4 = W >> 2, n_groups = H * W4;
    const bool from_a = c < Ca;
    if (t == 0) s_ch = from_a ? chan_fwd(a, k, c) : chan_fwd(b, k, c - Ca);
    __syncthreads();
    const ChanFwd ch = s_ch;
    float* __restrict__ o = out.data + (long long)k * out.sstride + (long long)c * HW;
    double sum = 0.0, sq = 0.0;
#pragma unroll
    for (int it = 0; it < V_GROUPS; ++it) {
        const int gi = (blockIdx.x * V_GROUPS + it) * 256 + t;
        if (gi >= n_groups) continue;
        float v[4];
        if (from_a) {
            const float4 y = *reinterpret_cast<const float4*>(a.data + (long long)k * a.sstride + (long long)c * HW + (long long)gi * 4);
            v[0] = apply_fwd(ch, y.x, a.act, a.slope); v[1] = apply_fwd(ch, y.y, a.act, a.slope);
            v[2] = apply_fwd(ch, y.z, a.act, a.slope); v[3] = apply_fwd(ch, y.w, a.act, a.slope);
        } else {
            const int r = gi / W4, q = (gi - r * W4) * 4;
            int y0, y1; float ly;
            up_coef(r, b.H, y0, y1, ly);
            if (nearest) { y0 = y1 = r >> 1; ly = 0.f; }
            const float* __restrict__ p = b.data + (long long)k * b.sstride + (long long)(c - Ca) * b.H * b.W;
            const int xb = (q >> 1) - 1;
            float lo[4], hi[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xc = min(max(xb + j, 0), b.W - 1);
                lo[j] = apply_fwd(ch, p[y0 * b.W + xc], b.act, b.slope);
                hi[j] = apply_fwd(ch, p[y1 * b.W + xc], b.act, b.slope);
            }
            // output column q+j blends low-res columns (x0, x1) with weight lx: interior pattern (0,1,.75) (1,2,.25) (1,2,.75) (2,3,.25);
            // at q == 0 the clamped source of column 0 is (1,2) with lx = 0 (up_coef) so the value is exact
            const bool left = q == 0;
            const int i0[4] = {0, 1, 1, 2};
            const float lxs[4] = {left ? 0.f : 0.75f, 0.25f, 0.75f, 0.25f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // mode='nearest': output columns q..q+3 copy low-res columns q/2, q/2, q/2+1, q/2+1 (lo[1], lo[1], lo[2], lo[2])
                const int ia = nearest ? 1 + (j >> 1) : ((j == 0 && left) ? 1 : i0[j]);
                const float lx = nearest ? 0.f : lxs[j];
                const float v00 = lo[ia], v01 = lo[ia + 1], v10 = hi[ia], v11 = hi[ia + 1];
                v[j] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
            }
        }
        *reinterpret_cast<float4*>(o + (long long)gi * 4) = make_float4(v[0], v[1], v[2], v[3]);
        // BN statistics: float partials over the 4 pixels of the group, folded into the thread's fp64 sums (the per-element fp64 converts /
        // adds ran at half rate and made this kernel VALU-bound)
        { const float s4 = (v[0] + v[1]) + (v[2] + v[3]);
          const float q4 = __builtin_fmaf(v[0], v[0], v[1] * v[1]) + __builtin_fmaf(v[2], v[2], v[3] * v[3]);
          sum += (double)s4; sq += (double)q4; }
    }
    if (out.stats != nullptr) {
        const double sa = block_sum_d(sum, s_red);
        const double sb = block_sum_d(sq, s_red);
        if (t == 0) { double* st = out.stats + ((long long)k * Ct + c) * 2; atomicAdd(st, sa); atomicAdd(st + 1, sb); }
    }
}

// concat_up forward for maps at least 128 wide (round 4): LDS-tiled.  A block = (sample, concat channel, tile of CF_TH x CF_TW LOW-res
// pixels = 2 CF_TH x 2 CF_TW outputs).  The vector kernel above fetches eight scalars per float4 of output (two low-res rows x four clamped
// columns per lane, every low-res value loaded by four lanes): 2.2 load instructions per output row segment, 3 TB/s at 256^2.  Here the
// low-res window (CF_TH + 2 rows x CF_TW + 2 columns, clamped at the image border exactly as up_coef / the vector kernel clamp) is loaded ONCE,
// coalesced, transformed (deferred BN + LeakyReLU of the low-res tensor) and kept in LDS; every lane then reads its 2 x 4 taps as aligned
// float2 pairs and blends with the SAME arithmetic (results bit-identical to the vector kernel: tests/test_gpu_parity.py).  Channels of the
// skip branch are an element-wise copy with the view applied, as before.
constexpr int CF_TH = 8, CF_TW = 64, CF_PITCH = CF_TW + 4;      // even pitch: the float2 reads of consecutive lanes are consecutive
__global__ __launch_bounds__(256) void concat_up_fwd_tiled_kernel(TView a, int has_a, TView b, OutDesc out, int H, int W, int nearest, int tiles_x, int n_tiles)
{
    __shared__ ChanFwd s_ch;
    __shared__ double s_red[8];
    __shared__ __align__(8) float s_lo[CF_TH + 2][CF_PITCH];
    const int t = threadIdx.x, k = blockIdx.z, c = blockIdx.y;
    const int Ca = has_a ? a.C : 0, Ct = Ca + b.C;
    const long long HW = (long long)H * W;
    const bool from_a = c < Ca;
    if (t == 0) s_ch = from_a ? chan_fwd(a, k, c) : chan_fwd(b, k, c - Ca);
    float* __restrict__ o = out.data + (long long)k * out.sstride + (long long)c * HW;
    double sum = 0.0, sq = 0.0;
    constexpr int NG = 2 * CF_TH * (2 * CF_TW / 4) / 256;      // float4 groups per thread
    __syncthreads();
    const ChanFwd ch = s_ch;
    // tiles blockIdx.x, blockIdx.x + gridDim.x, ... : several tiles per block amortise the channel constants and the block reduction
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int m0 = (tile / tiles_x) * CF_TH, n0 = (tile % tiles_x) * CF_TW;      // low-res tile origin
    const int r0 = 2 * m0, q0 = 2 * n0;
    if (from_a) {
        const float* __restrict__ src = a.data + (long long)k * a.sstride + (long long)c * HW;
        float4 y[NG]; bool ok[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int idx = t + 256 * j, r = r0 + idx / (2 * CF_TW / 4), q = q0 + 4 * (idx % (2 * CF_TW / 4));
            ok[j] = r < H && q < W;
            y[j] = ok[j] ? *reinterpret_cast<const float4*>(src + (long long)r * W + q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            if (!ok[j]) continue;
            const int idx = t + 256 * j, r = r0 + idx / (2 * CF_TW / 4), q = q0 + 4 * (idx % (2 * CF_TW / 4));
            float v[4] = {apply_fwd(ch, y[j].x, a.act, a.slope), apply_fwd(ch, y[j].y, a.act, a.slope), apply_fwd(ch, y[j].z, a.act, a.slope), apply_fwd(ch, y[j].w, a.act, a.slope)};
            *reinterpret_cast<float4*>(o + (long long)r * W + q) = make_float4(v[0], v[1], v[2], v[3]);
            const float s4 = (v[0] + v[1]) + (v[2] + v[3]);
            const float q4 = __builtin_fmaf(v[0], v[0], v[1] * v[1]) + __builtin_fmaf(v[2], v[2], v[3] * v[3]);
            sum += (double)s4; sq += (double)q4;
        }
    } else {
        const float* __restrict__ p = b.data + (long long)k * b.sstride + (long long)(c - Ca) * b.H * b.W;
        constexpr int NL = ((CF_TH + 2) * (CF_TW + 2) + 255) / 256;
        float raw[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int idx = min(t + 256 * j, (CF_TH + 2) * (CF_TW + 2) - 1), lr = idx / (CF_TW + 2), lc = idx - lr * (CF_TW + 2);
            const int yy = min(max(m0 - 1 + lr, 0), b.H - 1), xx = min(max(n0 - 1 + lc, 0), b.W - 1);
            raw[j] = p[yy * b.W + xx];
        }
        __syncthreads();                                   // the previous tile's reads of s_lo are done
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int idx = t + 256 * j, lr = idx / (CF_TW + 2), lc = idx - lr * (CF_TW + 2);
            if (idx < (CF_TH + 2) * (CF_TW + 2)) s_lo[lr][lc] = apply_fwd(ch, raw[j], b.act, b.slope);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int idx = t + 256 * j, col4 = idx % (2 * CF_TW / 4), r = r0 + idx / (2 * CF_TW / 4), q = q0 + 4 * col4;
            if (r >= H || q >= W) continue;
            int y0, y1; float ly;
            up_coef(r, b.H, y0, y1, ly);
            if (nearest) { y0 = y1 = r >> 1; ly = 0.f; }
            // low-res columns q/2 - 1 .. q/2 + 2 (clamped when staged) = local columns 2 col4 .. 2 col4 + 3
            const float2 l0 = *reinterpret_cast<const float2*>(&s_lo[y0 - m0 + 1][2 * col4]), l1 = *reinterpret_cast<const float2*>(&s_lo[y0 - m0 + 1][2 * col4 + 2]);
            const float2 h0 = *reinterpret_cast<const float2*>(&s_lo[y1 - m0 + 1][2 * col4]), h1 = *reinterpret_cast<const float2*>(&s_lo[y1 - m0 + 1][2 * col4 + 2]);
            const float lo[4] = {l0.x, l0.y, l1.x, l1.y}, hi[4] = {h0.x, h0.y, h1.x, h1.y};
            const bool left = q == 0;
            const int i0[4] = {0, 1, 1, 2};
            const float lxs[4] = {left ? 0.f : 0.75f, 0.25f, 0.75f, 0.25f};
            float v[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int ia = nearest ? 1 + (jj >> 1) : ((jj == 0 && left) ? 1 : i0[jj]);
                const float lx = nearest ? 0.f : lxs[jj];
                const float v00 = lo[ia], v01 = lo[ia + 1], v10 = hi[ia], v11 = hi[ia + 1];
                v[jj] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
            }
            *reinterpret_cast<float4*>(o + (long long)r * W + q) = make_float4(v[0], v[1], v[2], v[3]);
            const float s4 = (v[0] + v[1]) + (v[2] + v[3]);
            const float q4 = __builtin_fmaf(v[0], v[0], v[1] * v[1]) + __builtin_fmaf(v[2], v[2], v[3] * v[3]);
            sum += (double)s4; sq += (double)q4;
        }
    }
    }
    if (out.stats != nullptr) {
        const double sa = block_sum_d(sum, s_red);
        const double sb = block_sum_d(sq, s_red);
        if (t == 0) { double* st = out.stats + ((long long)k * Ct + c) * 2; atomicAdd(st, sa); atomicAdd(st + 1, sb); }
    }
}

// ---------------------------------------------------------------------------------------------
// concat_up backward, LDS-tiled.  A block owns one (sample, concat channel) and a tile of CB_TH x CB_TW LOW-res pixels
// (= 2*CB_TH x 2*CB_TW hi-res pixels of the concat gradient):
//   channels of A  : elementwise over the hi-res tile (BN-backward of the concat BN on load, LeakyReLU'/BN sums of A);
//   channels of B  : the hi-res window rows 2*m0-1 .. 2*m0+2*CB_TH, cols 2*n0-1 .. 2*n0+2*CB_TW of dy is staged ONCE
//                    (coalesced float2 loads), split into even/odd column planes so the stride-2 taps of the bilinear
//                    adjoint read LDS conflict-free; each thread then gathers its 4x4 taps from LDS.
// Tile shapes (round 4): 16 x 64 low-res pixels for maps at least 48 wide; 32 x 32 and 16 x 16 for the 32-, 16- and 8-wide low-res maps of the
// deeper scales, where a 64-wide tile left half to seven eighths of every block's lanes without a pixel (concat_bwd of the 64^2 / 32^2 /
// 16^2 scales: 39 / 24 / 21 us for 73 / 18 / 5 MB).  256 threads, CB_TH * CB_TW / 256 outputs per thread.
template <int CB_TH, int CB_TW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void concat_up_bwd_kernel(GView gc, TView a, int has_a, float* __restrict__ ga_a,
                                                            long long ga_a_sstride, double* __restrict__ bsums_a,
                                                            TView b, float* __restrict__ ga_b, long long ga_b_sstride,
                                                            double* __restrict__ bsums_b, int tiles_x, int nearest, int pairs)
{
    constexpr int CB_ROWS = 2 * CB_TH + 2, CB_PITCH = CB_TW + 1;      // staged rows; entries per even / odd plane row
    constexpr int RSTEP = 256 / CB_TW, NQ = CB_TH / RSTEP;            // a thread's outputs: rows t / CB_TW + RSTEP * q of column t % CB_TW
    static_assert(256 % CB_TW == 0 && CB_TH % RSTEP == 0 && NQ >= 1, "tile geometry");
    __shared__ ChanFwd s_ch;
    __shared__ ChanBwd s_cg;
    __shared__ double s_red[8];
    __shared__ float s_e[CB_ROWS][CB_PITCH], s_o[CB_ROWS][CB_PITCH];
    const int t = threadIdx.x, k = blockIdx.z, c = blockIdx.y;
    const int Ca = has_a ? a.C : 0;
    const bool from_a = c < Ca;
    const int H = gc.H, W = gc.W;
    const long long HW = (long long)H * W;
    // channel constants (fp64 divisions and square roots) by one lane of two different waves, needed only behind the batch of loads below
    if (t == 0) s_ch = from_a ? chan_fwd(a, k, c) : chan_fwd(b, k, c - Ca);
    if (t == 64) s_cg = chan_bwd(gc, k, c);
    const float* __restrict__ gap = gc.ga + (long long)k * gc.gstride + (long long)c * HW;
    const float* __restrict__ ycp = gc.y + (long long)k * gc.ystride + (long long)c * HW;
    const bool cat_bn = gc.stats != nullptr;
    const int m0 = (blockIdx.x / tiles_x) * CB_TH, n0 = (blockIdx.x % tiles_x) * CB_TW;      // low-res tile origin
    double sg = 0.0, sgx = 0.0;
    const TView& dst = from_a ? a : b;
    const int cd = from_a ? c : c - Ca;
    const long long HWd = (long long)dst.H * dst.W;
    const float* __restrict__ yd = dst.data + (long long)k * dst.sstride + (long long)cd * HWd;
    float* __restrict__ go = (from_a ? ga_a + (long long)k * ga_a_sstride : ga_b + (long long)k * ga_b_sstride) + (long long)cd * HWd;
    const bool dst_bn = dst.stats != nullptr;

    // All global loads of a phase are issued before anything depends on them (address math first, then the batch): with one
    // dependent load per loop trip a block paid one memory latency per trip (9 trips of staging + 4 of output) and the kernel ran at
    // latency x trips instead of bandwidth.
    if (from_a) {
        // hi-res tile rows 2*m0 .. 2*m0+2*CB_TH-1, cols 2*n0 .. 2*n0+2*CB_TW-1, as float2 (W is even)
        const int r0 = 2 * m0, q0 = 2 * n0;
        constexpr int NA = 2 * CB_TH * CB_TW / 256;
        f2a g2[NA], y2[NA], yv2[NA]; int pixs[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int i = t + 256 * j, r = r0 + i / CB_TW, q = q0 + (i % CB_TW) * 2;
            const bool ok = r < H && q < W;
            pixs[j] = ok ? r * W + q : -1;
            const int pix = ok ? pixs[j] : 0;
            if (pairs) {
                g2[j] = *reinterpret_cast<const f2a*>(gap + pix);
                if (cat_bn) y2[j] = *reinterpret_cast<const f2a*>(ycp + pix);
                if (dst_bn) yv2[j] = *reinterpret_cast<const f2a*>(yd + pix);
            } else {      // odd width (Concat crop) / unaligned rows: element loads, the second element may lie beyond the row
                const int p1 = (ok && q + 1 < W) ? pix + 1 : pix;
                g2[j].x = gap[pix]; g2[j].y = gap[p1];
                if (cat_bn) { y2[j].x = ycp[pix]; y2[j].y = ycp[p1]; }
                if (dst_bn) { yv2[j].x = yd[pix]; yv2[j].y = yd[p1]; }
                if (ok && q + 1 >= W) pixs[j] = -2 - pix;      // only the first element exists
            }
        }
        __syncthreads();
        const ChanFwd ch = s_ch; const ChanBwd cg = s_cg;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            if (pixs[j] == -1) continue;
            const bool single = pixs[j] < -1;
            const int pix0 = single ? -2 - pixs[j] : pixs[j];
            float d[2] = {g2[j].x, g2[j].y};
            if (cat_bn) { d[0] = apply_bwd(cg, g2[j].x, y2[j].x); d[1] = apply_bwd(cg, g2[j].y, y2[j].y); }
            if (dst_bn) {
                const float yy[2] = {yv2[j].x, yv2[j].y};
#pragma unroll
                for (int l = 0; l < 2; ++l) {
                    const float v = __builtin_fmaf(yy[l] - ch.mean, ch.scale, ch.beta);
                    if (dst.act && !(v > 0.f)) d[l] *= dst.slope;
                }
                if (single) d[1] = 0.f;
                // float partials over the pair, folded into the thread's fp64 sums
                sg += (double)(d[0] + d[1]);
                sgx += (double)__builtin_fmaf(d[0], (yy[0] - ch.mean) * ch.rstd, d[1] * ((yy[1] - ch.mean) * ch.rstd));
            }
            if (pairs) { f2a o2; o2.x = d[0]; o2.y = d[1]; *reinterpret_cast<f2a*>(go + pix0) = o2; }
            else { go[pix0] = d[0]; if (!single) go[pix0 + 1] = d[1]; }
        }
    } else {
        // ---- stage dy of the hi-res window: local row lr <-> hi-res row 2*m0-1+lr, local col lc <-> hi-res col 2*n0-1+lc;
        //      even lc -> s_e[lr][lc/2], odd lc -> s_o[lr][lc/2].  Slot 0: lc 0; slots 1..CB_TW: lc (2s-1, 2s); slot CB_TW+1: lc 2*CB_TW+1.
        const int gr0 = 2 * m0 - 1;
        constexpr int SLOTS = CB_TW + 2, NS = (CB_ROWS * SLOTS + 255) / 256;
        // slot sl of a window row = the aligned pair at hi-res columns (2*n0 - 2 + 2*sl, +1): .x -> s_o[lr][sl-1], .y -> s_e[lr][sl]
        // (slot 0 only contributes its .y, the last slot only its .x); branch-free: clamped address, zeroed when outside the image
        f2a g2[NS], y2[NS]; int ok[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int i = min(t + 256 * j, CB_ROWS * SLOTS - 1), lr = i / SLOTS, sl = i - lr * SLOTS;
            const int gr = gr0 + lr, gq = 2 * n0 - 2 + 2 * sl;
            ok[j] = gr >= 0 && gr < H && gq >= 0 && gq < W;
            const int pix = ok[j] ? gr * W + gq : 0;
            if (pairs) {
                g2[j] = *reinterpret_cast<const f2a*>(gap + pix);
                if (cat_bn) y2[j] = *reinterpret_cast<const f2a*>(ycp + pix);
            } else {      // odd width / unaligned rows: element loads; column gq + 1 may be the (dropped) column W
                const bool two = ok[j] && gq + 1 < W;
                const int p1 = two ? pix + 1 : pix;
                g2[j].x = gap[pix]; g2[j].y = two ? gap[p1] : 0.f;
                if (cat_bn) { y2[j].x = ycp[pix]; y2[j].y = ycp[p1]; }
                if (ok[j] && !two) ok[j] = 2;      // second element outside the image: staged as zero
            }
        }
        // raw destination values of this thread's outputs (LeakyReLU' and x-hat of the BN-backward sums), requested with the batch above
        const int nl = t % CB_TW, n = n0 + nl;
        float ydv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int m = m0 + t / CB_TW + RSTEP * q;
            ydv[q] = (dst_bn && m < dst.H && n < dst.W) ? yd[m * dst.W + n] : 0.f;
        }
        __syncthreads();
        const ChanFwd ch = s_ch; const ChanBwd cg = s_cg;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int i = t + 256 * j, lr = i / SLOTS, sl = i - lr * SLOTS;
            if (i >= CB_ROWS * SLOTS) continue;
            float d0 = g2[j].x, d1 = g2[j].y;
            if (cat_bn) { d0 = apply_bwd(cg, g2[j].x, y2[j].x); d1 = apply_bwd(cg, g2[j].y, y2[j].y); }
            if (!ok[j]) { d0 = 0.f; d1 = 0.f; }
            if (ok[j] == 2) d1 = 0.f;
            if (sl >= 1) s_o[lr][sl - 1] = d0;
            if (sl <= CB_TW) s_e[lr][sl] = d1;
        }
        __syncthreads();
        // ---- adjoint of the bilinear x2 gather (align_corners=False).  Low-res row m feeds hi-res rows 2m-1..2m+2 with weights
        //      .25 .75 .75 .25; at the borders the clamped taps collapse: row 0 gets 1.0 from hi-res row 0 and the last row 1.0
        //      from the last hi-res row (columns alike).
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) {
            const int ml = t / CB_TW + RSTEP * qq;
            const int m = m0 + ml;
            if (m >= dst.H || n >= dst.W) continue;
            float wy[4], wx[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int oy = 2 * m - 1 + q, ox = 2 * n - 1 + q;
                float v = (q == 0 || q == 3) ? 0.25f : 0.75f;
                if (oy < 0 || oy >= H) v = 0.f; else if ((m == 0 && q == 1) || (m == dst.H - 1 && q == 2)) v = 1.f;
                float u = (q == 0 || q == 3) ? 0.25f : 0.75f;
                if (ox < 0 || ox >= W) u = 0.f; else if ((n == 0 && q == 1) || (n == dst.W - 1 && q == 2)) u = 1.f;
                if (nearest) { v = (q == 1 || q == 2) ? 1.f : 0.f; u = v; }      // adjoint of the 2x2 replication: rows 2m, 2m+1 only
                wy[q] = v; wx[q] = u;
            }
            float d = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int lr = 2 * ml + q;
                float rowacc = 0.f;
                rowacc = __builtin_fmaf(s_e[lr][nl], wx[0], rowacc);
                rowacc = __builtin_fmaf(s_o[lr][nl], wx[1], rowacc);
                rowacc = __builtin_fmaf(s_e[lr][nl + 1], wx[2], rowacc);
                rowacc = __builtin_fmaf(s_o[lr][nl + 1], wx[3], rowacc);
                d = __builtin_fmaf(rowacc, wy[q], d);
            }
            const long long pix = (long long)m * dst.W + n;
            if (dst_bn) {
                const float yv = ydv[qq];
                const float v = __builtin_fmaf(yv - ch.mean, ch.scale, ch.beta);
                if (dst.act && !(v > 0.f)) d *= dst.slope;
                sg += (double)d; sgx += (double)d * (double)((yv - ch.mean) * ch.rstd);
            }
            go[pix] = d;
        }
    }
    if (dst_bn) {
        const double sa = block_sum_d(sg, s_red);
        const double sb = block_sum_d(sgx, s_red);
        if (t == 0) {
            double* o = (from_a ? bsums_a : bsums_b) + ((long long)k * dst.C + cd) * 2;
            atomicAdd(o, sa); atomicAdd(o + 1, sb);
        }
    }
}

// ---------------------------------------------------------------------------------------------
__global__ void bn_param_grads_kernel(const BnGradEntry* __restrict__ table, const double* __restrict__ bsums_base,
                                      int n_samples, float* __restrict__ dbn)
{
    bn_param_grads_entry(table[blockIdx.x], bsums_base, n_samples, dbn);
}

// nn.BatchNorm2d(momentum = 0.1) in training mode (models/common.py:96-97) after each of the n_samples batch-1 forwards, in order:
//   running_mean = (1 - m) running_mean + m mean_k;  running_var = (1 - m) running_var + m var_k * n / (n - 1)   (unbiased, n = H * W)
// running is laid out like the BN block: mean at the gamma slots, variance at the beta slots.
__global__ void bn_update_running_kernel(const BnGradEntry* __restrict__ table, const double* __restrict__ fstats, int n_samples, float momentum,
                                         float* __restrict__ running)
{
    const BnGradEntry e = table[blockIdx.x];
    const double n = (double)e.hw;
    for (int c = threadIdx.x; c < e.C; c += blockDim.x) {
        float rm = running[e.bn_off + c], rv = running[e.bn_off + e.C + c];
        for (int k = 0; k < n_samples; ++k) {
            const double* s = fstats + e.bsums_off + ((long long)k * e.C + c) * 2;
            const double m = s[0] / n;
            double var = s[1] / n - m * m; if (var < 0) var = 0;
            const float unb = (float)(n > 1 ? var * n / (n - 1) : var);
            rm = (1.f - momentum) * rm + momentum * (float)m;
            rv = (1.f - momentum) * rv + momentum * unb;
        }
        running[e.bn_off + c] = rm; running[e.bn_off + e.C + c] = rv;
    }
}
// eval mode: the consumers form their channel constants from (sum, sum of squares); write the pair that reproduces mean = running_mean,
// biased variance = running_var for every sample, and let no producer add to it
__global__ void bn_eval_fill_kernel(const BnGradEntry* __restrict__ table, double* __restrict__ fstats, int n_samples, const float* __restrict__ running)
{
    const BnGradEntry e = table[blockIdx.x];
    const double n = (double)e.hw;
    for (int c = threadIdx.x; c < e.C; c += blockDim.x) {
        const double rm = running[e.bn_off + c], rv = running[e.bn_off + e.C + c];
        for (int k = 0; k < n_samples; ++k) {
            double* s = fstats + e.bsums_off + ((long long)k * e.C + c) * 2;
            s[0] = rm * n; s[1] = (rv + rm * rm) * n;
        }
    }
}

// ---- local reparameterisation (BayTorch/modules/reparam_layers.py:59-72) ---------------------------------------------------
__global__ __launch_bounds__(256) void lrt_sigma2_kernel(const float* __restrict__ rho, long long n, float* __restrict__ sig2)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float s = softplus_f(rho[i]);
        sig2[i] = s * s;
    }
}
__global__ __launch_bounds__(256) void lrt_drho_kernel(const float* __restrict__ dsig2, const float* __restrict__ rho, long long n, float* __restrict__ drho)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = dsig2[i];
        if (d != 0.f) { const float r = rho[i]; drho[i] += d * 2.f * softplus_f(r) * sigmoid_f(r); }
    }
}
// eps of element j = c * HW + p of the layer's output: lane j & 3 of Philox block j >> 2 (RNG domain LRT, stream layer_id, sample k0 + k)
__device__ __forceinline__ void lrt_eps4(RngKey key, int layer_id, int k, long long j0, float z[4])
{
    key.stream = ((uint32_t)DOMAIN_LRT << 24) | (uint32_t)layer_id; key.sample += (uint32_t)k;
    spec_normal4(key, (uint32_t)(j0 >> 2), z);
}
__global__ __launch_bounds__(256) void lrt_combine_kernel(const float* __restrict__ a, const float* __restrict__ s2, long long sstride, int C, long long HW,
                                                          RngKey key, int layer_id, OutDesc y)
{
    key = key_now(key);
    __shared__ double s_red[8];
    const int k = blockIdx.z, c = blockIdx.y;
    const float* __restrict__ ap = a + (long long)k * sstride + (long long)c * HW;
    const float* __restrict__ sp = s2 + (long long)k * sstride + (long long)c * HW;
    float* __restrict__ yp = y.data + (long long)k * y.sstride + (long long)c * HW;
    double sum = 0.0, sq = 0.0;
    // 4 consecutive elements per thread; (c * HW + p) & 3 need not be 0, so every element looks up its own Philox block lane
    for (int it = 0; it < 4; ++it) {
        const long long p = ((long long)blockIdx.x * 4 + it) * 256 + threadIdx.x;
        if (p >= HW) break;
        const long long j = (long long)c * HW + p;
        float z[4]; lrt_eps4(key, layer_id, k, j, z);
        const float v = ap[p] + sqrtf(1e-16f + sp[p]) * z[j & 3];
        yp[p] = v; sum += (double)v; sq += (double)v * (double)v;
    }
    if (y.stats) {
        const double s0 = block_sum_d(sum, s_red);
        const double s1 = block_sum_d(sq, s_red);
        if (threadIdx.x == 0) { double* o = y.stats + ((long long)k * C + c) * 2; atomicAdd(o, s0); atomicAdd(o + 1, s1); }
    }
}
__global__ __launch_bounds__(256) void lrt_ds2_kernel(GView gy, const float* __restrict__ s2, long long sstride, long long HW, RngKey key, int layer_id,
                                                      float* __restrict__ ds2)
{
    key = key_now(key);
    __shared__ ChanBwd s_cb;
    const int k = blockIdx.z, c = blockIdx.y;
    if (threadIdx.x == 0) s_cb = chan_bwd(gy, k, c);
    __syncthreads();
    const ChanBwd cb = s_cb;
    const float* __restrict__ gp = gy.ga + (long long)k * gy.gstride + (long long)c * HW;
    const float* __restrict__ yp = gy.y ? gy.y + (long long)k * gy.ystride + (long long)c * HW : nullptr;
    const float* __restrict__ sp = s2 + (long long)k * sstride + (long long)c * HW;
    float* __restrict__ dp = ds2 + (long long)k * sstride + (long long)c * HW;
    for (int it = 0; it < 4; ++it) {
        const long long p = ((long long)blockIdx.x * 4 + it) * 256 + threadIdx.x;
        if (p >= HW) break;
        const long long j = (long long)c * HW + p;
        float z[4]; lrt_eps4(key, layer_id, k, j, z);
        const float dy = (gy.stats && yp) ? apply_bwd(cb, gp[p], yp[p]) : gp[p];
        dp[p] = dy * z[j & 3] / (2.f * sqrtf(1e-16f + sp[p]));
    }
}

}  // namespace

int launch_finalize_dx(const FoldSrc* srcs, int n_src, const TView& x, float* ga, long long ga_sstride, double* bsums,
                       int n_samples, hipStream_t st)
{
    if (n_src < 1 || n_src > MAX_FOLD_SRC) { set_error("finalize_dx: %d gradient sources (1..%d supported)", n_src, MAX_FOLD_SRC); return -1; }
    for (int i = 0; i < n_src; ++i)
        if (srcs[i].pad < 0 || srcs[i].pad > 2) { set_error("finalize_dx: pad %d unsupported", srcs[i].pad); return -1; }
    for (int i = 0; i < n_src; ++i)
        if (srcs[i].pad > 0 && (x.H <= srcs[i].pad || x.W <= srcs[i].pad)) { set_error("finalize_dx: reflection padding %d needs H,W > %d", srcs[i].pad, srcs[i].pad); return -1; }
    const long long HW = (long long)x.H * x.W;
    FoldSrc s0 = srcs[0], s1 = n_src > 1 ? srcs[1] : srcs[0];
    const auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    int maxpad = 0, any_mul = 0;
    for (int i = 0; i < n_src; ++i) { if (srcs[i].pad > maxpad) maxpad = srcs[i].pad; any_mul |= srcs[i].mul2v; }      // the float4 kernel folds pad <= 1
    if (n_src <= 2 && !any_mul && maxpad <= 1 && (x.W & 3) == 0 && x.H >= 2 && ((x.sstride | ga_sstride) & 3) == 0 && al16(x.data) && al16(ga)) {
        dim3 grid((unsigned)((HW / 4 + 256 * V_GROUPS - 1) / (256 * V_GROUPS)), x.C, n_samples);
        mfvi_launch(finalize_dx_vec_kernel, grid, dim3(256), 0, st, s0, s1, n_src, x, ga, ga_sstride, bsums);
        return (int)hipGetLastError();
    }
    dim3 grid((unsigned)((HW + 256 * EW_ITEMS - 1) / (256 * EW_ITEMS)), x.C, n_samples);
    FoldSrcs S; S.n = n_src;
    for (int i = 0; i < MAX_FOLD_SRC; ++i) S.s[i] = srcs[i < n_src ? i : 0];
    mfvi_launch(finalize_dx_kernel, grid, dim3(256), 0, st, S, x, ga, ga_sstride, bsums);
    return (int)hipGetLastError();
}

// One padded source + the backward-data of a 1x1 consumer with cs1 <= 8 output channels in place (finalize_dx_vec1_kernel).  -2: shape not served
int launch_finalize_dx_inline1x1(const FoldSrc& s0, const GView& g1, const float* w1, long long w1_sstride, int cs1, const TView& x, float* ga,
                                 long long ga_sstride, double* bsums, int n_samples, hipStream_t st)
{
    const auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if (cs1 < 1 || cs1 > 8 || s0.mul2v || s0.pad > 1 || (x.W & 3) || x.H < 2 || ((x.sstride | ga_sstride | g1.gstride) & 3) || !al16(x.data) || !al16(ga) ||
        !al16(g1.ga) || (g1.stats && (!al16(g1.y) || (g1.ystride & 3))) || g1.H != x.H || g1.W != x.W) return -2;
    const long long HW = (long long)x.H * x.W;
    dim3 grid((unsigned)((HW / 4 + 256 * V1_GROUPS - 1) / (256 * V1_GROUPS)), x.C, n_samples);
    if (cs1 <= 4) mfvi_launch(finalize_dx_vec1_kernel<4>, grid, dim3(256), 0, st, s0, g1, w1, w1_sstride, cs1, x, ga, ga_sstride, bsums);
    else mfvi_launch(finalize_dx_vec1_kernel<8>, grid, dim3(256), 0, st, s0, g1, w1, w1_sstride, cs1, x, ga, ga_sstride, bsums);
    return (int)hipGetLastError();
}

int launch_lrt_sigma2(const float* rho, long long n, float* sig2, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(lrt_sigma2_kernel, dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, st, rho, n, sig2);
    return (int)hipGetLastError();
}

int launch_lrt_combine(const float* a, const float* s2, long long sstride, int C, long long HW, RngKey key, int layer_id, OutDesc y,
                       int n_samples, hipStream_t st)
{
    dim3 grid((unsigned)((HW + 1023) / 1024), C, n_samples);
    hipLaunchKernelGGL(lrt_combine_kernel, grid, dim3(256), 0, st, a, s2, sstride, C, HW, key, layer_id, y);
    return (int)hipGetLastError();
}

int launch_lrt_ds2(const GView& gy, const float* s2, long long sstride, RngKey key, int layer_id, float* ds2, int n_samples, hipStream_t st)
{
    const long long HW = (long long)gy.H * gy.W;
    dim3 grid((unsigned)((HW + 1023) / 1024), gy.C, n_samples);
    hipLaunchKernelGGL(lrt_ds2_kernel, grid, dim3(256), 0, st, gy, s2, sstride, HW, key, layer_id, ds2);
    return (int)hipGetLastError();
}

int launch_lrt_drho(const float* dsig2, const float* rho, long long n, float* drho, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(lrt_drho_kernel, dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, st, dsig2, rho, n, drho);
    return (int)hipGetLastError();
}

int launch_concat_up_fwd(const TView* a, const TView& b, OutDesc out, int nearest, int n_samples, hipStream_t st)
{
    const int H = a ? a->H : 2 * b.H, W = a ? a->W : 2 * b.W;          // Concat's centre-crop: the up-sampled branch loses its last row / column when the skip branch is odd-sized
    const int Ct = (a ? a->C : 0) + b.C;
    const long long HW = (long long)H * W;
    TView av = a ? *a : b;
    const auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    static const int tiled_min_w = [] { const char* e = getenv("MFVI_CONCAT_TILED"); return e ? atoi(e) : 128; }();      // MFVI_CONCAT_TILED=0: never (A/B)
    if (tiled_min_w > 0 && (W & 3) == 0 && W >= tiled_min_w && (out.sstride & 3) == 0 && al16(out.data) && (!a || ((a->sstride & 3) == 0 && al16(a->data)))) {
        const int tiles_x = (b.W + CF_TW - 1) / CF_TW, tiles_y = (b.H + CF_TH - 1) / CF_TH;
        static const int tpb = [] { const char* e = getenv("MFVI_CONCAT_TPB"); return e ? atoi(e) : 4; }();
        const int n_tiles = tiles_x * tiles_y;
        hipLaunchKernelGGL(concat_up_fwd_tiled_kernel, dim3((n_tiles + tpb - 1) / tpb, Ct, n_samples), dim3(256), 0, st, av, a ? 1 : 0, b, out, H, W, nearest, tiles_x, n_tiles);
        return (int)hipGetLastError();
    }
    if ((W & 3) == 0 && (out.sstride & 3) == 0 && al16(out.data) && (!a || ((a->sstride & 3) == 0 && al16(a->data)))) {
        dim3 grid((unsigned)((HW / 4 + 256 * V_GROUPS - 1) / (256 * V_GROUPS)), Ct, n_samples);
        hipLaunchKernelGGL(concat_up_fwd_vec_kernel, grid, dim3(256), 0, st, av, a ? 1 : 0, b, out, H, W, nearest);
        return (int)hipGetLastError();
    }
    dim3 grid((unsigned)((HW + 256 * EW_ITEMS - 1) / (256 * EW_ITEMS)), Ct, n_samples);
    hipLaunchKernelGGL(concat_up_fwd_kernel, grid, dim3(256), 0, st, av, a ? 1 : 0, b, out, H, W, nearest);
    return (int)hipGetLastError();
}

int launch_concat_up_bwd(const GView& gc, const TView* a, float* ga_a, long long ga_a_sstride, double* bsums_a,
                         const TView& b, float* ga_b, long long ga_b_sstride, double* bsums_b, int nearest, int n_samples, hipStream_t st)
{
    const int Ct = (a ? a->C : 0) + b.C;
    // aligned pairs (float2) when every row starts on an even element; odd widths (Concat's crop) / odd strides take element loads
    const auto al8 = [](const void* q) { return ((uintptr_t)q & 7) == 0; };
    const int pairs = !((gc.W & 1) || ((gc.gstride | gc.ystride) & 1) || !al8(gc.ga) || (gc.y && !al8(gc.y)) ||
                        (a && (((a->sstride | ga_a_sstride) & 1) || !al8(a->data) || !al8(ga_a))));
    TView av = a ? *a : b;
    static const int force = [] { const char* e = getenv("MFVI_CONCAT_BWD_TILE"); return e ? atoi(e) : 0; }();      // 64 / 32 / 16: A/B of the tile shapes
    const int tw = force ? force : (b.W >= 48 ? 64 : b.W >= 24 ? 32 : 16);
#define CB_GO(TH_, TW_) { \
        const int tiles_x = (b.W + TW_ - 1) / TW_, tiles_y = (b.H + TH_ - 1) / TH_; \
        dim3 grid((unsigned)(tiles_x * tiles_y), Ct, n_samples); \
        mfvi_launch((concat_up_bwd_kernel<TH_, TW_>), grid, dim3(256), 0, st, gc, av, a ? 1 : 0, ga_a, ga_a_sstride, bsums_a, b, ga_b, \
                    ga_b_sstride, bsums_b, tiles_x, nearest, pairs); }
    if (tw == 64) CB_GO(16, 64) else if (tw == 32) CB_GO(32, 32) else CB_GO(16, 16)
#undef CB_GO
    return (int)hipGetLastError();
}

int launch_bn_update_running(const BnGradEntry* table_dev, int n_entries, int max_c, const double* fstats_base, int n_samples, float momentum,
                             float* running, hipStream_t st)
{
    if (n_entries == 0) return 0;
    hipLaunchKernelGGL(bn_update_running_kernel, dim3(n_entries), dim3(max_c < 64 ? 64 : (max_c > 256 ? 256 : ((max_c + 63) / 64) * 64)), 0, st,
                       table_dev, fstats_base, n_samples, momentum, running);
    return (int)hipGetLastError();
}

int launch_bn_eval_fill(const BnGradEntry* table_dev, int n_entries, int max_c, double* fstats_base, int n_samples, const float* running, hipStream_t st)
{
    if (n_entries == 0) return 0;
    hipLaunchKernelGGL(bn_eval_fill_kernel, dim3(n_entries), dim3(max_c < 64 ? 64 : (max_c > 256 ? 256 : ((max_c + 63) / 64) * 64)), 0, st,
                       table_dev, fstats_base, n_samples, running);
    return (int)hipGetLastError();
}

int launch_bn_param_grads(const BnGradEntry* table_dev, int n_entries, int max_c, const double* bsums_base, int n_samples,
                          float* dbn, hipStream_t st)
{
    if (n_entries == 0) return 0;
    hipLaunchKernelGGL(bn_param_grads_kernel, dim3(n_entries), dim3(max_c < 64 ? 64 : (max_c > 256 ? 256 : ((max_c + 63) / 64) * 64)), 0, st,
                       table_dev, bsums_base, n_samples, dbn);
    return (int)hipGetLastError();
}
