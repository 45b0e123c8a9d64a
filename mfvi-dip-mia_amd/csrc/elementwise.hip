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

namespace {

constexpr int EW_ITEMS = 4;     // pixels per thread

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void finalize_dx_kernel(FoldSrc s0, FoldSrc s1, int n_src, TView x,
                                                          float* __restrict__ ga, long long ga_sstride,
                                                          double* __restrict__ bsums)
{
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
        for (int s = 0; s < n_src; ++s) {
            const FoldSrc src = s == 0 ? s0 : s1;
            const int p = src.pad, Hp = H + 2 * p, Wp = W + 2 * p;
            const float* __restrict__ base = src.d + (long long)k * src.sstride + (long long)c * Hp * Wp;
            if (p == 0) { d += base[(long long)r * Wp + q]; continue; }
            // rows of the padded gradient that fold onto r: r+1 always, 0 if r == 1, H+1 if r == H-2
            int rows[3], cols[3], nr = 0, nc = 0;
            rows[nr++] = r + 1; if (r == 1) rows[nr++] = 0; if (r == H - 2) rows[nr++] = H + 1;
            cols[nc++] = q + 1; if (q == 1) cols[nc++] = 0; if (q == W - 2) cols[nc++] = W + 1;
            for (int a = 0; a < nr; ++a)
                for (int b = 0; b < nc; ++b) d += base[(long long)rows[a] * Wp + cols[b]];
        }
        if (has_bn) {
            const float yv = yx[pix];
            const float v = __builtin_fmaf(yv - ch.mean, ch.scale, ch.beta);
            if (x.act && !(v > 0.f)) d *= x.slope;
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

// bilinear x2, align_corners=False: src = (dst+0.5)/2-0.5 clamped at 0, i1 = min(i0+1, n-1)
__device__ __forceinline__ void up_coef(int d, int n, int& i0, int& i1, float& l1)
{
    float s = ((float)d + 0.5f) * 0.5f - 0.5f; s = s < 0.f ? 0.f : s;
    i0 = (int)s; i1 = min(i0 + 1, n - 1); l1 = s - (float)i0;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void concat_up_fwd_kernel(TView a, int has_a, TView b, OutDesc out, int H, int W)
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

// ---------------------------------------------------------------------------------------------
// grid: x = pixel blocks of the DESTINATION tensor (A: H x W, B: H/2 x W/2), y = channel of the concat, z = sample
__global__ __launch_bounds__(256) void concat_up_bwd_kernel(GView gc, TView a, int has_a, float* __restrict__ ga_a,
                                                            long long ga_a_sstride, double* __restrict__ bsums_a,
                                                            TView b, float* __restrict__ ga_b, long long ga_b_sstride,
                                                            double* __restrict__ bsums_b)
{
    __shared__ ChanFwd s_ch;
    __shared__ ChanBwd s_cg;
    __shared__ double s_red[8];
    const int t = threadIdx.x, k = blockIdx.z, c = blockIdx.y;
    const int Ca = has_a ? a.C : 0;
    const bool from_a = c < Ca;
    const int H = gc.H, W = gc.W;
    const long long HW = (long long)H * W;
    if (t == 0) { s_ch = from_a ? chan_fwd(a, k, c) : chan_fwd(b, k, c - Ca); s_cg = chan_bwd(gc, k, c); }
    __syncthreads();
    const ChanFwd ch = s_ch; const ChanBwd cg = s_cg;
    const float* __restrict__ gap = gc.ga + (long long)k * gc.gstride + (long long)c * HW;
    const float* __restrict__ ycp = gc.y + (long long)k * gc.ystride + (long long)c * HW;
    const bool cat_bn = gc.stats != nullptr;
    double sg = 0.0, sgx = 0.0;
    const TView& dst = from_a ? a : b;
    const int cd = from_a ? c : c - Ca;
    const long long HWd = (long long)dst.H * dst.W;
    if ((long long)blockIdx.x * EW_ITEMS * 256 >= HWd) return;     // grid is sized for A; B has a quarter of the pixels
    const float* __restrict__ yd = dst.data + (long long)k * dst.sstride + (long long)cd * HWd;
    float* __restrict__ go = (from_a ? ga_a + (long long)k * ga_a_sstride : ga_b + (long long)k * ga_b_sstride) + (long long)cd * HWd;
    for (int it = 0; it < EW_ITEMS; ++it) {
        const long long pix = ((long long)blockIdx.x * EW_ITEMS + it) * 256 + t;
        if (pix >= HWd) break;
        float d;
        if (from_a) {
            d = cat_bn ? apply_bwd(cg, gap[pix], ycp[pix]) : gap[pix];
        } else {
            // Adjoint of the bilinear x2 gather (align_corners=False).  Low-res row m feeds hi-res rows 2m-1..2m+2 with
            // weights .25 .75 .75 .25; at the borders the clamped taps collapse: row 0 gets 1.0 from hi-res row 0 and
            // the last row 1.0 from the last hi-res row.
            const int m = (int)(pix / dst.W), n = (int)(pix - (long long)m * dst.W);
            float wy[4], wx[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int oy = 2 * m - 1 + a, ox = 2 * n - 1 + a;
                float v = (a == 0 || a == 3) ? 0.25f : 0.75f;
                if (oy < 0 || oy >= H) v = 0.f; else if ((m == 0 && a == 1) || (m == dst.H - 1 && a == 2)) v = 1.f;
                wy[a] = v;
                float u = (a == 0 || a == 3) ? 0.25f : 0.75f;
                if (ox < 0 || ox >= W) u = 0.f; else if ((n == 0 && a == 1) || (n == dst.W - 1 && a == 2)) u = 1.f;
                wx[a] = u;
            }
            d = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (wy[a] == 0.f) continue;
                const int oy = 2 * m - 1 + a;
                float rowacc = 0.f;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (wx[b] == 0.f) continue;
                    const long long hp = (long long)oy * W + (2 * n - 1 + b);
                    const float g = cat_bn ? apply_bwd(cg, gap[hp], ycp[hp]) : gap[hp];
                    rowacc = __builtin_fmaf(g, wx[b], rowacc);
                }
                d = __builtin_fmaf(rowacc, wy[a], d);
            }
        }
        if (dst.stats != nullptr) {
            const float yv = yd[pix];
            const float v = __builtin_fmaf(yv - ch.mean, ch.scale, ch.beta);
            if (dst.act && !(v > 0.f)) d *= dst.slope;
            sg += (double)d; sgx += (double)d * (double)((yv - ch.mean) * ch.rstd);
        }
        go[pix] = d;
    }
    if (dst.stats != nullptr) {
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
    const BnGradEntry e = table[blockIdx.x];
    for (int c = threadIdx.x; c < e.C; c += blockDim.x) {
        double sb = 0, sg = 0;
        for (int k = 0; k < n_samples; ++k) {
            const double* s = bsums_base + e.bsums_off + ((long long)k * e.C + c) * 2;
            sb += s[0]; sg += s[1];
        }
        dbn[e.bn_off + c] += (float)sg;            // d gamma = sum ga * xhat
        dbn[e.bn_off + e.C + c] += (float)sb;      // d beta  = sum ga
    }
}

}  // namespace

int launch_finalize_dx(const FoldSrc* srcs, int n_src, const TView& x, float* ga, long long ga_sstride, double* bsums,
                       int n_samples, hipStream_t st)
{
    if (n_src < 1 || n_src > 2) { set_error("finalize_dx: %d gradient sources (1..2 supported)", n_src); return -1; }
    for (int i = 0; i < n_src; ++i)
        if (srcs[i].pad < 0 || srcs[i].pad > 1) { set_error("finalize_dx: pad %d unsupported", srcs[i].pad); return -1; }
    if (n_src > 0 && (x.H < 2 || x.W < 2) && (srcs[0].pad == 1 || (n_src > 1 && srcs[1].pad == 1))) {
        set_error("finalize_dx: reflection padding needs H,W >= 2"); return -1;
    }
    const long long HW = (long long)x.H * x.W;
    dim3 grid((unsigned)((HW + 256 * EW_ITEMS - 1) / (256 * EW_ITEMS)), x.C, n_samples);
    FoldSrc s0 = srcs[0], s1 = n_src > 1 ? srcs[1] : srcs[0];
    hipLaunchKernelGGL(finalize_dx_kernel, grid, dim3(256), 0, st, s0, s1, n_src, x, ga, ga_sstride, bsums);
    return (int)hipGetLastError();
}

int launch_concat_up_fwd(const TView* a, const TView& b, OutDesc out, int n_samples, hipStream_t st)
{
    const int H = 2 * b.H, W = 2 * b.W;
    const int Ct = (a ? a->C : 0) + b.C;
    const long long HW = (long long)H * W;
    dim3 grid((unsigned)((HW + 256 * EW_ITEMS - 1) / (256 * EW_ITEMS)), Ct, n_samples);
    TView av = a ? *a : b;
    hipLaunchKernelGGL(concat_up_fwd_kernel, grid, dim3(256), 0, st, av, a ? 1 : 0, b, out, H, W);
    return (int)hipGetLastError();
}

int launch_concat_up_bwd(const GView& gc, const TView* a, float* ga_a, long long ga_a_sstride, double* bsums_a,
                         const TView& b, float* ga_b, long long ga_b_sstride, double* bsums_b, int n_samples, hipStream_t st)
{
    const int Ct = (a ? a->C : 0) + b.C;
    const long long HW = (long long)gc.H * gc.W;      // A's size; B blocks beyond its pixels exit immediately
    dim3 grid((unsigned)((HW + 256 * EW_ITEMS - 1) / (256 * EW_ITEMS)), Ct, n_samples);
    TView av = a ? *a : b;
    hipLaunchKernelGGL(concat_up_bwd_kernel, grid, dim3(256), 0, st, gc, av, a ? 1 : 0, ga_a, ga_a_sstride, bsums_a, b, ga_b,
                       ga_b_sstride, bsums_b);
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
