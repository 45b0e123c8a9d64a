// K5/K6/K9/K10 — HBM-bound reductions and element-wise passes of one ELBO iteration:
// Gaussian NLL (utils/bayesian_utils.py:29-32), KL(prior || posterior) (BayTorch/modules/module.py:64-80),
// AdamW(wd=0) (bayesian_optimization.py:1356-1357), RNG fills, per-iteration bookkeeping
// (bayesian_optimization.py:1374-1406, utils/common_utils.py:297-353).
#include "common.h"
#include <algorithm>
#include <type_traits>
#include "../../include/mfvi_hip.h"

namespace {

__device__ __forceinline__ void block_atomic_add(double v, double* dst, double* red)
{
    const double s = block_sum_d(v, red);
    if (threadIdx.x == 0) atomicAdd(dst, s);
}

// ---- gaussian_nll ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gaussian_nll_kernel(const float* __restrict__ out, const float* __restrict__ target,
                                                           int H, int W, int f, float grad_scale, float* __restrict__ dout,
                                                           double* __restrict__ nll_sum)
{
    __shared__ double s_red[8];
    const int k = blockIdx.y;
    const int h = H / f, w = W / f;
    const long long n = (long long)h * w, HW = (long long)H * W;
    const float* __restrict__ o = out + (long long)k * 2 * HW;
    float* __restrict__ d = dout ? dout + (long long)k * 2 * HW : nullptr;
    double acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i - (long long)y * w);
        const long long p = (long long)(y * f) * W + (long long)x * f;
        const float m = o[p], sraw = o[HW + p];
        const float s = fminf(fmaxf(sraw, -20.f), 20.f);
        const bool inside = (sraw >= -20.f) && (sraw <= 20.f);
        const float df = target[i] - m, e = expf(s);
        acc += (double)(e * df * df - s);
        if (d) {
            d[p] = grad_scale * (-2.f * e * df) / (float)n;
            d[HW + p] = inside ? grad_scale * (e * df * df - 1.f) / (float)n : 0.f;
        }
    }
    block_atomic_add(acc / (double)n, nll_sum, s_red);
}

// factor 1, W % 4 == 0, 16-byte aligned rows: float4 lanes, a thread's four groups requested before any is used (round 4: the scalar form was
// a chain of 16 dependent-latency iterations per thread, 15 us for 24 MB).  Same per-element arithmetic, float partial sums per group folded
// into the thread's fp64 sum.
__global__ __launch_bounds__(256) void gaussian_nll_vec_kernel(const float* __restrict__ out, const float* __restrict__ target, long long HW,
                                                               float grad_scale, float* __restrict__ dout, double* __restrict__ nll_sum)
{
    __shared__ double s_red[8];
    const int k = blockIdx.y;
    const long long ng = HW >> 2;
    const float* __restrict__ o = out + (long long)k * 2 * HW;
    float* __restrict__ d = dout ? dout + (long long)k * 2 * HW : nullptr;
    const float nf = (float)HW;
    double acc = 0;
    for (long long g0 = (long long)blockIdx.x * 256 + threadIdx.x; g0 < ng; g0 += (long long)gridDim.x * 256 * 4) {
        float4 m4[4], s4[4], t4[4]; long long gi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            gi[u] = g0 + (long long)u * gridDim.x * 256;
            const long long gc = gi[u] < ng ? gi[u] : ng - 1;
            m4[u] = *reinterpret_cast<const float4*>(o + 4 * gc); s4[u] = *reinterpret_cast<const float4*>(o + HW + 4 * gc); t4[u] = *reinterpret_cast<const float4*>(target + 4 * gc);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (gi[u] >= ng) continue;
            const float mm[4] = {m4[u].x, m4[u].y, m4[u].z, m4[u].w}, ss[4] = {s4[u].x, s4[u].y, s4[u].z, s4[u].w}, tt[4] = {t4[u].x, t4[u].y, t4[u].z, t4[u].w};
            float dm[4], ds[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const float sraw = ss[l];
                const float s_ = fminf(fmaxf(sraw, -20.f), 20.f);
                const bool inside = (sraw >= -20.f) && (sraw <= 20.f);
                const float df = tt[l] - mm[l], e = expf(s_);
                acc += (double)(e * df * df - s_);
                dm[l] = grad_scale * (-2.f * e * df) / nf;
                ds[l] = inside ? grad_scale * (e * df * df - 1.f) / nf : 0.f;
            }
            if (d) {
                *reinterpret_cast<float4*>(d + 4 * gi[u]) = make_float4(dm[0], dm[1], dm[2], dm[3]);
                *reinterpret_cast<float4*>(d + HW + 4 * gi[u]) = make_float4(ds[0], ds[1], ds[2], ds[3]);
            }
        }
    }
    block_atomic_add(acc / (double)HW, nll_sum, s_red);
}

// ---- gaussian_nll_inpainting: sigmoid on the 3 colour channels, one shared log-precision channel, mask ----
__global__ __launch_bounds__(256) void gaussian_nll_inp_kernel(const float* __restrict__ out, const float* __restrict__ target,
                                                               const float* __restrict__ mask, int mask_channels, long long HW,
                                                               float grad_scale, float* __restrict__ dout, double* __restrict__ nll_sum)
{
    __shared__ double s_red[8];
    const int k = blockIdx.y;
    const float* __restrict__ o = out + (long long)k * 4 * HW;
    float* __restrict__ d = dout ? dout + (long long)k * 4 * HW : nullptr;
    const float inv_n = 1.f / (float)(3 * HW);
    double acc = 0;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long long)gridDim.x * 256) {
        const float sraw = o[3 * HW + p];
        const float s = fminf(fmaxf(sraw, -20.f), 20.f);
        const bool inside = (sraw >= -20.f) && (sraw <= 20.f);
        const float e = expf(s);
        float ds = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float m = sigmoid_f(o[c * HW + p]);
            const float mk = mask[(mask_channels == 3 ? c : 0) * HW + p];
            const float df = target[c * HW + p] - m;
            acc += (double)((e * df * df - s) * mk);
            if (d) {
                d[c * HW + p] = grad_scale * (-2.f * e * df) * mk * m * (1.f - m) * inv_n;
                ds += (e * df * df - 1.f) * mk;
            }
        }
        if (d) d[3 * HW + p] = inside ? grad_scale * ds * inv_n : 0.f;
    }
    block_atomic_add(acc * (double)inv_n, nll_sum, s_red);
}


// ---- gaussian_nll / gaussian_nll_inpainting on separate mu / neg_logvar tensors (the call shape of the drop-in API) ----
// loss[c][p] = (exp(s) * (t[c][p] - mu[c][p])^2 - s) * mask,  s = clamp(s_raw[Cs == 1 ? 0 : c][p], -20, 20)
__global__ __launch_bounds__(256) void gnll_bcast_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ sraw, const float* __restrict__ target,
                                                             const float* __restrict__ mask, int C, int Cs, int Cm, long long HW, double scale,
                                                             double* __restrict__ loss_sum)
{
    __shared__ double s_red[8];
    double acc = 0;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long long)gridDim.x * 256)
        for (int c = 0; c < C; ++c) {
            const float s = fminf(fmaxf(sraw[(Cs == 1 ? 0 : c) * HW + p], -20.f), 20.f);
            const float df = target[c * HW + p] - mu[c * HW + p];
            const float mk = mask ? mask[(Cm == 1 ? 0 : c) * HW + p] : 1.f;
            acc += (double)((expf(s) * df * df - s) * mk);
        }
    block_atomic_add(acc * scale, loss_sum, s_red);
}
__global__ __launch_bounds__(256) void gnll_bcast_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ sraw, const float* __restrict__ target,
                                                             const float* __restrict__ mask, int C, int Cs, int Cm, long long HW, float scale,
                                                             const float* __restrict__ gout, float* __restrict__ dmu, float* __restrict__ ds)
{
    const float g = gout[0] * scale;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long long)gridDim.x * 256) {
        float dsum = 0.f;
        for (int c = 0; c < C; ++c) {
            const float sr = sraw[(Cs == 1 ? 0 : c) * HW + p];
            const float s = fminf(fmaxf(sr, -20.f), 20.f);
            const bool inside = (sr >= -20.f) && (sr <= 20.f);
            const float df = target[c * HW + p] - mu[c * HW + p], e = expf(s);
            const float mk = mask ? mask[(Cm == 1 ? 0 : c) * HW + p] : 1.f;
            dmu[c * HW + p] = g * (-2.f * e * df) * mk;
            const float d = inside ? g * (e * df * df - 1.f) * mk : 0.f;
            if (Cs == 1) dsum += d; else ds[c * HW + p] = d;
        }
        if (Cs == 1) ds[p] = dsum;
    }
}

// ---- run_inp_dip's loss: mse_loss(out[:, :3].sigmoid() * mask, img * mask), mean over 3*H*W ----
__global__ __launch_bounds__(256) void mse_sigmoid_masked_kernel(const float* __restrict__ out, const float* __restrict__ target,
                                                                 const float* __restrict__ mask, int mask_channels, long long HW,
                                                                 float grad_scale, float* __restrict__ dout, double* __restrict__ mse_sum)
{
    __shared__ double s_red[8];
    const int k = blockIdx.y;
    const float* __restrict__ o = out + (long long)k * 4 * HW;
    float* __restrict__ d = dout ? dout + (long long)k * 4 * HW : nullptr;
    const float inv_n = 1.f / (float)(3 * HW);
    double acc = 0;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long long)gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float m = sigmoid_f(o[c * HW + p]);
            const float mk = mask[(mask_channels == 3 ? c : 0) * HW + p];
            const float df = m * mk - target[c * HW + p] * mk;
            acc += (double)(df * df);
            if (d) d[c * HW + p] = grad_scale * 2.f * df * mk * m * (1.f - m) * inv_n;
        }
        if (d) d[3 * HW + p] = 0.f;
    }
    block_atomic_add(acc * (double)inv_n, mse_sum, s_red);
}

// ---- mse_loss on one output channel (DIP / SGLD siblings), optionally after the SR projection out[..., ::f, ::f] ----
__global__ __launch_bounds__(256) void mse_channel_kernel(const float* __restrict__ out, const float* __restrict__ target, int C, int H, int W,
                                                          int channel, int f, float grad_scale, float* __restrict__ dout, double* __restrict__ mse_sum)
{
    __shared__ double s_red[8];
    const int k = blockIdx.y;
    const long long HW = (long long)H * W;
    const int w = W / f; const long long n = (long long)(H / f) * w;
    const float* __restrict__ o = out + (long long)k * C * HW;
    float* __restrict__ d = dout ? dout + (long long)k * C * HW : nullptr;
    const float gs = grad_scale * 2.f / (float)n;
    double acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i - (long long)y * W);
        float g = 0.f;
        if (y % f == 0 && x % f == 0 && y / f < H / f && x / f < w) {
            const float df = o[channel * HW + i] - target[(long long)(y / f) * w + x / f];
            acc += (double)(df * df); g = gs * df;
        }
        if (d) for (int c = 0; c < C; ++c) d[c * HW + i] = c == channel ? g : 0.f;
    }
    block_atomic_add(acc / (double)n, mse_sum, s_red);
}

// ---- Dropout2d masks (MC-dropout sibling) ----
__global__ __launch_bounds__(64) void dropout_mask_kernel(const DropEntry* __restrict__ table, RngKey key, float* __restrict__ arena)
{
    key = key_now(key);
    const DropEntry e = table[blockIdx.x];
    const int k = blockIdx.y;
    key.stream = ((uint32_t)DOMAIN_DROPOUT << 24) | (uint32_t)e.layer_id;
    key.sample += (uint32_t)k;
    const float keep = 1.0f / (1.0f - e.p);
    float* __restrict__ d = arena + e.drop_off + (long long)k * e.C;
    for (int blk = threadIdx.x; blk * 4 < e.C; blk += 64) {
        uint32_t r[4];
        philox4x32_10((uint32_t)blk, key.stream, key.sample, key.step, key.k0, key.k1, r);
#pragma unroll
        for (int l = 0; l < 4; ++l)
            if (blk * 4 + l < e.C) d[blk * 4 + l] = (float)(r[l] >> 8) * 5.9604644775390625e-08f >= e.p ? keep : 0.0f;
    }
}

// ---- KL ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kl_kernel(const float* __restrict__ mu, const float* __restrict__ rho, long long n,
                                                 float m0, float s0, double* __restrict__ kl_out)
{
    __shared__ double s_red[8];
    double acc = 0;
    const float log_s0 = logf(s0), s0sq = s0 * s0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float s = softplus_f(rho[i]), d = mu[i] - m0;
        acc += (double)(logf(s) - log_s0) + (double)((s0sq + d * d) / (2.f * s * s)) - 0.5;
    }
    block_atomic_add(acc, kl_out, s_red);
}

__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ rho, long long n,
                                                     float m0, float s0, float scale, float* __restrict__ dmu,
                                                     float* __restrict__ drho)
{
    const float s0sq = s0 * s0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float r = rho[i], s = softplus_f(r), d = mu[i] - m0, inv = 1.f / s;
        dmu[i] += scale * d * inv * inv;
        drho[i] += scale * (inv - (s0sq + d * d) * inv * inv * inv) * sigmoid_f(r);
    }
}

// ---- AdamW -----------------------------------------------------------------------------------------
// NaN guard of the CT runners (bayesian_optimization.py:380, 581-582, 792, 994: `if not torch.isnan(loss): optimizer.step()`) without a
// host sync: the data-term scalar of this iteration is read on the device (double accumulator and / or the float that rode the
// all-reduce); a NaN there leaves p, m, v untouched and the step counter of APPLIED updates (which sets the bias corrections, like
// torch's per-parameter state['step']) is not advanced.
struct StepGuard { const int* t_applied; const double* loss_d; const float* loss_f; };
__device__ __forceinline__ bool guard_skip(const StepGuard& g)
{
    return (g.loss_d && isnan(*g.loss_d)) || (g.loss_f && isnan(*g.loss_f));
}
// bias corrections of update number t (1-based): step_size = lr / (1 - b1^t), inv_sqrt_bc2 = 1 / sqrt(1 - b2^t); one lane per block
__device__ __forceinline__ void adam_bias(const StepGuard& g, float lr, float b1, float b2, float* s_bc, float& step_size, float& inv_sqrt_bc2)
{
    if (g.t_applied == nullptr) return;              // host-computed values stay
    if (threadIdx.x == 0) {
        const int t = *g.t_applied + 1;
        s_bc[0] = (float)((double)lr / (1.0 - pow((double)b1, (double)t)));
        s_bc[1] = (float)(1.0 / sqrt(1.0 - pow((double)b2, (double)t)));
    }
    __syncthreads();
    step_size = s_bc[0]; inv_sqrt_bc2 = s_bc[1];
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float b1, float b2, float eps,
                                                   float step_size, float inv_sqrt_bc2, float decay, float lr, StepGuard guard)
{
    __shared__ float s_bc[2];
    if (guard_skip(guard)) return;
    adam_bias(guard, lr, b1, b2, s_bc, step_size, inv_sqrt_bc2);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = p[i] * decay - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));       // decay = 1 - lr*wd (exactly 1 for wd = 0)
    }
}
__global__ void step_advance_kernel(int* __restrict__ t_applied, StepGuard guard)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && !guard_skip(guard)) *t_applied += 1;
}

// ---- fused tail of an ELBO iteration: KL, its gradient and AdamW(wd = 0) over [MU | RHO | BN] in one pass ----
// Same per-element arithmetic as kl_kernel / kl_bwd_kernel / adam_kernel.  The KL sum goes through per-block partials that a
// one-block kernel adds up in block order right behind: deterministic, and no same-address atomic chain (a ticket / atomicAdd
// per block serialises at ~70 ns each: 2048 blocks cost 150 us on MI355X).
constexpr int ELBO_UPDATE_MAX_BLOCKS = 2048;
struct ElboUpdateScratch { double partial[ELBO_UPDATE_MAX_BLOCKS]; };

__global__ __launch_bounds__(256) void elbo_update_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                          long long n_vi, long long n_bn, float m0, float s0, float temp, float b1, float b2,
                                                          float eps, float step_size, float inv_sqrt_bc2, ElboUpdateScratch* __restrict__ sc, float lr,
                                                          StepGuard guard)
{
    __shared__ double s_red[8];
    __shared__ float s_bc[2];
    const bool skip = guard_skip(guard);             // NaN data term: KL is still reported, nothing is written
    adam_bias(guard, lr, b1, b2, s_bc, step_size, inv_sqrt_bc2);
    const float log_s0 = logf(s0), s0sq = s0 * s0;
    auto adam = [&](long long i, float gi) {
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = p[i] - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    };
    double acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_vi; i += (long long)gridDim.x * 256) {
        const float mu = p[i], r = p[n_vi + i];
        const float s = softplus_f(r), d = mu - m0, inv = 1.f / s;
        acc += (double)(logf(s) - log_s0) + (double)((s0sq + d * d) / (2.f * s * s)) - 0.5;
        if (skip) continue;
        const float gmu = g[i] + temp * d * inv * inv;
        const float grho = g[n_vi + i] + temp * (inv - (s0sq + d * d) * inv * inv * inv) * sigmoid_f(r);
        g[i] = gmu; g[n_vi + i] = grho;
        adam(i, gmu); adam(n_vi + i, grho);
    }
    if (!skip)
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_bn; i += (long long)gridDim.x * 256) adam(2 * n_vi + i, g[2 * n_vi + i]);
    const double tot = block_sum_d(acc, s_red);
    if (threadIdx.x == 0) sc->partial[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void elbo_update_finish_kernel(const ElboUpdateScratch* __restrict__ sc, int n_blocks, double* __restrict__ kl_out,
                                                                 int* __restrict__ t_applied, StepGuard guard)
{
    __shared__ double s_red[8];
    double t = 0;
    for (int b = threadIdx.x; b < n_blocks; b += 256) t += sc->partial[b];
    t = block_sum_d(t, s_red);
    if (threadIdx.x == 0) { *kl_out = t; if (t_applied && !guard_skip(guard)) *t_applied += 1; }
}

// nearest /f projection x[..., ::f, ::f] (the SR runner's downsampler: bayesian_optimization.py:2095-2099)
__global__ __launch_bounds__(256) void decimate_kernel(const float* __restrict__ src, int W, int f, int h, int w, float* __restrict__ dst)
{
    const long long n = (long long)h * w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i - (long long)y * w);
        dst[i] = src[(long long)(y * f) * W + (long long)x * f];
    }
}

// ---- RNG fills ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void normal_fill_kernel(RngKey key, long long n, float a, float b, const float* __restrict__ base,
                                                          float* __restrict__ out)
{
    key = key_now(key);
    const long long nblk = (n + 3) >> 2;
    for (long long blk = (long long)blockIdx.x * 256 + threadIdx.x; blk < nblk; blk += (long long)gridDim.x * 256) {
        float z[4]; spec_normal4(key, (uint32_t)blk, z);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const long long j = blk * 4 + l;
            if (j < n) out[j] = (base ? base[j] : a) + b * z[l];
        }
    }
}
__global__ __launch_bounds__(256) void uniform_fill_kernel(RngKey key, long long n, float scale, float* __restrict__ out, float lo = 0.f)
{
    key = key_now(key);
    const long long nblk = (n + 3) >> 2;
    for (long long blk = (long long)blockIdx.x * 256 + threadIdx.x; blk < nblk; blk += (long long)gridDim.x * 256) {
        uint32_t r[4]; philox4x32_10((uint32_t)blk, key.stream, key.sample, key.step, key.k0, key.k1, r);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const long long j = blk * 4 + l;
            if (j < n) out[j] = lo + scale * ((float)(r[l] >> 8) * 5.9604644775390625e-08f);
        }
    }
}

// ---- bookkeeping -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sq_err_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                     double* __restrict__ out)
{
    __shared__ double s_red[8];
    double acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = a[i] - b[i]; acc += (double)(d * d);
    }
    block_atomic_add(acc, out, s_red);
}

struct SsimWin { float g[11]; };

__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ a, const float* __restrict__ b, int H, int W,
                                                   SsimWin win, double* __restrict__ out)
{
    __shared__ double s_red[8];
    const long long n = (long long)H * W;
    double acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i - (long long)y * W);
        float m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
        for (int dy = 0; dy < 11; ++dy) {
            const int yy = y + dy - 5; if (yy < 0 || yy >= H) continue;
            for (int dx = 0; dx < 11; ++dx) {
                const int xx = x + dx - 5; if (xx < 0 || xx >= W) continue;
                const float wv = win.g[dy] * win.g[dx], p = a[(long long)yy * W + xx], q = b[(long long)yy * W + xx];
                m1 += wv * p; m2 += wv * q; s11 += wv * p * p; s22 += wv * q * q; s12 += wv * p * q;
            }
        }
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float v1 = s11 - m1 * m1, v2 = s22 - m2 * m2, v12 = s12 - m1 * m2;
        acc += (double)(((2.f * m1 * m2 + C1) * (2.f * v12 + C2)) / ((m1 * m1 + m2 * m2 + C1) * (v1 + v2 + C2)));
    }
    block_atomic_add(acc, out, s_red);
}

__global__ __launch_bounds__(256) void post_step_kernel(float* __restrict__ out, int n, int C, long long HW, float* __restrict__ ema,
                                                        float w, int first)
{
    // out[:, 1:] = exp(-out[:, 1:]) (bayesian_optimization.py:1374-1375); EMA over sample 0 (:1378-1381)
    const long long per = (long long)C * HW, total = (long long)n * per;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i % per; const int c = (int)(r / HW);
        float v = out[i];
        if (c >= 1) { v = expf(-v); out[i] = v; }
        if (ema && i < per) ema[i] = first ? v : ema[i] * w + v * (1.f - w);
    }
}

// Per-iteration bookkeeping of the runner loop (bayesian_optimization.py:1374-1396), one pass over H*W:
//   m = mean_k out[k][0];  a = mean_k exp(-out[k][1])          (the K-sample generalisation of `out`; K = 1: the reference)
//   ema = first ? (m, a) : ema*w + (m, a)*(1-w);  clipped copies for PSNR/SSIM;  ring-buffer slot writes
__global__ __launch_bounds__(256) void bookkeep_kernel(const float* __restrict__ out, int n, int C, long long HW, float* __restrict__ ema,
                                                       float w, int first, float* __restrict__ out_clip, float* __restrict__ ale_clip,
                                                       float* __restrict__ avg_clip, float* __restrict__ ring_epi, float* __restrict__ ring_ale)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)gridDim.x * 256) {
        float m = 0.f, a = 0.f;
        for (int k = 0; k < n; ++k) {
            m += out[(long long)k * C * HW + i];
            if (C > 1) a += expf(-out[(long long)k * C * HW + HW + i]);
        }
        m /= (float)n; a /= (float)n;
        const float e0 = first ? m : ema[i] * w + m * (1.f - w);
        ema[i] = e0;
        if (C > 1) ema[HW + i] = first ? a : ema[HW + i] * w + a * (1.f - w);
        const float mc = fminf(fmaxf(m, 0.f), 1.f), ac = fminf(fmaxf(a, 0.f), 1.f);
        out_clip[i] = mc; avg_clip[i] = fminf(fmaxf(e0, 0.f), 1.f);
        if (ring_epi) ring_epi[i] = mc;
        if (C > 1) { ale_clip[i] = ac; if (ring_ale) ring_ale[i] = ac; }
    }
}


// inpainting runner bookkeeping (bayesian_optimization.py:3039-3064): colour channels through the sigmoid, masked copies for the metrics
__global__ __launch_bounds__(256) void bookkeep_inp_kernel(const float* __restrict__ out, int n, long long HW, const float* __restrict__ img,
                                                           const float* __restrict__ mask, int mask_channels, float* __restrict__ ema, float w,
                                                           int first, float* __restrict__ out_clip, float* __restrict__ ale_clip,
                                                           float* __restrict__ avg_clip, float* __restrict__ img_masked,
                                                           float* __restrict__ out_masked, float* __restrict__ avg_masked,
                                                           float* __restrict__ ring_epi, float* __restrict__ ring_ale)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)gridDim.x * 256) {
        float a = 0.f;
        for (int k = 0; k < n; ++k) a += expf(-out[((long long)k * 4 + 3) * HW + i]);
        a /= (float)n;
        ema[3 * HW + i] = first ? a : ema[3 * HW + i] * w + a * (1.f - w);
        const float ac = fminf(fmaxf(a, 0.f), 1.f);
        ale_clip[i] = ac; if (ring_ale) ring_ale[i] = ac;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float m = 0.f;
            for (int k = 0; k < n; ++k) m += sigmoid_f(out[((long long)k * 4 + c) * HW + i]);
            m /= (float)n;
            const float e0 = first ? m : ema[c * HW + i] * w + m * (1.f - w);
            ema[c * HW + i] = e0;
            const float mc = fminf(fmaxf(m, 0.f), 1.f), ec = fminf(fmaxf(e0, 0.f), 1.f);
            const float mk = mask[(mask_channels == 3 ? c : 0) * HW + i];
            out_clip[c * HW + i] = mc; avg_clip[c * HW + i] = ec;
            if (ring_epi) ring_epi[c * HW + i] = mc;
            img_masked[c * HW + i] = img[c * HW + i] * mk; out_masked[c * HW + i] = mc * mk; avg_masked[c * HW + i] = ec * mk;
        }
    }
}

// torch.var(ring, dim=0) (unbiased) and torch.mean(ring, dim=0) over the R-slot ring buffers (bayesian_optimization.py:1412-1413)
__global__ __launch_bounds__(256) void ring_stats_kernel(const float* __restrict__ ring, int R, long long HW, float* __restrict__ var_out,
                                                         float* __restrict__ mean_out)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)gridDim.x * 256) {
        double s = 0, q = 0;
        for (int r = 0; r < R; ++r) { const double v = ring[(long long)r * HW + i]; s += v; q += v * v; }
        const double mean = s / R;
        if (mean_out) mean_out[i] = (float)mean;
        if (var_out) var_out[i] = R > 1 ? (float)((q - R * mean * mean) / (R - 1)) : 0.f;
    }
}

inline int nblocks(long long n, int cap = 2048) { long long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > cap ? cap : b)); }
inline RngKey make_key(uint64_t seed, uint32_t domain, uint32_t stream, uint32_t sample, uint32_t step)
{
    RngKey k; k.k0 = (uint32_t)seed; k.k1 = (uint32_t)(seed >> 32); k.stream = (domain << 24) | stream; k.sample = sample; k.step = step; k.step_dev = nullptr;
    return k;
}


// Entry of a per-layer table that owns block `blk` (entries sorted by first_block).  The first_block column is staged in LDS with ONE
// round of global loads: walking the table in global memory is a chain of up to n_entries dependent loads per block (~1 us each
// under load), which dominated the table-driven kernels.  Contains a __syncthreads(): call from all threads of the block.
constexpr int TABLE_LDS = 128;
template <typename Entry>
__device__ __forceinline__ int find_entry(const Entry* __restrict__ table, int n_entries, int blk, int* s_first)
{
    const int n_lds = min(n_entries, TABLE_LDS);
    for (int i = threadIdx.x; i < n_lds; i += blockDim.x) s_first[i] = table[i].first_block;
    __syncthreads();
    int ei = 0;
    while (ei + 1 < n_lds && s_first[ei + 1] <= blk) ++ei;
    while (ei + 1 < n_entries && ei + 1 >= n_lds && table[ei + 1].first_block <= blk) ++ei;
    return ei;
}

// ---------------------------------------------------------------------------------------------
// grad_finalize: reduce the per-(strip, sample) partial dW slabs written by the MFMA backward-weight kernels of ALL layers
// of one backward pass and apply the reparameterisation chain rule (reparam_layers.py:26-37 under autograd):
//   d mu[j] += sum_{s,k} P[s,k,j]          d rho[j] += sigmoid(rho[j]) * sum_k eps_k[j] * sum_s P[s,k,j]
// eps_k[j] comes from the sampled-weight slab of this pass when it is still resident (wsamp != nullptr):
//   eps_k * softplus(rho) = W_k - mu, one float4 load instead of a Philox + 2 Box-Muller evaluation per quad and sample;
// otherwise it is re-derived from the counter RNG (same key as the forward draw).  A block owns GRAD_FIN_QUADS quads of 4
// consecutive weights of one layer; its 4 waves split the samples and are summed through LDS.
template <bool BF16>
__global__ __launch_bounds__(256) void grad_finalize_kernel(const GradFinEntry* __restrict__ table, int n_entries,
                                                            const float* __restrict__ part_base, const void* __restrict__ rho_v,
                                                            RngKey key, int sample_weights, int n_samples,
                                                            float* __restrict__ dmu, float* __restrict__ drho,
                                                            const float* __restrict__ wsamp, long long wstride, const void* __restrict__ mu_v,
                                                            int n_main_blocks, const BnGradEntry* __restrict__ bn_table, const double* __restrict__ bsums_base,
                                                            float* __restrict__ dbn)
{
    // blocks behind the weight-gradient blocks: the BatchNorm parameter gradients, one table entry each (round 4: bn_param_grads_kernel was a
    // dependent launch of its own at the tail of every iteration)
    if ((int)blockIdx.x >= n_main_blocks) { bn_param_grads_entry(bn_table[(int)blockIdx.x - n_main_blocks], bsums_base, n_samples, dbn); return; }
    key = key_now(key);
    typedef typename std::conditional<BF16, bf16_t, float>::type PT;
    const PT* __restrict__ rho = static_cast<const PT*>(rho_v); const PT* __restrict__ mu = static_cast<const PT*>(mu_v);
    auto ld = [](const PT* q) -> float { if constexpr (BF16) return bf16_to_f32(*q); else return *q; };
    __shared__ float s_mu[3][GRAD_FIN_QUADS][4], s_rh[3][GRAD_FIN_QUADS][4];
    __shared__ int s_first[TABLE_LDS];
    const GradFinEntry e = table[find_entry(table, n_entries, (int)blockIdx.x, s_first)];
    const int t = threadIdx.x, ql = t & (GRAD_FIN_QUADS - 1), wv = t / GRAD_FIN_QUADS;
    const int item = ((int)blockIdx.x - e.first_block) * GRAD_FIN_QUADS + ql;
    const int nq_w = e.n_w >> 2, nq_b = (e.n_b + 3) >> 2;
    const bool is_w = item < nq_w, is_b = !is_w && item < nq_w + nq_b;
    const int quad = is_w ? item : item - nq_w;                    // quad index inside the weight / bias tensor
    const long long col = is_w ? 4LL * quad : (long long)e.n_w + 4LL * quad;
    const int nv = is_w ? 4 : min(4, e.n_b - 4 * quad);            // valid elements of the quad
    float am[4] = {0.f, 0.f, 0.f, 0.f}, ar[4] = {0.f, 0.f, 0.f, 0.f};
    const bool from_slab = sample_weights && wsamp != nullptr;
    const long long jq = (is_w ? e.w_off : e.b_off) + 4LL * quad;      // first parameter of the quad inside MU / RHO
    float mq[4] = {0.f, 0.f, 0.f, 0.f};
    if (from_slab && (is_w || is_b))
        for (int l = 0; l < nv; ++l) mq[l] = ld(mu + jq + l);
    if (is_w || is_b) {
        const float* __restrict__ P = part_base + e.part_off + col;
        // the wave's samples in batches of 4, fully unrolled: the loads of a batch are independent, so up to 16 float4 are in
        // flight per lane (one load per sample iteration left the kernel latency-bound at ~1 TB/s)
        for (int kb = wv; kb < n_samples; kb += 16)
#pragma unroll
        for (int ku = 0; ku < 4; ++ku) {
            const int k = kb + 4 * ku;
            if (k >= n_samples) break;
            float sk[4] = {0.f, 0.f, 0.f, 0.f};
            const float* __restrict__ q0 = P + (long long)k * e.stride;
            const long long sstep = (long long)n_samples * e.stride;          // next pixel strip of the same sample
            // weights and biases alike: the bias columns start at n_w (a multiple of 4) and the slab row is padded to a multiple of 4,
            // so a bias quad is one aligned float4 too.  (A scalar loop over strips x elements for the bias quads was a chain of up to
            // 4 x 64 x 4 dependent loads in a handful of lanes and set the duration of the whole launch: 120 us.)
            {
                int sidx = 0;
                for (; sidx + 4 <= e.strips; sidx += 4) {                     // 4 independent loads in flight
                    const float4 v0 = *reinterpret_cast<const float4*>(q0 + (sidx + 0) * sstep), v1 = *reinterpret_cast<const float4*>(q0 + (sidx + 1) * sstep);
                    const float4 v2 = *reinterpret_cast<const float4*>(q0 + (sidx + 2) * sstep), v3 = *reinterpret_cast<const float4*>(q0 + (sidx + 3) * sstep);
                    sk[0] += (v0.x + v1.x) + (v2.x + v3.x); sk[1] += (v0.y + v1.y) + (v2.y + v3.y);
                    sk[2] += (v0.z + v1.z) + (v2.z + v3.z); sk[3] += (v0.w + v1.w) + (v2.w + v3.w);
                }
                for (; sidx < e.strips; ++sidx) {
                    const float4 v = *reinterpret_cast<const float4*>(q0 + sidx * sstep);
                    sk[0] += v.x; sk[1] += v.y; sk[2] += v.z; sk[3] += v.w;
                }
#pragma unroll
                for (int l = 0; l < 4; ++l) if (l >= nv) sk[l] = 0.f;         // padding columns of the last bias quad are never written
            }
#pragma unroll
            for (int l = 0; l < 4; ++l) am[l] += sk[l];
            if (from_slab) {
                const float* __restrict__ wk = wsamp + (long long)k * wstride + jq;
                if (is_w) {          // w_off % 4 == 0 for every layer of the slab: aligned float4
                    const float4 w = *reinterpret_cast<const float4*>(wk);
                    ar[0] = __builtin_fmaf(sk[0], w.x - mq[0], ar[0]); ar[1] = __builtin_fmaf(sk[1], w.y - mq[1], ar[1]);
                    ar[2] = __builtin_fmaf(sk[2], w.z - mq[2], ar[2]); ar[3] = __builtin_fmaf(sk[3], w.w - mq[3], ar[3]);
                } else
                    for (int l = 0; l < nv; ++l) ar[l] = __builtin_fmaf(sk[l], wk[l] - mq[l], ar[l]);
            } else if (sample_weights) {
                RngKey kw = key; kw.sample += (uint32_t)k;
                kw.stream = ((uint32_t)DOMAIN_EPS << 24) | (uint32_t)(2 * e.layer_id + (is_w ? 0 : 1));
                float z[4]; spec_normal4(kw, (uint32_t)quad, z);
#pragma unroll
                for (int l = 0; l < 4; ++l) ar[l] = __builtin_fmaf(sk[l], z[l], ar[l]);
            }
        }
    }
    if (wv > 0) {
#pragma unroll
        for (int l = 0; l < 4; ++l) { s_mu[wv - 1][ql][l] = am[l]; s_rh[wv - 1][ql][l] = ar[l]; }
    }
    __syncthreads();
    if (wv == 0 && (is_w || is_b)) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int l = 0; l < 4; ++l) { am[l] += s_mu[w][ql][l]; ar[l] += s_rh[w][ql][l]; }
        const long long j0 = (is_w ? e.w_off : e.b_off) + 4LL * quad;
        for (int l = 0; l < nv; ++l) {
            dmu[j0 + l] += am[l];
            if (from_slab) {         // ar = sum_k dW_k * (W_k - mu) = softplus(rho) * sum_k dW_k * eps_k, with the softplus the draw used
                const float r = ld(rho + j0 + l), sp = is_w ? softplus_fast(r) : softplus_f(r);
                drho[j0 + l] += ar[l] / sp * sigmoid_f(r);
            } else if (sample_weights) drho[j0 + l] += ar[l] * sigmoid_f(ld(rho + j0 + l));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// sample_weights: the reparameterisation draw of one pass, once per (layer, MC sample) instead of once per conv block:
//   W_k[j] = mu[j] + softplus(rho[j]) * eps_k[j]       (BayTorch reparam_layers.py:26-37, modules/module.py:64-80)
// eps_k from the counter RNG (domain EPS, stream 2*layer [+1 bias], sample k0+k, step); weights use the hardware-exp
// softplus_fast, biases the libm-grade softplus_f — the same split the conv kernels used when they sampled in place.
// grid: x = blocks of SAMPLE_QUADS quads over all layers of the table, y = sample.
constexpr int SAMPLE_KPT = 4;      // samples per thread of the draw
template <bool BF16>
__global__ __launch_bounds__(256) void sample_weights_kernel(const SampleEntry* __restrict__ table, int n_entries,
                                                             const void* __restrict__ mu_v, const void* __restrict__ rho_v,
                                                             RngKey key, float* __restrict__ wsamp, long long wstride, int sample,
                                                             double* __restrict__ zero, long long n_zero, int n_k)
{
    key = key_now(key);
    // the pass's statistics buffers, cleared by the draw's own threads (plan.hip, mfvi_forward)
    for (long long i = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < n_zero; i += (long long)gridDim.x * gridDim.y * 256) zero[i] = 0.0;
    typedef typename std::conditional<BF16, bf16_t, float>::type PT;
    const PT* __restrict__ mu = static_cast<const PT*>(mu_v); const PT* __restrict__ rho = static_cast<const PT*>(rho_v);
    __shared__ int s_first[TABLE_LDS];
    const SampleEntry e = table[find_entry(table, n_entries, (int)blockIdx.x, s_first)];
    // a thread draws its quad for SAMPLE_KPT consecutive samples (grid y = groups of samples): mu, rho and softplus(rho) — a fifth of the
    // work per (quad, sample) — once per thread instead of once per sample (round 4; the draws themselves are per (sample, quad) as before)
    const int k0 = (int)blockIdx.y * SAMPLE_KPT, k1 = min(k0 + SAMPLE_KPT, n_k);
    const int item = ((int)blockIdx.x - e.first_block) * SAMPLE_QUADS + (int)threadIdx.x;
    const int nq_w = e.n_w >> 2, nq_b = (e.n_b + 3) >> 2;
    if (item >= nq_w + nq_b) return;
    if (item < nq_w) {
        const long long j = e.w_off + 4LL * item;
        float4 m, r;
        if constexpr (BF16) { m = bf16x4_to_f32(mu + j); r = bf16x4_to_f32(rho + j); }
        else { m = *reinterpret_cast<const float4*>(mu + j); r = *reinterpret_cast<const float4*>(rho + j); }
        float sp[4] = {0.f, 0.f, 0.f, 0.f};
        if (sample) { sp[0] = softplus_fast(r.x); sp[1] = softplus_fast(r.y); sp[2] = softplus_fast(r.z); sp[3] = softplus_fast(r.w); }
        for (int k = k0; k < k1; ++k) {
            RngKey kw = key; kw.sample += (uint32_t)k;
            kw.stream = ((uint32_t)DOMAIN_EPS << 24) | (uint32_t)(2 * e.layer_id);
            float z[4] = {0.f, 0.f, 0.f, 0.f};
            if (sample) spec_normal4(kw, (uint32_t)item, z);
            float4 w = m;
            if (sample) { w.x = __builtin_fmaf(sp[0], z[0], m.x); w.y = __builtin_fmaf(sp[1], z[1], m.y); w.z = __builtin_fmaf(sp[2], z[2], m.z); w.w = __builtin_fmaf(sp[3], z[3], m.w); }
            *reinterpret_cast<float4*>(wsamp + (long long)k * wstride + j) = w;
        }
    } else {
        const int q = item - nq_w;
        for (int k = k0; k < k1; ++k) {
            RngKey kw = key; kw.sample += (uint32_t)k;
            kw.stream = ((uint32_t)DOMAIN_EPS << 24) | (uint32_t)(2 * e.layer_id + 1);
            float z[4] = {0.f, 0.f, 0.f, 0.f};
            if (sample) spec_normal4(kw, (uint32_t)q, z);
            float* __restrict__ o = wsamp + (long long)k * wstride;
            for (int l = 0; l < 4 && 4 * q + l < e.n_b; ++l) {
                const long long j = e.b_off + 4 * q + l;
                float mj, rj;
                if constexpr (BF16) { mj = bf16_to_f32(mu[j]); rj = bf16_to_f32(rho[j]); } else { mj = mu[j]; rj = rho[j]; }
                o[j] = sample ? mj + softplus_f(rj) * z[l] : mj;
            }
        }
    }
}

__global__ __launch_bounds__(256) void expand_bf16_kernel(const bf16_t* __restrict__ src, long long n, float* __restrict__ dst)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = bf16_to_f32(src[i]);
}
__global__ __launch_bounds__(256) void round_bf16_kernel(const float* __restrict__ src, long long n, bf16_t* __restrict__ dst)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = f32_to_bf16_rne(src[i]);
}

// ---- fused tail of an ELBO iteration on bf16 mu / rho (BASELINE configs[4]) ----
// Same arithmetic as elbo_update_kernel on the float32 values the bf16 parameters denote: KL terms in fp32, summed in fp64; gradients,
// both Adam moments and the update in fp32; the new mu / rho go back to bf16 by STOCHASTIC rounding with a 16-bit word of the counter
// RNG: domain ROUND, stream 0 (MU) / 1 (RHO), element j = lane j & 3 of block j >> 2, sample 0, step = t (upper 16 bits of the word).
// The BatchNorm parameters stay float32.
__global__ __launch_bounds__(256) void elbo_update_bf16_kernel(bf16_t* __restrict__ pm, bf16_t* __restrict__ pr, float* __restrict__ bn, float* __restrict__ g, float* __restrict__ m,
                                                               float* __restrict__ v, long long n_vi, long long n_bn, float m0, float s0, float temp,
                                                               float b1, float b2, float eps, float step_size, float inv_sqrt_bc2, RngKey key,
                                                               ElboUpdateScratch* __restrict__ sc)
{
    key = key_now(key);
    __shared__ double s_red[8];
    const float log_s0 = logf(s0), s0sq = s0 * s0;
    auto adam = [&](long long i, float gi, float pi) -> float {
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        return pi - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    };
    double acc = 0;
    const long long nq = (n_vi + 3) >> 2;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long long)gridDim.x * 256) {
        uint32_t rm[4], rr[4];
        RngKey k0 = key; k0.stream = ((uint32_t)DOMAIN_ROUND << 24) | 0u;
        philox4x32_10((uint32_t)q, k0.stream, k0.sample, k0.step, k0.k0, k0.k1, rm);
        philox4x32_10((uint32_t)q, k0.stream | 1u, k0.sample, k0.step, k0.k0, k0.k1, rr);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const long long i = 4 * q + l;
            if (i >= n_vi) break;
            const float mu = bf16_to_f32(pm[i]), r = bf16_to_f32(pr[i]);
            const float s = softplus_f(r), d = mu - m0, inv = 1.f / s;
            acc += (double)(logf(s) - log_s0) + (double)((s0sq + d * d) / (2.f * s * s)) - 0.5;
            const float gmu = g[i] + temp * d * inv * inv;
            const float grho = g[n_vi + i] + temp * (inv - (s0sq + d * d) * inv * inv * inv) * sigmoid_f(r);
            g[i] = gmu; g[n_vi + i] = grho;
            pm[i] = f32_to_bf16_sr(adam(i, gmu, mu), rm[l] >> 16);
            pr[i] = f32_to_bf16_sr(adam(n_vi + i, grho, r), rr[l] >> 16);
        }
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_bn; i += (long long)gridDim.x * 256) bn[i] = adam(2 * n_vi + i, g[2 * n_vi + i], bn[i]);
    const double tot = block_sum_d(acc, s_red);
    if (threadIdx.x == 0) sc->partial[blockIdx.x] = tot;
}

}  // namespace

int launch_dropout_masks(const DropEntry* table_dev, int n_entries, RngKey key, int n_samples, float* arena, hipStream_t st)
{
    if (n_entries <= 0) return 0;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(n_entries, n_samples), dim3(64), 0, st, table_dev, key, arena);
    return (int)hipGetLastError();
}

int launch_sample_weights(const SampleEntry* table_dev, int n_entries, int n_blocks, const void* mu, const void* rho, RngKey key,
                          int n_samples, float* wsamp, long long wstride, hipStream_t st, int bf16, int sample, double* zero, long long n_zero)
{
    if (n_entries < 1 || n_blocks < 1) { if (n_zero > 0) return (int)hipMemsetAsync(zero, 0, sizeof(double) * n_zero, st); return 0; }
    const dim3 grid(n_blocks, (n_samples + SAMPLE_KPT - 1) / SAMPLE_KPT);
    if (bf16) hipLaunchKernelGGL(sample_weights_kernel<true>, grid, dim3(256), 0, st, table_dev, n_entries, mu, rho, key, wsamp, wstride, sample, zero, n_zero, n_samples);
    else hipLaunchKernelGGL(sample_weights_kernel<false>, grid, dim3(256), 0, st, table_dev, n_entries, mu, rho, key, wsamp, wstride, sample, zero, n_zero, n_samples);
    return (int)hipGetLastError();
}

int launch_expand_bf16(const void* src, long long n, float* dst, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(expand_bf16_kernel, dim3(nblocks(n)), dim3(256), 0, st, (const bf16_t*)src, n, dst);
    return (int)hipGetLastError();
}

int launch_grad_finalize(const GradFinEntry* table_dev, int n_entries, int n_blocks, const float* part_base, const void* rho, RngKey key,
                         int sample_weights, int n_samples, float* dmu, float* drho, const float* wsamp, long long wstride, const void* mu,
                         hipStream_t st, int bf16, const BnGradEntry* bn_table, int n_bn, const double* bsums_base, float* dbn)
{
    if (n_entries < 1 || n_blocks < 1) return 0;      // (the caller launches bn_param_grads itself when there is nothing to reduce)
    if (!bn_table || !dbn) n_bn = 0;
    if (bf16) hipLaunchKernelGGL(grad_finalize_kernel<true>, dim3(n_blocks + n_bn), dim3(256), 0, st, table_dev, n_entries, part_base, rho, key, sample_weights,
                                 n_samples, dmu, drho, wsamp, wstride, mu, n_blocks, bn_table, bsums_base, dbn);
    else hipLaunchKernelGGL(grad_finalize_kernel<false>, dim3(n_blocks + n_bn), dim3(256), 0, st, table_dev, n_entries, part_base, rho, key, sample_weights,
                            n_samples, dmu, drho, wsamp, wstride, mu, n_blocks, bn_table, bsums_base, dbn);
    return (int)hipGetLastError();
}

extern "C" {

int mfvi_gaussian_nll(const float* out, const float* target, int n, int H, int W, int factor, float grad_scale, float* dout,
                      double* nll_sum, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 1 || factor < 1 || H % factor || W % factor) { set_error("gaussian_nll: bad shape n=%d H=%d W=%d factor=%d", n, H, W, factor); return -1; }
    if (dout && factor > 1) { hipError_t e = hipMemsetAsync(dout, 0, sizeof(float) * (size_t)n * 2 * H * W, st); if (e) return (int)e; }
    const long long npix = (long long)(H / factor) * (W / factor);
    // few blocks per sample: every block ends in one fp64 atomic on the SAME address, and those serialise (~0.2 us each)
    if (factor == 1 && (W & 3) == 0 && !(((uintptr_t)out | (uintptr_t)target | (uintptr_t)dout) & 15)) {
        hipLaunchKernelGGL(gaussian_nll_vec_kernel, dim3(nblocks(npix / 4, 16), n), dim3(256), 0, st, out, target, npix, grad_scale, dout, nll_sum);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(gaussian_nll_kernel, dim3(nblocks(npix, 16), n), dim3(256), 0, st, out, target, H, W, factor, grad_scale, dout, nll_sum);
    return (int)hipGetLastError();
}

int mfvi_gaussian_nll_inpainting(const float* out, const float* target, const float* mask, int mask_channels, int n, int H, int W,
                                 float grad_scale, float* dout, double* nll_sum, void* stream)
{
    if (!out || !target || !mask || !nll_sum || n < 1 || H < 1 || W < 1 || (mask_channels != 1 && mask_channels != 3)) {
        set_error("gaussian_nll_inpainting: bad arguments (n=%d H=%d W=%d mask_channels=%d)", n, H, W, mask_channels); return -1; }
    const long long HW = (long long)H * W;
    hipLaunchKernelGGL(gaussian_nll_inp_kernel, dim3(nblocks(HW, 16), n), dim3(256), 0, (hipStream_t)stream, out, target, mask, mask_channels, HW,
                       grad_scale, dout, nll_sum);
    return (int)hipGetLastError();
}


int mfvi_gaussian_nll_tensors(const float* mu, const float* neg_logvar, const float* target, const float* mask, int C, int Cs, int Cm, int64_t HW,
                              int reduction_mean, double* loss_out, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!mu || !neg_logvar || !target || !loss_out || C < 1 || HW < 1 || (Cs != 1 && Cs != C) || (mask && Cm != 1 && Cm != C)) {
        set_error("gaussian_nll_tensors: bad arguments (C=%d Cs=%d Cm=%d HW=%lld)", C, Cs, Cm, (long long)HW); return -1; }
    hipError_t e = hipMemsetAsync(loss_out, 0, sizeof(double), st); if (e) return (int)e;
    hipLaunchKernelGGL(gnll_bcast_fwd_kernel, dim3(nblocks(HW, 16)), dim3(256), 0, st, mu, neg_logvar, target, mask, C, Cs, Cm, (long long)HW,
                       reduction_mean ? 1.0 / ((double)C * (double)HW) : 1.0, loss_out);
    return (int)hipGetLastError();
}

int mfvi_gaussian_nll_tensors_backward(const float* mu, const float* neg_logvar, const float* target, const float* mask, int C, int Cs, int Cm,
                                       int64_t HW, int reduction_mean, const float* grad_out, float* dmu, float* dneg_logvar, void* stream)
{
    if (!mu || !neg_logvar || !target || !grad_out || !dmu || !dneg_logvar || C < 1 || HW < 1 || (Cs != 1 && Cs != C) || (mask && Cm != 1 && Cm != C)) {
        set_error("gaussian_nll_tensors_backward: bad arguments"); return -1; }
    hipLaunchKernelGGL(gnll_bcast_bwd_kernel, dim3(nblocks(HW)), dim3(256), 0, (hipStream_t)stream, mu, neg_logvar, target, mask, C, Cs, Cm, (long long)HW,
                       reduction_mean ? (float)(1.0 / ((double)C * (double)HW)) : 1.f, grad_out, dmu, dneg_logvar);
    return (int)hipGetLastError();
}

int mfvi_kl(const float* mu, const float* rho, int64_t n, float prior_mu, float prior_sigma, double* kl_out, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || !(prior_sigma > 0.f)) { set_error("kl: bad arguments"); return -1; }
    hipError_t e = hipMemsetAsync(kl_out, 0, sizeof(double), st); if (e) return (int)e;
    if (n == 0) return 0;
    hipLaunchKernelGGL(kl_kernel, dim3(nblocks(n, 96)), dim3(256), 0, st, mu, rho, (long long)n, prior_mu, prior_sigma, kl_out);
    return (int)hipGetLastError();
}

int mfvi_kl_backward(const float* mu, const float* rho, int64_t n, float prior_mu, float prior_sigma, float scale, float* dmu,
                     float* drho, void* stream)
{
    if (n < 0 || !(prior_sigma > 0.f)) { set_error("kl_backward: bad arguments"); return -1; }
    if (n == 0) return 0;
    hipLaunchKernelGGL(kl_bwd_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, mu, rho, (long long)n, prior_mu, prior_sigma, scale, dmu, drho);
    return (int)hipGetLastError();
}

int mfvi_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int t,
                   void* stream)
{
    if (n < 0 || t < 1) { set_error("adam_step: bad arguments (t is 1-based)"); return -1; }
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
    hipLaunchKernelGGL(adam_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, beta1, beta2, eps,
                       (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), 1.0f, lr, StepGuard{nullptr, nullptr, nullptr});
    return (int)hipGetLastError();
}

int64_t mfvi_elbo_update_scratch_bytes(void) { return (int64_t)sizeof(ElboUpdateScratch); }

int mfvi_elbo_update(float* params, float* grads, float* m, float* v, int64_t n_vi, int64_t n_bn, float prior_mu, float prior_sigma, float temp,
                     float lr, float beta1, float beta2, float eps, int t, double* kl_out, void* scratch, void* stream)
{
    if (!params || !grads || !m || !v || !kl_out || !scratch || n_vi < 0 || n_bn < 0 || t < 1 || !(prior_sigma > 0.f)) {
        set_error("elbo_update: bad arguments (t is 1-based, scratch of mfvi_elbo_update_scratch_bytes() bytes)"); return -1; }
    const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
    const long long work = n_vi > n_bn ? n_vi : n_bn;
    const int nb = nblocks(work, ELBO_UPDATE_MAX_BLOCKS);
    hipLaunchKernelGGL(elbo_update_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, (long long)n_vi, (long long)n_bn,
                       prior_mu, prior_sigma, temp, beta1, beta2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), (ElboUpdateScratch*)scratch, lr,
                       StepGuard{nullptr, nullptr, nullptr});
    hipLaunchKernelGGL(elbo_update_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const ElboUpdateScratch*)scratch, nb, kl_out, (int*)nullptr,
                       StepGuard{nullptr, nullptr, nullptr});
    return (int)hipGetLastError();
}

int mfvi_elbo_update_guarded(float* params, float* grads, float* m, float* v, int64_t n_vi, int64_t n_bn, float prior_mu, float prior_sigma, float temp,
                             float lr, float beta1, float beta2, float eps, int32_t* t_applied, const double* loss_d, const float* loss_f,
                             double* kl_out, void* scratch, void* stream)
{
    if (!params || !grads || !m || !v || !kl_out || !scratch || !t_applied || n_vi < 0 || n_bn < 0 || !(prior_sigma > 0.f)) {
        set_error("elbo_update_guarded: bad arguments (t_applied: device counter of applied updates)"); return -1; }
    const long long work = n_vi > n_bn ? n_vi : n_bn;
    const int nb = nblocks(work, ELBO_UPDATE_MAX_BLOCKS);
    const StepGuard guard{t_applied, loss_d, loss_f};
    hipLaunchKernelGGL(elbo_update_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, (long long)n_vi, (long long)n_bn,
                       prior_mu, prior_sigma, temp, beta1, beta2, eps, 0.f, 0.f, (ElboUpdateScratch*)scratch, lr, guard);
    hipLaunchKernelGGL(elbo_update_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const ElboUpdateScratch*)scratch, nb, kl_out, t_applied, guard);
    return (int)hipGetLastError();
}

int mfvi_adamw_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                            const int32_t* t_applied, const double* loss_d, const float* loss_f, float weight_decay, void* stream)
{
    if (n < 0 || !t_applied || weight_decay < 0.f) { set_error("adamw_step_guarded: bad arguments"); return -1; }
    if (n == 0) return 0;
    hipLaunchKernelGGL(adam_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, beta1, beta2, eps, 0.f, 0.f,
                       1.0f - lr * weight_decay, lr, StepGuard{t_applied, loss_d, loss_f});
    return (int)hipGetLastError();
}

int mfvi_step_advance(int32_t* t_applied, const double* loss_d, const float* loss_f, void* stream)
{
    if (!t_applied) { set_error("step_advance: null counter"); return -1; }
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t_applied, StepGuard{t_applied, loss_d, loss_f});
    return (int)hipGetLastError();
}

int mfvi_decimate(const float* src, int H, int W, int factor, float* dst, void* stream)
{
    if (!src || !dst || factor < 1 || H < factor || W < factor) { set_error("decimate: bad arguments"); return -1; }
    const int h = H / factor, w = W / factor;
    hipLaunchKernelGGL(decimate_kernel, dim3(nblocks((long long)h * w)), dim3(256), 0, (hipStream_t)stream, src, W, factor, h, w, dst);
    return (int)hipGetLastError();
}

int mfvi_elbo_update_bf16(void* mu_bf16, void* rho_bf16, float* bn, float* grads, float* m, float* v, int64_t n_vi, int64_t n_bn, float prior_mu,
                          float prior_sigma, float temp, float lr, float beta1, float beta2, float eps, int t, uint64_t seed, double* kl_out, void* scratch,
                          void* stream)
{
    if (!mu_bf16 || !rho_bf16 || !grads || !m || !v || !kl_out || !scratch || n_vi < 0 || n_bn < 0 || (n_bn > 0 && !bn) || t < 1 || !(prior_sigma > 0.f)) {
        set_error("elbo_update_bf16: bad arguments (t is 1-based)"); return -1; }
    const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
    const long long work = std::max<long long>((n_vi + 3) / 4, n_bn);
    const int nb = nblocks(work, ELBO_UPDATE_MAX_BLOCKS);
    hipLaunchKernelGGL(elbo_update_bf16_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (bf16_t*)mu_bf16, (bf16_t*)rho_bf16, bn, grads, m, v, (long long)n_vi,
                       (long long)n_bn, prior_mu, prior_sigma, temp, beta1, beta2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)),
                       make_key(seed, DOMAIN_ROUND, 0, 0, (uint32_t)t), (ElboUpdateScratch*)scratch);
    hipLaunchKernelGGL(elbo_update_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const ElboUpdateScratch*)scratch, nb, kl_out, (int*)nullptr,
                       StepGuard{nullptr, nullptr, nullptr});
    return (int)hipGetLastError();
}

int mfvi_bf16_to_f32(const void* src, int64_t n, float* dst, void* stream)
{
    if (n < 0 || (n > 0 && (!src || !dst))) { set_error("bf16_to_f32: bad arguments"); return -1; }
    return launch_expand_bf16(src, (long long)n, dst, (hipStream_t)stream);
}

int mfvi_f32_to_bf16(const float* src, int64_t n, void* dst, void* stream)
{
    if (n < 0 || (n > 0 && (!src || !dst))) { set_error("f32_to_bf16: bad arguments"); return -1; }
    if (n == 0) return 0;
    hipLaunchKernelGGL(round_bf16_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, src, (long long)n, (bf16_t*)dst);
    return (int)hipGetLastError();
}

int mfvi_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int t,
                    float weight_decay, void* stream)
{
    if (n < 0 || t < 1 || weight_decay < 0.f) { set_error("adamw_step: bad arguments (t is 1-based)"); return -1; }
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
    hipLaunchKernelGGL(adam_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, beta1, beta2, eps,
                       (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), 1.0f - lr * weight_decay, lr, StepGuard{nullptr, nullptr, nullptr});
    return (int)hipGetLastError();
}

int mfvi_mse_sigmoid_masked(const float* out, const float* target, const float* mask, int mask_channels, int n, int H, int W,
                            float grad_scale, float* dout, double* mse_sum, void* stream)
{
    if (!out || !target || !mask || !mse_sum || n < 1 || H < 1 || W < 1 || (mask_channels != 1 && mask_channels != 3)) {
        set_error("mse_sigmoid_masked: bad arguments"); return -1; }
    const long long HW = (long long)H * W;
    hipLaunchKernelGGL(mse_sigmoid_masked_kernel, dim3(nblocks(HW, 16), n), dim3(256), 0, (hipStream_t)stream, out, target, mask, mask_channels, HW,
                       grad_scale, dout, mse_sum);
    return (int)hipGetLastError();
}

int mfvi_mse_channel(const float* out, const float* target, int n, int C, int H, int W, int channel, int factor, float grad_scale, float* dout,
                     double* mse_sum, void* stream)
{
    if (!out || !target || !mse_sum || n < 1 || C < 1 || channel < 0 || channel >= C || H < 1 || W < 1 || factor < 1 || H / factor < 1 || W / factor < 1) {
        set_error("mse_channel: bad arguments"); return -1; }
    hipLaunchKernelGGL(mse_channel_kernel, dim3(nblocks((long long)H * W, 16), n), dim3(256), 0, (hipStream_t)stream, out, target, C, H, W, channel, factor,
                       grad_scale, dout, mse_sum);
    return (int)hipGetLastError();
}

int mfvi_uniform_fill_range(uint64_t seed, uint32_t stream_id, uint32_t sample, uint32_t step, int64_t n, float lo, float hi, float* out,
                            void* stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(uniform_fill_kernel, dim3(nblocks((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       make_key(seed, DOMAIN_UNIFORM, stream_id, sample, step), (long long)n, hi - lo, out, lo);
    return (int)hipGetLastError();
}

int mfvi_add_normal(float* x, uint64_t seed, uint32_t stream_id, uint32_t step, int64_t n, float std, void* stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(normal_fill_kernel, dim3(nblocks((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       make_key(seed, DOMAIN_SGLD, stream_id, 0, step), (long long)n, 0.f, std, (const float*)x, x);
    return (int)hipGetLastError();
}

int mfvi_normal_fill(uint64_t seed, uint32_t domain, uint32_t stream_id, uint32_t sample, uint32_t step, int64_t n, float a, float b,
                     float* out, void* stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(normal_fill_kernel, dim3(nblocks((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       make_key(seed, domain, stream_id, sample, step), (long long)n, a, b, (const float*)nullptr, out);
    return (int)hipGetLastError();
}

int mfvi_uniform_fill(uint64_t seed, uint32_t stream_id, uint32_t sample, uint32_t step, int64_t n, float scale, float* out, void* stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(uniform_fill_kernel, dim3(nblocks((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       make_key(seed, DOMAIN_UNIFORM, stream_id, sample, step), (long long)n, scale, out);
    return (int)hipGetLastError();
}

int mfvi_perturb_input(const float* z0, uint64_t seed, uint32_t step, int64_t n, float std, float* z, void* stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(normal_fill_kernel, dim3(nblocks((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       make_key(seed, DOMAIN_INPUT, 0, 0, step), (long long)n, 0.f, std, z0, z);
    return (int)hipGetLastError();
}

int mfvi_perturb_input_dev(const float* z0, uint64_t seed, const int32_t* step_dev, uint32_t step_offset, int64_t n, float std, float* z, void* stream)
{
    if (!step_dev) { set_error("perturb_input_dev: null step counter"); return -1; }
    if (n <= 0) return 0;
    RngKey key = make_key(seed, DOMAIN_INPUT, 0, 0, step_offset); key.step_dev = step_dev;
    hipLaunchKernelGGL(normal_fill_kernel, dim3(nblocks((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, key, (long long)n, 0.f, std, z0, z);
    return (int)hipGetLastError();
}

int mfvi_sq_err_sum(const float* a, const float* b, int64_t n, double* sum_out, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(sum_out, 0, sizeof(double), st); if (e) return (int)e;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sq_err_kernel, dim3(nblocks(n, 32)), dim3(256), 0, st, a, b, (long long)n, sum_out);
    return (int)hipGetLastError();
}

int mfvi_ssim_sum(const float* a, const float* b, int H, int W, double* ssim_sum, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(ssim_sum, 0, sizeof(double), st); if (e) return (int)e;
    SsimWin win; float s = 0.f;
    for (int i = 0; i < 11; ++i) { win.g[i] = expf(-(float)((i - 5) * (i - 5)) / (2.f * 1.5f * 1.5f)); s += win.g[i]; }
    for (int i = 0; i < 11; ++i) win.g[i] /= s;
    hipLaunchKernelGGL(ssim_kernel, dim3(nblocks((long long)H * W, 128)), dim3(256), 0, st, a, b, H, W, win, ssim_sum);
    return (int)hipGetLastError();
}

int mfvi_bookkeep(const float* out, int n, int C, int H, int W, float* ema, float ema_weight, int first, float* out_clip, float* ale_clip,
                  float* avg_clip, float* ring_epi_slot, float* ring_ale_slot, void* stream)
{
    if (n < 1 || C < 1 || C > 2 || !out || !ema || !out_clip || !avg_clip || (C > 1 && !ale_clip)) { set_error("bookkeep: bad arguments"); return -1; }
    const long long HW = (long long)H * W;
    hipLaunchKernelGGL(bookkeep_kernel, dim3(nblocks(HW)), dim3(256), 0, (hipStream_t)stream, out, n, C, HW, ema, ema_weight, first, out_clip,
                       ale_clip, avg_clip, ring_epi_slot, ring_ale_slot);
    return (int)hipGetLastError();
}

int mfvi_bookkeep_inpainting(const float* out, int n, int H, int W, const float* img, const float* mask, int mask_channels, float* ema,
                             float ema_weight, int first, float* out_clip, float* ale_clip, float* avg_clip, float* img_masked,
                             float* out_masked, float* avg_masked, float* ring_epi_slot, float* ring_ale_slot, void* stream)
{
    if (n < 1 || !out || !img || !mask || !ema || !out_clip || !ale_clip || !avg_clip || !img_masked || !out_masked || !avg_masked ||
        (mask_channels != 1 && mask_channels != 3)) { set_error("bookkeep_inpainting: bad arguments"); return -1; }
    const long long HW = (long long)H * W;
    hipLaunchKernelGGL(bookkeep_inp_kernel, dim3(nblocks(HW)), dim3(256), 0, (hipStream_t)stream, out, n, HW, img, mask, mask_channels, ema,
                       ema_weight, first, out_clip, ale_clip, avg_clip, img_masked, out_masked, avg_masked, ring_epi_slot, ring_ale_slot);
    return (int)hipGetLastError();
}

int mfvi_ring_stats(const float* ring, int R, int H, int W, float* var_out, float* mean_out, void* stream)
{
    if (R < 1 || !ring) { set_error("ring_stats: bad arguments"); return -1; }
    const long long HW = (long long)H * W;
    hipLaunchKernelGGL(ring_stats_kernel, dim3(nblocks(HW)), dim3(256), 0, (hipStream_t)stream, ring, R, HW, var_out, mean_out);
    return (int)hipGetLastError();
}

int mfvi_post_step(float* out, int n, int C, int H, int W, float* ema, float ema_weight, int first, void* stream)
{
    const long long total = (long long)n * C * H * W;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(post_step_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, out, n, C, (long long)H * W, ema, ema_weight, first);
    return (int)hipGetLastError();
}

}  // extern "C"
