// K7 — FastRadonTransform (radon/radon.py:23-55) and its adjoint as gather kernels.
// forward : sino[t][j] = sum_i bilinear(img, R_t(i, j))      (affine_grid + grid_sample(zeros) + sum over rows);
//           the 23.6 MB rotation grid of the reference is never built: coordinates are recomputed from theta.
// adjoint : dimg[y][x] = sum_t sum_{(i,j) in 3x3 around R_t^-1(y,x)} dsino[t][j] * (1-|ix-x|)+ * (1-|iy-y|)+
//           (exact transpose of the forward, no atomics: a rotation is an isometry, so only the 3x3 grid
//           neighbours of the back-rotated pixel can touch it).
#include "common.h"
#include "../../include/mfvi_hip.h"

namespace {

struct Rot { double c, s; };

__device__ __forceinline__ Rot rot_of(float theta_deg)
{
    const float th = theta_deg * 0.017453292519943295f;      // torch.deg2rad in fp32 (radon/radon.py:31)
    Rot r; r.c = (double)cosf(th); r.s = (double)sinf(th); return r;
}
__device__ __forceinline__ void src_of(const Rot& r, int H, int W, int i, int j, double& ix, double& iy)
{
    const double xb = (2.0 * j + 1.0) / W - 1.0, yb = (2.0 * i + 1.0) / H - 1.0;
    const double gx = r.c * xb - r.s * yb, gy = r.s * xb + r.c * yb;
    ix = ((gx + 1.0) * W - 1.0) * 0.5; iy = ((gy + 1.0) * H - 1.0) * 0.5;
}

__global__ __launch_bounds__(256) void radon_fwd_kernel(const float* __restrict__ img, const float* __restrict__ theta, int H, int W,
                                                        int T, float* __restrict__ sino)
{
    const int k = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * W) return;
    const int t = idx / W, j = idx - t * W;
    const Rot r = rot_of(theta[t]);
    const float* __restrict__ im = img + (long long)k * H * W;
    double acc = 0;
    for (int i = 0; i < H; ++i) {
        double ix, iy; src_of(r, H, W, i, j, ix, iy);
        const double fx = floor(ix), fy = floor(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        const double lx = ix - fx, ly = iy - fy;
        double v = 0;
        if (y0 >= 0 && y0 < H) {
            if (x0 >= 0 && x0 < W) v += (1 - lx) * (1 - ly) * im[y0 * W + x0];
            if (x0 + 1 >= 0 && x0 + 1 < W) v += lx * (1 - ly) * im[y0 * W + x0 + 1];
        }
        if (y0 + 1 >= 0 && y0 + 1 < H) {
            if (x0 >= 0 && x0 < W) v += (1 - lx) * ly * im[(y0 + 1) * W + x0];
            if (x0 + 1 >= 0 && x0 + 1 < W) v += lx * ly * im[(y0 + 1) * W + x0 + 1];
        }
        acc += v;
    }
    sino[(long long)k * T * W + idx] = (float)acc;
}

__global__ __launch_bounds__(256) void radon_adj_kernel(const float* __restrict__ dsino, const float* __restrict__ theta, int H, int W,
                                                        int T, float* __restrict__ dimg)
{
    const int k = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W;
    const float* __restrict__ ds = dsino + (long long)k * T * W;
    const double gx = (2.0 * x + 1.0) / W - 1.0, gy = (2.0 * y + 1.0) / H - 1.0;
    double acc = 0;
    for (int t = 0; t < T; ++t) {
        const Rot r = rot_of(theta[t]);
        const double xb = r.c * gx + r.s * gy, yb = -r.s * gx + r.c * gy;
        const int jc = (int)rint(((xb + 1.0) * W - 1.0) * 0.5), ic = (int)rint(((yb + 1.0) * H - 1.0) * 0.5);
        for (int i = ic - 1; i <= ic + 1; ++i) {
            if (i < 0 || i >= H) continue;
            for (int j = jc - 1; j <= jc + 1; ++j) {
                if (j < 0 || j >= W) continue;
                double ix, iy; src_of(r, H, W, i, j, ix, iy);
                const double wx = 1.0 - fabs(ix - x), wy = 1.0 - fabs(iy - y);
                if (wx > 0 && wy > 0) acc += (double)ds[t * W + j] * wx * wy;
            }
        }
    }
    dimg[(long long)k * H * W + idx] = (float)acc;
}

// s and ds may alias (in-place)
__global__ __launch_bounds__(256) void mse_grad_kernel(const float* s, const float* __restrict__ target, long long n_per,
                                                       float grad_scale, float* ds, double* __restrict__ mse_sum)
{
    __shared__ double s_red[8];
    const int k = blockIdx.y;
    double acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_per; i += (long long)gridDim.x * 256) {
        const float d = s[(long long)k * n_per + i] - target[i];
        acc += (double)(d * d);
        ds[(long long)k * n_per + i] = grad_scale * 2.f * d / (float)n_per;
    }
    const double tot = block_sum_d(acc / (double)n_per, s_red);
    if (threadIdx.x == 0) atomicAdd(mse_sum, tot);
}

int check(int n, int H, int W, int T)
{
    if (n < 1 || H < 1 || T < 1 || H != W) { set_error("radon: need n>=1, T>=1 and a square image (got n=%d H=%d W=%d T=%d)", n, H, W, T); return -1; }
    return 0;
}

}  // namespace

extern "C" {

int mfvi_radon_forward(const float* img, const float* theta_deg, int n, int H, int W, int T, float* sino, void* stream)
{
    if (check(n, H, W, T)) return -1;
    hipLaunchKernelGGL(radon_fwd_kernel, dim3((T * W + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, img, theta_deg, H, W, T, sino);
    return (int)hipGetLastError();
}

int mfvi_radon_adjoint(const float* dsino, const float* theta_deg, int n, int H, int W, int T, float* dimg, void* stream)
{
    if (check(n, H, W, T)) return -1;
    hipLaunchKernelGGL(radon_adj_kernel, dim3((H * W + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, dsino, theta_deg, H, W, T, dimg);
    return (int)hipGetLastError();
}

int mfvi_radon_mse(const float* out, const float* sino, const float* theta_deg, int n, int H, int W, int T, float grad_scale,
                   float* scratch, float* dout, double* mse_sum, void* stream)
{
    if (check(n, H, W, T)) return -1;
    hipStream_t st = (hipStream_t)stream;
    int rc = mfvi_radon_forward(out, theta_deg, n, H, W, T, scratch, stream); if (rc) return rc;
    const long long n_per = (long long)T * W;
    long long nb = (n_per + 255) / 256; if (nb > 256) nb = 256;
    hipLaunchKernelGGL(mse_grad_kernel, dim3((unsigned)nb, n), dim3(256), 0, st, scratch, sino, n_per, grad_scale, scratch, mse_sum);
    rc = (int)hipGetLastError(); if (rc) return rc;
    if (dout) return mfvi_radon_adjoint(scratch, theta_deg, n, H, W, T, dout, stream);
    return 0;
}

}  // extern "C"
