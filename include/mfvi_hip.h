/*
 * mfvi_hip.h — C ABI of libmfvi_hip.so: the MI355X (gfx950) implementation of the
 * mean-field-VI deep-image-prior hot path of Cardio-AI/mfvi-dip-mia.
 *
 * The reference is pure Python/PyTorch, so there is no existing FFI to mirror; these entry points are
 * what a ctypes/cffi binding of the reference's path binds (see INTEGRATION.md for the stub).  Each
 * one names the reference code it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *  - plain C: opaque handles, raw DEVICE pointers, sizes, a hipStream_t passed as void*.
 *  - every call is asynchronous on `stream` and allocates nothing; the caller owns all buffers.  One exception: mfvi_plan_create
 *    hipMallocs the plan's small device tables (BatchNorm, weight-sampling, gradient-reduction, Dropout2d entries; a few KB) and the
 *    plan lazily creates one low-priority side stream with its events; mfvi_plan_destroy releases them.
 *  - return value: 0 = ok, <0 = argument/shape error, >0 = hipError_t.  mfvi_last_error() returns a
 *    thread-local message for the last non-zero return.  Nothing throws across the boundary.
 *  - tensors are fp32, NCHW with N = MC sample index; parameters live in three flat fp32 blocks
 *    MU[n_vi], RHO[n_vi], BN[n_bn] (per VI layer: W then bias, in module order; per BatchNorm: gamma
 *    then beta).  Gradients use the same layout.
 *  - random numbers follow "RNG spec v1" (DESIGN.md): Philox4x32-10 keyed by seed, counter
 *    (block, domain<<24|stream, sample, step); eps of VI layer l, tensor t (0 = W, 1 = bias),
 *    element j is lane j&3 of block j>>2 of stream 2l+t in domain 0.
 *  - process-wide switches.  The library holds no mutable global state, but it reads the environment variables below ONCE (first use,
 *    function-local statics); none of them changes a result beyond summation-order rounding, all exist for A/B timing and for the
 *    cross-check tests.  Per-plan state has setters (mfvi_plan_set_*) instead.
 *      MFVI_DISABLE_MFMA=1     every convolution on the generic fp32 VALU kernels
 *      MFVI_AUTOTUNE=0         mfvi_plan_autotune returns at once (heuristic tilings)
 *      MFVI_TUNE=mf,th,T       force one tiling of the round-2 MFMA forward / backward-data kernels where the plan holds none
 *      MFVI_TUNE_W=nb,w,tgt    the same for the backward-weight kernels
 *      MFVI_RP=0               keep the 3x3 stride-1 layers on the round-2 kernels (row-phase kernels of conv_rp.hip off)
 *      MFVI_X6=0               heuristic tilings (no autotune): backward-weight never on the bf16x6 kernel (conv_bww_x6.hip); the autotuner's
 *                              candidates and explicit tilings (w = 11; forward: tune bit 25, conv_x6.hip) are not affected
 *      MFVI_TUNE_RP=mf,r,T[,rem[,ks]]  force one row-phase tiling where the plan holds none
 *      MFVI_RP_INTERLEAVE=0    row-phase kernels: a block takes a contiguous run of T tiles (default: tiles b, b + nx, b + 2 nx, ... so that the
 *                              blocks of an XCD work on adjacent tiles at every moment: halo rows and straddled cache lines are L2 hits)
 *      MFVI_PHASE=0            stride-2 backward-data: zero-stuffed formulation instead of the phase decomposition
 *      MFVI_FOLD_FUSION=0      1x1 backward-data writes the padded gradient + a finalize_dx launch (fold not fused)
 *      MFVI_FOLD_FUSION3=0     the same for the 3x3 stride-1 layers
 *      MFVI_GRAD_FROM_SLAB=1   grad_finalize reads eps * softplus(rho) as W_k - mu from the sampled-weight slab instead of re-deriving eps
 *      MFVI_SIDE_STREAM=0      backward-weight kernels on the caller's stream (default: the plan's side stream; mfvi_plan_set_side_stream)
 *      MFVI_SIDE_MAXPIX=n      only layers with at most n output pixels fork onto the side stream
 *      MFVI_SIDE_PRIO=0        side stream at default priority (default: lowest)
 *      MFVI_FWD_FORK=n         forward pass: skip-branch 1x1 convolutions on maps of up to n pixels run on the side stream (default 16384, 0 = off)
 *      MFVI_FORK_ON_PACKET=0   fork / join events as separate hipEventRecord packets instead of riding on kernel dispatch packets
 *      MFVI_FUSE_SKIP_BWD=0    the narrow 1x1 skip convolutions keep their own backward-data launch (default: formed inside the fold of the
 *                              tensor they share with the scale's stride-2 convolution, kernel family 5)
 *      MFVI_CONCAT_TILED=w     concat forward: LDS-tiled kernel for maps at least w wide (default 128; 0 = never)
 *      MFVI_CONCAT_TPB=n       concat forward, tiled kernel: tiles per block (default 4)
 *      MFVI_CONCAT_BWD_TILE=w  concat backward: force the 16 x 64 / 32 x 32 / 16 x 16 low-res tile (w = 64 / 32 / 16; default: by map width)
 *      MFVI_INKERNEL_MAX_W     (compile-time, csrc/common.h) largest layer the autotuner tries on the in-kernel-eps generic kernels (tune bit 27)
 *      MFVI_DEBUG_FIN          print the gradient-reduction table to stderr
 */
#ifndef MFVI_HIP_H
#define MFVI_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFVI_ABI_VERSION 6

typedef struct mfvi_plan mfvi_plan;

enum { MFVI_OP_CONV = 1, MFVI_OP_CONCAT_UP = 2, MFVI_OP_CONV_LRT = 3 };
enum { MFVI_UP_BILINEAR = 0, MFVI_UP_NEAREST = 1 };   /* nn.Upsample(scale_factor=2, mode=...) (models/skip.py:102) */
enum { MFVI_TASK_DENOISE = 0, MFVI_TASK_SR = 1 };
enum { MFVI_PARAM_F32 = 0, MFVI_PARAM_BF16 = 1 };     /* storage of the MU / RHO blocks (mfvi_plan_set_param_dtype) */

/* An activation tensor of the layer program, [C][H][W] per MC sample, stored RAW (conv output or
 * concat result).  Consumers read view(T) = LeakyReLU?(BatchNorm_train?(T)): the BatchNorm2d (training
 * mode, batch 1: models/common.py:96-97) and LeakyReLU(0.2) (models/common.py:83) that follow T in
 * models/skip.py:68-119 are folded into the loads of whatever consumes T. */
typedef struct {
    int32_t C, H, W;
    int32_t has_bn;      /* train-mode BN with per-sample statistics over H*W */
    int32_t has_act;     /* LeakyReLU(slope) after the BN */
    float   slope;
    float   eps;
    float   drop_p;      /* > 0: nn.Dropout2d(p) between the producing conv and the BN (models/common.py:125-131, the
                          * MC-dropout sibling's nets); needs has_bn.  Channel c of sample k is kept iff u >= p with u the
                          * U[0,1) of element c of stream <producing layer_id> in RNG domain 5, and scaled by 1/(1-p). */
    int64_t bn_off;      /* offset of gamma[C] (beta[C] follows) inside the BN block */
} mfvi_tensor_desc;

/* One fused op.
 * MFVI_OP_CONV       out = conv2d(reflection_pad(view(in0), ksize/2), w, b, stride), w = mu + softplus(rho)*eps
 *                    = ReflectionPad2d + Conv2dRT: models/common.py:100-135, BayTorch/modules/reparam_layers.py:26-37,
 *                      BayTorch/modules/module.py:82-85.
 * MFVI_OP_CONV_LRT   the same layer with the LOCAL reparameterisation (Conv2dLRT: BayTorch/modules/reparam_layers.py:39-72, conv.py:74-106;
 *                    MeanFieldVI(reparam='local')): with v = reflection_pad(view(in0)),
 *                        out = conv2d(v, mu, mu_b) + sqrt(1e-16 + conv2d(v**2, softplus(rho)**2, softplus(rho_b)**2)) * eps,
 *                    eps ~ N(0,1) in OUTPUT space: element j of [Cout][Ho][Wo] = lane j & 3 of block j >> 2 of RNG domain 7, stream
 *                    layer_id.  sample_weights = 0: out = conv2d(v, mu, mu_b) (LRTLayer's eval branch).  Same parameters, same KL.
 * MFVI_OP_CONCAT_UP  out = cat(view(in0), upsample2x_bilinear(view(in1)))  (in0 = -1: upsample only)
 *                    = Concat + nn.Upsample: models/common.py:23-43, models/skip.py:102.  out has in0's H x W, which may be one
 *                    row / column short of 2 x in1's (odd sizes): Concat's centre-crop (models/common.py:31-41) then drops the
 *                    up-sampled branch's last row / column.  Any other size relation is rejected. */
typedef struct {
    int32_t type;
    int32_t in0, in1, out;
    int32_t ksize, stride;
    int32_t layer_id;        /* VI layer index = RNG stream / 2 */
    int32_t up_mode;
    int64_t w_off, b_off;    /* offsets of W and bias inside MU / RHO; b_off < 0: no bias */
} mfvi_op_desc;

/* ---- layer program (replaces MeanFieldVI.forward / autograd of it: BayTorch/freq_to_bayes.py:40-41) ---- */
int mfvi_plan_create(const mfvi_tensor_desc* tensors, int n_tensors, const mfvi_op_desc* ops, int n_ops,
                     int input_tensor, int output_tensor, int64_t n_vi, int64_t n_bn, int max_samples,
                     mfvi_plan** plan);
void mfvi_plan_destroy(mfvi_plan* plan);
/* mfvi_backward runs the backward-weight kernels on a low-priority side stream owned by the plan (forked per layer behind the event
 * "dy of this layer is final" on the caller's stream, joined before the gradient reduction), so they overlap the backward-data /
 * fold chain that carries the critical path.  Results are unchanged.  enabled = 0 keeps every launch on the caller's stream
 * (e.g. to time a kernel alone); MFVI_SIDE_STREAM=0 in the environment does the same for every plan.  Once that stream exists,
 * mfvi_forward uses it too: a skip-branch convolution on a small map (<= MFVI_FWD_FORK pixels, default 128 x 128; 0 = never) runs there
 * beside the down path of its scale and is joined in front of its concat. */
int mfvi_plan_set_side_stream(mfvi_plan* plan, int enabled);
/* Device-resident iteration (ABI v6).  The reference's hot loop draws fresh noise every iteration from the loop index i
 * (bayesian_optimization.py:1360-1372); here that index is the `step` word of the counter RNG.  With a step source set, the plan's
 * kernels read the counter from DEVICE memory at run time (counter = step argument + *step_dev), so ONE captured HIP graph of an
 * iteration can be replayed for every i: the caller advances *step_dev on the device (one tiny launch inside the graph) and passes
 * step = 0.  nullptr (default): the `step` argument is the counter itself.  Results are bit-identical to passing the same counter
 * values from the host. */
int mfvi_plan_set_step_source(mfvi_plan* plan, const int32_t* step_dev);
/* enabled != 0 while the caller captures mfvi_forward / mfvi_backward into a HIP graph (hipStreamBeginCapture on `stream`): fork / join
 * events between the caller's stream and the plan's side stream are plain hipEventRecord / hipStreamWaitEvent pairs, which stream capture
 * turns into graph edges (default: they ride on the kernels' dispatch packets, hipExtLaunchKernelGGL, which capture does not record).
 * The plan must have run once un-captured (tables uploaded, events and the side stream created: none of that may happen under capture). */
int mfvi_plan_set_capture_mode(mfvi_plan* plan, int enabled);
/* Gradient split for an overlapped exchange (K sharded over ranks: DESIGN.md section 7; the reference has one process per fit and no exchange,
 * bayesian_optimization.py:3760-3775).  The flat layout is in op order and the backward pass runs the ops last to first, so the weight /
 * bias gradients of the ops >= first_op — the tail [offset, n_vi) of dmu and of drho — are complete long before the pass ends.  With a
 * split set, mfvi_backward reduces that group into dmu / drho on comm_stream as soon as the kernels producing it are enqueued (comm_stream
 * waits for them through events), and the rest at the end on the caller's stream as before: a collective enqueued on comm_stream after
 * mfvi_backward returns runs under the tail of the pass.  The caller joins comm_stream before it reads the gradients and before the
 * next mfvi_backward.  Results are bit-identical with and without a split.  first_op < 0 removes the split.  Not available for plans with
 * local-reparameterisation layers (their d rho is completed in one pass at the end).  mfvi_plan_grad_split_offset returns the tail's
 * first parameter index (-1 and an error if the ops >= first_op do not own a tail of the layout). */
int mfvi_plan_set_grad_split(mfvi_plan* plan, int first_op, void* comm_stream);
int mfvi_plan_grad_split_offset(const mfvi_plan* plan, int first_op, int64_t* offset);
/* Dropout2d layers of the program are active by default (the reference keeps its MC-dropout nets in train mode);
 * enabled = 0 makes them the identity (nn.Dropout2d in eval mode). */
int mfvi_plan_set_dropout(mfvi_plan* plan, int enabled);
/* nn.BatchNorm2d's running statistics (models/common.py:96-97: momentum 0.1, unbiased variance; the reference never reads them, but they
 * are part of the state_dict a reference run produces).  `running`: n_bn floats laid out like the BN block — running_mean at the gamma
 * slots, running_var at the beta slots.  mfvi_plan_bn_update_running applies the update of the n_samples batch-1 forwards just run by
 * mfvi_forward (their batch sums are still in the workspace), sample by sample in order.  mfvi_plan_set_bn_eval(plan, running) makes
 * mfvi_forward normalise with these statistics instead (module.eval(); NULL: back to batch statistics); forward only — mfvi_backward
 * refuses while it is set.  The pointer is kept, the caller keeps the buffer alive. */
int mfvi_plan_bn_update_running(const mfvi_plan* plan, const void* workspace, int n_samples, float momentum, float* running, void* stream);
int mfvi_plan_set_bn_eval(mfvi_plan* plan, const float* running);
/* bfloat16 storage of mu / rho (BASELINE configs[4]: "bf16 mu/rho with fp32 KL accumulate"; the reference keeps float32 Parameters:
 * BayTorch/modules/module.py:45-62).  With MFVI_PARAM_BF16 the `mu` / `rho` arguments of mfvi_forward / mfvi_backward / mfvi_plan_autotune
 * point to arrays of n_vi bf16 values (uint16_t: the upper half of the float32 pattern); the reparameterisation draw reads them directly
 * (half the parameter bytes per pass) and every value computed from them — sampled weights, activations, gradients, KL — is float32
 * exactly as with float32 storage of the same (bf16-representable) numbers.  Gradients, Adam moments and the BatchNorm block stay
 * float32.  The update rule is mfvi_elbo_update_bf16: no float32 master copy, stochastic rounding.  mu and rho must be 8-byte aligned. */
int mfvi_plan_set_param_dtype(mfvi_plan* plan, int dtype);
/* bytes of caller-provided device workspace (activations, gradients, BN statistics) for max_samples */
int64_t mfvi_plan_workspace_bytes(const mfvi_plan* plan);

/* n_samples MC forwards of the net on the SAME input z[Cin][H][W]; sample i uses eps of global sample
 * index k0+i.  sample_weights = 0 reproduces RTLayer's eval branch (w = mu).  out: [n_samples][Cout][H][W]. */
int mfvi_forward(mfvi_plan* plan, const void* mu, const void* rho, const float* bn, const float* z,
                 uint64_t seed, uint32_t step, uint32_t k0, int n_samples, int sample_weights,
                 void* workspace, float* out, void* stream);
/* Backward of the same call (workspace must still hold its activations).  dout: [n_samples][Cout][H][W].
 * dmu/drho/dbn are ACCUMULATED into (+=).  dz (optional): [n_samples][Cin][H][W].
 * mu / rho / bn must still hold the values the forward saw: the first backward after a forward with the same (pointers, seed, step,
 * k0, n_samples) re-uses the weights that forward drew (they are in the workspace) instead of drawing them again; any further
 * backward of the same forward draws them afresh from mu / rho. */
int mfvi_backward(mfvi_plan* plan, const void* mu, const void* rho, const float* bn, const float* z,
                  uint64_t seed, uint32_t step, uint32_t k0, int n_samples, int sample_weights,
                  void* workspace, const float* dout, float* dmu, float* drho, float* dbn, float* dz, void* stream);
/* Debug/parity access: copy tensor `tensor_id` of sample `sample` from the workspace to dst (device):
 * which = 0: raw activation [C][H][W]; 1: gradient wrt its BN output (valid after mfvi_backward);
 * 2: its forward BN sums as doubles [C][2] (dst must hold 2*C doubles). */
int mfvi_plan_read_tensor(const mfvi_plan* plan, const void* workspace, int tensor_id, int sample, int which, void* dst, void* stream);

/* Optional per-kernel timing with HIP events recorded on the caller's stream around the plan's launches.
 * mode 0: off; 1: every kernel; 2: only kernel (op, pass).  pass: 0 forward, 1 backward-weight, 2 backward-data,
 * 3 fold/finalize, 4 concat backward, 5 gradient finalize, 6 weight sampling (both op -1).  mfvi_plan_profile_read synchronises the recorded events, writes up to
 * `capacity` records (op index, pass, milliseconds) and clears the log. */
int mfvi_plan_profile(mfvi_plan* plan, int mode, int op, int pass);
int mfvi_plan_profile_read(mfvi_plan* plan, int capacity, int* n_records, int* ops, int* passes, float* ms);

/* Optional, once per plan: run one forward + backward, then time every valid MFMA tiling (output-channel fragments x tile
 * rows x tiles per block) of every conv op's forward, backward-data and backward-weight kernel with HIP events on `stream` and keep the
 * fastest per (op, pass).  Tilings do not change the forward result (same accumulation order per output element); gradients agree to
 * summation-order rounding (partial sums per pixel strip, the 4x4x1 variant below).
 * out_scratch: 2 * n_samples * numel(output tensor) floats; grad_scratch: 2 * n_vi + n_bn floats.  Synchronises `stream`.
 * Contents of workspace / scratch are undefined afterwards.  MFVI_AUTOTUNE=0 in the environment makes this a no-op. */
int mfvi_plan_autotune(mfvi_plan* plan, const void* mu, const void* rho, const float* bn, const float* z, int n_samples,
                       void* workspace, float* out_scratch, float* grad_scratch, void* stream);
/* Tiling in use for conv op `op`: which 0 forward, 1 backward-data (mf | th << 8 | T << 16; th bit 128 = tiles of whole rows for
 * narrow maps, th bit 64 (backward-data of layers with 16n + 4 input channels) = the last 4 channels on the 4x4x1 matrix instruction
 * instead of a padded 16-channel fragment), 2 backward-weight (input tiles | waves << 8 | block target/256 << 16; waves: 4, 8, 9 = 8 waves
 * producer/consumer specialised, 10 = the fragment-split variant with input tiles = 2: one block per 32-36 input channels, the accumulator
 * fragments dealt to the consumer waves); 0 = built-in heuristic. */
int mfvi_plan_get_tune(const mfvi_plan* plan, int op, int which);
int mfvi_plan_set_tune(mfvi_plan* plan, int op, int which, int tune);
/* Kernel family that served conv op `op` in its last forward (which 0) / backward-data (1) / backward-weight (2) launch: 0 = generic fp32
 * VALU kernels (also: a tiling the shape does not admit falls back to them), 1 = fp32 MFMA kernels, 2 = row-phase fp32 MFMA kernels
 * (3x3 stride 1, maps a multiple of 64 wide or exactly 32 / 16 wide), 3 = bf16x6 kernels (fp32 operands as three bf16 pieces on the bf16
 * matrix instruction: forward tune bit 25 = mf | rows << 8 | strips per block << 16, bit 12 = the 4-channel remainder plane rides on the
 * last 32-channel group's pass (mf = 1, Cin = 32 n + 4); backward-data with the fold tune bit 25 = strips per block | rows per strip << 8
 * (8 / 4 / 2 for 16 / 32 / 64 output channels), bit 16 = the strip-resident form for 32 (+ 4) -> 16 layers (conv_bwd_x6s.hip);
 * backward-weight w = 11, input tiles field = output fragments per block), 4 = one-stage kernels with the block's whole reduction in LDS
 * (tune = 1 | 1 << 26: 3x3 stride 1 on 8- / 16-wide maps with <= 144 reduction channels, conv_small.hip; 1x1 layers with 16 | Cin, Cout <= 128
 * and 2 / 4 / 8 output fragments on maps of 64 n pixels, conv_1x1.hip), 5 (backward-data slot only) = no launch of its own: the gradient of a
 * narrow 1x1 convolution (<= 8 output channels) wrt an input it shares with one other convolution is formed inside that tensor's fold
 * (finalize_dx_vec1_kernel; MFVI_FUSE_SKIP_BWD=0 keeps the separate launch), 6 (forward slot only) = streaming forward of a narrow 1x1
 * layer (tune = 1 | 1 << 28: Cin in {4 ... 16, 32, 64}, at most 16 output channels (32 for 16 or 32 input channels), maps of 64 n pixels; conv_1x1.hip).  -1: bad arguments.  Tests use it to prove that the kernel
 * under test is the one that ran. */
int mfvi_plan_last_kernel(const mfvi_plan* plan, int op, int which);

/* ---- losses -------------------------------------------------------------------------------------------- */
/* gaussian_nll (utils/bayesian_utils.py:29-32) for n samples of out[n][2][H][W] against target[H/f][W/f];
 * f > 1 applies the SR projection out[..., ::f, ::f] first (bayesian_optimization.py:2095-2099,2182-2185).
 * nll_sum (device double) += sum_i nll_i.  dout (optional) = grad_scale * d nll_i / d out_i. */
int mfvi_gaussian_nll(const float* out, const float* target, int n, int H, int W, int factor,
                      float grad_scale, float* dout, double* nll_sum, void* stream);
/* mse_loss(radon(out), sino) (bayesian_optimization.py:576, radon/radon.py:49-53): out[n][1][H][W],
 * theta_deg[T], sino[T][W]; scratch: n*T*W floats.  mse_sum += sum_i mse_i; dout = grad_scale * d mse_i/d out_i. */
/* gaussian_nll_inpainting (utils/bayesian_utils.py:35-39) with the runner's sigmoid on the colour channels
 * (bayesian_optimization.py:3033-3036): out[n][4][H][W] = 3 colour logits + 1 shared neg-log-variance, target[3][H][W],
 * mask[mask_channels][H][W] (1 or 3, broadcast); mean over the 3*H*W elements.  Same accumulation / gradient contract as
 * mfvi_gaussian_nll. */
int mfvi_gaussian_nll_inpainting(const float* out, const float* target, const float* mask, int mask_channels, int n, int H, int W,
                                 float grad_scale, float* dout, double* nll_sum, void* stream);
/* The same two losses in the call shape of the reference's functions — separate tensors, as the drop-in `gaussian_nll(out[:, :1],
 * out[:, 1:], target)` / `gaussian_nll_inpainting(out_pred, out[:, 3:], img, mask)` hand them over (utils/bayesian_utils.py:29-39):
 * mu / target [C][HW], neg_logvar [Cs][HW] with Cs == C or 1 (broadcast over channels), mask NULL or [Cm][HW] with Cm == C or 1.
 * loss_out (device double, overwritten) = mean (reduction_mean) or sum of (exp(s) (target - mu)^2 - s) * mask, s = clamp(neg_logvar, -20, 20).
 * The backward multiplies by the upstream gradient read from DEVICE memory (grad_out[0]): no host sync inside autograd. */
int mfvi_gaussian_nll_tensors(const float* mu, const float* neg_logvar, const float* target, const float* mask, int C, int Cs, int Cm, int64_t HW,
                              int reduction_mean, double* loss_out, void* stream);
int mfvi_gaussian_nll_tensors_backward(const float* mu, const float* neg_logvar, const float* target, const float* mask, int C, int Cs, int Cm,
                                       int64_t HW, int reduction_mean, const float* grad_out, float* dmu, float* dneg_logvar, void* stream);
int mfvi_radon_mse(const float* out, const float* sino, const float* theta_deg, int n, int H, int W, int T,
                   float grad_scale, float* scratch, float* dout, double* mse_sum, void* stream);
int mfvi_radon_forward(const float* img, const float* theta_deg, int n, int H, int W, int T, float* sino, void* stream);
int mfvi_radon_adjoint(const float* dsino, const float* theta_deg, int n, int H, int W, int T, float* dimg, void* stream);

/* ---- KL (VIModule._kl / MeanFieldVI.kl: BayTorch/modules/module.py:64-80, freq_to_bayes.py:43-48) ------- */
/* kl_out (device double, overwritten) = sum_j log(s_j/s0) + (s0^2 + (mu_j-m0)^2)/(2 s_j^2) - 1/2 */
int mfvi_kl(const float* mu, const float* rho, int64_t n, float prior_mu, float prior_sigma, double* kl_out, void* stream);
/* dmu += scale * dKL/dmu, drho += scale * dKL/drho */
int mfvi_kl_backward(const float* mu, const float* rho, int64_t n, float prior_mu, float prior_sigma, float scale,
                     float* dmu, float* drho, void* stream);

/* ---- optimizer (torch.optim.AdamW(lr, weight_decay=0): bayesian_optimization.py:1356-1357,1372) --------- */
int mfvi_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, int t, void* stream);
/* The deterministic tail of one ELBO iteration in one pass over the parameters (bayesian_optimization.py:1368-1372): kl_out = KL of all n_vi
 * (mu, rho) pairs, grads += temp * dKL (written back), then AdamW(weight_decay = 0) on the flat blocks [MU | RHO | BN] of
 * params / grads / m / v (2*n_vi + n_bn floats each).  Same per-element arithmetic as mfvi_kl + mfvi_kl_backward +
 * mfvi_adam_step; the KL sum is reduced in a fixed order (bit-reproducible).  scratch: mfvi_elbo_update_scratch_bytes()
 * bytes of device memory (per-block partial sums; contents need not be preserved between calls). */
int64_t mfvi_elbo_update_scratch_bytes(void);
int mfvi_elbo_update(float* params, float* grads, float* m, float* v, int64_t n_vi, int64_t n_bn, float prior_mu, float prior_sigma,
                     float temp, float lr, float beta1, float beta2, float eps, int t, double* kl_out, void* scratch, void* stream);
/* mfvi_elbo_update for bf16 mu / rho: mu_bf16 / rho_bf16 = n_vi bf16 values each, bn = the float32 BatchNorm block; grads / m / v
 * keep the float32 layout [MU | RHO | BN].  KL terms and the gradient in float32 on the values the bf16 parameters denote, KL summed in
 * float64, Adam in float32; the new mu / rho are rounded to bf16 STOCHASTICALLY (an fp32 master copy would double the parameter state;
 * round-to-nearest would freeze rho: lr = 1e-3 against ulp_bf16(-3) = 1.6e-2): magnitude rounded up with probability
 * (discarded 16 bits) / 2^16, the 16-bit word = upper half of lane j & 3 of Philox block j >> 2, RNG domain 6, stream 0 (MU) / 1 (RHO),
 * sample 0, step t — deterministic and reproducible on the CPU (oracle/). */
int mfvi_elbo_update_bf16(void* mu_bf16, void* rho_bf16, float* bn, float* grads, float* m, float* v, int64_t n_vi, int64_t n_bn, float prior_mu,
                          float prior_sigma, float temp, float lr, float beta1, float beta2, float eps, int t, uint64_t seed, double* kl_out,
                          void* scratch, void* stream);
/* bf16 <-> float32 copies of n values (float32 -> bf16 rounds to nearest even) */
int mfvi_bf16_to_f32(const void* src, int64_t n, float* dst, void* stream);
int mfvi_f32_to_bf16(const float* src, int64_t n, void* dst, void* stream);
/* The CT runners' NaN guard, `if not torch.isnan(loss): optimizer.step()` (bayesian_optimization.py:380, 581-582, 792, 994), without a host
 * sync.  loss = data term + temp * KL, and KL is a function of the parameters alone (finite while every earlier update was), so the guard
 * reads the DATA-TERM scalar of this iteration on the device: loss_d (the double accumulator of mfvi_radon_mse & co) and / or loss_f (the
 * float that rode the K-sharded all-reduce); either may be NULL.  NaN there: params / m / v are left untouched (kl_out is still written)
 * and *t_applied — the device counter of APPLIED updates that sets Adam's bias corrections, like torch's per-parameter state['step'] —
 * is not advanced.  Otherwise the update is number *t_applied + 1 and mfvi_elbo_update_guarded advances the counter itself;
 * mfvi_adamw_step_guarded only reads it (one optimizer step may cover several parameter blocks): call mfvi_step_advance once after them. */
int mfvi_elbo_update_guarded(float* params, float* grads, float* m, float* v, int64_t n_vi, int64_t n_bn, float prior_mu, float prior_sigma,
                             float temp, float lr, float beta1, float beta2, float eps, int32_t* t_applied, const double* loss_d,
                             const float* loss_f, double* kl_out, void* scratch, void* stream);
int mfvi_adamw_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                            const int32_t* t_applied, const double* loss_d, const float* loss_f, float weight_decay, void* stream);
int mfvi_step_advance(int32_t* t_applied, const double* loss_d, const float* loss_f, void* stream);
/* AdamW with decoupled weight decay (the SGLD sibling: bayesian_optimization.py:1765-1766): p *= 1 - lr*weight_decay first */
int mfvi_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, int t, float weight_decay, void* stream);
/* run_inp_dip's loss (bayesian_optimization.py:2824-2826): F.mse_loss(out[:, :3].sigmoid() * mask, img * mask), mean over 3*H*W;
 * out[n][4][H][W], target[3][H][W], mask[mask_channels][H][W] (1 or 3).  Same accumulation / gradient contract as
 * mfvi_gaussian_nll_inpainting (the 4th output channel gets a zero gradient). */
int mfvi_mse_sigmoid_masked(const float* out, const float* target, const float* mask, int mask_channels, int n, int H, int W,
                            float grad_scale, float* dout, double* mse_sum, void* stream);
/* F.mse_loss(out[:, channel], target) of the non-Bayesian siblings (bayesian_optimization.py:1177, 1780; SR :1983-1985 with
 * factor > 1 = the projection out[..., ::f, ::f] first): out[n][C][H][W], target[H/f][W/f]; mse_sum += sum_i mse_i;
 * dout (optional, all C channels written) = grad_scale * d mse_i / d out_i. */
int mfvi_mse_channel(const float* out, const float* target, int n, int C, int H, int W, int channel, int factor, float grad_scale,
                     float* dout, double* mse_sum, void* stream);

/* ---- RNG spec v1 on the device ----------------------------------------------------------------------------- */
/* out[j] = a + b * N(0,1)  (e.g. z = z0 + 0.1*noise uses mfvi_axpy_normal) */
int mfvi_normal_fill(uint64_t seed, uint32_t domain, uint32_t stream_id, uint32_t sample, uint32_t step, int64_t n,
                     float a, float b, float* out, void* stream);
int mfvi_uniform_fill(uint64_t seed, uint32_t stream_id, uint32_t sample, uint32_t step, int64_t n, float scale,
                      float* out, void* stream);
/* out[j] = lo + (hi - lo) * U[0,1)  (RNG domain 3): nn.Conv2d's default kaiming-uniform init of the DIP / SGLD siblings */
int mfvi_uniform_fill_range(uint64_t seed, uint32_t stream_id, uint32_t sample, uint32_t step, int64_t n, float lo, float hi,
                            float* out, void* stream);
/* x[j] += std * N(0,1), RNG domain 4 (SGLD's add_noise on the conv weights: bayesian_optimization.py:166-170) */
int mfvi_add_normal(float* x, uint64_t seed, uint32_t stream_id, uint32_t step, int64_t n, float std, void* stream);
/* net_input = net_input_saved + std * N(0,1)  (bayesian_optimization.py:1363-1364), RNG domain 1 */
int mfvi_perturb_input(const float* z0, uint64_t seed, uint32_t step, int64_t n, float std, float* z, void* stream);
/* the same with the counter word read on the device: step = step_offset + *step_dev (mfvi_plan_set_step_source's convention) */
int mfvi_perturb_input_dev(const float* z0, uint64_t seed, const int32_t* step_dev, uint32_t step_offset, int64_t n, float std, float* z,
                           void* stream);

/* ---- per-iteration bookkeeping (bayesian_optimization.py:1374-1406; utils/common_utils.py:297-353) --------- */
/* acc[0] += mse(a,b) numerator pieces: returns sum (a-b)^2 over n elements into sum_out (device double, overwritten) */
int mfvi_sq_err_sum(const float* a, const float* b, int64_t n, double* sum_out, void* stream);
/* SSIM map mean (11x11 Gaussian sigma 1.5, zero padding) of two [H][W] images; ssim_sum overwritten with the
 * sum of the SSIM map (divide by H*W). */
int mfvi_ssim_sum(const float* a, const float* b, int H, int W, double* ssim_sum, void* stream);
/* dst[h][w] = src[y*factor][x*factor]: the SR runner's nearest-neighbour downsampler (bayesian_optimization.py:2095-2099) applied to the
 * smoothed / clipped outputs for the low-resolution metrics (:2203, :2207, :2214-2217); h = H / factor, w = W / factor. */
int mfvi_decimate(const float* src, int H, int W, int factor, float* dst, void* stream);
/* One pass of the runner's per-iteration bookkeeping (bayesian_optimization.py:1374-1396) with no host sync: sample means of
 * out[:,0] and exp(-out[:,1]), EMA (weight w; first = 1 copies), clipped copies for the metrics, ring-buffer slot writes
 * (slot pointers may be NULL).  C = 2 (den/SR) or 1 (CT: no aleatoric channel). */
int mfvi_bookkeep(const float* out, int n, int C, int H, int W, float* ema, float ema_weight, int first, float* out_clip,
                  float* ale_clip, float* avg_clip, float* ring_epi_slot, float* ring_ale_slot, void* stream);
/* The same pass for the inpainting runner (bayesian_optimization.py:3039-3064): out[n][4][H][W] = 3 colour logits + 1 log-precision;
 * m = mean_k sigmoid(out[k][:3]), a = mean_k exp(-out[k][3]); ema[4][HW]; out_clip[3][HW], ale_clip[HW], avg_clip[3][HW] clipped to
 * [0,1]; masked copies (x * mask, mask[mask_channels][HW]) of img, out_clip and avg_clip for the masked PSNR/SSIM; ring slots [3][HW] / [HW]. */
int mfvi_bookkeep_inpainting(const float* out, int n, int H, int W, const float* img, const float* mask, int mask_channels, float* ema,
                             float ema_weight, int first, float* out_clip, float* ale_clip, float* avg_clip, float* img_masked,
                             float* out_masked, float* avg_masked, float* ring_epi_slot, float* ring_ale_slot, void* stream);
/* torch.var(ring, dim=0) (unbiased) / torch.mean(ring, dim=0) over R ring-buffer slots (bayesian_optimization.py:1412-1413) */
int mfvi_ring_stats(const float* ring, int R, int H, int W, float* var_out, float* mean_out, void* stream);
/* post-step output handling: mean[n][HW] kept, out[:,1] <- exp(-out[:,1]); ema = ema*w + out*(1-w) (first: copy) */
int mfvi_post_step(float* out, int n, int C, int H, int W, float* ema, float ema_weight, int first, void* stream);

const char* mfvi_last_error(void);
int mfvi_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
