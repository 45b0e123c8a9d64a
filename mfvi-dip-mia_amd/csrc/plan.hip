// Host side of libmfvi_hip: the layer-program planner/executor behind mfvi_plan_* / mfvi_forward /
// mfvi_backward (include/mfvi_hip.h).  It validates the fused-op program emitted by the Python front-end
// (which walks the reference's module tree: models/skip.py:58-134), lays the activations, gradients and
// BN statistics out in one caller-provided workspace, and issues the kernels on the caller's stream.
#include "common.h"
#include "../../include/mfvi_hip.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}

thread_local hipEvent_t mfvi_tl_stop_event = nullptr;      // see mfvi_launch (common.h)
thread_local int mfvi_tl_family = 0;

namespace {

struct TensorInfo {
    mfvi_tensor_desc d;
    long long numel = 0;
    long long act_off = -1, ga_off = -1;       // floats, from the float arena base
    long long stats_off = -1;                  // doubles, inside the fwd-stats block (same offset in the bsums block)
    long long drop_off = -1;                   // floats: Dropout2d factors [max_samples][C] (drop_p > 0 only)
    int producer = -1;
    std::vector<int> consumers;                // op indices, forward order
};

struct OpInfo {
    mfvi_op_desc d;
    ConvGeom g;
    long long scratch_off = -1;                // floats: padded input-gradient scratch of this conv
    long long scratch2_off = -1, s2_off = -1;  // LRT: padded scratch of the variance convolution's input gradient; s2 = conv(v^2, sigma^2) kept for the backward
    long long part_off = -1, part_stride = 0;  // floats: partial-dW slabs of the MFMA backward-weight kernel [strip][sample][stride]
    int max_strips = 0;
    long long x6w_off = -1;                    // floats: split weight pieces of the bf16x6 forward (conv_x6.hip), -1: shape not served
    long long x6bw_off = -1;                   // floats: split weight pieces of the bf16x6 backward-data (conv_bwd_x6.hip), -1: shape not served
    mutable int family[3] = {0, 0, 0};         // kernel family of the last forward / backward-data / backward-weight launch (mfvi_plan_last_kernel)
};

inline long long align_up(long long v, long long a) { return (v + a - 1) / a * a; }

}  // namespace

struct mfvi_plan {
    std::vector<TensorInfo> t;
    std::vector<OpInfo> ops;
    int input = -1, output = -1, max_samples = 0;
    long long n_vi = 0, n_bn = 0;
    long long stats_doubles = 0;               // per block (fwd stats | bsums), for max_samples
    const void* bsums_clean_ws = nullptr;      // workspace whose BN-backward sums the last forward zeroed (one memset for both blocks) with no backward since
    long long float_base = 0;                  // byte offset of the float arena
    long long total_bytes = 0;
    BnGradEntry* table_dev = nullptr; int n_entries = 0, max_c = 1;
    SampleEntry* samp_dev = nullptr; int n_samp = 0, samp_blocks = 0;    // layers whose weights are drawn once per pass
    long long wsamp_off = -1;                  // floats: sampled weights [max_samples][n_vi]
    X6SplitEntry* x6_dev = nullptr; std::vector<X6SplitEntry> x6_uploaded;      // table of the bf16x6 forward layers' weight split (conv_x6.hip)
    X6BSplitEntry* x6b_dev = nullptr; std::vector<X6BSplitEntry> x6b_uploaded;   // the same for the bf16x6 backward-data layers (conv_bwd_x6.hip)
    int param_dtype = MFVI_PARAM_F32;          // storage of mu / rho handed to forward / backward (MFVI_PARAM_BF16: bf16_t arrays)
    const int32_t* step_dev = nullptr;         // device-resident step counter (mfvi_plan_set_step_source): the `step` argument of forward / backward is an offset to it
    bool capture_mode = false;                 // the calls are being captured into a HIP graph: fork / join events as plain records (no events on kernel packets)
    int n_generic = 0;                         // conv layers outside the sampling table (served by the generic fp32 kernels)
    long long p32_off = -1;                    // floats: [mu | rho] expanded to float32 for those kernels when mu / rho are bf16
    const float* bn_eval = nullptr;            // BatchNorm in eval mode: running statistics used by mfvi_forward (nullptr: batch statistics)
    int n_lrt = 0;                             // local-reparameterisation layers
    long long sig2_off = -1, dsig2_off = -1;   // floats [n_vi] each: softplus(rho)^2 of this pass / gradient wrt it
    long long lrt_tmp_off = -1, lrt_tmp_n = 0; // floats: mean-convolution output (forward) / ds2 (backward) of the LRT layer in flight
    // identity of the draw currently held in the sampled-weight slab (set by forward, reused by the matching backward)
    const void* samp_mu = nullptr; const void* samp_rho = nullptr; const void* samp_ws = nullptr;
    uint64_t samp_seed = 0; uint32_t samp_step = 0, samp_k0 = 0; int samp_n = 0;
    DropEntry* drop_dev = nullptr; int n_drop = 0; bool dropout_on = true;   // Dropout2d layers (MC-dropout sibling)
    // Backward-weight launches are off the critical path of the backward pass (only grad_finalize needs them): they run on a side
    // stream of the plan, forked per layer behind the event that marks "dy of this layer is final" and joined before grad_finalize,
    // so they fill the CUs the latency-bound backward-data / fold kernels of the small maps leave idle.
    hipStream_t side = nullptr; std::vector<hipEvent_t> fork_events; hipEvent_t join_event = nullptr; bool side_enabled = true;
    std::vector<hipEvent_t> fwd_events;        // forward pass: skip-branch convolutions beside the down path (MFVI_FWD_FORK)
    GradFinEntry* fin_dev = nullptr;           // table of the layers whose partial dW slabs grad_finalize reduces
    std::vector<GradFinEntry> fin_uploaded;
    // Gradient split for an overlapped exchange (mfvi_plan_set_grad_split): the backward pass reduces the weight gradients of the ops
    // >= split_op on split_stream as soon as their backward-weight kernels have been enqueued, the rest at the end as before.  The early
    // group has its own half of the device table (fin_dev + n_conv).
    int split_op = -1; hipStream_t split_stream = nullptr; hipEvent_t split_ev[2] = {nullptr, nullptr}; int n_conv = 0;
    // three tables, each with its own device slot (fin_dev + {0, 1, 2} * n_conv) and cached host copy: the whole pass (no split), the
    // early group of a split pass, the late group of a split pass.  An engine that splits only the LAST launch of an iteration
    // (K_local > samples per launch) alternates between "whole" and "early + late": with one slot shared by "whole" and "late" the cache
    // missed twice per iteration, and the reassigned host vector was the source of a copy still in flight.
    std::vector<GradFinEntry> fin_uploaded_early, fin_uploaded_late;
    // optional per-kernel timing with HIP events on the caller's stream (bench.py's roofline leg)
    struct Rec { int op, pass; hipEvent_t a, b; };
    int prof_mode = 0, prof_op = -1, prof_pass = -1;      // 0 off, 1 every kernel, 2 only (prof_op, prof_pass)
    std::vector<Rec> recs;
    std::vector<hipEvent_t> free_events;
};

namespace {

bool fail(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return false;
}

bool build(mfvi_plan& p, const mfvi_tensor_desc* td, int n_t, const mfvi_op_desc* od, int n_ops)
{
    if (n_t < 2 || n_ops < 1) return fail("plan: need >= 2 tensors and >= 1 op");
    if (p.input < 0 || p.input >= n_t || p.output < 0 || p.output >= n_t || p.input == p.output)
        return fail("plan: bad input/output tensor ids %d/%d", p.input, p.output);
    if (p.max_samples < 1) return fail("plan: max_samples must be >= 1");
    p.t.resize(n_t); p.ops.resize(n_ops);
    for (int i = 0; i < n_t; ++i) {
        TensorInfo& ti = p.t[i]; ti.d = td[i];
        if (ti.d.C < 1 || ti.d.H < 1 || ti.d.W < 1 || ti.d.C > MFVI_MAX_C) return fail("plan: tensor %d has bad shape (%d,%d,%d)", i, ti.d.C, ti.d.H, ti.d.W);
        if (ti.d.has_act && !ti.d.has_bn) return fail("plan: tensor %d: activation without BatchNorm is not part of the skip() family", i);
        if (ti.d.has_bn && (ti.d.bn_off < 0 || ti.d.bn_off + 2LL * ti.d.C > p.n_bn)) return fail("plan: tensor %d: bn_off out of range", i);
        if (!(ti.d.drop_p >= 0.f && ti.d.drop_p < 1.f)) return fail("plan: tensor %d: dropout probability %g outside [0, 1)", i, (double)ti.d.drop_p);
        if (ti.d.has_act && !(ti.d.slope >= 0.f && ti.d.slope <= 1.f)) return fail("plan: tensor %d: LeakyReLU slope %g outside [0, 1] (the kernels form it as max(v, slope * v))", i, (double)ti.d.slope);
        if (ti.d.drop_p > 0.f && !ti.d.has_bn) return fail("plan: tensor %d: Dropout2d without a following BatchNorm is not part of the skip() family", i);
        ti.numel = (long long)ti.d.C * ti.d.H * ti.d.W;
    }
    if (p.t[p.input].d.has_bn) return fail("plan: the input tensor cannot carry a BatchNorm");
    if (p.t[p.output].d.has_bn) return fail("plan: the output tensor must be raw (no BatchNorm/activation)");
    for (int i = 0; i < n_ops; ++i) {
        OpInfo& o = p.ops[i]; o.d = od[i];
        const mfvi_op_desc& d = o.d;
        if (d.out < 0 || d.out >= n_t || d.out == p.input) return fail("plan: op %d: bad output tensor", i);
        if (p.t[d.out].producer >= 0) return fail("plan: tensor %d produced twice", d.out);
        p.t[d.out].producer = i;
        if (d.type == MFVI_OP_CONV || d.type == MFVI_OP_CONV_LRT) {
            if (d.in0 < 0 || d.in0 >= n_t) return fail("plan: op %d: bad input tensor", i);
            if (d.type == MFVI_OP_CONV_LRT) { ++p.n_lrt; if (d.b_off < 0) return fail("plan: op %d: a local-reparameterisation layer needs its bias (skip() builds every conv with one)", i); }
            const TensorInfo& x = p.t[d.in0]; const TensorInfo& y = p.t[d.out];
            if (!(((d.ksize == 3 || d.ksize == 5) && (d.stride == 1 || d.stride == 2)) || (d.ksize == 1 && d.stride == 1)))
                return fail("plan: op %d: conv ksize %d stride %d not supported (3x3 / 5x5 s1/s2, 1x1 s1)", i, d.ksize, d.stride);
            const int P = d.ksize / 2;
            const int Ho = (x.d.H + 2 * P - d.ksize) / d.stride + 1, Wo = (x.d.W + 2 * P - d.ksize) / d.stride + 1;
            if (Ho != y.d.H || Wo != y.d.W) return fail("plan: op %d: output spatial size (%d,%d) != expected (%d,%d)", i, y.d.H, y.d.W, Ho, Wo);
            if (P > 0 && (x.d.H <= P || x.d.W <= P)) return fail("plan: op %d: reflection padding %d needs H,W > %d", i, P, P);
            const long long nw = (long long)y.d.C * x.d.C * d.ksize * d.ksize;
            if (d.w_off < 0 || d.w_off + nw > p.n_vi) return fail("plan: op %d: w_off out of range", i);
            if (d.b_off >= 0 && d.b_off + y.d.C > p.n_vi) return fail("plan: op %d: b_off out of range", i);
            if (d.layer_id < 0 || d.layer_id >= (1 << 22)) return fail("plan: op %d: bad layer_id", i);
            o.g = ConvGeom{x.d.C, y.d.C, x.d.H, x.d.W, Ho, Wo, d.ksize, d.stride, d.w_off, d.b_off, d.layer_id};
            p.t[d.in0].consumers.push_back(i);
        } else if (d.type == MFVI_OP_CONCAT_UP) {
            if (d.in1 < 0 || d.in1 >= n_t || d.in0 >= n_t) return fail("plan: op %d: bad input tensors", i);
            if (d.up_mode != MFVI_UP_BILINEAR && d.up_mode != MFVI_UP_NEAREST) return fail("plan: op %d: unknown upsampling mode %d (bilinear, nearest)", i, d.up_mode);
            const TensorInfo& b = p.t[d.in1]; const TensorInfo& y = p.t[d.out];
            int Ca = 0;
            if (d.in0 >= 0) {
                const TensorInfo& a = p.t[d.in0]; Ca = a.d.C;
                // Concat centre-crops to the smaller input (models/common.py:31-41).  In skip() the up-sampled branch is 2*ceil(H/2) against
                // the skip branch's H, so the crop offset (size - target) / 2 is 0 and at most its last row / column is dropped.
                if (2 * b.d.H < a.d.H || 2 * b.d.H - a.d.H > 1 || 2 * b.d.W < a.d.W || 2 * b.d.W - a.d.W > 1)
                    return fail("plan: op %d: concat inputs %dx%d vs 2*%dx%d: only the crop of the up-sampled branch by one row / column is built", i, a.d.H, a.d.W, b.d.H, b.d.W);
                if (d.in0 == p.input) return fail("plan: op %d: the net input cannot feed a concat", i);
                p.t[d.in0].consumers.push_back(i);
            }
            if (d.in1 == p.input) return fail("plan: op %d: the net input cannot feed an upsample", i);
            const int cH = d.in0 >= 0 ? p.t[d.in0].d.H : 2 * b.d.H, cW = d.in0 >= 0 ? p.t[d.in0].d.W : 2 * b.d.W;
            if (y.d.C != Ca + b.d.C || y.d.H != cH || y.d.W != cW) return fail("plan: op %d: concat output shape mismatch", i);
            p.t[d.in1].consumers.push_back(i);
        } else return fail("plan: op %d: unknown type %d", i, d.type);
        // inputs must already be produced (program order = execution order)
        const int ins[2] = {d.in0, d.type == MFVI_OP_CONCAT_UP ? d.in1 : -1};
        for (int q = 0; q < 2; ++q)
            if (ins[q] >= 0 && ins[q] != p.input && (p.t[ins[q]].producer < 0 || p.t[ins[q]].producer >= i))
                return fail("plan: op %d reads tensor %d before it is produced", i, ins[q]);
    }
    for (int i = 0; i < n_t; ++i) {
        if (p.t[i].d.drop_p > 0.f && (p.t[i].producer < 0 || p.ops[p.t[i].producer].d.type == MFVI_OP_CONCAT_UP))
            return fail("plan: tensor %d: Dropout2d must follow a convolution", i);
        if (i != p.input && p.t[i].producer < 0) return fail("plan: tensor %d is never produced", i);
        if (i != p.output && p.t[i].consumers.empty()) return fail("plan: tensor %d is never consumed", i);
        if (i == p.output && !p.t[i].consumers.empty()) return fail("plan: the output tensor has consumers");
        bool cat = false;
        for (int c : p.t[i].consumers) cat |= p.ops[c].d.type == MFVI_OP_CONCAT_UP;
        if (p.n_lrt && p.t[i].d.drop_p > 0.f) return fail("plan: Dropout2d and local-reparameterisation layers are not combined by any runner");
        if (cat && p.t[i].consumers.size() != 1) return fail("plan: tensor %d feeds a concat and something else", i);
        if (p.t[i].consumers.size() > 2) return fail("plan: tensor %d has %d consumers (max 2)", i, (int)p.t[i].consumers.size());
    }
    // ---- workspace layout ----
    long long sd = 0;
    std::vector<BnGradEntry> table;
    for (int i = 0; i < n_t; ++i)
        if (p.t[i].d.has_bn) {
            p.t[i].stats_off = sd; sd += (long long)p.max_samples * p.t[i].d.C * 2;
            BnGradEntry e; e.bsums_off = p.t[i].stats_off; e.bn_off = p.t[i].d.bn_off; e.C = p.t[i].d.C; e.hw = p.t[i].d.H * p.t[i].d.W;
            table.push_back(e); if (p.t[i].d.C > p.max_c) p.max_c = p.t[i].d.C;
        }
    p.stats_doubles = align_up(sd, 32);
    p.float_base = 2 * p.stats_doubles * (long long)sizeof(double);
    long long fo = 0;
    auto take = [&](long long n) { const long long o = fo; fo += align_up(n, 64); return o; };
    for (int i = 0; i < n_t; ++i) {
        if (i != p.input && i != p.output) p.t[i].act_off = take(p.t[i].numel * p.max_samples);
        if (i != p.output && i != p.input) p.t[i].ga_off = take(p.t[i].numel * p.max_samples);
    }
    std::vector<DropEntry> drops;
    for (int i = 0; i < n_t; ++i)
        if (p.t[i].d.drop_p > 0.f) {
            p.t[i].drop_off = take((long long)p.t[i].d.C * p.max_samples);
            DropEntry e{}; e.drop_off = p.t[i].drop_off; e.C = p.t[i].d.C; e.layer_id = p.ops[p.t[i].producer].d.layer_id; e.p = p.t[i].d.drop_p;
            drops.push_back(e);
        }
    p.n_drop = (int)drops.size();
    if (p.n_drop) {
        hipError_t e = hipMalloc((void**)&p.drop_dev, sizeof(DropEntry) * drops.size());
        if (e == hipSuccess) e = hipMemcpy(p.drop_dev, drops.data(), sizeof(DropEntry) * drops.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail("plan: dropout table setup failed: %s", hipGetErrorString(e));
    }
    long long shared_scratch = 0;
    for (auto& o : p.ops)
        if (o.d.type != MFVI_OP_CONCAT_UP) {
            const int P = o.g.ks / 2;
            const bool lrt = o.d.type == MFVI_OP_CONV_LRT;
            const long long n = (long long)o.g.Cin * (o.g.H + 2 * P) * (o.g.W + 2 * P) * p.max_samples;
            if (p.t[o.d.in0].consumers.size() > 1) { o.scratch_off = take(n); if (lrt) o.scratch2_off = take(n); }       // live until the fold of in0
            else if ((lrt ? 2 : 1) * n > shared_scratch) shared_scratch = (lrt ? 2 : 1) * n;
            if (lrt) {
                const long long no = (long long)o.g.Cout * o.g.Ho * o.g.Wo * p.max_samples;
                o.s2_off = take(no);
                if (no > p.lrt_tmp_n) p.lrt_tmp_n = no;
            }
        }
    const long long shared_off = take(shared_scratch);
    for (auto& o : p.ops)
        if (o.d.type != MFVI_OP_CONCAT_UP && o.scratch_off < 0) {
            const int P = o.g.ks / 2;
            o.scratch_off = shared_off;
            if (o.d.type == MFVI_OP_CONV_LRT) o.scratch2_off = shared_off + (long long)o.g.Cin * (o.g.H + 2 * P) * (o.g.W + 2 * P) * p.max_samples;
        }
    if (p.n_lrt) { p.sig2_off = take(p.n_vi); p.dsig2_off = take(p.n_vi); p.lrt_tmp_off = take(p.lrt_tmp_n); }
    // weights of the MFMA-served layers are sampled once per pass into [max_samples][n_vi]
    std::vector<SampleEntry> samp;
    for (auto& o : p.ops)      // (LRT layers draw nothing in weight space: their convolutions read mu and softplus(rho)^2)
        if (o.d.type == MFVI_OP_CONV && !(o.g.Cin & 3) && !(o.g.w_off & 3) && o.g.Cin <= MFVI_MAX_C && o.g.Cout <= MFVI_MAX_C) {
            SampleEntry e{};
            e.w_off = o.g.w_off; e.b_off = o.g.b_off; e.n_w = o.g.Cout * o.g.Cin * o.g.ks * o.g.ks; e.n_b = o.g.b_off >= 0 ? o.g.Cout : 0;
            e.layer_id = o.g.layer_id; e.first_block = p.samp_blocks;
            p.samp_blocks += ((e.n_w >> 2) + ((e.n_b + 3) >> 2) + SAMPLE_QUADS - 1) / SAMPLE_QUADS;
            samp.push_back(e);
        }
    p.n_samp = (int)samp.size();
    for (auto& o : p.ops) if (o.d.type != MFVI_OP_CONCAT_UP) ++p.n_generic;
    p.n_generic -= p.n_samp;
    if (p.n_generic > 0) p.p32_off = take(2 * p.n_vi);
    if (p.n_samp) {
        p.wsamp_off = take(p.n_vi * p.max_samples);
        hipError_t e = hipMalloc((void**)&p.samp_dev, sizeof(SampleEntry) * samp.size());
        if (e == hipSuccess) e = hipMemcpy(p.samp_dev, samp.data(), sizeof(SampleEntry) * samp.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail("plan: sampling table setup failed: %s", hipGetErrorString(e));
    }
    // partial-dW slabs: up to ~4M floats per layer, at least one pixel strip
    int n_conv = 0;
    for (auto& o : p.ops)
        if (o.d.type != MFVI_OP_CONCAT_UP) {
            ++n_conv;
            const long long n_w = (long long)o.g.Cout * o.g.Cin * o.g.ks * o.g.ks;
            o.part_stride = n_w + (o.g.b_off >= 0 ? align_up(o.g.Cout, 4) : 0);
            const long long ms = (4LL << 20) / (o.part_stride * p.max_samples);
            o.max_strips = (int)(ms < 1 ? 1 : (ms > 64 ? 64 : ms));
            o.part_off = take(o.part_stride * p.max_samples * o.max_strips);
            if (o.d.type == MFVI_OP_CONV) { const long long xf = x6_fwd_scratch_floats(o.g, p.max_samples); if (xf > 0) o.x6w_off = take(xf); }
            if (o.d.type == MFVI_OP_CONV && o.d.in0 != p.input && p.t[o.d.in0].consumers.size() == 1) {      // (the fused-fold path: the conv's input feeds nothing else)
                const long long xb = x6_bwd_scratch_floats(o.g, p.max_samples); if (xb > 0) o.x6bw_off = take(xb); }
        }
    p.total_bytes = p.float_base + fo * (long long)sizeof(float);
    if (n_conv) {
        p.n_conv = n_conv;
        const hipError_t e = hipMalloc((void**)&p.fin_dev, sizeof(GradFinEntry) * n_conv * 3);      // whole pass | early group | late group of a gradient split
        if (e != hipSuccess) return fail("plan: hipMalloc of the gradient table failed: %s", hipGetErrorString(e));
    }
    p.n_entries = (int)table.size();
    if (p.n_entries) {
        hipError_t e = hipMalloc((void**)&p.table_dev, sizeof(BnGradEntry) * table.size());
        if (e != hipSuccess) return fail("plan: hipMalloc of the BN table failed: %s", hipGetErrorString(e));
        e = hipMemcpy(p.table_dev, table.data(), sizeof(BnGradEntry) * table.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail("plan: hipMemcpy of the BN table failed: %s", hipGetErrorString(e));
    }
    return true;
}

enum { PASS_FWD = 0, PASS_BWD_WEIGHT = 1, PASS_BWD_DATA = 2, PASS_FINALIZE = 3, PASS_CONCAT_BWD = 4, PASS_GRAD_FINALIZE = 5, PASS_SAMPLE = 6 };

struct ProfScope {
    mfvi_plan* p; hipStream_t st; bool on; hipEvent_t a, b; int op, pass;
    ProfScope(mfvi_plan* p_, int op_, int pass_, hipStream_t st_) : p(p_), st(st_), on(false), op(op_), pass(pass_)
    {
        on = p->prof_mode == 1 || (p->prof_mode == 2 && p->prof_op == op && p->prof_pass == pass);
        if (!on) return;
        auto get = [&]() { hipEvent_t e; if (!p->free_events.empty()) { e = p->free_events.back(); p->free_events.pop_back(); } else (void)hipEventCreate(&e); return e; };
        a = get(); b = get();
        (void)hipEventRecord(a, st);
    }
    ~ProfScope() { if (on) { (void)hipEventRecord(b, st); p->recs.push_back({op, pass, a, b}); } }
};

struct Ctx {
    const mfvi_plan& p; char* ws; const float* bn; const float* z; int n;
    double* fstats() const { return (double*)ws; }
    double* bsums() const { return (double*)ws + p.stats_doubles; }
    float* farena() const { return (float*)(ws + p.float_base); }
    float* wsamp() const { return p.wsamp_off >= 0 ? farena() + p.wsamp_off : nullptr; }
    TView view(int i, const float* out_ptr = nullptr) const
    {
        const TensorInfo& t = p.t[i]; TView v;
        if (i == p.input) { v.data = z; v.sstride = 0; }
        else if (i == p.output) { v.data = out_ptr; v.sstride = t.numel; }
        else { v.data = farena() + t.act_off; v.sstride = t.numel; }
        v.C = t.d.C; v.H = t.d.H; v.W = t.d.W;
        v.stats = t.d.has_bn ? fstats() + t.stats_off : nullptr;
        v.gamma = t.d.has_bn ? bn + t.d.bn_off : nullptr;
        v.eps = t.d.eps; v.slope = t.d.slope; v.act = t.d.has_act;
        v.drop = (t.drop_off >= 0 && p.dropout_on) ? farena() + t.drop_off : nullptr;
        return v;
    }
    GView gview(int i, const float* dout) const
    {
        const TensorInfo& t = p.t[i]; GView g;
        g.ga = (i == p.output) ? dout : farena() + t.ga_off; g.gstride = t.numel;
        g.y = (i == p.output) ? nullptr : farena() + t.act_off; g.ystride = t.numel;
        g.C = t.d.C; g.H = t.d.H; g.W = t.d.W;
        g.stats = t.d.has_bn ? fstats() + t.stats_off : nullptr;
        g.bsums = t.d.has_bn ? bsums() + t.stats_off : nullptr;
        g.gamma = t.d.has_bn ? bn + t.d.bn_off : nullptr;
        g.eps = t.d.eps;
        g.drop = (t.drop_off >= 0 && p.dropout_on) ? farena() + t.drop_off : nullptr;
        return g;
    }
};

bool check_call(const mfvi_plan* p, int n_samples, const void* ws)
{
    if (!p) return fail("null plan");
    if (!ws) return fail("null workspace");
    if (n_samples < 1 || n_samples > p->max_samples) return fail("n_samples %d outside 1..%d", n_samples, p->max_samples);
    return true;
}

// MFVI_DISABLE_MFMA=1 forces the generic fp32 VALU kernels (A/B timing and parity cross-checks)
bool use_mfma()
{
    static const bool on = [] { const char* e = getenv("MFVI_DISABLE_MFMA"); return !(e && e[0] == '1'); }();
    return on;
}

// MFVI_GRAD_FROM_SLAB=1: grad_finalize reads eps * softplus(rho) as W_k - mu from the sampled-weight slab instead of re-deriving eps
// from the counter RNG.  Measured slower on MI355X (87 vs 72 us: the extra 66 MB of loads cost more than the Philox work they save),
// kept as an A/B switch.
// MFVI_FOLD_FUSION=0: 1x1 backward-data always goes through the padded-gradient scratch + finalize_dx (A/B and parity cross-checks)
bool fold_fusion_on()
{
    static const bool on = [] { const char* e = getenv("MFVI_FOLD_FUSION"); return !(e && e[0] == '0'); }();
    return on;
}

// MFVI_FOLD_FUSION3=0: 3x3 stride-1 backward-data keeps the padded-gradient scratch + finalize_dx (A/B and parity cross-checks)
bool fold_fusion3_on()
{
    static const bool on = [] { const char* e = getenv("MFVI_FOLD_FUSION3"); return !(e && e[0] == '0'); }();
    return on;
}

bool grad_from_slab()
{
    static const bool on = [] { const char* e = getenv("MFVI_GRAD_FROM_SLAB"); return e && e[0] == '1'; }();
    return on;
}

// conv2d(reflection_pad(view), w, b) with EXPLICIT float32 weights (w_base + g.w_off, bias at w_base + g.b_off), no sampling: the two
// convolutions of a local-reparameterisation layer.  MFMA kernel when the shape is served, else the generic one (w = "mu", eval branch).
int conv_fwd_plain(const TView& v, const ConvGeom& g, const float* w_base, OutDesc od, int n, hipStream_t st)
{
    int rc = use_mfma() ? launch_conv_fwd_mfma(v, g, w_base, 0, od, n, st) : -2;
    if (rc == -2 || rc == -3) { RngKey none{}; rc = launch_conv_fwd(v, g, w_base, w_base, none, 0, od, n, st); }
    return rc;
}
int conv_bwd_data_plain(const GView& gy, const ConvGeom& g, const float* w_base, float* dxp, long long per, int n, hipStream_t st)
{
    int rc = use_mfma() ? launch_conv_bwd_data_mfma(gy, g, w_base, 0, dxp, per, n, st) : -2;
    if (rc == -2 || rc == -3) { RngKey none{}; rc = launch_conv_bwd_data(gy, g, w_base, w_base, none, 0, dxp, per, n, st); }
    return rc;
}

// Gradient wrt tensor `tid` once every consumer has written its padded input gradient: reflection-pad adjoint fold, sum over the
// consumers (an LRT consumer contributes two sources, the variance branch with the factor 2 * view(x)), LeakyReLU', BN-backward sums.
// inline_op >= 0: that consumer (a narrow 1x1 convolution) wrote no padded gradient — its backward-data is formed inside the fold from its
// output gradient gy1 and its weights w1 (launch_finalize_dx_inline1x1); -2 from there: the caller launches the consumer after all
int fold_consumers(mfvi_plan* plan, const Ctx& c, int tid, const TView& xin, float* dz, int sample_weights, int op_index, int n_samples, hipStream_t st,
                   int inline_op = -1, const GView* gy1 = nullptr, const float* w1 = nullptr, long long w1_sstride = 0)
{
    const TensorInfo& x = plan->t[tid];
    FoldSrc srcs[MAX_FOLD_SRC]; int ns = 0;
    for (int ci : x.consumers) {
        if (ci == inline_op) continue;
        const OpInfo& co = plan->ops[ci]; const int Pc = co.g.ks / 2;
        const long long per = (long long)co.g.Cin * (co.g.H + 2 * Pc) * (co.g.W + 2 * Pc);
        srcs[ns++] = FoldSrc{c.farena() + co.scratch_off, per, Pc, 0};
        if (co.d.type == MFVI_OP_CONV_LRT && sample_weights) srcs[ns++] = FoldSrc{c.farena() + co.scratch2_off, per, Pc, 1};
    }
    float* ga = (tid == plan->input) ? dz : c.farena() + x.ga_off;
    ProfScope ps(plan, op_index, PASS_FINALIZE, st);
    if (inline_op >= 0) {
        if (ns != 1) return -2;
        const OpInfo& io = plan->ops[inline_op];
        return launch_finalize_dx_inline1x1(srcs[0], *gy1, w1 + io.g.w_off, w1_sstride, io.g.Cout, xin, ga, x.numel, x.d.has_bn ? c.bsums() + x.stats_off : nullptr, n_samples, st);
    }
    return launch_finalize_dx(srcs, ns, xin, ga, x.numel, x.d.has_bn ? c.bsums() + x.stats_off : nullptr, n_samples, st);
}

RngKey base_key(uint64_t seed, uint32_t step, uint32_t k0, const int32_t* step_dev = nullptr)
{
    RngKey k; k.k0 = (uint32_t)seed; k.k1 = (uint32_t)(seed >> 32); k.stream = 0; k.sample = k0; k.step = step; k.step_dev = step_dev; return k;
}

}  // namespace

extern "C" {

int mfvi_plan_create(const mfvi_tensor_desc* tensors, int n_tensors, const mfvi_op_desc* ops, int n_ops, int input_tensor,
                     int output_tensor, int64_t n_vi, int64_t n_bn, int max_samples, mfvi_plan** plan)
{
    if (!tensors || !ops || !plan) { set_error("plan_create: null argument"); return -1; }
    mfvi_plan* p = new mfvi_plan();
    p->input = input_tensor; p->output = output_tensor; p->n_vi = n_vi; p->n_bn = n_bn; p->max_samples = max_samples;
    if (!build(*p, tensors, n_tensors, ops, n_ops)) { if (p->table_dev) (void)hipFree(p->table_dev); if (p->drop_dev) (void)hipFree(p->drop_dev); if (p->fin_dev) (void)hipFree(p->fin_dev); if (p->samp_dev) (void)hipFree(p->samp_dev); delete p; *plan = nullptr; return -1; }
    *plan = p;
    return 0;
}

void mfvi_plan_destroy(mfvi_plan* plan)
{
    if (!plan) return;
    for (auto& r : plan->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : plan->free_events) (void)hipEventDestroy(e);
    if (plan->table_dev) (void)hipFree(plan->table_dev);
    if (plan->fin_dev) (void)hipFree(plan->fin_dev);
    if (plan->x6_dev) (void)hipFree(plan->x6_dev);
    if (plan->x6b_dev) (void)hipFree(plan->x6b_dev);
    if (plan->samp_dev) (void)hipFree(plan->samp_dev);
    if (plan->drop_dev) (void)hipFree(plan->drop_dev);
    for (auto e : plan->fork_events) (void)hipEventDestroy(e);
    for (auto e : plan->fwd_events) (void)hipEventDestroy(e);
    if (plan->join_event) (void)hipEventDestroy(plan->join_event);
    for (auto e : plan->split_ev) if (e) (void)hipEventDestroy(e);
    if (plan->side) (void)hipStreamDestroy(plan->side);
    delete plan;
}

int64_t mfvi_plan_workspace_bytes(const mfvi_plan* plan) { return plan ? plan->total_bytes : -1; }

int mfvi_plan_set_side_stream(mfvi_plan* plan, int enabled)
{
    if (!plan) { set_error("set_side_stream: null plan"); return -1; }
    plan->side_enabled = enabled != 0;
    return 0;
}

int mfvi_plan_grad_split_offset(const mfvi_plan* plan, int first_op, int64_t* offset)
{
    if (!plan || !offset || first_op < 0 || first_op >= (int)plan->ops.size()) { set_error("grad_split_offset: bad arguments"); return -1; }
    if (plan->n_lrt) { set_error("grad_split_offset: plans with local-reparameterisation layers reduce d rho in one pass at the end"); return -1; }
    long long lo = plan->n_vi, head_end = 0;
    for (int i = 0; i < (int)plan->ops.size(); ++i) {
        const OpInfo& o = plan->ops[i];
        if (o.d.type != MFVI_OP_CONV) continue;
        const long long nw = (long long)o.g.Cout * o.g.Cin * o.g.ks * o.g.ks;
        const long long a = o.g.b_off >= 0 ? std::min<long long>(o.g.w_off, o.g.b_off) : o.g.w_off;
        const long long b = std::max<long long>(o.g.w_off + nw, o.g.b_off >= 0 ? o.g.b_off + o.g.Cout : 0);
        if (i >= first_op) lo = std::min(lo, a); else head_end = std::max(head_end, b);
    }
    if (head_end > lo) { set_error("grad_split_offset: the parameters of the ops >= %d are not a tail of the flat layout", first_op); return -1; }
    *offset = lo;
    return 0;
}

int mfvi_plan_set_grad_split(mfvi_plan* plan, int first_op, void* comm_stream)
{
    if (!plan) { set_error("set_grad_split: null plan"); return -1; }
    if (first_op < 0) { plan->split_op = -1; plan->split_stream = nullptr; return 0; }
    int64_t off = 0;
    if (mfvi_plan_grad_split_offset(plan, first_op, &off)) return -1;
    for (auto& e : plan->split_ev)
        if (!e) { const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming); if (rc != hipSuccess) { set_error("set_grad_split: event creation failed: %s", hipGetErrorString(rc)); return (int)rc; } }
    plan->split_op = first_op; plan->split_stream = (hipStream_t)comm_stream;
    return 0;
}

int mfvi_plan_set_step_source(mfvi_plan* plan, const int32_t* step_dev)
{
    if (!plan) { set_error("set_step_source: null plan"); return -1; }
    plan->step_dev = step_dev; plan->samp_n = 0;
    return 0;
}

int mfvi_plan_set_capture_mode(mfvi_plan* plan, int enabled)
{
    if (!plan) { set_error("set_capture_mode: null plan"); return -1; }
    plan->capture_mode = enabled != 0;
    return 0;
}

int mfvi_plan_set_param_dtype(mfvi_plan* plan, int dtype)
{
    if (!plan || (dtype != MFVI_PARAM_F32 && dtype != MFVI_PARAM_BF16)) { set_error("set_param_dtype: bad arguments"); return -1; }
    plan->param_dtype = dtype; plan->samp_n = 0;
    return 0;
}

int mfvi_plan_set_bn_eval(mfvi_plan* plan, const float* running)
{
    if (!plan) { set_error("set_bn_eval: null plan"); return -1; }
    plan->bn_eval = running;
    return 0;
}

int mfvi_plan_bn_update_running(const mfvi_plan* plan, const void* workspace, int n_samples, float momentum, float* running, void* stream)
{
    if (!plan || !workspace || !running || n_samples < 1 || n_samples > plan->max_samples || !(momentum >= 0.f && momentum <= 1.f)) {
        set_error("bn_update_running: bad arguments"); return -1; }
    if (plan->bn_eval) { set_error("bn_update_running: the plan is in BatchNorm eval mode (no batch statistics were formed)"); return -1; }
    const int rc = launch_bn_update_running(plan->table_dev, plan->n_entries, plan->max_c, (const double*)workspace, n_samples, momentum, running, (hipStream_t)stream);
    if (rc) set_error("bn_update_running: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}

int mfvi_plan_set_dropout(mfvi_plan* plan, int enabled)
{
    if (!plan) { set_error("set_dropout: null plan"); return -1; }
    plan->dropout_on = enabled != 0;
    return 0;
}

// float32 view of mu / rho for the generic kernels: the caller's arrays, or their expansion when the parameters are stored in bf16
static int generic_params(mfvi_plan* plan, const Ctx& c, const void* mu, const void* rho, const float** mu32, const float** rho32, hipStream_t st)
{
    *mu32 = static_cast<const float*>(mu); *rho32 = static_cast<const float*>(rho);
    if (plan->param_dtype != MFVI_PARAM_BF16) return 0;
    *mu32 = nullptr; *rho32 = nullptr;
    if (plan->n_generic <= 0) return 0;
    float* dst = c.farena() + plan->p32_off;
    int rc = launch_expand_bf16(mu, plan->n_vi, dst, st);
    if (!rc) rc = launch_expand_bf16(rho, plan->n_vi, dst + plan->n_vi, st);
    *mu32 = dst; *rho32 = dst + plan->n_vi;
    return rc;
}

int mfvi_forward(mfvi_plan* plan, const void* mu_v, const void* rho_v, const float* bn, const float* z, uint64_t seed, uint32_t step,
                 uint32_t k0, int n_samples, int sample_weights, void* workspace, float* out, void* stream)
{
    if (!check_call(plan, n_samples, workspace)) return -1;
    if (!mu_v || !rho_v || !z || !out || (plan->n_bn > 0 && !bn)) { set_error("forward: null pointer argument"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    Ctx c{*plan, (char*)workspace, bn, z, n_samples};
    const bool bf16 = plan->param_dtype == MFVI_PARAM_BF16;
    if (bf16 && !use_mfma()) { set_error("forward: bf16 parameters need the MFMA path (MFVI_DISABLE_MFMA is set)"); return -1; }
    if (bf16 && (((uintptr_t)mu_v | (uintptr_t)rho_v) & 7)) { set_error("forward: bf16 mu / rho must be 8-byte aligned"); return -1; }
    const float* mu = nullptr; const float* rho = nullptr;
    { const int rc = generic_params(plan, c, mu_v, rho_v, &mu, &rho, st); if (rc) { set_error("forward: bf16 expansion failed: %s", hipGetErrorString((hipError_t)rc)); return rc; } }
    // MFMA-served layers: draw every weight once per (layer, sample); without sampling the kernels read mu (stride 0)
    // (bf16 parameters: the slab also serves w = mu, as one float32 copy shared by all samples)
    const bool presample = use_mfma() && (sample_weights || bf16) && plan->n_samp > 0;
    // The forward statistics and (adjacent) the BN-backward sums of the backward pass that follows start every pass from zero.  With a weight
    // draw in front of the pass the draw's kernel clears them with its own threads (round 4: the memset was a dependent 6 us launch at the
    // head of every iteration); eval-mode BatchNorm fills the statistics in front of the draw and keeps the memset.
    const bool zero_in_draw = plan->stats_doubles && presample && !(plan->bn_eval && plan->n_entries);
    if (plan->stats_doubles) {
        if (!zero_in_draw) {
            hipError_t e = hipMemsetAsync(c.fstats(), 0, sizeof(double) * 2 * plan->stats_doubles, st);
            if (e != hipSuccess) { set_error("forward: memset failed: %s", hipGetErrorString(e)); return (int)e; }
        }
        plan->bsums_clean_ws = workspace;
    }
    const RngKey key = base_key(seed, step, k0, plan->step_dev);
    if (plan->bn_eval && plan->n_entries) {   // nn.BatchNorm2d in eval mode: the running statistics stand in for every sample's batch sums
        const int rc = launch_bn_eval_fill(plan->table_dev, plan->n_entries, plan->max_c, c.fstats(), n_samples, plan->bn_eval, st);
        if (rc) { set_error("forward: bn_eval_fill launch failed: %s", hipGetErrorString((hipError_t)rc)); return rc; }
    }
    if (plan->n_lrt && sample_weights) {      // weights of the variance convolutions of this pass
        if (!rho) { set_error("forward: local-reparameterisation layers take float32 parameters"); return -1; }
        const int rc = launch_lrt_sigma2(rho, plan->n_vi, c.farena() + plan->sig2_off, st);
        if (rc) { set_error("forward: sigma^2 launch failed: %s", hipGetErrorString((hipError_t)rc)); return rc; }
    }
    if (presample) {
        ProfScope ps(plan, -1, PASS_SAMPLE, st);
        const int rc = launch_sample_weights(plan->samp_dev, plan->n_samp, plan->samp_blocks, mu_v, rho_v, key, sample_weights ? n_samples : 1, c.wsamp(),
                                             plan->n_vi, st, bf16, sample_weights, zero_in_draw ? c.fstats() : nullptr, zero_in_draw ? 2 * plan->stats_doubles : 0);
        if (rc) { set_error("forward: sample_weights launch failed: %s", hipGetErrorString((hipError_t)rc)); return rc; }
        plan->samp_mu = mu_v; plan->samp_rho = rho_v; plan->samp_ws = workspace; plan->samp_seed = seed; plan->samp_step = step; plan->samp_k0 = k0;
        plan->samp_n = sample_weights ? n_samples : -n_samples;
    }
    if (plan->n_drop && plan->dropout_on) {      // Dropout2d factors of this pass; the backward reads them from the workspace
        const int rc = launch_dropout_masks(plan->drop_dev, plan->n_drop, key, n_samples, c.farena(), st);
        if (rc) { set_error("forward: dropout mask launch failed: %s", hipGetErrorString((hipError_t)rc)); return rc; }
    }
    bool x6_ready = false;
    if (presample) {      // weight pieces of the layers whose forward runs on the bf16x6 kernel: one launch behind the draw
        std::vector<X6SplitEntry> tab; int nb = 0;
        for (auto& o : plan->ops)
            if (o.d.type == MFVI_OP_CONV && o.x6w_off >= 0 && (o.g.tune[0] & MFVI_TUNE_X6)) {
                X6SplitEntry e; if (!x6_split_entry(o.g, o.x6w_off, &e)) continue;
                e.first_block = nb; nb += (e.units + 255) / 256; tab.push_back(e);
            }
        if (!tab.empty()) {
            hipError_t e = hipSuccess;
            if (!plan->x6_dev) e = hipMalloc((void**)&plan->x6_dev, sizeof(X6SplitEntry) * plan->ops.size());
            const bool same = tab.size() == plan->x6_uploaded.size() && memcmp(tab.data(), plan->x6_uploaded.data(), sizeof(X6SplitEntry) * tab.size()) == 0;
            if (e == hipSuccess && !same) {      // (copied from the plan's own vector: it outlives the asynchronous copy)
                plan->x6_uploaded = tab;
                e = hipMemcpyAsync(plan->x6_dev, plan->x6_uploaded.data(), sizeof(X6SplitEntry) * tab.size(), hipMemcpyHostToDevice, st);
            }
            if (e != hipSuccess) { set_error("forward: weight-piece table setup failed: %s", hipGetErrorString(e)); return (int)e; }
            const int rc = launch_x6_split_all(plan->x6_dev, (int)tab.size(), nb, c.wsamp(), sample_weights ? plan->n_vi : 0, sample_weights ? n_samples : 1, c.farena(), st);
            if (rc) { set_error("forward: weight-piece launch failed: %s", hipGetErrorString((hipError_t)rc)); return rc; }
            x6_ready = true;
        }
    }
    const float* wsrc = presample ? c.wsamp() : mu; const long long wstride = (presample && sample_weights) ? plan->n_vi : 0;
    // A skip-branch convolution (its only consumer is a later concat) on a map of up to MFVI_FWD_FORK pixels (default 128 x 128; 0 = never)
    // runs on the plan's side stream beside the down path of its scale and is joined in front of that concat: at those sizes both are
    // latency-bound launches that leave most of the chip idle (with the events on the kernels' packets: 3.306 ms per iteration without,
    // 3.282 / 3.274 / 3.279 with the maps up to 64^2 / 128^2 / 256^2).  The side stream exists once a backward pass has run.
    static const long long fwd_fork = [] { const char* e = getenv("MFVI_FWD_FORK"); return e ? atoll(e) : 16384; }();
    std::vector<int> join_at(plan->ops.size(), -1); size_t n_fev = 0;
    auto fwd_event = [&](hipEvent_t* ev) -> hipError_t {
        if (n_fev == plan->fwd_events.size()) { hipEvent_t e2; const hipError_t e = hipEventCreateWithFlags(&e2, hipEventDisableTiming); if (e != hipSuccess) return e; plan->fwd_events.push_back(e2); }
        *ev = plan->fwd_events[n_fev++]; return hipSuccess;
    };
    auto forks = [&](size_t j) {
        if (j >= plan->ops.size()) return false;
        const OpInfo& oj = plan->ops[j]; const TensorInfo& yj = plan->t[oj.d.out];
        return fwd_fork > 0 && plan->side && plan->side_enabled && oj.d.type == MFVI_OP_CONV && use_mfma() && (long long)oj.g.Ho * oj.g.Wo <= fwd_fork &&
               yj.consumers.size() == 1 && yj.consumers.front() > (int)j + 1 && plan->ops[yj.consumers.front()].d.type == MFVI_OP_CONCAT_UP;
    };
    // events on the kernels' own packets where the launch goes through mfvi_launch (as in mfvi_backward): the fork event of op i + 1 on op i's
    // launch, the join event on the forked launch itself
    static const bool on_packet_env = [] { const char* e = getenv("MFVI_FORK_ON_PACKET"); return !(e && e[0] == '0'); }();
    const bool on_packet = on_packet_env && !plan->capture_mode;
    hipEvent_t pre_ev = nullptr; size_t pre_for = (size_t)-1; bool pre_done = false; int pre_idx = -1;
    for (size_t i = 0; i < plan->ops.size(); ++i) {
        const OpInfo& o = plan->ops[i];
        const TensorInfo& y = plan->t[o.d.out];
        OutDesc od;
        od.data = (o.d.out == plan->output) ? out : c.farena() + y.act_off; od.sstride = y.numel;
        od.stats = (y.d.has_bn && !plan->bn_eval) ? c.fstats() + y.stats_off : nullptr;
        int rc;
        hipStream_t stc = st;      // stream of this op
        if (join_at[i] >= 0) {     // a forked producer of this op's input: wait for it
            const hipError_t e = hipStreamWaitEvent(st, plan->fwd_events[join_at[i]], 0);
            if (e != hipSuccess) { set_error("forward: join failed: %s", hipGetErrorString(e)); return (int)e; }
        }
        if (forks(i)) {
            hipEvent_t ef = nullptr;
            hipError_t e = hipSuccess;
            if (pre_for == i) { ef = pre_ev; if (!pre_done) e = hipEventRecord(ef, st); }      // reserved on the previous launch (recorded there, or here if that launch took another path)
            else { e = fwd_event(&ef); if (e == hipSuccess) e = hipEventRecord(ef, st); }
            if (e == hipSuccess) e = hipStreamWaitEvent(plan->side, ef, 0);
            if (e != hipSuccess) { set_error("forward: fork failed: %s", hipGetErrorString(e)); return (int)e; }
            stc = plan->side;
        }
        pre_for = (size_t)-1;
        hipEvent_t ej = nullptr; int ej_idx = -1; bool armed = false;
        if (stc != st) {           // forked: its completion event rides on its own launch
            const hipError_t e = fwd_event(&ej);
            if (e != hipSuccess) { set_error("forward: fork failed: %s", hipGetErrorString(e)); return (int)e; }
            ej_idx = (int)n_fev - 1;
            if (on_packet && plan->prof_mode != 1) { mfvi_tl_stop_event = ej; armed = true; }
        } else if (on_packet && plan->prof_mode != 1 && o.d.type != MFVI_OP_CONV_LRT && forks(i + 1)) {
            const hipError_t e = fwd_event(&pre_ev);
            if (e != hipSuccess) { set_error("forward: fork failed: %s", hipGetErrorString(e)); return (int)e; }
            pre_for = i + 1; pre_idx = (int)n_fev - 1; (void)pre_idx;
            mfvi_tl_stop_event = pre_ev; armed = true;
        }
        hipStream_t st_main = st; (void)st_main;
        {
        hipStream_t st = stc;      // (shadows the caller's stream for this op's launches)
        ProfScope ps(plan, (int)i, PASS_FWD, st);
        if (o.d.type == MFVI_OP_CONV_LRT) {
            // LRTLayer.forward (reparam_layers.py:59-72): act_mu = conv(v, mu, mu_b); training: + sqrt(1e-16 + conv(v^2, sigma^2, sigma_b^2)) * eps
            if (!mu) { set_error("forward: local-reparameterisation layers take float32 parameters"); return -1; }
            if (!sample_weights) rc = conv_fwd_plain(c.view(o.d.in0), o.g, mu, od, n_samples, st);
            else {
                OutDesc oa; oa.data = c.farena() + plan->lrt_tmp_off; oa.sstride = y.numel; oa.stats = nullptr;
                OutDesc os; os.data = c.farena() + o.s2_off; os.sstride = y.numel; os.stats = nullptr;
                TView v2 = c.view(o.d.in0); v2.act |= MFVI_ACT_SQUARE;
                rc = conv_fwd_plain(c.view(o.d.in0), o.g, mu, oa, n_samples, st);
                if (!rc) rc = conv_fwd_plain(v2, o.g, c.farena() + plan->sig2_off, os, n_samples, st);
                if (!rc) rc = launch_lrt_combine(oa.data, os.data, y.numel, y.d.C, (long long)y.d.H * y.d.W, key, o.g.layer_id, od, n_samples, st);
            }
        } else if (o.d.type == MFVI_OP_CONV) {
            mfvi_tl_x6w = (presample && o.x6w_off >= 0) ? c.farena() + o.x6w_off : nullptr; mfvi_tl_x6w_ready = x6_ready;
            mfvi_tl_family = 1;
            rc = use_mfma() ? launch_conv_fwd_mfma(c.view(o.d.in0), o.g, wsrc, wstride, od, n_samples, st) : -2;
            mfvi_tl_x6w = nullptr; mfvi_tl_x6w_ready = false;
            o.family[0] = (rc == -2 || rc == -3) ? 0 : mfvi_tl_family;
            if ((rc == -2 || rc == -3) && !mu) { set_error("forward: op %d needs the generic fp32 kernels, which bf16 parameters reach only for layers outside the sampling table (use H, W multiples of 4)", (int)i); if (plan->side) (void)hipStreamSynchronize(plan->side); return -1; }
            if (rc == -2 || rc == -3) rc = launch_conv_fwd(c.view(o.d.in0), o.g, mu, rho, key, sample_weights, od, n_samples, st);
        } else {
            TView a; if (o.d.in0 >= 0) a = c.view(o.d.in0);
            rc = launch_concat_up_fwd(o.d.in0 >= 0 ? &a : nullptr, c.view(o.d.in1), od, o.d.up_mode == MFVI_UP_NEAREST, n_samples, st);
        }
        if (rc) {      // forked skip-branch work may still be writing activations / BN statistics: join it before handing the buffers back
            mfvi_tl_stop_event = nullptr; if (rc > 0) set_error("forward: op %d launch failed: %s", (int)i, hipGetErrorString((hipError_t)rc));
            if (plan->side) (void)hipStreamSynchronize(plan->side);
            return rc; }
        }
        const bool consumed = armed && mfvi_tl_stop_event == nullptr;      // the event went out on the launch's packet
        mfvi_tl_stop_event = nullptr;
        if (stc != st) {           // forked: its completion event, waited for in front of the consumer
            if (!consumed) {
                const hipError_t e = hipEventRecord(ej, stc);
                if (e != hipSuccess) { set_error("forward: fork failed: %s", hipGetErrorString(e)); if (plan->side) (void)hipStreamSynchronize(plan->side); return (int)e; }
            }
            join_at[y.consumers.front()] = ej_idx;
        } else if (pre_for == i + 1) pre_done = consumed;
    }
    return 0;
}

int mfvi_backward(mfvi_plan* plan, const void* mu_v, const void* rho_v, const float* bn, const float* z, uint64_t seed, uint32_t step,
                  uint32_t k0, int n_samples, int sample_weights, void* workspace, const float* dout, float* dmu, float* drho,
                  float* dbn, float* dz, void* stream)
{
    if (!check_call(plan, n_samples, workspace)) return -1;
    if (!mu_v || !rho_v || !z || !dout || !dmu || !drho || (plan->n_bn > 0 && (!bn || !dbn))) { set_error("backward: null pointer argument"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    Ctx c{*plan, (char*)workspace, bn, z, n_samples};
    const bool bf16 = plan->param_dtype == MFVI_PARAM_BF16;
    if (plan->bn_eval) { set_error("backward: BatchNorm is in eval mode (mfvi_plan_set_bn_eval); the kernels implement the training-mode backward only"); return -1; }
    if (bf16 && !use_mfma()) { set_error("backward: bf16 parameters need the MFMA path (MFVI_DISABLE_MFMA is set)"); return -1; }
    if (bf16 && (((uintptr_t)mu_v | (uintptr_t)rho_v) & 7)) { set_error("backward: bf16 mu / rho must be 8-byte aligned"); return -1; }
    const float* mu = nullptr; const float* rho = nullptr;
    { const int rc = generic_params(plan, c, mu_v, rho_v, &mu, &rho, st); if (rc) { set_error("backward: bf16 expansion failed: %s", hipGetErrorString((hipError_t)rc)); return rc; } }
    if (plan->stats_doubles && plan->bsums_clean_ws != workspace) {      // a second backward after one forward (gradients accumulate): the sums start from zero again
        hipError_t e = hipMemsetAsync(c.bsums(), 0, sizeof(double) * plan->stats_doubles, st);
        if (e != hipSuccess) { set_error("backward: memset failed: %s", hipGetErrorString(e)); return (int)e; }
    }
    plan->bsums_clean_ws = nullptr;
    const RngKey key = base_key(seed, step, k0, plan->step_dev);
    if (plan->n_lrt && sample_weights) {
        if (!rho) { set_error("backward: local-reparameterisation layers take float32 parameters"); return -1; }
        hipError_t e = hipMemsetAsync(c.farena() + plan->dsig2_off, 0, sizeof(float) * plan->n_vi, st);
        if (e != hipSuccess) { set_error("backward: memset failed: %s", hipGetErrorString(e)); return (int)e; }
    }
    // the weights of this pass: the slab still holds them when the preceding forward was this very pass (same parameter
    // buffers, counters, sample range and workspace); otherwise they are re-drawn from the same counters
    const bool presample = use_mfma() && (sample_weights || bf16) && plan->n_samp > 0;
    const bool held = plan->samp_mu == mu_v && plan->samp_rho == rho_v && plan->samp_ws == workspace && plan->samp_seed == seed &&
                      plan->samp_step == step && plan->samp_k0 == k0 && plan->samp_n == (sample_weights ? n_samples : -n_samples);
    if (presample && !held) {
        ProfScope ps(plan, -1, PASS_SAMPLE, st);
        const int rc = launch_sample_weights(plan->samp_dev, plan->n_samp, plan->samp_blocks, mu_v, rho_v, key, sample_weights ? n_samples : 1, c.wsamp(),
                                             plan->n_vi, st, bf16, sample_weights);
        if (rc) { set_error("backward: sample_weights launch failed: %s", hipGetErrorString((hipError_t)rc)); return rc; }
    }
    // one use per draw: the parameters are updated in place in the same buffers, so a later backward with the same counters (a second
    // backward through a retained graph, a caller re-using a step index after an optimizer step) must re-draw from what mu / rho hold now
    plan->samp_n = 0;
    const float* wsrc = presample ? c.wsamp() : mu; const long long wstride = (presample && sample_weights) ? plan->n_vi : 0;
    bool x6b_ready = false;
    if (presample) {      // weight pieces of the layers whose backward-data runs on the bf16x6 kernel: one launch in front of the pass
        std::vector<X6BSplitEntry> tab; int nb = 0;
        for (auto& o : plan->ops)
            if (o.d.type == MFVI_OP_CONV && o.x6bw_off >= 0 && (o.g.tune[1] & MFVI_TUNE_X6) && ((o.d.in0 != plan->input) || dz != nullptr)) {
                X6BSplitEntry e; if (!x6b_split_entry(o.g, o.x6bw_off, &e)) continue;
                e.first_block = nb; nb += (e.units + e.rem_units + 255) / 256; tab.push_back(e);
            }
        if (!tab.empty()) {
            hipError_t e = hipSuccess;
            if (!plan->x6b_dev) e = hipMalloc((void**)&plan->x6b_dev, sizeof(X6BSplitEntry) * plan->ops.size());
            const bool same = tab.size() == plan->x6b_uploaded.size() && memcmp(tab.data(), plan->x6b_uploaded.data(), sizeof(X6BSplitEntry) * tab.size()) == 0;
            if (e == hipSuccess && !same) {
                if (!plan->x6b_uploaded.empty()) (void)hipStreamSynchronize(st);      // a previous upload may still be reading the vector
                plan->x6b_uploaded = tab;
                e = hipMemcpyAsync(plan->x6b_dev, plan->x6b_uploaded.data(), sizeof(X6BSplitEntry) * tab.size(), hipMemcpyHostToDevice, st);
            }
            if (e != hipSuccess) { set_error("backward: weight-piece table setup failed: %s", hipGetErrorString(e)); return (int)e; }
            const int rc = launch_x6b_split_all(plan->x6b_dev, (int)tab.size(), nb, c.wsamp(), sample_weights ? plan->n_vi : 0, sample_weights ? n_samples : 1, c.farena(), st);
            if (rc) { set_error("backward: weight-piece launch failed: %s", hipGetErrorString((hipError_t)rc)); return rc; }
            x6b_ready = true;
        }
    }
    std::vector<GradFinEntry> fin; int fin_blocks = 0;      // layers whose dW went to partial slabs in this pass
    // side stream for the backward-weight kernels (MFVI_SIDE_STREAM=0: everything on the caller's stream)
    static const bool side_on = [] { const char* e = getenv("MFVI_SIDE_STREAM"); return !(e && e[0] == '0'); }();
    // MFVI_SIDE_MAXPIX: layers with more output pixels per sample keep their backward-weight on the caller's stream (a kernel that
    // fills the chip by itself gains nothing from sharing it, and its launch duration stays meaningful for the roofline)
    static const long long side_maxpix = [] { const char* e = getenv("MFVI_SIDE_MAXPIX"); return e ? atoll(e) : (1LL << 40); }();
    hipStream_t side = st, sw = st; size_t n_fork = 0;
    if (side_on && plan->side_enabled) {
        if (!plan->side) {
            // lowest priority: the caller's stream carries the critical path, the side stream only fills what it leaves idle
            int prio_least = 0, prio_greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
            static const bool low = [] { const char* e2 = getenv("MFVI_SIDE_PRIO"); return !(e2 && e2[0] == '0'); }();
            hipError_t e = hipStreamCreateWithPriority(&plan->side, hipStreamNonBlocking, low ? prio_least : 0);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&plan->join_event, hipEventDisableTiming);
            if (e != hipSuccess) { set_error("backward: side stream setup failed: %s", hipGetErrorString(e)); return (int)e; }
        }
        side = plan->side;
    }
    // The fork event of the NEXT op's backward-weight kernel rides on the packet of this op's last launch on the caller's stream when that
    // launch goes through mfvi_launch (armed right before it; a launcher that takes another path leaves it armed and the fork falls back to
    // hipEventRecord).  MFVI_FORK_ON_PACKET=0: always hipEventRecord.
    static const bool fork_on_packet_env = [] { const char* e = getenv("MFVI_FORK_ON_PACKET"); return !(e && e[0] == '0'); }();
    const bool fork_on_packet = fork_on_packet_env && !plan->capture_mode;
    int armed_idx = -1;
    auto will_fork = [&](int j) {
        if (j < 0 || side == st || plan->ops[j].d.type != MFVI_OP_CONV) return false;
        const OpInfo& oj = plan->ops[j];
        const bool bww_only_j = (oj.d.in0 == plan->input) && dz == nullptr;
        return !bww_only_j && (long long)oj.g.Ho * oj.g.Wo <= side_maxpix;
    };
    auto arm = [&](int i_cur) -> int {       // call right before the LAST launch of op i_cur on `st`
        armed_idx = -1; mfvi_tl_stop_event = nullptr;
        if (!fork_on_packet || plan->prof_mode == 1 || !will_fork(i_cur - 1)) return 0;      // (mode 1 brackets every launch with its own events)
        if (n_fork == plan->fork_events.size()) {
            hipEvent_t ev; const hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess) { set_error("backward: event creation failed: %s", hipGetErrorString(e)); return (int)e; }
            plan->fork_events.push_back(ev);
        }
        mfvi_tl_stop_event = plan->fork_events[n_fork]; armed_idx = (int)n_fork;
        return 0;
    };
    auto settle = [&]() {                     // after that launch: consumed (the event is on the kernel's packet) or not
        if (mfvi_tl_stop_event) { mfvi_tl_stop_event = nullptr; armed_idx = -1; }
    };
    // reduction of the partial dW slabs of the layers collected in `fin` into dmu / drho, on stream fs from the device table tab
    bool bn_done = false;      // the BatchNorm parameter gradients went out with the last grad_finalize launch
    auto finalize = [&](hipStream_t fs, GradFinEntry* tab, std::vector<GradFinEntry>& uploaded, bool with_bn = false) -> int {
        if (fin.empty()) return 0;
        // longest blocks first: a block's work grows with the number of pixel strips of its layer
        std::stable_sort(fin.begin(), fin.end(), [](const GradFinEntry& a, const GradFinEntry& b) { return a.strips > b.strips; });
        fin_blocks = 0;
        for (auto& e : fin) { e.first_block = fin_blocks; fin_blocks += ((e.n_w >> 2) + ((e.n_b + 3) >> 2) + GRAD_FIN_QUADS - 1) / GRAD_FIN_QUADS; }
        ProfScope ps(plan, -1, PASS_GRAD_FINALIZE, fs);
        const bool same = fin.size() == uploaded.size() && memcmp(fin.data(), uploaded.data(), sizeof(GradFinEntry) * fin.size()) == 0;
        if (!same) {      // tilings change only when the plan is (re)tuned: each of the three tables is uploaded once in steady state
            if (getenv("MFVI_DEBUG_FIN")) for (auto& e : fin) fprintf(stderr, "fin layer %d n_w %d strips %d first_block %d\n", e.layer_id, e.n_w, e.strips, e.first_block);
            // the previous upload of this slot may still be reading the vector about to be reassigned (pageable source of an async copy)
            if (!uploaded.empty()) (void)hipStreamSynchronize(fs);
            uploaded = fin;      // (the copy reads the plan-owned vector: `fin` is reused by the caller)
            const hipError_t e = hipMemcpyAsync(tab, uploaded.data(), sizeof(GradFinEntry) * uploaded.size(), hipMemcpyHostToDevice, fs);
            if (e != hipSuccess) { uploaded.clear(); set_error("backward: gradient table upload failed: %s", hipGetErrorString(e)); return (int)e; }
        }
        // every layer of `fin` (MFMA backward-weight) is also in the sampling table (same shape conditions), so its W_k sit in the slab
        const int rc = launch_grad_finalize(tab, (int)fin.size(), fin_blocks, c.farena(), rho_v, key, sample_weights, n_samples, dmu, drho,
                                            presample && sample_weights && grad_from_slab() ? c.wsamp() : nullptr, plan->n_vi, mu_v, fs, bf16,
                                            with_bn ? plan->table_dev : nullptr, with_bn ? plan->n_entries : 0, c.bsums(), dbn);
        if (rc) { set_error("backward: grad_finalize launch failed: %s", hipGetErrorString((hipError_t)rc)); return rc; }
        if (with_bn && plan->n_entries) bn_done = true;
        fin.clear(); fin_blocks = 0;
        return 0;
    };
    // the join event rides on the side stream's last launch (the backward-weight kernel of the last op that forks)
    int last_fork_op = -1;
    for (int j = 0; j < (int)plan->ops.size(); ++j) if (will_fork(j)) { last_fork_op = j; break; }
    bool join_on_packet = false;
    for (int i = (int)plan->ops.size() - 1; i >= 0; --i) {
        const OpInfo& o = plan->ops[i];
        int rc = 0;
        if (o.d.type == MFVI_OP_CONV_LRT) {
            // autograd of LRTLayer.forward: d act_mu = dy, d act_var = dy * eps / (2 std); the two convolutions' weight gradients go to
            // d mu and (through sigma^2 = softplus(rho)^2) to d rho; their input gradients meet in the fold, the variance branch with
            // the factor 2 * view(x) of its x**2 operand.  Caller's stream throughout (an alternative estimator, not the hot path).
            const GView gy = c.gview(o.d.out, dout);
            const TView xin = c.view(o.d.in0);
            const TensorInfo& x = plan->t[o.d.in0];
            const TensorInfo& yt = plan->t[o.d.out];
            const int P = o.g.ks / 2;
            const long long per = (long long)o.g.Cin * (o.g.H + 2 * P) * (o.g.W + 2 * P);
            const bool need_dx = (o.d.in0 != plan->input) || dz != nullptr;
            if (!mu) { set_error("backward: local-reparameterisation layers take float32 parameters"); return -1; }
            RngKey none{};
            { ProfScope ps(plan, i, PASS_BWD_WEIGHT, st);
              rc = launch_conv_bwd_weight(xin, gy, o.g, rho, none, 0, dmu, drho, n_samples, st); }          // d mu, d mu_b
            if (!rc && need_dx) { ProfScope ps(plan, i, PASS_BWD_DATA, st);
              rc = conv_bwd_data_plain(gy, o.g, mu, c.farena() + o.scratch_off, per, n_samples, st); }
            if (!rc && sample_weights) {
                float* ds2 = c.farena() + plan->lrt_tmp_off;
                rc = launch_lrt_ds2(gy, c.farena() + o.s2_off, yt.numel, key, o.g.layer_id, ds2, n_samples, st);
                GView g2{}; g2.ga = ds2; g2.gstride = yt.numel; g2.y = nullptr; g2.ystride = 0; g2.C = yt.d.C; g2.H = yt.d.H; g2.W = yt.d.W;
                g2.stats = nullptr; g2.bsums = nullptr; g2.gamma = nullptr; g2.eps = 0.f; g2.drop = nullptr;
                TView v2 = xin; v2.act |= MFVI_ACT_SQUARE;
                float* dsig2 = c.farena() + plan->dsig2_off;
                if (!rc) { ProfScope ps(plan, i, PASS_BWD_WEIGHT, st);
                  rc = launch_conv_bwd_weight(v2, g2, o.g, rho, none, 0, dsig2, dsig2, n_samples, st); }    // d sigma^2 (weights and bias variance)
                if (!rc && need_dx) { ProfScope ps(plan, i, PASS_BWD_DATA, st);
                  rc = conv_bwd_data_plain(g2, o.g, c.farena() + plan->sig2_off, c.farena() + o.scratch2_off, per, n_samples, st); }
            }
            if (!rc && need_dx && x.consumers.front() == i) rc = fold_consumers(plan, c, o.d.in0, xin, dz, sample_weights, i, n_samples, st);
        } else
        if (o.d.type == MFVI_OP_CONV) {
            const GView gy = c.gview(o.d.out, dout);
            const TView xin = c.view(o.d.in0);
            // Layers that read the network input have no backward-data (unless dz is asked for): their backward-weight kernel is all the
            // caller's stream would do for them, so it runs there — at the end of the pass the side stream is still working off the last
            // layers' kernels while the caller's stream would sit idle (a ~100 us tail of three serial launches otherwise).
            const bool bww_only = (o.d.in0 == plan->input) && dz == nullptr;
            sw = (!bww_only && (long long)o.g.Ho * o.g.Wo <= side_maxpix) ? side : st;
            if (sw != st) {      // fork: everything this layer's backward-weight reads (dy, BN-backward sums) is final at this point of `st`
                if (n_fork == plan->fork_events.size()) {
                    hipEvent_t ev; const hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
                    if (e != hipSuccess) { set_error("backward: event creation failed: %s", hipGetErrorString(e)); return (int)e; }
                    plan->fork_events.push_back(ev);
                }
                hipError_t e = hipSuccess;
                if (armed_idx != (int)n_fork) e = hipEventRecord(plan->fork_events[n_fork], st);      // else: already on the previous launch's packet
                if (e == hipSuccess) e = hipStreamWaitEvent(sw, plan->fork_events[n_fork], 0);
                if (e != hipSuccess) { set_error("backward: fork failed: %s", hipGetErrorString(e)); return (int)e; }
                ++n_fork;
            }
            armed_idx = -1;
            { ProfScope ps(plan, i, PASS_BWD_WEIGHT, sw);
              int strips = 0;
              const bool arm_join = fork_on_packet && plan->prof_mode != 1 && sw != st && i == last_fork_op;
              if (arm_join) mfvi_tl_stop_event = plan->join_event;
              mfvi_tl_family = 1;
              rc = use_mfma() ? launch_conv_bwd_weight_mfma(xin, gy, o.g, BwwPart{c.farena() + o.part_off, o.part_stride, o.max_strips}, &strips, n_samples, sw) : -2;
              o.family[2] = (rc == -2 || rc == -3) ? 0 : mfvi_tl_family;
              if (arm_join) { join_on_packet = mfvi_tl_stop_event == nullptr; mfvi_tl_stop_event = nullptr; }
              if (rc == 0) {
                  GradFinEntry e{};
                  e.w_off = o.g.w_off; e.b_off = o.g.b_off; e.part_off = o.part_off; e.stride = o.part_stride;
                  e.n_w = o.g.Cout * o.g.Cin * o.g.ks * o.g.ks; e.n_b = o.g.b_off >= 0 ? o.g.Cout : 0; e.strips = strips; e.layer_id = o.g.layer_id;
                  e.first_block = fin_blocks;
                  fin_blocks += ((e.n_w >> 2) + ((e.n_b + 3) >> 2) + GRAD_FIN_QUADS - 1) / GRAD_FIN_QUADS;
                  fin.push_back(e);
              }
              if ((rc == -2 || rc == -3) && !mu) { set_error("backward: op %d needs the generic fp32 kernels, which bf16 parameters reach only for layers outside the sampling table", i); rc = -1; }
              if (rc == -2 || rc == -3) rc = launch_conv_bwd_weight(xin, gy, o.g, rho, key, sample_weights, dmu, drho, n_samples, sw); }
            const bool need_dx = (o.d.in0 != plan->input) || dz != nullptr;
            if (!rc && need_dx) {
                const int P = o.g.ks / 2;
                const long long per = (long long)o.g.Cin * (o.g.H + 2 * P) * (o.g.W + 2 * P);
                const TensorInfo& x = plan->t[o.d.in0];
                bool folded = false;
                if ((o.g.ks == 1 || (o.g.ks == 3 && o.g.stride == 1 && fold_fusion3_on())) && x.consumers.size() == 1 && use_mfma() && fold_fusion_on()) {
                    // the conv's input feeds nothing else: backward-data with the fold in its epilogue (no scratch round trip, no finalize_dx
                    // launch); 3x3 stride-1 layers compute on the un-padded domain with the reflection adjoint on the pixel operand
                    FoldFuse ff; ff.x = xin; ff.ga = (o.d.in0 == plan->input) ? dz : c.farena() + x.ga_off; ff.ga_sstride = x.numel;
                    ff.bsums = x.d.has_bn ? c.bsums() + x.stats_off : nullptr;
                    ProfScope ps(plan, i, PASS_BWD_DATA, st);
                    { const int ra = arm(i); if (ra) return ra; }
                    mfvi_tl_family = 1;
                    mfvi_tl_x6bw = (presample && o.x6bw_off >= 0) ? c.farena() + o.x6bw_off : nullptr; mfvi_tl_x6bw_ready = x6b_ready;
                    const int r2 = launch_conv_bwd_data_mfma(gy, o.g, wsrc, wstride, nullptr, 0, n_samples, st, &ff);
                    mfvi_tl_x6bw = nullptr; mfvi_tl_x6bw_ready = false;
                    settle();
                    if (r2 == 0) o.family[1] = mfvi_tl_family;
                    if (r2 == 0) folded = true; else { armed_idx = -1; if (r2 != -2 && r2 != -3) rc = r2; }
                }
                const bool fold_here = x.consumers.front() == i;
                // The tensor's other consumer has written its padded gradient and this one is a narrow 1x1 convolution (the 4-channel skip
                // branch of a down-path tensor): no launch of its own — its backward-data is formed inside the fold (elementwise.hip,
                // finalize_dx_vec1_kernel).  MFVI_FUSE_SKIP_BWD=0: the separate launch (A/B, parity cross-checks).
                static const bool fuse_skip = [] { const char* e = getenv("MFVI_FUSE_SKIP_BWD"); return !(e && e[0] == '0'); }();
                if (!rc && !folded && fold_here && fuse_skip && o.d.type == MFVI_OP_CONV && o.g.ks == 1 && o.g.stride == 1 && o.g.Cout <= 8 && x.consumers.size() == 2 &&
                    plan->ops[x.consumers.back()].d.type == MFVI_OP_CONV && use_mfma() &&
                    (presample ? (!(o.g.Cin & 3) && !(o.g.w_off & 3) && o.g.Cin <= MFVI_MAX_C) : (!sample_weights && mu != nullptr))) {      // (its weights are in the slab, or w = mu)
                    ProfScope ps(plan, i, PASS_BWD_DATA, st);      // (booked on the op's backward-data slot: the fold now holds both)
                    { const int ra = arm(i); if (ra) return ra; }
                    const int r2 = fold_consumers(plan, c, o.d.in0, xin, dz, sample_weights, i, n_samples, st, i, &gy, wsrc, wstride);
                    settle();
                    if (r2 == 0) { folded = true; o.family[1] = 5; }
                    else { armed_idx = -1; if (r2 != -2) rc = r2; }
                }
                if (!rc && !folded) {
                  ProfScope ps(plan, i, PASS_BWD_DATA, st);
                  if (!fold_here) { const int ra = arm(i); if (ra) return ra; }      // no fold behind it: this is the op's last launch on `st`
                  mfvi_tl_family = 1;
                  rc = use_mfma() ? launch_conv_bwd_data_mfma(gy, o.g, wsrc, wstride, c.farena() + o.scratch_off, per, n_samples, st) : -2;
                  o.family[1] = (rc == -2 || rc == -3) ? 0 : mfvi_tl_family;
                  if (!fold_here) { settle(); if (rc) armed_idx = -1; }
                  if ((rc == -2 || rc == -3) && !mu) { set_error("backward: op %d needs the generic fp32 kernels, which bf16 parameters reach only for layers outside the sampling table", i); rc = -1; }
                  if (rc == -2 || rc == -3) rc = launch_conv_bwd_data(gy, o.g, mu, rho, key, sample_weights, c.farena() + o.scratch_off, per, n_samples, st); }
                if (!rc && !folded && fold_here) {         // all consumers of in0 have run: fold + act' + BN sums
                    { const int ra = arm(i); if (ra) return ra; }
                    rc = fold_consumers(plan, c, o.d.in0, xin, dz, sample_weights, i, n_samples, st);
                    settle();
                }
            }
        } else {
            const GView gc = c.gview(o.d.out, dout);
            const TensorInfo& b = plan->t[o.d.in1];
            TView a; float* ga_a = nullptr; long long sa = 0; double* bs_a = nullptr;
            if (o.d.in0 >= 0) {
                const TensorInfo& ta = plan->t[o.d.in0];
                a = c.view(o.d.in0); ga_a = c.farena() + ta.ga_off; sa = ta.numel; bs_a = ta.d.has_bn ? c.bsums() + ta.stats_off : nullptr;
            }
            ProfScope ps(plan, i, PASS_CONCAT_BWD, st);
            { const int ra = arm(i); if (ra) return ra; }
            rc = launch_concat_up_bwd(gc, o.d.in0 >= 0 ? &a : nullptr, ga_a, sa, bs_a, c.view(o.d.in1), c.farena() + b.ga_off, b.numel,
                                      b.d.has_bn ? c.bsums() + b.stats_off : nullptr, o.d.up_mode == MFVI_UP_NEAREST, n_samples, st);
            settle();
        }
        if (rc) {
            mfvi_tl_stop_event = nullptr;
            if (rc > 0) set_error("backward: op %d launch failed: %s", i, hipGetErrorString((hipError_t)rc));
            if (side != st) (void)hipStreamSynchronize(side);   // leave no side-stream work behind a failed call
            return rc;
        }
        if (i == plan->split_op && plan->split_stream && plan->split_stream != st) {
            // gradient split: every kernel that writes a weight gradient of the ops >= i (partial slabs on the side stream, generic
            // kernels' atomics on either stream) has been enqueued; the exchange stream waits for them and reduces that group now
            hipError_t e = hipEventRecord(plan->split_ev[0], st);
            if (e == hipSuccess) e = hipStreamWaitEvent(plan->split_stream, plan->split_ev[0], 0);
            if (e == hipSuccess && side != st) { e = hipEventRecord(plan->split_ev[1], side); if (e == hipSuccess) e = hipStreamWaitEvent(plan->split_stream, plan->split_ev[1], 0); }
            if (e != hipSuccess) { set_error("backward: gradient split failed: %s", hipGetErrorString(e)); if (side != st) (void)hipStreamSynchronize(side); return (int)e; }
            const int r2 = finalize(plan->split_stream, plan->fin_dev + plan->n_conv, plan->fin_uploaded_early);
            if (r2) { if (side != st) (void)hipStreamSynchronize(side); return r2; }
        }
    }
    if (side != st) {        // join: grad_finalize (and the caller) see every partial slab / accumulated gradient
        hipError_t e = join_on_packet ? hipSuccess : hipEventRecord(plan->join_event, side);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, plan->join_event, 0);
        if (e != hipSuccess) { set_error("backward: join failed: %s", hipGetErrorString(e)); return (int)e; }
    }
    {   // the rest (or all) of the layers: the late group of a split pass has its own table slot
        const bool was_split = plan->split_op >= 0 && plan->split_stream && plan->split_stream != st;
        // (the BatchNorm parameter gradients ride on this launch: every fold that feeds the BN-backward sums ran on `st` in front of it.
        //  Not with local-reparameterisation layers: their d rho kernel in between touches neither, but keeps the old order for its tests)
        const bool with_bn = plan->n_entries > 0 && plan->n_lrt == 0;
        const int rc = was_split ? finalize(st, plan->fin_dev + 2 * plan->n_conv, plan->fin_uploaded_late, with_bn) : finalize(st, plan->fin_dev, plan->fin_uploaded, with_bn);
        if (rc) return rc; }
    if (plan->n_lrt && sample_weights) {      // d rho += d sigma^2 * 2 softplus(rho) sigmoid(rho)
        const int rc = launch_lrt_drho(c.farena() + plan->dsig2_off, rho, plan->n_vi, drho, st);
        if (rc) { set_error("backward: lrt_drho launch failed: %s", hipGetErrorString((hipError_t)rc)); return rc; }
    }
    if (plan->n_entries && !bn_done) {
        const int rc = launch_bn_param_grads(plan->table_dev, plan->n_entries, plan->max_c, c.bsums(), n_samples, dbn, st);
        if (rc) { set_error("backward: bn_param_grads launch failed: %s", hipGetErrorString((hipError_t)rc)); return rc; }
    }
    return 0;
}

int mfvi_plan_read_tensor(const mfvi_plan* plan, const void* workspace, int tensor_id, int sample, int which, void* dst, void* stream)
{
    if (!plan || !workspace || !dst) { set_error("read_tensor: null argument"); return -1; }
    if (tensor_id < 0 || tensor_id >= (int)plan->t.size() || tensor_id == plan->input || tensor_id == plan->output) {
        set_error("read_tensor: tensor %d is not a workspace tensor", tensor_id); return -1; }
    if (sample < 0 || sample >= plan->max_samples) { set_error("read_tensor: bad sample"); return -1; }
    const TensorInfo& t = plan->t[tensor_id];
    const char* ws = (const char*)workspace;
    const void* src; size_t bytes;
    if (which == 0 || which == 1) {
        const float* base = (const float*)(ws + plan->float_base) + (which == 0 ? t.act_off : t.ga_off);
        src = base + (long long)sample * t.numel; bytes = sizeof(float) * t.numel;
    } else if (which == 2 || which == 3) {
        if (!t.d.has_bn) { set_error("read_tensor: tensor %d has no BatchNorm", tensor_id); return -1; }
        const double* base = (const double*)ws + (which == 3 ? plan->stats_doubles : 0) + t.stats_off;
        src = base + (long long)sample * t.d.C * 2; bytes = sizeof(double) * t.d.C * 2;
    } else { set_error("read_tensor: bad selector %d", which); return -1; }
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("read_tensor: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

int mfvi_plan_profile(mfvi_plan* plan, int mode, int op, int pass)
{
    if (!plan || mode < 0 || mode > 2) { set_error("plan_profile: bad arguments"); return -1; }
    plan->prof_mode = mode; plan->prof_op = op; plan->prof_pass = pass;
    return 0;
}

int mfvi_plan_profile_read(mfvi_plan* plan, int capacity, int* n_records, int* ops, int* passes, float* ms)
{
    if (!plan || capacity < 0 || !n_records) { set_error("plan_profile_read: bad arguments"); return -1; }
    int n = 0;
    for (auto& r : plan->recs) {
        float t = 0.f;
        hipError_t e = hipEventSynchronize(r.b);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, r.a, r.b);
        if (e != hipSuccess) { set_error("plan_profile_read: %s", hipGetErrorString(e)); return (int)e; }
        if (n < capacity) { ops[n] = r.op; passes[n] = r.pass; ms[n] = t; }
        ++n;
        plan->free_events.push_back(r.a); plan->free_events.push_back(r.b);
    }
    plan->recs.clear();
    *n_records = n;      /* may exceed capacity: only the first `capacity` were written */
    return 0;
}

int mfvi_plan_last_kernel(const mfvi_plan* plan, int op, int which)
{
    if (!plan || op < 0 || op >= (int)plan->ops.size() || which < 0 || which > 2 || plan->ops[op].d.type != MFVI_OP_CONV) return -1;
    return plan->ops[op].family[which];
}

int mfvi_plan_get_tune(const mfvi_plan* plan, int op, int which)
{
    if (!plan || op < 0 || op >= (int)plan->ops.size() || which < 0 || which > 2 || plan->ops[op].d.type != MFVI_OP_CONV) return -1;
    return plan->ops[op].g.tune[which];
}

int mfvi_plan_set_tune(mfvi_plan* plan, int op, int which, int tune)
{
    if (!plan || op < 0 || op >= (int)plan->ops.size() || which < 0 || which > 2 || tune < 0 || plan->ops[op].d.type != MFVI_OP_CONV) {
        set_error("plan_set_tune: bad arguments"); return -1; }
    plan->ops[op].g.tune[which] = tune;
    return 0;
}

int mfvi_plan_autotune(mfvi_plan* plan, const void* mu, const void* rho, const float* bn, const float* z, int n_samples,
                       void* workspace, float* out_scratch, float* grad_scratch, void* stream)
{
    if (!check_call(plan, n_samples, workspace)) return -1;
    if (!mu || !rho || !z || !out_scratch || !grad_scratch || (plan->n_bn > 0 && !bn)) { set_error("autotune: null pointer argument"); return -1; }
    { const char* e = getenv("MFVI_AUTOTUNE"); if ((e && e[0] == '0') || !use_mfma()) return 0; }
    hipStream_t st = (hipStream_t)stream;
    const long long n_out = plan->t[plan->output].numel * n_samples;
    float* out = out_scratch; float* dout = out_scratch + n_out;
    float* dmu = grad_scratch; float* drho = dmu + plan->n_vi; float* dbn = drho + plan->n_vi;
    // every tensor, statistic and gradient the kernels read holds finite data: one real forward + backward
    int rc = mfvi_forward(plan, mu, rho, bn, z, 1, 0, 0, n_samples, 1, workspace, out, stream);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(dout, out, sizeof(float) * n_out, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(grad_scratch, 0, sizeof(float) * (2 * plan->n_vi + plan->n_bn), st);
    if (e != hipSuccess) { set_error("autotune: %s", hipGetErrorString(e)); return (int)e; }
    rc = mfvi_backward(plan, mu, rho, bn, z, 1, 0, 0, n_samples, 1, workspace, dout, dmu, drho, dbn, nullptr, stream);
    if (rc) return rc;
    // a cold GPU ramps its clocks over the first ~100 ms of work: candidates timed during the ramp would look slow and the
    // choice would depend on their order, so run the real passes until the device has been busy for a while
    for (int warm = 0; warm < 24 && !rc; ++warm) {
        rc = mfvi_forward(plan, mu, rho, bn, z, 1, 0, 0, n_samples, 1, workspace, out, stream);
        if (!rc) rc = mfvi_backward(plan, mu, rho, bn, z, 1, 0, 0, n_samples, 1, workspace, dout, dmu, drho, dbn, nullptr, stream);
    }
    if (rc) return rc;
    Ctx c{*plan, (char*)workspace, bn, z, n_samples};
    const RngKey key = base_key(1, 0, 0);
    hipEvent_t ea, eb;
    if (hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) { set_error("autotune: hipEventCreate failed"); return -1; }
    const int reps = 6;
    for (size_t i = 0; i < plan->ops.size(); ++i) {
        OpInfo& o = plan->ops[i];
        if (o.d.type != MFVI_OP_CONV) continue;
        const TensorInfo& y = plan->t[o.d.out];
        OutDesc od; od.data = (o.d.out == plan->output) ? out : c.farena() + y.act_off; od.sstride = y.numel;
        od.stats = y.d.has_bn ? c.fstats() + y.stats_off : nullptr;
        const GView gy = c.gview(o.d.out, dout);
        const int P = o.g.ks / 2;
        const long long per = (long long)o.g.Cin * (o.g.H + 2 * P) * (o.g.W + 2 * P);
        const TView xin = c.view(o.d.in0);
        for (int which = 0; which < 3; ++which) {
            if (which == 1 && o.d.in0 == plan->input) continue;
            int strips_used = 0;
            const float* mu32 = plan->param_dtype == MFVI_PARAM_F32 ? static_cast<const float*>(mu) : nullptr;
            const float* rho32 = plan->param_dtype == MFVI_PARAM_F32 ? static_cast<const float*>(rho) : nullptr;
            const bool tiny = mu32 && (long long)o.g.Cout * o.g.Cin * o.g.ks * o.g.ks <= MFVI_INKERNEL_MAX_W;
            auto launch = [&]() {
                if (o.g.tune[which] & MFVI_TUNE_GENERIC) {      // in-kernel eps: the generic kernels, exactly as mfvi_forward / mfvi_backward reach them
                    if (which == 0) return launch_conv_fwd(xin, o.g, mu32, rho32, key, 1, od, n_samples, st);
                    if (which == 1) return launch_conv_bwd_data(gy, o.g, mu32, rho32, key, 1, c.farena() + o.scratch_off, per, n_samples, st);
                    return launch_conv_bwd_weight(xin, gy, o.g, rho32, key, 1, dmu, drho, n_samples, st);
                }
                if (which == 0) {
                    mfvi_tl_x6w = o.x6w_off >= 0 ? c.farena() + o.x6w_off : nullptr;
                    const int r0 = launch_conv_fwd_mfma(xin, o.g, c.wsamp(), plan->n_vi, od, n_samples, st);
                    mfvi_tl_x6w = nullptr;
                    return r0;
                }
                if (which == 1) {
                    const TensorInfo& x = plan->t[o.d.in0];
                    if ((o.g.ks == 1 || (o.g.ks == 3 && o.g.stride == 1 && fold_fusion3_on())) && x.consumers.size() == 1 && fold_fusion_on() && o.d.in0 != plan->input) {
                        // the fused-fold variant mfvi_backward will launch (it accumulates into the BN-backward sums: contents undefined afterwards)
                        FoldFuse ff; ff.x = xin; ff.ga = c.farena() + x.ga_off; ff.ga_sstride = x.numel;
                        ff.bsums = x.d.has_bn ? c.bsums() + x.stats_off : nullptr;
                        mfvi_tl_x6bw = o.x6bw_off >= 0 ? c.farena() + o.x6bw_off : nullptr; mfvi_tl_x6bw_ready = false;      // (the launcher splits this layer's weights itself)
                        const int r2 = launch_conv_bwd_data_mfma(gy, o.g, c.wsamp(), plan->n_vi, nullptr, 0, n_samples, st, &ff);
                        mfvi_tl_x6bw = nullptr;
                        if (r2 != -2) return r2;
                    }
                    return launch_conv_bwd_data_mfma(gy, o.g, c.wsamp(), plan->n_vi, c.farena() + o.scratch_off, per, n_samples, st);
                }
                return launch_conv_bwd_weight_mfma(xin, gy, o.g, BwwPart{c.farena() + o.part_off, o.part_stride, o.max_strips}, &strips_used, n_samples, st);
            };
            // candidate tilings: fwd / bwd-data (mf, th, T) = fragments x tile rows x tiles per block;
            //                    bwd-weight (nb, waves, target/256) = input tiles per block x waves x block-count target
            std::vector<int> cands;
            if (which < 2) {
                for (int th : {8, 16, 8 | 128, 16 | 128, 4 | 128, 2 | 128}) for (int mf = 1; mf <= 4; ++mf) for (int T = 1; T <= 8; T *= 2) cands.push_back(mf | th << 8 | T << 16);
                // backward-data of the 4 + 16n-channel concat layers: the last 4 output channels on the 4x4x1 matrix instruction (th bit 64)
                if (which == 1 && (o.g.Cin & 15) == 4) for (int mf : {1, 2, 4}) for (int T = 1; T <= 8; T *= 2) cands.push_back(mf | (8 | 64) << 8 | T << 16);
                // row-phase kernels (conv_rp.hip) for 3x3 stride-1 layers on maps whose width is a multiple of 64:
                // (mf, rows per wave, 4 extra channels on the 4x4x1 instruction, tiles per block); -3 = not valid for the shape
                if (o.g.ks == 3 && o.g.stride == 1 && ((o.g.W & 63) == 0 || o.g.W == 32 || o.g.W == 16) && rp_default_tune(o.g, which, n_samples))
                    for (int mf : {1, 2, 4}) for (int r : {1, 2, 4}) for (int rem = 0; rem <= ((which == 1 && (o.g.Cin & 15) == 4) ? 1 : 0); ++rem)
                        for (int T = 1; T <= 8; T *= 2) cands.push_back(mf | r << 8 | rem << 12 | T << 16 | MFVI_TUNE_RP);
                if (o.g.ks == 3 && o.g.stride == 1 && o.g.W == 16 && rp_default_tune(o.g, which, n_samples))      // 16-wide maps: 2 / 4 k-steps per stage
                    for (int mf : {1, 2}) for (int ks : {2, 4}) for (int rem = 0; rem <= ((which == 1 && (o.g.Cin & 15) == 4) ? 1 : 0); ++rem)
                        cands.push_back(mf | 1 << 8 | rem << 12 | ks << 13 | 1 << 16 | MFVI_TUNE_RP);
                // small-map forward (conv_small.hip): one stage, the block's whole reduction in LDS
                if (which <= 1 && o.g.ks == 3 && o.g.stride == 1 && o.g.W <= 16) cands.push_back(1 | MFVI_TUNE_SM);
                // streaming forward of the narrow 1x1 layers (conv_1x1.hip, conv1_stream_kernel): at most 32 output channels
                if (which == 0 && o.g.ks == 1 && o.g.stride == 1 && o.g.Cout <= 32 && (o.g.Cin & 3) == 0 && o.g.Cin <= 64 && (((long long)o.g.H * o.g.W) & 63) == 0)
                    cands.push_back(1 | MFVI_TUNE_ST);
                // one-stage 1x1 kernel (conv_1x1.hip): the `up` 1x1 layers of 32 ... 128 channels; same tune bit
                if (which <= 1 && o.g.ks == 1 && o.g.stride == 1 && (o.g.Cin & 15) == 0 && (o.g.Cout & 15) == 0 && o.g.Cin <= 128 && o.g.Cout <= 128
                    && (((long long)o.g.H * o.g.W) & 63) == 0) cands.push_back(1 | MFVI_TUNE_SM);
                // bf16x6 forward (conv_x6.hip): output fragments per block, 8 output rows per block
                if (which == 0 && o.x6w_off >= 0) for (int mf : {1, 2}) for (int T = 1; T <= 16; T *= 2) cands.push_back(mf | 8 << 8 | T << 16 | MFVI_TUNE_X6);
                if (which == 0 && o.x6w_off >= 0 && (o.g.Cin & 31) == 4) for (int T = 1; T <= 16; T *= 2) cands.push_back(1 | 8 << 8 | 1 << 12 | T << 16 | MFVI_TUNE_X6);      // remainder plane on the last group's pass
                // bf16x6 backward-data with the fold (conv_bwd_x6.hip): strips per block; rows per strip follow the output-channel count
                if (which == 1 && o.x6bw_off >= 0) for (int T : {1, 2, 4, 8, 16, 32}) cands.push_back(T | (o.g.Cout == 16 ? 8 : o.g.Cout == 32 ? 4 : 2) << 8 | MFVI_TUNE_X6);
                if (which == 1 && o.x6bw_off >= 0 && x6s_shape_ok(o.g)) for (int T : {2, 4, 8, 16, 32}) cands.push_back(T | 8 << 8 | 1 << 16 | MFVI_TUNE_X6);      // strip-resident form (conv_bwd_x6s.hip)
            }
            else {
                for (int nb = 1; nb <= 3; ++nb) for (int nw : {4, 8, 9}) for (int tb = 1; tb <= 8; tb *= 2) cands.push_back(nb | nw << 8 | tb << 16);
                for (int tb = 1; tb <= 8; tb *= 2) cands.push_back(2 | 10 << 8 | tb << 16);      // fragment-split variant (3x3 stride 1, full-width tiles)
                if (o.g.ks == 3 && o.g.stride == 1 && (o.g.W & 31) == 0)                         // bf16x6 kernel (conv_bww_x6.hip)
                    for (int cof = 1; cof <= 2; ++cof) for (int tb = 1; tb <= 4; tb *= 2) cands.push_back(cof | 11 << 8 | tb << 16);
            }
            if (tiny && which != 1) cands.push_back(MFVI_TUNE_GENERIC);      // (backward-data: the fused fold of the matrix-core path is not what the generic kernel replaces)
            int best = 0; float best_ms = 1e30f;
            for (int cand : cands) {
                o.g.tune[which] = cand;
                rc = launch();                                   // warm-up; -2/-3: shape or tiling not served
                if (rc == -2) break;
                if (rc == -3) continue;
                if (rc) { set_error("autotune: op %d launch failed: %s", (int)i, rc > 0 ? hipGetErrorString((hipError_t)rc) : "bad arguments"); goto done; }
                float ms = 1e30f;
                for (int trial = 0; trial < 3 && !rc; ++trial) {      // best of three timings of `reps` launches: the choice must not flip on noise
                    (void)hipEventRecord(ea, st);
                    for (int r = 0; r < reps && !rc; ++r) rc = launch();
                    (void)hipEventRecord(eb, st);
                    float t_ms = 0.f;
                    e = hipEventSynchronize(eb);
                    if (e == hipSuccess) e = hipEventElapsedTime(&t_ms, ea, eb);
                    if (e != hipSuccess) break;
                    if (t_ms < ms) ms = t_ms;
                }
                if (rc || e != hipSuccess) { set_error("autotune: op %d timing failed: %s", (int)i, hipGetErrorString(rc ? (hipError_t)rc : e)); rc = rc ? rc : (int)e; goto done; }
                // backward-weight: every extra pixel strip is one more slab grad_finalize has to read (~2 TB/s there)
                if (which == 2) ms += reps * (float)((double)strips_used * n_samples * o.part_stride * 4.0 / 2.0e12 * 1e3);
                if (ms < best_ms) { best_ms = ms; best = cand; }
            }
            o.g.tune[which] = best;
            rc = 0;
        }
    }
done:
    plan->bsums_clean_ws = nullptr;      // the timed launches accumulated into the BN-backward sums
    (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    if (rc) for (auto& o : plan->ops) { o.g.tune[0] = 0; o.g.tune[1] = 0; o.g.tune[2] = 0; }
    return rc;
}

const char* mfvi_last_error(void) { return g_err; }
int mfvi_abi_version(void) { return MFVI_ABI_VERSION; }

}  // extern "C"
