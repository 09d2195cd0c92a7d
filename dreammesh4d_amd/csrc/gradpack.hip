// gradpack.hip -- pack / unpack of the data-parallel gradient message (gfx950).
//
// The one exchange step of the path is an all-reduce of the parameter gradients.  The
// message is a flat float32 buffer of up to 64 segments: a segment is a whole gradient tensor or, for the
// HexPlane spatial planes, only the elements the (static, rank-identical) graph nodes touch.  One launch packs
// all segments, one launch unpacks them (and applies the 1/world scale) -- instead of ~80 torch copy / index
// kernels per step.
#include <string.h>

#include <mutex>

#include "common.h"
#include "../../include/dm4d.h"

namespace dm4d {

constexpr int kMaxSeg = DM4D_MAX_GRAD_SEGMENTS;

struct PackDesc {
    int n_seg;
    float *grad[kMaxSeg];              // gradient tensor of the segment (pack: may be NULL = zeros)
    const long long *index[kMaxSeg];   // NULL: dense, else the flat element indices exchanged
    long long count[kMaxSeg];          // elements of the segment in the message
    long long offset[kMaxSeg];         // start of the segment in the message
};

__global__ __launch_bounds__(256) void k_grad_pack(PackDesc d, float *__restrict__ flat)
{
    const int k = blockIdx.y;
    const long long n = d.count[k];
    const float *__restrict__ src = d.grad[k];
    const long long *__restrict__ ix = d.index[k];
    float *__restrict__ dst = flat + d.offset[k];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dst[i] = src ? (ix ? src[ix[i]] : src[i]) : 0.f;
}

__global__ __launch_bounds__(256) void k_grad_unpack(PackDesc d, const float *__restrict__ flat, float scale)
{
    const int k = blockIdx.y;
    const long long n = d.count[k];
    float *__restrict__ dst = d.grad[k];
    const long long *__restrict__ ix = d.index[k];
    const float *__restrict__ src = flat + d.offset[k];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = src[i] * scale;
        if (ix) dst[ix[i]] = v;
        else dst[i] = v;
    }
}

// ---------------------------------------------------------------------------------------- AdamW in message space
// The optimiser step of the path (AdamW(betas (0.9, 0.99), eps 1e-15) over the deformation network,
// geometry/sugar.py:406-416) for the elements that can RECEIVE gradient -- the message of the exchange step: the MLP, the time
// planes, and of the 134 MB of spatial planes only the texels the static nodes touch (3.4 M of 35.76 M elements at the shipped
// size).  Everything else has zero gradient and zero moments for ever: its whole update is the weight decay, which the caller
// keeps as a pending factor per group (distributed.ShardedAdamW.materialize).  The dense step streams 35.76 M x (parameter,
// gradient, two moments, three writes) = 1 GB per iteration (0.2 ms); this one 32 bytes x 3.4 M.
//   k_adamw_scalars (one thread): the step counter (device: a step skipped by found_inf does not advance the bias corrections),
//   the two bias corrections in float64, the pending decay factors.
//   k_adamw_message: torch/optim/adamw.py::_single_tensor_adamw operation for operation per element, gathered from and scattered
//   to the parameter / gradient STORAGE through the segment's index list.
struct AdamDesc {
    PackDesc seg;                       // grad[]: gradient storage of the segment (NULL: zeros)
    float *param[kMaxSeg];
    int group[kMaxSeg];
    float lr[8];
    int n_groups;
    float beta1, beta2, eps, weight_decay, grad_scale;
    float *exp_avg, *exp_avg_sq;
    double *step, *pending_decay;
    const float *found_inf;
    float *scal;                        // [0] bias correction 1, [1] sqrt(bias correction 2), [2] 1 = apply / 0 = skip
};

__global__ void k_adamw_scalars(AdamDesc d)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool keep = !d.found_inf || *d.found_inf == 0.f;
    const double step = *d.step + (keep ? 1.0 : 0.0);
    *d.step = step;
    d.scal[0] = (float)(1.0 - pow((double)d.beta1, step));
    d.scal[1] = (float)sqrt(1.0 - pow((double)d.beta2, step));
    d.scal[2] = keep ? 1.f : 0.f;
    if (keep && d.pending_decay)
        for (int g = 0; g < d.n_groups; ++g) d.pending_decay[g] *= 1.0 - (double)d.lr[g] * (double)d.weight_decay;
}

__global__ __launch_bounds__(256) void k_adamw_message(AdamDesc d)
{
    if (d.scal[2] == 0.f) return;
    const int k = blockIdx.y;
    const long long n = d.seg.count[k];
    const float *__restrict__ grad = d.seg.grad[k];
    const long long *__restrict__ ix = d.seg.index[k];
    float *__restrict__ par = d.param[k];
    float *__restrict__ m = d.exp_avg + d.seg.offset[k], *__restrict__ v = d.exp_avg_sq + d.seg.offset[k];
    const float lr = d.lr[d.group[k]], bc1 = d.scal[0], bc2s = d.scal[1];
    const float w1 = 1.0f - d.beta1, w2 = 1.0f - d.beta2, decay = 1.0f - lr * d.weight_decay, step_size = -(lr / bc1);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long e = ix ? ix[i] : i;
        const float g = (grad ? grad[e] : 0.f) * d.grad_scale;
        const float p = par[e] * decay;
        const float m0 = m[i];
        const float m1 = w1 < 0.5f ? m0 + w1 * (g - m0) : g - (g - m0) * (1.0f - w1);       // torch.lerp(exp_avg, grad, 1 - beta1)
        const float v1 = v[i] * d.beta2 + (w2 * g) * g;                                      // mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(v1) / bc2s + d.eps;
        par[e] = p + (m1 * step_size) / denom;
        m[i] = m1;
        v[i] = v1;
    }
}

// ---------------------------------------------------------------------------------------- AdamW, per-group hyperparameters
// Round 5 (dm4d_adamw_step).  What the reference's optimiser EFFECTIVELY runs is not one set of hyperparameters: training_setup's
// torch.optim.Adam(l, lr=0, eps=1e-15) fills the group dicts of optimize_list IN PLACE with Adam's defaults (betas (0.9, 0.999),
// weight_decay 0), and merge_optimizer's AdamW(l, betas=[0.9, 0.99], eps=1e-15) only setdefault()s -- the geometry / deformation
// groups keep beta2 = 0.999 and no decay, groups appended later get (0.9, 0.99) and 0.01 (geometry/sugar.py:382,406-416,
// geometry/dynamic_sugar.py:231-235).  So beta1, beta2, eps and weight_decay are per GROUP here, like lr.  Also new:
//   * skip[k]: a parameter no gradient reached is skipped by torch.optim entirely (no decay, no moment decay, its own step counter
//     not advanced): step[] and pending_decay[] are per SEGMENT;
//   * grad_in_message[k] + param_out[k]: the data-parallel slice form -- the gradient is this rank's reduce-scattered slice of the
//     message (read at [i]), the parameter is updated IN its storage through index[k] and the new value also lands in the
//     all-gather's send slice, so the sharded step needs no parameter pack and no torch operator.
struct AdamStepDesc {
    PackDesc seg;
    float *param[kMaxSeg];
    float *param_out[kMaxSeg];
    unsigned char group[kMaxSeg], grad_in_message[kMaxSeg], skip[kMaxSeg];
    float lr[8], beta1[8], beta2[8], eps[8], weight_decay[8];
    float grad_scale;
    float *exp_avg, *exp_avg_sq;
    double *step, *pending_decay;
    const float *found_inf;
    float *scal;                        // [0] 1 = apply / 0 = skipped by found_inf; [1 + 2 k] bias correction 1 of segment k, [2 + 2 k] sqrt(bias correction 2)
};
static_assert(sizeof(AdamStepDesc) <= 4096, "kernel arguments are limited to 4 KB");

__global__ __launch_bounds__(64) void k_adamw_step_scalars(AdamStepDesc d)
{
    const int k = threadIdx.x;
    const bool apply = !d.found_inf || *d.found_inf == 0.f;
    if (k == 0) d.scal[0] = apply ? 1.f : 0.f;
    if (k >= d.seg.n_seg) return;
    const int g = d.group[k];
    const bool keep = apply && !d.skip[k];
    const double step = d.step[k] + (keep ? 1.0 : 0.0);
    d.step[k] = step;
    d.scal[1 + 2 * k] = (float)(1.0 - pow((double)d.beta1[g], step));
    d.scal[2 + 2 * k] = (float)sqrt(1.0 - pow((double)d.beta2[g], step));
    if (keep && d.pending_decay) d.pending_decay[k] *= 1.0 - (double)d.lr[g] * (double)d.weight_decay[g];
}

__global__ __launch_bounds__(256) void k_adamw_step(AdamStepDesc d)
{
    const int k = blockIdx.y;
    const bool off = d.scal[0] == 0.f || d.skip[k];
    float *__restrict__ pout = d.param_out[k];
    if (off && !pout) return;
    const long long n = d.seg.count[k];
    const float *__restrict__ grad = d.seg.grad[k];
    const long long *__restrict__ ix = d.seg.index[k];
    float *__restrict__ par = d.param[k];
    if (off) {          // a step masked on the device: nothing changes, but the all-gather's send slice still has to hold the current values
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) pout[i] = par[ix ? ix[i] : i];
        return;
    }
    const bool gmsg = d.grad_in_message[k] != 0;
    float *__restrict__ m = d.exp_avg + d.seg.offset[k], *__restrict__ v = d.exp_avg_sq + d.seg.offset[k];
    const int gq = d.group[k];
    const float lr = d.lr[gq], beta2 = d.beta2[gq], eps = d.eps[gq], bc1 = d.scal[1 + 2 * k], bc2s = d.scal[2 + 2 * k];
    const float w1 = 1.0f - d.beta1[gq], w2 = 1.0f - beta2, decay = 1.0f - lr * d.weight_decay[gq], step_size = -(lr / bc1);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long e = ix ? ix[i] : i;
        const float g = (grad ? grad[gmsg ? i : e] : 0.f) * d.grad_scale;
        const float p = par[e] * decay;
        const float m0 = m[i];
        const float m1 = w1 < 0.5f ? m0 + w1 * (g - m0) : g - (g - m0) * (1.0f - w1);       // torch.lerp(exp_avg, grad, 1 - beta1)
        const float v1 = v[i] * beta2 + (w2 * g) * g;                                        // mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(v1) / bc2s + eps;
        const float pn = p + (m1 * step_size) / denom;
        par[e] = pn;
        if (pout) pout[i] = pn;
        m[i] = m1;
        v[i] = v1;
    }
}

static int fill_pack(PackDesc &d, const dm4d_grad_segments *s, bool need_grad);

static int fill_pack(PackDesc &d, const dm4d_grad_segments *s, bool need_grad)
{
    if (!s || s->n_segments < 0 || s->n_segments > kMaxSeg) { set_error("grad segments: bad count"); return DM4D_ERR_INVALID; }
    memset(&d, 0, sizeof(d));
    d.n_seg = s->n_segments;
    for (int k = 0; k < d.n_seg; ++k) {
        if (s->count[k] < 0 || s->offset[k] < 0) { set_error("grad segments: negative count/offset"); return DM4D_ERR_INVALID; }
        if (need_grad && s->count[k] > 0 && !s->grad[k]) { set_error("grad segments: unpack needs every gradient tensor"); return DM4D_ERR_INVALID; }
        d.grad[k] = s->grad[k];
        d.index[k] = (const long long *)s->index[k];
        d.count[k] = s->count[k];
        d.offset[k] = s->offset[k];
    }
    return DM4D_OK;
}

}  // namespace dm4d

using namespace dm4d;

extern "C" {

int dm4d_grad_pack(const dm4d_grad_segments *segments, float *flat, dm4d_stream_t stream)
{
    PackDesc d;
    int rc = fill_pack(d, segments, false);
    if (rc) return rc;
    if (d.n_seg == 0) return DM4D_OK;
    if (!flat) { set_error("grad pack: null message"); return DM4D_ERR_INVALID; }
    hipLaunchKernelGGL(k_grad_pack, dim3(128, d.n_seg), dim3(256), 0, (hipStream_t)stream, d, flat);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_grad_unpack(const dm4d_grad_segments *segments, const float *flat, float scale, dm4d_stream_t stream)
{
    PackDesc d;
    int rc = fill_pack(d, segments, true);
    if (rc) return rc;
    if (d.n_seg == 0) return DM4D_OK;
    if (!flat) { set_error("grad unpack: null message"); return DM4D_ERR_INVALID; }
    hipLaunchKernelGGL(k_grad_unpack, dim3(128, d.n_seg), dim3(256), 0, (hipStream_t)stream, d, flat, scale);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_adamw_message(const dm4d_grad_segments *segments, const dm4d_adamw_args *a, float grad_scale, dm4d_stream_t stream)
{
    AdamDesc d;
    memset(&d, 0, sizeof(d));
    int rc = fill_pack(d.seg, segments, false);
    if (rc) return rc;
    if (!a || !a->exp_avg || !a->exp_avg_sq || !a->step || !a->scratch) { set_error("adamw: null state"); return DM4D_ERR_INVALID; }
    if (a->n_groups < 1 || a->n_groups > 8) { set_error("adamw: %d groups (1..8)", a->n_groups); return DM4D_ERR_INVALID; }
    for (int k = 0; k < d.seg.n_seg; ++k) {
        if (d.seg.count[k] > 0 && !a->param[k]) { set_error("adamw: null parameter storage (segment %d)", k); return DM4D_ERR_INVALID; }
        if (a->group[k] < 0 || a->group[k] >= a->n_groups) { set_error("adamw: segment %d in group %d", k, a->group[k]); return DM4D_ERR_INVALID; }
        d.param[k] = a->param[k];
        d.group[k] = a->group[k];
    }
    for (int g = 0; g < a->n_groups; ++g) d.lr[g] = a->lr[g];
    d.n_groups = a->n_groups;
    d.beta1 = a->beta1; d.beta2 = a->beta2; d.eps = a->eps; d.weight_decay = a->weight_decay; d.grad_scale = grad_scale;
    d.exp_avg = a->exp_avg; d.exp_avg_sq = a->exp_avg_sq; d.step = a->step; d.pending_decay = a->pending_decay;
    d.found_inf = a->found_inf; d.scal = a->scratch;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_adamw_scalars, dim3(1), dim3(64), 0, st, d);
    if (d.seg.n_seg > 0) hipLaunchKernelGGL(k_adamw_message, dim3(128, d.seg.n_seg), dim3(256), 0, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_adamw_step(const dm4d_grad_segments *segments, const dm4d_adamw_step_args *a, float grad_scale, dm4d_stream_t stream)
{
    static AdamStepDesc d;           // (3.9 KB: not on the stack of a ctypes caller's thread for no reason; filled and launched under one lock)
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    memset(&d, 0, sizeof(d));
    int rc = fill_pack(d.seg, segments, false);
    if (rc) return rc;
    if (!a || !a->exp_avg || !a->exp_avg_sq || !a->step || !a->scratch) { set_error("adamw step: null state"); return DM4D_ERR_INVALID; }
    if (a->n_groups < 1 || a->n_groups > 8) { set_error("adamw step: %d groups (1..8)", a->n_groups); return DM4D_ERR_INVALID; }
    for (int k = 0; k < d.seg.n_seg; ++k) {
        if (d.seg.count[k] > 0 && !a->param[k]) { set_error("adamw step: null parameter storage (segment %d)", k); return DM4D_ERR_INVALID; }
        if (a->group[k] < 0 || a->group[k] >= a->n_groups) { set_error("adamw step: segment %d in group %d", k, a->group[k]); return DM4D_ERR_INVALID; }
        if (a->grad_in_message[k] && !d.seg.grad[k] && d.seg.count[k] > 0) { set_error("adamw step: segment %d: a message-layout gradient cannot be NULL", k); return DM4D_ERR_INVALID; }
        d.param[k] = a->param[k];
        d.param_out[k] = a->param_out[k];
        d.group[k] = (unsigned char)a->group[k];
        d.grad_in_message[k] = a->grad_in_message[k] ? 1 : 0;
        d.skip[k] = a->skip[k] ? 1 : 0;
    }
    for (int g = 0; g < a->n_groups; ++g) {
        if (!(a->beta1[g] >= 0.f && a->beta1[g] < 1.f && a->beta2[g] >= 0.f && a->beta2[g] < 1.f && a->eps[g] >= 0.f)) {
            set_error("adamw step: group %d: betas (%g, %g) / eps %g out of range", g, a->beta1[g], a->beta2[g], a->eps[g]);
            return DM4D_ERR_INVALID;
        }
        d.lr[g] = a->lr[g]; d.beta1[g] = a->beta1[g]; d.beta2[g] = a->beta2[g]; d.eps[g] = a->eps[g]; d.weight_decay[g] = a->weight_decay[g];
    }
    d.grad_scale = grad_scale;
    d.exp_avg = a->exp_avg; d.exp_avg_sq = a->exp_avg_sq; d.step = a->step; d.pending_decay = a->pending_decay;
    d.found_inf = a->found_inf; d.scal = a->scratch;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_adamw_step_scalars, dim3(1), dim3(64), 0, st, d);
    if (d.seg.n_seg > 0) hipLaunchKernelGGL(k_adamw_step, dim3(128, d.seg.n_seg), dim3(256), 0, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // extern "C"
