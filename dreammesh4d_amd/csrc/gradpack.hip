// gradpack.hip -- pack / unpack of the data-parallel gradient message (gfx950).
//
// The one exchange step of the path is an all-reduce of the parameter gradients.  The
// message is a flat float32 buffer of up to 64 segments: a segment is a whole gradient tensor or, for the
// HexPlane spatial planes, only the elements the (static, rank-identical) graph nodes touch.  One launch packs
// all segments, one launch unpacks them (and applies the 1/world scale) -- instead of ~80 torch copy / index
// kernels per step.
#include <string.h>

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

}  // extern "C"
