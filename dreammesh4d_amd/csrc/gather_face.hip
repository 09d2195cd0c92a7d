// gather_face.hip -- B2 and the face part of the face -> Gaussian backward as ONE kernel (batched path).
//
// B2 (k_gather_bwd, raster_preprocess.hip) sums a Gaussian's backward records and runs the preprocess backward per (view,
// Gaussian), writing dL/dmean3D, dL/drotation and dL/dcolour per VIEW (52 B x N x views: 83 MB per 8-view step on the bench
// scene); k_face_bwd_face (skinning.hip) reads them back, sums the views of a frame and reduces the six Gaussians of a face to
// its corner records: 32 us of the step's serial chain.  Here B2's body (same code: gather_gaussian) keeps the three gradients in
// registers and finishes the face (same code: face_bwd_finish) -- see the kernel for how the views of a frame meet.
//
// One translation unit with the two files whose device code it calls.
#include "raster_preprocess.hip"
#include "skinning.hip"

namespace dm4d {

// Round 4: a thread owns (VIEW, Gaussian) -- B2's own parallelism (round 3's version gave a thread a FRAME's views in a loop: half the
// loads in flight of B2, which is bandwidth-bound: 214 us against 168 + 32 for the two kernels, and stayed off).  The face's corner
// records are therefore per VIEW ([views][F][3][6]: 19 MB per 8-view step on the bench scene where the per-view Gaussian gradients
// were 64 MB written and read back); the backward is linear, so k_face_bwd_vertex adds a frame's views when it sums a vertex's
// corners.  Equal to the two-kernel path up to the order of those additions (tests/test_views_gpu.py).
template <int PARTS>
__global__ __launch_bounds__(kSkinThreads) void k_gather_face_bwd(BatchDesc d, int F, int G, int V, const int32_t *__restrict__ faces,
                                                                  const float *__restrict__ vxyz, const float *__restrict__ vrot,
                                                                  const float *__restrict__ q_static, float *__restrict__ rec, int pypose)
{
    __shared__ float sV[16], sP[16];
    // ONE block of LDS for both phases: the waves' record chunks while the records are summed, then (after a barrier) the face part's
    // exchange arrays -- 24.6 KB instead of 37 KB per workgroup: five workgroups per CU (what the 92 VGPRs allow) instead of four
    constexpr int kChunk = kGatherChunkFloats(PARTS);
    static_assert((kSkinThreads / 64) * kChunk >= kSkinThreads * 12, "the face part's arrays must fit the record chunks");
    __shared__ __attribute__((aligned(16))) float s_mem[(kSkinThreads / 64) * kChunk];
    const int tid = threadIdx.x;
    const int faces_per_wg = kSkinThreads / G;
    const int fl = tid / G, sl = tid - fl * G;      // face in workgroup, slot
    const int f = blockIdx.x * faces_per_wg + fl;
    const bool live = fl < faces_per_wg && f < F;
    const int bv = blockIdx.y;
    const int frame = d.frame_index ? d.frame_index[bv] : bv;
    const int i = blockIdx.x * faces_per_wg * G + tid;      // == f * G + sl: the wave's Gaussians are consecutive
    const ViewCtx c = resolve(d, bv);
    if (tid < 16) { sV[tid] = c.vp.view[tid]; sP[tid] = c.vp.proj[tid]; }
    __syncthreads();
    GatherOut res;
    gather_gaussian<PARTS>(d, c, i, live, sV, sP, s_mem + (tid >> 6) * kChunk, res);
    __syncthreads();              // every wave is done with its record chunk: the memory changes hands
    const v3 gm = mk3(res.dmean[0], res.dmean[1], res.dmean[2]), gn = mk3(res.dcol[3], res.dcol[4], res.dcol[5]);
    const q4 go = q4{res.drot[1], res.drot[2], res.drot[3], res.drot[0]};      // (w, x, y, z) -> (x, y, z, w)
    face_bwd_finish(F, G, faces, vxyz + (size_t)frame * V * 3, vrot + (size_t)frame * V * 4, q_static, true, true, true, live, f, sl, fl * G,
                    gm, go, gn, rec + (size_t)bv * F * 3 * kCornerRec, pypose, reinterpret_cast<float (*)[3]>(s_mem),
                    reinterpret_cast<float (*)[9]>(s_mem + kSkinThreads * 3));
}

// d: the backward's batch (per-view outputs dL_dmeans3D / dL_drotations / dL_dcolors NULL, dL_dmeans2D optional);
// n_frames: frames of the batch (d.frame_index maps views to them; NULL: view == frame)
int launch_gather_face_bwd(const BatchDesc &d, int n_frames, int F, int G, int V, const int32_t *faces, const float *vxyz, const float *vrot,
                           const float *qs, float *face_scratch, hipStream_t st)
{
    const int pypose = (G & kPypose) ? 1 : 0;
    G &= 0xff;
    if (F <= 0 || n_frames <= 0) return DM4D_OK;
    if (d.C != 6) { set_error("the fused gather + face backward needs the 6-channel batch"); return DM4D_ERR_INVALID; }
    ProfScope prof_(kKGatherBwd, st);
    const int fpw = kSkinThreads / G;
    (void)n_frames;
    const dim3 grid((F + fpw - 1) / fpw, d.B);       // a workgroup row per VIEW
    if (grad_stride(d.C, d.lean) == 8)
        hipLaunchKernelGGL(k_gather_face_bwd<2>, grid, dim3(kSkinThreads), 0, st, d, F, G, V, faces, vxyz, vrot, qs, face_scratch, pypose);
    else if (grad_stride(d.C, d.lean) == 12)
        hipLaunchKernelGGL(k_gather_face_bwd<3>, grid, dim3(kSkinThreads), 0, st, d, F, G, V, faces, vxyz, vrot, qs, face_scratch, pypose);
    else
        hipLaunchKernelGGL(k_gather_face_bwd<4>, grid, dim3(kSkinThreads), 0, st, d, F, G, V, faces, vxyz, vrot, qs, face_scratch, pypose);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // namespace dm4d
