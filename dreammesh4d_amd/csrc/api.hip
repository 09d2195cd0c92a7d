// api.hip -- the C ABI of libdm4d_hip.so (declared in include/dm4d.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "raster.h"

namespace dm4d {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- per-kernel HIP-event timing (off by default) ----------------------------------------
struct ProfPair { hipEvent_t a, b; int id; };
static unsigned g_prof_mask = 0;
static std::vector<ProfPair> g_prof_pairs;

ProfScope::ProfScope(int id_, hipStream_t st_) : id(id_), st(st_), slot(nullptr)
{
    if (!(g_prof_mask & (1u << id))) return;
    ProfPair p;
    p.id = id;
    if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
    (void)hipEventRecord(p.a, st);
    g_prof_pairs.push_back(p);
    slot = (void *)(uintptr_t)g_prof_pairs.size();
}
ProfScope::~ProfScope()
{
    if (!slot) return;
    (void)hipEventRecord(g_prof_pairs[(size_t)(uintptr_t)slot - 1].b, st);
}

static int check_settings(const dm4d_raster_settings *s, const dm4d_raster_inputs *in)
{
    if (!s || !in) { set_error("null settings/inputs"); return DM4D_ERR_INVALID; }
    if (s->image_height <= 0 || s->image_width <= 0) { set_error("bad image size %dx%d", s->image_height, s->image_width); return DM4D_ERR_INVALID; }
    if ((int64_t)((s->image_height + kTile - 1) / kTile) * ((s->image_width + kTile - 1) / kTile) > kMaxTiles) {
        set_error("image %dx%d has more than %d tiles", s->image_height, s->image_width, kMaxTiles);
        return DM4D_ERR_UNSUPPORTED;
    }
    if (in->N < 0) { set_error("negative N"); return DM4D_ERR_INVALID; }
    if (in->N > (1 << kGidBits)) { set_error("N = %d: at most %d Gaussians per view", in->N, 1 << kGidBits); return DM4D_ERR_UNSUPPORTED; }
    if (!s->bg || !s->viewmatrix || !s->projmatrix) { set_error("bg/viewmatrix/projmatrix must be device pointers"); return DM4D_ERR_INVALID; }
    if (in->N > 0) {
        if (!in->means3D || !in->opacities) { set_error("means3D/opacities missing"); return DM4D_ERR_INVALID; }
        if ((in->shs != nullptr) == (in->colors_precomp != nullptr)) {
            set_error("Please provide excatly one of either SHs or precomputed colors!");
            return DM4D_ERR_INVALID;
        }
        const bool sr = in->scales != nullptr && in->rotations != nullptr;
        if ((in->scales != nullptr) != (in->rotations != nullptr) || sr == (in->cov3D_precomp != nullptr)) {
            set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
            return DM4D_ERR_INVALID;
        }
        if (in->n_channels != 0 && in->n_channels != 3 && in->n_channels != 6) {
            set_error("n_channels must be 3 or 6 (got %d)", in->n_channels);
            return DM4D_ERR_INVALID;
        }
        if (in->n_channels == 6 && in->shs) { set_error("6 channels need colors_precomp"); return DM4D_ERR_INVALID; }
        if (in->shs && (s->sh_degree != 0 || in->sh_coeffs < 1)) {
            set_error("sh_degree %d with %d coefficients is not supported (only degree 0)", s->sh_degree, in->sh_coeffs);
            return DM4D_ERR_UNSUPPORTED;
        }
    }
    return DM4D_OK;
}

// single-view call -> batch of one
static BatchDesc single_view_batch(const dm4d_raster_settings *s, const dm4d_raster_inputs *in)
{
    BatchDesc d;
    memset(&d, 0, sizeof(d));
    d.B = 1;
    d.N = in->N;
    d.C = in->n_channels > 3 ? 6 : 3;
    d.W = s->image_width;
    d.H = s->image_height;
    d.sh_coeffs = in->sh_coeffs;
    d.tanfovx = s->tanfovx;
    d.tanfovy = s->tanfovy;
    d.scale_modifier = s->scale_modifier;
    d.bg = s->bg;
    d.view = s->viewmatrix;
    d.proj = s->projmatrix;
    d.campos = s->campos;
    d.means3D = in->means3D;
    d.rotations = in->rotations;
    d.scales = in->scales;
    d.opacities = in->opacities;
    d.colors = in->colors_precomp;
    d.shs = in->shs;
    d.cov3D = in->cov3D_precomp;
    return d;
}

}  // namespace dm4d

using namespace dm4d;

extern "C" {

int dm4d_version(void) { return DM4D_ABI_VERSION; }
const char *dm4d_last_error(void) { return g_err; }

void dm4d_profile_enable(unsigned kernel_mask) { g_prof_mask = kernel_mask; }

/* Sums the recorded intervals of kernel `kernel_id` (synchronises the events), frees them.
 * Returns the number of launches; *total_ms receives their summed duration. */
int64_t dm4d_profile_collect(int kernel_id, double *total_ms)
{
    double tot = 0.0;
    int64_t n = 0;
    std::vector<ProfPair> keep;
    for (auto &p : g_prof_pairs) {
        if (p.id != kernel_id) { keep.push_back(p); continue; }
        float ms = 0.f;
        (void)hipEventSynchronize(p.b);
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { tot += ms; ++n; }
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    g_prof_pairs.swap(keep);
    if (total_ms) *total_ms = tot;
    return n;
}

int dm4d_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int dm4d_device_arch(int dev, char *buf, int buflen)
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) { set_error("hipGetDeviceProperties failed"); return DM4D_ERR_HIP; }
    snprintf(buf, (size_t)buflen, "%s", p.gcnArchName);
    return DM4D_OK;
}

size_t dm4d_raster_geom_bytes(int32_t N, int32_t H, int32_t W) { return geom_layout(N, H, W).total; }
size_t dm4d_raster_binning_bytes(int64_t capacity) { return binning_bytes(capacity); }
size_t dm4d_raster_image_bytes(int32_t H, int32_t W) { return image_bytes(H, W); }
size_t dm4d_raster_grad_bytes(int64_t n_records, int32_t n_channels) { return grad_bytes(n_records, n_channels > 3 ? 6 : 3); }

int dm4d_rasterize_prepare(const dm4d_raster_settings *s, const dm4d_raster_inputs *in, int32_t *radii, void *geom,
                           size_t geom_bytes_, dm4d_stream_t stream)
{
    int rc = check_settings(s, in);
    if (rc) return rc;
    const GeomLayout L = geom_layout(in->N, s->image_height, s->image_width);
    if (!geom || geom_bytes_ < L.total) { set_error("geom workspace too small (%zu < %zu)", geom_bytes_, L.total); return DM4D_ERR_CAPACITY; }
    if (in->N > 0 && !radii) { set_error("radii missing"); return DM4D_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    BatchDesc d = single_view_batch(s, in);
    d.radii = radii;
    d.geom = (char *)geom;
    DM4D_HIP_CHECK(hipMemsetAsync((char *)geom + L.zero_begin, 0, L.zero_bytes, st));
    rc = launch_preprocess(d, st);
    if (rc) return rc;
    return launch_colscan(d, st);
}

int64_t dm4d_rasterize_num_rendered(const void *geom, dm4d_stream_t stream)
{
    if (!geom) { set_error("null geom"); return DM4D_ERR_INVALID; }
    uint32_t D = 0;   // counters live at offset 0 of the geom workspace
    hipStream_t st = (hipStream_t)stream;
    DM4D_HIP_CHECK(hipMemcpyAsync(&D, (const char *)geom + kCntD * 4, 4, hipMemcpyDeviceToHost, st));
    DM4D_HIP_CHECK(hipStreamSynchronize(st));
    return (int64_t)D;
}

int dm4d_rasterize_counts(const void *geom, int64_t *num_rendered, int64_t *num_records, dm4d_stream_t stream)
{
    if (!geom) { set_error("null geom"); return DM4D_ERR_INVALID; }
    uint32_t cnt[4] = {0, 0, 0, 0};   // counters live at offset 0 of the geom workspace
    hipStream_t st = (hipStream_t)stream;
    DM4D_HIP_CHECK(hipMemcpyAsync(cnt, geom, sizeof(cnt), hipMemcpyDeviceToHost, st));
    DM4D_HIP_CHECK(hipStreamSynchronize(st));
    if (num_rendered) *num_rendered = (int64_t)cnt[kCntD];
    if (num_records) *num_records = (int64_t)cnt[kCntR];
    return DM4D_OK;
}

int64_t dm4d_rasterize_num_records(const void *geom, dm4d_stream_t stream)
{
    int64_t R = 0;
    const int rc = dm4d_rasterize_counts(geom, nullptr, &R, stream);
    return rc ? (int64_t)rc : R;
}

int dm4d_rasterize_overflowed(const void *geom, dm4d_stream_t stream)
{
    if (!geom) { set_error("null geom"); return DM4D_ERR_INVALID; }
    uint32_t f = 0;
    hipStream_t st = (hipStream_t)stream;
    DM4D_HIP_CHECK(hipMemcpyAsync(&f, (const char *)geom + kCntOverflow * 4, 4, hipMemcpyDeviceToHost, st));
    DM4D_HIP_CHECK(hipStreamSynchronize(st));
    return f ? 1 : 0;
}

int dm4d_rasterize_render(const dm4d_raster_settings *s, const dm4d_raster_inputs *in, const int32_t *radii,
                          void *geom, void *binning, int64_t capacity, void *image, float *out_color,
                          float *out_depth, float *out_alpha, dm4d_stream_t stream)
{
    int rc = check_settings(s, in);
    if (rc) return rc;
    if (!geom || !binning || !image || !out_color || !out_depth || !out_alpha || (in->N > 0 && !radii)) { set_error("null workspace/output"); return DM4D_ERR_INVALID; }
    if (capacity < 0 || capacity > 0xFFFFFFF0ll) { set_error("capacity out of range"); return DM4D_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    BatchDesc d = single_view_batch(s, in);
    d.radii = const_cast<int32_t *>(radii);
    d.geom = (char *)geom;
    d.binning = (char *)binning;
    d.cap = (uint32_t)capacity;
    d.rec_cap = 0xFFFFFFFFu;   // the caller sizes the backward scratch from dm4d_rasterize_num_records
    d.image = (char *)image;
    d.out_color = out_color;
    d.out_depth = out_depth;
    d.out_alpha = out_alpha;
    rc = launch_scatter(d, st);
    if (rc) return rc;
    return launch_sort_and_forward(d, st);
}

int dm4d_rasterize_backward(const dm4d_raster_settings *s, const dm4d_raster_inputs *in, const int32_t *radii,
                            const void *geom, const void *binning, int64_t capacity, const void *image, void *grad,
                            int64_t record_capacity, const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha,
                            float *dL_dmeans2D, float *dL_dmeans3D, float *dL_dopacity, float *dL_dcolors,
                            float *dL_dsh, float *dL_dscales, float *dL_drotations, float *dL_dcov3D,
                            dm4d_stream_t stream)
{
    int rc = check_settings(s, in);
    if (rc) return rc;
    if (!geom || !binning || !image || !grad || !dL_dcolor) { set_error("null workspace/grad input"); return DM4D_ERR_INVALID; }
    if (in->N > 0 && (!dL_dmeans2D || !dL_dmeans3D || !radii)) { set_error("dL_dmeans2D/dL_dmeans3D/radii required"); return DM4D_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    BatchDesc d = single_view_batch(s, in);
    d.radii = const_cast<int32_t *>(radii);
    d.geom = (char *)const_cast<void *>(geom);
    d.binning = (char *)const_cast<void *>(binning);
    d.cap = (uint32_t)capacity;
    d.image = (char *)const_cast<void *>(image);
    d.dL_dcolor = dL_dcolor;
    d.dL_ddepth = dL_ddepth;
    d.dL_dalpha = dL_dalpha;
    d.dLq = (float *)grad;
    if (record_capacity < 0 || record_capacity > 0xFFFFFFF0ll) { set_error("record_capacity out of range"); return DM4D_ERR_INVALID; }
    d.rec_cap = (uint32_t)record_capacity;
    d.o = BwdOutputs{dL_dmeans2D, dL_dmeans3D, dL_dopacity, dL_dcolors, dL_dsh, dL_dscales, dL_drotations, dL_dcov3D};
    rc = launch_render_bwd(d, st);
    if (rc) return rc;
    return launch_gather_bwd(d, st);
}

int64_t dm4d_rasterize_forward(const dm4d_raster_settings *s, const dm4d_raster_inputs *in, float *out_color,
                               float *out_depth, float *out_alpha, int32_t *radii, dm4d_alloc_fn alloc, void *ctx,
                               dm4d_stream_t stream)
{
    int rc = check_settings(s, in);
    if (rc) return rc;
    if (!alloc) { set_error("alloc callback missing"); return DM4D_ERR_INVALID; }
    const size_t gb = dm4d_raster_geom_bytes(in->N, s->image_height, s->image_width);
    void *geom = alloc(ctx, 0, gb);
    if (!geom) { set_error("alloc(geom) failed"); return DM4D_ERR_CAPACITY; }
    rc = dm4d_rasterize_prepare(s, in, radii, geom, gb, stream);
    if (rc) return rc;
    const int64_t D = dm4d_rasterize_num_rendered(geom, stream);
    if (D < 0) return D;
    void *binning = alloc(ctx, 1, dm4d_raster_binning_bytes(D));
    void *image = alloc(ctx, 2, dm4d_raster_image_bytes(s->image_height, s->image_width));
    if (!binning || !image) { set_error("alloc(binning/image) failed"); return DM4D_ERR_CAPACITY; }
    rc = dm4d_rasterize_render(s, in, radii, geom, binning, D, image, out_color, out_depth, out_alpha, stream);
    if (rc) return rc;
    return D;
}

int dm4d_raster_read_sorted(const void *geom, const void *binning, int32_t N, int32_t H, int32_t W, int64_t D,
                            uint64_t *keys, uint32_t *values, uint32_t *ranges, dm4d_stream_t stream)
{
    // Rebuild upstream's (key, value) view of the sorted list: key = tile<<32 | depth bits.
    hipStream_t st = (hipStream_t)stream;
    const GeomLayout L = geom_layout(N, H, W);
    const int T = L.T;
    std::vector<uint32_t> start((size_t)T + 1), dbits((size_t)(N > 0 ? N : 1));
    DM4D_HIP_CHECK(hipMemcpyAsync(start.data(), (const char *)geom + L.tile_start, ((size_t)T + 1) * 4, hipMemcpyDeviceToHost, st));
    if (N > 0) DM4D_HIP_CHECK(hipMemcpyAsync(dbits.data(), (const char *)geom + L.depth, (size_t)N * 4, hipMemcpyDeviceToHost, st));
    const BinPtrs b = bin_ptrs(const_cast<void *>(binning), D);
    if (D > 0) DM4D_HIP_CHECK(hipMemcpyAsync(values, b.point_list, (size_t)D * 4, hipMemcpyDeviceToHost, st));
    DM4D_HIP_CHECK(hipStreamSynchronize(st));
    for (int t = 0; t < T; ++t) {
        ranges[2 * t] = start[t];
        ranges[2 * t + 1] = start[t + 1];
        for (uint32_t e = start[t]; e < start[t + 1] && (int64_t)e < D; ++e)
            keys[e] = ((uint64_t)t << 32) | (values[e] < (uint32_t)N ? dbits[values[e]] : 0u);
    }
    return DM4D_OK;
}

int dm4d_raster_read_geom(const void *geom, int32_t N, int32_t H, int32_t W, float *xy, float *depths,
                          float *conic_opacity, uint32_t *tiles_touched, dm4d_stream_t stream)
{
    hipStream_t st = (hipStream_t)stream;
    const GeomLayout L = geom_layout(N, H, W);
    const char *b = (const char *)geom;
    if (N > 0) {
        DM4D_HIP_CHECK(hipMemcpyAsync(xy, b + L.xy, (size_t)N * 8, hipMemcpyDeviceToHost, st));
        DM4D_HIP_CHECK(hipMemcpyAsync(depths, b + L.depth, (size_t)N * 4, hipMemcpyDeviceToHost, st));
        DM4D_HIP_CHECK(hipMemcpyAsync(conic_opacity, b + L.conic_opacity, (size_t)N * 16, hipMemcpyDeviceToHost, st));
        DM4D_HIP_CHECK(hipMemcpyAsync(tiles_touched, b + L.tiles_touched, (size_t)N * 4, hipMemcpyDeviceToHost, st));
    }
    DM4D_HIP_CHECK(hipStreamSynchronize(st));
    return DM4D_OK;
}

int dm4d_raster_read_image_state(const void *geom, const void *binning, const void *image, int32_t N, int32_t H, int32_t W,
                                 int64_t D, uint32_t *n_contrib, float *final_T, dm4d_stream_t stream)
{
    hipStream_t st = (hipStream_t)stream;
    const ImgPtrs im = img_ptrs(const_cast<void *>(image), H, W);
    const size_t P = (size_t)H * W;
    // the kernels count contributors in cell-list positions; upstream's n_contrib counts tile-list positions
    uint32_t *tmp = nullptr;
    DM4D_HIP_CHECK(hipMalloc(&tmp, (P > 0 ? P : 1) * 4));
    int rc = launch_n_contrib_tile_positions(const_cast<void *>(geom), const_cast<void *>(binning), const_cast<void *>(image), N, H, W, D, tmp, st);
    if (rc == DM4D_OK) {
        hipError_t e = hipMemcpyAsync(final_T, im.final_T, P * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(n_contrib, tmp, P * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("read_image_state: %s", hipGetErrorString(e)); rc = DM4D_ERR_HIP; }
    }
    (void)hipFree(tmp);
    return rc;
}

int dm4d_debug_trace(void *trace, uint32_t min_work) { return set_trace_buffer(trace, min_work); }
int dm4d_debug_sort_trace(void *trace) { return set_sort_trace_buffer(trace); }

int dm4d_mark_visible(int32_t N, const float *means3D, const float *viewmatrix, uint8_t *present, dm4d_stream_t stream)
{
    if (N < 0 || (N > 0 && (!means3D || !viewmatrix || !present))) { set_error("bad arguments"); return DM4D_ERR_INVALID; }
    return launch_mark_visible(N, means3D, viewmatrix, present, (hipStream_t)stream);
}

}  // extern "C"
