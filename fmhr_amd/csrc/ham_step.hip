// fmhr_amd/csrc/ham_step.hip — the fused stage-2 HAM iteration (forward + backward) as one C-ABI call pair.
// Replaces the loop body /root/reference/mesh_sfs_optim.py:253-308 for the views a rank owns; every launch goes to the
// caller's stream, nothing synchronises, so the pair is hipGraph-capturable.
#include "common.h"

namespace fmhr {
// raster.hip / antialias.hip / mesh.hip / shade.hip
int rasterize_fwd_impl(const float*, const int32_t*, int, int, int, int, int, float*, void*, hipStream_t);
int aa_fwd_dual(const float*, const float*, const float*, const float*, const int32_t*, const int32_t*, int, int, int, int, int,
                float*, float*, int32_t*, float*, int32_t*, int, hipStream_t);
int aa_bwd_dual(const float*, const float*, const float*, const float*, const int32_t*, const float*, const float*,
                const int32_t*, const float*, const int32_t*, int, int, int, int, int, float*, float*, hipStream_t);
int normals_fwd_impl(const float*, const int32_t*, int, int, int, float*, float*, hipStream_t);
int normals_bwd_impl(const float*, const int32_t*, const float*, const float*, int, int, int, int, float*, float*, hipStream_t);
int lap_fwd_impl(const float*, const int32_t*, const int32_t*, int, int, float, float*, float*, hipStream_t);
int lap_bwd_impl(const float*, const int32_t*, const int32_t*, int, int, float, float*, float*, hipStream_t);
int add_delta_impl(const float*, const float*, int, float*, hipStream_t);
int transform_fwd_impl(const float*, const float*, const float*, int, int, float*, hipStream_t);
int transform_bwd_impl(const float*, const float*, const float*, int, int, float*, hipStream_t);
int edge_hinge_impl(const float*, const int32_t*, int, int, float, float, float, float*, float*, hipStream_t);
int delta_term_impl(const float*, int, float, float, float*, float*, hipStream_t);
int shade_fwd_impl(const float*, const int32_t*, const float*, const float*, const float*, const float*, int, int, int, int,
                   float*, float*, int*, hipStream_t);
int loss_pix_impl(const float*, const float*, const float*, const float*, const float*, const float*, long long, const float*,
                  float, float, float, float*, float*, float*, hipStream_t);
int shade_bwd_impl(const float*, const int32_t*, const float*, const float*, const float*, const float*, const float*,
                   const float*, int, int, int, int, int, float*, float*, hipStream_t);
int unpack_albedo_grad_impl(const float*, int, float*, hipStream_t);
int finish_losses_impl(float*, const float*, hipStream_t);

// workspace carving: every block 256-B aligned
struct Layout {
    size_t vertices, nsum, normals, pos, raster_ws, rast, tmp_img, mask_col, aa_img, aa_mask, aa_items, aa_alphas, aa_counter,
        g_aa_img, g_aa_mask, g_tmp_img, gpos, gattr, gvert, gtmp, lx_v, lx_a, rowg, total;
};

static size_t take(size_t& off, size_t bytes)
{
    size_t at = off;
    off += (bytes + 255) & ~(size_t)255;
    return at;
}

static Layout make_layout(const fmhr_ham_dims* d)
{
    Layout L;
    size_t off = 0;
    size_t V = d->V, nP = (size_t)d->n * d->H * d->W, nV = (size_t)d->n * d->V;
    L.vertices = take(off, V * 12);
    L.nsum = take(off, V * 12);
    L.normals = take(off, V * 12);
    L.pos = take(off, nV * 16);
    L.raster_ws = take(off, fmhr_rasterize_workspace_bytes(d->n, d->F, d->H, d->W));
    L.rast = take(off, nP * 16);
    L.tmp_img = take(off, nP * 12);
    L.mask_col = take(off, nP * 4);
    L.aa_img = take(off, nP * 12);
    L.aa_mask = take(off, nP * 4);
    L.aa_items = take(off, (size_t)d->aa_cap * 16);
    L.aa_alphas = take(off, (size_t)d->aa_cap * 4);
    L.aa_counter = take(off, 16);
    L.g_aa_img = take(off, nP * 12);
    L.g_aa_mask = take(off, nP * 4);
    L.g_tmp_img = take(off, nP * 12);
    L.gpos = take(off, nV * 16);
    L.gattr = take(off, V * 32);
    L.gvert = take(off, V * 12);
    L.gtmp = take(off, V * 12);
    L.lx_v = take(off, V * 12);
    L.lx_a = take(off, V * 12);
    L.rowg = take(off, V * 12);
    L.total = off;
    return L;
}

static int check_dims(const fmhr_ham_dims* d)
{
    FMHR_CHECK_ARG(d != nullptr, "ham: dims is null");
    FMHR_CHECK_ARG(d->n >= 1 && d->V >= 1 && d->F >= 1 && d->H >= 1 && d->W >= 1 && d->aa_cap >= 1,
                   "ham: bad dims n=%d V=%d F=%d H=%d W=%d aa_cap=%d", d->n, d->V, d->F, d->H, d->W, d->aa_cap);
    FMHR_CHECK_ARG(d->H <= 8192 && d->W <= 8192 && d->F < (1 << 24), "ham: H,W <= 8192 and F < 2^24 required");
    return FMHR_OK;
}

#define TRY(x)                 \
    do {                       \
        int rc__ = (x);        \
        if (rc__) return rc__; \
    } while (0)
#define HIPTRY(x, what)                                   \
    do {                                                  \
        hipError_t e__ = (x);                             \
        if (e__ != hipSuccess) return hip_fail(e__, what); \
    } while (0)

}  // namespace fmhr

using namespace fmhr;

extern "C" size_t fmhr_ham_workspace_bytes(const fmhr_ham_dims* d)
{
    if (check_dims(d)) return 0;
    return make_layout(d).total;
}

extern "C" int fmhr_ham_workspace_views(const fmhr_ham_dims* d, void* ws, size_t ws_bytes, fmhr_ham_views* out)
{
    TRY(check_dims(d));
    Layout L = make_layout(d);
    FMHR_CHECK_ARG(ws && out && ws_bytes >= L.total, "ham_workspace_views: workspace too small or null");
    char* w = (char*)ws;
    out->pos = (float*)(w + L.pos);
    out->normals = (float*)(w + L.normals);
    out->rast = (float*)(w + L.rast);
    out->tmp_img = (float*)(w + L.tmp_img);
    out->aa_img = (float*)(w + L.aa_img);
    out->aa_mask = (float*)(w + L.aa_mask);
    out->aa_counter = (int32_t*)(w + L.aa_counter);
    return FMHR_OK;
}

extern "C" int fmhr_ham_forward(const fmhr_ham_dims* d, const float* vertices_tmp, const float* delta, const float* albedo,
                                const float* sh, const float* w2c, const float* proj, const float* img, const float* mask,
                                const int32_t* tri, const int32_t* opp, int32_t* n_valid_local, void* ws, size_t ws_bytes,
                                void* stream)
{
    (void)img;
    TRY(check_dims(d));
    Layout L = make_layout(d);
    FMHR_CHECK_ARG(ws && ws_bytes >= L.total, "ham_forward: workspace too small (%zu < %zu)", ws_bytes, L.total);
    FMHR_CHECK_ARG(vertices_tmp && delta && albedo && sh && w2c && proj && mask && tri && opp && n_valid_local,
                   "ham_forward: null pointer");
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    float* vertices = (float*)(w + L.vertices);
    float* pos = (float*)(w + L.pos);
    float* rast = (float*)(w + L.rast);
    size_t nP = (size_t)d->n * d->H * d->W;
    {
        ProfScope ps(FMHR_SLOT_GEOMETRY, st);
        TRY(add_delta_impl(vertices_tmp, delta, d->V * 3, vertices, st));
        TRY(normals_fwd_impl(vertices, tri, 1, d->V, d->F, (float*)(w + L.nsum), (float*)(w + L.normals), st));
        TRY(transform_fwd_impl(vertices, w2c, proj, d->n, d->V, pos, st));
    }
    TRY(rasterize_fwd_impl(pos, tri, d->n, d->V, d->F, d->H, d->W, rast, w + L.raster_ws, st));
    {
        ProfScope ps(FMHR_SLOT_SHADE_FWD, st);
        HIPTRY(hipMemsetAsync(n_valid_local, 0, 4, st), "memset n_valid");
        TRY(shade_fwd_impl(rast, tri, (float*)(w + L.normals), albedo, sh, mask, d->n, d->F, d->H, d->W, (float*)(w + L.tmp_img),
                           (float*)(w + L.mask_col), n_valid_local, st));
    }
    ProfScope ps_aa(FMHR_SLOT_AA_FWD, st);
    HIPTRY(hipMemcpyAsync(w + L.aa_img, w + L.tmp_img, nP * 12, hipMemcpyDeviceToDevice, st), "copy tmp_img");
    HIPTRY(hipMemcpyAsync(w + L.aa_mask, w + L.mask_col, nP * 4, hipMemcpyDeviceToDevice, st), "copy mask_col");
    HIPTRY(hipMemsetAsync(w + L.aa_counter, 0, 16, st), "memset aa_counter");
    TRY(aa_fwd_dual((float*)(w + L.mask_col), (float*)(w + L.tmp_img), rast, pos, tri, opp, d->n, d->V, d->F, d->H, d->W,
                    (float*)(w + L.aa_mask), (float*)(w + L.aa_img), (int32_t*)(w + L.aa_items), (float*)(w + L.aa_alphas),
                    (int32_t*)(w + L.aa_counter), d->aa_cap, st));
    return FMHR_OK;
}

extern "C" int fmhr_ham_backward(const fmhr_ham_dims* d, const fmhr_ham_weights* wt, const float* vertices_tmp,
                                 const float* delta, const float* albedo, const float* sh, const float* w2c, const float* proj,
                                 const float* img, const float* mask, const float* valid_mask, const int32_t* tri,
                                 const int32_t* opp, const int32_t* row_ptr, const int32_t* col_idx,
                                 const float* n_valid_global, float* losses, float* gdelta, float* galbedo, void* ws,
                                 size_t ws_bytes, void* stream)
{
    (void)vertices_tmp; (void)opp;
    TRY(check_dims(d));
    Layout L = make_layout(d);
    FMHR_CHECK_ARG(ws && ws_bytes >= L.total, "ham_backward: workspace too small (%zu < %zu)", ws_bytes, L.total);
    FMHR_CHECK_ARG(wt && delta && albedo && sh && w2c && proj && img && mask && valid_mask && tri && row_ptr && col_idx &&
                       n_valid_global && losses && gdelta && galbedo, "ham_backward: null pointer");
    FMHR_CHECK_ARG(wt->mse_denominator > 0.f, "ham_backward: mse_denominator must be > 0");
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    float* vertices = (float*)(w + L.vertices);
    float* pos = (float*)(w + L.pos);
    float* rast = (float*)(w + L.rast);
    float* gpos = (float*)(w + L.gpos);
    float* gattr = (float*)(w + L.gattr);
    float* gvert = (float*)(w + L.gvert);
    size_t nP = (size_t)d->n * d->H * d->W, V = d->V;
    const float rg = wt->reg_grad_scale;

    {
        ProfScope ps(FMHR_SLOT_LOSS_PIX, st);
        HIPTRY(hipMemsetAsync(losses, 0, 32, st), "memset losses");
        HIPTRY(hipMemsetAsync(gpos, 0, (size_t)d->n * V * 16, st), "memset gpos");
        HIPTRY(hipMemsetAsync(gattr, 0, V * 32, st), "memset gattr");
        HIPTRY(hipMemsetAsync(gvert, 0, V * 12, st), "memset gvert");
        HIPTRY(hipMemsetAsync(galbedo, 0, V * 12, st), "memset galbedo");
        // losses on the antialiased images + their per-pixel gradients
        TRY(loss_pix_impl(rast, (float*)(w + L.aa_img), img, mask, (float*)(w + L.aa_mask), valid_mask, (long long)nP,
                          n_valid_global, wt->sfs, wt->mask, wt->mse_denominator, (float*)(w + L.g_aa_img),
                          (float*)(w + L.g_aa_mask), losses, st));
    }
    {
        // antialias backward (both images share one queue): colour grads of the shaded image, position grads of both
        ProfScope ps(FMHR_SLOT_AA_BWD, st);
        HIPTRY(hipMemcpyAsync(w + L.g_tmp_img, w + L.g_aa_img, nP * 12, hipMemcpyDeviceToDevice, st), "copy g_aa_img");
        TRY(aa_bwd_dual((float*)(w + L.mask_col), (float*)(w + L.tmp_img), rast, pos, tri, (float*)(w + L.g_aa_mask),
                        (float*)(w + L.g_aa_img), (int32_t*)(w + L.aa_items), (float*)(w + L.aa_alphas),
                        (int32_t*)(w + L.aa_counter), d->aa_cap, d->n, d->V, d->H, d->W, (float*)(w + L.g_tmp_img), gpos, st));
    }
    {
        // shade + interpolate + rasterize backward
        ProfScope ps(FMHR_SLOT_SHADE_BWD, st);
        TRY(shade_bwd_impl(rast, tri, pos, (float*)(w + L.normals), albedo, sh, mask, (float*)(w + L.g_tmp_img), d->n, d->V,
                           d->F, d->H, d->W, gattr, gpos, st));
    }
    {
        // clip transform backward (sum over views)
        ProfScope ps(FMHR_SLOT_TRANSFORM_BWD, st);
        TRY(transform_bwd_impl(gpos, w2c, proj, d->n, d->V, gvert, st));
    }
    {
        ProfScope ps(FMHR_SLOT_NORMALS_BWD, st);
        TRY(normals_bwd_impl(vertices, tri, (float*)(w + L.nsum), gattr, 8, 1, d->V, d->F, (float*)(w + L.gtmp), gvert, st));
        TRY(unpack_albedo_grad_impl(gattr, d->V, galbedo, st));
    }
    ProfScope ps(FMHR_SLOT_REGULARISERS, st);
    // view-independent regularisers (mesh_sfs_optim.py:292-293, 296-304)
    TRY(lap_fwd_impl(vertices, row_ptr, col_idx, d->V, 3, wt->lap, (float*)(w + L.lx_v), losses + FMHR_LOSS_LAP, st));
    TRY(lap_fwd_impl(albedo, row_ptr, col_idx, d->V, 3, wt->albedo, (float*)(w + L.lx_a), losses + FMHR_LOSS_ALBEDO, st));
    if (rg != 0.f) {
        TRY(lap_bwd_impl((float*)(w + L.lx_v), row_ptr, col_idx, d->V, 3, wt->lap * rg, (float*)(w + L.rowg), gvert, st));
        TRY(lap_bwd_impl((float*)(w + L.lx_a), row_ptr, col_idx, d->V, 3, wt->albedo * rg, (float*)(w + L.rowg), galbedo, st));
    }
    TRY(edge_hinge_impl(vertices, tri, d->V, d->F, wt->edge_length_mean, wt->edge, wt->edge * rg, losses + FMHR_LOSS_EDGE, gvert, st));
    TRY(delta_term_impl(delta, d->V, wt->delta, wt->delta * rg, losses + FMHR_LOSS_DELTA, gvert, st));
    HIPTRY(hipMemcpyAsync(gdelta, gvert, V * 12, hipMemcpyDeviceToDevice, st), "copy gdelta");
    TRY(finish_losses_impl(losses, n_valid_global, st));
    return FMHR_OK;
}
