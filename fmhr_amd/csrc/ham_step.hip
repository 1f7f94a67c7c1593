// fmhr_amd/csrc/ham_step.hip — the fused stage-2 HAM iteration (forward + backward) as one C-ABI call pair.
// Replaces the loop body /root/reference/mesh_sfs_optim.py:253-308 for the views a rank owns; every launch goes to the
// caller's stream, nothing synchronises, so the pair is hipGraph-capturable.
#include "common.h"
#include "shade_common.h"
#include "adam.h"
#include <string.h>

namespace fmhr {
struct MeshTailArgs {
    const float4* part; float* gvert_aa; int G;
    float* gvert_aa2;
    float* vacc;
    const float* verts; const float* delta; const float* albedo; const float* nsum;
    const int32_t* tri; const int32_t* vf_ptr; const int32_t* vf_idx; const int32_t* row_ptr; const int32_t* col;
    const int32_t* col_mult;
    int V, F;
    int v0, v1;
    float lap_w, alb_w, edge_w, delta_w, edge_mean, rg;
    float* gs; float* rowg_v; float* rowg_a; float* gacc_v; float* gacc_a; float* gacc_r;
    float* loss_part;
    float* gdelta; float* galbedo; float* gd_rest; float* ga_rest;
    AdamTensor ad_d, ad_a;
};
struct FinishArgs {
    float* losses; const float* n_valid; const float* part_l1; const float* part_l2; int nb;
    const float* tail_part; int tail_nb; float w_sfs, w_mask, mse_den;
    float reg_scale;
    const float* nvalid_report; float nvalid_scale;
    int split;
};
int mesh_tail_impl(const MeshTailArgs&, const FinishArgs&, hipStream_t);
int mesh_tail_blocks(int V);
int ham_combine_impl(float*, int, const fmhr_adam_param*, const fmhr_adam_param*, hipStream_t);
// raster.hip / antialias.hip / mesh.hip / shade.hip
size_t raster_core_ws_bytes(int n, int F, int H, int W);
size_t raster_keys_bytes(int n, int H, int W);
int raster_sparse_init(void*, unsigned char*, int, int, int, hipStream_t);
int rasterize_fwd_impl(const float*, const int2*, const int32_t*, const int32_t*, unsigned char*, int, int, int, int, int, int, float*,
                       void*, unsigned char*, unsigned long long*, const RasterFused*, hipStream_t);
int valid_mask_segments_impl(const float*, int, int, int, int, unsigned char*, hipStream_t);
int active_segments_impl(const unsigned long long*, const unsigned char*, long long, int, unsigned char*, int*, int*, const int*,
                         float*, float*, const ViewSel, hipStream_t);
int reshade_dense_impl(const unsigned long long*, const float*, const int32_t*, const float*, const float*, int, int, int, float*, hipStream_t);
int active_segments_blocks(long long N);
int densify_impl(const unsigned long long*, const unsigned char*, long long, const float*, float*, float*, hipStream_t);
int aa_fwd_rgba(const float*, const float*, float*, const float*, const float*, const float*, const float*, const int32_t*, const int32_t*, const unsigned char*, const int*,
                const int*, const unsigned long long*, const unsigned long long*, int, int, int, int, int, int, float*, int32_t*, float*, int32_t*, int32_t*, int,
                const float*, int, int*, float*, hipStream_t);
int aa_bwd_rgba(const float*, const float*, const float*, const float*, const int32_t*, const float*, const float*, const float*, const float*,
                const unsigned long long*, const float*, float, float, float, const int32_t*, const float*, const int32_t*, int, int, int,
                int, int, int, float*, const float*, float*, float*, const int32_t*, hipStream_t);
int clip_positions_impl(const float*, const float*, int, int, float*, hipStream_t);
int geometry_impl(const float*, const float*, const float*, const float*, int, int, float*, int, int, int2*, float*, const int32_t*,
                  const int32_t*, const int32_t*, float*, float*, const float*, float*, float*, int*, const int32_t*, const float*, float*, int, int32_t*, hipStream_t);
int loss_pix_impl(const unsigned long long*, const float*, const float*, const float*, const float*, long long, int, int, const ViewSel, const int*, const int*,
                  const float*, float, float, float, float*, float*, float*, float*, hipStream_t);
int face_bwd_impl(const float*, const int32_t*, const float*, const float*, const float*, const float*, const void*, const unsigned char*,
                  const int*, const int32_t*, const float*, const float*, int, int, int, int, int, int, float*,
                  const int2*, const int2*, const int*, const unsigned long long*, float*, hipStream_t);
int face_bwd_groups(int n, int H, int W);
int face_bwd_blocks(int F);

// workspace carving: every block 256-B aligned
struct Layout {
    size_t vertices, nsum, normals, attr8, sh_local, view_idx_ws, pos, snap, silh, raster_ws, dirty, covw, act, vmnz, act_list, act_count, cov16, own8, stamp, pix16, rast, shaded, aa, aa_items, aa_alphas, aa_dcol, aa_counter, aa_region_cnt,
        g_shaded, cmat, gvert_aa, gvert_aa2, vacc, part, partials, gnormal, gvert, gvert_r, gtmp, lx_v, lx_a, rowg, tail_part, n_valid_f, one_third, seg_valid, cnt_part, silw, bins, bin_cursor, bin_state, bin_list, total;
};

static size_t take(size_t& off, size_t bytes)
{
    size_t at = off;
    off += (bytes + 255) & ~(size_t)255;
    return at;
}

// the row pitch of the step's per-pixel buffers (ABI 5): fmhr_ham_dims.pitch, 0 = W rounded up to a multiple of 64
static int eff_pitch(const fmhr_ham_dims* d) { return d->pitch > 0 ? d->pitch : default_row_pitch(d->W); }

static Layout make_layout(const fmhr_ham_dims* d)
{
    Layout L;
    size_t off = 0;
    const int Wp = eff_pitch(d);
    size_t V = d->V, nP = (size_t)d->n * d->H * Wp, nV = (size_t)d->n * d->V;     // per-pixel buffers: [n, H, Wp]
    L.vertices = take(off, V * 12);
    L.nsum = take(off, V * 12);
    L.normals = take(off, V * 12);
    L.attr8 = take(off, V * 32);
    L.sh_local = take(off, (size_t)d->n * 36);      // the step's n SH rows in the step's own order (ABI 6: the caller's may be the capture's, selected by view_idx)
    L.view_idx_ws = take(off, (size_t)d->n * 4);    // the step's view selection, clamped to the capture's views by the geometry kernel: what every later kernel reads
    L.pos = take(off, nV * 16);
    L.snap = take(off, nV * 8);
    L.silh = take(off, (size_t)d->n * d->F);
    L.raster_ws = take(off, raster_core_ws_bytes(d->n, d->F, d->H, Wp));
    L.dirty = take(off, (nP + 63) / 64);
    L.covw = take(off, ((nP + 63) / 64) * 8);
    L.act = take(off, (nP + 63) / 64);
    // valid_mask flags per segment: of the step's n views, or (ABI 6, n_src > 0) of all the capture's views — built once, not per batch
    L.vmnz = take(off, (((size_t)(d->n_src > d->n ? d->n_src : d->n) * d->H * Wp) + 63) / 64);
    L.act_list = take(off, ((nP + 63) / 64) * 4);
    L.act_count = take(off, 16);
    L.cov16 = take(off, (size_t)d->n * d->F * 8);  // coverage entries [n, F]: candidate-pixel mask | box origin and width (raster.hip)
    L.own8 = take(off, (size_t)d->n * d->F);      // per (view, face): low byte of the last step in which it owned a pixel
    L.stamp = take(off, 16);                       // step counter tagging the coverage entries
    L.pix16 = take(off, nP * 16);             // per covered pixel: (u, v, z/w, packed face id | flags | silhouette bits)
    L.rast = take(off, nP * 16);
    L.shaded = take(off, nP * 16);            // inspection only (fmhr_ham_densify re-shades into it): the step keeps no shaded image
    L.aa = take(off, nP * 16);
    L.aa_items = take(off, (size_t)d->aa_cap * 16);   // antialias work items (px, py, view, code) of the pairs that blend
    L.aa_alphas = take(off, (size_t)d->aa_cap * 4);   // and their alphas
    L.aa_dcol = take(off, (size_t)d->aa_cap * 16);    // and the colour difference of each pair (the backward's d alpha)
    L.aa_counter = take(off, 16);
    L.aa_region_cnt = take(off, 2048 * 4);            // items per region of the list (one region per workgroup of the AA kernels)
    L.g_shaded = take(off, nP * 16);
    L.cmat = take(off, (size_t)d->n * 64);
    L.gvert_aa = take(off, V * 12 * kAaReplicas);
    L.gvert_aa2 = take(off, V * 12 * kAaReplicas);             // view-sharded step: the silhouette share of the antialias position gradient
    L.vacc = take(off, V * 48);                                  // per vertex 3 float4: sums of the large-box faces' backward (shade.hip), self-cleaning
    L.part = take(off, (size_t)face_bwd_groups(d->n, d->H, Wp) * d->F * 3 * 48);   // [view group, corner slot (vf CSR order)] x 3 float4
    L.partials = take(off, 2048 * 4 * 3);           // per-workgroup partial sums of the persistent pixel kernels
    L.gnormal = take(off, V * 12);
    L.gvert = take(off, V * 12);
    L.gvert_r = take(off, V * 12);
    L.gtmp = take(off, V * 12);
    L.lx_v = take(off, V * 12);
    L.lx_a = take(off, V * 12);
    L.rowg = take(off, V * 12);
    L.tail_part = take(off, (size_t)mesh_tail_blocks(d->V) * 16);    // per-workgroup regulariser loss sums [4, blocks]
    L.n_valid_f = take(off, 16);                    // float copy of n_valid_local (a ready n_valid_global for the one-rank case)
    L.one_third = take(off, 16);                    // the constant 1/3: the "valid-pixel count" that leaves the L1 divisor out (view-sharded step)
    L.seg_valid = take(off, ((nP + 63) / 64) * 4);   // valid-pixel count of every segment (resolve pass)
    L.cnt_part = take(off, (((nP + 63) / 64 + 1023) / 1024) * 4);   // the same per workgroup of k_active_segments
    L.silw = take(off, ((nP + 63) / 64) * 8);        // per segment: pixels of silhouette faces (resolve pass -> antialias scan)
    // binned rasterizer (raster.hip): per 64-pixel segment a cursor (one per cache line) and room for bin_cap_entries() 8-B fragment
    // entries (8 KB per segment: 1.6 GB at the metric, of which a step touches the ~50 MB its fragments fill — sized for the
    // 288-GB part, not for the footprint)
    const bool binned = raster_binned_ok(d->F, d->H, d->W, Wp);
    const size_t tiles = binned ? raster_bin_count(d->n, d->H, Wp) : 0;
    L.bin_list = take(off, tiles * 8 + 16);                          // round 5: the step's list of segments with fragments (k_bins_compact -> k_raster_bins<true>)
    L.bin_state = take(off, tiles * 4 + 16);                         // per-segment state words, addressed DOWNWARDS from the cursors (raster.hip: seg_state_word)
    L.bin_cursor = take(off, tiles * 4 * kCursorStride + 16);
    L.bins = take(off, tiles * (size_t)bin_cap_entries(d->bin_cap) * sizeof(BinEntry));
    L.total = off;
    return L;
}

static int check_dims(const fmhr_ham_dims* d)
{
    FMHR_CHECK_ARG(d != nullptr, "ham: dims is null");
    FMHR_CHECK_ARG(d->n >= 1 && d->V >= 1 && d->F >= 1 && d->H >= 1 && d->W >= 1 && d->aa_cap >= 2048,
                   "ham: bad dims n=%d V=%d F=%d H=%d W=%d aa_cap=%d", d->n, d->V, d->F, d->H, d->W, d->aa_cap);
    FMHR_CHECK_ARG(d->H <= 8192 && d->W <= 8192 && d->F < (1 << 24) && d->V < (1 << 27),
                   "ham: H,W <= 8192, F < 2^24 and V < 2^27 required (face ids in fp32 / 24 bits; 32-bit byte offsets into the per-vertex tables)");
    FMHR_CHECK_ARG(d->bin_cap >= 0 && d->bin_cap <= (1 << 20), "ham: bin_cap must be 0 (default) or 64 .. 2^20, got %d", d->bin_cap);
    FMHR_CHECK_ARG(d->pitch == 0 || d->pitch == d->W || (d->pitch > d->W && d->pitch % 64 == 0 && d->pitch <= 8192),
                   "ham: pitch must be 0 (default: W rounded up to a multiple of 64), W, or a multiple of 64 above W; got %d (W=%d)", d->pitch, d->W);
    const int Wp = eff_pitch(d);
    FMHR_CHECK_ARG(d->n_src >= 0 && d->n_src <= 65535, "ham: n_src must be 0 (the arrays hold the step's views) or the number of views of the capture, got %d", d->n_src);
    FMHR_CHECK_ARG(d->n_src == 0 || ((long long)d->H * Wp) % 64 == 0,
                   "ham: a view selection (n_src > 0) needs H * pitch to be a multiple of 64 (every 64-pixel segment inside one view); H=%d pitch=%d", d->H, Wp);
    FMHR_CHECK_ARG(((long long)d->n * d->H * Wp) % 4 == 0, "ham: n*H*pitch must be a multiple of 4");
    FMHR_CHECK_ARG(d->n <= 65535 && ((long long)d->n * d->H * Wp) < (1ll << 30), "ham: n <= 65535 and n*H*pitch < 2^30 required (32-bit channel offsets in k_aa_bwd_items)");
    return FMHR_OK;
}

// the view selection of a call (ABI 6): view_idx goes with dims.n_src > 0
static int view_sel_from(const fmhr_ham_dims* d, const int32_t* view_idx, const char* who, ViewSel& sel)
{
    FMHR_CHECK_ARG((view_idx != nullptr) == (d->n_src > 0), "%s: view_idx must be given exactly when dims.n_src > 0 (n_src=%d, view_idx %s)", who,
                   d->n_src, view_idx ? "given" : "null");
    sel.idx = view_idx;
    sel.segv = make_udiv((unsigned)(((long long)d->H * eff_pitch(d) + 63) / 64));
    sel.H = d->H;
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

extern "C" int fmhr_ham_rasterizer_form(const fmhr_ham_dims* d)
{
    if (check_dims(d)) return -1;
    return (raster_binned_ok(d->F, d->H, d->W, eff_pitch(d)) && raster_binned_wanted(true)) ? 1 : 0;
}

extern "C" int fmhr_ham_pitch(const fmhr_ham_dims* d)
{
    if (check_dims(d)) return -1;
    return eff_pitch(d);
}

extern "C" size_t fmhr_ham_workspace_bytes(const fmhr_ham_dims* d)
{
    if (check_dims(d)) return 0;
    return make_layout(d).total;
}

extern "C" long long fmhr_ham_workspace_offset(const fmhr_ham_dims* d, const char* name)
{
    if (check_dims(d) || name == nullptr) return -1;
    Layout L = make_layout(d);
#define FMHR_OFF(x) if (!strcmp(name, #x)) return (long long)L.x
    FMHR_OFF(vertices); FMHR_OFF(nsum); FMHR_OFF(normals); FMHR_OFF(sh_local); FMHR_OFF(pos); FMHR_OFF(snap); FMHR_OFF(silh); FMHR_OFF(raster_ws);
    FMHR_OFF(dirty); FMHR_OFF(covw); FMHR_OFF(act); FMHR_OFF(cov16); FMHR_OFF(own8); FMHR_OFF(stamp); FMHR_OFF(pix16); FMHR_OFF(rast); FMHR_OFF(shaded); FMHR_OFF(aa);
    FMHR_OFF(aa_items); FMHR_OFF(aa_alphas); FMHR_OFF(aa_counter); FMHR_OFF(aa_region_cnt); FMHR_OFF(g_shaded); FMHR_OFF(cmat); FMHR_OFF(gvert_aa); FMHR_OFF(vacc); FMHR_OFF(part);
    FMHR_OFF(gnormal); FMHR_OFF(gvert); FMHR_OFF(n_valid_f); FMHR_OFF(silw); FMHR_OFF(bins); FMHR_OFF(bin_cursor); FMHR_OFF(bin_state); FMHR_OFF(total);
#undef FMHR_OFF
    set_error("ham_workspace_offset: unknown buffer '%s'", name);
    return -1;
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
    out->shaded = (float*)(w + L.shaded);
    out->aa = (float*)(w + L.aa);
    out->aa_counter = (int32_t*)(w + L.aa_counter);
    return FMHR_OK;
}

extern "C" int fmhr_ham_workspace_init(const fmhr_ham_dims* d, void* ws, size_t ws_bytes, void* stream)
{
    TRY(check_dims(d));
    Layout L = make_layout(d);
    FMHR_CHECK_ARG(ws && ws_bytes >= L.total, "ham_workspace_init: workspace too small or null");
    char* w = (char*)ws;
    HIPTRY(hipMemsetAsync(w + L.aa_counter, 0, 16, (hipStream_t)stream), "init aa_counter");
    HIPTRY(hipMemsetAsync(w + L.aa_region_cnt, 0, 2048 * 4, (hipStream_t)stream), "init aa_region_cnt");   // regions beyond the grid stay 0
    // self-cleaning accumulator: k_mesh_tail_a zeroes each entry after consuming it
    HIPTRY(hipMemsetAsync(w + L.gvert_aa, 0, (size_t)d->V * 12 * kAaReplicas, (hipStream_t)stream), "init gvert_aa");
    HIPTRY(hipMemsetAsync(w + L.gvert_aa2, 0, (size_t)d->V * 12 * kAaReplicas, (hipStream_t)stream), "init gvert_aa2");
    HIPTRY(hipMemsetAsync(w + L.vacc, 0, (size_t)d->V * 48, (hipStream_t)stream), "init vacc");
    {
        const float third = 1.0f / 3.0f;           // 3 * (1/3) == 1 exactly in fp32: the kernels' divisor 3 * n_valid becomes 1
        HIPTRY(hipMemcpyAsync(w + L.one_third, &third, 4, hipMemcpyHostToDevice, (hipStream_t)stream), "init one_third");
        HIPTRY(hipStreamSynchronize((hipStream_t)stream), "init one_third sync");      // `third` lives on this stack frame
    }
    // coverage entries are recognised by the step stamp: start with no entry valid, counter at 0
    HIPTRY(hipMemsetAsync(w + L.cov16, 0xFF, (size_t)d->n * d->F * 8, (hipStream_t)stream), "init cov16");
    HIPTRY(hipMemsetAsync(w + L.stamp, 0, 16, (hipStream_t)stream), "init stamp");
    HIPTRY(hipMemsetAsync(w + L.own8, 0, (size_t)d->n * d->F, (hipStream_t)stream), "init own8");
    HIPTRY(hipMemsetAsync(w + L.bin_cursor, 0, L.bins - L.bin_cursor, (hipStream_t)stream), "init bin cursors");   // self-cleaning afterwards
    // every segment's words count as live until the second pass has seen the segment once (bit 1 of every byte: kSegWordsLive, bit 0 clear)
    HIPTRY(hipMemsetAsync(w + L.bin_state, 0x02, L.bin_cursor - L.bin_state, (hipStream_t)stream), "init bin state");
    if (raster_binned_ok(d->F, d->H, d->W, eff_pitch(d)))   // the bin-overflow word of the launch pair, two words below the lowest state word
        HIPTRY(hipMemsetAsync(w + L.bin_cursor - 4 * (raster_bin_count(d->n, d->H, eff_pitch(d)) + 2), 0, 4, (hipStream_t)stream), "init bin overflow word");
    return raster_sparse_init(w + L.raster_ws, (unsigned char*)(w + L.dirty), d->n, d->H, eff_pitch(d), (hipStream_t)stream);
}

extern "C" int fmhr_ham_densify(const fmhr_ham_dims* d, const int32_t* tri, const float* sh, void* ws, size_t ws_bytes, void* stream)
{
    TRY(check_dims(d));
    Layout L = make_layout(d);
    FMHR_CHECK_ARG(ws && ws_bytes >= L.total && tri, "ham_densify: workspace too small or null pointer");
    char* w = (char*)ws;
    hipStream_t st = (hipStream_t)stream;
    // the step stores neither clip positions nor a shaded image: both views are produced here, bit-identical to what the
    // kernels computed on the fly (same FMA chains / same shading statements)
    TRY(clip_positions_impl((const float*)(w + L.vertices), (const float*)(w + L.cmat), d->n, d->V, (float*)(w + L.pos), st));
    (void)sh;                                                        // (ABI 6: the step's SH rows are in the workspace, in the step's order)
    TRY(reshade_dense_impl((const unsigned long long*)(w + L.covw), (const float*)(w + L.pix16), tri, (const float*)(w + L.attr8), (const float*)(w + L.sh_local), d->n,
                           d->H, eff_pitch(d), (float*)(w + L.shaded), st));
    return densify_impl((const unsigned long long*)(w + L.covw), (const unsigned char*)(w + L.act), (long long)d->n * d->H * eff_pitch(d),
                        (const float*)(w + L.pix16), (float*)(w + L.rast), (float*)(w + L.aa), st);
}

extern "C" int fmhr_ham_forward(const fmhr_ham_dims* d, const float* vertices_tmp, const float* delta, const float* albedo,
                                const float* sh, const float* w2c, const float* proj, const float* mask,
                                const float* valid_mask, const int32_t* view_idx, const int32_t* tri, const int32_t* opp, const int32_t* vf_ptr,
                                const int32_t* vf_idx, int valid_mask_changed, int32_t* n_valid_local, void* ws,
                                size_t ws_bytes, void* stream)
{
    TRY(check_dims(d));
    ViewSel sel;
    TRY(view_sel_from(d, view_idx, "ham_forward", sel));
    Layout L = make_layout(d);
    FMHR_CHECK_ARG(ws && ws_bytes >= L.total, "ham_forward: workspace too small (%zu < %zu)", ws_bytes, L.total);
    FMHR_CHECK_ARG(vertices_tmp && delta && albedo && sh && w2c && proj && mask && valid_mask && tri && opp && vf_ptr && vf_idx &&
                       n_valid_local, "ham_forward: null pointer");
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    float* vertices = (float*)(w + L.vertices);
    float* rast = (float*)(w + L.rast);
    const int Wp = eff_pitch(d);                                     // row pitch of every per-pixel buffer of the step; d->W stays the image width
    const long long nP = (long long)d->n * d->H * Wp;
    unsigned char* act = (unsigned char*)(w + L.act);
    unsigned long long* covw = (unsigned long long*)(w + L.covw);
    {
        ProfScope ps(FMHR_SLOT_GEOMETRY, st);
        // one launch: clip transform of all views + vertex normals / packed attributes; vertices = vertices_tmp + delta is formed
        // on the fly by both parts and stored for everything after them
        TRY(geometry_impl(vertices_tmp, delta, w2c, proj, d->n, d->V, nullptr, d->H, d->W, (int2*)(w + L.snap), (float*)(w + L.cmat), tri,
                          vf_ptr, vf_idx, (float*)(w + L.nsum), (float*)(w + L.normals), albedo, (float*)(w + L.attr8), vertices,
                          (int*)(w + L.stamp), view_idx, sh, (float*)(w + L.sh_local), d->n_src, (int32_t*)(w + L.view_idx_ws), st));
        sh = (const float*)(w + L.sh_local);                         // from here on: the step's n rows, in the step's order
        if (view_idx != nullptr) sel.idx = (const int32_t*)(w + L.view_idx_ws);     // ... and the selection the geometry kernel clamped to [0, n_src)
    }
    {
        // rasterize; the resolve pass also shades the covered pixels, leaves the per-pixel backward word, marks the owning
        // (view, face)s and rearms the two segment-list cursors
        RasterFused fu;
        fu.cov16 = (int2*)(w + L.cov16); fu.stamp = (int*)(w + L.stamp); fu.own8 = (unsigned char*)(w + L.own8);
        fu.pix16 = (float4*)(w + L.pix16); fu.mask = mask;
        fu.silh = (const unsigned char*)(w + L.silh);
        fu.attr8 = (const float4*)(w + L.attr8); fu.sh = sh;
        fu.aa_init = (float4*)(w + L.aa); fu.seg_valid = (int*)(w + L.seg_valid); fu.silw = (unsigned long long*)(w + L.silw);
        fu.verts = vertices; fu.cmat = (const float4*)(w + L.cmat);
        fu.rearm2 = (int*)(w + L.act_count); fu.rearm_aa = (int*)(w + L.aa_counter);
        fu.bins = nullptr; fu.bin_cursor = nullptr; fu.bin_flag = nullptr; fu.bin_cap = 0; fu.bin_list = nullptr; fu.bin_list_cnt = nullptr;
        fu.segx = make_udiv((unsigned)((Wp + 63) / 64));            // used when Wp != d->W: segment -> row of the caller's [n,H,W] mask
        fu.sel = sel;
        if (raster_binned_ok(d->F, d->H, d->W, Wp)) {
            fu.bins = (BinEntry*)(w + L.bins); fu.bin_cursor = (int*)(w + L.bin_cursor); fu.bin_cap = bin_cap_entries(d->bin_cap);
            fu.bin_flag = (int*)(w + L.aa_counter) + 2;             // sticky overflow flag next to the antialias list's (read by the host)
            fu.bin_list = (int2*)(w + L.bin_list); fu.bin_list_cnt = (int*)(w + L.stamp) + 2;   // (cursor zeroed by the geometry kernel's first thread)
        }
        TRY(rasterize_fwd_impl(nullptr, (const int2*)(w + L.snap), tri, opp, (unsigned char*)(w + L.silh), d->n, d->V, d->F, d->H, d->W, Wp,
                               rast, w + L.raster_ws, (unsigned char*)(w + L.dirty), covw, &fu, st));
    }
    {
        ProfScope ps(FMHR_SLOT_SHADE_FWD, st);
        // (with a view selection the flags are those of ALL the capture's views, addressed through it: built once for every batch)
        if (valid_mask_changed) TRY(valid_mask_segments_impl(valid_mask, d->n_src > 0 ? d->n_src : d->n, d->H, d->W, Wp, (unsigned char*)(w + L.vmnz), st));
        TRY(active_segments_impl(covw, (const unsigned char*)(w + L.vmnz), nP, Wp, act, (int*)(w + L.act_list),
                                 (int*)(w + L.act_count), (const int*)(w + L.seg_valid), (float*)(w + L.cnt_part),
                                 (float*)(w + L.aa), sel, st));
    }
    ProfScope ps_aa(FMHR_SLOT_AA_FWD, st);
    TRY(aa_fwd_rgba((const float*)(w + L.attr8), sh, (float*)(w + L.aa_dcol), (const float*)(w + L.pix16), nullptr, vertices, (const float*)(w + L.cmat), tri, opp, (const unsigned char*)(w + L.silh), (const int*)(w + L.act_list),
                    (const int*)(w + L.act_count), covw, (const unsigned long long*)(w + L.silw), d->n, d->V, d->F,
                    d->H, d->W, Wp, (float*)(w + L.aa), (int32_t*)(w + L.aa_items), (float*)(w + L.aa_alphas),
                    (int32_t*)(w + L.aa_counter), (int32_t*)(w + L.aa_region_cnt), d->aa_cap, (const float*)(w + L.cnt_part),
                    active_segments_blocks(nP), n_valid_local, (float*)(w + L.n_valid_f), st));
    return FMHR_OK;
}

extern "C" int fmhr_ham_backward(const fmhr_ham_dims* d, const fmhr_ham_weights* wt, const float* vertices_tmp,
                                 const float* delta, const float* albedo, const float* sh, const float* w2c, const float* proj,
                                 const float* img, const float* mask, const float* valid_mask, const int32_t* view_idx, const int32_t* tri,
                                 const int32_t* opp, const int32_t* row_ptr, const int32_t* col_idx, const int32_t* col_mult,
                                 const int32_t* vf_ptr, const int32_t* vf_idx, const int32_t* corner_slot,
                                 const float* n_valid_global, float* losses, float* gdelta,
                                 float* galbedo, float* gdelta_rest, float* galbedo_rest, const fmhr_adam_param* adam_delta,
                                 const fmhr_adam_param* adam_albedo, void* ws, size_t ws_bytes, void* stream)
{
    (void)vertices_tmp; (void)opp;
    TRY(check_dims(d));
    ViewSel sel;
    TRY(view_sel_from(d, view_idx, "ham_backward", sel));
    Layout L = make_layout(d);
    FMHR_CHECK_ARG(ws && ws_bytes >= L.total, "ham_backward: workspace too small (%zu < %zu)", ws_bytes, L.total);
    FMHR_CHECK_ARG(wt && delta && albedo && sh && w2c && proj && img && mask && valid_mask && tri && row_ptr && col_idx && col_mult &&
                       vf_ptr && vf_idx && corner_slot && losses && gdelta && galbedo, "ham_backward: null pointer");
    FMHR_CHECK_ARG((gdelta_rest != nullptr) == (galbedo_rest != nullptr), "ham_backward: gdelta_rest and galbedo_rest go together");
    const bool split = gdelta_rest != nullptr;         // view-sharded caller: photometric share apart, L1 divisor left out
    FMHR_CHECK_ARG(split || n_valid_global != nullptr, "ham_backward: n_valid_global is null");
    FMHR_CHECK_ARG(!split || (adam_delta == nullptr && adam_albedo == nullptr),
                   "ham_backward: the fused optimiser update belongs to fmhr_ham_combine when the outputs are split");
    FMHR_CHECK_ARG(wt->mse_denominator > 0.f, "ham_backward: mse_denominator must be > 0");
    FMHR_CHECK_ARG(wt->reg_v0 >= 0 && wt->reg_v0 <= wt->reg_v1 && wt->reg_v1 <= d->V, "ham_backward: bad regulariser vertex range [%d, %d)",
                   wt->reg_v0, wt->reg_v1);
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    float* vertices = (float*)(w + L.vertices);
    float* gvert_aa = (float*)(w + L.gvert_aa);
    // the count the per-pixel kernels divide the L1 term by: the caller's batch-global one, or "1/3" (divisor 3 * n = 1) when split
    const float* nv_div = split ? (const float*)(w + L.one_third) : n_valid_global;
    float* gvert = (float*)(w + L.gvert);
    const int Wp = eff_pitch(d);
    size_t nP = (size_t)d->n * d->H * Wp;
    const float rg = wt->reg_grad_scale;
    const size_t nB = 2048;
    sh = (const float*)(w + L.sh_local);                             // the forward left the step's n SH rows here, in the step's order (ABI 6)
    if (view_idx != nullptr) {                                       // ... and the view selection it rendered, clamped to the capture's views: the backward
        view_idx = (const int32_t*)(w + L.view_idx_ws);              // differentiates exactly what the forward rendered, whatever the caller passes here
        sel.idx = view_idx;
    }

    {
        ProfScope ps(FMHR_SLOT_LOSS_PIX, st);
        // losses on the antialiased rgba + their per-pixel gradients
        TRY(loss_pix_impl((const unsigned long long*)(w + L.covw), (float*)(w + L.aa), img, mask, valid_mask, (long long)nP, d->W, Wp, sel,
                          (const int*)(w + L.act_list), (const int*)(w + L.act_count), nv_div,
                          wt->sfs, wt->mask, wt->mse_denominator, (float*)(w + L.g_shaded),
                          (float*)(w + L.partials) + nB, (float*)(w + L.partials) + 2 * nB, losses, st));
    }
    {
        // antialias backward: colour grads into g_shaded (atomics over the short queue), position grads (clip-transformed) into gvert_aa
        ProfScope ps(FMHR_SLOT_AA_BWD, st);
        TRY(aa_bwd_rgba((const float*)(w + L.aa_dcol), (const float*)(w + L.pix16), nullptr, (const float*)(w + L.vertices), tri, (const float*)(w + L.aa), img, mask, valid_mask,
                        (const unsigned long long*)(w + L.covw), nv_div, wt->sfs, wt->mask, wt->mse_denominator,
                        (const int32_t*)(w + L.aa_items),
                        (const float*)(w + L.aa_alphas), (const int32_t*)(w + L.aa_region_cnt), d->aa_cap, d->n, d->V, d->H, d->W, Wp,
                        (float*)(w + L.g_shaded), (const float*)(w + L.cmat), gvert_aa, split ? (float*)(w + L.gvert_aa2) : nullptr, view_idx, st));
    }
    {
        // shade + interpolate + rasterize + clip-transform backward, face-parallel (shade.hip: k_face_bwd)
        ProfScope ps(FMHR_SLOT_SHADE_BWD, st);
        TRY(face_bwd_impl((const float*)(w + L.pix16), tri, nullptr, (float*)(w + L.attr8), sh, (float*)(w + L.g_shaded),
                          w + L.cov16, (const unsigned char*)(w + L.own8), (const int*)(w + L.stamp), corner_slot,
                          (const float*)(w + L.cmat), vertices, d->n, d->V, d->F, d->H, d->W, Wp, (float*)(w + L.part),
                          (const int2*)(w + L.snap), (const int2*)(w + L.raster_ws + raster_keys_bytes(d->n, d->H, Wp) + 16),
                          (const int*)(w + L.raster_ws + raster_keys_bytes(d->n, d->H, Wp)), (const unsigned long long*)(w + L.covw),
                          (float*)(w + L.vacc), st));
    }
    {
        // everything per-vertex after the face-parallel backward, in two kernels (mesh.hip: k_mesh_tail_a / _b)
        ProfScope ps(FMHR_SLOT_REGULARISERS, st);
        MeshTailArgs A;
        A.part = (const float4*)(w + L.part); A.gvert_aa = gvert_aa; A.G = face_bwd_groups(d->n, d->H, Wp);
        A.gvert_aa2 = split ? (float*)(w + L.gvert_aa2) : nullptr; A.col_mult = col_mult;
        A.vacc = (float*)(w + L.vacc);
        A.gacc_r = (float*)(w + L.gvert_r);
        A.gdelta = gdelta; A.galbedo = galbedo; A.gd_rest = gdelta_rest; A.ga_rest = galbedo_rest;
        TRY(adam_tensor_from(A.ad_d, adam_delta, 3ll * d->V));
        TRY(adam_tensor_from(A.ad_a, adam_albedo, 3ll * d->V));
        A.verts = vertices; A.delta = delta; A.albedo = albedo; A.nsum = (const float*)(w + L.nsum);
        A.tri = tri; A.vf_ptr = vf_ptr; A.vf_idx = vf_idx; A.row_ptr = row_ptr; A.col = col_idx;
        A.V = d->V; A.F = d->F;
        A.v0 = (rg != 0.f) ? wt->reg_v0 : 0; A.v1 = (rg != 0.f) ? wt->reg_v1 : 0;
        A.lap_w = wt->lap / (float)d->V; A.alb_w = wt->albedo / (float)d->V; A.edge_w = wt->edge / (3.0f * (float)d->F);
        A.delta_w = wt->delta / (float)d->V; A.edge_mean = wt->edge_length_mean; A.rg = rg;
        A.gs = (float*)(w + L.gtmp); A.rowg_v = (float*)(w + L.rowg); A.rowg_a = (float*)(w + L.lx_a);
        A.gacc_v = gvert; A.gacc_a = (float*)(w + L.lx_v); A.loss_part = (float*)(w + L.tail_part);
        // one extra workgroup of the second tail kernel turns the per-workgroup partials into the loss vector
        FinishArgs fin;
        fin.losses = losses; fin.n_valid = nv_div; fin.part_l1 = (const float*)(w + L.partials) + nB;
        fin.part_l2 = (const float*)(w + L.partials) + 2 * nB; fin.nb = (int)nB; fin.tail_part = (const float*)(w + L.tail_part);
        fin.tail_nb = mesh_tail_blocks(d->V); fin.w_sfs = wt->sfs; fin.w_mask = wt->mask; fin.mse_den = wt->mse_denominator;
        fin.reg_scale = (rg != 0.f) ? 1.f : 0.f;
        // the n_valid slot: one rank reports the count it was given; a view-sharded rank reports its LOCAL count (the slots are summed)
        fin.split = split ? 1 : 0;
        fin.nvalid_report = split ? (const float*)(w + L.n_valid_f) : n_valid_global;
        fin.nvalid_scale = split ? 1.f : ((rg != 0.f && wt->reg_v0 == 0) ? 1.f : 0.f);
        TRY(mesh_tail_impl(A, fin, st));
    }
    return FMHR_OK;
}

extern "C" int fmhr_ham_combine(int V, float* bucket, const fmhr_adam_param* adam_delta, const fmhr_adam_param* adam_albedo, void* stream)
{
    FMHR_CHECK_ARG(V >= 1 && bucket != nullptr, "ham_combine: bad V or null bucket");
    return ham_combine_impl(bucket, V, adam_delta, adam_albedo, (hipStream_t)stream);
}
