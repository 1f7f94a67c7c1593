/*
 * include/fmhr_ham.h — C ABI of the MI355X-native HAM renderer (libfmhr_ham.so, gfx950 only).
 *
 * This is the drop-in boundary for the one hot path of agnJason/FMHR: the stage-2 "HAM" iteration of
 * /root/reference/mesh_sfs_optim.py:253-310.  The reference reaches its native code through nvdiffrast's Python ops
 * (`import nvdiffrast.torch as dr`, mesh_sfs_optim.py:14) and the torch helpers of models/utils.py; each entry point
 * below names the reference interface it replaces.  Python binding: fmhr_amd/_lib.py (ctypes) — see INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer (hipMalloc'd; PyTorch-ROCm tensors' data_ptr()), fp32 / int32, contiguous;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); nothing synchronises the host,
 *     nothing allocates, nothing keeps global mutable state: calls are re-entrant across streams and graph-capturable;
 *   - return 0 on success, <0 on error (FMHR_E_*); fmhr_last_error() gives the thread-local message;
 *   - gradient outputs documented "+=" are ACCUMULATED into caller-zeroed buffers, the others are overwritten;
 *   - scratch memory is caller-allocated: query the size with the matching *_workspace_bytes().
 * There is no CPU back end: without a GPU every compute call fails with FMHR_E_HIP.
 */
#ifndef FMHR_HAM_H
#define FMHR_HAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMHR_ABI_VERSION 6
#define FMHR_OK 0
#define FMHR_E_ARG (-1)   /* bad shape / null pointer / workspace too small */
#define FMHR_E_HIP (-2)   /* HIP runtime error (message has hipGetErrorString) */

int fmhr_abi_version(void);
const char* fmhr_last_error(void);

/* ------------------------------------------------------------------------------------------------ topology (once per mesh)
 * Replaces nvdiffrast's per-call edge hash (`dr.antialias(..., topology_hash=None)`, mesh_sfs_optim.py:274,287 rebuild it
 * twice per step).  opp[f*3+k] = the vertex opposite edge k (= edge (tri[f][k+1], tri[f][k+2])) in the neighbouring face,
 * or -1 on a boundary edge.  An edge shared by >2 faces keeps the two lowest codes 3*f+k. */
size_t fmhr_topology_workspace_bytes(int F);
int fmhr_topology_build(const int32_t* tri, int F, int V, int32_t* opp, void* ws, size_t ws_bytes, void* stream);

/* Unique-neighbour adjacency (CSR, rows sorted) for the uniform Laplacian of models/utils.py:661-693 (`compute_laplacian`,
 * which the reference rebuilds twice per step, mesh_sfs_optim.py:292-293).  Needs `opp` from fmhr_topology_build.
 * row_ptr [V+1], col_idx [>= 6F] (nnz = row_ptr[V] <= 6F). */
size_t fmhr_adjacency_workspace_bytes(int V, int F);
int fmhr_adjacency_build(const int32_t* tri, int F, int V, int32_t* row_ptr, int32_t* col_idx,
                         void* ws, size_t ws_bytes, void* stream);

/* faces per adjacency entry: col_mult[e] = number of faces that hold the edge (v, col_idx[e]) of row v (1 on the boundary, 2 inside
 * a manifold mesh).  The edge-length hinge of mesh_sfs_optim.py:296-302 is a mean over the 3F directed FACE edges; with this table
 * it is evaluated in the same neighbour walk as the uniform Laplacians.  col_mult [nnz], overwritten. */
int fmhr_edge_multiplicity_build(const int32_t* tri, int F, int V, const int32_t* row_ptr, const int32_t* col_idx, int nnz,
                                 int32_t* col_mult, void* stream);

/* vertex -> incident face corners (CSR, rows sorted; entry = 3*face + corner): the per-vertex kernels of the fused step walk it
 * instead of scattering float atomics.  vf_ptr [V+1], vf_idx [3F]. */
size_t fmhr_vertex_faces_workspace_bytes(int V);
int fmhr_vertex_faces_build(const int32_t* tri, int F, int V, int32_t* vf_ptr, int32_t* vf_idx,
                            void* ws, size_t ws_bytes, void* stream);

/* inverse of vf_idx: corner_slot [3F], corner_slot[3*face + corner] = position of that corner in its vertex's CSR row.  The
 * face-parallel backward of the fused step stores each corner's gradient sums at its slot. */
int fmhr_corner_slots_build(const int32_t* vf_idx, int F, int32_t* corner_slot, void* stream);

/* ------------------------------------------------------------------------------------------------ nvdiffrast-shaped primitives
 * dr.rasterize(glctx, pos, tri, resolution=(H,W))  — mesh_sfs_optim.py:142,212,267; get_data.py:251.
 * pos [n,V,4] clip space, tri [F,3]; rast [n,H,W,4] = (u, v, z/w, tri_id+1) or zeros.  Rules: DESIGN.md §3. */
size_t fmhr_rasterize_workspace_bytes(int n, int V, int F, int H, int W);
int fmhr_rasterize_fwd(const float* pos, const int32_t* tri, int n, int V, int F, int H, int W,
                       float* rast, void* ws, size_t ws_bytes, void* stream);
/* backward of rast[...,0:2] w.r.t. pos: dy [n,H,W,4] (only .x,.y read); gpos [n,V,4] += */
int fmhr_rasterize_bwd(const float* pos, const int32_t* tri, const float* rast, const float* dy,
                       int n, int V, int F, int H, int W, float* gpos, void* stream);

/* dr.interpolate(attr, rast, tri) — mesh_sfs_optim.py:143,214,269; get_data.py:253.  attr [an,V,C], an in {1,n}. */
int fmhr_interpolate_fwd(const float* attr, const float* rast, const int32_t* tri,
                         int n, int an, int V, int F, int H, int W, int C, float* out, void* stream);
/* gattr [an,V,C] +=, grast [n,H,W,4] overwritten (.z,.w = 0) */
int fmhr_interpolate_bwd(const float* attr, const float* rast, const int32_t* tri, const float* dy,
                         int n, int an, int V, int F, int H, int W, int C, float* gattr, float* grast, void* stream);

/* dr.antialias(color, rast, pos, tri) — mesh_sfs_optim.py:146,147,217,219,274,287.
 * items [cap,4] int32 = (px, py, view, code), alphas [cap], counter [2] int32 = (items found, overflow flag): the work queue
 * the backward replays.  counter must be zeroed by the caller (hipMemsetAsync) before the forward. */
int fmhr_antialias_fwd(const float* color, const float* rast, const float* pos, const int32_t* tri, const int32_t* opp,
                       int n, int V, int F, int H, int W, int C, float* out,
                       int32_t* items, float* alphas, int32_t* counter, int cap, void* stream);
/* gcolor [n,H,W,C] overwritten (= dy + blend terms); gpos [n,V,4] += */
int fmhr_antialias_bwd(const float* color, const float* rast, const float* pos, const int32_t* tri, const float* dy,
                       const int32_t* items, const float* alphas, const int32_t* counter, int cap,
                       int n, int V, int F, int H, int W, int C, float* gcolor, float* gpos, void* stream);

/* ------------------------------------------------------------------------------------------------ models/utils.py helpers
 * get_normals(vertices, faces) — models/utils.py:508-548.  verts [b,V,3]; nsum [b,V,3] = un-normalised sums (saved for the
 * backward), normals [b,V,3] = nsum / max(|nsum|, 1e-6). */
int fmhr_vertex_normals_fwd(const float* verts, const int32_t* tri, int b, int V, int F,
                            float* nsum, float* normals, void* stream);
/* gverts [b,V,3] += ; gtmp [b,V,3] scratch */
int fmhr_vertex_normals_bwd(const float* verts, const int32_t* tri, const float* nsum, const float* gnormals,
                            int b, int V, int F, float* gtmp, float* gverts, void* stream);

/* get_radiance(coeff, normal, 3) — models/utils.py:208-226.  coeff [P,9] (per_pixel=1) or [9]; normal [P,3]; out [P]. */
int fmhr_sh_radiance_fwd(const float* coeff, const float* normal, int P, int per_pixel, float* out, void* stream);
/* gcoeff [P,9] overwritten when per_pixel, else [9] += ; gnormal [P,3] overwritten */
int fmhr_sh_radiance_bwd(const float* coeff, const float* normal, const float* dy, int P, int per_pixel,
                         float* gcoeff, float* gnormal, void* stream);

/* laplacian_smoothing(x, faces, "uniform") — models/utils.py:696-722 on a cached CSR.  x [V,C] (C<=4);
 * loss[0] += weight * (1/V) sum_i |(Lx)_i|;  lx [V,C] scratch/out.  Backward: gx [V,C] += gscale * d loss / d x. */
int fmhr_laplacian_uniform_fwd(const float* x, const int32_t* row_ptr, const int32_t* col_idx, int V, int C,
                               float weight, float* lx, float* loss, void* stream);
int fmhr_laplacian_uniform_bwd(const float* lx, const int32_t* row_ptr, const int32_t* col_idx, int V, int C,
                               float gscale, float* rowg, float* gx, void* stream);

/* NCC(ref, src, ref_valid_mask, src_valid_mask) — models/ncc_utils.py:4-35 (never called by the reference; named by the
 * north star).  ref [1,Np,Npx], src / src_valid_mask [nv,Np,Npx] -> out [nv,Np].  ref_valid_mask is unused there too.
 * Backward: gref [1,Np,Npx] += (caller-zeroed; summed over views), gsrc [nv,Np,Npx] overwritten; either may be null. */
int fmhr_ncc_fwd(const float* ref, const float* src, const float* src_valid_mask, int nviews, int npoints, int npixels,
                 float* out, void* stream);
int fmhr_ncc_bwd(const float* ref, const float* src, const float* src_valid_mask, const float* dy, int nviews, int npoints,
                 int npixels, float* gref, float* gsrc, void* stream);

/* torch.optim.Adam(params, lr, betas, eps).step() for one parameter tensor (amsgrad=False, weight_decay=0) —
 * mesh_sfs_optim.py:244,310.  `step` is the 1-based step count; exp_avg / exp_avg_sq are the caller-owned state (zeros at start). */
int fmhr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                   float beta1, float beta2, float eps, int step, void* stream);
/* the same for two parameters in ONE launch (the HAM loops step exactly two per iteration) */
int fmhr_adam_step2(float* param_a, const float* grad_a, float* exp_avg_a, float* exp_avg_sq_a, long long n_a, float lr_a,
                    float beta1_a, float beta2_a, float eps_a, int step_a, float* param_b, const float* grad_b,
                    float* exp_avg_b, float* exp_avg_sq_b, long long n_b, float lr_b, float beta1_b, float beta2_b,
                    float eps_b, int step_b, void* stream);

/* Optional fused optimiser update (same rule as fmhr_adam_step): passed to fmhr_ham_backward (one-rank form) or fmhr_ham_combine
 * (view-sharded form), the parameter is updated IN PLACE by the kernel that forms its final gradient — no extra launch (the two
 * [V,3] parameters of a stage-2 iteration are 0.3 M floats: a launch of their own costs more than the arithmetic).  The gradient
 * outputs are still written.  NULL: no update. */
typedef struct fmhr_adam_param {
    float* param; float* exp_avg; float* exp_avg_sq;     /* [V,3] each; the moments are caller-owned state (zeros at start) */
    float lr, beta1, beta2, eps;
    int32_t step;                                        /* 1-based count of THIS update */
} fmhr_adam_param;

/* ------------------------------------------------------------------------------------------------ fused stage-2 step
 * One full forward+backward of mesh_sfs_optim.py:253-308 for the n views this rank owns.  See fmhr_ham_forward() / fmhr_ham_backward() below. */
typedef struct fmhr_ham_dims {
    int32_t n, V, F, H, W;      /* views on this rank, vertices, faces, image size */
    int32_t nnz;                /* adjacency nnz (row_ptr[V]) */
    int32_t aa_cap;             /* capacity of the antialias work-item list (pairs that blend, ∝ silhouette length); on overflow
                                   aa_counter[1] of fmhr_ham_views is set and the excess pairs get no gradient */
    int32_t bin_cap;            /* ABI 4: entries per fragment bin of the binned rasterizer (one bin per 64-pixel segment, 8 B per
                                   entry; 0 = the default 512 = 8 fragments per pixel; the host loop grows it on overflow).  Part of the workspace layout: the SAME
                                   value must be passed to every call on a workspace.  A full bin sets aa_counter[2] (sticky) and
                                   drops the excess fragments */
    int32_t pitch;              /* ABI 5: row pitch, in pixels, of the step's OWN per-pixel buffers (pixel words, antialiased image,
                                   image gradient, coverage words, fragment bins ...).  0 = the default: W rounded up to a multiple
                                   of 64, so that every 64-pixel segment — one wave of every per-pixel kernel, one fragment bin — lies
                                   in ONE row of ONE view whatever W is (W = 334: conf/ih_sfs.conf:14, conf/ih_sfsseq.conf:11) and
                                   such images run the binned rasterizer and the row-aligned antialias kernels.  Explicit values: W
                                   (flat segments that straddle rows: the ABI-4 layout, kept for A/B runs) or any multiple of 64
                                   above W.  Part of the workspace layout like bin_cap.  The CALLER'S tensors are not affected: img
                                   stays [n,H,W,3], mask / valid_mask [n,H,W]; only the inspection views of fmhr_ham_workspace_views
                                   have rows of `pitch` pixels (fmhr_ham_pitch) */
    int32_t n_src;              /* ABI 6: views of the CAPTURE the caller's per-view arrays hold (sh, w2c, proj, img, mask, valid_mask:
                                   [n_src, ...]) when a step renders a SUBSET of them, chosen per step by `view_idx` of fmhr_ham_forward /
                                   fmhr_ham_backward — the minibatch of mesh_sfs_optim.py:249,255-260 (`perm[k:k+batch]`) without the six
                                   gathers.  0 = the arrays hold exactly the n views of the step (view_idx must then be NULL).  Part of
                                   the workspace layout (the per-segment valid_mask flags are kept per SOURCE view: built once, not
                                   once per batch); n_src > 0 needs H * pitch % 64 == 0 (always true with the default pitch) */
} fmhr_ham_dims;

typedef struct fmhr_ham_weights {
    float sfs, lap, albedo, mask, edge, delta;   /* conf/demo_sfs.conf:16-22 */
    float edge_length_mean;                      /* mesh_sfs_optim.py:188 */
    float mse_denominator;                       /* n_global * H * W  (mesh_sfs_optim.py:295) */
    float reg_grad_scale;                        /* 1: add the view-independent regulariser gradients (lap, albedo-lap, edge,
                                                    delta) to gdelta/galbedo; 0: leave them out (all ranks but one in the
                                                    view-sharded loop, so the all-reduce sums them once) */
    int32_t reg_v0, reg_v1;                      /* the view-independent terms (losses and gradients) are evaluated for the
                                                    vertices [reg_v0, reg_v1) only: ranks of a view-sharded loop take disjoint
                                                    ranges that cover [0, V) and the all-reduce puts them together (a
                                                    Laplacian row, a vertex's edge-hinge gradient, a face's edge loss via its
                                                    corner 0 all belong to one vertex).  One rank: 0, V */
} fmhr_ham_weights;

/* loss vector layout (device, float[8]): sfs, lap, albedo, mask, edge, delta, total, n_valid_global as used */
#define FMHR_LOSS_SFS 0
#define FMHR_LOSS_LAP 1
#define FMHR_LOSS_ALBEDO 2
#define FMHR_LOSS_MASK 3
#define FMHR_LOSS_EDGE 4
#define FMHR_LOSS_DELTA 5
#define FMHR_LOSS_TOTAL 6
#define FMHR_LOSS_NVALID 7

size_t fmhr_ham_workspace_bytes(const fmhr_ham_dims* d);
/* ABI 5: the row pitch in effect for these dims (d->pitch, or W rounded up to a multiple of 64 when that is 0); -1 on bad dims */
int fmhr_ham_pitch(const fmhr_ham_dims* d);
/* ABI 4: which rasterizer fmhr_ham_forward takes for these dims: 1 = per-segment fragment bins resolved in LDS (micro-triangle mesh:
 * H*W <= 16 F, pitch % 64 == 0 — any W with the default pitch — F < 2^22; the environment variable FMHR_RASTER_BINNED=0 switches it
 * off for A/B runs), 0 = global
 * atomicMin depth keys + resolve pass (everything else, and the stand-alone dr.rasterize op: nvdiffrast mesh_sfs_optim.py:267). */
int fmhr_ham_rasterizer_form(const fmhr_ham_dims* d);

/* Phase A: geometry + rasterize + shade + antialias (forward up to the AA images) and the local valid-pixel count.
 * Phase B: losses + the whole backward.
 *   vertices_tmp, delta, albedo [V,3]; sh [n,9]; w2c, proj [n,4,4] (row-vector convention, get_data.py:96-97);
 *   img [n,H,W,3]; mask, valid_mask [n,H,W]; tri [F,3]; opp [F,3]; row_ptr [V+1]; col_idx, col_mult [nnz]; vf_ptr [V+1]; vf_idx [3F];
 *   corner_slot [3F] (fmhr_corner_slots_build).
 *   view_idx (ABI 6): NULL, or — with dims.n_src > 0 — device int32 [n], all in [0, n_src): view b of the step is view view_idx[b] of
 *   the caller's arrays, which are then [n_src, ...] (the whole capture, resident): sh, w2c, proj, img, mask, valid_mask.  The kernels
 *   index through it (one scalar load per wave); nothing is gathered or copied.  The forward CLAMPS every entry to [0, n_src) and keeps the
 *   clamped selection in the workspace: that copy is what every later kernel of the step — the backward included — reads, so an index out
 *   of range is never an out-of-bounds access; it sets a sticky word (int32 at fmhr_ham_workspace_offset(d, "stamp") + 4) that the host
 *   reads with the overflow flags (HamStage2.check_overflow raises).  The backward differentiates the selection the forward rendered;
 *   its own view_idx argument only has to be non-NULL.  Replaces the loop's per-batch gathers
 *   `w2cs[perm[k:k+batch]]` ... `sh_coeffs[perm[k:k+batch]]` (mesh_sfs_optim.py:255-260).
 *   n*H*pitch must be a multiple of 4 (always true with the default pitch).
 *   valid_mask_changed: pass 1 on the first call on a workspace and whenever valid_mask's CONTENTS may differ from the previous
 *   call (its per-segment non-zero flags are cached in the workspace; the mask is constant over a HAM run, :158).
 *   n_valid_local: device int32[1] written by phase A, which also leaves float(n_valid_local) in the workspace at
 *   fmhr_ham_workspace_offset(d, "n_valid_f").
 * F.l1_loss divides by the batch-GLOBAL 3*Pv (mesh_sfs_optim.py:289).  Two ways to call phase B:
 *   ONE RANK (gdelta_rest = galbedo_rest = NULL): n_valid_global = device float[1], normally the "n_valid_f" address.
 *     losses: device float[8]; gdelta, galbedo [V,3] overwritten with the whole gradient.
 *   VIEW-SHARDED (gdelta_rest, galbedo_rest non-NULL; n_valid_global ignored): no exchange between the phases.  gdelta / galbedo
 *     receive the PHOTOMETRIC share P of this rank's views with the L1 divisor left out (per-pixel gradient +-w_sfs), gdelta_rest /
 *     galbedo_rest everything else, R: the silhouette term of its views and the view-independent terms of its vertex range
 *     [reg_v0, reg_v1).  losses[FMHR_LOSS_SFS] = w_sfs * sum |d| (no divisor), losses[FMHR_LOSS_NVALID] = the LOCAL count.  The
 *     caller lays [gdelta | galbedo | gdelta_rest | galbedo_rest | losses] out as ONE bucket of 12V+8 floats, all-reduces it (SUM)
 *     once and calls fmhr_ham_combine. */
/* Must be called once on a fresh workspace (and after any aborted step): sets up the self-cleaning state (depth keys, dirty
 * flags, fragment-bin cursors, antialias-backward vertex accumulator) that every completed forward+backward pair leaves clean again. */
int fmhr_ham_workspace_init(const fmhr_ham_dims* d, void* ws, size_t ws_bytes, void* stream);
int fmhr_ham_forward(const fmhr_ham_dims* d, const float* vertices_tmp, const float* delta, const float* albedo,
                     const float* sh, const float* w2c, const float* proj, const float* mask, const float* valid_mask,
                     const int32_t* view_idx, const int32_t* tri, const int32_t* opp, const int32_t* vf_ptr, const int32_t* vf_idx,
                     int valid_mask_changed, int32_t* n_valid_local, void* ws, size_t ws_bytes, void* stream);
int fmhr_ham_backward(const fmhr_ham_dims* d, const fmhr_ham_weights* w, const float* vertices_tmp, const float* delta,
                      const float* albedo, const float* sh, const float* w2c, const float* proj, const float* img,
                      const float* mask, const float* valid_mask, const int32_t* view_idx, const int32_t* tri, const int32_t* opp,
                      const int32_t* row_ptr, const int32_t* col_idx, const int32_t* col_mult, const int32_t* vf_ptr,
                      const int32_t* vf_idx, const int32_t* corner_slot, const float* n_valid_global,
                      float* losses, float* gdelta, float* galbedo, float* gdelta_rest, float* galbedo_rest,
                      const fmhr_adam_param* adam_delta, const fmhr_adam_param* adam_albedo,   /* one-rank form only; may be NULL */
                      void* ws, size_t ws_bytes, void* stream);
/* After the all-reduce of a view-sharded step's bucket [P_delta | P_albedo | R_delta | R_albedo | 8 loss slots] (12V+8 floats):
 * g = P / (3 * n_valid_global) + R written over P (the first 6V floats are then gdelta | galbedo), the photometric loss slot gets the
 * same divisor, the total slot is formed; n_valid_global = the summed FMHR_LOSS_NVALID slot (0: the photometric term is 0). */
int fmhr_ham_combine(int V, float* bucket, const fmhr_adam_param* adam_delta, const fmhr_adam_param* adam_albedo, void* stream);
/* Views of the step's intermediates inside the workspace (for parity tests / rendering): returns device pointers.  The per-pixel
 * ones have rows of fmhr_ham_pitch(d) pixels: element (b, y, x, c) of `rast` is at ((b*H + y)*pitch + x)*4 + c, x < W; the pad
 * columns hold zeros after fmhr_ham_densify. */
typedef struct fmhr_ham_views {
    float* pos;        /* [n,V,4] clip positions: NOT stored by the step (every kernel recomputes them from the vertices and the
                          views' combined matrices) — filled, bit-identical to what the kernels used, by fmhr_ham_densify */
    float* normals;    /* [V,3]     */
    float* rast;       /* [n,H,pitch,4] */
    float* shaded;     /* [n,H,pitch,4] rgb = shaded image before antialias (tmp_img, :285-286), a = coverage colour (:272); not kept by
                          the step: filled by fmhr_ham_densify */
    float* aa;         /* [n,H,pitch,4] the same after antialias: rgb = tmp_img (:287), a = pred_mask (:274) */
    int32_t* aa_counter; /* [4]: antialias items found (rearmed per step), antialias-list overflow of the LAST step, fragment-bin
                            overflow (sticky), antialias-list overflow in ANY step since fmhr_ham_workspace_init (sticky) */
} fmhr_ham_views;
int fmhr_ham_workspace_views(const fmhr_ham_dims* d, void* ws, size_t ws_bytes, fmhr_ham_views* out);
/* byte offset of a named internal buffer inside the workspace (tests / debugging), -1 if unknown */
long long fmhr_ham_workspace_offset(const fmhr_ham_dims* d, const char* name);
/* The step only writes shaded / aa inside "active" 64-pixel segments (covered pixels, their 4-neighbours, non-zero valid_mask)
 * and stores neither the dr.rasterize image, nor the clip positions, nor the shaded image before antialias; this fills the zeros,
 * rebuilds `rast` from the per-pixel words, computes `pos` and re-shades `shaded` (tri: the array the last forward was given; sh: ignored
 * since ABI 6 — the forward keeps the step's n SH rows in the workspace, may be NULL), so the views above can be inspected as dense arrays.
 * Call it after the step whose state you want. */
int fmhr_ham_densify(const fmhr_ham_dims* d, const int32_t* tri, const float* sh, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------ fused stage-1 step
 * The "albedo" iteration, mesh_sfs_optim.py:205-236 (vertices constant; parameters albedo [V,3] and sh_coeffs [n,9]; loss =
 * sfs_weight * L1 over the valid pixels of the batch).  With the geometry fixed, interpolate + antialias of the albedo is one fixed
 * sparse linear map per valid pixel; the PLAN holds it (and its per-view transpose) and is built ONCE per stage from the outputs of
 * the generic ops above, in four calls with two host reads of a count in between:
 *   1. fmhr_stage1_rows      rast / mask -> rowid [n*H*W] (row of a valid pixel = mask > 0 and covered, else -1), view_row0 [n+1]
 *                            (first row of each view, multiples of 256; [n] = total rows, padding included), view_valid [n]
 *   2. fmhr_stage1_row_ptr   + the antialias work items (fmhr_antialias_fwd's items / alphas of ANY colour on this rast) ->
 *                            row_ptr [rows+1]; row_ptr[rows] = number of entries
 *   3. fmhr_stage1_fill      -> entries (ent_col, ent_w) and the rows' constants: row_n [rows,3] = the antialiased, normalised normal
 *                            (:217-218, computed by the caller with the generic ops), row_img [rows,3] = the target colour
 *   4. fmhr_stage1_transpose -> t_ptr [V*n+1], t_row / t_w [entries]: per (vertex, view) the rows it contributes to, sorted by row
 * A step is then fmhr_stage1_step: two launches, gather-only (deterministic), the two Adam updates optionally fused in. */
size_t fmhr_stage1_rows_workspace_bytes(int n, int H, int W);
/* row_flags: scratch of row_flags_len >= n*H*W + n*255 ints, kept by the caller until step 3 */
int fmhr_stage1_rows(const float* rast, const float* mask, int n, int H, int W, int32_t* rowid, int32_t* view_row0,
                     int32_t* view_valid, int32_t* row_flags, long long row_flags_len, void* ws, size_t ws_bytes, void* stream);
size_t fmhr_stage1_row_ptr_workspace_bytes(int rows);
int fmhr_stage1_row_ptr(const float* rast, const int32_t* rowid, const int32_t* items, const float* alphas, int n_items, int n,
                        int H, int W, int rows, int32_t* row_flags, int32_t* row_ptr, void* ws, size_t ws_bytes, void* stream);
int fmhr_stage1_fill(const float* rast, const int32_t* tri, const int32_t* rowid, const int32_t* items, const float* alphas,
                     int n_items, const float* normals, const float* img, int n, int V, int F, int H, int W, int rows,
                     const int32_t* row_flags, const int32_t* row_ptr, int32_t* ent_col, float* ent_w, float* row_n,
                     float* row_img, void* stream);
size_t fmhr_stage1_transpose_workspace_bytes(int n, int V);
int fmhr_stage1_transpose(const int32_t* row_ptr, const int32_t* ent_col, const float* ent_w, const int32_t* view_row0, int n, int V,
                          int rows, int32_t* t_ptr, int32_t* t_row, float* t_w, void* ws, size_t ws_bytes, void* stream);
typedef struct fmhr_stage1_plan {
    int32_t n, V, rows, reserved;           /* views, vertices, rows (a multiple of 256) */
    const int32_t* view_row0; const int32_t* view_valid; const int32_t* row_ptr; const int32_t* ent_col; const float* ent_w;
    const float* row_n; const float* row_img; const int32_t* t_ptr; const int32_t* t_row; const float* t_w;
} fmhr_stage1_plan;
size_t fmhr_stage1_step_workspace_bytes(const fmhr_stage1_plan* plan);
/* One iteration on the views with view_sel[b] != 0 (view_sel == NULL: all n).  albedo [V,3], sh [n,9] (row b = view b).
 * Out: galbedo [V,3], gsh [n,9] (zero rows outside the batch), loss_view [n]: the views' shares of the loss (their sum is the
 * loss value of :233).  adam_albedo / adam_sh (may be NULL): the parameters are updated in place, every row of sh included (a
 * view outside the batch has a zero gradient, its moments still decay — torch.optim.Adam on the dense gradient). */
int fmhr_stage1_step(const fmhr_stage1_plan* plan, const float* albedo, const float* sh, const int32_t* view_sel, float sfs_weight,
                     float* galbedo, float* gsh, float* loss_view, const fmhr_adam_param* adam_albedo,
                     const fmhr_adam_param* adam_sh, void* ws, size_t ws_bytes, void* stream);

/* SH least-squares initialisation, mesh_sfs_optim.py:150-154,164-166,173-174: the reference fits 9 coefficients per view (and once
 * over all views) with np.linalg.lstsq on the host.  The normal equations need only sum phi phi^T and sum phi gray over the valid
 * pixels (mask > 0 and covered): accumulated here in float64 as per-workgroup partials, part [n, nblk, 55] (45 upper-triangle
 * entries row by row, 9 right-hand sides, the pixel count); the caller adds the nblk partials and solves the 9 x 9 systems.
 * normals [n,H,W,3] = the antialiased, normalised normals (:147-148), rast [n,H,W,4], mask / gray [n,H,W]. */
int fmhr_sh_normal_equations(const float* normals, const float* rast, const float* mask, const float* gray, int n, int H, int W,
                             int nblk, double* part, void* stream);
/* sums of img / radiance(sh, normal) per channel over the valid pixels + their count (albedo_mean, :173-174): part [n, nblk, 4] */
int fmhr_sh_albedo_sums(const float* normals, const float* rast, const float* mask, const float* img, const float* sh, int n,
                        int H, int W, int nblk, double* part, void* stream);

/* ------------------------------------------------------------------------------------------------ one-shot peer-to-peer all-reduce
 * the view-sharded step's ONE exchange (the 12V+8 bucket of fmhr_ham_backward / fmhr_ham_combine) as one launch per rank
 * over the direct xGMI links instead of a ring: every rank exports a block of uncached device memory (staging copy of its bucket, its
 * reduced shard, arrival flags) by hipIpcGetMemHandle, maps the peers' blocks, and fmhr_peer_all_reduce stages, reduces its shard from
 * all ranks' staging blocks (fixed rank order: bit-identical results on every rank) and gathers the peers' reduced shards — two flag
 * exchanges, every wait bounded (a dead peer ends in an error the host reads, not in a hung GPU).  One process per GPU; the handles (64
 * bytes each) travel over the caller's own channel (fmhr_amd/dist.py: one all_gather of torch.distributed).  Replaces ncclAllReduce of
 * the bucket where every rank could set it up (negotiated, RCCL is the fallback).
 *   fmhr_peer_create      allocate + initialise this rank's block for a bucket of n_floats floats (a multiple of 4); world <= 16
 *   fmhr_peer_export      this rank's 64-byte IPC handle (fmhr_peer_handle_bytes)
 *   fmhr_peer_import      map peer `peer`'s block from its handle (every peer but oneself, once)
 *   fmhr_peer_all_reduce  in-place SUM of `bucket` (this rank's device memory, 16-byte aligned) over the ranks, enqueued on `stream`;
 *                         every rank must call it the same number of times (the flags carry the call count)
 *   fmhr_peer_staging     the staging area of this rank's block as a bucket: a caller that lets fmhr_ham_backward write its bucket THERE
 *                         (uncached device memory, n_floats floats) and passes that pointer to fmhr_peer_all_reduce saves the staging
 *                         copy and one grid barrier of every exchange; any other 16-byte aligned bucket is staged by the launch
 *   fmhr_peer_status      *gave_up = 1 if a wait of any call so far ran into its bound (synchronises; the state is unusable afterwards)
 *   fmhr_peer_destroy     unmap, free */
size_t fmhr_peer_handle_bytes(void);
int fmhr_peer_create(int rank, int world, long long n_floats, void** state_out);
int fmhr_peer_export(void* state, void* handle_out);
int fmhr_peer_import(void* state, int peer, const void* handle);
int fmhr_peer_all_reduce(void* state, float* bucket, long long n_floats, void* stream);
int fmhr_peer_staging(void* state, void** ptr_out, long long* n_floats_out);
int fmhr_peer_status(void* state, int* gave_up);
int fmhr_peer_destroy(void* state);

/* ------------------------------------------------------------------------------------------------ per-kernel timing (optional)
 * Attach an array of 2*FMHR_SLOT_COUNT hipEvent_t (created by the caller with timing enabled) to the CALLING THREAD: while
 * attached, fmhr_ham_forward/backward record events[2*slot] / events[2*slot+1] around launch group `slot` on the step's
 * stream (asynchronously: nothing synchronises).  only_slot >= 0 restricts recording to that group; -1 records all.
 * The caller reads hipEventElapsedTime after synchronising.  Thread-local: other threads/streams are unaffected. */
#define FMHR_SLOT_GEOMETRY 0        /* vertices = tmp+delta, vertex normals, clip transform */
#define FMHR_SLOT_RASTER_CLEAR 1
#define FMHR_SLOT_RASTER_TRIS 2
#define FMHR_SLOT_RASTER_LARGE 3    /* queued large-box triangles (coarse meshes only; empty when they are swept inside RASTER_TRIS) */
#define FMHR_SLOT_RASTER_RESOLVE 4
#define FMHR_SLOT_SHADE_FWD 5
#define FMHR_SLOT_AA_FWD 6          /* the two image copies + the antialias kernel */
#define FMHR_SLOT_LOSS_PIX 7        /* + the gradient buffer clears */
#define FMHR_SLOT_AA_BWD 8
#define FMHR_SLOT_SHADE_BWD 9
#define FMHR_SLOT_TRANSFORM_BWD 10
#define FMHR_SLOT_NORMALS_BWD 11
#define FMHR_SLOT_REGULARISERS 12
#define FMHR_SLOT_COUNT 13
int fmhr_timing_attach(void** events, int n_events, int only_slot);
int fmhr_timing_detach(void);

#ifdef __cplusplus
}
#endif
#endif /* FMHR_HAM_H */
