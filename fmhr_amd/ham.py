"""fmhr_amd.ham — the fused stage-2 HAM iteration (render + six losses + backward) and the optimisation loop.

Replaces the loop body of /root/reference/mesh_sfs_optim.py:253-310.  One `HamStage2.step()` issues two C-ABI calls
(fmhr_ham_forward / fmhr_ham_backward, include/fmhr_ham.h) on the current stream.  With the views sharded over ranks (SURVEY.md
§8e) the step has exactly ONE collective: the backward keeps the photometric share of the gradient apart with the batch-global
L1 divisor left out, the bucket [P | R | losses] is all-reduced once and fmhr_ham_combine applies P / (3 * n_valid_global) + R.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import HamDims, HamViews, HamWeights, call, ptr, stream_ptr

LOSS_NAMES = ("sfs", "lap", "albedo", "mask", "edge", "delta", "total", "n_valid")
BIN_CAP_DEFAULT = 512                  # fmhr_amd/csrc/shade_common.h: kBinCapDefault


class WorkListOverflow(RuntimeError):
    """a work list of the step was full in some step (the steps themselves never synchronise: the sticky flags are read where the caller
    synchronises anyway).  which = "bins" (fragment bins of the rasterizer: `needed` = the fullest segment's depth — run_ham rebuilds the
    workspaces with room for it and repeats stage 2) or "aa" (the antialias work-item list)."""

    def __init__(self, msg, which, needed=None):
        super().__init__(msg)
        self.which, self.needed = which, needed


def edge_length_mean(vertices, faces):
    """mean squared length of the 3F directed face edges of the initial mesh (mesh_sfs_optim.py:184-188); one-off."""
    fl = faces.long()
    a, b, c = vertices[fl[:, 0]], vertices[fl[:, 1]], vertices[fl[:, 2]]
    return float(torch.cat([((a - b) ** 2).sum(1), ((c - b) ** 2).sum(1), ((a - c) ** 2).sum(1)]).mean())


class VertexOrder:
    """A renumbering of the mesh's VERTICES for the step's gathers — the faces keep their order (the rasterizer breaks depth ties
    by face index, and `rast` carries face indices), every face keeps its corner order.

    The step's kernels gather per-vertex data through the face table: the snapped vertices of 64 consecutive faces in the triangle
    kernel, positions and attributes of the faces under 64 consecutive pixels in the second rasterizer pass, CSR neighbours in the mesh
    tails.  A Loop-subdivided mesh (mesh_sfs_optim.py:82,106) appends each level's edge midpoints behind all older vertices, so the
    three corners of a face sit in three distant parts of every per-vertex array.  Numbering the vertices in the order the face list
    first touches them puts the corners of neighbouring faces in neighbouring rows: 17.6 -> 7.3 cache lines per 64 faces of the 8-byte
    snapped-vertex array at the metric mesh, 2 880 -> 2 930 it/s; a random numbering: 2 530.

    The caller applies it ONCE to what it holds per vertex (`to_internal`: template vertices, initial parameters), builds HamStage2 /
    HamStage1 on `order.faces`, optimises in that numbering and maps the results back (`to_caller`) — `fmhr_amd.pipeline.run_ham` and
    `bench.py` do exactly this.  Nothing but floating-point summation order depends on the numbering (neighbour sums of the
    Laplacians); `rast` is bit-identical."""

    def __init__(self, faces, V, enabled=None):
        if enabled is None:
            enabled = os.environ.get("FMHR_VERTEX_ORDER", "1") != "0"
        self.V = int(V)
        self.device = dev = faces.device
        # (on the faces' own device: as NumPy on the host — unique + argsort over 3 F indices and three transfers — this was 20-30 ms
        # of a 260-ms demo run; tools/experiments/r05_run_ham_cold.py)
        flat = faces.detach().reshape(-1).to(torch.int64)
        if flat.numel() and (int(flat.min()) < 0 or int(flat.max()) >= self.V):
            raise RuntimeError("VertexOrder: face table refers to a vertex outside [0, V)")
        ar = torch.arange(self.V, dtype=torch.int64, device=dev)
        if enabled and flat.numel():
            # where the face list first touches each vertex; a vertex in no face keeps 3 F and sorts behind the others, in its own order
            first = torch.full((self.V,), flat.numel(), dtype=torch.int64, device=dev)
            first.scatter_reduce_(0, flat, torch.arange(flat.numel(), dtype=torch.int64, device=dev), reduce="amin", include_self=True)
            old_of_new = torch.argsort(first, stable=True)
        else:
            old_of_new = ar
        new_of_old = torch.empty_like(old_of_new)
        new_of_old[old_of_new] = ar
        self.identity = bool((old_of_new == ar).all())
        self.old_of_new, self.new_of_old = old_of_new, new_of_old
        self.faces = (faces if self.identity else new_of_old[faces.detach().to(torch.int64)].to(faces.dtype)).contiguous()

    def _index(self, x, idx, dim):
        if self.identity:
            return x
        if x.shape[dim] != self.V:
            raise RuntimeError(f"VertexOrder: dimension {dim} has {x.shape[dim]} entries, the mesh {self.V} vertices")
        return x.index_select(dim, idx.to(x.device)).contiguous()

    def to_internal(self, x, dim=0):
        """per-vertex data in the caller's numbering -> the step's (a copy; x itself when the order is the identity)"""
        return self._index(x, self.old_of_new, dim)

    def to_caller(self, x, dim=0):
        """per-vertex data in the step's numbering -> the caller's"""
        return self._index(x, self.new_of_old, dim)


class HamStage2:
    """Owns the per-mesh tables (edge/opposite-vertex table, adjacency CSR) and the step workspace for `n` local views.

    faces: int32 [F,3] on the GPU.  `n_global` = number of views in the whole batch (== n unless views are sharded).
    `n_src` (ABI 6): the object steps MINIBATCHES of n views out of a capture of n_src — the per-view arguments of step() / forward() /
    backward() (sh, w2c, proj, img, mask, valid_mask) are then the capture's resident [n_src, ...] tensors and `view_idx` (int32 [n] on
    the GPU, e.g. perm[k:k+batch] of mesh_sfs_optim.py:249,255-260) says which views the step renders; the kernels index through it,
    nothing is gathered, and the per-segment valid_mask table is built once for all n_src views."""

    def __init__(self, faces, V, n, H, W, weights, edge_len_mean, n_global=None, aa_cap=None, bin_cap=0, pitch=None, n_src=None):
        if not faces.is_cuda or faces.dtype != torch.int32:
            raise RuntimeError("HamStage2: faces must be an int32 GPU tensor")
        self.lib = _lib.load()
        self.dev = faces.device
        self.faces = faces.contiguous()
        self.V, self.F, self.n, self.H, self.W = int(V), int(faces.shape[0]), int(n), int(H), int(W)
        self.n_global = int(n_global or n)
        self.weights = dict(weights)
        self.edge_len_mean = float(edge_len_mean)
        st = stream_ptr(self.dev)
        # per-mesh tables, built once (the reference rebuilds both twice per step)
        self.opp = torch.full((self.F, 3), -1, dtype=torch.int32, device=self.dev)
        nb = self.lib.fmhr_topology_workspace_bytes(self.F)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=self.dev)
        call("fmhr_topology_build", ptr(self.faces), self.F, self.V, ptr(self.opp), ptr(ws), nb, st)
        self.row_ptr = torch.zeros(self.V + 1, dtype=torch.int32, device=self.dev)
        self.col_idx = torch.zeros(6 * self.F, dtype=torch.int32, device=self.dev)
        nb = self.lib.fmhr_adjacency_workspace_bytes(self.V, self.F)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=self.dev)
        call("fmhr_adjacency_build", ptr(self.faces), self.F, self.V, ptr(self.row_ptr), ptr(self.col_idx), ptr(ws), nb, st)
        self.nnz = int(self.row_ptr[-1])
        # faces per adjacency entry: the edge-length hinge (a mean over FACE edges) rides in the Laplacians' neighbour walk
        self.col_mult = torch.zeros(max(self.nnz, 1), dtype=torch.int32, device=self.dev)
        call("fmhr_edge_multiplicity_build", ptr(self.faces), self.F, self.V, ptr(self.row_ptr), ptr(self.col_idx), self.nnz,
             ptr(self.col_mult), st)
        self.vf_ptr = torch.zeros(self.V + 1, dtype=torch.int32, device=self.dev)
        self.vf_idx = torch.zeros(3 * self.F, dtype=torch.int32, device=self.dev)
        nb = self.lib.fmhr_vertex_faces_workspace_bytes(self.V)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=self.dev)
        call("fmhr_vertex_faces_build", ptr(self.faces), self.F, self.V, ptr(self.vf_ptr), ptr(self.vf_idx), ptr(ws), nb, st)
        self.corner_slot = torch.zeros(3 * self.F, dtype=torch.int32, device=self.dev)
        call("fmhr_corner_slots_build", ptr(self.vf_idx), self.F, ptr(self.corner_slot), st)
        # antialias work items: only pairs that blend are listed (∝ silhouette length); 1/4 of the pixels is a generous bound
        cap = int(aa_cap or max(2048 * 256, self.n * self.H * self.W // 4))     # 768 or 1280 regions (one per workgroup of the AA kernels) of >= 409 items
        # bin_cap: entries per fragment bin of the binned rasterizer (0 = the default 512 = 8 fragments per pixel of a 64-pixel segment; run_ham grows it on overflow)
        # pitch (ABI 5): row pitch of the step's own per-pixel buffers; None / 0 = W rounded up to a multiple of 64 (every 64-pixel
        # segment in one row: binned rasterizer + row-aligned antialias kernels for any W); FMHR_ROW_PITCH=0 in the environment
        # selects the ABI-4 layout (pitch = W: flat segments) for A/B runs.  The caller's img / mask / valid_mask stay [n,H,W].
        if pitch is None:
            pitch = self.W if os.environ.get("FMHR_ROW_PITCH", "1") == "0" else 0
        if not bin_cap and os.environ.get("FMHR_BIN_CAP"):             # experiments: the default capacity of every workspace
            bin_cap = int(os.environ["FMHR_BIN_CAP"])
        self.n_src = int(n_src or 0)
        self.dims = HamDims(self.n, self.V, self.F, self.H, self.W, self.nnz, cap, int(bin_cap), int(pitch), self.n_src)
        self.pitch = self.lib.fmhr_ham_pitch(ctypes.byref(self.dims))
        if self.pitch < 0:
            raise RuntimeError("fmhr_ham_pitch failed: " + self.lib.fmhr_last_error().decode())
        self.ws_bytes = self.lib.fmhr_ham_workspace_bytes(ctypes.byref(self.dims))
        if self.ws_bytes == 0:
            raise RuntimeError("fmhr_ham_workspace_bytes failed: " + self.lib.fmhr_last_error().decode())
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.dev)
        call("fmhr_ham_workspace_init", ctypes.byref(self.dims), ptr(self.ws), self.ws_bytes, st)
        self.n_valid_local = torch.zeros(1, dtype=torch.int32, device=self.dev)
        # fmhr_ham_forward also leaves float(n_valid_local) in the workspace: it IS n_valid_global on one rank, and the
        # all-reduce target (in place) when the views are sharded
        self.n_valid_global = self.raw("n_valid_f", torch.float32, 1)
        # one contiguous bucket [gdelta | galbedo | gdelta_rest | galbedo_rest | losses]: one rank uses the first two and the loss
        # slots; a view-sharded step fills all of it (photometric share | the rest), all-reduces it in place ONCE and combines
        self._bind_bucket(torch.zeros(12 * self.V + 8, dtype=torch.float32, device=self.dev))
        # which rasterizer fmhr_ham_forward takes for these dims: per-segment fragment bins resolved in LDS where the mesh is a
        # micro-triangle one and the pitch a multiple of 64 (FMHR_RASTER_BINNED=0 switches them off), else global depth keys + resolve pass
        self.binned = self.lib.fmhr_ham_rasterizer_form(ctypes.byref(self.dims)) == 1
        self._views = None
        self._aa_flag = None
        self._vm_key, self._vm_ref = None, None
        self._last_sh = None

    def _bind_bucket(self, bucket):
        """the step's output buffer [gdelta | galbedo | gdelta_rest | galbedo_rest | 8 loss slots]; every step writes all of it"""
        V3 = 3 * self.V
        self.bucket = bucket
        self.gdelta = bucket[:V3].view(self.V, 3)
        self.galbedo = bucket[V3: 2 * V3].view(self.V, 3)
        self.gdelta_rest = bucket[2 * V3: 3 * V3].view(self.V, 3)
        self.galbedo_rest = bucket[3 * V3: 4 * V3].view(self.V, 3)
        self.losses = bucket[4 * V3:]

    # -- the two phases -------------------------------------------------------------------------------------------------
    def _view_idx(self, view_idx):
        if (view_idx is not None) != (self.n_src > 0):
            raise RuntimeError("HamStage2: view_idx goes with n_src (a minibatch of a resident capture): construct with n_src=<views of the "
                               "capture> and pass view_idx to every step, or neither")
        if view_idx is None:
            return None
        if view_idx.dtype != torch.int32 or not view_idx.is_cuda or tuple(view_idx.shape) != (self.n,) or not view_idx.is_contiguous():
            raise RuntimeError(f"HamStage2: view_idx must be a contiguous int32 GPU tensor of shape ({self.n},), got {tuple(view_idx.shape)} "
                               f"{view_idx.dtype} {view_idx.device}")
        return view_idx

    def forward(self, vertices_tmp, delta, albedo, sh, w2c, proj, mask, valid_mask, view_idx=None):
        view_idx = self._view_idx(view_idx)
        key = (valid_mask.data_ptr(), valid_mask._version)
        changed = 1 if key != self._vm_key else 0            # the tensor is kept alive below, so its address is not reused
        self._vm_key, self._vm_ref = key, valid_mask
        self._last_sh = sh                                   # views() re-shades the dense `shaded` image with it
        call("fmhr_ham_forward", ctypes.byref(self.dims), ptr(vertices_tmp), ptr(delta), ptr(albedo), ptr(sh), ptr(w2c),
             ptr(proj), ptr(mask), ptr(valid_mask), ptr(view_idx), ptr(self.faces), ptr(self.opp), ptr(self.vf_ptr), ptr(self.vf_idx),
             changed, ptr(self.n_valid_local), ptr(self.ws), self.ws_bytes, stream_ptr(self.dev))

    def backward(self, vertices_tmp, delta, albedo, sh, w2c, proj, img, mask, valid_mask, reg_grads=True, reg_range=None, split=False,
                 adam=(None, None), view_idx=None):
        """split=True: the view-sharded form — gdelta / galbedo get the photometric share without the L1 divisor, gdelta_rest /
        galbedo_rest the rest (include/fmhr_ham.h); the caller all-reduces self.bucket and calls combine()."""
        w = self.weights
        view_idx = self._view_idx(view_idx)
        v0, v1 = (0, self.V) if reg_range is None else reg_range
        wt = HamWeights(w["sfs"], w["lap"], w["albedo"], w["mask"], w["edge"], w["delta"], self.edge_len_mean,
                        float(self.n_global * self.H * self.W), 1.0 if reg_grads else 0.0, int(v0), int(v1))
        call("fmhr_ham_backward", ctypes.byref(self.dims), ctypes.byref(wt), ptr(vertices_tmp), ptr(delta), ptr(albedo),
             ptr(sh), ptr(w2c), ptr(proj), ptr(img), ptr(mask), ptr(valid_mask), ptr(view_idx), ptr(self.faces), ptr(self.opp),
             ptr(self.row_ptr), ptr(self.col_idx), ptr(self.col_mult), ptr(self.vf_ptr), ptr(self.vf_idx), ptr(self.corner_slot),
             ptr(self.n_valid_global),
             ptr(self.losses), ptr(self.gdelta),
             ptr(self.galbedo), ptr(self.gdelta_rest if split else None), ptr(self.galbedo_rest if split else None),
             ctypes.byref(adam[0]) if adam[0] is not None else None, ctypes.byref(adam[1]) if adam[1] is not None else None,
             ptr(self.ws), self.ws_bytes, stream_ptr(self.dev))

    def combine(self, adam=(None, None)):
        """after the all-reduce of self.bucket (view-sharded step): g = P / (3 n_valid_global) + R in place, loss vector finished."""
        call("fmhr_ham_combine", self.V, ptr(self.bucket), ctypes.byref(adam[0]) if adam[0] is not None else None,
             ctypes.byref(adam[1]) if adam[1] is not None else None, stream_ptr(self.dev))

    def step(self, vertices_tmp, delta, albedo, sh, w2c, proj, img, mask, valid_mask, group=None, reg_grads=True, reg_range=None,
             optimizer=None, params=None, view_idx=None):
        """Full forward + backward.  Returns (losses[8] device tensor, gdelta [V,3], galbedo [V,3]) — buffers owned by this
        object and overwritten by the next step.  `group`: torch.distributed process group when views are sharded: the results
        are then the batch-global ones (summed over the ranks), identical on every rank.
        `group` may also be a fmhr_amd.dist.StreamAllReduce (RCCL on this step's own stream: no cross-stream hand-over).
        `optimizer`, `params`: apply the Adam update of (delta, albedo) inside the step (no separate optimizer.step() launch).
        `reg_range` = (v0, v1): evaluate the view-independent terms for these vertices only (ranks take disjoint ranges covering
        [0, V): fmhr_amd.dist.shard_views(V, rank, world)); `reg_grads=False` switches them off on this rank.
        `view_idx` (object built with n_src): int32 [n] GPU tensor — the step renders views view_idx[b] of the capture-resident
        sh / w2c / proj / img / mask / valid_mask [n_src, ...] (the loop's `perm[k:k+batch]`, mesh_sfs_optim.py:255-260, without the gathers)."""
        self._check(vertices_tmp, delta, albedo, sh, w2c, proj, img, mask, valid_mask)
        adam = (None, None)
        if optimizer is not None:
            # fused optimiser step: `params` = the (delta, albedo) parameter tensors `optimizer` (fmhr_amd.optim.Adam) owns, whose
            # storage the `delta` / `albedo` arguments view; they are updated in place by the kernel that forms their final gradient
            p_delta, p_albedo = params
            if p_delta.data_ptr() != delta.data_ptr() or p_albedo.data_ptr() != albedo.data_ptr():
                raise RuntimeError("HamStage2.step: `params` must be the tensors behind the `delta` / `albedo` arguments")
            adam = (optimizer.fused_param(p_delta), optimizer.fused_param(p_albedo))
        if group is not None and hasattr(group, "staging") and getattr(self, "_staged_for", None) is not group:
            # a one-shot peer-to-peer exchange (fmhr_amd.dist.PeerAllReduce): the backward writes the bucket straight into the block the
            # peers read — no staging copy inside the exchange.  One owner per exchange object; anyone else's bucket is staged by the launch
            self._staged_for = group
            st = group.staging(self) if os.environ.get("FMHR_PEER_IN_PLACE", "1") != "0" else None
            if st is not None and st.numel() == self.bucket.numel():
                self._bind_bucket(st)
        self.forward(vertices_tmp, delta, albedo, sh, w2c, proj, mask, valid_mask, view_idx=view_idx)
        self.backward(vertices_tmp, delta, albedo, sh, w2c, proj, img, mask, valid_mask, reg_grads=reg_grads, reg_range=reg_range,
                      split=group is not None, adam=adam if group is None else (None, None), view_idx=view_idx)
        if group is not None:
            # the step's ONE collective (1.18 -> 2.35 MB at V = 49k: still latency-bound on xGMI), then P / (3 Pv) + R
            if hasattr(group, "all_reduce_"):
                group.all_reduce_(self.bucket, stream_ptr(self.dev))
            else:
                torch.distributed.all_reduce(self.bucket, group=group)
            self.combine(adam)
        return self.losses, self.gdelta, self.galbedo

    def _check(self, vertices_tmp, delta, albedo, sh, w2c, proj, img, mask, valid_mask):
        n, V, H, W = self.n_src or self.n, self.V, self.H, self.W        # (n_src: the per-view arguments are the capture's)
        exp = dict(vertices_tmp=(V, 3), delta=(V, 3), albedo=(V, 3), sh=(n, 9), w2c=(n, 4, 4), proj=(n, 4, 4),
                   img=(n, H, W, 3), mask=(n, H, W), valid_mask=(n, H, W))
        got = dict(vertices_tmp=vertices_tmp, delta=delta, albedo=albedo, sh=sh, w2c=w2c, proj=proj, img=img, mask=mask,
                   valid_mask=valid_mask)
        for k, shp in exp.items():
            t = got[k]
            if tuple(t.shape) != shp or t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                raise RuntimeError(f"HamStage2.step: {k} must be a contiguous float32 GPU tensor of shape {shp}, "
                                   f"got {tuple(t.shape)} {t.dtype} {t.device}")

    # -- intermediates (views into the workspace; valid until the next step) ----------------------------------------------
    def views(self):
        """dense views of the last step's intermediates (zero-fills what the sparse step did not write)."""
        if self._last_sh is None:
            raise RuntimeError("HamStage2.views: no forward has run on this object yet")
        call("fmhr_ham_densify", ctypes.byref(self.dims), ptr(self.faces), ptr(self._last_sh), ptr(self.ws), self.ws_bytes,
             stream_ptr(self.dev))
        if self._views is None:
            hv = HamViews()
            call("fmhr_ham_workspace_views", ctypes.byref(self.dims), ptr(self.ws), self.ws_bytes, ctypes.byref(hv))
            base = self.ws.data_ptr()
            n, V, H, W, Wp = self.n, self.V, self.H, self.W, self.pitch

            def view(p, shape, dtype=torch.float32):
                numel = 1
                for s in shape:
                    numel *= s
                off = p - base
                return self.ws[off: off + numel * 4].view(dtype).view(*shape)

            # the per-pixel views have rows of `pitch` pixels inside the workspace: sliced to the image's W columns
            img = lambda p: view(p, (n, H, Wp, 4))[:, :, :W]
            shaded, aa = img(hv.shaded), img(hv.aa)
            self._views = dict(pos=view(hv.pos, (n, V, 4)), normals=view(hv.normals, (V, 3)), rast=img(hv.rast),
                               tmp_img_pre_aa=shaded[..., :3], tmp_img=aa[..., :3], pred_mask=aa[..., 3],
                               aa_counter=view(hv.aa_counter, (4,), torch.int32))
        return self._views

    def raw(self, name, dtype, count):
        """debug: `count` elements of a named internal workspace buffer as a tensor view."""
        off = self.lib.fmhr_ham_workspace_offset(ctypes.byref(self.dims), name.encode())
        if off < 0:
            raise RuntimeError(self.lib.fmhr_last_error().decode())
        nbytes = count * torch.empty((), dtype=dtype).element_size()
        return self.ws[off: off + nbytes].view(dtype)

    def aa_item_count(self):
        """number of antialias work items (silhouette pixel pairs that blend) of the last step: a device tensor, no synchronisation.
        It grows while the optimisation roughens the contour (2.4 k -> ~55 k over the first ~50 steps of the metric workload), and
        the antialias kernels' time with it: benchmarks pre-roll until it is stationary."""
        if getattr(self, "_aa_regions", None) is None:
            off = self.lib.fmhr_ham_workspace_offset(ctypes.byref(self.dims), b"aa_region_cnt")
            self._aa_regions = self.ws[off: off + 2048 * 4].view(torch.int32)
        return self._aa_regions.sum()

    def _flags(self):
        if self._aa_flag is None:
            off = self.lib.fmhr_ham_workspace_offset(ctypes.byref(self.dims), b"aa_counter")
            self._aa_flag_all = self.ws[off: off + 16].view(torch.int32)
            self._aa_flag = self._aa_flag_all[:2]
        return self._aa_flag_all

    def bin_depth_max(self):
        """The most fragments ONE 64-pixel segment has held in any step since the workspace was initialised (binned rasterizer; 0 with the
        atomic-key form) — what `bin_cap` (default 512 = 4 KB per segment) has to cover.  Synchronises (one 4-byte copy): call it where the
        run synchronises anyway; run_ham reports it at the end."""
        return int(self.raw("stamp", torch.int32, 4)[3])

    def aa_overflowed(self):
        """True if the antialias work-item list was full in ANY step since the workspace was initialised (sticky, like the
        fragment-bin flag; one device -> host copy): those steps dropped silhouette pairs from the gradient."""
        return int(self._flags()[3]) != 0

    def aa_overflowed_last_step(self):
        """the same for the last step only (the kernel's per-step word, rearmed by every forward)."""
        return int(self._flags()[1]) != 0

    def overflow_flags(self):
        """(antialias list, fragment bins) sticky overflow flags with ONE device -> host copy: what an optimisation loop checks once
        per epoch and at its end (fmhr_amd.pipeline.run_ham)."""
        f = self._flags().tolist()
        return f[3] != 0, f[2] != 0

    def check_overflow(self):
        """raise if a work list of the step overflowed in any step so far (the step itself never synchronises)."""
        aa, bins = self.overflow_flags()
        if self.n_src and int(self.raw("stamp", torch.int32, 4)[1]) != 0:
            raise RuntimeError(f"HamStage2: a step was given a view index outside [0, {self.n_src}) (view_idx selects views of the resident capture): "
                               "the kernels clamped it — that step rendered the wrong view")
        if bins:
            need = self.bin_depth_max()
            raise WorkListOverflow(f"HamStage2: a fragment bin of the rasterizer overflowed (more than {self.dims.bin_cap or BIN_CAP_DEFAULT} fragments "
                                   f"in one 64-pixel segment; the fullest segment so far held {need}): fragments were dropped; "
                                   "construct HamStage2 with a larger bin_cap", "bins", need)
        if aa:
            raise WorkListOverflow(f"HamStage2: the antialias work-item list overflowed (aa_cap={self.dims.aa_cap}): some silhouette pairs "
                                   "got no gradient in at least one step; construct HamStage2 with a larger aa_cap", "aa")

    def raster_overflowed(self):
        """True if a fragment bin of the binned rasterizer was full in ANY step since the workspace was initialised (sticky; one device
        -> host copy): some fragments were dropped — construct HamStage2 with a larger `bin_cap` (entries per 64-pixel segment,
        default 512 = 8 fragments per pixel)."""
        return int(self._flags()[2]) != 0

    def loss_dict(self):
        """The loss vector of the last step as a dict (one device -> host copy).  Also the place where a full antialias work-item
        list is reported: the step never synchronises on its own, so the flag is checked when the caller reads results anyway."""
        vals = self.losses.tolist()
        self.check_overflow()
        return dict(zip(LOSS_NAMES, vals))


class _HamStepFn(torch.autograd.Function):
    """The whole stage-2 loop body (mesh_sfs_optim.py:253-306) as ONE differentiable op: forward runs fmhr_ham_forward +
    fmhr_ham_backward (the fused step always forms the gradients: the loss vector is a by-product of the backward's reductions),
    backward hands the stored gradients to autograd.  `loss.backward()` then fills `delta.grad` / `albedo.grad` exactly as the
    reference's `loss.backward()` (:309) does, and `optimizer.step()` (:310) stays the caller's."""

    @staticmethod
    def forward(ctx, hs, vertices_tmp, delta, albedo, sh, w2c, proj, img, mask, valid_mask, group, reg_range):
        alb = albedo.detach()
        ctx.albedo_shape = tuple(albedo.shape)
        if alb.dim() == 3:                               # the reference keeps albedo as [1,V,3] (:176)
            if alb.shape[0] != 1:
                raise RuntimeError("ham_step: albedo must be [V,3] or [1,V,3]")
            alb = alb[0]
        losses, gdelta, galbedo = hs.step(vertices_tmp, delta.detach(), alb, sh, w2c, proj, img, mask, valid_mask, group=group,
                                          reg_range=reg_range)
        # the step's buffers are overwritten by the next step: autograd gets its own copies (2 x 0.6 MB at V = 49k)
        ctx.save_for_backward(gdelta.clone(), galbedo.clone())
        out = losses.clone()
        ctx.mark_non_differentiable(out)
        return out[6].clone(), out

    @staticmethod
    def backward(ctx, g_total, _g_losses):
        gdelta, galbedo = ctx.saved_tensors
        return (None, None, gdelta * g_total, (galbedo * g_total).view(ctx.albedo_shape), None, None, None, None, None, None,
                None, None)


def ham_step(hs, vertices_tmp, delta, albedo, sh, w2c, proj, img, mask, valid_mask, group=None, reg_range=None):
    """-> (loss, losses[8]): `loss` is the scalar of mesh_sfs_optim.py:306, differentiable w.r.t. `delta` [V,3] and `albedo`
    ([V,3] or the reference's [1,V,3]); `losses` = [sfs, lap, albedo, mask, edge, delta, total, n_valid] (detached; read it every
    k steps instead of six .item() per step, :312-314).  `hs`: the HamStage2 that owns the mesh tables and the workspace."""
    return _HamStepFn.apply(hs, vertices_tmp, delta, albedo, sh, w2c, proj, img, mask, valid_mask, group, reg_range)


def render_forward(faces, vertices, albedo, sh, w2c, proj, mask, H, W):
    """Stage-2 forward only (for target generation / re-rendering): returns dict(tmp_img, pred_mask, rast) clones.  The sticky overflow
    flags are read before the result is handed out (this call synchronises anyway): a fragment bin deeper than the default capacity means
    a workspace with room for it and a second render — never an image with dropped fragments."""
    n, V = w2c.shape[0], vertices.shape[0]
    cap = 0
    for _ in range(4):
        hs = HamStage2(faces, V, n, H, W, dict(sfs=0, lap=0, albedo=0, mask=0, edge=0, delta=0), 0.0, bin_cap=cap)
        zero = torch.zeros_like(vertices)
        hs.forward(vertices.contiguous(), zero, albedo.contiguous(), sh.contiguous(), w2c.contiguous(), proj.contiguous(),
                   mask.contiguous(), torch.zeros_like(mask))
        try:
            hs.check_overflow()
        except WorkListOverflow as e:
            if e.which != "bins":
                raise
            cap = max(2 * (cap or BIN_CAP_DEFAULT), (int(1.25 * e.needed) + 63) // 64 * 64)
            continue
        v = hs.views()
        return dict(tmp_img=v["tmp_img"].clone(), pred_mask=v["pred_mask"].clone(), rast=v["rast"].clone())
    raise RuntimeError("render_forward: the fragment bins overflowed at every capacity tried")
