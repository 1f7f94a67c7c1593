"""fmhr_amd.pipeline — the whole HAM optimisation of /root/reference/mesh_sfs_optim.py:30-343 on the MI355X library
(SURVEY.md §8f rows N3 / N4): SH least-squares initialisation (:124-174), stage 1 "albedo" (:193-240, the fused HamStage1 step;
`stage1_loss` is its op-by-op form on fmhr_amd.dr / fmhr_amd.utils with autograd), stage 2 "sfs" (:242-317, the fused HamStage2
step), export (:319-343).

`init_sh` and `stage1_loss` take the operator module as an argument (anything with the nvdiffrast surface + the models/utils.py
helpers), so the parity tests can run the very same code on the CPU checker; the product passes fmhr_amd.dr / fmhr_amd.utils.
"""
import math
import os
import types

import numpy as np
import torch
import torch.nn.functional as F


def product_ops():
    from . import dr, utils
    return types.SimpleNamespace(RasterizeGLContext=dr.RasterizeGLContext, rasterize=dr.rasterize, interpolate=dr.interpolate,
                                 antialias=dr.antialias, get_normals=utils.get_normals, get_radiance=utils.get_radiance,
                                 get_matrix=utils.get_matrix)


def clip_positions(vertices, w2c, proj):
    """mesh_sfs_optim.py:137-140 / :207-209 / :262-264 (row-vector convention)."""
    n = w2c.shape[0]
    vertsw = torch.cat([vertices, torch.ones_like(vertices[:, 0:1])], dim=1).unsqueeze(0).expand(n, -1, -1)
    rot = torch.einsum('ijk,ikl->ijl', vertsw, w2c)
    return vertsw, torch.einsum('ijk,ikl->ijl', rot, proj).contiguous()


def init_sh(ops, vertices, faces, imgs, grayimgs, masks, w2cs, projs, resolution, degree=3):
    """Per-view and global SH least squares + mean albedo + the antialiased start silhouettes (mesh_sfs_optim.py:124-174).
    Returns dict(sh_coeffs [num,9], sh_global [9], albedo_mean [3], valid_masks [num,H,W])."""
    glctx = ops.RasterizeGLContext()
    num = imgs.shape[0]
    v_normals, v_gray, v_imgs, v_masks, sh_coeffs = [], [], [], [], []
    with torch.no_grad():
        for k in range(num):
            w2c, proj, mask, img, gray = w2cs[k:k + 1], projs[k:k + 1], masks[k:k + 1], imgs[k:k + 1], grayimgs[k:k + 1]
            vertsw, proj_verts = clip_positions(vertices, w2c, proj)
            normals = ops.get_normals(vertsw[:, :, :3].contiguous(), faces.long())
            rast_out, _ = ops.rasterize(glctx, proj_verts, faces, resolution=resolution)
            feat, _ = ops.interpolate(torch.cat([normals, torch.ones_like(vertsw[:, :, :1])], 2).contiguous(), rast_out, faces)
            pred_normals = feat[:, :, :, :3].contiguous()
            pred_mask = ops.antialias(feat[:, :, :, 3:4].contiguous(), rast_out, proj_verts, faces).squeeze(-1)
            pred_normals = F.normalize(ops.antialias(pred_normals, rast_out, proj_verts, faces), p=2, dim=3)
            valid = (mask > 0) & (rast_out[:, :, :, 3] > 0)
            vn = pred_normals[valid].detach().cpu().numpy().astype(np.float64)
            vg = gray[valid].detach().cpu().numpy().astype(np.float64)
            matrix = ops.get_matrix(torch.from_numpy(vn), degree).numpy()
            sh = np.linalg.lstsq(matrix, vg, rcond=None)[0]
            v_normals.append(vn); v_gray.append(vg); v_imgs.append(img[valid]); v_masks.append(pred_mask)
            sh_coeffs.append(torch.from_numpy(sh.astype(np.float32)).to(imgs.device).unsqueeze(0))
    vn = np.concatenate(v_normals, 0)
    sh_global = np.linalg.lstsq(ops.get_matrix(torch.from_numpy(vn), degree).numpy(), np.concatenate(v_gray, 0), rcond=None)[0]
    sh_global = torch.from_numpy(sh_global.astype(np.float32)).to(imgs.device)
    radiance = ops.get_radiance(sh_global, torch.from_numpy(vn.astype(np.float32)).to(imgs.device), degree).unsqueeze(-1)
    albedo_mean = (torch.cat(v_imgs, 0) / radiance).mean(0).detach()
    return dict(sh_coeffs=torch.cat(sh_coeffs, 0), sh_global=sh_global, albedo_mean=albedo_mean, valid_masks=torch.cat(v_masks, 0))


def stage1_loss(ops, glctx, vertices, faces, albedo, sh_coeff, w2c, proj, img, mask, resolution, sfs_weight, degree=3):
    """One stage-1 batch (mesh_sfs_optim.py:205-233): vertices are constants, antialias is applied to the interpolated normals
    and albedo BEFORE shading; loss = sfs_weight * L1 over the valid pixels.  albedo [1,V,3], sh_coeff [n,9] carry the graph."""
    n = w2c.shape[0]
    vertsw, proj_verts = clip_positions(vertices, w2c, proj)
    normals = ops.get_normals(vertsw[:, :, :3].contiguous(), faces.long())
    rast_out, _ = ops.rasterize(glctx, proj_verts, faces, resolution=resolution)
    feat, _ = ops.interpolate(torch.cat([normals, albedo.expand(n, -1, -1)], dim=2).contiguous(), rast_out, faces)
    pred_normals = F.normalize(ops.antialias(feat[:, :, :, :3].contiguous(), rast_out, proj_verts, faces), p=2, dim=3)
    rast_albedo = ops.antialias(feat[:, :, :, 3:6].contiguous(), rast_out, proj_verts, faces)
    valid_idx = torch.where((mask > 0) & (rast_out[:, :, :, 3] > 0))
    radiance = ops.get_radiance(sh_coeff[valid_idx[0]].contiguous(), pred_normals[valid_idx].contiguous(), degree).unsqueeze(-1)
    pred_img = radiance * rast_albedo[valid_idx]
    return sfs_weight * F.l1_loss(pred_img, img[valid_idx])


class Stage1Cache:
    """Everything of the stage-1 batch (mesh_sfs_optim.py:205-233) that does not depend on the optimised variables.  Stage 1 moves
    only the albedo and the SH coefficients: the vertices, hence the clip positions, the rasterization, the antialiased +
    normalised normals, their SH basis and the validity mask are the same in every one of its iterations.  They are computed once
    per view here; an iteration is then interpolate(albedo) -> antialias -> (basis . sh) * albedo -> masked L1 — the reference
    (and `stage1_loss`) re-rasterizes and re-derives the normals 150 x num / batch times."""

    def __init__(self, ops, glctx, vertices, faces, w2cs, projs, masks, resolution, degree=3, chunk=8):
        num = w2cs.shape[0]
        self.faces, self.resolution = faces, resolution
        rast, pos, basis, valid = [], [], [], []
        with torch.no_grad():
            for k in range(0, num, chunk):
                w2c, proj = w2cs[k:k + chunk], projs[k:k + chunk]
                vertsw, proj_verts = clip_positions(vertices, w2c, proj)
                normals = ops.get_normals(vertsw[:, :, :3].contiguous(), faces.long())
                r, _ = ops.rasterize(glctx, proj_verts, faces, resolution=resolution)
                feat, _ = ops.interpolate(normals.contiguous(), r, faces)
                pn = F.normalize(ops.antialias(feat.contiguous(), r, proj_verts, faces), p=2, dim=3)
                rast.append(r); pos.append(proj_verts)
                basis.append(ops.get_matrix(pn.reshape(-1, 3), degree).reshape(*pn.shape[:3], -1))     # [c,H,W,9]
                valid.append(((masks[k:k + chunk] > 0) & (r[..., 3] > 0)).to(torch.float32))
        self.rast, self.pos = torch.cat(rast), torch.cat(pos)
        self.basis, self.valid = torch.cat(basis), torch.cat(valid)


def stage1_loss_cached(ops, cache, idx, albedo, sh_coeff, img, sfs_weight):
    """`stage1_loss` for the views `idx` of a Stage1Cache: same value and gradients (the constant sub-graph is cached, the boolean
    gather of the valid pixels becomes a mask — no host synchronisation)."""
    rast, pos = cache.rast[idx], cache.pos[idx]
    feat, _ = ops.interpolate(albedo.expand(idx.shape[0], -1, -1).contiguous(), rast, cache.faces)
    rast_albedo = ops.antialias(feat.contiguous(), rast, pos, cache.faces)
    radiance = (cache.basis[idx] * sh_coeff[:, None, None, :]).sum(-1, keepdim=True)
    valid = cache.valid[idx].unsqueeze(-1)
    diff = ((radiance * rast_albedo - img) * valid).abs().sum()
    return sfs_weight * diff / (3.0 * valid.sum())


class PermChunks:
    """The loops' per-epoch permutations (mesh_sfs_optim.py:195,249: `perm = torch.randperm(num).cuda()`) without a host-to-device copy
    per epoch: uniform permutations drawn from the run's generator kChunk epochs at a time (one vectorised draw: argsort of i.i.d.
    uniforms), into ONE pinned staging buffer allocated once, and copied with one non-blocking transfer on the step's stream (stream
    order protects the rows the queued steps still read).  What it replaces, measured at the metric config: a
    pageable host-to-device copy per epoch — the reference's `.cuda()` — stalls the launch queue for ~25 us = 9 % of a 48-view stage-2
    step; a freshly pinned buffer per chunk (hipHostMalloc) ~1 ms per 64 epochs = 5 %; 64 torch.randperm calls + 64 row copies ~1 ms of
    host time per refill.  next() -> (perm [num] int32, masks [ceil(num / batch), num] int32): row j of `masks` is 1 on the views of the
    epoch's j-th batch `perm[j*batch:(j+1)*batch]` — the selection form HamStage1.step(sel=...) takes (no per-step index kernels)."""
    kChunk = 64

    def __init__(self, num, batch, gen, dev):
        self.num, self.batch, self.gen, self.dev = int(num), int(batch), gen, dev
        self.nb = (self.num + self.batch - 1) // self.batch
        words = self.num + self.nb * self.num
        self._pin = torch.empty((self.kChunk, words), dtype=torch.int32)
        if dev.type == "cuda":
            self._pin = self._pin.pin_memory()
        self._dev = torch.empty((self.kChunk, words), dtype=torch.int32, device=dev)
        self._at, self._copied = self.kChunk, None

    def next(self):
        if self._at == self.kChunk:
            if self._copied is not None:
                self._copied.synchronize()                                              # the previous chunk's transfer has read the staging buffer (64 epochs ago)
            perms = torch.rand(self.kChunk, self.num, generator=self.gen).argsort(dim=1)
            batch_of_view = perms.argsort(dim=1) // self.batch                          # [chunk, num]: which batch of its epoch a view is in
            masks = (batch_of_view[:, None, :] == torch.arange(self.nb)[None, :, None]).to(torch.int32)
            self._pin[:, :self.num].copy_(perms)
            self._pin[:, self.num:].copy_(masks.reshape(self.kChunk, -1))
            self._dev.copy_(self._pin, non_blocking=True)
            if self.dev.type == "cuda":
                self._copied = torch.cuda.Event()
                self._copied.record()
            self._at = 0
        row = self._dev[self._at]
        self._at += 1
        return row[:self.num], row[self.num:].view(self.nb, self.num)


class FixedPerms:
    """PermChunks' interface over GIVEN permutations ([epochs, num] integers: e.g. the sequence a reference run drew with torch.randperm,
    mesh_sfs_optim.py:195,249) — run_ham(perms=...) replays them instead of drawing its own."""

    def __init__(self, perms, batch, dev):
        perms = torch.as_tensor(perms).to(torch.int64).cpu()
        num = perms.shape[1]
        if not bool((perms.sort(dim=1).values == torch.arange(num)[None]).all()):
            raise RuntimeError("FixedPerms: every row must be a permutation of range(num)")
        nb = (num + int(batch) - 1) // int(batch)
        batch_of_view = perms.argsort(dim=1) // int(batch)
        masks = (batch_of_view[:, None, :] == torch.arange(nb)[None, :, None]).to(torch.int32)
        self._perms, self._masks, self._at = perms.to(torch.int32).to(dev), masks.to(dev), 0

    def next(self):
        if self._at >= self._perms.shape[0]:
            raise RuntimeError(f"FixedPerms: the run asks for epoch {self._at + 1}, {self._perms.shape[0]} permutations were given")
        self._at += 1
        return self._perms[self._at - 1], self._masks[self._at - 1]


class Stage2Loop:
    """The stage-2 loop of mesh_sfs_optim.py:242-317 as the product runs it: per epoch a fresh permutation of the capture's views (:249),
    per batch ONE fused step on `perm[k:k+batch]` — the capture's images, masks, start silhouettes, cameras and SH rows stay where they
    are and the step indexes them through the batch's view indices (HamStage2(n_src=num), ABI 6: no per-batch gathers, the per-segment
    valid_mask table built once), the Adam update of (delta, albedo) applied by the kernel that forms their gradients.  The loop never
    synchronises: the sticky overflow flags of every workspace are read every `check_every` epochs (one 16-byte copy each), before a
    workspace is dropped, and at the end.  run_ham drives it; bench.py times it (`run_ham_stage2`)."""

    def __init__(self, vertices_tmp, faces, delta, albedo, sh_fixed, imgs, masks, valid_masks, w2cs, projs, resolution, weights, elm,
                 optimizer, batch, epoch_sfs, gen, aa_cap=None, bin_cap=0, check_every=64, group=None, exchange=None, perms=None):
        """perms: explicit permutations [epochs, num] instead of draws from `gen` (FixedPerms).
        group: torch.distributed process group over whose ranks every batch's views are sharded (None: one rank).  Every rank holds the
        whole capture and the same `gen` seed — the permutations are drawn identically everywhere — and steps views
        shard_batch(len(batch), rank, world) of perm[k:k+batch] with n_global = len(batch); `exchange`: what carries the step's ONE
        all-reduce (fmhr_amd.dist.make_stream_all_reduce's result; None: negotiated here)."""
        from .ham import HamStage2
        self.HamStage2 = HamStage2
        self.vertices_tmp, self.faces, self.delta, self.albedo = vertices_tmp, faces, delta, albedo
        self.args = tuple(t.contiguous() for t in (sh_fixed, w2cs, projs, imgs, masks, valid_masks))
        self.H, self.W = resolution
        self.weights, self.elm, self.opt = dict(weights), elm, optimizer
        self.num, self.batch, self.epoch_sfs, self.gen = imgs.shape[0], int(batch), int(epoch_sfs), gen
        self.aa_cap, self.bin_cap, self.check_every = aa_cap, bin_cap, check_every
        self.steppers, self.epoch, self.hs, self.perm, self._pending = {}, 0, None, None, []
        self._k, self._cur = None, None
        self.dev = vertices_tmp.device
        self._perms = FixedPerms(perms, self.batch, self.dev) if perms is not None else None
        self.group, self.rank, self.world, self.exchange, self.monitor, self._zero = group, 0, 1, None, None, None
        if group is not None:
            import torch.distributed as tdist
            from . import dist as fdist
            self.rank, self.world = tdist.get_rank(group), tdist.get_world_size(group)
            self.exchange = exchange if exchange is not None else fdist.make_stream_all_reduce(group, numel=12 * vertices_tmp.shape[0] + 8)
            self.monitor = fdist.ExchangeMonitor(group, self.exchange if hasattr(self.exchange, "gave_up") else None, self.dev)

    def _next_perm(self):
        """the epoch's permutation (:249) as an int32 device row (PermChunks below)"""
        if self._perms is None:
            self._perms = PermChunks(self.num, self.batch, self.gen, self.dev)
        return self._perms.next()[0]

    def check(self):
        """synchronous: raise if a work list of any live workspace overflowed in any step so far (one device -> host copy each) or — a
        view-sharded run — if the ranks no longer hold identical parameters / the exchange reported a failure (ExchangeMonitor).  On N ranks
        the overflow verdict is COLLECTIVE (MAX over the ranks of the flags and of the fullest bin): every rank raises the same
        WorkListOverflow, so that run_ham's grow-and-repeat is taken by all of them together and no rank is left in the next collective."""
        from .ham import BIN_CAP_DEFAULT, WorkListOverflow
        self._pending = []
        if self.monitor is None:
            for hs_ in self.steppers.values():
                hs_.check_overflow()
            return
        flags = [0, 0, 0, 0]                                                         # antialias list, fragment bins, fullest bin, a clamped view index
        for hs_ in self.steppers.values():
            try:
                hs_.check_overflow()
            except WorkListOverflow as e:
                flags[0 if e.which == "aa" else 1] = 1
                flags[2] = max(flags[2], int(e.needed or 0))
            except RuntimeError:
                flags[3] = 1
        v = torch.tensor(flags, dtype=torch.int64, device=self.dev)
        torch.distributed.all_reduce(v, op=torch.distributed.ReduceOp.MAX, group=self.group)
        aa, bins, need, bad_idx = v.tolist()
        state = [st_[k_] for p_ in (self.delta, self.albedo) for st_ in [self.opt.state.get(p_)] if st_ for k_ in ("exp_avg", "exp_avg_sq")]
        self.monitor.verify([self.delta, self.albedo], "stage-2 parameters (delta, albedo) after epoch %d" % self.epoch,
                            resync=[self.delta, self.albedo] + state)
        if bad_idx:
            raise RuntimeError("Stage2Loop: a step on some rank was given a view index outside the capture")
        if bins:
            raise WorkListOverflow(f"Stage2Loop: a fragment bin of the rasterizer overflowed on some rank (capacity {self.bin_cap or BIN_CAP_DEFAULT}, the fullest "
                                   f"segment held {need}): fragments were dropped", "bins", need)
        if aa:
            raise WorkListOverflow("Stage2Loop: the antialias work-item list overflowed on some rank: construct with a larger aa_cap", "aa")

    def poll(self):
        """the periodic check WITHOUT a synchronisation: inspect the copies of the sticky flags issued at the previous poll —
        their transfers completed epochs ago — and issue this poll's (16 bytes per workspace into pinned memory, non-blocking, on the
        step's stream).  An overflow is reported one polling interval late at most; check() before a workspace is dropped and at the end
        of the run is synchronous and exact.  Cadence: a device-to-host copy, however small, ends in a system-scope release — the L2s
        are written back — and costs the steps around it ~58 us at the metric config (tools/experiments/r05_loop_cost.py: +3.3 % at
        every 16 epochs, +2.2 % without any poll); every 64 epochs — the cadence of the permutation refill — it is below 0.3 %."""
        if self.monitor is not None:                                                # a view-sharded run synchronises here: one small collective per
            return self.check()                                                     # `check_every` epochs (>= 128 steps) is what keeps a dead exchange from going unnoticed
        for hs_, pin, ev in getattr(self, "_pending", []):
            ev.synchronize()
            f = pin.tolist()
            if f[2] != 0 or f[3] != 0:
                hs_.check_overflow()                                                    # raises with the message that names the list
        self._pending = []
        if self.dev.type != "cuda":
            return self.check()
        for hs_ in self.steppers.values():
            if getattr(hs_, "_flag_pin", None) is None:
                hs_._flag_pin = torch.zeros(4, dtype=torch.int32).pin_memory()
            hs_._flag_pin.copy_(hs_._flags(), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending.append((hs_, hs_._flag_pin, ev))

    def _zero_share(self):
        """this rank has no view in the batch (fewer views than ranks — a ragged last batch): it still enters the step's exchange, with a
        zero share, and applies the same combine + Adam update as every other rank"""
        import ctypes
        from ._lib import call, ptr, stream_ptr
        V = self.vertices_tmp.shape[0]
        if self._zero is None:
            self._zero = _ZeroShare(V, self.dev)
        z = self._zero
        z.bucket.zero_()
        if hasattr(self.exchange, "all_reduce_"):
            self.exchange.all_reduce_(z.bucket, stream_ptr(self.dev))
        else:
            torch.distributed.all_reduce(z.bucket, group=self.exchange)
        ad, aa = self.opt.fused_param(self.delta), self.opt.fused_param(self.albedo)
        call("fmhr_ham_combine", V, ptr(z.bucket), ctypes.byref(ad), ctypes.byref(aa), stream_ptr(self.dev))
        return z

    def step_batch(self, perm, k, after_step=None):
        """ONE fused step on the batch perm[k:k+batch] (this rank's share of it)"""
        V = self.vertices_tmp.shape[0]
        sh, w2cs, projs, imgs, masks, valid_masks = self.args
        nb = min(self.batch, self.num - k)
        s0, s1, reg = 0, nb, None
        if self.group is not None:
            from .dist import shard_batch
            part = shard_batch(nb, V, self.rank, self.world)
            (s0, s1), reg = part["views"], part["reg_range"]
        n = s1 - s0
        if n == 0:
            self.hs = self._zero_share()
            return
        idx = perm[k + s0: k + s1]                                                  # a view of the permutation: no copy
        hs = self.steppers.get(n)
        if hs is None:
            hs = self.steppers[n] = self.HamStage2(self.faces, V, n, self.H, self.W, self.weights, self.elm, aa_cap=self.aa_cap,
                                                   bin_cap=self.bin_cap, n_src=self.num)
        hs.n_global = nb                                                            # the MSE divisor is the BATCH's n * H * W (:295), ragged or sharded
        hs.step(self.vertices_tmp, self.delta.detach(), self.albedo.detach()[0], sh, w2cs, projs, imgs, masks, valid_masks,
                view_idx=idx, optimizer=self.opt, params=(self.delta, self.albedo), group=self.exchange, reg_range=reg)
        self.hs = hs
        if after_step is not None:
            after_step(hs, idx)

    def step_next(self, after_step=None):
        """the loop's next fused step: the next batch of the running epoch, or — at an epoch boundary — a fresh permutation first (:249; the
        albedo-weight switch of :250-251 at epoch_sfs // 2).  Returns True when the step ended an epoch (the overflow / exchange checks run
        there every `check_every` epochs).  run_epoch() = step_next() until it returns True; bench.py times step_next() itself."""
        if self._k is None:
            self._cur = self._next_perm()
            if self.epoch == self.epoch_sfs // 2:
                self.weights["albedo"] = self.weights["albedo"] / 10000               # :250-251
                for hs_ in self.steppers.values():                                      # the weights are arguments of every backward call: the
                    hs_.weights = dict(self.weights)                                    # workspaces and their mesh tables stay (rebuilding them: 8 ms each)
            self._k = 0
        self.step_batch(self._cur, self._k, after_step)
        self._k += self.batch
        if self._k < self.num:
            return False
        self.perm, self._k = self._cur, None
        self.epoch += 1
        if self.check_every and self.epoch % self.check_every == 0:
            self.poll()
        return True

    def run_epoch(self, after_step=None):
        """one epoch: ceil(num / batch) fused steps on a fresh permutation (the last batch ragged when batch does not divide num)"""
        if self._k is not None:
            raise RuntimeError("Stage2Loop.run_epoch: an epoch is under way (step_next)")
        while not self.step_next(after_step):
            pass


class _ZeroShare:
    """the bucket [P | R | losses] of a rank without a view in the batch: all zero going into the exchange, the batch-global gradients and
    loss vector coming out of it (fmhr_ham_combine)"""

    def __init__(self, V, dev):
        self.V = V
        self.bucket = torch.zeros(12 * V + 8, dtype=torch.float32, device=dev)
        self.losses = self.bucket[12 * V:]

    def loss_dict(self):
        from .ham import LOSS_NAMES
        return dict(zip(LOSS_NAMES, self.losses.tolist()))


def run_ham(vertices, faces, imgs, grayimgs, masks, w2cs, projs, resolution, conf, out_dir=None, log_every=0, seed=0, scan_id=None,
            check_every=64, aa_cap=None, bin_cap=0, vertex_order=None, group=None, exchange=None, perms=None):
    """The HAM optimisation end to end on GPU tensors.  conf: dict with the keys of conf/demo_sfs.conf that the loop reads.
    `group` (a torch.distributed process group, e.g. dist.group.WORLD under torchrun; None: one rank): the run is VIEW-SHARDED over its
    ranks (north_star: "views shard one-per-GPU ... all-reduce ... each optimizer step").  Every rank calls run_ham with the
    SAME arguments — the whole capture and the same seed, so that all ranks draw the same permutations (mesh_sfs_optim.py:195,249) — and
      * init + stage 1: rank r builds the constant sub-graph of views shard_views(num, r, world) only; the global SH least squares and the
        mean albedo come from all-reduced float64 sums, the start silhouettes are gathered once; an iteration is the rank's fused step on
        the batch's views it holds, ONE all-reduce of the albedo gradient and the replicated Adam update (fmhr_amd.dist.ShardedStage1);
      * stage 2: of every batch perm[k:k+batch] rank r steps views shard_batch(len(batch), r, world) with the MSE divisor of the whole
        batch, the regularisers split by vertex range; ONE exchange of [P | R | losses] per step, then combine + Adam on every rank; a rank
        that has no view of a ragged batch enters the exchange with a zero share (Stage2Loop);
      * every `check_every` epochs and at the end the ranks' parameters are fingerprinted against each other and the exchange's failure
        state is read (fmhr_amd.dist.ExchangeMonitor): a dead or diverged exchange raises ExchangeError on every rank;
      * every rank returns the same result; only rank 0 writes files.
    `exchange`: the object that carries the stage-2 all-reduce (default: fmhr_amd.dist.make_stream_all_reduce(group, 12 V + 8)).
    `perms`: dict(stage1=[epoch_albedo, num], stage2=[epoch_sfs, num]) of explicit permutations replayed instead of the run's own draws
    (either key optional) — e.g. the sequence a reference run drew.
    A step never synchronises, so a work list that overflows (antialias pairs, fragment bins: `aa_cap` / `bin_cap` of HamStage2) drops
    gradient terms without an error of its own: the STICKY overflow flags of every stage-2 workspace are read every `check_every`
    epochs (one 16-byte copy per workspace — a device-to-host copy drains the launch queue, so not every epoch: the demo's epoch is two
    steps, 0.7 ms of GPU work), before a workspace is discarded and at the end, and a set flag raises — whatever `log_every` is.
    Stage 2 steps MINIBATCHES of the resident capture (Stage2Loop: HamStage2(n_src=num) + view_idx = perm[k:k+batch], ABI 6): the six
    per-batch gathers of mesh_sfs_optim.py:255-260 do not exist, the Adam update rides in the step.
    The optimisation runs on a RENUMBERING of the vertices (fmhr_amd.ham.VertexOrder: first-touch order of the face list, the faces and
    their corner order untouched — the gathers of every kernel stay within few cache lines) and every per-vertex result is mapped back:
    what is returned and written has the caller's numbering, i.e. the child-face / appended-midpoint layout of the subdivided mesh that
    repose.py:14-24 relies on.  `vertex_order=False` (or FMHR_VERTEX_ORDER=0) optimises in the caller's numbering.
    Returns dict(vertices, delta, albedo, sh_coeffs, history, valid_masks, bin_depth_max).  With `out_dir` the outputs of mesh_sfs_optim.py:319-343 are written
    under the reference's names: '<scan_id>.pt' (sh_coeff, albedo), '<scan_id>.obj', '<scan_id>_c.obj' (vertex colours) and
    'rerender/mesh_<view>.png' (the last epoch's rendered batches); scan_id = None keeps the stem 'ham'."""
    from . import optim
    from .ham import HamStage2, VertexOrder, edge_length_mean
    from .data import save_obj, save_obj_mesh_with_color, save_rerender_png
    ops = product_ops()
    dev = vertices.device
    order = VertexOrder(faces, vertices.shape[0], enabled=vertex_order)
    faces_caller = faces
    vertices, faces = order.to_internal(vertices), order.faces
    degree, batch = int(conf.get("degree", 3)), int(conf["batch"])
    num = imgs.shape[0]
    H, W = resolution
    gen = torch.Generator(device="cpu").manual_seed(seed)
    history = []
    delta = torch.zeros_like(vertices).requires_grad_(True)
    vertices_tmp = vertices.detach().clone()
    elm = edge_length_mean(vertices_tmp, faces)                                      # :184-188
    rank, world = 0, 1
    if group is not None:
        import torch.distributed as tdist
        from . import dist as fdist
        rank, world = tdist.get_rank(group), tdist.get_world_size(group)
        if world > num:
            raise RuntimeError(f"run_ham: {world} ranks for a capture of {num} views: stage 1 shards the capture's views, one at least per rank")
        if exchange is None:
            exchange = fdist.make_stream_all_reduce(group, numel=12 * vertices.shape[0] + 8)
    d2 = degree * degree
    fused_s1 = os.environ.get("FMHR_STAGE1_OPBYOP", "0") != "1"                      # (the op-by-op stage 1 on the generic ops stays selectable: A/B, tests)
    # degree 1 / 2 (conf key `degree`; models/utils.py:188-226): the same fused stage-1 kernels on coefficient rows zero-padded to order 3,
    # with the gradient of the bands the degree does not have masked out before the (then separate) Adam update — a zero coefficient with a
    # zero gradient stays zero under Adam, so the padded rows ARE the lower-degree polynomial throughout (round 6; until then: op by op)
    band = torch.zeros(9, dtype=torch.float32, device=dev)
    band[:d2] = 1.0
    if fused_s1 and group is not None and world > 1:
        # ---- view-sharded init + stage 1: this rank's views of the capture
        from . import stage1 as s1
        v0, v1 = fdist.shard_views(num, rank, world)
        hs1 = s1.HamStage1(vertices.detach(), faces, w2cs[v0:v1].contiguous(), projs[v0:v1].contiguous(), imgs[v0:v1].contiguous(),
                           masks[v0:v1].contiguous(), resolution, conf["sfs_weight"], want_pred_masks=True)
        init = s1.init_sh(hs1, grayimgs[v0:v1].contiguous(), group=group, view0=v0, n_views_global=num, degree=degree)
        sh_local = F.pad(init["sh_coeffs"], (0, 9 - d2)).contiguous().requires_grad_(True)   # this rank's rows of :168-170 (order-3 form)
        albedo = (torch.zeros_like(vertices) + init["albedo_mean"]).unsqueeze(0).contiguous().requires_grad_(True)
        opt = optim.Adam([{'params': albedo, 'lr': conf["albedo_lr"]}, {'params': sh_local, 'lr': conf["sh_lr"]}])
        # (the one-shot peer-to-peer exchange is built for the stage-2 bucket: stage 1's 3 V + 1 floats go through RCCL / the torch group)
        sh1 = fdist.ShardedStage1(hs1, v0, num, group=exchange if isinstance(exchange, fdist.StreamAllReduce) else group)
        monitor = fdist.ExchangeMonitor(group, exchange if isinstance(exchange, fdist.StreamAllReduce) else None, dev)
        perms1 = FixedPerms(perms["stage1"], batch, dev) if (perms or {}).get("stage1") is not None else PermChunks(num, batch, gen, dev)
        ones = torch.ones(num, dtype=torch.int32, device=dev)
        for ep in range(int(conf["epoch_albedo"])):
            _, batch_masks = perms1.next()                                              # :195 — the same draw on every rank
            for j in range(batch_masks.shape[0]):
                loss, galbedo, gsh = sh1.step_sel(albedo, sh_local, ones if batch >= num else batch_masks[j])
                albedo.grad, sh_local.grad = galbedo.view_as(albedo), gsh if degree == 3 else gsh * band
                opt.step()
            if log_every and ep % log_every == 0:
                history.append(("albedo", ep, float(loss)))
            if check_every and (ep + 1) % check_every == 0:
                monitor.verify([albedo], "stage-1 albedo after epoch %d" % (ep + 1))
        monitor.verify([albedo], "stage-1 albedo at the end of stage 1")
        sh9 = torch.zeros(num, 9, dtype=torch.float32, device=dev)                     # every rank needs every view's row: stage 2 picks any view
        sh9[v0:v1] = sh_local.detach()
        tdist.all_reduce(sh9, group=group)
        albedo = albedo.detach().clone().requires_grad_(True)
        del hs1, sh1
        sh_fixed = sh9.contiguous()
        sh_coeffs = sh9[:, :d2].contiguous()
    elif fused_s1:
        # The constant sub-graph of init + stage 1 (:137-148 = :207-218: rasterization, antialiased + normalised normals, validity,
        # start silhouettes) is computed ONCE for all views by HamStage1; the SH least squares of :150-174 then run on it (normal
        # equations accumulated on the GPU, 9 x 9 solves on the host: `init_sh` above — the reference's own per-view host loop — took
        # 2.8 of the 3.0 s of a 48-view run)
        from . import stage1 as s1
        hs1 = s1.HamStage1(vertices.detach(), faces, w2cs, projs, imgs, masks, resolution, conf["sfs_weight"], want_pred_masks=True)
        init = s1.init_sh(hs1, grayimgs, degree=degree)
        sh9 = F.pad(init["sh_coeffs"], (0, 9 - d2)).contiguous().requires_grad_(True)   # :168-170 (the rows in their order-3 form)
        albedo = (torch.zeros_like(vertices) + init["albedo_mean"]).unsqueeze(0).contiguous().requires_grad_(True)   # :175-176
        # ---- stage 1: albedo + SH (:193-240), the fused step (fmhr_amd.stage1.HamStage1): the constant sub-graph and the linear
        # interpolate + antialias operator are built once, an iteration is one C call (two launches), the Adam update of both
        # parameters applied inside it.  `stage1_loss` / `stage1_loss_cached` above are the op-by-op forms the parity tests compare it with.
        opt = optim.Adam([{'params': albedo, 'lr': conf["albedo_lr"]}, {'params': sh9, 'lr': conf["sh_lr"]}])
        perms1 = FixedPerms(perms["stage1"], batch, dev) if (perms or {}).get("stage1") is not None else PermChunks(num, batch, gen, dev)   # (no per-epoch copy, no per-step index kernels)
        for ep in range(int(conf["epoch_albedo"])):
            _, batch_masks = perms1.next()                                              # :195
            for j in range(batch_masks.shape[0]):
                sel = None if batch >= num else batch_masks[j]
                if degree == 3:
                    loss_view, _, _ = hs1.step(albedo, sh9, sel=sel, optimizer=opt, params=(albedo, sh9))
                else:                                                                   # the step's gradients, the absent bands masked, then Adam
                    loss_view, galbedo, gsh = hs1.step(albedo, sh9, sel=sel)
                    albedo.grad, sh9.grad = galbedo.view_as(albedo), gsh * band
                    opt.step()
            if log_every and ep % log_every == 0:
                history.append(("albedo", ep, float(loss_view.sum())))
        del hs1
        sh_fixed = sh9.detach()
        sh_coeffs = sh9.detach()[:, :d2].contiguous()
    else:
        # FMHR_STAGE1_OPBYOP=1 (comparison runs; until round 5 the path of degrees 1 / 2): init + stage 1 op by op on the generic ops with
        # degree-d get_matrix / get_radiance and torch autograd (cached constant sub-graph); stage 2 freezes the SH coefficients (:243) and a
        # zero-padded row IS the lower-degree polynomial
        init = init_sh(ops, vertices.detach(), faces, imgs, grayimgs, masks, w2cs, projs, resolution, degree)
        sh_coeffs = init["sh_coeffs"].clone().requires_grad_(True)
        albedo = (torch.zeros_like(vertices) + init["albedo_mean"]).unsqueeze(0).contiguous().requires_grad_(True)
        cache = Stage1Cache(ops, ops.RasterizeGLContext(), vertices.detach(), faces, w2cs, projs, masks, resolution, degree)
        opt = optim.Adam([{'params': albedo, 'lr': conf["albedo_lr"]}, {'params': sh_coeffs, 'lr': conf["sh_lr"]}])
        for ep in range(int(conf["epoch_albedo"])):
            perm = torch.randperm(num, generator=gen).to(dev)
            for k in range(0, num, batch):
                idx = perm[k:k + batch]
                loss = stage1_loss_cached(ops, cache, idx, albedo, sh_coeffs[idx], imgs[idx], conf["sfs_weight"])
                opt.zero_grad()
                loss.backward()
                opt.step()
            if log_every and ep % log_every == 0:
                history.append(("albedo", ep, float(loss)))
        del cache
        sh_fixed = F.pad(sh_coeffs.detach(), (0, 9 - degree ** 2)).contiguous()
    # ---- stage 2: vertex offsets + albedo (:242-317): the fused step on minibatches of the resident capture (Stage2Loop above)
    weights = dict(sfs=conf["sfs_weight"], lap=conf["lap_weight"], albedo=conf["albedo_weight"], mask=conf["mask_weight"],
                   edge=conf["edge_weight"], delta=conf["delta_weight"])
    epoch_sfs = int(conf["epoch_sfs"])
    # A fragment bin that overflows drops fragments (the sticky flag says so where the loop synchronises): with the default capacity — 512
    # entries per 64-pixel segment, the fullest one measured holds 336-448 — stage 2 is then REPEATED from its start state with room for the
    # depth the run needed (x 1.25, at least 2 x): at most one stage-2 run is lost (0.2 s for the demo's 600 steps), and the reserved span
    # follows the measured depth instead of being sized for the worst case.  An explicit `bin_cap` is the caller's
    # decision: its overflow raises.
    from .ham import WorkListOverflow
    start = dict(albedo=albedo.detach().clone(), gen=gen.get_state(), history=len(history))
    cap, grown = bin_cap, []
    while True:
        opt = optim.Adam([{'params': delta, 'lr': conf["lr"]}, {'params': albedo, 'lr': conf["albedo_lr"]}])
        loop = Stage2Loop(vertices_tmp, faces, delta, albedo, sh_fixed, imgs, masks, init["valid_masks"], w2cs, projs, (H, W), weights, elm, opt,
                          batch, epoch_sfs, gen, aa_cap=aa_cap, bin_cap=cap, check_every=check_every, group=group if world > 1 else None,
                          exchange=exchange if world > 1 else None, perms=(perms or {}).get("stage2"))
        rendered_img, rendered_idx, perm_last = [], [], None                         # :315-317
        keep = None
        if out_dir is not None:
            def keep(hs_, idx_):
                rendered_img.append(hs_.views()["tmp_img"].clone())
                rendered_idx.append(idx_.clone())
        try:
            for ep in range(epoch_sfs):
                loop.run_epoch(after_step=keep if ep == epoch_sfs - 1 else None)     # the last epoch's renders are the "rerender" images
                if log_every and ep % log_every == 0:
                    # (on N ranks the losses are read WITHOUT the local overflow check of loss_dict(): a rank that raised alone would leave
                    # the others in the next exchange — the flags are judged collectively by loop.check() / poll())
                    from .ham import LOSS_NAMES
                    history.append(("sfs", ep, dict(zip(LOSS_NAMES, loop.hs.losses.tolist())) if world > 1 else loop.hs.loss_dict()))
            loop.check()
            break
        except WorkListOverflow as e:
            if e.which != "bins" or bin_cap or len(grown) >= 3:
                raise
            from .ham import BIN_CAP_DEFAULT
            cap = max(2 * (cap or BIN_CAP_DEFAULT), (int(1.25 * (e.needed or 0)) + 63) // 64 * 64)
            grown.append(cap)
            with torch.no_grad():                                                    # back to the start of stage 2: parameters, draws, log
                delta.zero_()
                albedo.copy_(start["albedo"])
            gen.set_state(start["gen"])
            del history[start["history"]:]
            del loop
    if out_dir is not None and loop.perm is not None:
        perm_last = loop.perm.cpu().numpy()
    if out_dir is not None and world > 1:
        # the last epoch's renders live on the ranks that stepped their views: summed into one [num,H,W,3] array (every view is rendered
        # by exactly one rank of exactly one batch), once
        full = torch.zeros(num, H, W, 3, dtype=torch.float32, device=dev)
        if rendered_img:
            full[torch.cat(rendered_idx).long()] = torch.cat(rendered_img, 0)
        tdist.all_reduce(full, group=group)
        rendered_img = [full[torch.from_numpy(perm_last).to(dev).long()]]
    out_vertices = order.to_caller((vertices_tmp + delta).detach())
    albedo_out = order.to_caller(albedo.detach(), 1)
    result = dict(vertices=out_vertices, delta=order.to_caller(delta.detach()), albedo=albedo_out, sh_coeffs=sh_coeffs.detach(),
                  history=history, valid_masks=init["valid_masks"],
                  # the fullest fragment bin of the run (0: atomic-key rasterizer) — what `bin_cap` has to cover (HamStage2.bin_depth_max)
                  bin_depth_max=max([hs_.bin_depth_max() for hs_ in loop.steppers.values()] or [0]),
                  # the capacities stage 2 was repeated with after a fragment-bin overflow ([]: the default sufficed)
                  bin_cap_grown=grown)
    if world > 1:
        result["exchange"] = dict(loop.monitor.last or {}, path=type(loop.exchange).__name__, ranks=world)
    if out_dir is not None and rank == 0:                                            # :319-343
        os.makedirs(out_dir, exist_ok=True)
        stem = 'ham' if scan_id is None else '%d' % scan_id
        torch.save({'sh_coeff': sh_coeffs.detach().cpu(), 'albedo': albedo_out.cpu()}, os.path.join(out_dir, stem + '.pt'))   # :321
        nv, nf = out_vertices.cpu().numpy(), faces_caller.cpu().numpy()
        save_obj(os.path.join(out_dir, stem + '.obj'), nv, nf)                                                             # :327
        color = torch.clamp(0.5 * albedo_out, 0, 1)[0].cpu().numpy()
        save_obj_mesh_with_color(os.path.join(out_dir, stem + '_c.obj'), nv, nf, color[:, 2::-1])                          # :334-335
        os.makedirs(os.path.join(out_dir, 'rerender'), exist_ok=True)                                                     # :337
        if rendered_img:
            rendered = torch.cat(rendered_img, 0).cpu().numpy()
            for i, idx in enumerate(perm_last):                                                                           # :342-343
                save_rerender_png(os.path.join(out_dir, 'rerender', 'mesh_%02d.png' % idx), rendered[i])
        result["files"] = sorted(os.listdir(out_dir))
    return result
