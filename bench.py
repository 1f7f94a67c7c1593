#!/usr/bin/env python3
"""bench.py — HAM stage-2 iterations/sec on MI355X (BASELINE.json metric).

One "step" = one full stage-2 iteration of /root/reference/mesh_sfs_optim.py:253-310 on the metric workload (48 views,
512x512, sub3 mesh V=49281/F=98432 — the sizes train_mlp.py:150-154 hard-codes): render + six losses + backward + gradient
exchange + Adam step.  Before --warmup the optimisation is PRE-ROLLED untimed until the antialias work list is stationary (the
synthetic targets roughen the contour: 2.4 k -> ~55 k silhouette pairs over the first ~50 steps, and the step slows by 6-7 % with
it), so that a 20-step run and a 200-step run time the same workload.
`python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches it under torch.distributed.run, one rank per GPU,
views sharded contiguously over ranks (strong scaling: the 48-view batch is fixed), ONE packed all-reduce per step (the
photometric share of the gradient travels with its L1 divisor left out: fmhr_amd/dist.py).

Prints ONE JSON line on rank 0 with the contract fields plus:
  roofline     — dominant launch group of the step: algorithmic bytes (SURVEY.md §8d per-primitive table) / its mean HIP-event
                 duration over the timed region, vs the 8 TB/s HBM peak
  step_roofline— the whole step: `traffic_frac` = MEASURED HBM bytes (PMC profile of this command) / device time / peak and
                 `fused_bound_frac` = the fused lower bound (52 P + 96 V + 24 F per view) / device time / peak are the
                 physical fractions; `op_granular_equivalent` prices the step by the reference's primitive chain (7.4x more bytes
                 than the fused step moves: it exceeds the peak and is NOT a roofline fraction)
  collective   — N > 1: per-rank milliseconds of the step's one all-reduce and of the combine (+ Adam) kernel (HIP events)
  kernels      — mean HIP-event milliseconds per launch group (from an untimed profiling pass)
  cpu_baseline — the CPU oracle restatement ("port") timed on this box's host cores on a bounded sample (rank 0, N=1 only)
  stage1       — side measurement (N=1 only, after the timed region): the fused stage-1 iteration of the same optimisation
                 (mesh_sfs_optim.py:205-236) on the same workload; not part of `value`
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from fmhr_amd import _lib, dist as fdist, synthetic  # noqa: E402
from fmhr_amd.ham import HamStage2, VertexOrder, edge_length_mean, render_forward  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
SLOTS = ["geometry", "raster_clear", "raster_tris", "raster_large", "raster_resolve", "shade_fwd", "aa_fwd", "loss_pix",
         "aa_bwd", "shade_bwd", "transform_bwd", "normals_bwd", "regularisers"]


def alg_bytes(slot, n, P, V, F, nnz):
    """Algorithmic bytes per launch of a launch group = SURVEY.md §8d per-primitive table, regrouped by launch group
    (the groups sum to n*(508 P + 252 V + 72 F) + the ~34 MB/step view-independent constant)."""
    per_view = {
        "geometry": 28 * V,                                   # transform fwd: 12V r + 16V w
        "raster_tris": 16 * V + 12 * F,                       # rasterize fwd reads
        # the resolve pass also does the forward shading: rasterize fwd write + interpolate C=7 fwd + shade fwd
        # (the "shade_fwd" launch group is only the segment lists, the zero fill and the valid-pixel count: no algorithmic bytes)
        "raster_resolve": 16 * P + 92 * P + 28 * V + 12 * F,
        "aa_fwd": 64 * P + 32 * V + 24 * F,                   # antialias mask + image fwd
        "loss_pix": 88 * P,                                   # losses fwd + bwd
        "aa_bwd": 80 * P + 32 * V,                            # antialias bwd
        "shade_bwd": 168 * P + 88 * V + 24 * F,               # shade + interpolate + rasterize bwd
        "transform_bwd": 28 * V,
    }
    const = {
        "geometry": 24 * V + 12 * F,                          # normals fwd
        "normals_bwd": 24 * V + 12 * F,
        "regularisers": 4 * (8 * nnz + 24 * V) + 24 * F + 60 * V,
    }
    return n * per_view.get(slot, 0) + const.get(slot, 0)


def step_alg_bytes(n, P, V, F, nnz):
    return n * (508 * P + 252 * V + 72 * F) + (48 * V + 24 * F) + 4 * (8 * nnz + 24 * V) + (24 * F + 60 * V) + 2 * 7 * 12 * V


def traffic_binned(path):
    """which rasterizer form the committed traffic profile was taken with"""
    try:
        return bool(json.load(open(path)).get("_binned", False))
    except Exception:  # noqa: BLE001
        return False


def step_fused_bound_bytes(n, P, V, F, nnz):
    """the fully fused lower bound (inputs once, rast written + re-read, gradients once) + the per-step constant."""
    return n * (52 * P + 96 * V + 24 * F) + (48 * V + 24 * F) + 4 * (8 * nnz + 24 * V) + (24 * F + 60 * V) + 2 * 7 * 12 * V


class Timing:
    """HIP events for the per-launch-group timing hook of the C ABI (fmhr_timing_attach)."""

    def __init__(self, n_sets, stream):
        import ctypes
        self.ct = ctypes
        self.sets = []
        for _ in range(n_sets):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(2 * len(SLOTS))]
            for e in evs:
                e.record(stream)                      # creates the underlying hipEvent_t
            arr = (ctypes.c_void_p * len(evs))(*[e.cuda_event for e in evs])
            self.sets.append((evs, arr))

    def attach(self, i, only_slot=-1):
        _lib.call("fmhr_timing_attach", self.sets[i][1], 2 * len(SLOTS), only_slot)

    @staticmethod
    def detach():
        _lib.call("fmhr_timing_detach")

    def ms(self, i, slot):
        evs = self.sets[i][0]
        return evs[2 * slot].elapsed_time(evs[2 * slot + 1])


def cpu_baseline(scene, views_sample, seed, views_per_step=None):
    """Time the CPU oracle (restatement of the reference loop body on CPU torch + the C rasterizer) on a bounded sample.
    views_per_step: the views ONE step of the timed workload renders (loop mode: the minibatch; default: all the scene's views)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ham_oracle as oracle
    oracle.lib()
    # the GPU box gives one-GPU jobs a 16-CPU share whatever os.cpu_count() says: over-subscribing 256 threads is 30x slower
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = int(os.environ.get("FMHR_CPU_THREADS", min(avail, 16)))
    torch.set_num_threads(cores)
    cores_c = oracle.set_threads(cores)          # the C rasterizer / interpolate / antialias loop over VIEWS with OpenMP: one view per thread
    S = views_sample
    w2c, proj = scene["w2c"][:S].contiguous(), scene["proj"][:S].contiguous()
    sh = scene["sh0"][:S].contiguous()
    H, W = scene["H"], scene["W"]
    faces = scene["faces"]
    with torch.no_grad():
        ones = torch.ones(S, H, W)
        tg = oracle.ham_stage2_losses(scene["vertices_true"], torch.zeros_like(scene["vertices"]), scene["albedo_true"][None],
                                      scene["sh_true"][:S], w2c, proj, torch.zeros(S, H, W, 3), ones, ones, faces, (H, W),
                                      dict(sfs=0, lap=0, albedo=0, mask=0, edge=0, delta=0), 0.0)
    img, mask = tg["tmp_img"], (tg["rast"][..., 3] > 0).float()
    valid_mask = tg["pred_mask"]
    elm = float(oracle.edge_length_mean(scene["vertices"], faces))
    times = []
    for it in range(5):
        d = torch.zeros_like(scene["vertices"]).requires_grad_(True)
        a = scene["albedo0"][None].clone().requires_grad_(True)
        t0 = time.perf_counter()
        out = oracle.ham_stage2_losses(scene["vertices"], d, a, sh, w2c, proj, img, mask, valid_mask, faces, (H, W),
                                       scene["weights"], elm)
        out["loss"].backward()
        times.append(time.perf_counter() - t0)
    t = sorted(times)[len(times) // 2]
    n = views_per_step or scene["views"]
    return dict(value=1.0 / (t * n / S), unit="iterations/s", cores=cores, kind="port",
                sample=f"{S} of {n} views ({H}x{W}, V={scene['V']}), fwd+bwd, median of 5 = {t:.3f}s, scaled x{n / S:.0f}; "
                       f"C rasterize / interpolate / antialias: {cores_c} OpenMP threads over the views, torch ops: {cores} threads")


def stage1_side_measurement(scene, faces, w2c, proj, sh, tg, vtmp, H, W, dev, steps=100):
    """Not the headline metric: the OTHER loop of the HAM optimisation, stage 1 (mesh_sfs_optim.py:196-240: albedo + SH, vertices
    frozen), on the same workload — the fused HamStage1 step (two launches, Adam inside) timed with HIP events after the timed
    region, and priced against the bytes of its own tables (every table read once, the gradient image written and gathered once)."""
    from fmhr_amd.optim import Adam
    from fmhr_amd.stage1 import HamStage1
    t0 = time.perf_counter()
    hs1 = HamStage1(vtmp, faces, w2c, proj, tg["img"], tg["mask"], (H, W), scene["weights"]["sfs"])
    torch.cuda.synchronize()
    plan_ms = (time.perf_counter() - t0) * 1e3
    albedo = scene["albedo0"][None].to(dev).clone().requires_grad_(True)
    shc = sh.clone().requires_grad_(True)
    lr = scene["lr"]
    opt = Adam([{'params': albedo, 'lr': lr["albedo_lr"]}, {'params': shc, 'lr': lr.get("sh_lr", lr["albedo_lr"])}])
    for _ in range(5):
        hs1.step(albedo, shc, optimizer=opt, params=(albedo, shc))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        lv, _, _ = hs1.step(albedo, shc, optimizer=opt, params=(albedo, shc))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    # ... and the loop run_ham drives around it when a batch is PART of the capture (mesh_sfs_optim.py:195-204: a fresh permutation per
    # epoch, batches of it): half the views per step — the batch masks of 64 epochs drawn and transferred at once
    # (fmhr_amd.pipeline.PermChunks), handed to the step in the C ABI's own form — against the same selection with fixed masks, and
    # against the reference's form (torch.randperm(num).cuda() per epoch, index tensors per step), by the same clock
    from fmhr_amd.pipeline import PermChunks
    n_all = int(w2c.shape[0])
    loop = None
    if n_all >= 2:
        half = n_all // 2
        gen = torch.Generator().manual_seed(0)
        pc = PermChunks(n_all, half, gen, dev)
        fixed = pc.next()[1].clone()

        def epoch_masks():
            m = pc.next()[1]
            for j in range(2):
                hs1.step(albedo, shc, sel=m[j], optimizer=opt, params=(albedo, shc))

        def epoch_fixed():
            for j in range(2):
                hs1.step(albedo, shc, sel=fixed[j], optimizer=opt, params=(albedo, shc))

        def epoch_reference_form():
            perm = torch.randperm(n_all, generator=gen).to(dev)
            for k in (0, half):
                hs1.step(albedo, shc, idx=perm[k:k + half], optimizer=opt, params=(albedo, shc))
        res = {}
        for _ in range(2):                                           # two alternating rounds
            for name, fn in (("fixed_masks", epoch_fixed), ("loop", epoch_masks), ("reference_form", epoch_reference_form)):
                for _ in range(8):
                    fn()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(48):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(name, []).append(e0.elapsed_time(e1) / 96)
        loop = {"what": f"half-batch steps ({half} of {n_all} views): ms per step with fixed masks | run_ham's loop (PermChunks: a fresh permutation "
                        "per epoch, masks of 64 epochs per transfer) | the reference's form (randperm().cuda() per epoch + index tensors per step)",
                **{k: round(min(v), 4) for k, v in res.items()},
                "loop_over_fixed": round(min(res["loop"]) / min(res["fixed_masks"]), 4),
                "reference_form_over_fixed": round(min(res["reference_form"]) / min(res["fixed_masks"]), 4)}
    V = albedo.shape[1]
    # forward: row_ptr + entries + normal + target + R per row; backward: transposed entries + one R gather per entry + Adam state
    nbytes = hs1.rows * (4 + 12 + 12 + 12) + hs1.entries * 8 + hs1.entries * (8 + 12) + V * 3 * 4 * 7
    return {"what": "fused stage-1 (albedo + SH) step, mesh_sfs_optim.py:205-236, same views / mesh", "ms_per_step": round(ms, 4),
            "iterations_per_s": round(1e3 / ms, 1), "plan_ms_once": round(plan_ms, 1), "rows": hs1.rows, "entries": hs1.entries,
            "antialias_items": hs1.n_items, "table_bytes_per_step": nbytes, "gbs": round(nbytes / (ms * 1e-3) / 1e9, 1),
            "frac_of_hbm_peak": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "loss": round(float(lv.sum()), 6), "steps": steps,
            "run_ham_stage1": loop}


def run_length_side_measurement(Problem, stream, length):
    """Not the headline metric: the SAME optimisation from its first step over the reference's demo length (300 epochs x 2 batches
    of stage 2 = 600 steps, /root/reference/conf/demo_sfs.conf:9,24, mesh_sfs_optim.py:247-252) on a fresh workspace with fresh
    parameters.  `value` is timed on the stationary (pre-rolled) workload; a real run starts with a short antialias work list and grows
    sliver triangles as it goes: this is the mean over the whole run and the triangle kernel at its end."""
    p = Problem(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    slot = SLOTS.index("raster_tris")
    marks = sorted({min(32, length), length})
    probe = {}
    torch.cuda.synchronize()
    e0.record(stream)
    done = 0
    for m in marks:                                       # the last 8 steps before each mark carry an event pair around k_raster_tris
        for _ in range(max(0, m - 8 - done)):
            p.step()
        k = min(8, m - done - max(0, m - 8 - done))
        tm = Timing(k, stream)
        for i in range(k):
            tm.attach(i, slot)
            p.step()
        Timing.detach()
        probe[m] = (tm, k)
        done = m
    e1.record(stream)
    torch.cuda.synchronize()
    p.hs.check_overflow()
    out = {"what": f"steps 1..{length} of the same optimisation from scratch (the reference's demo run: 300 epochs x 2 batches), fresh "
                   "workspace; HIP events around the whole run (16 of the steps carry an event pair around k_raster_tris)",
           "steps": length, "ms_per_step_mean": round(e0.elapsed_time(e1) / length, 4),
           "iterations_per_s_mean": round(length / e0.elapsed_time(e1) * 1e3, 1),
           "aa_items_at_end": int(p.hs.aa_item_count())}
    for m, (tm, k) in probe.items():
        out[f"raster_tris_ms_at_step_{m}"] = round(sum(tm.ms(i, slot) for i in range(k)) / max(k, 1), 4)
    return out


def caller_numbering_side_measurement(scene_c, dev, stream, steps=100, preroll=192):
    """Not the headline metric: the same workload with the mesh in the numbering it ARRIVED in (the subdivision's: each level's midpoints
    behind all older vertices) instead of the product's first-touch renumbering (fmhr_amd.ham.VertexOrder, what run_ham applies and what
    `value` is timed on) — so that the record holds both.  Own targets, workspace and parameters; pre-rolled, HIP events around the run."""
    n, H, W, V = scene_c["views"], scene_c["H"], scene_c["W"], scene_c["V"]
    faces = scene_c["faces"].to(dev)
    w2c, proj, sh = scene_c["w2c"].to(dev).contiguous(), scene_c["proj"].to(dev).contiguous(), scene_c["sh0"].to(dev).contiguous()
    vt = scene_c["vertices"].to(dev).contiguous()
    tg = synthetic.make_targets(scene_c, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
    hs = HamStage2(faces, V, n, H, W, scene_c["weights"], edge_length_mean(vt, faces), n_global=n)
    from fmhr_amd.optim import Adam
    delta = torch.zeros(V, 3, device=dev, requires_grad=True)
    albedo = scene_c["albedo0"][None].to(dev).clone().requires_grad_(True)
    opt = Adam([{'params': delta, 'lr': scene_c["lr"]["lr"]}, {'params': albedo, 'lr': scene_c["lr"]["albedo_lr"]}])

    def step():
        hs.step(vt, delta.detach(), albedo.detach()[0], sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"], optimizer=opt, params=(delta, albedo))
    for _ in range(preroll):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(steps):
        step()
    e1.record(stream)
    torch.cuda.synchronize()
    hs.check_overflow()
    ms = e0.elapsed_time(e1) / steps
    return {"what": "the same workload with the vertices in the numbering the mesh arrived in (FMHR_VERTEX_ORDER=0) and, by the same clock, "
                    f"in the product's renumbering; each pre-rolled {preroll} steps, HIP events around {steps} steps",
            "ms_per_step": round(ms, 4), "iterations_per_s": round(1e3 / ms, 1)}


def run_ham_stage2_side_measurement(scene_c, batch, dev, stream, epochs_timed, preroll_steps=192, label=""):
    """Not the headline metric: the stage-2 loop THE PRODUCT RUNS (fmhr_amd.pipeline.Stage2Loop, what run_ham drives) — per epoch a fresh
    permutation of the capture's views (mesh_sfs_optim.py:249), per batch one fused step on perm[k:k+batch] with the capture's tensors
    resident and the views selected INSIDE the step (ABI 6: no per-batch gathers), fused Adam, overflow flags polled every 64 epochs — by
    HIP events, next to the bare step on fixed tensors of the same batch size by the same clock (within 3 %)."""
    from fmhr_amd.optim import Adam
    from fmhr_amd.pipeline import Stage2Loop
    num, H, W, V = scene_c["views"], scene_c["H"], scene_c["W"], scene_c["V"]
    order = VertexOrder(scene_c["faces"], V)                       # as run_ham: the optimisation runs on the first-touch renumbering
    sc = dict(scene_c)
    for k in ("vertices", "vertices_true", "albedo_true", "albedo0"):
        sc[k] = order.to_internal(sc[k])
    sc["faces"] = order.faces
    faces = sc["faces"].to(dev)
    w2c, proj, sh = sc["w2c"].to(dev).contiguous(), sc["proj"].to(dev).contiguous(), sc["sh0"].to(dev).contiguous()
    vt = sc["vertices"].to(dev).contiguous()
    tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
    elm = edge_length_mean(vt, faces)
    lr = sc["lr"]

    def fresh():
        delta = torch.zeros(V, 3, device=dev, requires_grad=True)
        albedo = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
        return delta, albedo, Adam([{'params': delta, 'lr': lr["lr"]}, {'params': albedo, 'lr': lr["albedo_lr"]}])

    def timed(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    steps_per_epoch = -(-num // batch)
    # the loop ...
    delta, albedo, opt = fresh()
    loop = Stage2Loop(vt, faces, delta, albedo, sh, tg["img"], tg["mask"], tg["valid_mask"], w2c, proj, (H, W), sc["weights"], elm, opt, batch,
                      1 << 30, torch.Generator(device="cpu").manual_seed(0))
    for _ in range(-(-preroll_steps // steps_per_epoch)):
        loop.run_epoch()
    # ... and the bare step: the first `batch` views, gathered ONCE, fixed tensors (what `value` times at the metric config)
    delta_b, albedo_b, opt_b = fresh()
    n = min(batch, num)
    hs = HamStage2(faces, V, n, H, W, sc["weights"], elm)
    fixed = tuple(t[:n].contiguous() for t in (sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"]))

    def bare():
        hs.step(vt, delta_b.detach(), albedo_b.detach()[0], *fixed, optimizer=opt_b, params=(delta_b, albedo_b))
    for _ in range(preroll_steps):
        bare()
    # four alternating blocks of each, by the same clock (the boxes of the pool drift by a per cent or two within a second)
    blocks, per = 4, max(1, epochs_timed // 4)
    tl = tb = 0.0
    for _ in range(blocks):
        tl += timed(loop.run_epoch, per)
        tb += timed(bare, per * steps_per_epoch)
    its = blocks * per * steps_per_epoch
    loop_ms, bare_ms = tl / its, tb / its
    loop.check()
    hs.check_overflow()
    aa_loop = int(loop.hs.aa_item_count())
    epochs_timed = blocks * per
    return {"what": f"{label}: {batch} of {num} views x {H}x{W}, V={V}: ms per ITERATION of the product's stage-2 loop (Stage2Loop = run_ham: fresh "
                    "permutation per epoch, views selected inside the step, fused Adam) | the bare step on fixed tensors of the same batch, same clock",
            "loop_ms_per_iteration": round(loop_ms, 4), "bare_step_ms": round(bare_ms, 4), "loop_over_bare": round(loop_ms / bare_ms, 4),
            "iterations_timed": epochs_timed * steps_per_epoch, "aa_items_last_step": aa_loop}


def import_swap_side_measurement(scene_c, dev, views=None, iters=5):
    """Not the headline metric: the import-swap experience of INTEGRATION.md — the stage-2 loop body of mesh_sfs_optim.py:253-308 on the
    generic fmhr_amd.dr / fmhr_amd.utils ops with torch autograd, op by op (fmhr_amd/examples/loop_body.py is the restated body the
    parity test runs), ms per iteration on this workload by the host clock (the path synchronises: torch.where, .item())."""
    from fmhr_amd.examples.loop_body import loop_body_on_dr as _loop_body_on_dr
    n, H, W = views or scene_c["views"], scene_c["H"], scene_c["W"]
    faces = scene_c["faces"].to(dev)
    w2c, proj = scene_c["w2c"][:n].contiguous().to(dev), scene_c["proj"][:n].contiguous().to(dev)
    sc = dict(scene_c, views=n, sh_true=scene_c["sh_true"][:n], sh0=scene_c["sh0"][:n], w2c=scene_c["w2c"][:n], proj=scene_c["proj"][:n])
    tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
    g = dict(vertices=sc["vertices"].to(dev), delta=torch.zeros_like(sc["vertices"]).to(dev), albedo=sc["albedo0"].to(dev), sh0=sc["sh0"].to(dev),
             w2c=w2c, proj=proj, img=tg["img"], mask=tg["mask"], valid_mask=tg["valid_mask"])
    elm = edge_length_mean(g["vertices"], faces)
    for _ in range(2):
        _loop_body_on_dr(g, faces, n, H, W, sc["weights"], elm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        _loop_body_on_dr(g, faces, n, H, W, sc["weights"], elm)
    torch.cuda.synchronize()
    return {"what": f"mesh_sfs_optim.py:253-308 op by op on fmhr_amd.dr + fmhr_amd.utils with torch autograd (the two-line import swap), {n} views x {H}x{W}",
            "import_swap_ms": round(1e3 * (time.perf_counter() - t0) / iters, 2), "iterations": iters}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # 0.1 s of device time at ~0.5 ms per step
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="metric", choices=list(synthetic.CONFIGS))
    ap.add_argument("--views", type=int, default=None, help="override the view count of the config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage1", action="store_true", help="skip the stage-1 side measurement (N = 1 only; after the timed region)")
    ap.add_argument("--no-run-length", action="store_true", help="skip the reference-length side measurement (N = 1 only)")
    ap.add_argument("--run-length", type=int, default=600, help="steps of the reference-length side measurement: the demo run is 300 "
                    "epochs x 2 batches of stage 2 (conf/demo_sfs.conf:9,24; mesh_sfs_optim.py:247-252)")
    ap.add_argument("--no-run-ham", action="store_true", help="skip the run_ham stage-2 loop side measurement (N = 1 only)")
    ap.add_argument("--cpu-views", type=int, default=48)   # the whole 48-view batch: 5 x ~2.6 s of CPU work, nothing extrapolated
    ap.add_argument("--preroll", type=int, default=320, help="upper bound of the untimed pre-roll (stops earlier once the antialias "
                    "work list is stationary; at least 128 steps; 0 = none)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--frame-streams", type=int, default=int(os.environ.get("FMHR_FRAME_STREAMS", "2")),
                    help="HIP streams the local frames of a sequence capture are spread over (independent problems: their kernels may "
                         "overlap: C5 on one GPU 364 -> 392 it/s with 2, 377 with 4); 1 = one after the other on the current stream")
    args = ap.parse_args()

    rank, world, local = fdist.init_from_env()
    # FMHR_FORCE_SHARDED=1 (rehearsal): one rank runs the view-sharded step — split backward, all-reduce over the real backend, combine
    sharded = world > 1 or os.environ.get("FMHR_FORCE_SHARDED", "0") == "1"
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(args.seed)

    scene = synthetic.scene_from_config(args.config, args.seed, args.views)
    n_total, H, W, V, F = scene["views"], scene["H"], scene["W"], scene["V"], scene["F"]
    # partition: the views of ONE problem over the ranks (metric, C1-C4), or — a sequence capture (C5: 8 frames x 32 views, every
    # frame an independent (delta, albedo) problem on the shared topology) — frames first, then views (fmhr_amd.dist.shard_frames_views)
    n_frames = int(os.environ.get("FMHR_DUP_FRAMES", scene.get("frames", 1)))       # FMHR_DUP_FRAMES: experiments only
    if n_frames > 1:
        part = fdist.shard_frames_views(n_frames, n_total, rank, world)
        fgroups = fdist.frame_groups(n_frames, world) if world > 1 else [None] * n_frames          # (one rank holds whole frames: no exchange)
        (f0, f1), (s0, s1) = part["frames"], part["views"]
        view_rank, view_world = part["view_rank"], part["view_world"]
    else:
        f0, f1 = 0, 1
        s0, s1 = fdist.shard_views(n_total, rank, world)
        # the step's one all-reduce goes straight into RCCL on the step's own stream where that communicator can be had (two cross-stream
        # hand-overs of torch's nccl backend = 14.6 us per call saved); otherwise the torch group itself
        fgroups = [fdist.make_stream_all_reduce(torch.distributed.group.WORLD, numel=12 * V + 8) if sharded else None]
        view_rank, view_world = rank, world
    n = s1 - s0
    # LOOP MODE (--config demo): a step is one MINIBATCH of the resident capture, drawn by the product's own loop (fmhr_amd.pipeline.Stage2Loop:
    # a fresh permutation per epoch, views selected inside the step — ABI 6 — and, on N ranks, the batch sharded over them): every rank
    # holds the whole capture, `n` is this rank's share of a batch
    loop_batch = synthetic.LOOP_BATCH.get(args.config) if (n_frames == 1 and args.views is None) else None
    if loop_batch:
        s0, s1 = 0, n_total
        n = fdist.shard_batch(loop_batch, V, rank, world)["n_local"]
    if n == 0:
        raise SystemExit("more ranks than views")
    # The optimisation runs on the product's renumbering of the VERTICES (fmhr_amd.ham.VertexOrder: first-touch order of the face list;
    # same triangles in the same face order, `rast` bit-identical, results mapped back with order.to_caller) — exactly what
    # fmhr_amd.pipeline.run_ham does with a caller's mesh, applied once here to everything the scene holds per vertex.
    # FMHR_VERTEX_ORDER=0: the subdivision's own numbering (2 880 against 2 930 it/s at the metric config).
    scene_caller = dict(scene)                          # (the tensors as generated: the side measurement in the mesh's own numbering)
    order = VertexOrder(scene["faces"], V)
    for k in ("vertices", "vertices_true", "albedo_true", "albedo0"):
        scene[k] = order.to_internal(scene[k])
    scene["faces"] = order.faces
    faces = scene["faces"].to(dev)
    w2c, proj = scene["w2c"][s0:s1].contiguous().to(dev), scene["proj"][s0:s1].contiguous().to(dev)
    sh = scene["sh0"][s0:s1].contiguous().to(dev)
    vtmp = scene["vertices"].to(dev).contiguous()
    elm = edge_length_mean(vtmp, faces)
    lr = scene["lr"]
    from fmhr_amd.optim import Adam          # same update rule as torch.optim.Adam (tests/test_gpu_utils.py::test_adam)
    reg_range = fdist.shard_views(V, view_rank, view_world)     # the view-independent terms are split over the ranks by vertex range
    if os.environ.get("FMHR_BENCH_REG_SHARE"):                  # experiments (tools/r6_tail_share.sh): the regulariser share of a rank of a K-rank run
        k_ = int(os.environ["FMHR_BENCH_REG_SHARE"])            # on ONE rank: K = 0 none at all, K = 8 an eighth of the vertices, K = 1 all
        reg_range = (0, 0) if k_ == 0 else fdist.shard_views(V, 0, k_)

    bench_bin_cap = [0]                  # entries per fragment bin of the workspaces below (0: the library's default, 512); grown after the pre-roll if it overflowed

    class Problem:
        """one (delta, albedo) optimisation: its targets, its step workspace, its optimiser state"""

        def __init__(self, frame):
            sc = dict(scene, views=n, sh_true=scene["sh_true"][s0:s1], sh0=scene["sh0"][s0:s1])
            if frame > 0:                      # another frame of the sequence: its own target geometry (seeded), same cameras / topology
                g = torch.Generator().manual_seed(1000 + frame)
                sc["vertices_true"] = scene["vertices"] + 0.0005 * torch.randn(scene["vertices"].shape, generator=g)
            self.tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
            self.hs = HamStage2(faces, V, n, H, W, scene["weights"], elm, n_global=n_total, bin_cap=bench_bin_cap[0])
            self.delta = torch.zeros(V, 3, device=dev, requires_grad=True)
            self.albedo = scene["albedo0"][None].to(dev).clone().requires_grad_(True)          # [1,V,3] as in the reference (:176)
            self.opt = Adam([{'params': self.delta, 'lr': lr["lr"]}, {'params': self.albedo, 'lr': lr["albedo_lr"]}])
            self.group = fgroups[frame]

        def step(self):
            # the Adam update of (delta, albedo) is applied by the kernel that forms their final gradients (the mesh tail; after the
            # step's one all-reduce the combine kernel): same rule as opt.step(), no launch of its own
            tg = self.tg
            losses, gd, ga = self.hs.step(vtmp, self.delta.detach(), self.albedo.detach()[0], sh, w2c, proj, tg["img"], tg["mask"],
                                          tg["valid_mask"], group=self.group, reg_range=reg_range, optimizer=self.opt,
                                          params=(self.delta, self.albedo))
            self.delta.grad, self.albedo.grad = gd, ga.view(1, V, 3)
            return losses

    class LoopProblem:
        """LOOP MODE: the same optimisation stepped by the product's loop on minibatches of the resident capture (fmhr_amd.pipeline.Stage2Loop)"""

        def __init__(self, frame):
            from fmhr_amd.pipeline import Stage2Loop
            self.tg = tgt = synthetic.make_targets(scene, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
            self.delta = torch.zeros(V, 3, device=dev, requires_grad=True)
            self.albedo = scene["albedo0"][None].to(dev).clone().requires_grad_(True)
            self.opt = Adam([{'params': self.delta, 'lr': lr["lr"]}, {'params': self.albedo, 'lr': lr["albedo_lr"]}])
            self.group = fgroups[frame]
            self.loop = Stage2Loop(vtmp, faces, self.delta, self.albedo, sh, tgt["img"], tgt["mask"], tgt["valid_mask"], w2c, proj, (H, W),
                                   scene["weights"], elm, self.opt, loop_batch, 1 << 30, torch.Generator(device="cpu").manual_seed(args.seed),
                                   check_every=0,      # (no polls inside the timed region: the flags and the exchange are checked after it)
                                   group=torch.distributed.group.WORLD if sharded else None, exchange=self.group, bin_cap=bench_bin_cap[0])
            self.loop.step_next()                                # the first step builds the workspace of this rank's batch share

        @property
        def hs(self):
            return self.loop.hs

        def step(self):
            self.loop.step_next()
            return self.loop.hs.losses

    problems = [(LoopProblem if loop_batch else Problem)(f) for f in range(f0, f1)]
    hs, tg, delta, albedo, opt, group = (problems[0].hs, problems[0].tg, problems[0].delta, problems[0].albedo, problems[0].opt,
                                          problems[0].group)
    nf_local = len(problems)
    bin_cap_grown = []

    n_fs = max(1, min(args.frame_streams, nf_local))
    fstreams = [torch.cuda.Stream(device=dev) for _ in range(n_fs)] if n_fs > 1 else []

    def step():
        if not fstreams:
            for p in problems:                 # the local frames one after the other (one frame: the plain stage-2 iteration)
                losses = p.step()
            return losses
        # independent frames on side streams, forked from and joined to the current stream
        cur = torch.cuda.current_stream()
        for s in fstreams:
            s.wait_stream(cur)
        for i, p in enumerate(problems):
            with torch.cuda.stream(fstreams[i % n_fs]):
                losses = p.step()
        for s in fstreams:
            cur.wait_stream(s)
        return losses

    def step_parts():
        """the view-sharded step with HIP events around its exchange: (kernels, all-reduce, combine + Adam) in ms — untimed
        profiling pass only (N > 1)."""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        a = (vtmp, delta.detach(), albedo.detach()[0], sh, w2c, proj)
        vi, rr = None, reg_range
        if loop_batch:                                           # LOOP MODE: this rank's share of the running epoch's first batch
            lp = problems[0].loop
            part = fdist.shard_batch(loop_batch, V, rank, world)
            perm = lp._cur if lp._cur is not None else lp.perm
            vi, rr = perm[part["views"][0]: part["views"][1]], part["reg_range"]
        ev[0].record(stream)
        hs.forward(*a, tg["mask"], tg["valid_mask"], view_idx=vi)
        hs.backward(*a, tg["img"], tg["mask"], tg["valid_mask"], reg_range=rr, split=True, view_idx=vi)
        ev[1].record(stream)
        if hasattr(group, "all_reduce_"):
            group.all_reduce_(hs.bucket)
        else:
            torch.distributed.all_reduce(hs.bucket, group=group)
        ev[2].record(stream)
        hs.combine((opt.fused_param(delta), opt.fused_param(albedo)))
        ev[3].record(stream)
        torch.cuda.synchronize()
        return [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]

    def barrier():
        if sharded:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    stream = torch.cuda.current_stream()
    # untimed pre-roll to the stationary workload: at least 128 optimiser steps, then until the antialias work-item count of a
    # 32-step block is within 0.5 % of the previous block's (every rank runs the same number of steps: the criterion is evaluated
    # on the all-reduced count), at most --preroll steps
    while True:
        preroll, aa_hist = 0, []
        while preroll < args.preroll:
            for _ in range(32):
                step()
            preroll += 32
            cnt = sum(p.hs.aa_item_count() for p in problems).to(torch.float32).reshape(1)
            if sharded:
                torch.distributed.all_reduce(cnt)
            aa_hist.append(float(cnt))
            if preroll >= 128 and len(aa_hist) >= 2 and abs(aa_hist[-1] - aa_hist[-2]) <= 0.005 * max(aa_hist[-1], 1.0):
                break
        # the product's policy for the fragment bins (fmhr_amd.pipeline.run_ham): the default capacity follows the measured depth — if the
        # pre-roll overflowed it on ANY rank, every rank rebuilds its workspaces with room for the fullest bin (x 1.25, at least 2 x) and
        # pre-rolls again from fresh parameters; the timed region then runs on workspaces that hold the workload
        need = torch.tensor([max([p.hs.bin_depth_max() if p.hs.raster_overflowed() else 0 for p in problems] + [0])], dtype=torch.int64, device=dev)
        if sharded:
            torch.distributed.all_reduce(need, op=torch.distributed.ReduceOp.MAX)
        if int(need) == 0 or len(bin_cap_grown) >= 3:
            break
        bench_bin_cap[0] = max(2 * (bench_bin_cap[0] or 512), (int(1.25 * int(need)) + 63) // 64 * 64)
        bin_cap_grown.append(bench_bin_cap[0])
        del problems, hs, tg, delta, albedo, opt
        torch.cuda.empty_cache()
        problems = [(LoopProblem if loop_batch else Problem)(f) for f in range(f0, f1)]
        hs, tg, delta, albedo, opt, group = (problems[0].hs, problems[0].tg, problems[0].delta, problems[0].albedo, problems[0].opt,
                                              problems[0].group)
    import gc
    gc.collect()
    gc.disable()
    for _ in range(args.warmup):
        step()
    barrier()

    # untimed profiling pass: per-launch-group HIP-event times -> dominant group
    prof_steps = 9
    tm = Timing(prof_steps, stream)
    for i in range(prof_steps):
        tm.attach(i)
        step()
    Timing.detach()
    torch.cuda.synchronize()
    # MEDIAN over the instrumented steps: with an event pair around every launch group a step now and then shows one group at
    # 1.5-2x (k_face_bwd 73 -> 150 us), which a 3-step mean turned into a different "dominant kernel" from run to run
    kernels = {name: sorted(tm.ms(i, k) for i in range(prof_steps))[prof_steps // 2] for k, name in enumerate(SLOTS)}
    collective = None
    if group is not None:
        parts = torch.tensor([step_parts() for _ in range(5)][1:], dtype=torch.float64).mean(0).to(dev)     # [kernels, all-reduce, combine]
        allp = [torch.zeros_like(parts) for _ in range(world)]
        torch.distributed.all_gather(allp, parts)
        collective = {"per_rank_ms": {"kernels": [round(float(p[0]), 4) for p in allp], "all_reduce": [round(float(p[1]), 4) for p in allp],
                                      "combine_adam": [round(float(p[2]), 4) for p in allp]},
                      "bytes": int(hs.bucket.numel() * 4), "note": "all_reduce includes the wait for the slowest rank's kernels",
                      "path": ("one-shot peer-to-peer all-reduce over the direct links (fmhr_peer_all_reduce) on the step's own stream" if type(group).__name__ == "PeerAllReduce" else
                               "ncclAllReduce (RCCL) on the step's own stream" if hasattr(group, "all_reduce_") else
                               "torch.distributed.all_reduce (the backend's own stream)"),
                      "path_negotiation": list(fdist.make_stream_all_reduce.last)}
    P = H * W
    # dominant launch group: the longest one.  Three kernels are within ~20 % of each other (k_raster_tris 84-88, the second rasterizer
    # pass 78, k_face_bwd 70-73 us by rocprofv3; until v89 k_face_bwd moved between 74 and 92 us with the number of sliver triangles the
    # pre-roll had grown): groups within 35 % of the longest count as tied and the tie goes to the earliest in the
    # pipeline, so that the reported kernel is the one the committed rocprofv3 summary shows as the longest (k_raster_tris) in every
    # run; all candidates are listed in the line
    cand = [k for k in range(len(SLOTS)) if alg_bytes(SLOTS[k], n, P, V, F, hs.nnz) > 0]
    tmax = max(kernels[SLOTS[k]] for k in cand)
    dom = min(k for k in cand if kernels[SLOTS[k]] >= 0.65 * tmax)

    # timed region: EXACTLY K steps between barrier+synchronize pairs; HIP events on the step's own stream bracket the
    # dominant launch group (recorded asynchronously by the library) and the whole step — on every `stride`-th step of the timed
    # region only: an event record costs ~5.7 us of device time (the kernel trace shows exactly that gap on either side of an
    # instrumented group), four of them per step were 3.8 % of a 0.6 ms step
    K = args.steps
    # (round 4: every 8th step carried the four event records = 2.5-3.5 us per step of the headline, 0.75-1 %; now K // 25 steps do,
    # at least two: 2 of the driver's 20, 8 of 200)
    stride = 1 if K <= 2 else max(1, K // max(2, K // 25))
    sampled = list(range(0, K, stride))[:max(2, K // 25)] if K > 2 else list(range(K))
    slot_of = {i: j for j, i in enumerate(sampled)}
    tk = Timing(len(sampled), stream)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in sampled]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in sampled]
    # no cyclic garbage collection inside the timed region: with torch loaded a full collection is a ~40 ms host pause, and whether
    # one falls into the region depended on how many event objects had been created before it (C4 with --steps 60: 1.85 ms per
    # step by the host clock, 1.15 ms on the device for every single step; --steps 100: 1.17)
    # (collected and switched off BEFORE the warm-up steps: a collection right before the region leaves the interpreter's working set
    # cold, and with 20 steps = 7 ms of device time the first step's enqueue latency is a visible share of the host clock)
    barrier()
    dbg = [] if os.environ.get("FMHR_BENCH_DEBUG") else None
    t0 = time.perf_counter()
    for i in range(K):
        if dbg is not None:
            dbg.append(time.perf_counter() - t0)
        if i in slot_of:
            j = slot_of[i]
            tk.attach(j, dom)
            ev0[j].record(stream)
            losses = step()
            ev1[j].record(stream)
            Timing.detach()
        else:
            losses = step()
    if dbg is not None:
        dbg.append(time.perf_counter() - t0)
    barrier()
    t1 = time.perf_counter()
    gc.enable()
    if dbg is not None and rank == 0:
        print("host time of each step's enqueue (ms since t0):", " ".join("%.2f" % (1e3 * x) for x in dbg), "| barrier at %.2f" % (1e3 * (t1 - t0)), file=sys.stderr)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if sharded:
        torch.distributed.all_reduce(elapsed, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(elapsed)
    # the launch duration of the dominant group = the RAW mean of its HIP-event pairs over the timed region (it agrees with
    # rocprofv3's kernel duration — 86.2 vs 85.1 us — while the "empty event pair" correction round 3 subtracted put it BELOW the
    # profiler's figure).  What an empty pair reads (the launch groups that are empty on this workload) is still reported, as
    # information only
    dom_ms = dom_ms_raw = sum(tk.ms(j, dom) for j in range(len(sampled))) / len(sampled)
    empty = sorted(kernels[s_] for s_ in ("raster_clear", "raster_large", "transform_bwd", "normals_bwd") if kernels[s_] < 0.008)
    ev_overhead = empty[len(empty) // 2] if empty else 0.0
    step_dev_ms = sum(ev0[j].elapsed_time(ev1[j]) for j in range(len(sampled))) / len(sampled)
    # a full antialias work-item list (some silhouette pairs without a gradient) invalidates the run — on EVERY rank
    ovf = torch.tensor([1.0 if any(p.hs.aa_overflowed() or p.hs.raster_overflowed() for p in problems) else 0.0], device=dev)
    if sharded:
        torch.distributed.all_reduce(ovf)
    if float(ovf) != 0:
        raise SystemExit(f"bench.py: the antialias work-item list or a rasterizer fragment bin overflowed on {int(float(ovf))} rank(s) (aa_cap={hs.dims.aa_cap})")
    # ... and so does a dead or diverged exchange (it would read as a FASTER number): after the timed region the
    # replicated parameters of every problem are fingerprinted across the ranks that share it and the exchange's own failure state is read
    # (fmhr_amd.dist.ExchangeMonitor) — ExchangeError on every rank, exit code != 0, no JSON line
    exchange_ok = None
    if sharded:
        for p_ in problems:
            tg_ = p_.group
            if tg_ is None:
                continue                                         # (a rank that holds whole frames exchanges nothing)
            torch_group = tg_.group if hasattr(tg_, "all_reduce_") else tg_
            mon = fdist.ExchangeMonitor(torch_group if torch_group is not None else torch.distributed.group.WORLD, tg_ if hasattr(tg_, "gave_up") else None, dev)
            exchange_ok = mon.verify([p_.delta, p_.albedo], "bench.py: (delta, albedo) after the timed region")
    final = dict(zip(("sfs", "lap", "albedo", "mask", "edge", "delta", "total", "n_valid"), losses.tolist()))
    aa_items = sum(p.hs.aa_item_count() for p in problems).to(torch.float32).reshape(1)
    if sharded:
        torch.distributed.all_reduce(aa_items)

    if rank == 0:
        # HBM traffic of the launch groups comes from the committed PMC profile of this same command (counters cannot be
        # collected live next to the timed region); null when the workload differs from the profiled one
        traffic = None
        tpath = next((os.path.join(ROOT, "profiles", f) for f in ("traffic_r06.json", "traffic_r05.json", "traffic_r04.json", "traffic_r03.json", "traffic_r02.json")
                      if os.path.exists(os.path.join(ROOT, "profiles", f))), "")
        if os.path.exists(tpath) and args.config == "metric" and world == 1 and args.views is None and hs.binned == traffic_binned(tpath):
            traffic = json.load(open(tpath))
        ms_per_step = elapsed / K * 1e3
        dom_bytes = alg_bytes(SLOTS[dom], n, P, V, F, hs.nnz)
        dom_gbs = dom_bytes / (dom_ms * 1e-3) / 1e9
        sb = nf_local * step_alg_bytes(n, P, V, F, hs.nnz)
        step_gbs = sb / (step_dev_ms * 1e-3) / 1e9
        fb = nf_local * step_fused_bound_bytes(n, P, V, F, hs.nnz)
        fused_gbs = fb / (step_dev_ms * 1e-3) / 1e9
        t_lo, t_hi = (traffic["step_total_low"], traffic["step_total_high"]) if traffic else (None, None)
        out = {
            "metric": "HAM render+backward iters/sec (48 views, 49k verts, 512x512)",
            "value": K / elapsed, "unit": "iterations/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {(str(loop_batch) + ' of ') if loop_batch else ''}{n_total} views x {H}x{W}, synthetic hand mesh V={V} F={F} "
                                   f"({'MANO sub3 counts' if V % 49281 == 0 else 'MANO base counts' if V % 778 == 0 else 'custom'}"
                                   f"{', two hands' if scene.get('hands', 1) == 2 else ''}), "
                                   f"stage-2 step = render + 6 losses + backward + grad all-reduce + Adam, stationary (pre-rolled {preroll} steps)"
                                   + (f"; LOOP MODE: each step is a minibatch of {loop_batch} views drawn by the product's loop (fresh permutation per epoch, views selected inside the step, the batch sharded over the ranks)" if loop_batch else ""),
                       "batch": loop_batch or n_total,
                       "views": n_total, "views_per_gpu": n, "frames": n_frames, "frames_per_gpu": nf_local, "frame_streams": n_fs, "H": H, "W": W, "V": V, "F": F,
                       "rasterizer": "binned (per-segment LDS bins)" if hs.binned else "atomic depth keys + resolve",
                       "vertex_order": ("caller's numbering" if order.identity else
                                        "first-touch renumbering of the vertices by the product (fmhr_amd.ham.VertexOrder, as run_ham applies it; faces unchanged)"),
                       "coverage": round(tg["coverage"], 4), "parallelism": (f"views sharded over {world} GPU(s)" if n_frames == 1 else
                                                                        f"{n_frames} frames x {n_total} views: frames first, then views, over {world} GPU(s)"),
                       "preroll_steps": preroll, "aa_items_per_step": int(float(aa_items)), "aa_items_during_preroll": [int(x) for x in aa_hist],
                       # the fullest 64-pixel segment of any step so far (fragments) against the bins' capacity (workspace span: 8 B x capacity per segment)
                       "bin_depth_max": max(p.hs.bin_depth_max() for p in problems), "bin_cap": int(hs.dims.bin_cap) or 512, "bin_cap_grown": bin_cap_grown, "workspace_bytes": int(hs.ws_bytes)},
            "roofline": {"bound": "hbm", "kernel": SLOTS[dom], "achieved": dom_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": dom_gbs / HBM_PEAK_GBS, "traffic": (traffic or {}).get(SLOTS[dom]),
                         # the same launch priced by its MEASURED HBM bytes (upper estimate of the PMC profile): how busy HBM really was
                         "traffic_gbs": ((traffic or {}).get(SLOTS[dom]) or 0) / (dom_ms * 1e-3) / 1e9 or None,
                         "traffic_frac": ((traffic or {}).get(SLOTS[dom]) or 0) / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS or None,
                         "traffic_source": (traffic or {}).get("_source"), "alg_bytes_per_launch": dom_bytes,
                         "avg_launch_ms": dom_ms, "avg_launch_ms_source": "raw mean of the HIP-event pairs around the launch group over the timed region",
                         "empty_event_pair_ms": ev_overhead,
                         "event_samples": len(sampled),
                         "longest_groups_ms": {SLOTS[k]: round(kernels[SLOTS[k]], 4) for k in cand if kernels[SLOTS[k]] >= 0.65 * tmax}},
            "step_roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "device_ms_per_step": step_dev_ms,
                              # physical: measured HBM bytes of the step (PMC profile; FETCH_SIZE x1 .. x2 + WRITE_SIZE) per device time
                              "traffic_low_high": [t_lo, t_hi] if traffic else None,
                              "traffic_frac": [t_lo / (step_dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, t_hi / (step_dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS] if traffic else None,
                              # physical: the least a fully fused step must move (SURVEY.md §8d lower bound) per device time
                              "fused_bound_bytes_per_step": fb, "fused_bound_gbs": fused_gbs, "fused_bound_frac": fused_gbs / HBM_PEAK_GBS,
                              # NOT a roofline fraction: the reference's op-by-op primitive chain priced at this step's time
                              "op_granular_equivalent": {"alg_bytes_per_step": sb, "gbs": step_gbs, "x_of_peak": step_gbs / HBM_PEAK_GBS}},
            "kernels_ms": {k: round(v, 4) for k, v in kernels.items()},
            # every launch group against the same yardstick (untimed profiling pass): algorithmic GB/s and, where the committed PMC
            # profile has it, the measured HBM bytes of the launch
            "groups": {k: {"ms": round(v, 4), "alg_gbs": round(alg_bytes(k, n, P, V, F, hs.nnz) / max(v, 1e-9) / 1e6, 1),
                           "hbm_bytes": (traffic or {}).get(k)} for k, v in kernels.items() if alg_bytes(k, n, P, V, F, hs.nnz) > 0},
            "losses": {k: (round(v, 6) if isinstance(v, float) else v) for k, v in final.items()},
        }
        if collective is not None:
            # the cross-rank check after the timed region: every rank holds the same bits of (delta, albedo), no exchange gave up
            collective["exchange_ok"] = bool(exchange_ok and exchange_ok["ok"])
            collective["exchange_check"] = exchange_ok
            out["collective"] = collective
        if loop_batch:                                          # LOOP MODE times the product's loop itself: the side measurements of the fixed-batch
            args.no_run_length = True                            # workload (run length, numbering, run_ham's loop against the bare step, import swap) do not apply
        if world == 1 and nf_local == 1 and not args.no_run_length:
            try:
                out["run_length"] = run_length_side_measurement(Problem, stream, args.run_length)
            except Exception as e:  # noqa: BLE001
                out["run_length"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and nf_local == 1 and not args.no_run_length and not order.identity:
            try:                                        # both numberings by the SAME clock (HIP events around 100 pre-rolled steps)
                out["vertex_numbering"] = {"as_arrived": caller_numbering_side_measurement(scene_caller, dev, stream),
                                           "first_touch (the timed workload)": {k: v for k, v in caller_numbering_side_measurement(scene, dev, stream).items() if k != "what"}}
            except Exception as e:  # noqa: BLE001
                out["vertex_numbering"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and nf_local == 1 and not args.no_run_ham and not args.no_run_length:     # (--no-run-length: the lean form the A/B and profile scripts use)
            try:                                        # the loop the product runs, at the metric batch and at the reference's demo batch
                rh = {"metric": run_ham_stage2_side_measurement(scene_caller, n_total, dev, stream, 100, label=args.config)}
                if args.config == "metric":
                    demo = synthetic.make_scene(16, 1024, 1280, "mano", 0, 3, seed=args.seed, hands=2)
                    rh["demo_conf"] = run_ham_stage2_side_measurement(demo, 8, dev, stream, 30, preroll_steps=96,
                                                                      label="conf/demo_sfs.conf:5-7,24 (num 16, batch 8, 1280x1024, two hands)")
                out["run_ham_stage2"] = rh
            except Exception as e:  # noqa: BLE001
                out["run_ham_stage2"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and nf_local == 1 and not args.no_run_ham and not args.no_run_length:
            try:
                out["import_swap"] = import_swap_side_measurement(scene_caller, dev)
            except Exception as e:  # noqa: BLE001
                out["import_swap"] = {"error": f"{type(e).__name__}: {e}"}
        # the two side measurements must never cost the headline line: a failure is reported in their place
        if world == 1 and not args.no_stage1:
            try:
                out["stage1"] = stage1_side_measurement(scene, faces, w2c, proj, sh, tg, vtmp, H, W, dev)
            except Exception as e:  # noqa: BLE001
                out["stage1"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(scene, max(1, min(args.cpu_views, loop_batch or n_total)), args.seed, views_per_step=loop_batch)
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out))
    if sharded:
        for g_ in fgroups:
            if hasattr(g_, "close"):
                g_.close()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
