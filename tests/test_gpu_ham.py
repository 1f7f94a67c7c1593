"""GPU parity of the fused stage-2 step (fmhr_ham_forward/backward through fmhr_amd.ham.HamStage2) against the oracle's
restatement of mesh_sfs_optim.py:253-306 (autograd over the oracle ops), plus the op-by-op `dr` composition (the unchanged
reference loop body running on fmhr_amd.dr / fmhr_amd.utils)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import log_parity, mesh_scene, rel_err
from checker_scenes import oracle_forward_fn, prepare              # noqa: F401  (re-exported: other test modules import them from here)
import checker_scenes
from fmhr_amd.examples.loop_body import loop_body_on_dr as _loop_body_on_dr

pytestmark = pytest.mark.gpu
TOL = 1e-4

# The gradient bar is a HARD 1e-4 (per-tensor max|a-b| / max|b| against fp64 values on the fp32 decisions) for EVERY case (round 5).
# Round 4 had two cases under a relaxed bar ("twice the fp32 twin"): the two-hand sub3 mesh at 512 x 334 (albedo gradient 4.0e-4) and
# the reference capture's own cameras at 1280 x 1024 (position gradient 9.9e-4) — rule R5's pixel-relative edge functions x - fx w,
# evaluated in fp32, lose ~10 bits to cancellation at the capture's clip-space scale and a thin triangle divides the rest by its area;
# the fp32 twin showed the same (8.2e-4).  Since round 5 the VALUES of rule R5 — the barycentrics a winner returns and their gradient
# with respect to the clip positions — are evaluated in double from the fp32 clip positions (the decisions — coverage,
# accept / reject, z/w and the depth order — stay fp32 and bit-identical): twin 2.3e-5 / 1.3e-5 on those scenes
# (tools/experiments/r05_r5_wide.py), and RELAXED is empty.  A case may only be listed here with its measured reason, and only while
# the fp32 twin itself is beyond 1e-4.  The measured numbers of every case are written to profiles/parity_r05.txt.
RELAXED = {}


def grad_bar(name, e_d, e_a, o_d, o_a, t_d=None, t_a=None):
    """e_*: fused step vs fp64 on the fp32 decisions; o_*: the oracle's fp32 twin vs the same; t_*: fused step vs the fp32 twin"""
    twin = "" if t_d is None else f" | fused vs fp32 twin: {t_d:.2e} {t_a:.2e}"
    log_parity(f"{name}: fused vs fp64-on-fp32-decisions: gdelta {e_d:.2e} galbedo {e_a:.2e} | fp32 oracle twin vs the same: "
               f"{o_d:.2e} {o_a:.2e}{twin} | bar {'2x twin (' + RELAXED[name] + ')' if name in RELAXED else '1e-4 hard'}")
    if name in RELAXED:
        assert o_d > TOL or o_a > TOL, f"{name}: listed as relaxed but the fp32 twin meets 1e-4: remove it from RELAXED"
        assert e_d < max(TOL, 2 * o_d) and e_a < max(TOL, 2 * o_a), (name, e_d, e_a, o_d, o_a)
    else:
        assert e_d < TOL and e_a < TOL, (name, e_d, e_a, o_d, o_a)


def gpu_loop_decisions(ref32, tmp_img_gpu, img, mask, name=""):
    """oracle/checker_scenes.gpu_loop_decisions, logging into the round's parity file"""
    return checker_scenes.gpu_loop_decisions(ref32, tmp_img_gpu, img, mask, name, log=log_parity)


# "views19/30/37/41": more views than one lane group holds at 1 view per lane -> 3, 4, 5 and 6 views per lane with ragged last slots
CASES = [("C1", (1, 256, 256, 24, 32), 0), ("C2x3", (3, 512, 334, 24, 32), 1), ("micro", (2, 192, 192, 64, 96), 2),
         ("views19", (19, 96, 80, 12, 16), 3), ("views30", (30, 64, 48, 10, 14), 4), ("views41", (41, 64, 48, 10, 14), 5),
         ("views37", (37, 64, 48, 10, 14), 15),
         ("ragged", (6, 50, 38, 10, 14), 6),       # H*W = 1900: the 64-pixel segments straddle view boundaries
         ("views53", (53, 48, 40, 10, 14), 7),     # more than 48 views: two view groups (two partial planes for the mesh tail)
         ("coarse50", (50, 96, 96, 4, 6), 8),      # every triangle a large-box one, in two view groups (queue + k_face_bwd_large)
         # BASELINE.json configs[4] in small: TWO hands = two connected components with a boundary loop each, subdivided per hand and
         # concatenated as mesh_sfs_optim.py:76-87 does, W = 334 (conf/ih_sfsseq.conf:11), more than 48 views (two view groups)
         ("two_hands50", (50, 128, 334, 10, 14, 1, 2), 9),
         ("two_hands_mano", (3, 512, 334, "mano", 0, 1, 2), 10),
         # BASELINE.json configs[0]: ONE view, 256 x 256, the 778-vertex / 1538-face base mesh, silhouette-only loss (every other
         # weight 0: the gradient reaches the vertices through the antialias of the mask alone)
         ("silhouette_only_C1", (1, 256, 256, "mano", 0, 0), 13)]


def binned_form(hs):
    """hs.binned — except that a suite run under FMHR_RASTER_BINNED=0 (the whole suite through the atomic-key form: same-library validation of
    the other rasterizer) is not held to the tests' expectation of the binned one"""
    return hs.binned or os.environ.get("FMHR_RASTER_BINNED") == "0"


def expect_layout(hs, flat=False):
    """ABI 5: the step's per-pixel buffers have a row pitch of W rounded up to 64 unless the ABI-4 layout (pitch = W) is asked for,
    and the binned rasterizer runs exactly where the mesh is a micro-triangle one (H*W <= 16 F) and the pitch a multiple of 64 —
    W = 334 (conf/ih_sfs.conf:14, conf/ih_sfsseq.conf:11) included."""
    assert hs.pitch == (hs.W if flat else (hs.W + 63) // 64 * 64), (hs.pitch, hs.W)
    micro = hs.H * hs.W <= 16 * hs.F
    if os.environ.get("FMHR_RASTER_BINNED") == "0":
        assert not hs.binned
        return
    assert hs.binned == (micro and hs.pitch % 64 == 0 and hs.F < (1 << 22)), (hs.binned, hs.H, hs.W, hs.F, hs.pitch)


# the ABI-4 layout (pitch = W: segments that straddle rows and views, atomic-key rasterizer, per-pixel antialias scan) stays
# selectable for A/B runs (FMHR_ROW_PITCH=0 / HamStage2(pitch=W)): these cases keep it under the same oracle bar
FLAT_CASES = ("C2x3", "ragged", "two_hands50", "views19")


@pytest.mark.parametrize("case", CASES + [(c[0] + "_flat",) + c[1:] for c in CASES if c[0] in FLAT_CASES], ids=lambda c: c[0])
def test_fused_step_vs_oracle(oracle, case, monkeypatch):
    from fmhr_amd.ham import HamStage2, LOSS_NAMES
    name, dims, seed = case
    flat = name.endswith("_flat")
    if flat:
        monkeypatch.setenv("FMHR_ROW_PITCH", "0")
        name = name[:-5]
    sc = prepare(oracle, dims, seed)
    if name.startswith("silhouette_only"):
        sc["weights"] = dict(sfs=0.0, lap=0.0, albedo=0.0, mask=1000.0, edge=0.0, delta=0.0)
        sc["delta"] = sc["delta"] * 10.0                     # move the contour off the start silhouette: a non-zero MSE
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    hs = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
    expect_layout(hs, flat)
    losses, gdelta, galbedo = hs.step(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"],
                                      g["valid_mask"])
    ld = hs.loss_dict()
    v = hs.views()
    pos_gpu = v["pos"].cpu()

    def run_oracle(dtype, decisions=None):
        # the oracle is evaluated AT the GPU's clip positions (its own einsum differs from the kernel in the last bit, which
        # would move snapped vertices); they are first checked to agree to 1e-6
        c = lambda t: t.to(dtype)
        d = c(sc["delta"]).clone().requires_grad_(True)
        a = c(sc["albedo"])[None].clone().requires_grad_(True)
        ref = oracle.ham_stage2_losses(c(sc["vertices"]), d, a, c(sc["sh0"]), c(sc["w2c"]), c(sc["proj"]), c(sc["img"]),
                                       c(sc["mask"]), c(sc["valid_mask"]), sc["faces"], (sc["H"], sc["W"]), sc["weights"],
                                       sc["elm"], pos_value=pos_gpu, decisions=decisions)
        ref["loss"].backward()
        return ref, d.grad, a.grad[0]

    ref, gd32, ga32 = run_oracle(torch.float32)             # fp32 twin: decisions must match bit for bit
    assert int(v["aa_counter"][1]) == 0, "antialias queue overflow"
    pos_cpu = oracle.clip_positions(sc["vertices"] + sc["delta"], sc["w2c"], sc["proj"])[1]
    assert rel_err(pos_gpu, pos_cpu) < 1e-6
    # ... and what the last-bit difference between the kernel's FMA chain and torch's einsum does to the DECISIONS' inputs: the
    # snapped vertices (rule R2, 1/256 px) of the two sets of positions differ at a handful of vertices and by one unit at most —
    # a transform defect that moved a vertex further would fail here, not be absorbed by evaluating the oracle at the GPU's positions
    def snap(p, n_):
        p = p.to(torch.float32)
        rw = 1.0 / p[..., 3]
        qx = (p[..., 0] * rw) * (n_[1] * 128.0) + (n_[1] * 128.0)
        qy = (p[..., 1] * rw) * (n_[0] * 128.0) + (n_[0] * 128.0)
        return torch.stack([torch.round(qx), torch.round(qy)], -1)
    ds = (snap(pos_gpu, (sc["H"], sc["W"])) - snap(pos_cpu, (sc["H"], sc["W"]))).abs()
    assert float(ds.max()) <= 1.0 and float((ds > 0).float().mean()) < 2e-3, (float(ds.max()), float((ds > 0).float().mean()))
    assert torch.equal(v["rast"].cpu(), ref["rast"]), "rast must be bit-identical to the fp32 oracle"
    assert ld["n_valid"] == ref["n_valid"] and ref["n_valid"] > 100
    with oracle.f32_decisions():                            # fp64 values on the same decisions: the truth for 1e-4
        ref64, gd64, ga64 = run_oracle(torch.float64, gpu_loop_decisions(ref, v["tmp_img"], sc["img"], sc["mask"], name))
    assert ref64["n_valid"] == ref["n_valid"]
    # images: tight against the fp32 twin; against fp64 only loosely — fp32 barycentrics of edge-on (near-degenerate)
    # silhouette triangles are ill-conditioned in ANY fp32 implementation of the formula (the oracle twin shows the same)
    assert rel_err(v["normals"], ref64["normals"][0]) < TOL
    assert rel_err(v["tmp_img"], ref["tmp_img"]) < 1e-5 and rel_err(v["pred_mask"], ref["pred_mask"]) < 1e-5
    assert rel_err(v["tmp_img"], ref64["tmp_img"]) < 5e-3 and rel_err(v["pred_mask"], ref64["pred_mask"]) < 5e-3
    for k in ("sfs", "lap", "albedo", "mask", "edge", "delta"):
        assert abs(ld[k] - float(ref64[k])) <= TOL * max(abs(float(ref64[k])), 1e-6), (k, ld[k], float(ref64[k]))
    assert abs(ld["total"] - float(ref64["loss"])) <= TOL * abs(float(ref64["loss"]))
    e_d, e_a = rel_err(gdelta, gd64), rel_err(galbedo, ga64)
    o_d, o_a = rel_err(gd32, gd64), rel_err(ga32, ga64)
    print(f"{name}: grad rel err vs fp64 oracle: delta {e_d:.2e} albedo {e_a:.2e}; fp32 oracle's own: {o_d:.2e} {o_a:.2e}; "
          f"vs fp32 oracle: {rel_err(gdelta, gd32):.2e} {rel_err(galbedo, ga32):.2e}")
    name = name + (" [flat ABI-4 layout]" if flat else f" [pitch {hs.pitch}, {'binned' if hs.binned else 'atomic keys'}]")
    log_parity(f"{name}: losses vs fp64: " + " ".join(f"{k} {abs(ld[k] - float(ref64[k])) / max(abs(float(ref64[k])), 1e-6):.1e}"
                                                       for k in ("sfs", "lap", "albedo", "mask", "edge", "delta"))
               + f" | tmp_img vs fp32 twin {rel_err(v['tmp_img'], ref['tmp_img']):.1e} vs fp64 {rel_err(v['tmp_img'], ref64['tmp_img']):.1e}"
               + f" | n_valid {ld['n_valid']:.0f} rast bit-identical")
    grad_bar(name, e_d, e_a, o_d, o_a, rel_err(gdelta, gd32), rel_err(galbedo, ga32))
    # run-to-run spread of the float-atomic accumulation stays far inside the tolerance
    g1 = gdelta.clone()
    hs.step(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    assert rel_err(hs.gdelta, g1) < 5e-5


@pytest.mark.parametrize("case", [CASES[2], CASES[6], CASES[8]], ids=lambda c: c[0])
def test_split_backward_combines_to_the_full_gradient(oracle, case):
    """The view-sharded form of the backward on one process: photometric share without its L1 divisor (P) + the rest (R), then
    fmhr_ham_combine — must give the one-rank gradient and loss vector (the all-reduce in between is a sum of such buckets)."""
    from fmhr_amd.ham import HamStage2
    name, dims, seed = case
    sc = prepare(oracle, dims, seed)
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    hs = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
    args = (g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    hs.step(*args)
    ref_l, ref_d, ref_a = hs.losses.clone(), hs.gdelta.clone(), hs.galbedo.clone()
    hs.forward(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["mask"], g["valid_mask"])
    hs.backward(*args, split=True)
    p_d, r_d = hs.gdelta.clone(), hs.gdelta_rest.clone()
    assert float(p_d.abs().max()) > 0 and float(r_d.abs().max()) > 0
    assert float(hs.losses[7]) == float(ref_l[7])                        # the local count
    hs.combine()
    assert rel_err(hs.gdelta, ref_d) < 2e-5 and rel_err(hs.galbedo, ref_a) < 2e-5, name
    assert rel_err(hs.losses[:7], ref_l[:7]) < 1e-5 and float(hs.losses[7]) == float(ref_l[7])
    # two "ranks" = the same bucket twice: the sum must give the same gradient direction with doubled count / doubled R
    hs.forward(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["mask"], g["valid_mask"])
    hs.backward(*args, split=True)
    hs.bucket.mul_(2.0)
    hs.combine()
    assert rel_err(hs.gdelta, p_d / (3.0 * float(ref_l[7])) + 2.0 * r_d) < 2e-5


def test_fused_adam_matches_separate_optimizer_step(oracle):
    """HamStage2.step(optimizer=...) applies Adam inside the mesh tail (one rank) or inside fmhr_ham_combine (view-sharded form):
    three steps must leave the same parameters and moments as step() + optimizer.step()."""
    from fmhr_amd.ham import HamStage2
    from fmhr_amd.optim import Adam
    name, dims, seed = CASES[2]
    sc = prepare(oracle, dims, seed)
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}

    def run(mode):
        hs = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
        delta = g["delta"].clone().requires_grad_(True)
        albedo = g["albedo"][None].clone().requires_grad_(True)
        opt = Adam([{'params': delta, 'lr': 1e-4}, {'params': albedo, 'lr': 5e-3}])
        for _ in range(3):
            args = (g["vertices"], delta.detach(), albedo.detach()[0], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
            if mode == "separate":
                _, gd, ga = hs.step(*args)
                delta.grad, albedo.grad = gd, ga.view(1, -1, 3)
                opt.step()
            elif mode == "fused":
                hs.step(*args, optimizer=opt, params=(delta, albedo))
            else:                                    # the view-sharded form on one process: split backward, combine with the update
                hs.forward(*args[:6], g["mask"], g["valid_mask"])
                hs.backward(*args, split=True)
                hs.combine((opt.fused_param(delta), opt.fused_param(albedo)))
        return delta.detach().clone(), albedo.detach().clone(), opt.state[delta]['exp_avg_sq'].clone()

    d0, a0, v0 = run("separate")
    d1, a1, v1 = run("fused")
    assert float((d0 - g["delta"]).abs().max()) > 0
    # run-to-run spread of the float-atomic parts of the gradient aside, the update rule is the same statements
    assert rel_err(d1 - g["delta"], d0 - g["delta"]) < 1e-4 and rel_err(a1, a0) < 1e-6 and rel_err(v1, v0) < 1e-4
    d2, a2, v2 = run("sharded-form")
    assert rel_err(d2 - g["delta"], d0 - g["delta"]) < 1e-4 and rel_err(a2, a0) < 1e-6 and rel_err(v2, v0) < 1e-4


def _step_vs_oracle(oracle, sc, name, osc=None):
    """one fused step against the fp32 oracle (bit-identical rast) and the fp64-on-fp32-decisions oracle (values).
    `osc`: the scene the oracle sees when it must differ from the GPU's (see the empty-valid-set case)."""
    osc = osc or sc
    from fmhr_amd.ham import HamStage2
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    hs = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
    if not os.environ.get("FMHR_RASTER_BINNED"):
        expect_layout(hs, os.environ.get("FMHR_ROW_PITCH") == "0")
    losses, gdelta, galbedo = hs.step(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"],
                                      g["valid_mask"])
    ld = hs.loss_dict()
    v = hs.views()
    pos_gpu = v["pos"].cpu()

    def run_oracle(dtype, decisions=None):
        c = lambda t: t.to(dtype)
        d = c(sc["delta"]).clone().requires_grad_(True)
        a = c(sc["albedo"])[None].clone().requires_grad_(True)
        ref = oracle.ham_stage2_losses(c(sc["vertices"]), d, a, c(sc["sh0"]), c(sc["w2c"]), c(sc["proj"]), c(sc["img"]),
                                       c(osc["mask"]), c(sc["valid_mask"]), sc["faces"], (sc["H"], sc["W"]), osc["weights"],
                                       sc["elm"], pos_value=pos_gpu, decisions=decisions)
        ref["loss"].backward()
        return ref, d.grad, a.grad[0]

    ref, gd32, ga32 = run_oracle(torch.float32)
    assert torch.equal(v["rast"].cpu(), ref["rast"]), name
    assert osc is not sc or ld["n_valid"] == ref["n_valid"], name
    with oracle.f32_decisions():
        # (osc: the oracle deliberately sees another scene than the GPU — its own decisions then)
        ref64, gd64, ga64 = run_oracle(torch.float64, gpu_loop_decisions(ref, v["tmp_img"], sc["img"], sc["mask"], name) if osc is sc else ref["decisions"])
    for k in ("sfs", "lap", "albedo", "mask", "edge", "delta"):
        assert abs(ld[k] - float(ref64[k])) <= TOL * max(abs(float(ref64[k])), 1e-6), (name, k, ld[k], float(ref64[k]))
    assert all(torch.isfinite(t).all() for t in (losses, gdelta, galbedo)), name
    e_d, e_a = rel_err(gdelta, gd64), rel_err(galbedo, ga64)
    o_d, o_a = rel_err(gd32, gd64), rel_err(ga32, ga64)
    print(f"{name}: grad rel err vs fp64 oracle: delta {e_d:.2e} albedo {e_a:.2e}; fp32 oracle's own: {o_d:.2e} {o_a:.2e}")
    grad_bar(name, e_d, e_a, o_d, o_a, rel_err(gdelta, gd32), rel_err(galbedo, ga32))
    v = dict(v, gdelta=gdelta.clone(), galbedo=galbedo.clone(), gd64=gd64, ga64=ga64, o_d=o_d, o_a=o_a, binned=binned_form(hs), pitch=hs.pitch)
    return ld, v


def test_fused_step_crumpled_mesh(oracle):
    """a heavily perturbed (self-intersecting, crumpled) mesh: many silhouette pairs, thin and flipped triangles, long candidate
    lists — the decisions must still match the fp32 oracle bit for bit and the antialias work-item list must not overflow."""
    sc = prepare(oracle, (4, 128, 96, 16, 24), 21)
    sc["delta"] = sc["delta"] * 40.0                      # ~8e-3: several edge lengths of this mesh
    ld, v = _step_vs_oracle(oracle, sc, "crumpled")
    assert int(v["aa_counter"][1]) == 0


def _sweep_cases():
    """seeded random shapes: view counts around the lane-group sizes (8 views x 1..6 per lane), resolutions with and without
    H*W % 64 == 0 (scalar-view resolve pass or not) and W % 64, meshes from 10-pixel triangles (all of them large-box: swept by
    their wave) to sub-pixel ones, subdivided or not."""
    rng = np.random.default_rng(20240807)
    out = []
    for k in range(14):
        n = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 13, 17, 25, 33, 47]))
        H = int(rng.choice([33, 40, 48, 64, 72, 96, 100, 128]))
        W = int(rng.choice([31, 38, 48, 64, 80, 96, 120, 128]))
        rings, segs = [(4, 6), (6, 8), (10, 14), (16, 24), (24, 32)][int(rng.integers(0, 5))]
        sub = int(rng.integers(0, 2)) if rings <= 10 else 0
        if n * H * W > 600_000:
            n = max(1, 600_000 // (H * W))
        out.append((f"s{k}_{n}x{H}x{W}_m{rings}x{segs}s{sub}", (n, H, W, rings, segs, sub), 100 + k))
    return out


@pytest.mark.parametrize("case", _sweep_cases(), ids=lambda c: c[0])
def test_fused_step_shape_sweep(oracle, case):
    name, dims, seed = case
    sc = prepare(oracle, dims, seed)
    if float((sc["mask"] > 0).float().mean()) == 0:
        pytest.skip("no photometric pixel in this random scene")
    _step_vs_oracle(oracle, sc, name)


def test_aa_list_overflow_is_reported(oracle):
    """an antialias work-item list that is too small must not pass silently: loss_dict() raises."""
    from fmhr_amd.ham import HamStage2
    sc = prepare(oracle, (2, 128, 96, 16, 24), 22)
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    hs = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"], aa_cap=2048)   # 1 item per region
    hs.step(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    with pytest.raises(RuntimeError, match="overflowed"):
        hs.loss_dict()


def test_fused_step_edge_cases(oracle):
    """views without any coverage, and a batch without a single valid pixel (the L1 mean over an empty set)."""
    # (a) the middle view looks past the mesh (clip x shifted by 5 w): no fragment, no coverage entry, empty segments
    sc = prepare(oracle, (3, 96, 80, 12, 16), 11)
    sc["proj"] = sc["proj"].clone()
    sc["proj"][1, :, 0] += 5.0 * sc["proj"][1, :, 3]
    sc["mask"] = sc["mask"].clone()
    sc["mask"][1] = 0
    ld, v = _step_vs_oracle(oracle, sc, "empty view")
    assert float(v["rast"][1].abs().sum()) == 0.0 and float(v["pred_mask"][1].abs().sum()) == 0.0
    # (b) no valid pixel at all: mask == 0 everywhere -> n_valid = 0.  The reference's F.l1_loss over an empty selection is NaN
    # (0 / 0); the build defines the term and its gradient as 0 there, i.e. the step must equal the one with sfs weight 0
    sc = prepare(oracle, (2, 96, 80, 12, 16), 12)
    osc = dict(sc, weights=dict(sc["weights"], sfs=0.0))
    sc = dict(sc, mask=torch.zeros_like(sc["mask"]))
    ld, v = _step_vs_oracle(oracle, sc, "no valid pixel", osc)
    assert ld["n_valid"] == 0 and ld["sfs"] == 0.0


def test_unchanged_loop_body_on_dr(oracle):
    """the reference's own loop body (restated verbatim in oracle.ham_stage2_losses) executed on fmhr_amd.dr / fmhr_amd.utils
    GPU ops must agree with the fused step and with the CPU oracle."""
    from fmhr_amd.ham import HamStage2
    sc = prepare(oracle, (2, 160, 120, 24, 32), 5)
    dev = torch.device("cuda:0")
    weights, elm = sc["weights"], sc["elm"]
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    faces = sc["faces"].to(dev)
    n = sc["views"]
    out = _loop_body_on_dr(g, faces, n, sc["H"], sc["W"], weights, elm)
    loss, delta, albedo = out["loss"], out["delta"], out["albedo"]
    # oracle (fp64 values on fp32 decisions) evaluated at the clip positions this composition produced (torch's GPU einsum)
    with oracle.f32_decisions():
        d = sc["delta"].double().clone().requires_grad_(True)
        al = sc["albedo"].double()[None].clone().requires_grad_(True)
        c = lambda t: t.double()
        ref = oracle.ham_stage2_losses(c(sc["vertices"]), d, al, c(sc["sh0"]), c(sc["w2c"]), c(sc["proj"]), c(sc["img"]),
                                       c(sc["mask"]), c(sc["valid_mask"]), sc["faces"], (sc["H"], sc["W"]), weights, elm,
                                       pos_value=out["pos"].cpu())
        ref["loss"].backward()
    assert torch.equal(out["rast"].cpu()[..., 3].double(), ref["rast"][..., 3])
    assert abs(float(loss) - float(ref["loss"])) <= TOL * abs(float(ref["loss"]))
    assert rel_err(delta.grad, d.grad) < TOL and rel_err(albedo.grad, al.grad) < TOL
    # and the fused step agrees with the op-by-op composition (their clip positions differ in the last bit only)
    hs = HamStage2(faces, sc["V"], n, sc["H"], sc["W"], weights, elm)
    hs.step(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    ld = hs.loss_dict()
    assert abs(ld["total"] - float(loss)) <= 1e-3 * abs(float(loss))
    assert rel_err(hs.gdelta, delta.grad) < 2e-2 and rel_err(hs.galbedo, albedo.grad[0]) < 2e-2
    # ... and TIGHTLY once both run on the same clip positions (the 2e-2 above is the price of a last-bit difference between torch's
    # einsum and the kernel's FMA chain: a snapped vertex that moves by one unit changes a few pixels' owners): the reference's loop
    # body on the generic ops, evaluated at the fused step's positions, must make the same decisions and agree to 1e-4
    gd_f, ga_f = hs.gdelta.clone(), hs.galbedo.clone()
    pos_f = hs.views()["pos"].clone()
    out2 = _loop_body_on_dr(g, faces, n, sc["H"], sc["W"], weights, elm, pos_value=pos_f)
    assert torch.equal(out2["rast"][..., 3], hs.views()["rast"][..., 3]) and out2["n_valid"] == ld["n_valid"]
    assert abs(ld["total"] - float(out2["loss"])) <= TOL * abs(float(out2["loss"]))
    e = (rel_err(gd_f, out2["delta"].grad), rel_err(ga_f, out2["albedo"].grad[0]))
    log_parity("fused step vs the unchanged loop body on the generic ops at identical clip positions: gdelta %.1e galbedo %.1e" % e)
    assert e[0] < TOL and e[1] < TOL, e


@pytest.mark.parametrize("shape", [(3, 512, 512), (2, 1024, 1024), (1, 1024, 1280)], ids=lambda s: "x".join(map(str, s)))
def test_full_size_step_vs_oracle_and_composed_ops(oracle, shape):
    """the metric's mesh (V=49281, F=98432) at the metric's 512x512, at configs[3]'s 1024x1024 and at the demo capture's
    1280x1024 (conf/demo_sfs.conf: w=1280, h=1024): (1) the fused step against the CPU oracle exactly as the small
    cases are checked (bit-identical rast vs the fp32 twin, values vs fp64-on-fp32-decisions); (2) against the reference's loop
    body composed from the generic fmhr_amd.dr / fmhr_amd.utils ops on identical clip positions: every discrete decision
    (coverage, depth winner, AA pairs, valid set) agrees exactly, values to 1e-4."""
    from fmhr_amd import synthetic as S
    from fmhr_amd.ham import render_forward, edge_length_mean
    dev = torch.device("cuda:0")
    n, H, W = shape
    sc = S.make_scene(n, H, W, "mano", 0, 3, seed=3)
    assert sc["V"] == 49281 and sc["F"] == 98432                                      # train_mlp.py:150-154
    faces = sc["faces"].to(dev)
    fwd = lambda v, a, s, m: render_forward(faces, v, a, s, sc["w2c"].to(dev), sc["proj"].to(dev), m, H, W)
    tg = S.make_targets(sc, fwd, device=dev)
    gen = torch.Generator().manual_seed(11)
    sc["delta"] = 2e-4 * torch.randn(sc["V"], 3, generator=gen)                       # off the start mesh: every term is live
    sc["albedo"] = sc["albedo0"] + 0.05 * torch.rand(sc["V"], 3, generator=gen)       # not constant: |L a| = 0 has no gradient
    sc.update({k: tg[k].cpu() for k in ("img", "mask", "valid_mask")})
    sc["elm"] = float(edge_length_mean(sc["vertices"].to(dev), faces))
    ld, v = _step_vs_oracle(oracle, sc, "metric_mesh_%dx%dx%d" % shape)
    assert ld["n_valid"] > 0.1 * n * H * W
    if shape[1] != 512:
        return

    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    # regularisers in fp64 (the checker's torch Laplacian): two fp32 evaluations of |L x| on this mesh differ by > 1e-4
    out = _loop_body_on_dr(g, faces, n, H, W, sc["weights"], sc["elm"], pos_value=v["pos"], reg_dtype=torch.float64,
                           lap_fn=oracle.laplacian_smoothing)
    assert torch.equal(out["pos"], v["pos"])
    assert torch.equal(out["rast"][..., 3], v["rast"][..., 3])                       # same winner in every pixel
    assert float((out["rast"][..., :3] - v["rast"][..., :3]).abs().max()) <= 1e-6    # u, v, z/w
    assert ld["n_valid"] == out["n_valid"]
    assert float((out["tmp_img"] - v["tmp_img"]).abs().max()) <= 1e-5
    for k in ("sfs", "lap", "albedo", "mask", "edge", "delta"):
        assert abs(ld[k] - out["terms"][k]) <= TOL * max(abs(out["terms"][k]), 1e-6), (k, ld[k], out["terms"][k])
    assert abs(ld["total"] - float(out["loss"])) <= TOL * abs(float(out["loss"]))
    assert rel_err(v["galbedo"], out["albedo"].grad[0]) < TOL
    # position gradients: the generic ops differentiate per pixel exactly as the fp32 oracle does, and share its conditioning
    # on the mesh's edge-on triangles (a handful of vertices, 1e-3 of the largest gradient here); the fused step, which passed
    # the fp64 bar in part (1), recomputes each face's clip positions bit-identically to the rasterized ones and is tighter
    e_c = rel_err(out["delta"].grad, v["gd64"])
    print(f"composed ops: gdelta rel err vs fp64 oracle {e_c:.2e} (fp32 oracle {v['o_d']:.2e}, fused "
          f"{rel_err(v['gdelta'], v['gd64']):.2e})")
    log_parity(f"metric_mesh_%dx%dx%d: GENERIC ops (import-swap path) gdelta vs fp64 {e_c:.2e} (fp32 twin {v['o_d']:.2e}), "
               f"galbedo vs fused {rel_err(v['galbedo'], out['albedo'].grad[0]):.2e}" % shape)
    assert e_c < max(TOL, 2 * v["o_d"])                     # the generic per-pixel ops ARE the fp32 formulation: bounded by its twin
    assert rel_err(v["gdelta"], out["delta"].grad) < max(TOL, 2 * v["o_d"])


def test_full_size_properties():
    """sub3 mesh (V=49281, F=98432) at 512x512, 4 views: size-independent properties (the oracle is too slow here)."""
    from fmhr_amd import synthetic as S
    from fmhr_amd.ham import HamStage2, render_forward, edge_length_mean
    dev = torch.device("cuda:0")
    sc = S.make_scene(4, 512, 512, "mano", 0, 3, seed=0)
    faces = sc["faces"].to(dev)
    fwd = lambda v, a, s, m: render_forward(faces, v, a, s, sc["w2c"].to(dev), sc["proj"].to(dev), m, 512, 512)
    tg = S.make_targets(sc, fwd, device=dev)
    assert 0.12 < tg["coverage"] < 0.45, tg["coverage"]
    elm = edge_length_mean(sc["vertices"].to(dev), faces)
    hs = HamStage2(faces, sc["V"], 4, 512, 512, sc["weights"], elm)
    args = [sc[k].to(dev).contiguous() for k in ("vertices",)] + [torch.zeros(sc["V"], 3, device=dev)] + \
           [sc[k].to(dev).contiguous() for k in ("albedo0", "sh0", "w2c", "proj")] + [tg["img"], tg["mask"], tg["valid_mask"]]
    hs.step(*args)
    v = hs.views()
    rast = v["rast"]
    ids = rast[..., 3]
    cov = ids > 0
    assert int(v["aa_counter"][1]) == 0
    assert (ids >= 0).all() and (ids <= sc["F"]).all() and (ids == ids.round()).all()
    assert (rast[..., 0][cov] >= 0).all() and (rast[..., 1][cov] >= 0).all() and (rast[..., 0][cov] <= 1).all()
    assert (rast[..., :3][~cov] == 0).all()
    assert (rast[..., 2][cov] < 0).all() and (rast[..., 2][cov] > -1).all()       # z/w = -0.1/z_cam in (-1, 0)
    # delta = 0 and start mesh: the silhouette term is exactly zero (valid_mask IS the AA silhouette of the start mesh)
    ld = hs.loss_dict()
    assert ld["mask"] == 0.0 and ld["delta"] == 0.0 and ld["sfs"] > 0 and ld["lap"] > 0
    assert torch.isfinite(hs.gdelta).all() and torch.isfinite(hs.galbedo).all()
    assert float(hs.galbedo.abs().max()) > 0 and float(hs.gdelta.abs().max()) > 0
    # pixels away from any id discontinuity are untouched by antialias
    same = torch.ones_like(cov)
    same[:, :, :-1] &= ids[:, :, :-1] == ids[:, :, 1:]
    same[:, :, 1:] &= ids[:, :, :-1] == ids[:, :, 1:]
    same[:, :-1, :] &= ids[:, :-1, :] == ids[:, 1:, :]
    same[:, 1:, :] &= ids[:, :-1, :] == ids[:, 1:, :]
    assert torch.equal(v["tmp_img"][same], v["tmp_img_pre_aa"][same])
    # linearity of the backward in the loss weights: doubling the photometric weight doubles its gradient share
    g1 = hs.galbedo.clone()
    w2 = dict(sc["weights"]); w2["sfs"] *= 2; w2["albedo"] *= 2
    hs2 = HamStage2(faces, sc["V"], 4, 512, 512, w2, elm)
    hs2.step(*args)
    assert float((hs2.galbedo - 2 * g1).abs().max()) <= 2e-4 * float(g1.abs().max())
    # determinism of the discrete part
    r1 = rast.clone()
    hs.step(*args)
    assert torch.equal(hs.views()["rast"], r1)


def test_full_size_workspace_state_after_600_steps():
    """The metric workload itself (48 views x 512 x 512, V = 49 281, binned rasterizer, 4 views per lane in the backward): 600
    optimiser steps on one workspace — two 256-step periods of the owner marks, four of their 128-step wipes, the antialias list grown
    to its stationary size — then the step at the final parameters must equal the same step on a fresh workspace, no list or bin may
    have overflowed, and the optimisation must have made progress."""
    from fmhr_amd import synthetic as S
    from fmhr_amd.ham import HamStage2, render_forward, edge_length_mean
    from fmhr_amd.optim import Adam
    dev = torch.device("cuda:0")
    sc = S.scene_from_config("metric", 0, None)
    n, H, W, V = sc["views"], sc["H"], sc["W"], sc["V"]
    faces = sc["faces"].to(dev)
    w2c, proj = sc["w2c"].contiguous().to(dev), sc["proj"].contiguous().to(dev)
    tg = S.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
    vt, sh = sc["vertices"].to(dev).contiguous(), sc["sh0"].contiguous().to(dev)
    elm = edge_length_mean(vt, faces)
    hs = HamStage2(faces, V, n, H, W, sc["weights"], elm)
    assert binned_form(hs)
    delta = torch.zeros(V, 3, device=dev, requires_grad=True)
    albedo = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
    opt = Adam([{'params': delta, 'lr': sc["lr"]["lr"]}, {'params': albedo, 'lr': sc["lr"]["albedo_lr"]}])
    rest = (sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"])
    first = None
    for it in range(600):
        losses, _, _ = hs.step(vt, delta.detach(), albedo.detach()[0], *rest, optimizer=opt, params=(delta, albedo))
        if it == 0:
            first = float(losses[6])
    ld = hs.loss_dict()                                      # raises on an antialias-list / fragment-bin overflow
    assert np.isfinite(ld["total"]) and ld["total"] < first
    assert int(hs.aa_item_count()) > 20000                   # the silhouette has roughened: the stationary list
    d_fin, a_fin = delta.detach().clone(), albedo.detach()[0].clone()
    hs.step(vt, d_fin, a_fin, *rest)
    l_old, gd_old, ga_old = hs.losses.clone(), hs.gdelta.clone(), hs.galbedo.clone()
    r_old = hs.views()["rast"].clone()
    fresh = HamStage2(faces, V, n, H, W, sc["weights"], elm)
    fresh.step(vt, d_fin, a_fin, *rest)
    assert torch.equal(fresh.views()["rast"], r_old)
    assert rel_err(l_old[:7], fresh.losses[:7]) < 1e-6 and float(l_old[7]) == float(fresh.losses[7])
    e = (rel_err(gd_old, fresh.gdelta), rel_err(ga_old, fresh.galbedo))
    log_parity("metric workload, step 601 on one workspace vs a fresh workspace: gdelta %.1e galbedo %.1e, antialias items %d" % (*e, int(hs.aa_item_count())))
    assert e[0] < 5e-5 and e[1] < 5e-5, e


def test_stale_owner_mark_after_256_steps(oracle):
    """ADVICE r2 (shade.hip box path): the owner mark of a (view, face) is the low BYTE of the step that wrote it.  A face that owns
    pixel p at step s and then covers no pixel centre any more, while p stays uncovered, keeps its coverage entry, its mark, the
    pixel word of p and p's image gradient — and at step s + 256 the mark matches again.  One workspace: step 1 on a zoomed-IN
    camera (large silhouette), steps 2..257 on the same mesh zoomed OUT (sub-pixel faces, the outer ring of step 1's silhouette
    uncovered): the gradient of step 257 must equal the one a fresh workspace gives."""
    from fmhr_amd.ham import HamStage2
    sc = prepare(oracle, (3, 96, 80, 24, 32), 31)
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    far = g["proj"].clone()
    far[:, 0, 0] *= 0.45                                   # zoom out about the principal point: fx, fy scaled (row-vector layout)
    far[:, 1, 1] *= 0.45
    mask1 = torch.ones_like(g["mask"])
    args_near = (g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], mask1, g["valid_mask"])
    args_far = (g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], far, g["img"], mask1, g["valid_mask"])
    hs = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
    hs.step(*args_near)
    cov_near = hs.views()["rast"][..., 3] > 0
    for _ in range(256):
        hs.step(*args_far)
    ld = hs.loss_dict()
    gd, ga = hs.gdelta.clone(), hs.galbedo.clone()
    cov_far = hs.views()["rast"][..., 3] > 0
    assert int((cov_near & ~cov_far).sum()) > 200          # the scenario exists: pixels that lost their coverage
    fresh = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
    fresh.step(*args_far)
    ld0 = fresh.loss_dict()
    e_d, e_a = rel_err(gd, fresh.gdelta), rel_err(ga, fresh.galbedo)
    log_parity(f"stale_owner_mark: step 257 on one workspace vs a fresh workspace: gdelta {e_d:.2e} galbedo {e_a:.2e}")
    assert abs(ld["total"] - ld0["total"]) <= 1e-6 * abs(ld0["total"])
    assert e_d < 5e-5 and e_a < 5e-5, (e_d, e_a)          # run-to-run spread of the antialias-backward atomics only


def test_fused_step_as_autograd_function(oracle):
    """north_star: the fused step "exposed as a torch.autograd.Function": loss = ham_step(...); loss.backward() fills delta.grad /
    albedo.grad ([1,V,3] as the reference keeps it, mesh_sfs_optim.py:176) and torch.optim.Adam steps on them (:308-310)."""
    from fmhr_amd.ham import HamStage2, ham_step
    sc = prepare(oracle, (2, 128, 96, 16, 24), 0)
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    hs = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
    hs.step(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    ref_d, ref_a, ref_l = hs.gdelta.clone(), hs.galbedo.clone(), hs.losses.clone()
    delta = g["delta"].clone().requires_grad_(True)
    albedo = g["albedo"][None].clone().requires_grad_(True)
    optimizer = torch.optim.Adam([{'params': delta, 'lr': 1e-4}, {'params': albedo, 'lr': 5e-3}])
    loss, losses = ham_step(hs, g["vertices"], delta, albedo, g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    assert loss.requires_grad and not losses.requires_grad and loss.dim() == 0
    optimizer.zero_grad()
    (2.0 * loss).backward()
    assert albedo.grad.shape == (1, sc["V"], 3)
    assert rel_err(delta.grad, 2.0 * ref_d) < 5e-5 and rel_err(albedo.grad[0], 2.0 * ref_a) < 1e-6
    assert rel_err(losses[:7], ref_l[:7]) < 1e-6 and float(loss) == float(losses[6])
    d0 = delta.detach().clone()
    optimizer.step()
    assert float((delta.detach() - d0).abs().max()) > 0
    # a second step overwrites the workspace buffers; the first step's saved gradients must be its own
    loss2, _ = ham_step(hs, g["vertices"], delta, albedo, g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    assert float(loss2) != float(loss)


# (3, 128, 128, 24, 32): just inside the micro-triangle bound H*W <= 16 F with ~7 fragments per face: a workgroup's 256 faces emit ~1700
# fragments = four rounds of the 512-entry task list / reservation table, and some boxes exceed 32 pixels (swept by their wave)
@pytest.mark.parametrize("dims", [(3, 192, 192, 64, 96), (2, 256, 320, "mano", 0, 2), (5, 64, 128, 48, 64), (3, 128, 128, 24, 32)],
                         ids=lambda d: "x".join(map(str, d)))
def test_binned_rasterizer_equals_the_atomic_key_form(oracle, dims, monkeypatch):
    """The fused step rasterizes through per-tile LDS bins where it can (micro-triangle mesh, W % 64 == 0, H % 4 == 0: raster.hip
    k_raster_tile) and through the global atomicMin depth keys + resolve pass elsewhere: on the same inputs both forms must give the
    SAME per-pixel words, images, losses and gradients (bit-identical images: same fragment values, same depth / tie order)."""
    from fmhr_amd.ham import HamStage2
    sc = prepare(oracle, dims, 41)
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    args = (g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FMHR_RASTER_BINNED", mode)
        hs = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
        for _ in range(3):                                   # the bins are self-cleaning: the third step must equal the first
            hs.step(*args)
        v = hs.views()
        out[mode] = dict(rast=v["rast"].clone(), img=v["tmp_img"].clone(), pm=v["pred_mask"].clone(), pre=v["tmp_img_pre_aa"].clone(),
                         gd=hs.gdelta.clone(), ga=hs.galbedo.clone(), ld=hs.loss_dict())
    a, b = out["1"], out["0"]
    assert float((a["rast"][..., 3] > 0).float().mean()) > 0.05
    # bit-identical: the rasterizer image and the shaded image before antialias; the antialiased images add their blends with float
    # atomics (a pixel that receives two blends sums them in either order: one ulp between ANY two runs)
    assert torch.equal(a["rast"], b["rast"]) and torch.equal(a["pre"], b["pre"])
    assert float((a["img"] - b["img"]).abs().max()) <= 1e-6 and float((a["pm"] - b["pm"]).abs().max()) <= 1e-6
    assert a["ld"]["n_valid"] == b["ld"]["n_valid"] and abs(a["ld"]["total"] - b["ld"]["total"]) <= 1e-6 * abs(b["ld"]["total"])
    assert rel_err(a["gd"], b["gd"]) < 5e-5 and rel_err(a["ga"], b["ga"]) < 1e-6      # antialias-backward atomics order only


def test_fragment_bin_overflow_is_reported(oracle, monkeypatch):
    """a fragment bin that is too small must not pass silently: loss_dict() raises (sticky flag next to the antialias list's)."""
    from fmhr_amd.ham import HamStage2
    sc = prepare(oracle, (2, 192, 192, 64, 96), 23)          # micro-triangle mesh, W % 64 == 0: the binned form
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    monkeypatch.setenv("FMHR_RASTER_BINNED", "1")
    # 64 entries per bin = one fragment per pixel: every doubly covered segment overflows
    hs = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"], bin_cap=64)
    assert binned_form(hs)
    hs.step(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    with pytest.raises(RuntimeError, match="fragment bin"):
        hs.loss_dict()
    # the high-water mark says what the capacity has to be: a workspace with that capacity holds the same step without overflow, and
    # its own mark is the same number (the count of a segment does not depend on the capacity)
    need = hs.bin_depth_max()
    assert need > 64
    hs2 = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"], bin_cap=need)
    hs2.step(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    hs2.loss_dict()
    assert hs2.bin_depth_max() == need
    if need > 65:
        hs3 = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"], bin_cap=need - 1)
        hs3.step(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
        with pytest.raises(RuntimeError, match="fragment bin"):
            hs3.loss_dict()


def test_two_hand_sub3_mesh_at_the_sequence_resolution(oracle):
    """BASELINE.json configs[4] at size: two hands, each the exact sub3 mesh (V = 2 x 49281, F = 2 x 98432, two boundary loops),
    334 x 512 (conf/ih_sfsseq.conf:11-12) — the fused step (row pitch 384: binned rasterizer, row-aligned antialias scan) against
    the CPU oracle exactly as the small cases are checked."""
    from fmhr_amd import synthetic as S
    from fmhr_amd.ham import render_forward, edge_length_mean
    dev = torch.device("cuda:0")
    n, H, W = 2, 512, 334
    sc = S.make_scene(n, H, W, "mano", 0, 3, seed=5, hands=2)
    assert sc["V"] == 98562 and sc["F"] == 196864
    faces = sc["faces"].to(dev)
    fwd = lambda v, a, s, m: render_forward(faces, v, a, s, sc["w2c"].to(dev), sc["proj"].to(dev), m, H, W)
    tg = S.make_targets(sc, fwd, device=dev)
    gen = torch.Generator().manual_seed(12)
    sc["delta"] = 2e-4 * torch.randn(sc["V"], 3, generator=gen)
    sc["albedo"] = sc["albedo0"] + 0.05 * torch.rand(sc["V"], 3, generator=gen)
    sc.update({k: tg[k].cpu() for k in ("img", "mask", "valid_mask")})
    sc["elm"] = float(edge_length_mean(sc["vertices"].to(dev), faces))
    ld, v = _step_vs_oracle(oracle, sc, "two_hands_sub3_2x512x334")
    assert v["binned"] and v["pitch"] == 384
    assert ld["n_valid"] > 0.1 * n * H * W


@pytest.mark.parametrize("variant", ["shuffled_faces", "two_large_triangles", "sliver_faces"])
def test_binned_rasterizer_rare_paths(oracle, variant, monkeypatch):
    """The binned rasterizer's slow paths against the atomic-key form AND the oracle: (a) faces in random order — the 256 faces of a
    workgroup are scattered over the image, their segments collide in the workgroup's direct-mapped LDS table and the fragments go
    through the per-wave run reservation (`bin_append`); (b) a micro-triangle mesh with two large triangles added — swept by their
    wave in the forward, queued for the backward's wave-per-face workers (extra workgroups of the k_face_bwd launch, sums through the
    per-vertex buffer the mesh tail consumes and re-zeroes: the comparison is made after the SECOND step on the workspace); (c) the
    same mesh with 32 vertices pushed out into spikes — the long thin triangles an optimisation run grows: dozens of
    large-box (view, face)s that own pixels, several per wave of the triangle kernel."""
    from fmhr_amd.ham import HamStage2
    sc = prepare(oracle, (4, 192, 192, 64, 96), 2)           # the mesh of the "micro" case, four views (one looks into the wrist)
    rng = np.random.default_rng(7)
    f = sc["faces"].clone()
    if variant == "shuffled_faces":
        f = f[torch.from_numpy(rng.permutation(f.shape[0]))].contiguous()
    elif variant == "sliver_faces":
        ring = lambda r, s: 1 + r * 96 + s
        # 32 vertices pushed out along their radius: spikes whose six incident faces are long thin triangles, visible from outside
        idx = torch.tensor([ring(r, s) for r in (20, 40) for s in range(0, 96, 6)])
        sc = dict(sc, delta=sc["delta"].clone())
        sc["delta"][idx] += 0.35 * sc["vertices"][idx]
    else:
        V = sc["V"]
        ring = lambda r, s: 1 + r * 96 + s                               # blob_mesh vertex numbering: pole 0, then rings x segments
        # two large triangles that cap the open wrist (rim = the last ring): in front of everything for a camera below the hand
        big = torch.tensor([[ring(63, 0), ring(63, 32), ring(63, 64)], [ring(63, 0), ring(63, 64), ring(63, 80)]], dtype=torch.int32)
        assert int(big.max()) < V
        f = torch.cat([f, big]).contiguous()
    sc = dict(sc, faces=f, F=f.shape[0])
    sc["elm"] = float(oracle.edge_length_mean(sc["vertices"], f))
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    args = (g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FMHR_RASTER_BINNED", mode)
        hs = HamStage2(f.to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
        assert hs.binned == (mode == "1")
        hs.step(*args)
        hs.step(*args)
        v = hs.views()
        out[mode] = dict(rast=v["rast"].clone(), img=v["tmp_img"].clone(), pre=v["tmp_img_pre_aa"].clone(), gd=hs.gdelta.clone(),
                         ga=hs.galbedo.clone(), ld=hs.loss_dict())
    a, b = out["1"], out["0"]
    assert torch.equal(a["rast"], b["rast"]) and torch.equal(a["pre"], b["pre"])
    assert float((a["img"] - b["img"]).abs().max()) <= 1e-6               # (float-atomic blend order: see the test above)
    assert rel_err(a["gd"], b["gd"]) < 5e-5 and rel_err(a["ga"], b["ga"]) < 1e-6
    monkeypatch.setenv("FMHR_RASTER_BINNED", "1")
    _step_vs_oracle(oracle, sc, "binned_" + variant)                     # bit-identical rast vs the fp32 oracle, values vs fp64
    if variant == "two_large_triangles":
        ids = a["rast"][..., 3]
        assert int((ids == sc["F"]).sum() + (ids == sc["F"] - 1).sum()) > 100      # the large triangles own pixels
    if variant == "sliver_faces":
        # the spikes' faces own pixels, and enough of them are large-box ones: count the (view, face)s whose owned pixels span a box above 32
        ids = a["rast"][..., 3].long()
        nl = 0
        for b in range(ids.shape[0]):
            ys, xs = torch.nonzero(ids[b] > 0, as_tuple=True)
            fid = ids[b][ys, xs]
            mnx = torch.full((sc["F"] + 1,), 10**6, device=ids.device).scatter_reduce(0, fid, xs, "amin")
            mxx = torch.full((sc["F"] + 1,), -1, device=ids.device).scatter_reduce(0, fid, xs, "amax")
            mny = torch.full((sc["F"] + 1,), 10**6, device=ids.device).scatter_reduce(0, fid, ys, "amin")
            mxy = torch.full((sc["F"] + 1,), -1, device=ids.device).scatter_reduce(0, fid, ys, "amax")
            nl += int((((mxx - mnx + 1) * (mxy - mny + 1)) > 32)[mxx >= 0].sum())
        assert nl >= 40, nl


@pytest.mark.parametrize("dims,flat", [((3, 192, 192, 64, 96), False), ((3, 96, 80, 24, 32), False), ((3, 96, 80, 24, 32), True)],
                         ids=["binned", "binned_pitch128", "atomic_flat"])
def test_workspace_state_after_a_long_optimisation(oracle, dims, flat, monkeypatch):
    """Everything the step keeps between calls cleans itself (depth keys / fragment-bin cursors, owner marks, coverage entries,
    segment words, antialias lists, accumulator replicas): after 700 optimiser steps on ONE workspace — past the 256-step period of
    the owner marks and their 128-step wipe — a step at the final parameters must equal the same step on a fresh workspace."""
    from fmhr_amd.ham import HamStage2
    from fmhr_amd.optim import Adam
    if flat:
        monkeypatch.setenv("FMHR_ROW_PITCH", "0")             # the ABI-4 layout: pitch = W = 80, atomic depth keys
    sc = prepare(oracle, dims, 61)
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    hs = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
    expect_layout(hs, flat)
    assert binned_form(hs) == (not flat) or os.environ.get("FMHR_RASTER_BINNED") == "0"
    delta = g["delta"].clone().requires_grad_(True)
    albedo = g["albedo"][None].clone().requires_grad_(True)
    opt = Adam([{'params': delta, 'lr': 1e-4}, {'params': albedo, 'lr': 5e-3}])
    rest = (g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    first = None
    for it in range(700):
        losses, _, _ = hs.step(g["vertices"], delta.detach(), albedo.detach()[0], *rest, optimizer=opt, params=(delta, albedo))
        if it == 0:
            first = float(losses[6])
    ld = hs.loss_dict()                                      # raises on an antialias-list / fragment-bin overflow
    assert np.isfinite(ld["total"]) and ld["total"] < first  # it optimises
    d_fin, a_fin = delta.detach().clone(), albedo.detach()[0].clone()
    hs.step(g["vertices"], d_fin, a_fin, *rest)
    l_old, gd_old, ga_old = hs.losses.clone(), hs.gdelta.clone(), hs.galbedo.clone()
    r_old = hs.views()["rast"].clone()
    fresh = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
    fresh.step(g["vertices"], d_fin, a_fin, *rest)
    assert torch.equal(fresh.views()["rast"], r_old)
    assert rel_err(l_old[:7], fresh.losses[:7]) < 1e-6 and float(l_old[7]) == float(fresh.losses[7])
    e = (rel_err(gd_old, fresh.gdelta), rel_err(ga_old, fresh.galbedo))
    log_parity("step 701 on one workspace vs a fresh workspace (%s rasterizer): gdelta %.1e galbedo %.1e" % ("binned" if hs.binned else "atomic", *e))
    assert e[0] < 5e-5 and e[1] < 5e-5, e


# ---- ABI 5: the row pitch ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dims", [(3, 256, 334, "mano", 0, 2, 2),      # two hands at W = 334, micro-triangle: bins vs atomic keys
                                  (2, 512, 334, 24, 32),               # config C2's shape (coarse mesh: the large-triangle queue either way)
                                  (5, 50, 38, 10, 14),                 # H*W = 1900: flat segments straddle rows AND views
                                  (4, 72, 120, 16, 24),                # two segments per row, the second one half pad
                                  (9, 40, 65, 10, 14, 1)],             # one pixel past a segment boundary
                         ids=lambda d: "x".join(map(str, d)))
def test_row_pitch_equals_the_flat_layout(oracle, dims):
    """W % 64 != 0: the default layout (rows padded to a multiple of 64 inside the workspace: row-aligned segments, binned rasterizer
    on a micro-triangle mesh, mask-scan antialias forward) against the ABI-4 layout (pitch = W) on the same inputs: the rasterizer
    image and the shaded image bit-identical, the antialiased images / losses / gradients equal up to the order of float atomics — and
    nothing in the pad columns.  Three steps on each workspace (self-cleaning state behind the pad)."""
    from fmhr_amd.ham import HamStage2
    sc = prepare(oracle, dims, 77)
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    args = (g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    n, H, W = sc["views"], sc["H"], sc["W"]
    out = {}
    for flat in (False, True):
        hs = HamStage2(sc["faces"].to(dev), sc["V"], n, H, W, sc["weights"], sc["elm"], pitch=W if flat else None)
        expect_layout(hs, flat)
        for _ in range(3):
            hs.step(*args)
        v = hs.views()
        out[flat] = dict(rast=v["rast"].clone(), img=v["tmp_img"].clone(), pm=v["pred_mask"].clone(), pre=v["tmp_img_pre_aa"].clone(),
                         gd=hs.gdelta.clone(), ga=hs.galbedo.clone(), ld=hs.loss_dict(), items=int(hs.aa_item_count()))
        if not flat:
            # the pad columns of the workspace images: never covered, never a blend target
            Wp = hs.pitch
            off = hs.lib.fmhr_ham_workspace_offset(__import__("ctypes").byref(hs.dims), b"aa")
            aa_full = hs.ws[off: off + n * H * Wp * 16].view(torch.float32).view(n, H, Wp, 4)
            off = hs.lib.fmhr_ham_workspace_offset(__import__("ctypes").byref(hs.dims), b"rast")
            rast_full = hs.ws[off: off + n * H * Wp * 16].view(torch.float32).view(n, H, Wp, 4)
            assert float(aa_full[:, :, W:].abs().max()) == 0.0 and float(rast_full[:, :, W:].abs().max()) == 0.0
    a, b = out[False], out[True]
    assert float((a["rast"][..., 3] > 0).float().mean()) > 0.03
    assert torch.equal(a["rast"], b["rast"]) and torch.equal(a["pre"], b["pre"])
    assert a["items"] == b["items"] and a["items"] > 0                     # the same silhouette pairs blend (none across the pad)
    assert float((a["img"] - b["img"]).abs().max()) <= 1e-6 and float((a["pm"] - b["pm"]).abs().max()) <= 1e-6
    assert a["ld"]["n_valid"] == b["ld"]["n_valid"]
    for k in ("sfs", "lap", "albedo", "mask", "edge", "delta", "total"):
        assert abs(a["ld"][k] - b["ld"][k]) <= 2e-6 * max(abs(b["ld"][k]), 1e-6), (k, a["ld"][k], b["ld"][k])
    e = (rel_err(a["gd"], b["gd"]), rel_err(a["ga"], b["ga"]))
    log_parity("row pitch vs flat layout %s: rast / shaded bit-identical, %d antialias items, gdelta %.1e galbedo %.1e" % ("x".join(map(str, dims)), a["items"], *e))
    assert e[0] < 5e-5 and e[1] < 2e-6, e


def test_bad_pitch_is_rejected(oracle):
    from fmhr_amd.ham import HamStage2
    sc = prepare(oracle, (2, 64, 48, 10, 14), 3)
    dev = torch.device("cuda:0")
    for bad in (47, 100, 16384):                                           # below W; above W but not a multiple of 64; above 8192
        with pytest.raises(RuntimeError, match="pitch"):
            HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"], pitch=bad)
    assert HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"], pitch=128).pitch == 128


def test_bin_cursor_reset_order(oracle):
    """k_raster_bins reads a bin's fragment count through the scalar cache and resets the same word with a vector store: the order is
    explicit in the kernel (s_waitcnt + a compiler barrier), and a violation shows as segments that come out empty at random — so
    every one of eight consecutive steps on one workspace must give the first step's image and valid-pixel count."""
    from fmhr_amd.ham import HamStage2
    sc = prepare(oracle, (6, 192, 192, 64, 96), 19)
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    args = (g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    hs = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
    assert binned_form(hs)
    hs.step(*args)
    r1, nv1 = hs.views()["rast"].clone(), float(hs.losses[7])
    assert float((r1[..., 3] > 0).float().mean()) > 0.05
    for k in range(7):
        hs.step(*args)
        assert float(hs.losses[7]) == nv1, k
        assert torch.equal(hs.views()["rast"], r1), k


def test_aa_overflow_flag_is_sticky(oracle):
    """the antialias list's overflow word is rearmed every step — an overflow in the MIDDLE of a run was invisible at its
    end.  aa_counter[3] stays set until the workspace is initialised again: overflow in step 1, none in step 2 (every view looks past
    the mesh: no pair), and loss_dict() / check_overflow() still raise."""
    from fmhr_amd.ham import HamStage2
    sc = prepare(oracle, (2, 128, 96, 16, 24), 22)
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    hs = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"], aa_cap=2048)   # 1 - 2 items per region
    hs.step(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    assert hs.aa_overflowed_last_step() and hs.aa_overflowed()
    away = g["proj"].clone()
    away[:, :, 0] += 5.0 * away[:, :, 3]                                   # clip x shifted by 5 w: nothing on screen
    hs.step(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], away, g["img"], g["mask"], g["valid_mask"])
    assert int(hs.aa_item_count()) == 0 and not hs.aa_overflowed_last_step()
    assert hs.aa_overflowed() and hs.overflow_flags() == (True, False)
    with pytest.raises(RuntimeError, match="overflowed"):
        hs.check_overflow()
    with pytest.raises(RuntimeError, match="overflowed"):
        hs.loss_dict()


def demo_capture_scene():
    """the scene of test_demo_capture_cameras_through_the_fused_step: two synthetic sub3 hands laid over the two hands' keypoints of the
    reference's capture, seen by two of the capture's own cameras at 1280 x 1024 (CPU tensors + the fixture + the chosen cameras)"""
    from fmhr_amd import synthetic as S
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "demo_ingest.npz"))
    W, H = int(g["res"][0]), int(g["res"][1])
    assert (W, H) == (1280, 1024)
    kp = g["keypoints"].astype(np.float64)                              # [42,3]: two hands x 21, in the capture's normalised space
    base = S.make_scene(2, H, W, "mano", 0, 3, seed=4)
    bv, bf = base["vertices"].numpy().astype(np.float64), base["faces"].numpy()
    bc = bv.mean(0)
    _, bs, bax = np.linalg.svd(bv - bc, full_matrices=False)            # the synthetic hand's own principal axes (rows of bax)
    verts, faces = [], []
    for h in range(2):
        k = kp[21 * h: 21 * (h + 1)]
        c = k.mean(0)
        _, ks, kax = np.linalg.svd(k - c, full_matrices=False)
        if np.linalg.det(kax) * np.linalg.det(bax) < 0:                 # keep the orientation (a reflection would flip every face)
            kax[2] *= -1
        scale = 1.25 * (ks[0] / np.sqrt(21)) / (bs[0] / np.sqrt(bv.shape[0]))     # the keypoints' spread along the hand, a margin for the fingers
        verts.append(((bv - bc) @ bax.T * scale) @ kax + c)
        faces.append(bf + h * bv.shape[0])
    verts, faces = np.concatenate(verts).astype(np.float32), np.concatenate(faces).astype(np.int32)
    # two cameras of the capture that see both hands well inside the image
    X = np.concatenate([kp, np.ones((42, 1))], 1)
    margin = []
    for i in range(16):
        clip = X @ g["w2cs"][i].astype(np.float64) @ g["projs"][i].astype(np.float64)          # row-vector convention, as stored
        ndc = clip[:, :2] / clip[:, 3:4]
        margin.append(1.0 - np.abs(ndc).max())
    cams = [int(i) for i in np.argsort(margin)[-2:]]
    assert min(margin[i] for i in cams) > 0.1, margin
    rng = np.random.default_rng(4)
    V, F = verts.shape[0], faces.shape[0]
    assert V == 2 * 49281 and F == 2 * 98432
    t = torch.from_numpy
    sc = dict(views=2, H=H, W=W, V=V, F=F, hands=2, vertices=t(verts), faces=t(faces), w2c=t(g["w2cs"][cams].copy()), proj=t(g["projs"][cams].copy()),
              vertices_true=t((verts + 0.0005 * scale * rng.standard_normal(verts.shape)).astype(np.float32)),
              albedo_true=t(rng.uniform(0.3, 0.9, (V, 3)).astype(np.float32)), albedo0=torch.full((V, 3), 0.6),
              weights=dict(S.DEMO_WEIGHTS), lr=dict(S.DEMO_LR), scale=float(scale))
    sc["sh_true"] = t(np.concatenate([np.full((2, 1), 0.8), 0.1 * rng.standard_normal((2, 8))], 1).astype(np.float32))
    sc["sh0"] = sc["sh_true"] + 0.02 * torch.randn(2, 9, generator=torch.Generator().manual_seed(4))
    return sc, g, cams, X


def demo_capture_targets(sc, dev):
    from fmhr_amd import synthetic as S
    from fmhr_amd.ham import render_forward, edge_length_mean
    H, W, V = sc["H"], sc["W"], sc["V"]
    fdev = sc["faces"].to(dev)
    fwd = lambda v, a, s, m: render_forward(fdev, v, a, s, sc["w2c"].to(dev), sc["proj"].to(dev), m, H, W)
    tg = S.make_targets(sc, fwd, device=dev)
    gen = torch.Generator().manual_seed(14)
    sc["delta"] = (2e-4 * sc["scale"]) * torch.randn(V, 3, generator=gen)
    sc["albedo"] = sc["albedo0"] + 0.05 * torch.rand(V, 3, generator=gen)
    sc.update({k: tg[k].cpu() for k in ("img", "mask", "valid_mask")})
    sc["elm"] = float(edge_length_mean(sc["vertices"].to(dev), fdev))
    return sc


def test_demo_capture_cameras_through_the_fused_step(oracle):
    """Row N1 on the GPU: the REAL cameras of the reference's capture — tests/golden/demo_ingest.npz, derived by
    fmhr_amd.data.get_demo_data from demo_data/1 exactly as /root/reference/get_data.py:64-73,96-97 builds w2c / proj (off-centre
    principal points, fx != fy, 1280 x 1024 as conf/demo_sfs.conf:6-7) — drive the fused step: one synthetic sub3 hand per hand of the
    capture, laid over that hand's 21 three-D keypoints (centroid, principal axes, extent), two views, against the CPU oracle like
    every other case (bit-identical rast, losses / gradients vs fp64 on the fp32 decisions) through the binned rasterizer; and the
    capture's keypoints must project inside the rendered silhouette."""
    dev = torch.device("cuda:0")
    sc, g, cams, X = demo_capture_scene()
    H, W = sc["H"], sc["W"]
    # the real proj: off-centre principal point and fx != fy (get_data.py:66-69 on the capture's intrinsics)
    p0 = g["projs"][cams[0]].T
    assert abs(p0[0, 2]) > 1e-3 and abs(p0[1, 2]) > 1e-3 and abs(p0[0, 0] - p0[1, 1]) > 0.1 and p0[2, 3] == np.float32(-0.1)
    sc = demo_capture_targets(sc, dev)
    ld, v = _step_vs_oracle(oracle, sc, "demo_capture_cameras_2x1024x1280")
    assert v["binned"] and v["pitch"] == 1280
    cov = (v["rast"][..., 3] > 0).cpu().numpy()
    assert 0.01 < cov.mean() < 0.5, cov.mean()
    inside = []
    for j, i in enumerate(cams):
        clip = X @ g["w2cs"][i].astype(np.float64) @ g["projs"][i].astype(np.float64)
        px = np.floor((clip[:, 0] / clip[:, 3] + 1) * W / 2).astype(int)
        py = np.floor((clip[:, 1] / clip[:, 3] + 1) * H / 2).astype(int)
        assert ((px >= 0) & (px < W) & (py >= 0) & (py < H)).all()
        inside.append(float(cov[j][py, px].mean()))
    log_parity("demo capture cameras %s at 1280x1024: coverage %.3f, keypoints inside the rendered silhouette %s" % (cams, cov.mean(), ["%.2f" % x for x in inside]))
    assert min(inside) > 0.6, inside                                      # (the blob is not the hand: finger tips may stick out)


def test_sliver_workers_over_consecutive_steps(oracle, monkeypatch):
    """Round 4: a large-box (sliver) face that was queued LAST step is rasterized by a worker wave at the head of this step's triangle
    launch instead of by its own wave (raster.hip: sweep_workers); the owning lane decides from the marker its predecessor left in the
    coverage entry, the worker from the queue + this step's box — both must agree in every transition.  One workspace, six steps:
    no sliver -> slivers APPEAR (swept in place: nothing was queued) -> PERSIST (worker) -> persist with moved vertices (worker, new
    boxes) -> VANISH (the worker finds a small box and leaves it to its lane) -> none; every step against a FRESH workspace that sweeps
    everything in place (FMHR_SWEEP_WORKERS=0): bit-identical rasterizer image and shaded image, gradients up to the float-atomic order."""
    from fmhr_amd.ham import HamStage2
    sc = prepare(oracle, (4, 192, 192, 64, 96), 2)
    ring = lambda r, s: 1 + r * 96 + s
    idx = torch.tensor([ring(r, s) for r in (20, 40) for s in range(0, 96, 6)])
    spike = sc["delta"].clone()
    spike[idx] += 0.35 * sc["vertices"][idx]
    spike2 = sc["delta"].clone()
    spike2[idx] += 0.30 * sc["vertices"][idx]
    deltas = [sc["delta"], spike, spike, spike2, sc["delta"], sc["delta"]]
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    faces = sc["faces"].to(dev)
    args = lambda d: (g["vertices"], d.to(dev).contiguous(), g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    monkeypatch.delenv("FMHR_SWEEP_WORKERS", raising=False)
    hs = HamStage2(faces, sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
    assert binned_form(hs)
    n, H, Wp = sc["views"], sc["H"], hs.pitch
    qcnt = hs.raw("raster_ws", torch.uint8, n * H * Wp * 8 + 16)[n * H * Wp * 8:].view(torch.int32)     # [pushed so far, last step's count]
    queued = []
    for k, d in enumerate(deltas):
        monkeypatch.delenv("FMHR_SWEEP_WORKERS", raising=False)
        hs.step(*args(d))
        v = hs.views()
        got = dict(rast=v["rast"].clone(), pre=v["tmp_img_pre_aa"].clone(), gd=hs.gdelta.clone(), ga=hs.galbedo.clone(), ld=hs.loss_dict())
        queued.append(int(qcnt[1]))
        monkeypatch.setenv("FMHR_SWEEP_WORKERS", "0")
        ref = HamStage2(faces, sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
        ref.step(*args(d))
        rv = ref.views()
        assert torch.equal(got["rast"], rv["rast"]) and torch.equal(got["pre"], rv["tmp_img_pre_aa"]), k
        assert got["ld"]["n_valid"] == ref.loss_dict()["n_valid"], k
        assert rel_err(got["gd"], ref.gdelta) < 5e-5 and rel_err(got["ga"], ref.galbedo) < 2e-6, k
    # the scenario exists: dozens of large-box (view, face)s while the spikes stand, (next to) none without them
    assert min(queued[1:4]) >= 40 and max(queued[0], queued[5]) <= 4, queued
    log_parity("sliver workers over six steps (none / appear / persist / persist moved / vanish / none): queued large-box (view, face)s per step %s, "
               "every step bit-identical to a fresh in-place workspace" % queued)


def test_vertex_order_gives_the_callers_results(oracle):
    """fmhr_amd.ham.VertexOrder: the step on the RENUMBERED mesh (first-touch order of the face list; what run_ham and bench.py
    optimise in) against the oracle on the caller's numbering — `rast` bit-identical (the faces keep their order, a vertex's
    arithmetic does not know its index), the six losses and the gradients mapped back within the same bars as the plain step."""
    from fmhr_amd.ham import HamStage2, VertexOrder
    dev = torch.device("cuda:0")
    sc = prepare(oracle, (3, 256, 256, "mano", 0, 2), 31)              # two subdivision levels: appended midpoints, a real permutation
    ld0, v0 = _step_vs_oracle(oracle, sc, "vertex_order_reference")    # the caller's numbering: asserts rast == oracle, grads vs fp64
    order = VertexOrder(sc["faces"].to(dev), sc["V"], enabled=True)
    assert not order.identity
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    hs = HamStage2(order.faces, sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
    losses, gd, ga = hs.step(order.to_internal(g["vertices"]), order.to_internal(g["delta"]), order.to_internal(g["albedo"]), g["sh0"],
                             g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    ld, v = hs.loss_dict(), hs.views()
    assert torch.equal(v["rast"], v0["rast"])                           # face indices, barycentrics and depths: the same bits
    assert torch.equal(order.to_caller(v["pos"], 1), v0["pos"])
    assert ld["n_valid"] == ld0["n_valid"]
    for k in ("sfs", "lap", "albedo", "mask", "edge", "delta"):
        assert abs(ld[k] - ld0[k]) <= 2e-6 * max(abs(ld0[k]), 1e-6), (k, ld[k], ld0[k])     # summation order only
    gd, ga = order.to_caller(gd), order.to_caller(ga)
    e_d, e_a = rel_err(gd, v0["gd64"]), rel_err(ga, v0["ga64"])
    print(f"vertex_order: grad rel err vs fp64 oracle (caller's numbering): delta {e_d:.2e} albedo {e_a:.2e}; plain step: "
          f"{rel_err(v0['gdelta'], v0['gd64']):.2e} {rel_err(v0['galbedo'], v0['ga64']):.2e}")
    grad_bar("vertex_order", e_d, e_a, v0["o_d"], v0["o_a"])


def test_segment_state_words_follow_moving_coverage(oracle, monkeypatch):
    """The binned rasterizer's per-segment state words (raster.hip: BinArgs — an empty segment leaves the second pass on one packed
    load, and zeroes its coverage / valid-count words only when the last step with fragments left them set): one workspace through a
    sequence in which segments gain, keep and lose their fragments — the mesh on screen, shifted by a third of the image, off screen,
    back, shifted the other way — every step equal to the same step on a FRESH workspace (`rast`, the antialiased image, the loss
    vector, the valid-pixel count, both gradients up to the float-atomic order); then the same workspace sent down the atomic-key
    form for a step (FMHR_RASTER_BINNED=0: segment words written behind the state words' back) and back."""
    from fmhr_amd.ham import HamStage2
    sc = prepare(oracle, (3, 192, 192, 64, 96), 41)
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}

    def shifted(dx):
        p = g["proj"].clone()
        p[:, :, 0] += dx * p[:, :, 3]                                    # clip x shifted by dx * w: the image moves by dx / 2 of its width
        return p

    def run(hs, proj):
        losses, gd, ga = hs.step(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], proj, g["img"], g["mask"], g["valid_mask"])
        v = hs.views()
        return losses.clone(), gd.clone(), ga.clone(), v["rast"].clone(), v["tmp_img"].clone(), v["pred_mask"].clone()

    def same(a, b, what):
        assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4]) and torch.equal(a[5], b[5]), what
        assert float(a[0][7]) == float(b[0][7]), what
        assert rel_err(a[0][:7], b[0][:7]) < 1e-6 and rel_err(a[1], b[1]) < 1e-5 and rel_err(a[2], b[2]) < 1e-5, what

    def make():
        return HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])

    hs = make()
    assert binned_form(hs)
    seq = [0.0, 0.66, 5.0, 0.0, -0.66, -0.66, 5.0, 5.0, 0.3]
    covered = []
    for k, dx in enumerate(seq):
        got = run(hs, shifted(dx))
        same(got, run(make(), shifted(dx)), (k, dx))
        covered.append(int((got[3][..., 3] > 0).sum()))
    assert covered[2] == 0 and covered[6] == 0 and covered[0] > 1000 and covered[1] != covered[0]
    monkeypatch.setenv("FMHR_RASTER_BINNED", "0")                       # one step of the atomic-key form on the SAME workspace ...
    flat = run(hs, shifted(0.66))
    monkeypatch.delenv("FMHR_RASTER_BINNED")
    same(flat, run(make(), shifted(0.66)), "atomic-key form on the binned workspace")
    for dx in (5.0, -0.66, 0.0):                                        # ... and the bins again: every segment word counts as live
        same(run(hs, shifted(dx)), run(make(), shifted(dx)), ("after the atomic-key step", dx))


def test_two_frames_on_two_streams_equal_their_sequential_results():
    """BASELINE.json configs[4] / conf/ih_sfsseq.conf:10-12,29 in the form bench.py --config C5 runs it: the frames
    of a two-hand sequence are independent (delta, albedo) problems on the shared topology, each with its own HamStage2 workspace, and a
    rank's local frames run CONCURRENTLY on side streams forked from / joined to the current stream every step.  Two frames of the
    two-hand mesh at W = 334 (row pitch 384, binned rasterizer), six dependent fused-Adam steps each: once one after the other on the
    current stream, once on two HIP streams — the last step's face ids equal and (u, v, z/w) to 1e-5, losses and parameters equal up to the float-atomic
    summation order, and frame A's results are not frame B's (the two workspaces do not leak into each other).  The per-launch-group
    timing hook (thread-local state of the library) is attached during the concurrent steps, as bench.py does."""
    from fmhr_amd import _lib, synthetic as S
    from fmhr_amd.ham import HamStage2, edge_length_mean, render_forward
    from fmhr_amd.optim import Adam
    dev = torch.device("cuda:0")
    n, H, W = 4, 256, 334
    sc = S.make_scene(n, H, W, "mano", 0, 2, seed=5, hands=2)
    V = sc["V"]
    faces = sc["faces"].to(dev)
    w2c, proj, sh = sc["w2c"].to(dev).contiguous(), sc["proj"].to(dev).contiguous(), sc["sh0"].to(dev).contiguous()
    vt = sc["vertices"].to(dev).contiguous()
    elm = edge_length_mean(vt, faces)
    fwd = lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W)
    targets = []
    for frame in range(2):                                   # another frame: its own target geometry, same cameras / topology (bench.py: Problem)
        scf = dict(sc)
        if frame:
            scf["vertices_true"] = sc["vertices"] + 0.0005 * torch.randn(sc["vertices"].shape, generator=torch.Generator().manual_seed(1000 + frame))
        targets.append(S.make_targets(scf, fwd, device=dev))

    class Problem:
        def __init__(self, frame):
            self.tg = targets[frame]
            self.hs = HamStage2(faces, V, n, H, W, sc["weights"], elm)
            self.delta = torch.zeros(V, 3, device=dev, requires_grad=True)
            self.albedo = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
            self.opt = Adam([{'params': self.delta, 'lr': sc["lr"]["lr"]}, {'params': self.albedo, 'lr': sc["lr"]["albedo_lr"]}])

        def step(self):
            return self.hs.step(vt, self.delta.detach(), self.albedo.detach()[0], sh, w2c, proj, self.tg["img"], self.tg["mask"],
                                self.tg["valid_mask"], optimizer=self.opt, params=(self.delta, self.albedo))[0]

        def result(self):
            return dict(losses=self.hs.losses.clone(), delta=self.delta.detach().clone(), albedo=self.albedo.detach().clone(),
                        rast=self.hs.views()["rast"].clone(), gd=self.hs.gdelta.clone())

    steps = 6
    seq = [Problem(0), Problem(1)]
    assert seq[0].hs.pitch == 384 and binned_form(seq[0].hs)
    for _ in range(steps):
        for p in seq:
            p.step()
    torch.cuda.synchronize()
    ref = [p.result() for p in seq]
    for p in seq:
        p.hs.check_overflow()
    # the concurrent form: bench.py:360-374
    con = [Problem(0), Problem(1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    import ctypes
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(26)]
    for e in evs:
        e.record()
    arr = (ctypes.c_void_p * 26)(*[e.cuda_event for e in evs])
    for it in range(steps):
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        if it == steps - 2:
            _lib.call("fmhr_timing_attach", arr, 26, -1)     # the library's thread-local timing sink while two streams are in flight
        for i, p in enumerate(con):
            with torch.cuda.stream(streams[i]):
                p.step()
        if it == steps - 2:
            _lib.call("fmhr_timing_detach")
        for s in streams:
            cur.wait_stream(s)
    torch.cuda.synchronize()
    got = [p.result() for p in con]
    for p in con:
        p.hs.check_overflow()
    errs = []
    for f in range(2):
        # six dependent steps later: the same triangles in the same pixels.  Not bit for bit: the float-atomic order of the antialias
        # backward moves a gradient in its last bits from run to run — two SEQUENTIAL runs of frame 1 differ the same way in one run of
        # three (tools/experiments/r05_two_stream_debug.py: one vertex one ulp apart after six Adam steps, (u, v) of the 16 pixels of its
        # faces in the last bits, every face id equal) — so: ids equal but for a handful of pixels at most, values to 1e-5
        ids_a, ids_b = got[f]["rast"][..., 3], ref[f]["rast"][..., 3]
        assert int((ids_a != ids_b).sum()) <= 8, (f, int((ids_a != ids_b).sum()))
        same = (ids_a == ids_b).unsqueeze(-1)
        assert float(((got[f]["rast"] - ref[f]["rast"]).abs() * same)[..., :3].max()) < 1e-5, f
        assert float(got[f]["losses"][7]) == float(ref[f]["losses"][7]) and float(ref[f]["losses"][7]) > 1000
        e = (rel_err(got[f]["losses"][:7], ref[f]["losses"][:7]), rel_err(got[f]["delta"], ref[f]["delta"]),
             rel_err(got[f]["albedo"], ref[f]["albedo"]), rel_err(got[f]["gd"], ref[f]["gd"]))
        errs.append(e)
        assert e[0] < 1e-5 and e[1] < 1e-4 and e[2] < 1e-5 and e[3] < 1e-4, (f, e)
    # the frames are different problems: nothing of one may show up in the other
    assert rel_err(got[0]["delta"], got[1]["delta"]) > 1e-2 and float((got[0]["rast"] - got[1]["rast"]).abs().max()) > 1e-3
    log_parity("two frames of the two-hand mesh (V = %d, %d x %dx%d, pitch 384, binned) on two HIP streams vs one after the other, %d fused-Adam "
               "steps: face ids equal, losses / delta / albedo / gdelta rel %s | %s" % (V, n, H, W, steps, ["%.1e" % x for x in errs[0]], ["%.1e" % x for x in errs[1]]))


@pytest.mark.parametrize("case", [("micro_sel", (7, 192, 176, 64, 96), 51, [5, 0, 3]),          # W = 176: row pitch 192, binned rasterizer
                                  ("coarse_sel", (6, 128, 334, 10, 14), 52, [4, 1, 5, 2]),      # W = 334: pitch 384, atomic keys + resolve pass, large-box queue
                                  ("flat512_sel", (5, 64, 128, 24, 32), 53, [2, 4])],           # pitch == W (a multiple of 64): the cidx == gid fast path is left
                         ids=lambda c: c[0])
def test_view_selection_equals_the_gathered_batch(oracle, case):
    """ABI 6: the loop's minibatch — `perm[k:k+batch]` of mesh_sfs_optim.py:249,255-260 — INSIDE the step.  The
    capture's per-view tensors (sh, w2c, proj, img, mask, valid_mask: all `num` views) stay resident, a device int32 array says which views
    the step renders, and the kernels index through it.  A permuted subset through HamStage2(n_src=num).step(view_idx=...) against (a) the
    oracle on the GATHERED inputs under the same bars as every other case and (b) the same step on gathered tensors: `rast` bit-identical,
    losses and gradients equal up to the float-atomic order.  Then a SECOND batch on the same workspace (other views, same object):
    the per-segment valid_mask table was built once for all source views and must serve it."""
    from fmhr_amd.ham import HamStage2
    name, dims, seed, pick = case
    sc = prepare(oracle, dims, seed)
    num = sc["views"]
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    hs = HamStage2(sc["faces"].to(dev), sc["V"], len(pick), sc["H"], sc["W"], sc["weights"], sc["elm"], n_src=num)
    for picked in (pick, [v for v in range(num) if v not in pick][:len(pick)][::-1] or pick):
        if len(picked) != len(pick):
            continue
        idx = torch.tensor(picked, dtype=torch.int32, device=dev)
        sub = dict(sc, views=len(picked))
        for k in ("sh0", "w2c", "proj", "img", "mask", "valid_mask"):
            sub[k] = sc[k][picked].contiguous()
        ld_ref, v = _step_vs_oracle(oracle, sub, f"{name}{picked}")            # the gathered batch: GPU vs oracle, hard 1e-4
        losses, gd, ga = hs.step(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"],
                                 view_idx=idx)
        ld = hs.loss_dict()
        vs = hs.views()
        assert torch.equal(vs["rast"], v["rast"]), picked
        assert torch.equal(vs["tmp_img_pre_aa"], v["tmp_img_pre_aa"]), picked   # the SH rows went through the selection too
        assert ld["n_valid"] == ld_ref["n_valid"]
        for k in ("sfs", "lap", "albedo", "mask", "edge", "delta", "total"):
            assert abs(ld[k] - ld_ref[k]) <= 2e-6 * max(abs(ld_ref[k]), 1e-6), (picked, k, ld[k], ld_ref[k])
        e = (rel_err(gd, v["gdelta"]), rel_err(ga, v["galbedo"]), rel_err(gd, v["gd64"]), rel_err(ga, v["ga64"]))
        log_parity(f"{name}: views {picked} of {num} selected inside the step vs the gathered batch: rast bit-identical, gdelta {e[0]:.1e} galbedo {e[1]:.1e}"
                   f" | vs fp64 on the gathered inputs: {e[2]:.1e} {e[3]:.1e}")
        assert e[0] < 2e-5 and e[1] < 2e-5 and e[2] < TOL and e[3] < TOL, (picked, e)


def test_view_selection_argument_errors(oracle):
    """view_idx goes with n_src, both ways; a selection needs every 64-pixel segment inside one view (H * pitch % 64 == 0)"""
    from fmhr_amd.ham import HamStage2
    sc = prepare(oracle, (3, 64, 64, 10, 14), 54)
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    args = (g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    idx = torch.tensor([2, 0], dtype=torch.int32, device=dev)
    plain = HamStage2(sc["faces"].to(dev), sc["V"], 3, 64, 64, sc["weights"], sc["elm"])
    with pytest.raises(RuntimeError, match="view_idx"):
        plain.step(*args, view_idx=idx)
    sel = HamStage2(sc["faces"].to(dev), sc["V"], 2, 64, 64, sc["weights"], sc["elm"], n_src=3)
    with pytest.raises(RuntimeError, match="view_idx"):
        sel.step(*args)
    with pytest.raises(RuntimeError, match="int32"):
        sel.step(*args, view_idx=idx.long())
    with pytest.raises(RuntimeError, match="multiple of 64"):                    # 50 x 38 pixels at pitch = W: segments straddle views
        HamStage2(sc["faces"].to(dev), sc["V"], 2, 50, 38, sc["weights"], sc["elm"], n_src=3, pitch=38)
    sel.step(*args, view_idx=idx)                                                 # ... and the well-formed call runs
    assert sel.loss_dict()["n_valid"] > 0


def test_view_index_out_of_range_is_clamped_and_reported(oracle):
    """view_idx is dereferenced on the device.  An index outside [0, n_src) must never read out of bounds: the geometry kernel
    clamps it (the step then renders the clamped view — exactly, for every later kernel reads the clamped copy) and sets a sticky word;
    check_overflow() / loss_dict() raise.  A well-formed selection on the same workspace afterwards still reports the earlier fault."""
    from fmhr_amd.ham import HamStage2
    sc = prepare(oracle, (3, 64, 64, 10, 14), 55)
    dev = torch.device("cuda:0")
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    args = (g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
    good = HamStage2(sc["faces"].to(dev), sc["V"], 2, 64, 64, sc["weights"], sc["elm"], n_src=3)
    good.step(*args, view_idx=torch.tensor([2, 0], dtype=torch.int32, device=dev))
    ref = (good.losses.clone(), good.gdelta.clone(), good.galbedo.clone())
    good.check_overflow()
    bad = HamStage2(sc["faces"].to(dev), sc["V"], 2, 64, 64, sc["weights"], sc["elm"], n_src=3)
    bad.step(*args, view_idx=torch.tensor([7, -4], dtype=torch.int32, device=dev))          # clamped to views 2 and 0
    torch.cuda.synchronize()
    assert float(bad.losses[7]) == float(ref[0][7]) and rel_err(bad.losses[:7], ref[0][:7]) < 1e-6      # (float atomics: not the same bits)
    assert rel_err(bad.gdelta, ref[1]) < 1e-5 and rel_err(bad.galbedo, ref[2]) < 1e-5
    with pytest.raises(RuntimeError, match="view index outside"):
        bad.check_overflow()
    bad.step(*args, view_idx=torch.tensor([2, 0], dtype=torch.int32, device=dev))
    with pytest.raises(RuntimeError, match="view index outside"):                              # sticky
        bad.loss_dict()
