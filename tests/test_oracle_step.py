"""The oracle's stage-2 step (mesh_sfs_optim.py:253-306 restated on the oracle ops): fp64 directional finite differences of
the total loss w.r.t. delta and albedo, and fp32-vs-fp64 agreement — this is the reference the fused HIP step is graded on."""
import numpy as np
import torch

from util import mesh_scene, rel_err
from checker_scenes import prepare


def test_step_fd_and_precision(oracle):
    sc = prepare(oracle, (2, 96, 80, 10, 14), 3)
    w = dict(sc["weights"])
    w["delta"] = 5e6          # make every term visible at this tiny scale
    f64 = {k: sc[k].double() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    elm = float(oracle.edge_length_mean(f64["vertices"], sc["faces"])) * 0.9

    def loss_fn(d, a, t=f64, den=None):
        return oracle.ham_stage2_losses(t["vertices"], d, a[None], t["sh0"], t["w2c"], t["proj"], t["img"], t["mask"],
                                        t["valid_mask"], sc["faces"], (sc["H"], sc["W"]), w, elm)

    d = f64["delta"].clone().requires_grad_(True)
    a = f64["albedo"].clone().requires_grad_(True)
    out = loss_fn(d, a)
    out["loss"].backward()
    assert all(float(out[k]) > 0 for k in ("sfs", "lap", "albedo", "mask", "edge", "delta"))
    rng = np.random.default_rng(0)
    ids0, n0 = out["rast"][..., 3].clone(), out["n_valid"]
    checked = 0
    for _ in range(12):
        dd = torch.tensor(rng.standard_normal(d.shape)) * 1e-8
        da = torch.tensor(rng.standard_normal(a.shape)) * 1e-6
        with torch.no_grad():
            p, m = loss_fn(d + dd, a + da), loss_fn(d - dd, a - da)
        if not (torch.equal(p["rast"][..., 3], ids0) and torch.equal(m["rast"][..., 3], ids0)):
            continue
        num = float(p["loss"] - m["loss"]) / 2
        ana = float((d.grad * dd).sum() + (a.grad * da).sum())
        assert abs(num - ana) <= 1e-3 * abs(ana) + 5e-9, (num, ana)   # L1 / hinge kinks crossed by the step add O(h) noise
        checked += 1
    assert checked >= 4
    # fp32 twin agrees with fp64 (same decisions on this scene)
    d32 = sc["delta"].clone().requires_grad_(True)
    a32 = sc["albedo"].clone().requires_grad_(True)
    t32 = {k: sc[k] for k in f64}
    o32 = loss_fn(d32, a32, t32)
    o32["loss"].backward()
    assert abs(float(o32["loss"]) - float(out["loss"])) < 1e-4 * float(out["loss"])
    assert rel_err(d32.grad, d.grad) < 5e-3 and rel_err(a32.grad, a.grad) < 5e-3


def test_loop_body_decisions_are_taken_from_the_fp32_run(oracle):
    """The loop body's own discrete decisions — the sign of every L1 residual (:289) and the active set of the edge hinge (:302) — can
    be handed from an fp32 run to the fp64 run ("fp64 values on fp32 decisions"): with the run's own decisions nothing changes; with
    one hinge entry forced the other way exactly that edge's gradient share appears / disappears."""
    sc = prepare(oracle, (2, 64, 48, 8, 10), 4)
    f64 = {k: sc[k].double() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
    elm = float(oracle.edge_length_mean(f64["vertices"], sc["faces"]))

    def run(t, dec=None):
        d = t["delta"].clone().requires_grad_(True)
        a = t["albedo"].clone().requires_grad_(True)
        o = oracle.ham_stage2_losses(t["vertices"], d, a[None], t["sh0"], t["w2c"], t["proj"], t["img"], t["mask"], t["valid_mask"],
                                     sc["faces"], (sc["H"], sc["W"]), sc["weights"], elm, decisions=dec)
        o["loss"].backward()
        return o, d.grad, a.grad

    o0, gd0, ga0 = run(f64)
    o1, gd1, ga1 = run(f64, o0["decisions"])
    assert float(o1["loss"]) == float(o0["loss"]) and torch.equal(gd0, gd1) and torch.equal(ga0, ga1)
    dec = dict(o0["decisions"], hinge_active=o0["decisions"]["hinge_active"].clone())
    k = int(torch.nonzero(dec["hinge_active"])[0])
    dec["hinge_active"][k] = False
    o2, gd2, _ = run(f64, dec)
    F3 = 3 * sc["F"]
    fl = sc["faces"].long()
    ends = [(0, 1), (2, 1), (0, 2)][k // sc["F"]]                        # order of the concatenation in :296-301
    i, j = int(fl[k % sc["F"], ends[0]]), int(fl[k % sc["F"], ends[1]])
    x = f64["vertices"] + f64["delta"]
    share = 2.0 * (x[i] - x[j]) * sc["weights"]["edge"] / F3
    diff = gd0 - gd2
    assert torch.allclose(diff[i], share, rtol=1e-9, atol=1e-12) and torch.allclose(diff[j], -share, rtol=1e-9, atol=1e-12)
    touched = torch.zeros(sc["V"], dtype=torch.bool); touched[i] = touched[j] = True
    assert float(diff[~touched].abs().max()) == 0.0
    # the fp32 twin's decisions on the fp64 values: where both agree this is the plain fp64 result
    with oracle.f32_decisions():
        o32, _, _ = run({k_: sc[k_] for k_ in f64})
        o3, gd3, _ = run(f64, o32["decisions"])
    same = torch.equal(o32["decisions"]["hinge_active"], o0["decisions"]["hinge_active"]) and \
        torch.equal(o32["decisions"]["l1_sign"].double(), o0["decisions"]["l1_sign"])
    if same and torch.equal(o3["rast"][..., 3], o0["rast"][..., 3]):
        assert rel_err(gd3, gd0) < 1e-9


def test_rule_r5_values_in_double_keep_the_decisions_and_meet_the_bar(oracle):
    """Round 5 (rule R5): the fp32 twin evaluates the VALUES of rule R5 — the barycentrics a winner returns and their
    gradient with respect to the clip positions — in double from the fp32 clip positions, its DECISIONS (coverage, accept / reject, z/w,
    depth order) in fp32 as before.  On a micro-triangle scene at a capture-like clip scale: ids and z/w are the same bits under both
    rules, (u, v) agree with exact arithmetic to half an fp32 ulp instead of ~1e-4, and the twin's gradients — beyond the north_star's
    1e-4 with the all-fp32 rule — meet it."""
    lib = oracle.lib()
    sc = prepare(oracle, (2, 192, 192, 64, 96), 2)
    # clip-space scale of the reference capture's cameras: |x| ~ 2 where the synthetic scenes have ~0.3 (an off-centre principal point)
    sc["proj"] = sc["proj"].clone()
    sc["proj"][:, :, 0] += 0.6 * sc["proj"][:, :, 3]
    pos32 = oracle.clip_positions(sc["vertices"] + sc["delta"], sc["w2c"], sc["proj"])[1].contiguous()

    def grads(dtype, decisions=None):
        c = lambda t: t.to(dtype)
        d = c(sc["delta"]).clone().requires_grad_(True)
        a = c(sc["albedo"])[None].clone().requires_grad_(True)
        ref = oracle.ham_stage2_losses(c(sc["vertices"]), d, a, c(sc["sh0"]), c(sc["w2c"]), c(sc["proj"]), c(sc["img"]), c(sc["mask"]),
                                       c(sc["valid_mask"]), sc["faces"], (sc["H"], sc["W"]), sc["weights"], sc["elm"], pos_value=pos32,
                                       decisions=decisions)
        ref["loss"].backward()
        return ref, d.grad, a.grad[0]

    res = {}
    try:
        for wide in (0, 1):
            lib.orc_set_r5_wide(wide)
            r32, gd32, ga32 = grads(torch.float32)
            with oracle.f32_decisions():
                r64, gd64, ga64 = grads(torch.float64, r32["decisions"])
            res[wide] = dict(rast=r32["rast"], uv=float((r32["rast"][..., :2].double() - r64["rast"][..., :2]).abs().max()),
                             e_d=rel_err(gd32, gd64), e_a=rel_err(ga32, ga64))
    finally:
        lib.orc_set_r5_wide(1)
    assert lib.orc_get_r5_wide() == 1                                   # the rule in force
    assert torch.equal(res[0]["rast"][..., 2:], res[1]["rast"][..., 2:])  # decisions: the same bits
    assert float((res[1]["rast"][..., 3] > 0).float().mean()) > 0.05
    assert res[1]["uv"] <= 6e-8 < res[0]["uv"], (res[0]["uv"], res[1]["uv"])
    assert res[1]["e_d"] < 2e-5 and res[1]["e_a"] < 2e-5, res[1]
    assert res[0]["e_d"] > 5 * res[1]["e_d"], (res[0]["e_d"], res[1]["e_d"])
