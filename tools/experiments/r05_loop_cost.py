"""Round 5: where the product's stage-2 loop (pipeline.Stage2Loop) loses time against the bare step, metric config.
    gpurun -- python tools/experiments/r05_loop_cost.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fmhr_amd import synthetic, pipeline                                          # noqa: E402
from fmhr_amd.ham import VertexOrder, edge_length_mean, render_forward          # noqa: E402
from fmhr_amd.optim import Adam                                                  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    sc = synthetic.scene_from_config(sys.argv[1] if len(sys.argv) > 1 else "metric", 0, None)
    n, H, W, V = sc["views"], sc["H"], sc["W"], sc["V"]
    order = VertexOrder(sc["faces"], V)
    for k in ("vertices", "vertices_true", "albedo_true", "albedo0"):
        sc[k] = order.to_internal(sc[k])
    sc["faces"] = order.faces
    faces = sc["faces"].to(dev)
    w2c, proj, sh = sc["w2c"].to(dev).contiguous(), sc["proj"].to(dev).contiguous(), sc["sh0"].to(dev).contiguous()
    vt = sc["vertices"].to(dev).contiguous()
    tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
    elm = edge_length_mean(vt, faces)
    from fmhr_amd.ham import HamStage2

    def timed(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            fn()
        t1 = time.perf_counter()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3, 1e6 * (t1 - t0) / reps

    def fresh():
        delta = torch.zeros(V, 3, device=dev, requires_grad=True)
        albedo = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
        return delta, albedo, Adam([{'params': delta, 'lr': sc["lr"]["lr"]}, {'params': albedo, 'lr': sc["lr"]["albedo_lr"]}])
    runs = {}
    d0, a0, o0 = fresh()
    hs0 = HamStage2(faces, V, n, H, W, sc["weights"], elm)
    runs["bare step, fixed tensors"] = lambda: hs0.step(vt, d0.detach(), a0.detach()[0], sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"], optimizer=o0, params=(d0, a0))
    d1, a1, o1 = fresh()
    hs1 = HamStage2(faces, V, n, H, W, sc["weights"], elm, n_src=n)
    perms = torch.stack([torch.randperm(n, generator=torch.Generator().manual_seed(1)) for _ in range(512)]).to(torch.int32).to(dev)
    it = [0]

    def sel():
        it[0] += 1
        hs1.step(vt, d1.detach(), a1.detach()[0], sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"], optimizer=o1, params=(d1, a1), view_idx=perms[it[0] % 512])
    runs["bare step, fresh permutation inside the step"] = sel
    for name, kw in (("Stage2Loop (poll every 16 epochs)", {}), ("Stage2Loop, check_every=0", dict(check_every=0))):
        d2, a2, o2 = fresh()
        lp = pipeline.Stage2Loop(vt, faces, d2, a2, sh, tg["img"], tg["mask"], tg["valid_mask"], w2c, proj, (H, W), sc["weights"], elm, o2, n, 1 << 30,
                                 torch.Generator().manual_seed(0), **kw)
        runs[name] = lp.run_epoch
    for fn in runs.values():
        for _ in range(192):
            fn()
    acc = {k: [0.0, 0.0] for k in runs}
    for rnd in range(4):
        for k, fn in runs.items():
            d, h = timed(fn, 64)
            acc[k][0] += d / 4
            acc[k][1] += h / 4
    for k, (d, h) in acc.items():
        print(f"{k:48s} {d:7.1f} us per iteration (device), host enqueue {h:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
