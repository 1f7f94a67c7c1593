"""Round 5: what the view selection inside the step (ABI 6) costs, per launch group, at the metric config:
plain step | identity selection (n_src = n, view_idx = arange) | a fresh permutation every step.  HIP events per launch group
(fmhr_timing_attach), median over the instrumented steps, plus the whole step by an event pair around 100 un-instrumented steps.

    gpurun -- python tools/experiments/r05_sel_cost.py [config]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B                                                    # noqa: E402
from fmhr_amd import synthetic                                       # noqa: E402
from fmhr_amd.ham import HamStage2, VertexOrder, edge_length_mean, render_forward   # noqa: E402
from fmhr_amd.optim import Adam                                      # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "metric"
    dev = torch.device("cuda:0")
    sc = synthetic.scene_from_config(cfg, 0, None)
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
    stream = torch.cuda.current_stream()
    gen = torch.Generator().manual_seed(0)
    perms = torch.stack([torch.randperm(n, generator=gen) for _ in range(512)]).to(torch.int32).to(dev)
    for mode in ("plain", "identity", "permuted", "plain", "identity", "permuted"):
        delta = torch.zeros(V, 3, device=dev, requires_grad=True)
        albedo = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
        opt = Adam([{'params': delta, 'lr': sc["lr"]["lr"]}, {'params': albedo, 'lr': sc["lr"]["albedo_lr"]}])
        hs = HamStage2(faces, V, n, H, W, sc["weights"], elm, n_src=None if mode == "plain" else n)
        ident = torch.arange(n, dtype=torch.int32, device=dev)
        it = [0]

        def step():
            idx = None if mode == "plain" else (ident if mode == "identity" else perms[it[0] % 512])
            it[0] += 1
            hs.step(vt, delta.detach(), albedo.detach()[0], sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"], optimizer=opt,
                    params=(delta, albedo), view_idx=idx)
        for _ in range(192):
            step()
        tm = B.Timing(9, stream)
        for i in range(9):
            tm.attach(i)
            step()
        B.Timing.detach()
        torch.cuda.synchronize()
        k = {name: sorted(tm.ms(i, s) for i in range(9))[4] for s, name in enumerate(B.SLOTS)}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            step()
        e1.record()
        torch.cuda.synchronize()
        hs.check_overflow()
        print(f"{mode:9s} step {e0.elapsed_time(e1) * 10:7.1f} us | " + " ".join(f"{a[:10]} {1e3 * b:5.1f}" for a, b in k.items() if b > 0.008), flush=True)


if __name__ == "__main__":
    main()
