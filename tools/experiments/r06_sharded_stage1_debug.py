"""Where does the view-sharded stage 1 leave the single-rank one on the reference's capture?  Two ranks on one GPU (gloo); rank 0 also runs the
single-rank fused step on all 16 views, step by step on the same masks, and prints the differences of the gradient and of the parameters.
    gpurun -- 'python tools/experiments/r06_sharded_stage1_debug.py'"""
import os, socket, sys
import torch, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from fmhr_amd import dist as fdist, optim, stage1 as s1
    from fmhr_amd.pipeline import PermChunks
    from fmhr_amd.ham import VertexOrder
    from test_gpu_sharded import _capture_problem
    fdist.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    cap, vertices, faces, conf = _capture_problem()
    order = VertexOrder(faces, vertices.shape[0])
    vertices, faces = order.to_internal(vertices), order.faces
    num, batch = 16, 8
    v0, v1 = fdist.shard_views(num, rank, world)
    hs1 = s1.HamStage1(vertices, faces, cap["w2cs"][v0:v1].contiguous(), cap["projs"][v0:v1].contiguous(), cap["imgs"][v0:v1].contiguous(),
                       cap["masks"][v0:v1].contiguous(), cap["res"], conf["sfs_weight"], want_pred_masks=True)
    init = s1.init_sh(hs1, cap["grayimgs"][v0:v1].contiguous(), group=dist.group.WORLD, view0=v0, n_views_global=num)
    sh_local = init["sh_coeffs"].clone().requires_grad_(True)
    albedo = (torch.zeros_like(vertices) + init["albedo_mean"]).unsqueeze(0).contiguous().requires_grad_(True)
    opt = optim.Adam([{'params': albedo, 'lr': conf["albedo_lr"]}, {'params': sh_local, 'lr': conf["sh_lr"]}])
    sh1 = fdist.ShardedStage1(hs1, v0, num, group=dist.group.WORLD)
    gen = torch.Generator(device="cpu").manual_seed(5)
    perms = PermChunks(num, batch, gen, dev)
    if rank == 0:
        full = s1.HamStage1(vertices, faces, cap["w2cs"], cap["projs"], cap["imgs"], cap["masks"], cap["res"], conf["sfs_weight"], want_pred_masks=True)
        finit = s1.init_sh(full, cap["grayimgs"])
        fsh = finit["sh_coeffs"].clone().requires_grad_(True)
        falb = (torch.zeros_like(vertices) + finit["albedo_mean"]).unsqueeze(0).contiguous().requires_grad_(True)
        fopt = optim.Adam([{'params': falb, 'lr': conf["albedo_lr"]}, {'params': fsh, 'lr': conf["sh_lr"]}])
        print("init: albedo_mean diff", float((finit["albedo_mean"] - init["albedo_mean"]).abs().max()), "sh rows", float((finit["sh_coeffs"][v0:v1] - init["sh_coeffs"]).abs().max()),
              "valid_masks", float((finit["valid_masks"] - init["valid_masks"]).abs().max()), "counts", (full.view_valid[v0:v1] - hs1.view_valid).abs().max().item(), flush=True)
    for ep in range(3):
        _, masks = perms.next()
        for j in range(masks.shape[0]):
            loss, galb, gsh = sh1.step_sel(albedo, sh_local, masks[j])
            if rank == 0:
                lv, fg, fgs = full.step(falb, fsh, sel=masks[j])            # gradients only (no optimiser): compare, then apply with the same rule
                gd = float((fg - galb).abs().max() / fg.abs().max())
                gsd = float((fgs[v0:v1] - gsh).abs().max() / fgs.abs().max().clamp_min(1e-30))
                ld = abs(float(lv.sum()) - float(loss)) / abs(float(lv.sum()))
                falb.grad, fsh.grad = fg.clone().view_as(falb), fgs.clone()
                fopt.step()
            albedo.grad, sh_local.grad = galb.view_as(albedo), gsh
            opt.step()
            if rank == 0:
                ad = float((falb - albedo).abs().max() / falb.abs().max())
                sd = float((fsh[v0:v1] - sh_local).abs().max() / fsh.abs().max())
                big = int(((falb - albedo).abs() > 1e-4 * falb.abs().max()).sum())
                print(f"epoch {ep} batch {j}: loss rel diff {ld:.2e} galbedo {gd:.2e} gsh(local rows) {gsd:.2e} | after Adam: albedo {ad:.2e} ({big} components beyond 1e-4) sh {sd:.2e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, port)) for r in range(2)]
    [p.start() for p in ps]
    [p.join(600) for p in ps]
    sys.exit(max(p.exitcode or 0 for p in ps))
