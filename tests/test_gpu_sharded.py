"""Two ranks on ONE GPU (gloo backend, CUDA tensors): the view-sharded fused step (HamStage2.step(group=...): split backward, ONE
all-reduce of [P | R | losses], fmhr_ham_combine; the view-independent terms split by vertex range) must reproduce the
single-rank full-batch result.  This is the code path bench.py runs per rank over RCCL at N > 1."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    sys.path.insert(0, ROOT)
    from fmhr_amd import synthetic as S
    from fmhr_amd.ham import render_forward
    dev = torch.device("cuda:0")
    sc = S.make_scene(6, 128, 96, 21, 37, 1, seed=3)
    faces = sc["faces"].to(dev)
    w2c, proj = sc["w2c"].to(dev), sc["proj"].to(dev)
    tg = S.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, 128, 96), device=dev)
    torch.manual_seed(0)
    sc["delta"] = (torch.randn(sc["V"], 3) * 2e-4).to(dev)
    return sc, tg, faces, w2c, proj


def _run(sc, tg, faces, w2c, proj, s0, s1, group, reg_range):
    from fmhr_amd.ham import HamStage2, edge_length_mean
    dev = faces.device
    vt = sc["vertices"].to(dev)
    hs = HamStage2(faces, sc["V"], s1 - s0, 128, 96, sc["weights"], edge_length_mean(vt, faces), n_global=sc["views"])
    sl = slice(s0, s1)
    out = hs.step(vt, sc["delta"], sc["albedo0"].to(dev), sc["sh0"][sl].to(dev).contiguous(), w2c[sl].contiguous(),
                  proj[sl].contiguous(), tg["img"][sl].contiguous(), tg["mask"][sl].contiguous(), tg["valid_mask"][sl].contiguous(),
                  group=group, reg_range=reg_range)
    return [t.clone() for t in out]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from fmhr_amd import dist as fdist
    fdist.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    sc, tg, faces, w2c, proj = _scene()
    s0, s1 = fdist.shard_views(sc["views"], rank, world)
    # the view-independent terms are split over the ranks by vertex range
    ls, gd, ga = _run(sc, tg, faces, w2c, proj, s0, s1, dist.group.WORLD, fdist.shard_views(sc["V"], rank, world))
    torch.cuda.synchronize()
    if rank == 0:
        # numpy arrays are pickled by value: tensors would be handed over through a file descriptor of THIS process, which may
        # have exited before the parent unpickles
        q.put((ls.cpu().numpy(), gd.cpu().numpy(), ga.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_match_single_rank():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import rel_err
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ls, gd, ga = (torch.from_numpy(x) for x in q.get(timeout=500))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    sc, tg, faces, w2c, proj = _scene()
    l1, gd1, ga1 = _run(sc, tg, faces, w2c, proj, 0, sc["views"], None, None)
    assert float(ls[7]) == float(l1[7]) and float(l1[7]) > 1000
    assert rel_err(ls[:7], l1[:7].cpu()) < 1e-5, (ls, l1)
    assert rel_err(gd, gd1.cpu()) < 5e-5 and rel_err(ga, ga1.cpu()) < 5e-5
