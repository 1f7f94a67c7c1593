#!/usr/bin/env python3
"""tools/rccl_probe.py — does RCCL (torch.distributed backend "nccl") come up on this box at all?  One rank, world size 1; every
stage leaves a time-stamped line in gpurun_out/rccl_probe.log (a hang is then located by the last line).  Run it under `timeout`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "rccl_probe.log"), "a")
T0 = time.time()


def say(msg):
    LOG.write("%7.2f s  %s\n" % (time.time() - T0, msg))
    LOG.flush()
    os.fsync(LOG.fileno())


mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29577"), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
say(f"mode {mode}; env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith(("NCCL", "RCCL", "HSA", "HIP", "ROCR"))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
say(f"torch {torch.__version__}, cuda available {torch.cuda.is_available()}, devices {torch.cuda.device_count()}")
torch.cuda.set_device(0)
x = torch.ones(1 << 20, device="cuda")
torch.cuda.synchronize()
say("first kernel done")
if mode == "eager":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("nccl", rank=0, world_size=1)
say("init_process_group returned")
dist.all_reduce(x)
say("all_reduce enqueued")
torch.cuda.synchronize()
say("all_reduce complete: x[0] = %g" % float(x[0]))
dist.barrier()
say("barrier returned")
if len(sys.argv) > 2 and sys.argv[2] == "step":
    # the sharded fused step over the backend: where does it stop, if it stops?  (stack of every thread after 40 s without progress)
    import faulthandler
    faulthandler.dump_traceback_later(40, exit=True, file=LOG)
    sys.path.insert(0, ROOT)
    from fmhr_amd import synthetic as S
    from fmhr_amd.ham import HamStage2, edge_length_mean, render_forward
    from fmhr_amd.optim import Adam
    dev = torch.device("cuda:0")
    sc = S.make_scene(6, 128, 96, 21, 37, 1, seed=3)
    faces = sc["faces"].to(dev)
    w2c, proj = sc["w2c"].to(dev).contiguous(), sc["proj"].to(dev).contiguous()
    tg = S.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, sc["H"], sc["W"]), device=dev)
    vt = sc["vertices"].to(dev)
    say("scene built")
    hs = HamStage2(faces, sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], edge_length_mean(vt, faces))
    delta = torch.zeros(sc["V"], 3, device=dev, requires_grad=True)
    albedo = sc["albedo0"].to(dev).clone().requires_grad_(True)
    opt = Adam([{'params': delta, 'lr': 1e-4}, {'params': albedo, 'lr': 5e-3}])
    use_opt = os.environ.get("PROBE_OPT", "1") == "1"
    for i in range(4):
        a = (vt, delta.detach(), albedo.detach(), sc["sh0"].to(dev), w2c, proj)
        hs.forward(*a, tg["mask"], tg["valid_mask"])
        say(f"step {i}: forward enqueued"); torch.cuda.synchronize(); say("  ... complete")
        hs.backward(*a, tg["img"], tg["mask"], tg["valid_mask"], reg_range=(0, sc["V"]), split=True)
        say("  backward enqueued"); torch.cuda.synchronize(); say("  ... complete")
        dist.all_reduce(hs.bucket, group=dist.group.WORLD)
        say("  all_reduce enqueued"); torch.cuda.synchronize(); say("  ... complete")
        hs.combine((opt.fused_param(delta), opt.fused_param(albedo)) if use_opt else (None, None))
        say("  combine enqueued"); torch.cuda.synchronize(); say("  ... complete: loss %g" % float(hs.losses[6]))
    for i in range(8):
        hs.step(*a, tg["img"], tg["mask"], tg["valid_mask"], group=dist.group.WORLD, reg_range=(0, sc["V"]),
                optimizer=opt if use_opt else None, params=(delta, albedo) if use_opt else None)
    say("8 unsynchronised steps enqueued"); torch.cuda.synchronize(); say("  ... complete: loss %g" % float(hs.losses[6]))
    faulthandler.cancel_dump_traceback_later()
if len(sys.argv) > 2 and sys.argv[2] == "direct":
    # a second communicator inside the same RCCL instance, driven through ctypes on the caller's stream (fmhr_amd.dist.StreamAllReduce)
    import traceback
    sys.path.insert(0, ROOT)
    from fmhr_amd import dist as fdist
    try:
        sar = fdist.StreamAllReduce(None)
        say("StreamAllReduce constructed")
        y = torch.arange(1000, device="cuda", dtype=torch.float32)
        sar.all_reduce_(y)
        torch.cuda.synchronize()
        say("direct all-reduce complete: y[999] = %g" % float(y[999]))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        big = torch.zeros(12 * 49281 + 8, device="cuda")
        for name, fn in (("direct", lambda: sar.all_reduce_(big)), ("torch", lambda: dist.all_reduce(big))):
            for _ in range(20):
                fn(); big.add_(1.0)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(200):
                fn(); big.add_(1.0)            # a kernel of the same stream on either side of the call
            e1.record()
            torch.cuda.synchronize()
            say("%s: %.2f us per (all-reduce + one elementwise kernel), world of one" % (name, e0.elapsed_time(e1) * 5.0))
        sar.close()
    except Exception:  # noqa: BLE001
        say("direct path FAILED:\n" + traceback.format_exc())
dist.destroy_process_group()
say("destroyed; done")
