"""fmhr_amd.dist — view sharding of the HAM step over ranks (one process per GPU, torch.distributed; backend "nccl" is RCCL
on ROCm) — SURVEY.md §8e.  The reference has no multi-GPU code; this is the build's data-parallel axis:

* the n views of a step are split contiguously over G ranks; every rank holds the whole mesh;
* ONE all-reduce(SUM) per step and nothing else.  F.l1_loss divides by the batch-GLOBAL 3*Pv (mesh_sfs_optim.py:289), which a
  rank only knows after an exchange; instead of exchanging the count between forward and backward (a second, blocking
  collective in the middle of a ~0.15 ms step) every rank back-propagates the photometric term with the divisor LEFT OUT and
  keeps that share of the gradient apart from the rest (the photometric gradient only flows through the rgb channels, the
  silhouette MSE only through alpha, so the split is free): bucket = [P_delta | P_albedo | R_delta | R_albedo | 8 loss slots] =
  12V+8 fp32 (2.35 MB at V = 49k: latency-bound on xGMI).  After the all-reduce, g = P / (3 * n_valid_global) + R
  (fmhr_ham_combine), n_valid_global being the summed count slot.  Every entry is the rank's share: the view terms of its
  views and the view-independent terms (regularisers) of its VERTEX range — shard_views(V, rank, world) — so the regulariser
  work is split over the ranks too.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "FMHR_FORCE_DEVICE" in os.environ:            # rehearsal: all ranks on one GPU
        local = int(os.environ["FMHR_FORCE_DEVICE"])
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            # FMHR_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get("FMHR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_views(n_views, rank, world):
    """Contiguous, balanced partition: returns (start, stop) of this rank's views."""
    base, rem = divmod(n_views, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def combine_bucket(bucket, V):
    """The arithmetic of fmhr_ham_combine on any device (the GPU step uses the kernel; this is the host-side statement the CPU
    rehearsal checks): bucket = [P_delta | P_albedo | R_delta | R_albedo | 8 loss slots] summed over the ranks."""
    n3 = 3 * V
    losses = bucket[4 * n3:]
    nv = float(losses[7])
    sc = 1.0 / (3.0 * nv) if nv > 0 else 0.0
    g = bucket[: 2 * n3] * sc + bucket[2 * n3: 4 * n3]
    losses = losses.clone()
    losses[0] = losses[0] * sc
    losses[6] = losses[0] + losses[1] + losses[2] + losses[3] + losses[4] + losses[5]
    return losses, g[:n3].view(V, 3), g[n3:].view(V, 3)


class ShardedStep:
    """The single-collective exchange on plain tensors (any device): pack the rank's shares, all-reduce once, combine.
    HamStage2.step(group=...) does the same in place on its own bucket with the combine kernel; this class is the host-side
    reference of the protocol (tests/test_dist_gloo.py runs it over gloo with the CPU oracle as the compute)."""

    def __init__(self, V, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bucket = torch.zeros(12 * V + 8, dtype=torch.float32, device=device)
        self.V = V

    def reduce(self, losses, p_delta, p_albedo, r_delta, r_albedo):
        """losses[8]: this rank's share of every slot — slot 0 = w_sfs * sum |d| over its valid pixels (no divisor), slot 7 = its LOCAL
        valid-pixel count; p_* = photometric gradient share without the divisor, r_* = everything else.  Returns the batch-global
        (losses, gdelta, galbedo)."""
        V, b = self.V, self.bucket
        n3 = 3 * V
        for k, t in enumerate((p_delta, p_albedo, r_delta, r_albedo)):
            b[k * n3: (k + 1) * n3].copy_(t.reshape(-1))
        b[4 * n3:].copy_(losses)
        if self.world > 1:
            dist.all_reduce(b, group=self.group)
        return combine_bucket(b, V)
