"""fmhr_amd.dist — view sharding of the HAM step over ranks (one process per GPU, torch.distributed; backend "nccl" is RCCL
on ROCm) — SURVEY.md §8e.  The reference has no multi-GPU code; this is the build's data-parallel axis:

* the n views of a step are split contiguously over G ranks; every rank holds the whole mesh;
* exchange 1: all-reduce(SUM) of the valid-pixel count (F.l1_loss divides by the batch-global 3*Pv, mesh_sfs_optim.py:289);
* exchange 2: ONE all-reduce(SUM) per step, in place on HamStage2's bucket [gdelta | galbedo | 8 loss slots] = 6V+8 fp32
  (1.18 MB at V = 49k: latency-bound on xGMI).  Every entry is the rank's share: the view terms of its views and the
  view-independent terms (regularisers) of its VERTEX range — shard_views(V, rank, world) — so nothing is fixed up afterwards
  and the regulariser work is split over the ranks too.
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


class ShardedStep:
    """Wraps a per-rank step function `local_step(n_valid_reduce, reg_grads) -> (losses[8], gdelta[V,3], galbedo[V,3])`
    and performs the gradient exchange.  `local_step` must call `n_valid_reduce(t)` on its valid-pixel count tensor between
    its forward and backward phases (HamStage2.step does, via `group`)."""

    def __init__(self, V, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # one packed bucket: [gdelta | galbedo | 8 loss slots]
        self.bucket = torch.zeros(2 * V * 3 + 8, dtype=torch.float32, device=device)
        self.V = V

    def reduce(self, losses, gdelta, galbedo):
        """ONE all-reduce(SUM) of [gdelta | galbedo | losses].  Contract: every entry is this rank's SHARE — view terms are
        partial sums over the rank's views, the view-independent terms (regulariser gradients and losses, the n_valid slot) come
        from one rank only (HamStage2.step(reg_grads=rank == 0)) — so the sum is the global result and nothing is fixed up
        afterwards.  When the three tensors already are HamStage2's contiguous bucket the collective runs in place on it (no
        pack / unpack kernels: at 6 views per GPU a step is ~0.16 ms, a dozen tiny kernels would be a third of it)."""
        V = self.V
        es = gdelta.element_size()
        in_place = (gdelta.is_contiguous() and galbedo.is_contiguous() and losses.is_contiguous() and
                    galbedo.data_ptr() == gdelta.data_ptr() + 3 * V * es and losses.data_ptr() == galbedo.data_ptr() + 3 * V * es and
                    gdelta._base is not None and gdelta._base is galbedo._base and gdelta._base is losses._base and
                    gdelta._base.numel() == 6 * V + 8 and gdelta._base.data_ptr() == gdelta.data_ptr())
        if in_place:
            b = gdelta._base
        else:
            b = self.bucket
            b[: V * 3].copy_(gdelta.reshape(-1))
            b[V * 3: 2 * V * 3].copy_(galbedo.reshape(-1))
            b[2 * V * 3:].copy_(losses)
        if self.world > 1:
            dist.all_reduce(b, group=self.group)
        return b[2 * V * 3:], b[: V * 3].view(V, 3), b[V * 3: 2 * V * 3].view(V, 3)
