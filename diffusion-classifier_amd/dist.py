"""(class x trial) grid sharding across the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the
CPU tests).  The trials of a stage are dealt round-robin to ranks — trial j belongs to rank
`(j - stage_start) % world` — so every rank keeps all classes of its (image, trial) pairs local
(q_sample and the class-shared trunk are never duplicated across ranks) and weights are
replicated.  The ONLY data-path exchange is one all-gather of the per-rank error slab
`[BS, classes, ceil(stage_len / world)]` per stage; every rank then holds the full
`errors[BS, classes, T]` tensor and performs the identical fixed-order mean / top-k, so labels
are bit-identical on all ranks and for any world size.  (An all-reduce of partial sums would
make the result depend on the reduction order.)  The reference itself only shards the
dataloader over ranks (diffusion_classifier.py:615-617) and all-reduces scalar metrics
(utils/metrics.py:56-58); that outer, per-image level composes with this one.
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def local_trials(stage_start, stage_end, rank, world_size):
    """Trials of [stage_start, stage_end) owned by `rank`."""
    return list(range(stage_start + rank, stage_end, world_size))


def slab_len(stage_start, stage_end, world_size):
    return -(-(stage_end - stage_start) // world_size)


def gather_stage_errors(errors, stage_start, stage_end, rank, world_size, group=None):
    """In place: fill errors[:, :, stage_start:stage_end] on every rank from the owners.

    `errors` is [BS, classes, T]; on entry each rank has filled only its own trials.
    """
    if world_size == 1:
        return errors
    n = slab_len(stage_start, stage_end, world_size)
    mine = local_trials(stage_start, stage_end, rank, world_size)
    slab = torch.full((errors.shape[0], errors.shape[1], n), float("inf"), dtype=errors.dtype, device=errors.device)
    if mine:
        slab[:, :, : len(mine)] = errors[:, :, mine]
    out = torch.empty((world_size,) + tuple(slab.shape), dtype=errors.dtype, device=errors.device)
    dist.all_gather_into_tensor(out, slab.contiguous(), group=group)
    for r in range(world_size):
        tr = local_trials(stage_start, stage_end, r, world_size)
        if tr:
            errors[:, :, tr] = out[r, :, :, : len(tr)]
    return errors
