"""(class x trial) grid sharding across the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the
CPU tests).  Within a stage the (trial, image) pairs are dealt round-robin to ranks — pair
g = (j - stage_start)*BS + b belongs to rank g % world — so every rank keeps ALL classes of its
pairs local (q_sample and the class-shared trunk are never duplicated across ranks), the load is
balanced to within one pair whatever T and BS are, and weights are replicated.  The ONLY
data-path exchange is one all-gather of the per-rank error slab [ceil(P/world), classes] per
stage (P = pairs in the stage); every rank then holds the full errors[BS, classes, T] tensor and
performs the identical fixed-order mean / top-k, so labels are bit-identical on all ranks and
for any world size.  (An all-reduce of partial sums would make results depend on the reduction
order.)  The reference itself only shards the dataloader over ranks
(diffusion_classifier.py:615-617) and all-reduces scalar metrics (utils/metrics.py:56-58);
that outer, per-image level composes with this one.
"""
import torch
import torch.distributed as dist


def world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


_replicated_calls = {}   # group -> number of assert_replicated calls made on it (identical on every rank: classify is collective)
REPLICATED_CHECK_FIRST = 2       # the first calls on a group are always compared ...
REPLICATED_CHECK_EVERY = 64      # ... and every 64th after that


def assert_replicated(x, group=None):
    """Grid sharding splits the (trial, image) pairs of ONE batch over the ranks: every rank must hold the same x.
    Compares a cheap fingerprint (shape, sum, sum of squares, a strided sample) across the group, bit for bit (the fp64 words are
    compared as int64, so a replicated batch that contains NaN still passes).  The check costs a pass over x, a small all-gather and
    a host sync, so it is not made on every call — but WHETHER it is made must be the same decision on every rank, or one rank's
    all-gather would pair with a peer's next collective (a key built from data_ptr() is not rank-invariant, allocator
    histories differ).  The decision therefore depends only on how many times this group has been asked: every rank of a group
    calls classify the same number of times (it is a collective), so the counter is the same everywhere.  The first
    REPLICATED_CHECK_FIRST calls and every REPLICATED_CHECK_EVERY-th call after them are compared: the guard is against a
    mis-configured launch (dataloader sharding with grid sharding switched on), which shows on the first batch."""
    k = id(group)
    n = _replicated_calls.get(k, 0)
    _replicated_calls[k] = n + 1
    if n >= REPLICATED_CHECK_FIRST and n % REPLICATED_CHECK_EVERY:
        return False
    xf = x.detach().reshape(-1).to(torch.float64)
    step = max(1, xf.numel() // 61)
    fp = torch.cat([torch.tensor([float(x.shape[0]), float(xf.numel())], dtype=torch.float64, device=xf.device),
                    xf.sum().reshape(1), (xf * xf).sum().reshape(1), xf[::step][:61]]).contiguous().view(torch.int64)
    ws = dist.get_world_size(group)
    if fp.is_cuda and dist.get_backend(group) == "gloo":
        fp = fp.cpu()                    # gloo has no device all-gather
    allfp = torch.empty((ws * fp.numel(),), dtype=fp.dtype, device=fp.device)
    dist.all_gather_into_tensor(allfp, fp.contiguous(), group=group)
    allfp = allfp.view(ws, -1)
    if not bool((allfp == allfp[0:1]).all().item()):
        raise RuntimeError("classify(shard_grid=True / group=...) needs the identical image batch on every rank of the group; "
                           "the ranks hold different x (dataloader-sharded launch?). Leave grid sharding off in that case.")
    return True


def stage_pairs(stage_start, stage_end, BS):
    """All (trial, image) pairs of a stage, trial-major (the reference's loop order)."""
    return [(j, b) for j in range(stage_start, stage_end) for b in range(BS)]


def local_pairs(stage_start, stage_end, BS, rank, world_size):
    """Pairs of the stage owned by `rank`."""
    P = (stage_end - stage_start) * BS
    return [(stage_start + g // BS, g % BS) for g in range(rank, P, world_size)]     # == stage_pairs(...)[rank::world_size]


def slab_len(stage_start, stage_end, BS, world_size):
    return -(-((stage_end - stage_start) * BS) // world_size)


def gather_stage_errors(errors, stage_start, stage_end, rank, world_size, group=None):
    """In place: complete errors[:, :, stage_start:stage_end] on every rank from the owners.

    `errors` is [BS, classes, T]; on entry each rank has filled only the cells of its own pairs.
    """
    if world_size == 1:
        return errors
    BS, ncls = errors.shape[0], errors.shape[1]
    n = slab_len(stage_start, stage_end, BS, world_size)
    P = (stage_end - stage_start) * BS
    dev = errors.device
    # pair g = (j - stage_start) * BS + b lives on rank g % world, row g // world of that rank's slab: index arithmetic on
    # the device instead of Python lists of pairs (6400 pairs x 8 ranks per call at N = 8 sat on the critical path)
    g_mine = torch.arange(rank, P, world_size, device=dev)
    slab = torch.full((n, ncls), float("inf"), dtype=errors.dtype, device=dev)
    if g_mine.numel():
        slab[: g_mine.numel()] = errors[g_mine % BS, :, stage_start + g_mine // BS]
    host = errors.is_cuda and dist.get_backend(group) == "gloo"     # gloo (single-GPU rehearsals, CPU tests) gathers on the host
    flat = torch.empty((world_size * n, ncls), dtype=errors.dtype, device="cpu" if host else dev)   # dim-0 concatenation (gloo needs this form)
    dist.all_gather_into_tensor(flat, slab.cpu() if host else slab.contiguous(), group=group)
    flat = flat.to(dev)
    g = torch.arange(P, device=dev)
    errors[g % BS, :, stage_start + g // BS] = flat[(g % world_size) * n + g // world_size]
    return errors
