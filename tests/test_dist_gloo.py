"""CPU, world_size 2 (gloo): the N>1 path of classify — pair sharding, the one all-gather per
stage, identical labels on every rank and equal to the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, q, mode="shard"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import diffusion_classifier_amd as dca
    from helpers import load_case, standin_from
    g, cfg = load_case(name)
    if mode == "shard":
        cfg["shard_grid"] = True                         # grid sharding is opt-in
    dc = dca.DiffusionClassifier(standin_from(g, cfg), dca.Config(**cfg))
    if dc.encoder is not None:
        dc.encoder.weight.data.copy_(torch.from_numpy(g["encoder.weight"]))
    fast = bool(g["fast"])
    x = torch.from_numpy(g["x"])
    kw = dict(t=torch.from_numpy(g["t"]), eps=torch.from_numpy(g["eps"]),
              fast_select=torch.from_numpy(g["fast_select"]) if fast else None, return_errors=True)
    lab = torch.from_numpy(g["labels"]) if fast else None
    if mode == "history":
        # the decision to compare the batch across ranks must not depend on anything rank-local.  Rank 0 allocates and
        # frees junk between calls (another allocator history, other data_ptr values) and re-uses one buffer; rank 1 feeds fresh
        # tensors.  Every rank must make the same collectives in the same order: five sharded classify calls complete with the
        # golden labels, the checks made are the same calls on both ranks, and a batch that differs on a CHECKED call is refused.
        from diffusion_classifier_amd import dist as D
        made, buf = [], x.clone()
        for i in range(5):
            junk = [torch.empty(1000 + 977 * i * (rank + 1)) for _ in range(3 * (1 - rank))]
            xi = buf if rank == 0 else x.clone()
            n0 = D._replicated_calls.get(id(dist.group.WORLD), 0)
            out, err = dc.classify(xi, lab, fast=fast, group=dist.group.WORLD, **kw)
            made.append(n0 < D.REPLICATED_CHECK_FIRST or n0 % D.REPLICATED_CHECK_EVERY == 0)
            del junk
        D._replicated_calls[id(dist.group.WORLD)] = D.REPLICATED_CHECK_EVERY        # the next call is a checked one on every rank
        try:
            dc.classify(x + rank, lab, fast=fast, group=dist.group.WORLD, **kw)
            verdict = "no error"
        except RuntimeError as e:
            verdict = "refused" if "identical image batch" in str(e) else repr(e)
        q.put((rank, out.numpy(), (made, verdict)))
    elif mode == "mismatch":
        # an accelerate-style launch (each rank holds its own images) with grid sharding requested: must be refused
        try:
            dc.classify(x + rank, lab, fast=fast, group=dist.group.WORLD, **kw)
            q.put((rank, "no error", None))
        except RuntimeError as e:
            q.put((rank, "refused" if "identical image batch" in str(e) else repr(e), None))
    else:
        # mode "off": a process group exists but sharding was not asked for -> every rank scores its whole batch alone
        out, err = dc.classify(x, lab, fast=fast, **kw)
        q.put((rank, out.numpy(), err.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("2stage_pruned", 2), ("1stage_eps", 3)])
def test_sharded_classify_equals_single_process(name, world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load_case
    g, _ = load_case(name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, err in res:
        np.testing.assert_array_equal(out, g["out"])          # same labels on every rank == reference golden
        fin = np.isfinite(g["errors"])
        assert np.array_equal(np.isfinite(err), fin)
        np.testing.assert_allclose(err[fin], g["errors"][fin], rtol=3e-7)   # sub-batches differ from the reference's batch of BS


def _spawn(world, name, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_grid_sharding_is_opt_in_and_checks_that_x_is_replicated():
    """A drop-in script launched by accelerate has a process group AND per-rank images (reference :615-617): classify must
    not shard the grid by default, and when asked to it must refuse batches that differ across ranks."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load_case
    g, _ = load_case("1stage_eps")
    for rank, out, err in _spawn(2, "1stage_eps", "off"):
        np.testing.assert_array_equal(out, g["out"])
        np.testing.assert_array_equal(err, g["errors"])       # the full grid, computed locally, bit-equal to the reference
    for rank, verdict, _ in _spawn(2, "1stage_eps", "mismatch"):
        assert verdict == "refused", verdict


def test_replicated_batch_check_is_a_rank_invariant_decision():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load_case
    g, _ = load_case("1stage_eps")
    res = _spawn(2, "1stage_eps", "history")
    for rank, out, (made, verdict) in res:
        np.testing.assert_array_equal(out, g["out"])
        assert made == [True, True, False, False, False], (rank, made)
        assert verdict == "refused", (rank, verdict)


def test_pair_ownership_is_a_balanced_partition():
    from diffusion_classifier_amd import dist as D
    for (s0, s1, BS, world) in [(0, 50, 8, 8), (3, 10, 5, 4), (0, 1, 1, 8), (0, 100, 16, 8)]:
        allp = D.stage_pairs(s0, s1, BS)
        got = [D.local_pairs(s0, s1, BS, r, world) for r in range(world)]
        assert sorted(sum(got, [])) == sorted(allp)
        sizes = [len(x) for x in got]
        assert max(sizes) - min(sizes) <= 1 and max(sizes) == D.slab_len(s0, s1, BS, world)
