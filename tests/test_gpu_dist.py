"""GPU: the HIP runner under grid sharding (SURVEY §4 "Distributed" row).  Only one GPU is reachable here, so world size 2
is rehearsed as two child processes that share cuda:0 and gather over gloo; the result must be bit-identical to world
size 1 — same kernels, same (seed, image, trial) noise, fixed-order mean / top-k on the gathered slab."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "hip_shard_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, tmp_path, arch="small"):
    port = _free_port()
    outs = [str(tmp_path / f"{arch}_w{world}_r{r}.npz") for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), outs[r], arch], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    return [dict(np.load(o)) for o in outs]


def test_hip_runner_world_size_2_is_bit_identical_to_world_size_1(tmp_path):
    one = _launch(1, tmp_path)[0]
    two = _launch(2, tmp_path)
    assert int(one["hip"][0]) == 1
    pruned = np.isinf(one["err"])
    assert pruned.any() and not pruned.all()                       # two-stage pruning really left cells unevaluated
    for r in two:
        for k in ("lab", "err", "lab_p", "err_p"):
            np.testing.assert_array_equal(r[k], one[k])            # bit-identical errors and labels on every rank


def test_cfg2_architecture_bf16_world_size_3_is_bit_identical_to_world_size_1(tmp_path):
    """BASELINE config 2's UNet in bf16 with 10 classes (class-shared trunk and skip halves in the plan), two stages, the 21 + 12
    (trial, image) pairs of the stages dealt to THREE ranks (uneven shares): every rank must end with the single-process errors
    and labels, bit for bit."""
    one = _launch(1, tmp_path, "cfg2")[0]
    three = _launch(3, tmp_path, "cfg2")
    assert np.isinf(one["err"]).any() and np.isfinite(one["err"]).any()
    for r in three:
        for k in ("lab", "err", "lab_p", "err_p"):
            np.testing.assert_array_equal(r[k], one[k])


def test_bench_harness_with_three_ranks_rehearsed_on_one_gpu(tmp_path):
    """`python bench.py --gpus 3` exactly as typed by hand — the parent starts torch.distributed.run as a child, the ranks rendezvous on
    127.0.0.1, shard the grid, take the max time over ranks, rank 0 prints ONE JSON line — rehearsed with DCAMD_BENCH_REHEARSE=1 (ranks
    share cuda:0 and talk over gloo, since RCCL refuses two ranks on one device).  The line must carry the contract's fields and say
    that it is a rehearsal."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, DCAMD_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1", "--workload", "small-unet-2x8"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["value"] > 0 and d["config"]["parallelism"].startswith("grid-shard x3") and d["config"]["images_per_step"] == 24
    assert "REHEARSAL" in d["data"] and "roofline" in d and "cpu_baseline" not in d
