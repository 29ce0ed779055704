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
