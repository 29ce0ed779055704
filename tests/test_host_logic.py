"""CPU: host logic of the product (class lists, draws, scatter, stage pruning), the C-ABI surface,
and the no-fallback rule.  No compute call touches the HIP kernels here."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

import diffusion_classifier_amd as dca
from helpers import CASES, load_case, standin_from

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", CASES)
def test_product_loop_matches_reference_goldens(name):
    """The product's classify driving a user-supplied (foreign) backbone reproduces the vectors
    captured from the reference's own classify bit for bit: same class lists, RNG order, scatter,
    stage mean / top-k."""
    g, cfg = load_case(name)
    dc = dca.DiffusionClassifier(standin_from(g, cfg), dca.Config(**cfg))
    if dc.encoder is not None:
        dc.encoder.weight.data.copy_(torch.from_numpy(g["encoder.weight"]))
    fast = bool(g["fast"])
    lab = torch.from_numpy(g["labels"]) if fast else None
    out, err = dc.classify(torch.from_numpy(g["x"]), lab, fast=fast, t=torch.from_numpy(g["t"]),
                           eps=torch.from_numpy(g["eps"]),
                           fast_select=torch.from_numpy(g["fast_select"]) if fast else None, return_errors=True)
    np.testing.assert_array_equal(out.numpy(), g["out"])
    np.testing.assert_array_equal(err.numpy(), g["errors"])
    torch.manual_seed(int(g["seed"]))                       # the loop's own draws, reference order
    out2 = dc.classify(torch.from_numpy(g["x"]), lab, fast=fast)
    np.testing.assert_array_equal(out2.numpy(), g["out"])


def test_simulated_rank_scores_exactly_that_ranks_share():
    """bench.py --simulate-rank r/N (config key `simulate_rank`): without a process group the call evaluates the (trial, image)
    pairs dist.local_pairs deals to rank r — those cells equal the full run's, every other cell stays +inf."""
    from diffusion_classifier_amd import dist as D
    g, cfg = load_case("1stage_eps")
    x, t, eps = torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), torch.from_numpy(g["eps"])
    T, BS = t.shape
    for r, n in ((0, 3), (2, 3), (1, 2)):
        dc = dca.DiffusionClassifier(standin_from(g, cfg), dca.Config(**dict(cfg, simulate_rank=(r, n))))
        dc.encoder.weight.data.copy_(torch.from_numpy(g["encoder.weight"]))
        _, err = dc.classify(x, t=t, eps=eps, return_errors=True)
        mine = torch.zeros(BS, T, dtype=torch.bool)
        for j, b in D.local_pairs(0, T, BS, r, n):
            mine[b, j] = True
        fin = torch.isfinite(err)
        assert torch.equal(fin, mine[:, None, :].expand_as(fin))
        # (sub-batches of a trial on the CPU stand-in: torch's kernels differ in the last bit with the batch size)
        np.testing.assert_allclose(err[fin].numpy(), torch.from_numpy(g["errors"])[fin].numpy(), rtol=1e-6)


def test_schedule_matches_reference_goldens():
    g = np.load(os.path.join(ROOT, "tests", "golden", "schedules.npz"))
    t = torch.from_numpy(g["t"])
    for nd, im in [(32, 32), (64, 32), (128, 256)]:
        _, cfg = load_case("1stage_eps")
        cfg.update(noise_d=nd, image_size=im)
        dc = dca.DiffusionClassifier(standin_from(*load_case("1stage_eps")), dca.Config(**cfg))
        np.testing.assert_array_equal(dc.logsnr_schedule_cosine(t).numpy(), g[f"cosine_{nd}_{im}"])
        np.testing.assert_array_equal(dc.logsnr_schedule_cosine_shifted(t).numpy(), g[f"shifted_{nd}_{im}"])


def test_classify_asserts_like_reference():
    g, cfg = load_case("1stage_eps")
    for bad in (dict(n_fast_classes=1), dict(n_keep_per_stage=[2]), dict(evaluation_per_stage=[1, 2])):
        c = dict(cfg); c.update(bad)
        dc = dca.DiffusionClassifier(standin_from(g, cfg), dca.Config(**c))
        with pytest.raises(AssertionError):
            dc.classify(torch.from_numpy(g["x"]))
    with pytest.raises(AssertionError):
        dca.DiffusionClassifier(standin_from(g, cfg), dca.Config(**dict(cfg, pred_param="x0")))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "dcamd.h")).read()
    declared = set(re.findall(r"\b(dc_[a-z0-9_]+)\s*\(", hdr)) - {"dc_stream"}
    assert declared == set(dca._lib.EXPORTS), declared ^ set(dca._lib.EXPORTS)
    lib = ctypes.CDLL(dca._lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dc_abi_version() == dca._lib.ABI_VERSION == 4
    lib.dc_arch.restype = ctypes.c_char_p
    assert lib.dc_arch() == b"gfx950"


def test_ctypes_structs_match_header_field_order():
    hdr = open(os.path.join(ROOT, "include", "dcamd.h")).read()

    def fields(struct):
        body = re.search(r"typedef struct \{([^{}]*)\} " + struct + ";", hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            parts = decl.split(",")
            first = parts[0].split()[-1].lstrip("*")
            names.append(first)
            names += [p.strip().lstrip("*") for p in parts[1:]]
        return names
    for struct, cls in [("dc_qsample_params", dca._lib.QsampleParams), ("dc_sinusoid_params", dca._lib.SinusoidParams),
                        ("dc_igemm_params", dca._lib.IgemmParams), ("dc_groupnorm_params", dca._lib.GroupnormParams),
                        ("dc_layernorm_params", dca._lib.LayernormParams), ("dc_attention_params", dca._lib.AttentionParams),
                        ("dc_eps_mse_params", dca._lib.EpsMseParams), ("dc_ddpm_step_params", dca._lib.DdpmStepParams),
                        ("dc_op", dca._lib.Op)]:
        assert fields(struct) == [n for n, _ in cls._fields_], struct


def test_arg_validation_needs_no_gpu():
    lib = dca._lib.lib()
    p = dca._lib.IgemmParams(dtype=1, taps=5)
    assert lib.dc_igemm(p, None) == -1
    assert b"taps" in lib.dc_last_error()
    assert lib.dc_igemm_cout_pad(3, 32) == 32 and lib.dc_igemm_cout_pad(130, 128) == 256
    assert lib.dc_run_plan(None, 0, None) == 0


def test_no_cpu_fallback_for_hip_backbones():
    m = dca.UNetCondition2D(**dca.small_unet_kwargs())
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dca._lib.DcamdError):
        m(torch.zeros(1, 3, 32, 32), torch.zeros(1), encoder_hidden_states=torch.zeros(1, 1, 64))
    cfg = dict(load_case("1stage_eps")[1], classes=3)
    dc = dca.DiffusionClassifier(m, dca.Config(**cfg))
    with pytest.raises(dca._lib.DcamdError):
        dc.classify(torch.zeros(2, 3, 32, 32))
    with pytest.raises(dca._lib.DcamdError):
        dca.wavelet_dec_2(torch.zeros(3, 8, 8))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "diffusion-classifier_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_unsupported_constructor_options_are_refused():
    with pytest.raises(NotImplementedError):
        dca.UNetCondition2D(**dict(dca.small_unet_kwargs(), use_linear_projection=True))
    with pytest.raises(NotImplementedError):
        dca.UNetCondition2D(**dict(dca.small_unet_kwargs(), down_block_types=("AttnDownBlock2D", "DownBlock2D")))


def test_geglu_perm_and_weight_packing():
    from diffusion_classifier_amd import engine as E
    perm = E.geglu_perm(32)
    assert perm.tolist()[:16] == list(range(16)) and perm.tolist()[16:32] == list(range(32, 48))
    assert sorted(perm.tolist()) == list(range(64))
    w = torch.arange(2 * 3 * 9, dtype=torch.float32).reshape(2, 3, 3, 3)
    m = w.permute(0, 2, 3, 1).reshape(2, 27)
    assert m[1, 5 * 3 + 2] == w[1, 2, 1, 2]        # k = (ky*3+kx)*Cin + c


def test_bench_starts_its_own_ranks_as_a_child_process(monkeypatch):
    """`python bench.py --gpus N` typed by hand (no WORLD_SIZE): the parent must start torch.distributed.run as a CHILD process
    (never exec: a process that has touched the GPU must not replace itself) on 127.0.0.1 and relay its exit code."""
    import importlib
    import subprocess
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and os.path.basename(cmd[-5]) == "bench.py"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
