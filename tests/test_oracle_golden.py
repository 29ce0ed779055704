"""CPU: pin the oracle against vectors captured from the reference itself
(tools/capture_goldens.py, tools/capture_dwt_goldens.py)."""
import os

import numpy as np
import pytest
import torch

import oracle
from helpers import CASES, GOLDEN, load_case, standin_from


def test_schedules_match_reference():
    g = np.load(os.path.join(GOLDEN, "schedules.npz"))
    t = torch.from_numpy(g["t"])
    for nd, im in [(32, 32), (64, 32), (128, 256), (256, 256)]:
        a = oracle.logsnr_schedule_cosine(t, nd, im).numpy()
        b = oracle.logsnr_schedule_cosine_shifted(t, nd, im).numpy()
        np.testing.assert_array_equal(a, g[f"cosine_{nd}_{im}"])
        np.testing.assert_array_equal(b, g[f"shifted_{nd}_{im}"])


def test_schedule_probe_values():
    # SURVEY §8 a-2 probe values (noise_d == image_d)
    lam = oracle.logsnr_schedule_cosine(torch.tensor([0.0, 0.25, 0.5, 0.75, 1.0]), 32, 32)
    np.testing.assert_allclose(lam.numpy(), [15.0, 1.7611834, 1.19e-07, -1.7611831, -14.999989],
                               rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("name", CASES)
def test_classify_matches_reference(name):
    g, cfg = load_case(name)
    bb = standin_from(g, cfg)
    dc = oracle.OracleDiffusionClassifier(bb, oracle.AttrBag(**cfg))
    if dc.encoder is not None:
        dc.encoder.weight.data.copy_(torch.from_numpy(g["encoder.weight"]))
    x = torch.from_numpy(g["x"])
    fast = bool(g["fast"])
    out, errors = dc.classify(
        x, torch.from_numpy(g["labels"]) if fast else None, fast=fast,
        t=torch.from_numpy(g["t"]), eps=torch.from_numpy(g["eps"]),
        fast_select=torch.from_numpy(g["fast_select"]) if fast else None, return_errors=True)
    np.testing.assert_array_equal(out.numpy(), g["out"])
    np.testing.assert_array_equal(errors.numpy(), g["errors"])   # same ops, same order: bit-exact


@pytest.mark.parametrize("name", ["1stage_eps", "fast"])
def test_classify_rng_replay(name):
    """The loop's own RNG consumption order (randint?, then rand(BS), randn_like(x) per trial)."""
    g, cfg = load_case(name)
    bb = standin_from(g, cfg)
    dc = oracle.OracleDiffusionClassifier(bb, oracle.AttrBag(**cfg))
    if dc.encoder is not None:
        dc.encoder.weight.data.copy_(torch.from_numpy(g["encoder.weight"]))
    fast = bool(g["fast"])
    torch.manual_seed(int(g["seed"]))
    out = dc.classify(torch.from_numpy(g["x"]), torch.from_numpy(g["labels"]) if fast else None, fast=fast)
    np.testing.assert_array_equal(out.numpy(), g["out"])


def test_dwt_matches_pywt():
    g = np.load(os.path.join(GOLDEN, "dwt_pywt.npz"))
    for k in ["rand_3x64x64", "rand_10x32x48", "ramp_1x4x4"]:
        dec = oracle.haar_dwt2(g[k + ".x"])
        np.testing.assert_allclose(dec, g[k + ".dec"], rtol=0, atol=2e-6 * max(1.0, np.abs(g[k + ".dec"]).max()))
        rec = oracle.haar_idwt2(g[k + ".dec"])
        np.testing.assert_allclose(rec, g[k + ".enc_of_dec"], rtol=0, atol=2e-6 * max(1.0, np.abs(g[k + ".x"]).max()))
        np.testing.assert_allclose(oracle.haar_idwt2(dec), g[k + ".x"], rtol=0, atol=2e-6 * max(1.0, np.abs(g[k + ".x"]).max()))


def test_dwt_ramp_known_answer():
    # cA=(a+b+c+d)/2 ... on arange(16)
    x = np.arange(16, dtype=np.float32).reshape(1, 4, 4)
    d = oracle.haar_dwt2(x)
    np.testing.assert_array_equal(d[0], [[5, 9], [21, 25]])
    np.testing.assert_array_equal(d[1], [[-4, -4], [-4, -4]])
    np.testing.assert_array_equal(d[2], [[-1, -1], [-1, -1]])
    np.testing.assert_array_equal(d[3], [[0, 0], [0, 0]])
