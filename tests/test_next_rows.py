"""CPU: the rows SURVEY §8f marks "next" that are built — checkpoint ingest (accelerate save_state layout),
the evaluate/inference driver contract and the metric counters."""
import os

import numpy as np
import pytest
import torch

import diffusion_classifier_amd as dca
from diffusion_classifier_amd.utils.metrics import Accuracy, F1, Precision, Recall
from helpers import load_case, standin_from


def _dc(name="1stage_eps"):
    g, cfg = load_case(name)
    dc = dca.DiffusionClassifier(standin_from(g, cfg), dca.Config(**cfg))
    dc.encoder.weight.data.copy_(torch.from_numpy(g["encoder.weight"]))
    return dc, g, cfg


def test_metrics_match_closed_forms():
    pred = torch.tensor([1, 0, 1, 1, 0, 0, 1, 0])
    true = torch.tensor([1, 0, 0, 1, 1, 0, 1, 1])
    batch = {"prompt": true}
    out = {}
    for m in (Accuracy("acc"), Precision(), Recall(), F1()):
        m.update((pred[:4], {"prompt": true[:4]}))
        m.update((pred[4:], {"prompt": true[4:]}))
        m.sync_across_processes(None)
        out.update(m.get_output())
    tp, fp, fn = 3, 1, 2
    assert float(out["acc"]) == pytest.approx(5 / 8)
    assert float(out["precision"]) == pytest.approx(tp / (tp + fp))
    assert float(out["recall"]) == pytest.approx(tp / (tp + fn))
    assert float(out["f1"]) == pytest.approx(2 * tp / (2 * tp + fp + fn))
    m = Precision()
    assert m.get_output()["precision"] == 0.0          # empty denominators read 0, like the reference
    m.update((pred, batch)); m.reset()
    assert int(m.tp) == 0


def test_checkpoint_round_trip_in_accelerate_layout(tmp_path):
    dc, g, cfg = _dc()
    with torch.no_grad():
        for p in dc.ema.ema_model.parameters():
            p.add_(0.25)                                 # EMA weights differ from the online model
    dc.save_checkpoint(str(tmp_path), epoch=6, best_metric=0.9)
    assert sorted(os.listdir(tmp_path)) == ["experiment_state.pth", "model.safetensors", "model_1.safetensors", "model_2.safetensors"]
    from safetensors.torch import load_file
    keys = set(load_file(str(tmp_path / "model_1.safetensors")))
    assert any(k.startswith("ema_model.") for k in keys) and "step" in keys
    dc2, _, _ = _dc()
    dc2.encoder.weight.data.zero_()
    epoch, best, key = dc2.load_checkpoint(str(tmp_path))          # the reference's triple (:805)
    assert (epoch, best, key) == (7, 0.9, None)
    for a, b in zip(dc.ema.ema_model.state_dict().values(), dc2.ema.ema_model.state_dict().values()):
        assert torch.equal(a, b)
    assert torch.equal(dc.encoder.weight, dc2.encoder.weight)
    x = torch.from_numpy(g["x"])
    t, eps = torch.from_numpy(g["t"]), torch.from_numpy(g["eps"])
    assert torch.equal(dc.classify(x, t=t, eps=eps), dc2.classify(x, t=t, eps=eps))   # inference uses the EMA copy


def test_checkpoint_bin_spelling_and_missing_ema_are_handled(tmp_path):
    """accelerate with safe_serialization=False writes pytorch_model*.bin; the alternate name is built from the file stem
    only (a parent directory called `models/` must not confuse it), and a checkpoint without EMA weights is refused —
    classification always runs on the EMA copy."""
    from safetensors.torch import load_file
    dc, g, cfg = _dc()
    d = tmp_path / "models" / "ckpt"
    dc.save_checkpoint(str(d))
    for stem in ("model", "model_1", "model_2"):
        sd = load_file(str(d / (stem + ".safetensors")))
        torch.save(sd, str(d / ("pytorch_" + stem + ".bin")))
        os.remove(d / (stem + ".safetensors"))
    dc2, _, _ = _dc()
    dc2.encoder.weight.data.zero_()
    dc2.load_checkpoint(str(d))
    assert torch.equal(dc.encoder.weight, dc2.encoder.weight)
    os.remove(d / "pytorch_model_1.bin")
    with pytest.raises(FileNotFoundError, match="EMA"):
        dc2.load_checkpoint(str(d))


def test_evaluate_and_inference_contract(tmp_path):
    dc, g, cfg = _dc()
    dc.config.experiment_path = str(tmp_path)
    dc.config.fast_classification = False
    x = torch.from_numpy(g["x"])
    loader = [{"images": x, "prompt": torch.from_numpy(g["out"])}, {"images": x, "prompt": torch.from_numpy(g["out"])}]
    torch.manual_seed(int(g["seed"]))
    samples, batches, metrics = dc.evaluate(loader, stop_idx=0, metrics=[Accuracy("accuracy")], classification=True)
    assert len(samples) == 1 and len(batches) == 1                     # stop_idx semantics of the reference (:573-574)
    np.testing.assert_array_equal(samples[0].numpy(), g["out"])
    assert float(metrics[0].get_output()["accuracy"]) == 1.0
    if not torch.cuda.is_available():
        out = dc.inference(None, None, loader, None, metrics=[Accuracy("accuracy")], classification=True)
        assert len(out) == 3 and "accuracy" in out[0][0]


def test_sample_matches_reference_golden():
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "sample_cases.npz")))
    for tag in ("eps", "v_from_t"):
        _, cfg = load_case("1stage_eps")
        cfg.update(pred_param="v" if tag.startswith("v") else "eps", cfg_w=1.5, sampling_steps=4)
        gc, _ = load_case("1stage_eps")
        dc = dca.DiffusionClassifier(standin_from(gc, cfg), dca.Config(**cfg))
        dc.encoder.weight.data.copy_(torch.from_numpy(gc["encoder.weight"]))
        torch.manual_seed(int(g[tag + ".seed"]))
        out = dc.sample(torch.from_numpy(g[tag + ".x"]), torch.from_numpy(g[tag + ".labels"]), from_t=float(g[tag + ".from_t"]))
        np.testing.assert_allclose(out.numpy(), g[tag + ".out"], rtol=0, atol=2e-6)


# ---- f-1 pinned: a checkpoint directory written by the REFERENCE's own save_checkpoint (tools/capture_checkpoint_fixture.py) ----
CKPT = os.path.join(os.path.dirname(__file__), "golden", "ckpt_tiny_unet")


def ckpt_case():
    g = dict(np.load(os.path.join(CKPT, "expected.npz")))
    cfg = {k[4:]: (v.tolist() if v.ndim else v.item()) for k, v in g.items() if k.startswith("cfg.")}
    arch = {k[5:]: (tuple(v.tolist()) if v.ndim else v.item()) for k, v in g.items() if k.startswith("arch.")}
    return g, cfg, arch


def test_reference_written_checkpoint_is_ingested_and_scores_like_the_reference():
    import oracle
    g, cfg, arch = ckpt_case()
    assert sorted(os.listdir(CKPT)) == ["expected.npz", "experiment_state.pth", "model.safetensors", "model_1.safetensors",
                                        "model_2.safetensors", "optimizer.bin", "random_states_0.pkl", "scheduler.bin"]
    torch.manual_seed(0)
    dc = dca.DiffusionClassifier(oracle.OracleUNetCondition2D(**arch), dca.Config(**cfg))     # foreign (plain nn.Module) backbone
    assert dc.load_checkpoint(CKPT) == (5, 0.75, None)                  # reference :805 (epoch + 1, best metric, comet key)
    assert int(dc.ema.step) == 123 and bool(dc.ema.initted)
    online, ema = dc.model.state_dict(), dc.ema.ema_model.state_dict()
    assert any(not torch.equal(online[k], ema[k]) for k in online)      # EMA weights are their own tensors in model_1
    lab, err = dc.classify(torch.from_numpy(g["x"]), t=torch.from_numpy(g["t"]), eps=torch.from_numpy(g["eps"]), return_errors=True)
    np.testing.assert_array_equal(lab.numpy(), g["labels"])
    np.testing.assert_allclose(err.numpy(), g["errors"], rtol=1e-6)     # same modules, same ops as the reference's run
