"""GPU: the "next" rows on the HIP backbone — checkpoint ingest (f-1) from a directory the reference itself wrote, and the
evaluate / inference driver with metrics (f-2)."""
import os
import shutil

import numpy as np
import pytest
import torch

import diffusion_classifier_amd as dca
from diffusion_classifier_amd.utils.metrics import Accuracy, F1
from test_next_rows import CKPT, ckpt_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _hip_dc(cfg, arch, **over):
    torch.manual_seed(0)
    return dca.DiffusionClassifier(dca.UNetCondition2D(**arch), dca.Config(**dict(cfg, compute_dtype="f32", **over)))


def test_hip_classify_on_reference_written_checkpoint_matches_the_reference_scores():
    """load_checkpoint(<dir written by the reference's save_checkpoint>) -> HIP classify == the errors / labels the reference's
    own loop produced with those (EMA) weights on the CPU."""
    g, cfg, arch = ckpt_case()
    dc = _hip_dc(cfg, arch)
    assert dc.load_checkpoint(CKPT) == (5, 0.75, None)
    dc = dc.to(DEV)
    lab, err = dc.classify(torch.from_numpy(g["x"]).to(DEV), t=torch.from_numpy(g["t"]), eps=torch.from_numpy(g["eps"]).to(DEV),
                           return_errors=True)
    rel = np.abs(err.numpy() - g["errors"]) / g["errors"]
    assert rel.max() < 1e-4, rel.max()
    np.testing.assert_array_equal(lab.cpu().numpy(), g["labels"])
    # scoring with the ONLINE weights instead would be visibly different: the EMA copy is what runs (reference :700)
    dc.ema.ema_model.load_state_dict(dc.model.state_dict())
    err_online = dc.classify(torch.from_numpy(g["x"]).to(DEV), t=torch.from_numpy(g["t"]), eps=torch.from_numpy(g["eps"]).to(DEV),
                             return_errors=True)[1]
    assert (np.abs(err_online.numpy() - g["errors"]) / g["errors"]).max() > 1e-3


def test_inference_driver_on_hip_backbone_with_metrics(tmp_path):
    """inference(): checkpoint load from experiment_path/checkpoint_folder, evaluate over a 2-batch loader (stop_idx from
    config.evaluation_batches), metric update / sync / output — reference :581-655 — on the HIP path."""
    g, cfg, arch = ckpt_case()
    shutil.copytree(CKPT, tmp_path / "best_checkpoint")
    dc = _hip_dc(cfg, arch, experiment_path=str(tmp_path), fast_classification=False, evaluation_batches=1)
    x = torch.from_numpy(g["x"])
    truth = torch.from_numpy(g["labels"])
    loader = [{"images": x, "prompt": truth}, {"images": x.flip(0), "prompt": truth.flip(0)}, {"images": x, "prompt": truth}]
    torch.manual_seed(99)                                       # the reference's draw order (rand(BS), randn_like(x) per trial) on the CPU generator
    real = torch.randn_like
    torch.randn_like = lambda t_, **k: torch.randn(t_.shape).to(t_.device)     # x lives on the GPU here; draw eps from the CPU stream like the fixture did
    try:
        out, samples, batches = dc.inference(None, None, loader, None, metrics=[Accuracy("accuracy"), F1()], classification=True,
                                             checkpoint_folder="best_checkpoint")
    finally:
        torch.randn_like = real
    assert len(samples) == 2 and len(batches) == 2              # stop_idx = evaluation_batches = 1 -> two batches (:573-574)
    assert samples[0].is_cuda and batches[0]["images"].is_cuda
    np.testing.assert_array_equal(samples[0].cpu().numpy(), g["labels"])       # first batch: same seed, same draws as the fixture run
    acc = float(out[0]["accuracy"])
    agree = float(torch.cat([s.cpu() for s in samples]).eq(torch.cat([truth, truth.flip(0)])).float().mean())
    assert acc == pytest.approx(agree) and acc >= 0.5
    assert os.path.isdir(tmp_path / "inference_images")


def test_inference_prefetches_the_next_batch_and_transforms_it_on_the_side_stream():
    """SURVEY §8f-2 / VERDICT r3 item 7: `inference()` moves batch i+1 to the device (pinned staging, non_blocking) and, with
    `dwt_on_device=True`, applies haar_dwt2(.)/2 to it on a side stream while batch i is being scored — what the reference does per
    item on the host (dataset/chexpert.py:146-147) before evaluate (:555-563) sees the batch.  Labels must equal the unpipelined loop
    (host -> device copy, DWT, classify, one after the other on one stream), and the next batch's copy + transform must have ENDED
    while the current batch was still being scored (event timestamps)."""
    from diffusion_classifier_amd.utils.wavelet import wavelet_dec_2
    arch = dict(dca.small_unet_kwargs(), in_channels=12, out_channels=12, sample_size=32)
    cfg = dict(pred_param="eps", schedule="cosine", noise_d=32, image_size=32, cfg_w=0.0, ema_beta=0.999, ema_warmup=0, ema_update_freq=1,
               encoder_type="nn", classes=4, n_stages=1, evaluation_per_stage=[24], n_keep_per_stage=[1], n_fast_classes=2,
               fast_classification=False, compute_dtype="bf16")
    torch.manual_seed(3)
    model = dca.UNetCondition2D(**arch)
    g = torch.Generator().manual_seed(4)
    loader = [{"images": (torch.rand(16, 3, 64, 64, generator=g) * 2 - 1), "prompt": torch.randint(0, 4, (16,), generator=g)} for _ in range(4)]

    def run(pipelined):
        torch.manual_seed(5)                                   # (the constructor draws the class-embedding table)
        dc = dca.DiffusionClassifier(model, dca.Config(**dict(cfg, dwt_on_device=pipelined))).to(DEV)
        torch.manual_seed(11)
        torch.cuda.manual_seed(11)
        if pipelined:
            samples, batches = dc.inference(None, None, loader, None, classification=True)
            assert batches[0]["images"].shape == (16, 12, 32, 32) and batches[0]["images"].is_cuda
            return [s.cpu() for s in samples], dc._prefetch_log
        out = []
        for b in loader:
            x = wavelet_dec_2(b["images"].to(DEV), scale=0.5)
            out.append(dc.classify(x, b["prompt"].to(DEV)).cpu())
        return out, None

    want, _ = run(False)
    got, log = run(True)
    torch.cuda.synchronize()
    assert len(got) == len(want) == 4
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    for i in range(1, len(log) - 0):
        e0, e1, eh = log[i]
        window = log[i - 1][2].elapsed_time(eh)                 # batch i-1 handed over -> batch i handed over = batch i-1's scoring
        assert e0.elapsed_time(eh) > 0 and window > 0
        assert e1.elapsed_time(eh) > 0.5 * window, (i, e1.elapsed_time(eh), window)     # batch i was ready in the first half of that window
