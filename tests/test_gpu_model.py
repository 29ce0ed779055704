"""GPU: whole-backbone and whole-classify parity of the HIP path against the CPU oracle.

Bars (BASELINE.json north_star): fp32 path — per-class eps-MSE within 1e-4 relative of the CPU
PyTorch path and bit-exact arg-min labels.  bf16/f16 paths — compared with the oracle run with
the SAME storage rounding (`lowp=True`), tolerance stated per test, plus label agreement.
"""
import numpy as np
import pytest
import torch

import diffusion_classifier_amd as dca
import oracle
from helpers import hip_preds, pred_rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_pair(kw, seed=0, lowp=False):
    torch.manual_seed(seed)
    m = dca.UNetCondition2D(**kw)
    # default inits leave GroupNorm/LayerNorm affine at (1,0) and biases tiny: randomise them so the
    # test can see a swapped gamma/beta or a dropped bias
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    o = oracle.OracleUNetCondition2D(**kw, lowp=lowp)
    o.load_state_dict(m.state_dict())
    return m, o


def relerr(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


@pytest.mark.parametrize("share", [True, False])
def test_small_unet_forward_f32(share):
    kw = dca.small_unet_kwargs()
    m, o = make_pair(kw)
    torch.manual_seed(1)
    N = 5
    x, lam, emb = torch.randn(N, 3, 32, 32), torch.tensor([14.9, 2.0, 0.0, -3.0, -14.0]), torch.randn(N, 1, 64)
    ref = o(x, lam, encoder_hidden_states=emb)
    m = m.to(DEV)
    m.share_trunk = share
    got = m(x.to(DEV), lam.to(DEV), encoder_hidden_states=emb.to(DEV)).cpu()
    assert got.shape == ref.shape
    assert relerr(got, ref) < 2e-5, relerr(got, ref)
    assert (got - ref).abs().max().item() < 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("size", [24, 20])
def test_small_unet_forward_f32_non_power_of_two_images(size):
    """Image extents the halo kernels do not take (they want powers of two): every conv falls to the tap-gather GEMM, attention runs on
    144 / 100 tokens — same parity bar."""
    kw = dict(dca.small_unet_kwargs(), sample_size=size)
    m, o = make_pair(kw, seed=6)
    torch.manual_seed(7)
    x, lam, emb = torch.randn(3, 3, size, size), torch.tensor([3.0, 0.0, -4.0]), torch.randn(3, 1, 64)
    ref = o(x, lam, encoder_hidden_states=emb)
    got = m.to(DEV)(x.to(DEV), lam.to(DEV), encoder_hidden_states=emb.to(DEV)).cpu()
    assert got.shape == ref.shape
    assert relerr(got, ref) < 2e-5, relerr(got, ref)


def test_small_unet_forward_f32_with_large_activation_offsets():
    """A checkpoint-like failure mode random weights never show: conv biases far larger than the activations' spread, so that every
    GroupNorm sees groups with |mean| >> std (tens of standard deviations here; the offsets differ per group).  The fp32 path must
    still sit on the oracle — the sum / sum-of-squares statistics of round 1 lost the variance to cancellation at this point."""
    kw = dca.small_unet_kwargs()
    m, o = make_pair(kw, seed=3)
    G = kw.get("norm_num_groups", 32)
    with torch.no_grad():
        torch.manual_seed(4)
        for n, p in m.named_parameters():
            if p.dim() == 1 and n.endswith("bias") and (".conv1." in n or ".conv2." in n or "conv_in" in n or ".conv." in n or "conv_shortcut" in n):
                cpg = max(p.numel() // G, 1)
                p.add_((torch.randn(p.numel() // cpg, 1) * 8 + 25).expand(-1, cpg).reshape(-1)[: p.numel()])
    o.load_state_dict(m.state_dict())
    torch.manual_seed(5)
    N = 5
    x, lam, emb = torch.randn(N, 3, 32, 32), torch.tensor([9.0, 2.0, 0.0, -3.0, -9.0]), torch.randn(N, 1, 64)
    ref = o(x, lam, encoder_hidden_states=emb)
    m = m.to(DEV)
    got = m(x.to(DEV), lam.to(DEV), encoder_hidden_states=emb.to(DEV)).cpu()
    assert torch.isfinite(got).all()
    assert relerr(got, ref) < 5e-5, relerr(got, ref)


@pytest.mark.parametrize("dtname,tol", [("bf16", 1.5e-2), ("f16", 2.5e-3)])
def test_small_unet_forward_lowp(dtname, tol):
    kw = dca.small_unet_kwargs()
    m, o = make_pair(kw, lowp=True)
    if dtname == "f16":
        o._q = lambda t: t.to(torch.float16).float()
    torch.manual_seed(2)
    N = 4
    x, lam, emb = torch.randn(N, 3, 32, 32), torch.tensor([5.0, 1.0, -1.0, -8.0]), torch.randn(N, 1, 64)
    ref = o(x, lam, encoder_hidden_states=emb)
    m = m.to(DEV).set_compute_dtype(dtname)
    got = m(x.to(DEV), lam.to(DEV), encoder_hidden_states=emb.to(DEV)).cpu()
    # same rounding points, fp32 accumulation in a different order -> a few low-precision ulps drift
    assert relerr(got, ref) < tol, relerr(got, ref)
    o32 = oracle.OracleUNetCondition2D(**kw)
    o32.load_state_dict(o.state_dict())
    ref32 = o32(x, lam, encoder_hidden_states=emb)
    assert relerr(got, ref32) < 4 * tol     # and it is a faithful low-precision version of the fp32 net


def test_cifar_unet_forward_f32():
    """BASELINE config-2 architecture (experiments/cifar10/inference.py:94-116), one forward."""
    kw = dca.cifar10_unet_kwargs()
    m, o = make_pair(kw, seed=3)
    torch.manual_seed(4)
    N = 11        # more samples than a wave of the 4x4 level holds (8): per-sample vectors must follow their own sample
    x, lam, emb = torch.randn(N, 3, 32, 32), torch.linspace(9.0, -9.0, N), torch.randn(N, 1, 128)
    ref = o(x, lam, encoder_hidden_states=emb)
    m = m.to(DEV)
    got = m(x.to(DEV), lam.to(DEV), encoder_hidden_states=emb.to(DEV)).cpu()
    assert relerr(got, ref) < 3e-5, relerr(got, ref)
    per = ((got - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)).max().item()
    assert per < 5e-5, per                                   # every sample, not only the batch as a whole


def test_cifar_unet_forward_bf16():
    """BASELINE config-2 architecture in the bench's compute type, against the oracle with the same storage rounding: covers
    the bf16 kernels as the bench runs them (halo / four-phase upsample convs, producer-side GroupNorm statistics, one-wave
    GroupNorm, folded shortcuts and LayerNorm).  Tolerance: a few bf16 ulps of drift over ~60 layers."""
    kw = dca.cifar10_unet_kwargs()
    m, o = make_pair(kw, seed=3, lowp=True)
    torch.manual_seed(4)
    N = 3
    x, lam, emb = torch.randn(N, 3, 32, 32), torch.tensor([3.0, -2.0, 7.5]), torch.randn(N, 1, 128)
    ref = o(x, lam, encoder_hidden_states=emb)
    m = m.to(DEV).set_compute_dtype("bf16")
    got = m(x.to(DEV), lam.to(DEV), encoder_hidden_states=emb.to(DEV)).cpu()
    assert torch.isfinite(got).all()
    assert relerr(got, ref) < 1.5e-2, relerr(got, ref)          # measured 6.7e-3 (6.5e-3 with qstats / up4 switched off)
    plan = next(iter(m._plans.values()))
    fams = {mt.get("family", "") for mt in plan.pb.meta}
    # the 8x8 -> 16x16 and 16x16 -> 32x32 upsample convs as four-phase halo convs; the 4x4 -> 8x8 one on mosaic halo patches too
    # (on the tap-gather kernel when the mosaic is switched off)
    import os
    assert any(f.startswith("conv3_up4<bf16,4w") for f in fams) and any(f.startswith("conv3_up4<bf16,8w") for f in fams), fams
    assert any(f.startswith("igemm_pipe_up4<bf16") for f in fams) == (os.environ.get("DCAMD_NO_MOSAIC") is not None), fams
    # GroupNorms fed by their producer's quad statistics: as a pass with the records, or stored normalised by the producer itself
    assert sum(1 for (_, _, f) in plan.pb.ops if "qparts" in f or f.get("pn_out") is not None) >= 20
    assert any(f.endswith(",pn>") for f in fams), fams


def test_class_shared_skip_halves_match_the_unsplit_plan(monkeypatch):
    """conv(cat(h, skip)) = conv_a(h) + conv_b(skip): the up-path ResNets whose skip comes from the class-shared trunk run
    the skip half (GroupNorm, 3x3 conv, 1x1 shortcut) once per pair.  Same network, different summation order only."""
    kw = dca.cifar10_unet_kwargs()
    torch.manual_seed(8)
    x, lam, emb = torch.randn(3, 3, 32, 32), torch.tensor([4.0, 0.5, -6.0]), torch.randn(3, 1, 128)
    outs, names = [], []
    for split in (True, False):
        if split:
            monkeypatch.delenv("DCAMD_NO_SKIP_SPLIT", raising=False)
        else:
            monkeypatch.setenv("DCAMD_NO_SKIP_SPLIT", "1")
        m, _ = make_pair(kw, seed=7)
        m = m.to(DEV)
        outs.append(m(x.to(DEV), lam.to(DEV), encoder_hidden_states=emb.to(DEV)).cpu())
        plan = next(iter(m._plans.values()))
        names.append([mt["name"] for mt in plan.pb.meta])
    assert sum(n.endswith(".conv1s") for n in names[0]) == 5 and not any(n.endswith(".conv1s") for n in names[1])
    assert relerr(outs[0], outs[1]) < 1e-5, relerr(outs[0], outs[1])


@pytest.mark.parametrize("dtname,tol", [("f32", 1e-5), ("bf16", 1.2e-2)])
def test_transformer_front_launch_proj_out_fold_and_shortcut_split_match_the_chain_plan(monkeypatch, dtname, tol):
    """Second half of round 4: (a) proj_in -> LayerNorm -> q/k/v -> attention -> to_out of a 64-token block as ONE launch (dc_tblock_front,
    16-bit only), (b) proj_out folded into the feed-forward's second linear ([Wpo W2 | Wpo] over [f | h]), (c) the 1x1 shortcut of a
    class-shared skip split although a GroupNorm group straddles the seam.  Same network; (b) and (c) are exact algebra (f32: summation order
    only), (a) rounds where the chain stores."""
    kw = dca.cifar10_unet_kwargs()
    torch.manual_seed(28)
    N = 2
    x, lam, emb = torch.randn(N, 3, 32, 32), torch.tensor([3.0, -5.0]), torch.randn(N, 1, 128)
    outs, fams, names = [], [], []
    for on in (True, False):
        for v in ("DCAMD_NO_TBLOCK", "DCAMD_NO_PO_FOLD", "DCAMD_NO_SKIP_SPLIT"):
            if on:
                monkeypatch.delenv(v, raising=False)
            else:
                monkeypatch.setenv(v, "1")
        m, _ = make_pair(kw, seed=7)
        m = m.to(DEV).set_compute_dtype(dtname)
        outs.append(m(x.to(DEV), lam.to(DEV), encoder_hidden_states=emb.to(DEV)).float().cpu())
        plan = next(iter(m._plans.values()))
        fams.append([mt["family"] for mt in plan.pb.meta])
        names.append([mt["name"] for mt in plan.pb.meta])
    n_front = sum(f == "tblock_front" for f in fams[0])
    # the four per-class 8x8-level blocks take the launch; the class-shared first transformer (its proj_in ... attention run per pair, to_out per class) keeps the chain
    assert n_front == (4 if dtname != "f32" else 0) and "tblock_front" not in fams[1], (n_front, dtname)
    assert sum(n.endswith(".ff_proj_out") for n in names[0]) == 11 and not any(n.endswith(".ff_proj_out") for n in names[1])
    assert sum(n.endswith(".proj_out") for n in names[1]) == 11
    print(f"{dtname}: plans with / without the three changes differ by {relerr(outs[0], outs[1]):.2e} (bound {tol:.1e}); {len(names[0])} vs {len(names[1])} launches")
    assert relerr(outs[0], outs[1]) < tol, relerr(outs[0], outs[1])


def test_producer_side_groupnorm_statistics_match_the_swept_plan(monkeypatch):
    """3x3 convs hand (sum, sumsq) quad statistics of their output to the GroupNorm that follows (dc_igemm qstats), which
    then streams the tensor once.  Same statistics up to fp32 summation order."""
    kw = dca.cifar10_unet_kwargs()
    torch.manual_seed(18)
    x, lam, emb = torch.randn(3, 3, 32, 32), torch.tensor([4.0, 0.5, -6.0]), torch.randn(3, 1, 128)
    outs, used = [], []
    for on in (True, False):
        if on:
            monkeypatch.delenv("DCAMD_NO_QSTATS", raising=False)
        else:
            monkeypatch.setenv("DCAMD_NO_QSTATS", "1")
        m, _ = make_pair(kw, seed=7)
        m = m.to(DEV)
        outs.append(m(x.to(DEV), lam.to(DEV), encoder_hidden_states=emb.to(DEV)).cpu())
        plan = next(iter(m._plans.values()))
        used.append(sum(1 for (_, _, f) in plan.pb.ops if "qparts" in f or f.get("pn_out") is not None))
    assert used[0] >= 20 and used[1] == 0, used          # (with the records: as a one-sweep pass, or normalised by the producer itself)
    assert relerr(outs[0], outs[1]) < 1e-5, relerr(outs[0], outs[1])


def test_producer_normalised_plan_matches_the_plan_with_groupnorm_passes(monkeypatch):
    """Producer-side GroupNorm (csrc/epi_pn.h, engine.PlanBuilder.pn_claim) against the round-3 plan (`DCAMD_NO_PN`: GroupNorm passes and
    the wave-specialised conv): the same network, the same statistics records, only the place where the tensor is normalised differs.  f32:
    rounding noise; the plan must really contain the three producer forms (4-wave exchange, upsample, 8-wave wave-local)."""
    kw = dca.cifar10_unet_kwargs()
    torch.manual_seed(19)
    x, lam, emb = torch.randn(5, 3, 32, 32), torch.tensor([4.0, 0.5, -6.0, 1.0, 9.0]), torch.randn(5, 1, 128)
    outs, fams = [], []
    for on in (True, False):
        if on:
            monkeypatch.delenv("DCAMD_NO_PN", raising=False)
        else:
            monkeypatch.setenv("DCAMD_NO_PN", "1")
        m, _ = make_pair(kw, seed=7)
        m = m.to(DEV)
        outs.append(m(x.to(DEV), lam.to(DEV), encoder_hidden_states=emb.to(DEV)).cpu())
        plan = next(iter(m._plans.values()))
        fams.append({mt.get("family", "") for mt in plan.pb.meta})
    assert {"conv3_halo<f32,4w,pn>", "conv3_halo<f32,8w,pn>", "conv3_up4<f32,4w,pn>"} <= fams[0], fams[0]
    assert not any(f.endswith(",pn>") for f in fams[1]) and any(f.startswith("conv3_ws<") for f in fams[1]), fams[1]
    assert relerr(outs[0], outs[1]) < 1e-5, relerr(outs[0], outs[1])


def test_one_token_cross_attention_shortcut_is_exact():
    """attn2 over a single class token == to_out(to_v(ctx)) for every query (what the engine uses)."""
    kw = dca.small_unet_kwargs()
    _, o = make_pair(kw, seed=5)
    t = o.mid_block.attentions[0]
    torch.manual_seed(6)
    h, ctx = torch.randn(3, 16, 128), torch.randn(3, 1, 64)
    full = o.cross_attn_exact(t, h, ctx)
    a2 = t.transformer_blocks[0].attn2
    short = a2.to_out[0](a2.to_v(ctx[:, 0]))[:, None, :].expand_as(full)
    assert (full - short).abs().max().item() < 1e-6


def _classifiers(kw, cfg, seed=0, lowp=False):
    m, o = make_pair(kw, seed=seed, lowp=lowp)
    dc = dca.DiffusionClassifier(m, dca.Config(**cfg))
    oc = oracle.OracleDiffusionClassifier(o, oracle.AttrBag(**cfg))
    if dc.encoder is not None:
        oc.encoder.load_state_dict(dc.encoder.state_dict())
    return dc.to(DEV), oc


BASE = dict(pred_param="eps", schedule="cosine", noise_d=32, image_size=32, cfg_w=0.0, ema_beta=0.999, ema_warmup=0,
            ema_update_freq=1, encoder_type="nn", classes=4, n_stages=1, evaluation_per_stage=[6], n_keep_per_stage=[1],
            n_fast_classes=2, compute_dtype="f32")


@pytest.mark.parametrize("variant", ["eps", "v_shifted", "two_stage", "fast", "tiny_launch"])
def test_classify_small_unet_f32_matches_oracle(variant):
    cfg = dict(BASE)
    fast = False
    if variant == "v_shifted":
        cfg.update(pred_param="v", schedule="shifted_cosine", noise_d=64)
    if variant == "two_stage":
        cfg.update(classes=6, n_stages=2, evaluation_per_stage=[3, 7], n_keep_per_stage=[2, 1])
    if variant == "fast":
        cfg.update(classes=6, n_fast_classes=3)
        fast = True
    if variant == "tiny_launch":
        cfg.update(units_per_launch=7)        # ragged micro-batches with padding rows
    dc, oc = _classifiers(dca.small_unet_kwargs(), cfg, seed=7)
    torch.manual_seed(8)
    BS, T = 3, cfg["evaluation_per_stage"][-1]
    x = torch.rand(BS, 3, 32, 32) * 2 - 1
    t, eps = torch.rand(T, BS), torch.randn(T, BS, 3, 32, 32)
    lab = torch.randint(0, cfg["classes"], (BS,))
    sel = torch.randint(0, cfg["classes"] - 1, (BS, cfg["n_fast_classes"] - 1)) if fast else None
    full = variant in ("eps", "v_shifted")          # one stage, every class, one micro-batch: the predictions can be lined up
    ref = oc.classify(x, lab if fast else None, fast=fast, t=t, eps=eps, fast_select=sel, return_errors=True, return_preds=full)
    ref_l, ref_e = ref[0], ref[1]
    got_l, got_e = dc.classify(x.to(DEV), lab.to(DEV) if fast else None, fast=fast, t=t, eps=eps.to(DEV), fast_select=sel,
                               return_errors=True)
    if full:
        # the per-cell eps-MSE is dominated by ||eps||^2 and barely moves when the prediction is slightly wrong: compare the
        # backbone outputs of the scored plan themselves, every (trial, image, class) sample on its own
        pr = pred_rel_l2(hip_preds(dc, T, BS), ref[2])
        assert pr < 5e-5, pr
    assert torch.equal(torch.isinf(got_e), torch.isinf(ref_e))          # same (class, trial) cells evaluated
    fin = torch.isfinite(ref_e)
    rel = ((got_e[fin] - ref_e[fin]).abs() / ref_e[fin]).max().item()
    assert rel < 1e-4, rel                                              # north-star bar: per-class eps-MSE within 1e-4
    assert got_l.device.type == "cuda" and got_l.dtype == torch.int64
    assert got_l.cpu().tolist() == ref_l.tolist()                       # bit-exact arg-min labels


def test_classify_bf16_against_lowp_oracle():
    cfg = dict(BASE, compute_dtype="bf16", evaluation_per_stage=[4])
    dc, oc = _classifiers(dca.small_unet_kwargs(), cfg, seed=9, lowp=True)
    torch.manual_seed(10)
    BS, T = 3, 4
    x = torch.rand(BS, 3, 32, 32) * 2 - 1
    t, eps = torch.rand(T, BS), torch.randn(T, BS, 3, 32, 32)
    ref_l, ref_e, ref_p = oc.classify(x, t=t, eps=eps, return_errors=True, return_preds=True)
    got_l, got_e = dc.classify(x.to(DEV), t=t, eps=eps.to(DEV), return_errors=True)
    rel = ((got_e - ref_e).abs() / ref_e).max().item()
    assert rel < 2e-2, rel               # bf16 storage (2^-8) at identical rounding points; fp32 accumulate
    pr = pred_rel_l2(hip_preds(dc, T, BS), ref_p)
    assert pr < 2e-2, pr                 # the scored plan's predictions, per sample
    means_g, means_r = got_e.mean(2), ref_e.mean(2)
    gap = (means_r.sort(1).values[:, 1] - means_r.sort(1).values[:, 0]) / means_r.min(1).values
    decided = gap > 5e-2                 # only well-separated images must agree at bf16
    assert (got_l.cpu()[decided] == ref_l[decided]).all()


def test_classify_philox_is_world_size_and_launch_size_independent():
    cfg = dict(BASE, evaluation_per_stage=[5])
    dc, _ = _classifiers(dca.small_unet_kwargs(), cfg, seed=11)
    torch.manual_seed(12)
    x = (torch.rand(2, 3, 32, 32) * 2 - 1).to(DEV)
    t = torch.rand(5, 2)
    l1, e1 = dc.classify(x, t=t, rng="philox", seed=99, return_errors=True)
    dc.config.units_per_launch = 8
    l2, e2 = dc.classify(x, t=t, rng="philox", seed=99, return_errors=True)
    assert torch.equal(e1, e2) and torch.equal(l1, l2)    # same (seed, image, trial) noise and same kernels -> bit-identical
    l3, e3 = dc.classify(x, t=t, rng="philox", seed=100, return_errors=True)
    assert not torch.equal(e1, e3)


def test_small_dit_forward_and_classify_f32():
    kw = dict(num_attention_heads=2, attention_head_dim=32, in_channels=4, num_layers=2, sample_size=16, patch_size=4,
              num_embeds_ada_norm=10)
    torch.manual_seed(13)
    m = dca.DiT(**kw)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    o = oracle.OracleDiT(**kw)
    o.load_state_dict(m.state_dict())
    N = 3
    x, lam, lab = torch.randn(N, 4, 16, 16), torch.tensor([4.0, 0.0, -6.0]), torch.tensor([1, 9, 3])
    ref = o(x, lam, lab)
    md = m.to(DEV)
    got = md(x.to(DEV), lam.to(DEV), lab.to(DEV)).cpu()
    assert relerr(got, ref) < 2e-5, relerr(got, ref)
    cfg = dict(BASE, encoder_type="DiT", classes=3, evaluation_per_stage=[4], image_size=16, noise_d=16)
    dc = dca.DiffusionClassifier(m.cpu(), dca.Config(**cfg)).to(DEV)
    oc = oracle.OracleDiffusionClassifier(o, oracle.AttrBag(**cfg))
    BS, T = 2, 4
    xx = torch.rand(BS, 4, 16, 16) * 2 - 1
    t, eps = torch.rand(T, BS), torch.randn(T, BS, 4, 16, 16)
    ref_l, ref_e = oc.classify(xx, t=t, eps=eps, return_errors=True)
    got_l, got_e = dc.classify(xx.to(DEV), t=t, eps=eps.to(DEV), return_errors=True)
    assert ((got_e - ref_e).abs() / ref_e).max().item() < 1e-4
    assert got_l.cpu().tolist() == ref_l.tolist()


def test_dit_b4_shaped_blocks_f16_against_lowp_oracle():
    """BASELINE config-5 geometry (models/chexpert-256-dit-b4.py: 12 heads x 64, patch 4, 12x128x128 DWT input ->
    1024 tokens) with 2 of the 12 layers: exercises the 1024-token flash attention path in fp16."""
    kw = dict(dca.chexpert_dit_b4_kwargs(True), num_layers=2, num_embeds_ada_norm=10)
    torch.manual_seed(21)
    m = dca.DiT(**kw)
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if p_.dim() == 1:
                p_.add_(torch.randn_like(p_) * 0.1)
    o = oracle.OracleDiT(**kw, lowp=True, lowp_dtype=torch.float16)
    o.load_state_dict(m.state_dict())
    N = 2
    x, lam, lab = torch.randn(N, 12, 128, 128) * 0.5, torch.tensor([2.0, -3.0]), torch.tensor([1, 7])
    ref = o(x, lam, lab)
    got = m.to(DEV).set_compute_dtype("f16")(x.to(DEV), lam.to(DEV), lab.to(DEV)).cpu()
    assert relerr(got, ref) < 5e-3, relerr(got, ref)


def test_chexpert_dwt_unet_forward_bf16():
    """BASELINE config-3 architecture (models/chexpert-256-unet-dwt-healthysick.py), 12x128x128, one forward in bf16
    against the oracle with the same storage rounding; also checks trunk sharing on a 5-level UNet."""
    kw = dca.chexpert_dwt_unet_kwargs()
    m, o = make_pair(kw, seed=22, lowp=True)
    torch.manual_seed(23)
    N = 2
    x, lam, emb = torch.randn(N, 12, 128, 128) * 0.5, torch.tensor([1.5, -2.5]), torch.randn(N, 1, 512)
    ref = o(x, lam, encoder_hidden_states=emb)
    got = m.to(DEV).set_compute_dtype("bf16")(x.to(DEV), lam.to(DEV), encoder_hidden_states=emb.to(DEV)).cpu()
    assert relerr(got, ref) < 2e-2, relerr(got, ref)


def test_sample_on_hip_backbone_matches_cpu_backbone():
    """Generation: the same sampler code driving the HIP UNet and the CPU oracle UNet."""
    kw = dca.small_unet_kwargs()
    cfg = dict(BASE, cfg_w=1.5, sampling_steps=3, classes=4)
    m, o = make_pair(kw, seed=31)
    dc_cpu = dca.DiffusionClassifier(o, dca.Config(**cfg))          # foreign (plain nn.Module) backbone, eager torch
    dc_hip = dca.DiffusionClassifier(m, dca.Config(**cfg))
    dc_hip.encoder.load_state_dict(dc_cpu.encoder.state_dict())
    dc_hip = dc_hip.to(DEV)
    x, lab = torch.zeros(2, 3, 32, 32), torch.tensor([1, 3])
    real = torch.randn_like
    torch.randn_like = lambda t_, **k: torch.randn(t_.shape).to(t_.device)     # same CPU noise stream for both runs
    try:
        torch.manual_seed(5)
        ref = dc_cpu.sample(x, lab)
        torch.manual_seed(5)
        got = dc_hip.sample(x.to(DEV), lab.to(DEV)).cpu()
    finally:
        torch.randn_like = real
    assert (got - ref).abs().max().item() < 2e-3       # 4 chained fp32 forwards with a clip in between


@pytest.mark.parametrize("kind", ["unet_v", "dit_eps"])
def test_sample_batch2_plan_with_fused_step_matches_the_two_call_path(kind):
    """`sample` on a HIP backbone = one batch-2 plan launch (class token || null token) + one dc_ddpm_step per step; the reference's
    form (two backbone calls + the elementwise torch expressions, :255-266) on the SAME backbone must give the same images."""
    cfg = dict(BASE, cfg_w=2.0, sampling_steps=3, classes=4, pred_param="v" if kind == "unet_v" else "eps")
    if kind == "unet_v":
        m, _ = make_pair(dca.small_unet_kwargs(), seed=33)
        dc = dca.DiffusionClassifier(m, dca.Config(**cfg)).to(DEV)
        x = torch.rand(3, 3, 32, 32) * 2 - 1
    else:
        dc, _ = _small_dit_classifiers(dict(cfg, encoder_type="DiT", image_size=16, noise_d=16))
        x = torch.rand(3, 4, 16, 16) * 2 - 1
    lab = torch.tensor([1, 3, 0])
    bb = dc.ema.ema_model
    assert hasattr(bb, "forward_pair")
    outs = []
    for fused in (True, False):
        torch.manual_seed(7)
        if not fused:
            real = type(bb).forward_pair
            del type(bb).forward_pair                   # no pair entry point: `sample` takes the two-call path
        try:
            outs.append(dc.sample(x.to(DEV), lab.to(DEV), from_t=0.8).cpu())
        finally:
            if not fused:
                type(bb).forward_pair = real
    assert outs[0].shape == x.shape and torch.isfinite(outs[0]).all()
    assert (outs[0] - outs[1]).abs().max().item() < 3e-4, (outs[0] - outs[1]).abs().max().item()    # fp32 plans of different launch shapes


def _small_dit_classifiers(cfg, seed=13):
    kw = dict(num_attention_heads=2, attention_head_dim=32, in_channels=4, num_layers=2, sample_size=16, patch_size=4, num_embeds_ada_norm=10)
    torch.manual_seed(seed)
    m = dca.DiT(**kw)
    with torch.no_grad():
        for _, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    o = oracle.OracleDiT(**kw)
    o.load_state_dict(m.state_dict())
    return dca.DiffusionClassifier(m, dca.Config(**cfg)).to(DEV), oracle.OracleDiffusionClassifier(o, oracle.AttrBag(**cfg))


@pytest.mark.parametrize("variant", ["dit_three_stage_v", "dit_fast", "unet_one_cell", "unet_three_stage_odd_batch"])
def test_classify_f32_schedule_corners_match_oracle(variant):
    """Corners of the scoring loop the main variants do not reach: the label-table (DiT) encoder under three pruning stages with
    v-prediction and under fast mode (3 of 6 classes), a 1 image x 2 classes x 1 trial grid, three stages over 9 classes with an odd
    batch — per-cell errors within 1e-4, the same cells evaluated, identical labels."""
    dcfg = dict(BASE, encoder_type="DiT", image_size=16, noise_d=16)
    fast, lab, fsel = False, None, None
    if variant == "dit_three_stage_v":
        dc, oc = _small_dit_classifiers(dict(dcfg, classes=8, pred_param="v", n_stages=3, evaluation_per_stage=[1, 3, 6], n_keep_per_stage=[4, 2, 1]))
        x, T = torch.rand(3, 4, 16, 16) * 2 - 1, 6
    elif variant == "dit_fast":
        dc, oc = _small_dit_classifiers(dict(dcfg, classes=6, n_fast_classes=3, evaluation_per_stage=[4]))
        x, T = torch.rand(3, 4, 16, 16) * 2 - 1, 4
        fast, lab, fsel = True, torch.tensor([1, 5, 0]), torch.tensor([[0, 3], [1, 2], [4, 0]])
    elif variant == "unet_one_cell":
        dc, oc = _classifiers(dca.small_unet_kwargs(), dict(BASE, classes=2, evaluation_per_stage=[1]), seed=21)
        x, T = torch.rand(1, 3, 32, 32) * 2 - 1, 1
    else:
        dc, oc = _classifiers(dca.small_unet_kwargs(), dict(BASE, classes=9, n_stages=3, evaluation_per_stage=[1, 2, 4], n_keep_per_stage=[5, 2, 1]), seed=22)
        x, T = torch.rand(5, 3, 32, 32) * 2 - 1, 4
    torch.manual_seed(99)
    t, eps = torch.rand(T, x.shape[0]), torch.randn(T, *x.shape)
    if fast:
        ref_l, ref_e = oc.classify(x, lab, fast=True, t=t, eps=eps, fast_select=fsel, return_errors=True)
        got_l, got_e = dc.classify(x.to(DEV), lab.to(DEV), fast=True, t=t, eps=eps.to(DEV), fast_select=fsel, return_errors=True)
    else:
        ref_l, ref_e = oc.classify(x, t=t, eps=eps, return_errors=True)
        got_l, got_e = dc.classify(x.to(DEV), t=t, eps=eps.to(DEV), return_errors=True)
    fin = torch.isfinite(ref_e)
    assert torch.equal(torch.isfinite(got_e), fin)
    assert ((got_e[fin] - ref_e[fin]).abs() / ref_e[fin]).max().item() < 1e-4
    assert got_l.cpu().tolist() == ref_l.tolist()
