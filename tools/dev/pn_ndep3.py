"""Developer probe: first plan tensor of rank 0's share (3 of 9 pairs) that differs from the single-process plan's same samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import diffusion_classifier_amd as dca
from diffusion_classifier_amd import engine as E

orig_tensor, orig_fin = E.PlanBuilder.tensor, E.PlanBuilder.finalize

def run(sim):
    names = []
    def rec(self, name, dom, H, W, Cc, dt):
        t = orig_tensor(self, name, dom, H, W, Cc, dt)
        names.append((self, t))
        return t
    def fin(self, keep_alive=()):
        extra = [t for (pb, t) in names if pb is self and t.first is not None]
        if os.environ.get("PROBE_NO_REUSE"):      # every arena tensor (quad records, workspaces too) lives to the end: no address is ever reused
            for _, _, f in self.ops:
                extra += [v for v in f.values() if isinstance(v, E.TRef) and v.base.ext is None and v.base.first is not None]
        return orig_fin(self, keep_alive=list(keep_alive) + extra)
    E.PlanBuilder.tensor, E.PlanBuilder.finalize = rec, fin
    cfg = dict(pred_param="eps", schedule="cosine", noise_d=32, image_size=32, cfg_w=0.0, ema_beta=0.999, ema_warmup=0,
               ema_update_freq=1, encoder_type="nn", classes=10, n_stages=1, evaluation_per_stage=[3], n_keep_per_stage=[1],
               n_fast_classes=2, compute_dtype="bf16", simulate_rank=sim)
    torch.manual_seed(5)
    m = dca.UNetCondition2D(**dca.cifar10_unet_kwargs())
    dc = dca.DiffusionClassifier(m, dca.Config(**cfg)).to("cuda:0")
    torch.manual_seed(6)
    BS, T = 3, 3
    x = (torch.rand(BS, 3, 32, 32) * 2 - 1).to("cuda:0")
    t, eps = torch.rand(7, BS)[:T], torch.randn(7, BS, 3, 32, 32).to("cuda:0")[:T]
    lab, err = dc.classify(x, t=t, eps=eps, return_errors=True)
    torch.cuda.synchronize()
    E.PlanBuilder.tensor, E.PlanBuilder.finalize = orig_tensor, orig_fin
    (sp,) = list(dc._score_plans.values())
    plan = sp["plan"]
    out = {}
    for pb, tt in names:
        if pb is not plan.pb or tt.first is None or tt.base is not tt or tt.H * tt.W * tt.C == 1:
            continue
        try:
            out[tt.name] = (tt.dom, plan.pb.tensor_view(tt).float().cpu().clone())
        except Exception:
            pass
        if tt.qs is not None and tt.qs[0].off is not None:
            q, parts = tt.qs
            n = plan.pb.n[tt.dom]
            nel = n * parts * (tt.C // 4) * 2
            out[tt.name + ".QS"] = (tt.dom, plan.pb.arena[q.off:q.off + nel * 4].view(torch.float32).view(n, parts, tt.C // 4, 2).cpu().clone())
    return out, plan

full, pf = run(None)
part, pp = run((0, 3))
print("families:", [ (m_["name"], m_["family"]) for m_ in pp.pb.meta if m_.get("pn")][:40])
nd = 0
for name, (dom, v) in part.items():
    if name not in full:
        print("only in part:", name); continue
    w = full[name][1]
    if dom == "bj":
        w = w[[0, 3, 6]]
    elif dom == "unit":
        w = w[[c + 10 * p for p in (0, 3, 6) for c in range(10)]]
    else:
        continue
    if v.shape != w.shape:
        print("shape", name, v.shape, w.shape); continue
    if not torch.equal(v, w):
        bad = (v != w).flatten(1).any(1).nonzero().flatten().tolist()
        print("DIFF", name, dom, tuple(v.shape), "samples", bad[:12], "max abs", (v - w).abs().max().item())
        nd += 1
        if nd >= 3:
            break
print("done", nd)
