"""MI355X-native drop-in for the scoring API of reference `diffusion/diffusion_classifier.py`.

Kept surface (same names, argument meaning, assertions and return types):
  DiffusionClassifier(backbone, config)                         reference :17-81
  .classify(x, text=None, fast=False) -> LongTensor[BS]         reference :657-725
  .evaluate(val_dataloader, stop_idx, metrics, classification)  reference :532-578
  .encode_text_prompt / .diffuse / .logsnr_schedule_cosine(_shifted)  :83-161
`config` is the reference's attribute bag (missing keys read as None).  Additive keys read
here: `compute_dtype` ("bf16" default | "f16" | "f32"), `units_per_launch`, `score_plan_cache` (launch plans kept, LRU;
default 6), `dwt_on_device` (True: `inference` applies `wavelet_dec_2(images) / 2` on the device to the batches its loader yields, on the
prefetch stream), `shard_grid` (opt-in: True
shards the (trial, image) grid of ONE replicated batch over the default process group),
`simulate_rank` ((r, N), bench.py only: time rank r's share of an N-rank sharded call on one GPU).

What differs is HOW classify runs.  The reference walks a Python double loop — T trials x C
classes sequential eager backbone calls at batch BS (:686-714).  Here the (image, trial, class)
grid is flattened into micro-batches of work units; each micro-batch is ONE native call
(`dc_run_plan`) that enqueues q_sample -> backbone -> eps-MSE on the HIP stream, with z_t / eps
materialised once per (image, trial), class-independent layers computed once per (image,
trial), and errors scattered straight into `errors[b, class, j]`.  The stage end (mean over
trials, top-k smallest, :718-721) runs ON THE DEVICE for the HIP backbones (`dc_stage_topk`,
`dc_reduce_argmin`: fixed summation order, ties to the lower class id, NaN last — identical on
every rank), and so do the next stage's work-unit maps (`dc_stage_maps`): nothing is copied to
the host between stages.  A foreign `nn.Module` backbone takes the reference's own torch ops
(`_ForeignRunner.stage_end`).

Extra keyword-only arguments (defaults keep the reference behaviour):
  t, eps         inject the per-trial draws ([T,BS] and [T,BS,C,H,W]) — parity tests
  fast_select    inject the `randint` draw of fast mode
  return_errors  also return errors[BS, classes, T] (copied to the CPU; with the multi-GPU gather's host leg under gloo
                 the only device -> host copies of a classify call)
  rng            "reference": draw rand(BS) / randn_like(x) per trial in the reference's order;
                 "philox":   t from the CPU generator, eps on device from Philox keyed by
                             (seed, image, trial) — no eps traffic, world-size independent
  group          a torch.distributed process group: shard the (trial, image) grid of THIS batch over its
                 ranks (every rank must pass the identical x; checked).  Default (None, `shard_grid`
                 unset): no grid sharding — under `accelerate launch` each rank scores its own images,
                 exactly like the reference (:615-617).
"""
import math
import os
import time

import torch
import torch.nn as nn

from .. import _lib as L
from .. import dist as D
from .._ema import EMA


def log(t, eps=1e-20):
    return torch.log(t.clamp(min=eps))


class DiffusionClassifier(nn.Module):
    def __init__(self, backbone: nn.Module, config):
        super().__init__()
        self.config = config
        pred_param = self.config.pred_param
        assert pred_param in ['v', 'eps'], "Invalid prediction parameterization. Must be 'v' or 'eps'"
        self.pred_param = pred_param
        schedule = self.config.schedule
        assert schedule in ['cosine', 'shifted_cosine'], "Invalid schedule. Must be 'cosine' or 'shifted_cosine'"
        self.schedule = self.logsnr_schedule_cosine if schedule == 'cosine' else self.logsnr_schedule_cosine_shifted
        self.noise_d = self.config.noise_d
        self.image_d = self.config.image_size
        self.cfg_w = self.config.cfg_w
        assert isinstance(backbone, nn.Module), "Model must be an instance of torch.nn.Module."
        self.model = backbone
        self.ema = EMA(self.model, beta=config.ema_beta, update_after_step=config.ema_warmup,
                       update_every=config.ema_update_freq)
        self.encoder_type = self.config.encoder_type
        if self.encoder_type == 't5':
            raise NotImplementedError("encoder_type='t5' fetches t5-base over the network; not part of the scoring path")
        elif self.encoder_type == 'nn':
            self.encoder = nn.Embedding(self.config.classes + 1, backbone.config.encoder_hid_dim)
            self.tokenizer = None
            self.null_token = self.config.classes
        elif self.encoder_type == 'DiT':
            self.tokenizer = None
            self.encoder = None
            self.null_token = self.config.classes
        print(f"Parameter count: {sum(p.numel() for p in self.model.parameters())}")
        self._score_plans = {}
        self._timed_sink = None      # bench.py: list collecting (plan, per-op event ms) instead of plain runs

    # ---- small reference-identical helpers (host side, fp32 torch like the reference) -------
    def encode_text_prompt(self, text):
        if self.encoder_type == 'nn':
            embeddings = self.encoder(text)
            embeddings.unsqueeze_(1)
        elif self.encoder_type == 'DiT':
            embeddings = text
        else:
            raise NotImplementedError(self.encoder_type)
        return embeddings

    def diffuse(self, x, alpha_t, sigma_t):
        eps_t = torch.randn_like(x)
        return alpha_t * x + sigma_t * eps_t, eps_t

    def logsnr_schedule_cosine(self, t, logsnr_min=-15, logsnr_max=15):
        logsnr_max = logsnr_max + math.log(self.noise_d / self.image_d)
        logsnr_min = logsnr_min + math.log(self.noise_d / self.image_d)
        t_min = math.atan(math.exp(-0.5 * logsnr_max))
        t_max = math.atan(math.exp(-0.5 * logsnr_min))
        return -2 * log(torch.tan((t_min + t * (t_max - t_min)).clone().detach()))

    def logsnr_schedule_cosine_shifted(self, t):
        return self.logsnr_schedule_cosine(t) + 2 * math.log(self.noise_d / self.image_d)

    # ---- the hot path ---------------------------------------------------------------------
    @torch.no_grad()
    def classify(self, x, text=None, fast=False, *, t=None, eps=None, fast_select=None, return_errors=False,
                 rng="reference", seed=0, group=None):
        cfg = self.config
        assert self.encoder_type is not None, "Encoder must be provided for classification."
        assert len(cfg.evaluation_per_stage) == cfg.n_stages, "Number of evaluations per stage must match the number of stages."
        assert len(cfg.n_keep_per_stage) == cfg.n_stages, "Number of classes to keep per stage must match the number of stages."
        assert cfg.n_keep_per_stage[-1] == 1, "Only one class should be selected at the end of the classification process."
        assert cfg.n_fast_classes <= cfg.classes and cfg.n_fast_classes >= 2, "Number of fast classes must be less than or equal to the total number of classes. Must be at least 2."
        assert rng in ("reference", "philox")
        backbone = self.ema.ema_model
        ends = [0] + list(cfg.evaluation_per_stage)
        T, ncls = ends[-1], cfg.classes
        BS = x.shape[0]
        # grid sharding is opt-in: the reference shards the DATALOADER over ranks, so an initialised process group alone
        # must not make ranks mix the errors of different images
        shard = group is not None or cfg.shard_grid is True
        rank, ws = D.world(group) if shard else (0, 1)
        # bench.py --simulate-rank r/N (measurement only): score exactly rank r's share of an N-rank grid-sharded call on this one
        # GPU, without a process group — no replication check, no gather, so the labels of such a call mean nothing
        sim = cfg.simulate_rank
        if sim:
            rank, ws = int(sim[0]), int(sim[1])
        elif ws > 1:
            D.assert_replicated(x, group)

        # candidate classes per image (host, exactly the reference's ops :671-679)
        if fast:
            text_c = text.detach().cpu().view(-1, 1)
            classes = torch.arange(ncls).repeat(BS, 1)
            wrong = classes[(classes == text_c) == False].view(BS, -1)  # noqa: E712
            sel = fast_select.cpu() if fast_select is not None else torch.randint(0, wrong.shape[1], (BS, cfg.n_fast_classes - 1))
            classes = torch.cat((text_c, torch.gather(wrong, 1, sel)), dim=1)
        else:
            classes = torch.arange(ncls).repeat(BS, 1)

        # per-trial draws.  reference order: t = rand(BS) then eps = randn_like(x), trial by trial (:688-692, :113)
        eps_of = {}
        if t is not None:
            t_all = torch.as_tensor(t, dtype=torch.float32).cpu().reshape(T, BS)
        elif rng == "reference":
            t_rows = []
            for j in range(T):
                t_rows.append(torch.rand(BS))
                if eps is None:
                    eps_of[j] = torch.randn_like(x)
            t_all = torch.stack(t_rows)
        else:
            t_all = torch.rand(T, BS)
        if eps is not None:
            eps_of = {j: eps[j] for j in range(T)}
        elif rng == "reference" and not eps_of:
            eps_of = {j: torch.randn_like(x) for j in range(T)}
        # fp32 on the host, one [BS] vector per trial exactly like the reference (:689-691): torch's CPU
        # kernels take different (vector / scalar-tail) code paths by element position, so evaluating the
        # whole [T,BS] grid at once would differ from the reference in the last bit.
        if rng == "philox" and t is None:
            # throughput mode draws its own t anyway: one vectorised evaluation (last-bit differences from the
            # per-trial form are irrelevant here and ~2 ms of host time per call are not)
            logsnr = self.schedule(t_all)
            alpha_all, sigma_all = torch.sqrt(torch.sigmoid(logsnr)), torch.sqrt(torch.sigmoid(-logsnr))
        else:
            logsnr = torch.stack([self.schedule(t_all[j].clone()) for j in range(T)])
            alpha_all = torch.stack([torch.sqrt(torch.sigmoid(logsnr[j].clone())) for j in range(T)])
            sigma_all = torch.stack([torch.sqrt(torch.sigmoid(-logsnr[j].clone())) for j in range(T)])
        draws = dict(logsnr=logsnr, alpha=alpha_all, sigma=sigma_all,
                     eps_of=eps_of, philox=(rng == "philox" and eps is None), seed=int(seed))

        if hasattr(backbone, "make_plan"):
            runner = _HipRunner(self, backbone, x, T, draws)           # libdcamd; raises without GPU / library
        else:
            runner = _ForeignRunner(self, backbone, x, T, draws)       # user-supplied nn.Module, eager torch
        for i in range(cfg.n_stages):
            pairs = D.local_pairs(ends[i], ends[i + 1], BS, rank, ws)
            runner.run_stage(pairs, classes, stage=(ends[i], rank, ws))
            errors = runner.errors()
            if not sim:
                D.gather_stage_errors(errors, ends[i], ends[i + 1], rank, ws, group=group)
            # stage end (:718-721), identically on every rank: mean over the trials so far, the k smallest classes per image.
            # HIP backbones: on the device (dc_stage_topk / dc_reduce_argmin; the next stage's work-unit maps are built
            # there too), so a multi-stage / fast classify never copies errors to the host between stages.
            classes = runner.stage_end(errors, ends[i + 1], cfg.n_keep_per_stage[i], last=i == cfg.n_stages - 1)
        assert classes.shape[1] == 1, "Only one class should be selected at the end of the classification process."
        out = classes[:, 0].to(device=x.device, dtype=torch.int64)
        if return_errors:
            err_host = errors.cpu()              # (synchronises: the cheap moment to look at the device-side failure counter)
            self.check_device_errors()
            return out, err_host
        return out

    def check_device_errors(self):
        """Raise if a producer-side-GroupNorm launch (csrc/epi_pn.h) gave up waiting for the other workgroups of a sample since the last
        check: its outputs were NaN-poisoned, so the scores of that call are invalid.  Synchronises with the device; called where the
        host synchronises anyway — `classify(return_errors=True)`, the end of `evaluate`, bench.py, the smoke test."""
        if torch.cuda.is_available() and hasattr(self.ema.ema_model, "make_plan"):
            n = int(L.lib().dc_pn_timeouts())
            if n:
                raise L.DcamdError(f"{n} wave(s) timed out inside a producer-side GroupNorm launch: the results of this call are invalid")

    # ---- callers of the hot path (reference :532-578) ----------------------------------------
    @torch.no_grad()
    def evaluate(self, val_dataloader, stop_idx=None, metrics=None, classification=False, from_t=1):
        val_samples, batches = [], []
        for idx, batch in enumerate(val_dataloader):
            batch = {k: v for k, v in batch.items()}
            x = batch["images"]
            p = batch["prompt"] if "prompt" in batch.keys() else None
            sample = self.classify(x, p, fast=self.config.fast_classification) if classification else self.sample(x, p, from_t)
            if metrics is not None:
                for metric in metrics:
                    metric.update((sample, batch))
            val_samples.append(sample)
            batches.append(batch)
            if stop_idx is not None and idx == stop_idx:
                break
        self.check_device_errors()
        return val_samples, batches, metrics

    def prefetch_to_device(self, loader, dev):
        """The batches of `loader` on `dev`, one batch AHEAD of the consumer: while batch i is being scored on the
        current stream, batch i+1 is copied host -> HBM on a side stream (pinned staging, non_blocking) and — with the additive config
        key `dwt_on_device=True`, for loaders that yield the RAW [-1, 1] images — transformed there by the HIP Haar kernel,
        `images := wavelet_dec_2(images) / 2`, what reference dataset/chexpert.py:146-147 does per item on the host before
        `evaluate` (:555-563) sees the batch.  The consumer's stream waits on the batch's event, so results are those of the
        unpipelined loop.  `self._prefetch_log` keeps (copy start, copy end, handed over) events per batch for the tests."""
        from ..utils.wavelet import wavelet_dec_2
        side = torch.cuda.Stream(device=dev)
        dwt = self.config.dwt_on_device is True
        self._prefetch_log = []

        def stage(batch):
            out = {}
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(side):
                e0.record(side)
                for k, v in batch.items():
                    if not torch.is_tensor(v):
                        out[k] = v
                        continue
                    d = v if v.is_cuda else (v if v.is_pinned() else v.pin_memory()).to(dev, non_blocking=True)
                    if dwt and k == "images":
                        d = wavelet_dec_2(d, scale=0.5)          # enqueued on the side stream (L.stream_ptr() = the current stream)
                    out[k] = d
                e1.record(side)
            return out, e0, e1

        it = iter(loader)
        first = next(it, None)
        nxt = stage(first) if first is not None else None
        while nxt is not None:
            cur, e0, e1 = nxt
            following = next(it, None)
            nxt = stage(following) if following is not None else None     # batch i+1 starts moving before batch i is handed over
            main = torch.cuda.current_stream(dev)
            main.wait_event(e1)
            for v in cur.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(main)
            eh = torch.cuda.Event(enable_timing=True)
            eh.record(main)
            self._prefetch_log.append((e0, e1, eh))
            yield cur

    # ---- generation (reference :163-293) ---------------------------------------------------
    # HIP backbones: ONE batch-2 plan launch per step (class token || null token, the class-independent layers once per
    # image) and ONE fused sampler-step kernel (`dc_ddpm_step`); a foreign nn.Module takes the reference's two eager calls
    # and its elementwise torch expressions (`ddpm_sampler_step`, kept for that path and as the fused kernel's statement).
    def clip(self, x):
        return torch.clamp(x, -1, 1)

    @torch.no_grad()
    def ddpm_sampler_step(self, z_t, pred, u_pred, logsnr_t, logsnr_s):
        c = -torch.special.expm1(logsnr_t - logsnr_s)
        alpha_t, alpha_s = torch.sqrt(torch.sigmoid(logsnr_t)), torch.sqrt(torch.sigmoid(logsnr_s))
        sigma_t, sigma_s = torch.sqrt(torch.sigmoid(-logsnr_t)), torch.sqrt(torch.sigmoid(-logsnr_s))
        w = self.cfg_w
        pred = (1 + w) * pred - w * u_pred                                   # classifier-free guidance mix
        x_pred = alpha_t * z_t - sigma_t * pred if self.pred_param == 'v' else (z_t - sigma_t * pred) / alpha_t
        x_pred = self.clip(x_pred)
        mu = alpha_s * (z_t * (1 - c) / alpha_t + c * x_pred)
        return mu, (sigma_s ** 2) * c

    def _fused_sampler_step(self, backbone, z_t, pair, lam_t, lam_s, noise):
        """`ddpm_sampler_step` + the update `z = mu + noise * sqrt(var)` (noise None: the clipped mean of the last pass) on the
        prediction pair of `forward_pair`, as one kernel.  The step's scalars are formed by the same fp32 torch ops as above."""
        lt, ls = lam_t.detach().float().cpu().reshape(()), lam_s.detach().float().cpu().reshape(())
        c = -torch.special.expm1(lt - ls)
        alpha_t, alpha_s = torch.sqrt(torch.sigmoid(lt)), torch.sqrt(torch.sigmoid(ls))
        sigma_t, sigma_s = torch.sqrt(torch.sigmoid(-lt)), torch.sqrt(torch.sigmoid(-ls))
        sd = torch.sqrt((sigma_s ** 2) * c)
        N, Cc, H, W = z_t.shape
        z = z_t.detach().to(torch.float32).contiguous()
        out = torch.empty_like(z)
        patch = int(getattr(backbone.config, "patch_size", 0) or 0)
        oc = int(getattr(backbone.config, "out_channels", Cc) or Cc)
        if oc != Cc:        # dc_ddpm_step indexes the prediction with z's channel count as the feature stride
            raise L.DcamdError(f"fused sampler step needs out_channels == in_channels (got {oc} vs {Cc})")
        p = L.DdpmStepParams(z=z.data_ptr(), pred=pair.data_ptr(), noise=None if noise is None else noise.data_ptr(), out=out.data_ptr(),
                             n=N, C=Cc, H=H, W=W, ld=pair.shape[-1], patch=patch, v_param=int(self.pred_param == 'v'),
                             w=float(self.cfg_w), one_plus_w=float(1.0 + float(self.cfg_w)), alpha_t=float(alpha_t), sigma_t=float(sigma_t), alpha_s=float(alpha_s), c=float(c), sd=float(sd))
        L.check(L.lib().dc_ddpm_step(p, L.stream_ptr()), "dc_ddpm_step")
        return out

    @torch.no_grad()
    def sample(self, x, text=None, from_t=1):
        """Ancestral DDPM sampling with classifier-free guidance (reference :210-293), same RNG consumption order as the reference."""
        dev = x.device
        if from_t == 1:
            z_t = torch.randn(x.shape).to(dev)
        else:
            lam = self.schedule(torch.ones(x.shape[0]) * from_t).to(dev)
            z_t, _ = self.diffuse(x, torch.sqrt(torch.sigmoid(lam)).view(-1, 1, 1, 1), torch.sqrt(torch.sigmoid(-lam)).view(-1, 1, 1, 1))
        cond = null = None
        if text is not None and self.encoder_type is not None:
            cond = self.encode_text_prompt(text).to(dev)
            null = self.encode_text_prompt(torch.full_like(text, self.null_token)).to(dev)
        backbone = self.ema.ema_model
        fused = hasattr(backbone, "forward_pair") and cond is not None and z_t.is_cuda
        steps = torch.linspace(from_t, 0.0, self.config.sampling_steps + 1)
        n = len(steps) - 1
        for i in range(n + 1):                                                # the last pass repeats step n-1 and keeps the mean
            u_t, u_s = (steps[i], steps[i + 1]) if i < n else (steps[-2], steps[-1])
            lam_t, lam_s = self.schedule(u_t).to(dev).unsqueeze(0), self.schedule(u_s).to(dev).unsqueeze(0)
            if fused:
                pair = backbone.forward_pair(z_t, lam_t, cond, null)
                if i == n:
                    return self._fused_sampler_step(backbone, z_t, pair, lam_t, lam_s, None).to(z_t.dtype)
                noise = torch.randn_like(z_t).to(torch.float32).contiguous()
                z_t = self._fused_sampler_step(backbone, z_t, pair, lam_t, lam_s, noise).to(z_t.dtype)
                continue
            pred = self.ema(z_t, lam_t, encoder_hidden_states=cond)
            u_pred = self.ema(z_t, lam_t, encoder_hidden_states=null)
            mu, var = self.ddpm_sampler_step(z_t, pred, u_pred, lam_t.clone().detach(), lam_s.clone().detach())
            if i == n:
                return self.clip(mu)
            z_t = mu + torch.randn_like(mu) * torch.sqrt(var)

    # ---- inference driver and checkpoint ingest (reference :581-655, :769-805) ----
    @torch.no_grad()
    def inference(self, optimizer=None, train_dataloader=None, val_dataloader=None, lr_scheduler=None, metrics=None,
                  plot_function=None, classification=False, from_t=1, checkpoint_folder="checkpoints"):
        """Same arguments and return value as the reference.  No accelerate object: the process's current HIP
        device is used, batches are moved to it, metrics are summed over `torch.distributed` when initialised —
        unless `config.shard_grid` is True (every rank then scored the same images: the counters are already global).
        A caller that grid-shards through `classify(..., group=pg)` instead of the config key drives `classify` itself
        and owns its metric reduction; this driver only knows the config key."""
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        if self.config.experiment_path is not None:
            os.makedirs(os.path.join(self.config.experiment_path, "inference_images/"), exist_ok=True)
            ckpt = os.path.join(self.config.experiment_path, checkpoint_folder)
            if os.path.isdir(ckpt):
                self.load_checkpoint(ckpt)
        self.to(dev)
        if metrics is not None:
            for metric in metrics:
                metric.set_device(dev)
        self.model.eval()

        on_device = (lambda loader: self.prefetch_to_device(loader, dev)) if dev.type == "cuda" else \
            (lambda loader: ({k: v for k, v in b.items()} for b in loader))
        val_samples, batches, metrics = self.evaluate(on_device(val_dataloader), metrics=metrics,
                                                      stop_idx=self.config.evaluation_batches,
                                                      classification=classification, from_t=from_t)
        metric_output = []
        if metrics is not None:
            for metric in metrics:
                # grid sharding scores ONE replicated batch stream on all ranks: the counters are already global, summing
                # them over ranks would multiply them by the world size.  Without it the ranks saw different images
                # (the reference's dataloader sharding) and the counters are summed (utils/metrics.py:56-58).
                if self.config.shard_grid is not True:
                    metric.sync_across_processes(None)
                metric_output.append(metric.get_output())
        if plot_function is not None and not classification:
            plot_function(output_dir=os.path.join(self.config.experiment_path, "inference_images/"), batches=batches,
                          samples=val_samples, epoch=0, process_idx=D.world()[0])
        return (metric_output, val_samples, batches) if metrics is not None else (val_samples, batches)

    # accelerate.save_state layout written by the reference's training run (:382-386, :741): the prepared modules in
    # registration order -> model.safetensors (backbone), model_1.safetensors (EMA wrapper: `ema_model.*`,
    # `online_model.*`, `initted`, `step`), model_2.safetensors (nn.Embedding encoder: `weight`); plus
    # experiment_state.pth {epoch, best_metric, experiment_key} (:744-752).
    _CKPT_FILES = ("model", "model_1", "model_2")

    @staticmethod
    def _read_state(directory, stem):
        """`<stem>.safetensors`, or accelerate's safe_serialization=False spelling `pytorch_<stem>.bin`, under `directory`."""
        st = os.path.join(directory, stem + ".safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            return load_file(st)
        alt = os.path.join(directory, "pytorch_" + stem + ".bin")
        if os.path.exists(alt):
            return torch.load(alt, map_location="cpu", weights_only=True)
        return None

    def load_checkpoint(self, checkpoint_path, accelerator=None):
        """Ingest a reference checkpoint directory (reference :769-805).  Returns the reference's triple
        (epoch, best_metric, experiment_key); (None, None, None) when experiment_state.pth is absent."""
        sd = self._read_state(checkpoint_path, "model")
        if sd is None:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {checkpoint_path}")
        self.model.load_state_dict(sd)
        ema_sd = self._read_state(checkpoint_path, "model_1")
        if ema_sd is None:
            # classification always runs on the EMA copy: scoring silently with the online weights would be a wrong answer
            raise FileNotFoundError(f"no model_1.safetensors / pytorch_model_1.bin (EMA weights) under {checkpoint_path}")
        inner = {k[len("ema_model."):]: v for k, v in ema_sd.items() if k.startswith("ema_model.")}
        self.ema.ema_model.load_state_dict(inner)
        for k in ("initted", "step"):
            if k in ema_sd:
                getattr(self.ema, k).copy_(ema_sd[k].reshape(()))
        enc_sd = self._read_state(checkpoint_path, "model_2")
        if self.encoder is not None:
            if enc_sd is None:
                raise FileNotFoundError(f"no model_2.safetensors (class-token encoder) under {checkpoint_path}")
            self.encoder.load_state_dict(enc_sd)
        st = os.path.join(checkpoint_path, "experiment_state.pth")
        if os.path.exists(st):
            state = torch.load(st, map_location="cpu", weights_only=False)
            print(f"Checkpoint loaded. Resuming from epoch {state.get('epoch')}. Best metric {state.get('best_metric')}")
            return state.get("epoch"), state.get("best_metric"), state.get("experiment_key")
        return None, None, None

    def save_checkpoint(self, checkpoint_dir, epoch=0, best_metric=None, experiment_key=None):
        """Write the same directory layout (used for round trips and for handing weights back to the reference)."""
        from safetensors.torch import save_file
        os.makedirs(checkpoint_dir, exist_ok=True)
        cpu = lambda sd: {k: v.detach().cpu().contiguous() for k, v in sd.items()}
        save_file(cpu(self.model.state_dict()), os.path.join(checkpoint_dir, "model.safetensors"))
        ema_sd = {"ema_model." + k: v for k, v in self.ema.ema_model.state_dict().items()}
        ema_sd.update({"online_model." + k: v for k, v in self.model.state_dict().items()})
        ema_sd.update(initted=self.ema.initted.reshape(1), step=self.ema.step.reshape(1))
        save_file(cpu(ema_sd), os.path.join(checkpoint_dir, "model_1.safetensors"))
        if self.encoder is not None:
            save_file(cpu(self.encoder.state_dict()), os.path.join(checkpoint_dir, "model_2.safetensors"))
        torch.save({"epoch": epoch + 1, "best_metric": best_metric, "experiment_key": experiment_key},
                   os.path.join(checkpoint_dir, "experiment_state.pth"))


class _ForeignRunner:
    """Backbones that are plain nn.Modules (not this package's HIP backbones): the reference's
    semantics in eager torch on x.device, one backbone call per (trial, class column) at batch BS
    (reference :686-714).  This is NOT a fallback for UNetCondition2D / DiT — those always take
    _HipRunner and raise when libdcamd or the GPU is missing."""

    def __init__(self, dc, backbone, x, T, draws):
        if draws["philox"]:
            raise L.DcamdError("rng='philox' needs a HIP backbone (UNetCondition2D / DiT)")
        self.dc, self.bb, self.x, self.T, self.d = dc, backbone, x, T, draws
        self.err = torch.full((x.shape[0], dc.config.classes, T), float("inf"), device=x.device)

    def errors(self):
        return self.err

    def stage_end(self, errors, t_end, num_keep, last=False):
        # identical torch ops to the reference (:718-721)
        end_of_stage_errors = errors.cpu()[:, :, :t_end].mean(dim=2)
        _, keep_indices = torch.topk(end_of_stage_errors, num_keep, dim=1, largest=False)
        return keep_indices

    def run_stage(self, pairs, classes, stage=None):
        dc, x, d = self.dc, self.x, self.d
        by_trial = {}
        for j, b in pairs:
            by_trial.setdefault(j, []).append(b)
        for j, bl in by_trial.items():
            bs = torch.tensor(bl)
            full = bl == list(range(x.shape[0]))       # the reference's own batch: use the tensors as they are
            al = (d["alpha"][j] if full else d["alpha"][j, bs]).view(-1, 1, 1, 1).to(x.device)
            sg = (d["sigma"][j] if full else d["sigma"][j, bs]).view(-1, 1, 1, 1).to(x.device)
            lam = (d["logsnr"][j] if full else d["logsnr"][j, bs]).to(x.device)
            e = d["eps_of"][j]
            e = (e if full else e[bs.to(e.device)]).to(x.device)
            z = al * (x if full else x[bs.to(x.device)]) + sg * e
            for c in range(classes.shape[1]):
                lab = classes[bs, c].to(x.device)
                emb = dc.encode_text_prompt(lab)
                pred = self.bb(x=z, noise_labels=lam, encoder_hidden_states=emb)
                eps_pred = sg * z + al * pred if dc.pred_param == 'v' else pred
                err = torch.norm((eps_pred - e).view(len(bl), -1), dim=1, p=2) ** 2
                self.err[bs.to(x.device), lab, j] = err


class _HipRunner:
    """Micro-batched execution of the (image, trial, class) grid through libdcamd plans."""

    def __init__(self, dc, backbone, x, T, draws):
        self.lib = L.require_gpu()
        self.dc, self.bb, self.x, self.T, self.d = dc, backbone, x, T, draws
        self.dev = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
        self.x_dev = x.detach().to(self.dev, torch.float32).contiguous()
        self.err_dev = None
        cfg = dc.config
        dt = cfg.compute_dtype or "bf16"
        if getattr(backbone, "compute_dtype", None) != dt:
            backbone.set_compute_dtype(dt)
        self.dt = dt
        if draws["philox"]:
            assert (x.shape[1] * x.shape[2] * x.shape[3]) % 4 == 0

    def _units_per_launch(self, H, W, k):
        u = self.dc.config.units_per_launch
        if u is None:
            # ~8M output pixel rows per launch at full resolution (measured on cfg2: 1M -> 4M = +13 %, 4M -> 8M = +2.4 %), but never fewer than 192
            # units: the deep levels of a 256x256 UNet see only units x 64 rows, and 64-unit launches left their GEMMs at
            # 0.2-0.35 PF (IPMSA: 1.14 -> 1.27 img/s; ~200 MB of arena per unit, far inside 288 GB)
            u = max(192, (1 << 23) // (H * W))
        return max(k, int(u))

    def _plan(self, n_bj, k):
        dc, dev = self.dc, self.dev
        cfg = dc.config
        BS, Cc, H, W = self.x.shape
        wver = getattr(self.bb, "_wver", 0)
        key = (BS, n_bj, k, self.dt, str(dev), (Cc, H, W), bool(getattr(self.bb, "share_trunk", True)), id(self.bb),
               self.T, cfg.classes, wver)
        sp = dc._score_plans.get(key)
        if sp is not None:
            dc._score_plans[key] = dc._score_plans.pop(key)      # most recently used last
            return sp
        # plans hold raw pointers into the packed weights they were built from: entries of an older weights version of this
        # backbone (load_state_dict / .to() / EMA.update since) are stale — drop them so their arenas are freed
        for old in [k_ for k_ in dc._score_plans if k_[7] == id(self.bb) and k_[10] != wver]:
            del dc._score_plans[old]
        U, T = n_bj * k, self.T
        # one int32 control block: | pair_id (int64 x n_bj) | lam | alpha | sigma | img_of_bj | ctx_of_unit | out_index |
        words = 2 * n_bj + 4 * n_bj + 2 * U
        ctl = torch.zeros(words, dtype=torch.int32, device=dev)
        o = [0]

        def take(n):
            v = ctl[o[0]:o[0] + n]
            o[0] += n
            return v
        pair_id = take(2 * n_bj).view(torch.int64)
        lam, alpha, sigma = (take(n_bj).view(torch.float32) for _ in range(3))
        img_of_bj, ctx_of_unit, out_index = take(n_bj), take(U), take(U)
        score = dict(
            x=torch.zeros((BS, Cc, H, W), dtype=torch.float32, device=dev),
            eps=torch.zeros((n_bj, Cc, H, W), dtype=torch.float32, device=dev),
            errors=torch.full((BS * cfg.classes * T + 1,), float("inf"), dtype=torch.float32, device=dev),
            lam=lam, alpha=alpha, sigma=sigma, img_of_bj=img_of_bj, ctx_of_unit=ctx_of_unit, out_index=out_index,
            v_param=dc.pred_param == 'v')
        plan = self.bb.make_plan(n_bj, k, cfg.classes, dev, score=score)
        sp = dict(plan=plan, score=score, ctl=ctl, pair_id=pair_id, words=words, n_bj=n_bj, k=k, U=U)
        dc._score_plans[key] = sp
        # each entry owns an arena, a workspace and an errors buffer in HBM, and n_bj follows the pair count (per rank, per stage, per
        # last batch of a dataloader): keep the most recently used few (a multi-stage classify alternates between one plan per stage).
        # (Scores that do not depend on n_bj — i.e. on the world size or the micro-batch split — rest on every kernel giving a sample the
        # same bits wherever it sits in a launch and whichever tile shape the launch size selects; tests/test_gpu_dist.py and
        # test_cfg2_full_grid_properties_at_bench_size hold that, and caught the one epilogue branch that did not in round 4.)
        cap = int(getattr(cfg, "score_plan_cache", None) or 6)
        while len(dc._score_plans) > max(cap, 1):
            del dc._score_plans[next(iter(dc._score_plans))]
        return sp

    def errors(self):
        BS, ncls = self.x.shape[0], self.dc.config.classes
        if self.err_dev is None:   # this rank owned no trial so far
            self.err_dev = torch.full((BS * ncls * self.T + 1,), float("inf"), dtype=torch.float32, device=self.dev)
        return self.err_dev[:-1].view(BS, ncls, self.T)

    def stage_end(self, errors, t_end, num_keep, last=False):
        """Mean over trials [0, t_end) and the num_keep smallest classes per image, on the device.  Returns [BS, num_keep]:
        int64 labels for the last stage (dc_reduce_argmin), else the int32 class lists the next stage's maps are built from."""
        BS, ncls, T = errors.shape
        assert errors.is_contiguous() and errors.dtype == torch.float32
        if last and num_keep == 1:
            keep = torch.empty((BS, 1), dtype=torch.int64, device=errors.device)
            L.check(self.lib.dc_reduce_argmin(errors.data_ptr(), BS, ncls, T, t_end, keep.data_ptr(), None, L.stream_ptr()), "dc_reduce_argmin")
        else:
            keep = torch.empty((BS, num_keep), dtype=torch.int32, device=errors.device)
            L.check(self.lib.dc_stage_topk(errors.data_ptr(), BS, ncls, T, t_end, num_keep, keep.data_ptr(), None, L.stream_ptr()), "dc_stage_topk")
        return keep

    def run_stage(self, pairs, classes, stage=None):
        """pairs: this rank's (trial, image) pairs of the stage = D.local_pairs(t0, t1, BS, rank, world); stage = (t0, rank, world)."""
        if not pairs:
            return
        dc, d, dev, T = self.dc, self.d, self.dev, self.T
        BS, Cc, H, W = self.x.shape
        ncls, k = dc.config.classes, classes.shape[1]
        on_dev = classes.is_cuda                         # stages >= 1: the surviving classes never left the device
        n_bj = min(len(pairs), max(1, self._units_per_launch(H, W, k) // k))
        # equal micro-batches: 800 pairs at a cap of 256 run as 4 x 200, not 3 x 256 + 32 padded to 256 (slots past the last pair
        # repeat a pair into the dump cell: 28 % wasted work on the CheXpert workload at 8 images per step)
        n_bj = -(-len(pairs) // -(-len(pairs) // n_bj))
        sp = self._plan(n_bj, k)
        plan, score, U = sp["plan"], sp["score"], sp["U"]
        if self.err_dev is None:
            score["errors"].fill_(float("inf"))
        elif score["errors"] is not self.err_dev:
            score["errors"].copy_(self.err_dev)
        self.err_dev = score["errors"]
        score["x"].copy_(self.x_dev)
        if dc.encoder is not None:
            plan.ctx.copy_(dc.encoder.weight[:ncls].detach().to(dev, torch.float32))
        plan.run_ctx()                                   # per-class vectors: once per stage, not per micro-batch
        n_mb = -(-len(pairs) // n_bj)
        host = sp.get("host")                            # pinned staging, reused across calls (pin_memory() is slow)
        ev = sp.get("host_ev")
        if ev is not None:
            ev.synchronize()                             # the previous call's asynchronous control-block copies have left `host`
        if host is None or host.shape[0] < n_mb:
            host = sp["host"] = torch.zeros((n_mb, sp["words"]), dtype=torch.int32).pin_memory()
        dump = BS * ncls * T
        # the index part of every control block (pair ids, image / class / output maps) depends on (pairs, classes) only:
        # built once and kept (stage 0 of every call sees the same pairs and the full class list); per call only the
        # three float rows (lambda, alpha, sigma of this call's t draws) are gathered.  Host time here is GPU idle time.
        ck = (n_bj, len(pairs), pairs[0], pairs[-1], T, ncls, b"dev" if on_dev else classes.numpy().tobytes())
        cache = sp.setdefault("idx_cache", {})
        ent = cache.get(ck)
        if ent is None:
            if len(cache) > 8:
                cache.clear()
            tmpl = torch.zeros((n_mb, sp["words"]), dtype=torch.int32)
            jsb = []
            for m in range(n_mb):
                chunk = pairs[m * n_bj:(m + 1) * n_bj]
                row = tmpl[m]
                pad = n_bj - len(chunk)
                js = torch.tensor([p[0] for p in chunk] + [chunk[0][0]] * pad)
                bs = torch.tensor([p[1] for p in chunk] + [chunk[0][1]] * pad)
                o = 2 * n_bj
                row[0:o].view(torch.int64).copy_(bs * T + js)
                o += 3 * n_bj
                row[o:o + n_bj].copy_(bs.to(torch.int32)); o += n_bj
                if not on_dev:
                    cl = classes[bs]                                       # [n_bj, k] class id of every unit
                    oi = (bs[:, None] * ncls + cl) * T + js[:, None]       # errors[b, class, j], flat
                    if pad:
                        oi[len(chunk):] = dump
                    row[o:o + U].copy_(cl.reshape(-1).to(torch.int32)); o += U
                    row[o:o + U].copy_(oi.reshape(-1).to(torch.int32))
                jsb.append((js, bs))
            ent = cache[ck] = (tmpl, jsb)
        tmpl, jsb = ent
        host[:n_mb].copy_(tmpl)
        for m in range(n_mb):
            js, bs = jsb[m]
            o = 2 * n_bj
            for src in (d["logsnr"], d["alpha"], d["sigma"]):
                host[m, o:o + n_bj].view(torch.float32).copy_(src[js, bs])
                o += n_bj
        CHW = Cc * H * W
        hw = 6 * n_bj                                    # words of a control block the host owns (pair ids, lambda / alpha / sigma, image map)
        if on_dev:
            # class-dependent maps of every micro-batch of the stage, built on the device from the surviving classes
            t0, rank, ws = stage
            maps = torch.empty((n_mb, 2 * U), dtype=torch.int32, device=dev)
            L.check(self.lib.dc_stage_maps(classes.data_ptr(), BS, ncls, T, k, t0, len(pairs), rank, ws, n_bj, n_mb, dump,
                                           maps.data_ptr(), L.stream_ptr()), "dc_stage_maps")
        for m in range(n_mb):
            chunk = pairs[m * n_bj:(m + 1) * n_bj]
            if on_dev:
                sp["ctl"][:hw].copy_(host[m, :hw], non_blocking=True)
                sp["ctl"][hw:].copy_(maps[m], non_blocking=True)
            else:
                sp["ctl"].copy_(host[m], non_blocking=True)
            if d["philox"]:
                L.check(self.lib.dc_philox_normal(score["eps"].data_ptr(), n_bj, CHW, sp["pair_id"].data_ptr(),
                                                  d["seed"], L.stream_ptr()), "dc_philox_normal")
            else:
                r = 0
                while r < len(chunk):                                  # runs of consecutive images of one trial
                    j, b0 = chunk[r]
                    r1 = r
                    while r1 < len(chunk) and chunk[r1][0] == j and chunk[r1][1] == b0 + (r1 - r):
                        r1 += 1
                    score["eps"][r:r1].copy_(d["eps_of"][j][b0:b0 + (r1 - r)].to(torch.float32), non_blocking=True)
                    r = r1
                if len(chunk) < n_bj:
                    score["eps"][len(chunk):].copy_(score["eps"][0:1].expand(n_bj - len(chunk), -1, -1, -1))
            if dc._timed_sink is not None:
                dc._timed_sink.append((plan, plan.pb.run_timed()))
            else:
                plan.run()
        # `host` (pinned) must outlive the asynchronous copies: instead of draining the stream here (the GPU then idles while the
        # host prepares the next call) an event marks the last copy, and the next user of the buffer waits for it — by then it has
        # long fired, and the host side of call i+1 runs under the kernels of call i
        if ev is None:
            ev = sp["host_ev"] = torch.cuda.Event()
        ev.record()
