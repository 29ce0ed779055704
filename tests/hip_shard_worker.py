"""One rank of the sharded-classify rehearsal on a single GPU (tests/test_gpu_dist.py).

`python hip_shard_worker.py RANK WORLD PORT OUT.npz [small|cfg2]`: builds the small UNet (or the CIFAR-10 UNet of BASELINE config 2, bf16) + classifier from fixed seeds, joins a gloo
group of WORLD ranks that all use cuda:0, runs a two-stage and a philox classify with grid sharding on and writes the
errors / labels it ended with.  WORLD == 1 (no process group) is the single-process result to compare with."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(rank, world, port, out, arch="small"):
    import numpy as np
    import torch
    import torch.distributed as dist
    import diffusion_classifier_amd as dca
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = "cuda:0"
    cfg = dict(pred_param="eps", schedule="cosine", noise_d=32, image_size=32, cfg_w=0.0, ema_beta=0.999, ema_warmup=0,
               ema_update_freq=1, encoder_type="nn", classes=6, n_stages=2, evaluation_per_stage=[3, 7], n_keep_per_stage=[2, 1],
               n_fast_classes=2, compute_dtype="f32", shard_grid=world > 1, units_per_launch=24)
    if arch == "cfg2":        # BASELINE config 2's architecture and dtype, 10 classes, class-shared trunk and skip halves, ragged deals at world 3
        cfg.update(classes=10, evaluation_per_stage=[3, 7], n_keep_per_stage=[4, 1], compute_dtype="bf16", units_per_launch=None)
    torch.manual_seed(5)
    m = dca.UNetCondition2D(**(dca.cifar10_unet_kwargs() if arch == "cfg2" else dca.small_unet_kwargs()))
    dc = dca.DiffusionClassifier(m, dca.Config(**cfg)).to(dev)
    torch.manual_seed(6)
    BS, T = 3, 7
    x = (torch.rand(BS, 3, 32, 32) * 2 - 1).to(dev)
    t, eps = torch.rand(T, BS), torch.randn(T, BS, 3, 32, 32).to(dev)
    lab, err = dc.classify(x, t=t, eps=eps, return_errors=True)
    lab_p, err_p = dc.classify(x, t=t, rng="philox", seed=77, return_errors=True)
    runner_cls = type(dc).__module__
    np.savez(out, lab=lab.cpu().numpy(), err=err.numpy(), lab_p=lab_p.cpu().numpy(), err_p=err_p.numpy(),
             hip=np.array([int(hasattr(dc.ema.ema_model, "make_plan"))]), module=np.array([runner_cls]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], *(sys.argv[5:6]))
