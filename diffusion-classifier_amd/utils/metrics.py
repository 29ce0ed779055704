"""Counters with the interface of reference `utils/metrics.py` (Metric / Accuracy / Precision /
Recall / F1: `update((pred, batch))`, `sync_across_processes(accelerator)`, `get_output()`,
`set_device`, `reset`), fed by `DiffusionClassifier.evaluate` (reference
diffusion_classifier.py:565-568, :639-643).  The cross-rank sum of the int64 counters is the
reference's only explicit collective (`accelerator.reduce`, utils/metrics.py:56-58): here it is
an all-reduce(SUM) over `torch.distributed` (RCCL on ROCm) when no accelerate object is given.
Binary metrics treat class 1 as positive, like the reference.
"""
import torch
import torch.distributed as dist


class Metric(torch.nn.Module):
    counters = ()

    def __init__(self, name, device=torch.device("cpu")):
        super().__init__()
        self.name = name
        self.device = device
        self.required_output_keys = ()
        self.reset()

    def reset(self):
        for c in self.counters:
            setattr(self, c, torch.tensor(0, dtype=torch.int64, device=self.device))

    def set_device(self, device):
        self.device = device
        for c in self.counters:
            setattr(self, c, getattr(self, c).to(device))

    def _count(self, **masks):
        for c, m in masks.items():
            setattr(self, c, getattr(self, c) + m.sum().to(self.device))

    def update(self, output):
        raise NotImplementedError

    def compute(self):
        raise NotImplementedError

    def get_output(self, reduce=True):
        return self.compute()

    def sync_across_processes(self, accelerator=None):
        for c in self.counters:
            v = getattr(self, c)
            if accelerator is not None and hasattr(accelerator, "reduce"):
                v = accelerator.reduce(v)
            elif dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                v = v.clone()
                dist.all_reduce(v, op=dist.ReduceOp.SUM)
            setattr(self, c, v)

    def __call__(self, output):
        self.update(output)
        return self.compute()

    @staticmethod
    def _pair(output, device):
        y_pred, batch = output
        return y_pred.to(device), batch["prompt"].to(device)

    @staticmethod
    def _ratio(num, den):
        return 0.0 if den == 0 else num.float() / den.float()


class Accuracy(Metric):
    counters = ("correct", "total")

    def update(self, output):
        y_pred, y_true = self._pair(output, self.device)
        self._count(correct=(y_pred == y_true), total=torch.ones_like(y_true, dtype=torch.bool))

    def compute(self):
        return {self.name: self.correct / self.total}


class Precision(Metric):
    counters = ("tp", "fp")

    def __init__(self, name="precision", device=torch.device("cpu")):
        super().__init__(name, device)

    def update(self, output):
        y_pred, y_true = self._pair(output, self.device)
        self._count(tp=(y_pred == 1) & (y_true == 1), fp=(y_pred == 1) & (y_true == 0))

    def compute(self):
        return {self.name: self._ratio(self.tp, self.tp + self.fp)}


class Recall(Metric):
    counters = ("tp", "fn")

    def __init__(self, name="recall", device=torch.device("cpu")):
        super().__init__(name, device)

    def update(self, output):
        y_pred, y_true = self._pair(output, self.device)
        self._count(tp=(y_pred == 1) & (y_true == 1), fn=(y_pred == 0) & (y_true == 1))

    def compute(self):
        return {self.name: self._ratio(self.tp, self.tp + self.fn)}


class F1(Metric):
    counters = ("tp", "fp", "fn")

    def __init__(self, name="f1", device=torch.device("cpu")):
        super().__init__(name, device)

    def update(self, output):
        y_pred, y_true = self._pair(output, self.device)
        self._count(tp=(y_pred == 1) & (y_true == 1), fp=(y_pred == 1) & (y_true == 0), fn=(y_pred == 0) & (y_true == 1))

    def compute(self):
        return {self.name: self._ratio(2 * self.tp, 2 * self.tp + self.fp + self.fn)}
