"""Training loop behind ``references/classification`` / ``references/detection`` (reference: holocron/trainer/core.py).

Same class, constructor, attributes and method names as the reference's ``Trainer`` so that its scripts run unchanged
(`references/classification/train.py:216-227`), with what the MI355X path needs folded in:

* **data parallelism**: when ``torch.distributed`` is initialised with more than one rank (one process per GPU, RCCL over xGMI),
  ``_backprop_step`` drives a ``parallel.GradReducer`` - bucketed all-reduce overlapped with backward, gradient-accumulation
  micro-steps under ``no_sync`` - before the optimizer step.  The reference has no distributed path at all.
* **no per-batch host synchronisation**: the reference formats ``batch_loss.item()`` into the progress bar every step
  (core.py:162), a device->host wait per iteration; here the loss stays on the device and is read every ``log_every`` steps.
* **bf16 is the storage type of the kernels**, so ``amp=True`` needs no ``GradScaler`` (bf16 has fp32's exponent range): the flag
  is accepted and recorded, ``scaler`` stays ``None``.
* ``save`` can include the optimizer state (``with_optimizer=True``); ``load`` keeps it and the next ``_reset_opt`` (every
  ``fit_n_epochs`` / ``find_lr`` / ``check_setup`` begins with one) re-applies it to the rebuilt parameter groups instead of
  starting the moments from zero.
* under data parallelism the decisions of a step are taken by all ranks together (NaN skip, evaluation sums) and only rank 0
  writes checkpoints.
"""
import math
import warnings
from collections import defaultdict
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import torch
from torch import Tensor, nn
from torch.optim.lr_scheduler import CosineAnnealingLR, MultiplicativeLR, OneCycleLR

from .utils import freeze_bn, freeze_model, split_normalization_params

__all__ = ["Trainer"]

ParamSeq = Sequence[nn.Parameter]


class _Bar:
    """Stand-in for fastprogress' bars (not installed on the training image): iterates, carries ``comment``."""

    def __init__(self, it, parent=None):
        self.it, self.parent, self.comment = it, parent, ""
        self.main_bar = self

    def __iter__(self):
        return iter(self.it)

    def __len__(self):
        return len(self.it)

    def write(self, msg):
        print(msg)


def _bars():
    try:
        from fastprogress import master_bar, progress_bar
        return master_bar, progress_bar
    except Exception:      # noqa: BLE001 - optional dependency
        return _Bar, _Bar


class Trainer:
    """Baseline trainer (core.py:26-104).  Arguments as in the reference; ``log_every``: how often the running loss is read back."""

    def __init__(self, model: nn.Module, train_loader, val_loader, criterion: nn.Module, optimizer: torch.optim.Optimizer,
                 gpu: Optional[int] = None, output_file: str = "./checkpoint.pth", amp: bool = False,
                 skip_nan_loss: bool = False, nan_tolerance: int = 5, gradient_acc: int = 1,
                 gradient_clip: Optional[float] = None, on_epoch_end: Optional[Callable[[Dict[str, float]], Any]] = None,
                 log_every: int = 50) -> None:
        # the attribute names are the reference's (scripts and callbacks read them)
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.train_loader, self.val_loader = train_loader, val_loader
        self.amp, self.scaler = amp, None                   # bf16 activations are always on: no GradScaler to keep
        self.skip_nan_loss, self.nan_tolerance = skip_nan_loss, nan_tolerance
        self.gradient_acc, self.grad_clip = gradient_acc, gradient_clip
        self.on_epoch_end, self.output_file = on_epoch_end, output_file
        self.log_every = max(1, int(log_every))
        self.step = self.start_epoch = self.epoch = self._grad_count = 0
        self.min_loss, self.gpu = math.inf, gpu
        self._params: Tuple[ParamSeq, ParamSeq] = ([], [])
        self.lr_recorder: List[float] = []
        self.loss_recorder: List[float] = []
        self._reducer = self._reducer_key = None
        self._pending_opt_state: Optional[Dict[str, Any]] = None
        self.set_device(gpu)
        self._reset_opt(optimizer.defaults["lr"])

    # ---------------------------------------------------------------- device / checkpoint
    def set_device(self, gpu: Optional[int] = None) -> None:
        """core.py:90-104"""
        if not isinstance(gpu, int):
            return
        if not torch.cuda.is_available():
            raise AssertionError("PyTorch cannot access your GPU. Please investigate!")      # the reference's messages
        if gpu >= torch.cuda.device_count():
            raise ValueError("Invalid device index")
        torch.cuda.set_device(gpu)
        self.model = self.model.cuda()
        if isinstance(self.criterion, nn.Module):
            self.criterion = self.criterion.cuda()

    def save(self, output_file: str, with_optimizer: bool = False) -> None:
        """Checkpoint with the reference's keys (core.py:106-121); ``with_optimizer`` adds the optimizer state."""
        if self._world() > 1 and self._rank() != 0:
            return                                  # replicas are identical: one writer, not one file race per rank
        state = {"epoch": self.epoch, "step": self.step, "min_loss": self.min_loss, "model": self.model.state_dict()}
        if with_optimizer:
            state["optimizer"] = self.optimizer.state_dict()
        torch.save(state, output_file)

    def load(self, state: Dict[str, Any]) -> None:
        """core.py:123-133 (+ the optimizer state when the checkpoint has one).

        Under data parallelism ``load`` is a COLLECTIVE, like the constructor: EVERY rank must call it (``save`` is the opposite - it
        writes on rank 0 only - so do not mirror its ``if rank == 0`` around ``load``: the other ranks would wait in the broadcast
        forever).  Whatever file each rank read, all ranks continue from RANK 0's checkpoint: parameters, buffers, epoch / step /
        min_loss and the optimizer state (moments and step counters) are broadcast from rank 0, and a rank whose checkpoint has an
        optimizer state where rank 0's has none (or the reverse) raises instead of training on with diverging moments."""
        dp = self._world() > 1
        meta = [state["epoch"], state["step"], state["min_loss"], "optimizer" in state]
        if dp:
            import torch.distributed as dist
            mine = meta[3]
            dist.broadcast_object_list(meta, src=0)
            if mine != meta[3]:
                raise RuntimeError("Trainer.load: this rank's checkpoint %s an optimizer state and rank 0's %s"
                                   % ("has" if mine else "lacks", "has one" if meta[3] else "has none"))
        self.start_epoch = self.epoch = meta[0]
        self.step, self.min_loss = meta[1], meta[2]
        self.model.load_state_dict(state["model"])
        if dp:
            from ..parallel import broadcast_parameters
            broadcast_parameters(self.model, 0)
        if "optimizer" in state:
            self.optimizer.load_state_dict(state["optimizer"])
            if dp:
                self._broadcast_optimizer_state()
            # fit_n_epochs / find_lr / check_setup rebuild the groups with an empty state first thing: keep the loaded one for them
            self._pending_opt_state = self.optimizer.state_dict() if dp else state["optimizer"]

    def _broadcast_optimizer_state(self) -> None:
        """Rank 0's optimizer state on every rank: tensors in (group, parameter, key) order, python scalars as one object list."""
        import torch.distributed as dist
        scalars, slots = [], []
        for group in self.optimizer.param_groups:
            for p in group["params"]:
                st = self.optimizer.state.get(p, {})
                for k in sorted(st):
                    v = st[k]
                    if torch.is_tensor(v):
                        dist.broadcast(v, src=0)
                    else:
                        scalars.append(v)
                        slots.append((st, k))
        if scalars:
            dist.broadcast_object_list(scalars, src=0)
            for (st, k), v in zip(slots, scalars):
                st[k] = v

    # ---------------------------------------------------------------- ranks
    @staticmethod
    def _world() -> int:
        import torch.distributed as dist
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    @staticmethod
    def _rank() -> int:
        import torch.distributed as dist
        return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0

    def _all_ranks_finite(self, loss: Tensor) -> bool:
        """``isfinite(loss)`` agreed on by every rank (MIN): a rank that skipped a step on its own would issue no bucket
        all-reduce while the others wait in theirs."""
        ok = torch.isfinite(loss.detach()).to(torch.int32)
        if self._world() > 1:
            import torch.distributed as dist
            ok = ok.reshape(1).clone()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        return bool(ok)

    def _sum_over_ranks(self, *tensors: Tensor) -> None:
        """In-place sum of evaluation accumulators over the ranks (each rank evaluates its shard of the validation set)."""
        if self._world() > 1:
            import torch.distributed as dist
            for t in tensors:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)

    # ---------------------------------------------------------------- one epoch
    def _fit_epoch(self, mb) -> None:
        """core.py:135-165"""
        freeze_bn(self.model.train())
        _, progress_bar = _bars()
        nan_cnt = 0
        pb = progress_bar(self.train_loader, parent=mb)
        for x, target in pb:
            x, target = self.to_cuda(x, target)
            batch_loss = self._get_loss(x, target)
            # `skip_nan_loss` needs the value on the host; without it nothing here waits for the device
            if not self.skip_nan_loss or self._all_ranks_finite(batch_loss):
                nan_cnt = 0
                self._backprop_step(batch_loss)
            else:
                nan_cnt += 1
                if nan_cnt > self.nan_tolerance:
                    raise ValueError(f"loss value has been NaN or inf for more than {self.nan_tolerance} steps.")
            self.scheduler.step()
            if self.step % self.log_every == 0:
                pb.comment = f"Training loss: {float(batch_loss.detach()):.4}"
            self.step += 1
        self.epoch += 1

    def to_cuda(self, x, target):
        if isinstance(self.gpu, int):
            if self.gpu >= torch.cuda.device_count():
                raise ValueError("Invalid device index")
            return self._to_cuda(x, target)
        return x, target

    @staticmethod
    def _to_cuda(x: Tensor, target: Tensor) -> Tuple[Tensor, Tensor]:
        return x.cuda(non_blocking=True), target.cuda(non_blocking=True)

    # ---------------------------------------------------------------- data parallelism
    def _grad_reducer(self):
        """The GradReducer over the parameters currently being optimised, or None on a single rank."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() <= 1:
            return None
        params = [p for g in self.optimizer.param_groups for p in g["params"] if p.requires_grad]
        key = tuple(id(p) for p in params)
        if self._reducer is None or self._reducer_key != key:
            from ..parallel import GradReducer, broadcast_parameters
            if self._reducer is not None:
                self._reducer.remove()
            else:
                # first use: replicas built from different seeds would stay different forever (the all-reduce averages
                # gradients, not weights) - everybody starts from rank 0's parameters and buffers
                broadcast_parameters(self.model)
                self._warn_unsharded_loader()
            self._reducer = GradReducer(params, overlap=True)
            self._reducer_key = key
        return self._reducer

    def _warn_unsharded_loader(self) -> None:
        from torch.utils.data.distributed import DistributedSampler
        sampler = getattr(self.train_loader, "sampler", None)
        batch_sampler = getattr(self.train_loader, "batch_sampler", None)
        inner = getattr(batch_sampler, "sampler", None)
        if not any(isinstance(s, DistributedSampler) for s in (sampler, inner)):
            warnings.warn("data-parallel training, but `train_loader` has no DistributedSampler: unless the loader shards by rank "
                          "itself, every rank trains on the same batches and the extra GPUs add nothing", stacklevel=3)

    def _backprop_step(self, loss: Tensor) -> None:
        """core.py:185-212: backward; every ``gradient_acc`` calls clip, (all-reduce,) step and zero the gradients."""
        self._grad_count += 1
        last = self._grad_count == self.gradient_acc
        red = self._grad_reducer()
        if red is not None and not last:
            with red.no_sync():
                loss.backward()
        else:
            loss.backward()
        if last:
            if red is not None:
                red.finalize()
            if isinstance(self.grad_clip, float):
                nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_clip)
            self.optimizer.step()
            self.optimizer.zero_grad()
            self._grad_count = 0

    def _get_loss(self, x: Tensor, target: Tensor, return_logits: bool = False) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        """core.py:214-233"""
        out = self.model(x)
        loss = self.criterion(out if out.dtype == torch.float32 else out.float(), target)
        return (loss, out) if return_logits else loss

    # ---------------------------------------------------------------- optimizer / scheduler plumbing
    def _set_params(self, norm_weight_decay: Optional[float] = None) -> None:
        if not any(p.requires_grad for p in self.model.parameters()):
            raise AssertionError("All parameters are frozen")
        if norm_weight_decay is None:
            self._params = [p for p in self.model.parameters() if p.requires_grad], []
        else:
            self._params = split_normalization_params(self.model)

    def _reset_opt(self, lr: float, norm_weight_decay: Optional[float] = None) -> None:
        """Point the optimizer at the currently trainable parameters with a fresh state (core.py:246-261)."""
        self.optimizer.defaults["lr"] = lr
        self.optimizer.state = defaultdict(dict)
        self.optimizer.param_groups = []
        self._set_params(norm_weight_decay)
        if norm_weight_decay is None:
            self.optimizer.add_param_group({"params": self._params[0]})
        else:
            decays = [norm_weight_decay, self.optimizer.defaults.get("weight_decay", 0)]
            for group, wd in zip(self._params, decays):
                if len(group) > 0:
                    self.optimizer.add_param_group({"params": group, "weight_decay": wd})
        self.optimizer.zero_grad()
        self._restore_opt_state()
        # data parallelism: build the reducer - and with it the rank-0 broadcast of parameters and buffers - HERE, before the first
        # forward of fit_n_epochs / find_lr / check_setup.  Done lazily inside `_backprop_step` it ran between the first batch's
        # forward and its backward: activations saved from the old weights met the new ones in that step's gradient.
        self._grad_reducer()

    def _restore_opt_state(self) -> None:
        """Re-apply a state that ``load`` brought in to the groups ``_reset_opt`` just rebuilt (moments, step counts; the groups'
        own hyper-parameters stay).  One shot; dropped with a warning when the parameter count no longer matches."""
        pending, self._pending_opt_state = self._pending_opt_state, None
        if pending is None:
            return
        groups = self.optimizer.state_dict()["param_groups"]
        if sum(len(g["params"]) for g in groups) != sum(len(g["params"]) for g in pending["param_groups"]):
            warnings.warn("the loaded optimizer state does not match the parameters being optimised now: starting from a fresh "
                          "state", stacklevel=3)
            return
        # state entries are keyed by the position of the parameter in the flattened groups: remap saved order -> current order
        saved_ids = [i for g in pending["param_groups"] for i in g["params"]]
        cur_ids = [i for g in groups for i in g["params"]]
        # ... which is only right when both orders list the same parameters: a different `norm_weight_decay` / freeze setting between
        # save and now keeps the count and permutes the order.  Every per-parameter tensor of a state entry must have the shape of the
        # parameter it is about to be attached to (scalars - step counts - aside); otherwise moments would land on the wrong weights
        # or fail later inside a fused optimizer kernel.
        cur_params = [p for g in self.optimizer.param_groups for p in g["params"]]
        for s, p in zip(saved_ids, cur_params):
            for v in pending["state"].get(s, {}).values():
                if torch.is_tensor(v) and v.dim() > 0 and v.numel() > 1 and tuple(v.shape) != tuple(p.shape):
                    warnings.warn("the loaded optimizer state lists the parameters in another order than the groups being optimised "
                                  "now (different `norm_weight_decay` / freeze settings?): starting from a fresh state", stacklevel=3)
                    return
        state = {c: pending["state"][s] for s, c in zip(saved_ids, cur_ids) if s in pending["state"]}
        self.optimizer.load_state_dict({"state": state, "param_groups": groups})

    @torch.inference_mode()
    def evaluate(self):
        raise NotImplementedError

    @staticmethod
    def _eval_metrics_str(eval_metrics) -> str:
        raise NotImplementedError

    def _reset_scheduler(self, lr: float, num_epochs: int, sched_type: str = "onecycle", **kwargs: Any) -> None:
        total = num_epochs * len(self.train_loader)
        if sched_type == "onecycle":
            self.scheduler = OneCycleLR(self.optimizer, lr, total, **kwargs)
        elif sched_type == "cosine":
            self.scheduler = CosineAnnealingLR(self.optimizer, total, **kwargs)
        else:
            raise ValueError(f"The following scheduler type is not supported: {sched_type}")

    # ---------------------------------------------------------------- public loops
    def fit_n_epochs(self, num_epochs: int, lr: float, freeze_until: Optional[str] = None, sched_type: str = "onecycle",
                     norm_weight_decay: Optional[float] = None, **kwargs: Any) -> None:
        """core.py:277-324"""
        freeze_model(self.model.train(), freeze_until)
        self._reset_opt(lr, norm_weight_decay)
        self._reset_scheduler(lr, num_epochs, sched_type, **kwargs)
        master_bar, _ = _bars()
        mb = master_bar(range(num_epochs))
        for _ in mb:
            self._fit_epoch(mb)
            eval_metrics = self.evaluate()
            mb.main_bar.comment = f"Epoch {self.epoch}/{self.start_epoch + num_epochs}"
            mb.write(f"Epoch {self.epoch}/{self.start_epoch + num_epochs} - {self._eval_metrics_str(eval_metrics)}")
            if eval_metrics["val_loss"] < self.min_loss:
                print(f"Validation loss decreased {self.min_loss:.4} --> {eval_metrics['val_loss']:.4}: saving state...")
                self.min_loss = eval_metrics["val_loss"]
                self.save(self.output_file)
            if self.on_epoch_end is not None:
                self.on_epoch_end(eval_metrics)

    def find_lr(self, freeze_until: Optional[str] = None, start_lr: float = 1e-7, end_lr: float = 1,
                norm_weight_decay: Optional[float] = None, num_it: int = 100) -> None:
        """Learning-rate range test (core.py:326-381): exponential sweep, losses in ``loss_recorder``."""
        if num_it > len(self.train_loader):
            raise ValueError("the value of `num_it` needs to be lower than the number of available batches")
        freeze_model(self.model.train(), freeze_until)
        self._reset_opt(start_lr, norm_weight_decay)
        gamma = (end_lr / start_lr) ** (1 / (num_it - 1))
        scheduler = MultiplicativeLR(self.optimizer, lambda step: gamma)
        self.lr_recorder = [start_lr * gamma ** idx for idx in range(num_it)]
        self.loss_recorder = []
        for batch_idx, (x, target) in enumerate(self.train_loader):
            x, target = self.to_cuda(x, target)
            batch_loss = self._get_loss(x, target)
            self._backprop_step(batch_loss)
            scheduler.step()
            value = float(batch_loss.detach())
            if not math.isfinite(value):
                if batch_idx == 0:
                    raise ValueError("loss value is NaN or inf.")
                break
            self.loss_recorder.append(value)
            if batch_idx + 1 == num_it:
                break
        self.lr_recorder = self.lr_recorder[: len(self.loss_recorder)]

    def plot_recorder(self, beta: float = 0.95, **kwargs: Any) -> None:
        """core.py:383-418 (needs matplotlib)"""
        if len(self.lr_recorder) != len(self.loss_recorder) or len(self.lr_recorder) == 0:
            raise AssertionError("Please run the `lr_find` method first")
        import matplotlib.pyplot as plt
        import numpy as np
        smoothed, avg = [], 0.0
        for idx, loss in enumerate(self.loss_recorder):
            avg = beta * avg + (1 - beta) * loss
            smoothed.append(avg / (1 - beta ** (idx + 1)))
        n = len(self.loss_recorder)
        sl = slice(min(n // 10, 10), -min(n // 20, 5) if n >= 20 else n)
        vals = np.array(smoothed[sl])
        lo = int(vals.argmin())
        hi = vals[: lo + 1].max()
        delta = hi - vals[lo]
        plt.plot(self.lr_recorder[sl], smoothed[sl])
        plt.xscale("log")
        plt.xlabel("Learning Rate")
        plt.ylabel("Training loss")
        plt.ylim(vals[lo] - 0.1 * delta, hi + 0.2 * delta)
        plt.grid(True, linestyle="--", axis="x")
        plt.show(**kwargs)

    def check_setup(self, freeze_until: Optional[str] = None, lr: float = 3e-4, norm_weight_decay: Optional[float] = None,
                    num_it: int = 100, plot: bool = True, **kwargs: Any) -> List[float]:
        """Overfit one batch (core.py:420-451); returns the losses (and plots them when matplotlib is there)."""
        freeze_model(self.model.train(), freeze_until)
        self._reset_opt(lr, norm_weight_decay)
        x, target = next(iter(self.train_loader))
        x, target = self.to_cuda(x, target)
        losses = []
        for _ in range(num_it):
            batch_loss = self._get_loss(x, target)
            self._backprop_step(batch_loss)
            value = float(batch_loss.detach())
            if not math.isfinite(value):
                raise ValueError("loss value is NaN or inf.")
            losses.append(value)
        if plot:
            try:
                import matplotlib.pyplot as plt
                plt.plot(range(len(losses)), losses)
                plt.xlabel("Optimization steps")
                plt.ylabel("Training loss")
                plt.grid(True, linestyle="--", axis="x")
                plt.show(**kwargs)
            except ImportError:
                pass
        return losses
