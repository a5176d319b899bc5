"""Parameter-freezing helpers of the trainers (reference: holocron/trainer/utils.py:14-98; same behaviour)."""
from typing import List, Optional, Sequence, Tuple

from torch import nn
from torch.nn.modules.batchnorm import _BatchNorm

__all__ = ["freeze_bn", "freeze_model", "split_normalization_params"]


def freeze_bn(mod: nn.Module) -> None:
    """Affine BatchNorm layers whose parameters are ALL frozen stop updating their running statistics and normalise with them
    (utils.py:14-30): ``track_running_stats = False`` + ``eval()``.  On the fused HIP units this selects the running-statistics
    forward; their backward in that mode is the frozen-statistics one (nn/convbn_op.py)."""
    for m in mod.modules():
        if not isinstance(m, _BatchNorm) or not m.affine:
            continue
        if any(p.requires_grad for p in m.parameters()):
            continue
        m.track_running_stats = False
        m.eval()


def freeze_model(model: nn.Module, last_frozen_layer: Optional[str] = None, frozen_bn_stat_update: bool = False) -> None:
    """Un-freeze everything, then freeze every parameter registered up to and including the layer whose name starts with
    ``last_frozen_layer`` (registration order = forward order, utils.py:33-70).  Unknown layer -> ValueError."""
    for p in model.parameters():
        p.requires_grad_(True)
    if isinstance(last_frozen_layer, str):
        seen = False
        for name, p in model.named_parameters():
            inside = name.startswith(last_frozen_layer)
            if seen and not inside:
                break                     # first parameter after the layer: done
            p.requires_grad_(False)
            seen = seen or inside
        if not seen:
            raise ValueError(f"Unable to locate child module {last_frozen_layer}")
    if not frozen_bn_stat_update:
        freeze_bn(model)


def split_normalization_params(model: nn.Module, norm_classes: Optional[Sequence[type]] = None
                               ) -> Tuple[List[nn.Parameter], List[nn.Parameter]]:
    """(normalisation parameters, all other parameters), trainable ones only (utils.py:73-98): used to give the norm layers
    their own weight decay."""
    classes = tuple(norm_classes) if norm_classes else (_BatchNorm, nn.LayerNorm, nn.GroupNorm)
    for t in classes:
        if not (isinstance(t, type) and issubclass(t, nn.Module)):
            raise ValueError(f"Class {t} is not a subclass of nn.Module.")
    norm: List[nn.Parameter] = []
    other: List[nn.Parameter] = []
    for module in model.modules():
        has_children = next(module.children(), None) is not None
        if has_children:                  # a container's own parameters; its children are visited on their own
            other.extend(p for p in module.parameters(recurse=False) if p.requires_grad)
        elif isinstance(module, classes):
            norm.extend(p for p in module.parameters() if p.requires_grad)
        else:
            other.extend(p for p in module.parameters() if p.requires_grad)
    return norm, other
