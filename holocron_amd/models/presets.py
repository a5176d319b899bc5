"""Dataset presets the training scripts read normalisation constants from (reference: holocron/models/presets.py:
``IMAGENET``, ``IMAGENETTE``, ``CIFAR10`` with ``mean``, ``std``, ``classes``; used at references/classification/train.py:85-95).

The mean / std triples are the standard ImageNet and CIFAR-10 channel statistics.  The reference ships the 1000 ImageNet class
names inline (1 000 lines of data); they are not part of the compute path and are not duplicated: ``classes`` is filled with
``class_<index>`` placeholders of the right length, and ``load_class_names(preset, path)`` replaces them from a text file (one name
per line) when real names are wanted for display."""
from dataclasses import dataclass
from typing import List, Tuple

__all__ = ["CIFAR10", "IMAGENET", "IMAGENETTE", "load_class_names"]


@dataclass
class _Dataset:
    mean: Tuple[float, ...]
    std: Tuple[float, ...]
    classes: List[str]


def _placeholder(n: int) -> List[str]:
    return [f"class_{i}" for i in range(n)]


IMAGENET = _Dataset(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), classes=_placeholder(1000))
IMAGENETTE = _Dataset(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225),
                      classes=["tench", "English springer", "cassette player", "chain saw", "church", "French horn",
                               "garbage truck", "gas pump", "golf ball", "parachute"])
CIFAR10 = _Dataset(mean=(0.5071, 0.4866, 0.4409), std=(0.2673, 0.2564, 0.2761),
                   classes=["airplane", "automobile", "bird", "cat", "deer", "dog", "frog", "horse", "ship", "truck"])


def load_class_names(preset: _Dataset, path: str) -> None:
    with open(path) as fh:
        names = [line.strip() for line in fh if line.strip()]
    if len(names) != len(preset.classes):
        raise ValueError(f"expected {len(preset.classes)} class names, got {len(names)}")
    preset.classes = names
