"""``parallel`` and ``find_image_size`` (reference: holocron/utils/misc.py): dataset-side helpers the training scripts import
(`references/classification/train.py:30-36`).  Pure host code; the plotting half needs matplotlib and is skipped without it."""
import multiprocessing as mp
from math import sqrt
from multiprocessing.pool import ThreadPool
from typing import Any, Callable, Iterable, Optional, Sequence, Tuple, TypeVar

Inp = TypeVar("Inp")
Out = TypeVar("Out")

__all__ = ["find_image_size", "parallel"]


def parallel(func: Callable[[Inp], Out], arr: Sequence[Inp], num_threads: Optional[int] = None, progress: bool = False,
             **kwargs: Any) -> Iterable[Out]:
    """Map ``func`` over ``arr`` on a thread pool (misc.py:23-52); ``num_threads < 2`` runs serially."""
    n = num_threads if isinstance(num_threads, int) else min(16, mp.cpu_count())
    wrap = (lambda it: it)
    if progress:
        try:
            from tqdm.auto import tqdm
            wrap = (lambda it: tqdm(it, total=len(arr), **kwargs))
        except ImportError:
            pass
    if n < 2:
        return list(map(func, wrap(arr)))
    with ThreadPool(n) as tp:
        return list(wrap(tp.imap(func, arr)))


def find_image_size(dataset: Sequence[Tuple[Any, Any]], plot: bool = True, **kwargs: Any) -> Tuple[int, int]:
    """Median-aspect-ratio / median-side target size of a dataset of (PIL image, target) pairs (misc.py:55-90).  Returns
    (height, width) - the reference only shows them in the figure title."""
    import numpy as np
    sizes = np.asarray(parallel(lambda s: s[0].size, dataset, progress=plot))[:, ::-1]      # PIL size is (w, h)
    ratios = sizes[:, 0] / sizes[:, 1]
    sides = np.sqrt(sizes[:, 0] * sizes[:, 1])
    ratio, side = float(np.median(ratios)), float(np.median(sides))
    height, width = round(side * sqrt(ratio)), round(side / sqrt(ratio))
    if plot:
        try:
            import matplotlib.pyplot as plt
            fig, axes = plt.subplots(1, 2)
            for ax, data, med, title in ((axes[0], ratios, ratio, f"Aspect ratio (median: {ratio:.2})"),
                                         (axes[1], sides, side, f"Side (median: {int(side)})")):
                ax.hist(data, bins=30, alpha=0.7)
                ax.title.set_text(title)
                ax.grid(True, linestyle="--", axis="x")
                ax.axvline(med, color="r")
            fig.suptitle(f"Median image size: ({height}, {width})")
            plt.show(**kwargs)
        except ImportError:
            pass
    return height, width
