"""Parameter structs of the SGD path, field for field as in the reference.

Reference: `PathSGDParams` src/sgd.rs:196-234, `LayoutSGDParams` src/sgd.rs:676-763,
`YgsParams` src/ygs.rs:16-93.
"""
from dataclasses import dataclass, field, replace

import numpy as np

from .graph import FlatGraph


@dataclass
class PathSGDParams:
    # defaults: sgd.rs:214-234
    iter_max: int = 100
    iter_with_max_learning_rate: int = 0
    min_term_updates: int = 100
    delta: float = 0.0
    eps: float = 0.01
    eta_max: float = 100.0
    theta: float = 0.99
    space: int = 100
    space_max: int = 100
    space_quantization_step: int = 100
    cooling_start: float = 0.5
    nthreads: int = 1
    progress: bool = False
    seed: int = 9399220


@dataclass
class LayoutSGDParams:
    # defaults: sgd.rs:709-729
    dimensions: int = 2
    iter_max: int = 30
    iter_with_max_learning_rate: int = 0
    min_term_updates: int = 100
    delta: float = 0.0
    eps: float = 0.01
    eta_max: float = 100.0
    theta: float = 0.99
    space: int = 100
    space_max: int = 1000
    space_quantization_step: int = 100
    cooling_start: float = 0.5
    nthreads: int = 1
    progress: bool = False
    seed: int = 9399220

    @staticmethod
    def from_graph(g: FlatGraph, dimensions: int, nthreads: int) -> "LayoutSGDParams":
        """sgd.rs:733-762"""
        counts = g.path_step_counts()
        s = int(counts.sum()) if counts.size else 0
        m = int(counts.max()) if counts.size else 0
        return LayoutSGDParams(dimensions=dimensions, iter_max=30, min_term_updates=10 * s,
                               eta_max=float(m * m), space=m, space_max=1000,
                               space_quantization_step=100, nthreads=nthreads)


def _ygs_default_sgd() -> PathSGDParams:
    # ygs.rs:23-45 (placeholders 0 are filled by from_graph)
    return PathSGDParams(iter_max=100, iter_with_max_learning_rate=0, min_term_updates=0, delta=0.0,
                         eps=0.01, eta_max=0.0, theta=0.99, space=0, space_max=100,
                         space_quantization_step=100, cooling_start=0.5, nthreads=1,
                         progress=False, seed=9399220)


@dataclass
class YgsParams:
    path_sgd: PathSGDParams = field(default_factory=_ygs_default_sgd)
    verbose: int = 0

    @staticmethod
    def from_graph(g: FlatGraph, verbose: int, nthreads: int) -> "YgsParams":
        """ygs.rs:50-93: min_term_updates = sum of path step counts, eta_max = (max step
        count)^2, space = longest path in bp."""
        p = YgsParams()
        p.verbose = verbose
        sgd = replace(p.path_sgd, nthreads=nthreads, progress=verbose >= 2)
        counts = g.path_step_counts()
        _, plen = g.step_positions()
        s = int(counts.sum()) if counts.size else 0
        m = int(counts.max()) if counts.size else 0
        sgd.min_term_updates = s
        sgd.eta_max = float(m * m)
        sgd.space = int(plen.max()) if plen.size else 0
        p.path_sgd = sgd
        return p


def first_cooling_iteration(p) -> int:
    """sgd.rs:297 / :857"""
    return int(np.floor(p.cooling_start * float(p.iter_max)))
