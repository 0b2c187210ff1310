"""`Layout` — the result type of path_linear_sgd_layout (reference: src/layout.rs:17-163).

coords[node*2*D + end*D + dim], end 0 = '+', end 1 = '-'.  The device keeps coordinates in
exactly this order, so no re-interleave (layout.rs:39-69 `from_vectors`) is needed on the way
out; `from_vectors` is kept for callers holding dim-major vectors.
"""
import io
from typing import List

import numpy as np

_DIM_NAMES = ["x", "y", "z", "w"]


def dim_name(dim: int) -> str:
    """layout.rs:248-256 (dims >= 4 are all called 'd' there)."""
    return _DIM_NAMES[dim] if dim < 4 else "d"


def _fmt_f64(v: float) -> str:
    """Rust `{}` (Display) on f64, as used by write_tsv (layout.rs:154,158): the shortest
    decimal that round-trips, never scientific notation, and no fraction when it is zero
    (1.0 prints as "1", 1e21 as "1000000000000000000000")."""
    if v != v:
        return "NaN"
    if v in (float("inf"), float("-inf")):
        return "inf" if v > 0 else "-inf"
    r = repr(float(v))
    if "e" in r or "E" in r:
        # expand scientific notation exactly
        from decimal import Decimal
        r = format(Decimal(r), "f")
    if r.endswith(".0"):
        r = r[:-2]
    return r


class Layout:
    def __init__(self, dimensions: int, num_nodes: int, coords=None):
        self.dimensions = int(dimensions)
        self.num_nodes = int(num_nodes)
        n = self.num_nodes * 2 * self.dimensions
        if coords is None:
            self.coords = np.zeros(n, dtype=np.float64)              # layout.rs:28-35
        else:
            self.coords = np.ascontiguousarray(coords, dtype=np.float64)
            assert self.coords.shape[0] == n

    @staticmethod
    def from_vectors(coord_vecs: List[np.ndarray]) -> "Layout":
        """layout.rs:39-69: one vector per dimension, each 2*num_nodes long."""
        assert len(coord_vecs) > 0, "Must have at least 1 dimension"
        entries = len(coord_vecs[0])
        assert entries % 2 == 0, "Must have even number of entries (2 per node)"
        for v in coord_vecs:
            assert len(v) == entries, "All dimension vectors must have same length"
        stacked = np.stack([np.asarray(v, dtype=np.float64) for v in coord_vecs], axis=1)   # [2N, D]
        return Layout(len(coord_vecs), entries // 2, stacked.reshape(-1))

    def index(self, node, end, dim):
        return node * 2 * self.dimensions + end * self.dimensions + dim

    def get(self, node, end, dim):
        return float(self.coords[self.index(node, end, dim)])

    def set(self, node, end, dim, value):
        self.coords[self.index(node, end, dim)] = value

    def get_coords(self, node, end):
        s = self.index(node, end, 0)
        return self.coords[s:s + self.dimensions]

    def x_plus(self, node):
        return self.get(node, 0, 0)

    def y_plus(self, node):
        return self.get(node, 0, 1)

    def x_minus(self, node):
        return self.get(node, 1, 0)

    def y_minus(self, node):
        return self.get(node, 1, 1)

    def distance(self, node_a, end_a, node_b, end_b):
        """layout.rs:126-133 (sequential sum over dims, then sqrt)."""
        s = 0.0
        for d in range(self.dimensions):
            delta = self.get(node_a, end_a, d) - self.get(node_b, end_b, d)
            s += delta * delta
        return float(np.sqrt(s))

    def write_tsv(self, writer):
        """layout.rs:138-163."""
        D = self.dimensions
        hdr = ["idx"] + [f"{dim_name(d)}+" for d in range(D)] + [f"{dim_name(d)}-" for d in range(D)]
        writer.write("\t".join(hdr) + "\n")
        c = self.coords.reshape(self.num_nodes, 2 * D)
        for node in range(self.num_nodes):
            writer.write(str(node) + "".join("\t" + _fmt_f64(v) for v in c[node]) + "\n")

    def to_tsv(self) -> str:
        buf = io.StringIO()
        self.write_tsv(buf)
        return buf.getvalue()

    @staticmethod
    def read_tsv(reader) -> "Layout":
        """layout.rs:166-217."""
        lines = reader.read().split("\n")
        if not lines or lines[0] == "":
            raise ValueError("Empty file")
        cols = lines[0].split("\t")
        if len(cols) < 3 or (len(cols) - 1) % 2 != 0:
            raise ValueError("Invalid header format")
        D = (len(cols) - 1) // 2
        rows = []
        for ln in lines[1:]:
            if not ln.strip():
                continue
            parts = ln.split("\t")
            if len(parts) != len(cols):
                raise ValueError(f"Row has {len(parts)} columns, expected {len(cols)}")
            rows.append([float(v) for v in parts[1:]])
        lay = Layout(D, len(rows))
        if rows:
            lay.coords[:] = np.array(rows, dtype=np.float64).reshape(-1)
        return lay

    def calculate_stress(self, target_distances):
        """layout.rs:224-244."""
        wsum, wtot = 0.0, 0.0
        for (na, ea, nb, eb, td) in target_distances:
            if td == 0.0:
                continue
            w = 1.0 / (td * td)
            err = self.distance(na, ea, nb, eb) - td
            wsum += err * err * w
            wtot += w
        return float(np.sqrt(wsum / wtot)) if wtot > 0 else 0.0
