"""Round 4, review item 6 (box-regime SchNet): which share of the pairs of the bulk-water box has BOTH atoms inside one block of B
consecutive centre atoms -- the pairs whose two directions could be accumulated in LDS instead of through float atomics -- for the
generator's order and for cell / Morton orders (CPU only: the neighbour list of the synthetic box, no kernels involved)."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from schnetpack_amd import synthetic as S   # noqa: E402


def morton(ix, iy, iz, bits=8):
    code = np.zeros_like(ix, dtype=np.int64)
    for b in range(bits):
        code |= ((ix >> b) & 1) << (3 * b) | ((iy >> b) & 1) << (3 * b + 1) | ((iz >> b) & 1) << (3 * b + 2)
    return code


def main(n_side=22):
    b = S.water_box(n_side)
    R = b["R"].numpy()
    ii, jj = b["idx_i"].numpy(), b["idx_j"].numpy()
    N, E = R.shape[0], ii.shape[0]
    box = float(b["cell"][0, 0]) if "cell" in b else float(R.max() - R.min())
    out = {"n_atoms": int(N), "n_pairs_directed": int(E), "rows": []}
    orders = {"generator (lattice) order": np.arange(N)}
    for cell in (5.0, 2.5):
        c = np.floor((R - R.min(0)) / cell).astype(np.int64)
        nc = c.max(0) + 1
        lin = (c[:, 0] * nc[1] + c[:, 1]) * nc[2] + c[:, 2]
        orders["cells of %.1f A, sorted" % cell] = np.argsort(lin, kind="stable")
        orders["Morton order of %.1f A cells" % cell] = np.argsort(morton(c[:, 0], c[:, 1], c[:, 2]), kind="stable")
    for name, perm in orders.items():
        rank = np.empty(N, dtype=np.int64)
        rank[perm] = np.arange(N)
        ri, rj = rank[ii], rank[jj]
        row = {"order": name}
        for B in (16, 32, 64, 128, 256):
            same = (ri // B) == (rj // B)
            row["B=%d" % B] = round(float(same.mean()), 4)
        out["rows"].append(row)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 22)
