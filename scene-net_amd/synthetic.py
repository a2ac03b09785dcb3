"""Synthetic workload of BASELINE.json configs C2/C3: seeded LiDAR-like tiles at a UTM-scale
origin and an explicit GENEO bank.  numpy only; shared by bench.py and the tests so both see the same
clouds and the same model."""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np


def synthetic_tile(t: int, n_points: int = 100_000) -> Tuple[np.ndarray, np.ndarray]:
    """Tile `t` of the C2/C3 workload: (xyz [N,3] f64 at UTM-like origin, labels [N] f64)."""
    rng = np.random.default_rng(1000 + t)
    origin = np.array([5.44e5, 4.634e6, 1.5e2])
    n = n_points - 2
    n_g, n_v, n_t = int(0.60 * n), int(0.25 * n), int(0.10 * n)
    n_l = n - n_g - n_v - n_t
    ground = np.stack([rng.uniform(0, 30, n_g), rng.uniform(0, 30, n_g), np.abs(rng.normal(0, 0.3, n_g))], 1)
    veg = np.stack([rng.uniform(0, 30, n_v), rng.uniform(0, 30, n_v), rng.uniform(0, 8, n_v)], 1)
    tower = np.stack([rng.normal(15, 0.6, n_t), rng.normal(15, 0.6, n_t), rng.uniform(0, 40, n_t)], 1)
    a, b = rng.uniform(0, 30, 2), rng.uniform(0, 30, 2)
    s = rng.uniform(0, 1, n_l)
    lines = np.stack([a[0] + s * (b[0] - a[0]), a[1] + s * (b[1] - a[1]), rng.normal(35, 0.5, n_l)], 1)
    sentinels = np.array([[0.0, 0.0, 0.0], [30.0, 30.0, 60.0]])
    xyz = np.concatenate([ground, veg, tower, lines, sentinels], 0)
    xyz[:, :2] = np.clip(xyz[:, :2], 0, 30)
    xyz[:, 2] = np.clip(xyz[:, 2], 0, 60)
    labels = np.concatenate([np.full(n_g, 2.0), np.full(n_v, 4.0), np.full(n_t, 15.0), np.full(n_l, 16.0),
                             np.full(2, 1.0)])
    perm = rng.permutation(n_points)
    return (xyz[perm] + origin), labels[perm]


def synthetic_bank_spec(geneo_num: Dict[str, int] = None, seed: int = 7):
    """Explicit model parameters from default_rng(seed): ([(kind, params)], names, lambdas, last)."""
    geneo_num = {"cy": 6, "cone": 5, "neg": 5} if geneo_num is None else geneo_num
    rng = np.random.default_rng(seed)
    specs: List[Tuple[str, Dict[str, float]]] = []
    names: List[str] = []
    for kind, n in geneo_num.items():
        for i in range(n):
            p = dict(radius=float(rng.uniform(0.5, 4)), sigma=float(rng.uniform(0.5, 2)))
            if kind == "cone":
                p.update(apex=float(rng.integers(4, 8)), cone_radius=float(rng.uniform(0.5, 4)),
                         cone_inc=float(rng.uniform(0.05, 0.45)))
            if kind == "neg":
                p.update(neg_factor=float(rng.uniform(0.1, 0.9)))
            specs.append((kind, p))
            names.append(f"{kind}_{i}")
    G = len(specs)
    lambdas = rng.uniform(-2 / G, 1 / G, G).astype(np.float32)
    return specs, names, lambdas, G - 1


def apply_bank_spec(model, specs, names, lambdas, last) -> None:
    """Writes an explicit spec into a SceneNet-shaped module (no dependence on torch RNG streams)."""
    import torch
    with torch.no_grad():
        for (kind, p), n in zip(specs, names):
            for k, v in p.items():
                model.geneos[n].geneo_params[k].fill_(v)
        for n, v in zip(names, lambdas):
            model.lambdas_dict[f"lambda_{n}"].fill_(float(v))
    model.last_lambda = f"lambda_{names[last]}"
