"""Portable deterministic fill -- independent of torch's RNG.

u01(i, seed): splitmix64 finaliser of (i + seed*0x9E3779B97F4A7C15) mod 2^64, top 24 bits / 2^24
(fp32-exact).  Check value: seed 7 -> [0.1738678217, 0.8773486614, 0.7263535857, 0.1351458430].
"""
import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def u01(n: int, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint64)
        z = i + np.uint64((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def fill(shape, seed: int, lo: float, hi: float) -> torch.Tensor:
    n = int(np.prod(shape))
    u = u01(n, seed).astype(np.float64)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def labels(shape, seed: int, num_classes: int) -> torch.Tensor:
    n = int(np.prod(shape))
    u = u01(n, seed).astype(np.float64)
    lab = np.clip(np.floor(num_classes * u), 0, num_classes - 1).astype(np.int64)
    return torch.from_numpy(lab.reshape(shape))


@torch.no_grad()
def fill_module(module: torch.nn.Module, base: int = 1000) -> None:
    """named_parameters() k=0,1,..: seed=base+k; >=2-D: +-1/sqrt(prod(shape[1:]));
    1-D *.weight: [0.9,1.1]; 1-D *.bias: [-0.1,0.1]."""
    for k, (name, p) in enumerate(module.named_parameters()):
        seed = base + k
        if p.dim() >= 2:
            b = 1.0 / float(np.sqrt(np.prod(p.shape[1:])))
            p.copy_(fill(tuple(p.shape), seed, -b, b))
        elif name.endswith("weight"):
            p.copy_(fill(tuple(p.shape), seed, 0.9, 1.1))
        else:
            p.copy_(fill(tuple(p.shape), seed, -0.1, 0.1))
