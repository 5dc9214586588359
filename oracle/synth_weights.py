"""Deterministic, name-keyed synthetic weights (test infrastructure).

The same function fills the reference model (in oracle/gen_golden.py), the oracle and the
HIP model in the tests, so golden fixtures need not store any weights: each tensor is
drawn from ``RandomState(crc32(name) ^ seed)`` and depends only on its name and shape.
Biases and norm affine parameters are non-trivial on purpose (catches dropped-bias bugs).
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import numpy as np

PAD = 1


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> np.ndarray:
    rs = np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    shape = tuple(int(s) for s in shape)
    if name.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    if name.endswith("running_mean"):
        return np.zeros(shape, dtype=np.float32)
    if name.endswith("running_var"):
        return np.ones(shape, dtype=np.float32)
    if name.endswith("_float_tensor"):
        return np.zeros(shape, dtype=np.float32)
    if name.endswith("version"):
        return np.full(shape, 3.0, dtype=np.float32)
    if name.endswith("pos_emb_alpha"):
        return np.full(shape, 1.25, dtype=np.float32)
    if name.endswith("mask_emb"):
        return rs.uniform(0, 1, size=shape).astype(np.float32)
    if len(shape) == 1:
        if name.endswith(".weight") or name.endswith("weight_g"):
            return (1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32)  # norm gain
        return (0.05 * rs.standard_normal(shape)).astype(np.float32)  # bias
    if name.endswith("embed_tokens.weight"):
        w = (rs.standard_normal(shape) * shape[1] ** -0.5).astype(np.float32)
        w[PAD] = 0
        return w
    if name.endswith("weight_g"):
        return (1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    return (rs.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0) -> Dict[str, np.ndarray]:
    return {k: synth_tensor(k, v, seed) for k, v in shapes.items()}


def load_synth(module, seed: int = 0, skip_prefix: str = None):
    """Fill a torch module (reference, oracle or HIP model) in place.  Entries under ``skip_prefix`` keep their
    current values (the frozen HuBERT inside the reference encoder is keyed by its own local names)."""
    import torch

    sd = module.state_dict()
    new = {k: (v if skip_prefix and k.startswith(skip_prefix) else
               torch.from_numpy(synth_tensor(k, tuple(v.shape), seed)).to(v.dtype))
           for k, v in sd.items()}
    module.load_state_dict(new, strict=True)
    return module
