"""Deterministic synthetic weights for golden fixtures.

Fixtures do not store model weights (that would be megabytes of noise); they store the list of
state_dict (name, shape) pairs dumped from the reference model plus a seed.  Both the fixture
generator (which loads the weights into the *reference* model) and the tests (which load them into
the oracle / the HIP modules) rebuild the same tensors from here.  A CRC of the result is stored in
each fixture to detect RNG drift.
"""
import json
import zlib

import numpy as np


def make_state_dict(names_shapes, seed):
    """names_shapes: list of [name, shape].  Returns {name: float32 ndarray}."""
    sd = {}
    for i, (name, shape) in enumerate(names_shapes):
        shape = tuple(shape)
        rng = np.random.default_rng([seed, i])
        leaf = name.rsplit(".", 1)[-1]
        if name.endswith("positional_encoding.pe"):
            continue  # buffer, recomputed by every implementation
        if "layer_norm" in name and leaf == "weight":
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif leaf == "bias":
            a = 0.1 * rng.standard_normal(shape)
        elif "tgt_word_emb" in name:
            a = 0.5 * rng.standard_normal(shape)
        elif len(shape) >= 2:
            fan_out = shape[0] * int(np.prod(shape[2:])) if len(shape) > 2 else shape[0]
            fan_in = int(np.prod(shape[1:]))
            a = rng.standard_normal(shape) * np.sqrt(2.0 / (fan_in + fan_out)) * 1.5
        else:
            a = 0.1 * rng.standard_normal(shape)
        sd[name] = a.astype(np.float32)
    return sd


def crc_of(sd):
    c = 0
    for k in sorted(sd):
        c = zlib.crc32(np.ascontiguousarray(sd[k]).tobytes(), c)
    return c


def names_shapes_to_json(ns):
    return json.dumps([[n, list(s)] for n, s in ns])


def names_shapes_from_json(s):
    return [(n, tuple(sh)) for n, sh in json.loads(str(s))]
