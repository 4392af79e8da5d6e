"""Helpers shared by the tests: golden decoding, limb packing, edge/random inputs."""
import random

import numpy as np

from oracle import pyoracle as po
from oracle.coracle import elem_bytes


def unhex(lst):
    return [int(v, 16) for v in lst]


def field_of(case):
    return po.Field(int(case['modulus'], 16), binary=case['binary'])


def pack(vals, eb):
    """ints -> raw little-endian numpy array ((n,), (n,2) uint64 for 16 bytes, (n,3) uint32 for 12 bytes)."""
    if eb == 24:
        buf = b''.join(int(v).to_bytes(24, 'little') for v in vals)
        return np.frombuffer(buf, dtype=np.uint64).reshape(len(vals), 3).copy()
    if eb == 16:
        buf = b''.join(int(v).to_bytes(16, 'little') for v in vals)
        return np.frombuffer(buf, dtype=np.uint64).reshape(len(vals), 2).copy()
    if eb == 12:
        buf = b''.join(int(v).to_bytes(12, 'little') for v in vals)
        return np.frombuffer(buf, dtype=np.uint32).reshape(len(vals), 3).copy()
    dt = {1: np.uint8, 4: np.uint32, 8: np.uint64}[eb]
    return np.array([int(v) for v in vals], dtype=object).astype(np.uint64).astype(dt) if len(vals) else np.zeros(0, dt)


def lshape(eb, *dims):
    """numpy shape of a raw limb array with the given leading dims"""
    return tuple(dims) + ({16: (2,), 12: (3,), 24: (3,)}.get(eb, ()))


def unpack(arr, eb):
    if eb == 24:
        a = np.ascontiguousarray(arr).reshape(-1, 3)
        return [int(a[i, 0]) | (int(a[i, 1]) << 64) | (int(a[i, 2]) << 128) for i in range(a.shape[0])]
    if eb == 16:
        a = np.ascontiguousarray(arr).reshape(-1, 2)
        return [int(a[i, 0]) | (int(a[i, 1]) << 64) for i in range(a.shape[0])]
    if eb == 12:
        a = np.ascontiguousarray(arr).view(np.uint32).reshape(-1, 3)
        return [int(a[i, 0]) | (int(a[i, 1]) << 32) | (int(a[i, 2]) << 64) for i in range(a.shape[0])]
    return [int(v) for v in np.asarray(arr).reshape(-1)]


def edge_values(F):
    q = F.order
    bits = max((q - 1).bit_length(), 1)
    e = [0, 1, 2, q - 1, q - 2, (q - 1) // 2, (q + 1) // 2, 2**32 - 1, 2**32, 2**63, 2**64 - 1, 2**bits - 1,
         2**(bits - 1), 2**31, 2**33 - 1, 2**64, 2**64 + 1, 2**96, 2**127]
    return sorted(set(v % q for v in e))


def rand_values(F, n, seed):
    r = random.Random(seed)
    return [r.randrange(F.order) for _ in range(n)]


def cross(vals):
    a, b = [], []
    for x in vals:
        for y in vals:
            a.append(x)
            b.append(y)
    return a, b


# fields exercised everywhere (name -> (modulus, binary)); superset of the golden file
P61, P64, P128 = 2**61 - 1, 2**64 - 189, 2**128 - 173
EXTRA_FIELDS = {
    'P61': (P61, False), 'P64': (P64, False), 'P128': (P128, False), 'P127': (2**127 - 1, False),
    'P96': (2**96 - 17, False), 'GF2_8': (0x11b, True), 'GF2_128': ((1 << 128) | 0x87, True),
}
