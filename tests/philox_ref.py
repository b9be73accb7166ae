"""Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) restated in numpy
from the paper's definition, and its known-answer vectors (Random123 1.x, examples/kat_vectors: the three
`philox4x32 10` lines).  Tests only.  The replay sampler (csrc/smx_replay.hip) draws row i of a uniform sample as
mulhi64((out[0] << 32) | out[1], len) with counter = (offset + i, 0) as two 32-bit words + two zero words and
key = seed as two 32-bit words."""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF

# (counter[4], key[2]) -> output[4]
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(ctr, key):
    c = [int(x) & MASK for x in ctr]
    k0, k1 = int(key[0]) & MASK, int(key[1]) & MASK
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c[3] ^ k1) & MASK, p0 & MASK]
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return tuple(c)


def uniform_index(i, length, seed, offset):
    """row i of smx_uniform_indices(len, seed, offset)"""
    ctr = (offset + i) & 0xFFFFFFFFFFFFFFFF
    o = philox4x32_10((ctr & MASK, ctr >> 32, 0, 0), (seed & MASK, (seed >> 32) & MASK))
    return (((o[0] << 32) | o[1]) * int(length)) >> 64


def uniform_indices(n, length, seed, offset):
    return np.array([uniform_index(i, length, seed, offset) for i in range(n)], dtype=np.int64)
