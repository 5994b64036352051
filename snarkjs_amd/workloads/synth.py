"""Synthetic input streams shared by the golden generator (oracle/gen_golden.js) and the tests.

word(seed, k) = fmix32(seed + k*0x9E3779B9)  (murmur3 finaliser), little-endian u32 stream;
element i = words 8i..8i+7 with the top byte masked (0x1f -> uniform 253-bit value, < r on both curves).
Definitions mirror SURVEY.md §8d (uniform scalars; 60/30/10 witness-like mix).
"""
import numpy as np


def _fmix32(h):
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h = (h * np.uint32(0x85EBCA6B)).astype(np.uint32)
    h ^= h >> np.uint32(13)
    h = (h * np.uint32(0xC2B2AE35)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return h


def words(seed, count, start=0):
    k = np.arange(start, start + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (np.uint64(seed) + k * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    return _fmix32(x.astype(np.uint32))


def elems(seed, n, mask_top=0x1F):
    """n x 32 bytes as a uint8 array of shape (n*32,)"""
    w = words(seed, n * 8).astype("<u4")
    b = w.view(np.uint8).copy()
    if n:
        b[31::32] &= np.uint8(mask_top)
    return b


def iota(n):
    """element i = integer (i+1), little-endian, written as-is"""
    b = np.zeros((n, 32), dtype=np.uint8)
    v = np.arange(1, n + 1, dtype=np.uint32)
    b[:, 0] = v & 255
    b[:, 1] = (v >> 8) & 255
    b[:, 2] = (v >> 16) & 255
    return b.reshape(-1)


def witness_like(seed, n):
    full = elems(seed, n).reshape(n, 32)
    k = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (np.uint64(seed ^ 0xABCDEF) + k * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    sel = _fmix32(x.astype(np.uint32)) % np.uint32(100)
    small = sel < 60
    mid = (sel >= 60) & (sel < 90)
    bit = full[:, 0] & 1
    full[small, :] = 0
    full[small, 0] = bit[small]
    full[mid, 8:] = 0
    return full.reshape(-1)


def to_int(b):
    return int.from_bytes(bytes(b), "little")
