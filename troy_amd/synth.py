"""Synthetic uniform residues -- the documented generator behind every bit-exact test and bench input.

value(row, n) = splitmix64_stream(seed ^ (row * 0xD1B54A32D192ED03), n) mod p_row, where the n-th output of a
splitmix64 stream seeded with s is mix(s + (n+1) * 0x9E3779B97F4A7C15).  The device twin is
fill_uniform_kernel (troy_amd/csrc/poly.hip); tests regenerate inputs from (seed, shape) instead of storing them.
"""
import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_R = np.uint64(0xD1B54A32D192ED03)


def splitmix64_stream(seed, n):
    """first n outputs of the stream as uint64 array"""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.arange(1, n + 1, dtype=np.uint64)) * _G
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform_rows(seed, row_primes, rows, N, row0=0, inner=1):
    """uint64 [rows][N]; row r reduced mod row_primes[(r // inner) % len(row_primes)]"""
    out = np.empty((rows, N), dtype=np.uint64)
    period = len(row_primes)
    with np.errstate(over="ignore"):
        for r in range(rows):
            p = int(row_primes[(r // inner) % period])
            s = np.uint64(seed) ^ (np.uint64(row0 + r) * _R)
            out[r] = splitmix64_stream(s, N) % np.uint64(p)
    return out


def uniform_ct(seed, primes, size, N, batch=1, row0=0):
    """uint64 [batch][size][limbs][N] of uniform residues (statistically what a ciphertext is)"""
    limbs = len(primes)
    return uniform_rows(seed, primes, batch * size * limbs, N, row0).reshape(batch, size, limbs, N)


def uniform_kswitch_key(seed, key_primes, N):
    """uint64 [K-1][2][K][N]: synthetic key-switching key in NTT form (the key-switch arithmetic is oblivious to key validity)"""
    K = len(key_primes)
    return uniform_rows(seed, key_primes, (K - 1) * 2 * K, N).reshape(K - 1, 2, K, N)
