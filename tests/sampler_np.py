"""numpy mirror of reference-seal-backend_amd/csrc/client/sampler.h (test helper): the counter-based RLWE samplers
the host client and the device encryption kernels share.  Lets a test hand the oracle exactly the randomness they used."""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def sample_word(seed, stream, count):
    with np.errstate(over="ignore"):
        base = splitmix64(np.uint64(seed) ^ splitmix64(np.uint64(stream)))
        return splitmix64(base + np.arange(count, dtype=np.uint64))


def sample_ternary(seed, stream, count):
    w = sample_word(seed, stream, count)
    out = np.zeros(count, dtype=np.int32)
    done = np.zeros(count, dtype=bool)
    for _ in range(32):
        f = (w & np.uint64(3)).astype(np.int32)
        take = (~done) & (f != 3)
        out[take] = f[take] - 1
        done |= take
        w = w >> np.uint64(2)
    return out


def _popcount21(v):
    v = v & np.uint64(0x1FFFFF)
    return np.array([bin(int(x)).count("1") for x in v], dtype=np.int32) if v.size < 64 else _popcount_vec(v)


def _popcount_vec(v):
    v = v.astype(np.uint64)
    c = np.zeros(v.shape, dtype=np.int32)
    for k in range(21):
        c += ((v >> np.uint64(k)) & np.uint64(1)).astype(np.int32)
    return c


def sample_cbd(seed, stream, count):
    w = sample_word(seed, stream, count)
    return _popcount_vec(w & np.uint64(0x1FFFFF)) - _popcount_vec((w >> np.uint64(21)) & np.uint64(0x1FFFFF))


def enc_streams(index):
    """streams of the asymmetric encryption with ciphertext index `index`: (u, e0, e1)"""
    return 3 * index, 3 * index + 1, 3 * index + 2
