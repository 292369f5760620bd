"""The rank-quantised path rests on two claims that need no GPU to check (numpy restatements of what
csrc/ddt_image.cpp builds and csrc/ddt_prepass.hip searches):

1. exactness: with the sorted distinct thresholds t_0 < t_1 < ... of one feature (signed-int32 order of the key),
   r(x) = #{k : t_k <= x} satisfies  (x < t_k)  <=>  (r(x) < k + 1)  for every x -- a node only needs the rank.
2. the two-level search: slicing [t_0, t_last] into NB equal pieces of 2^shift codes, starts[b] = #{keys in slices < b}
   and P = a power of two above the fullest slice, log2(P) branch-free probes from starts[bucket(x)] find r(x); keys past
   the slice are > x by construction, so no end test is needed (rank_kernel / fused_rank_kernel)."""
import numpy as np
import pytest


def build(keys, NB):
    k = np.unique(keys.astype(np.int64))                       # sorted distinct, as int32 values
    lo, hi = int(k[0]), int(k[-1])
    span, shift = hi - lo, 0
    while (span >> shift) >= NB:
        shift += 1
    cnt = np.bincount(((k - lo) >> shift).astype(np.int64), minlength=NB)
    starts = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    P = 1
    while P <= cnt.max():
        P <<= 1
    return k, lo, hi, shift, starts, P


def search(x, k, lo, hi, shift, starts, P, NB):
    table = np.concatenate([k, np.full(P, np.iinfo(np.int32).max, np.int64)])   # INT_MAX pads behind the keys
    b = np.clip((x - lo) >> shift, 0, NB - 1)
    b = np.where(x < lo, 0, b)
    pos = starts[b].astype(np.int64)
    step = P >> 1
    while step >= 1:
        probe = np.minimum(pos + step - 1, len(table) - 1)
        pos = np.where(table[probe] <= x, pos + step, pos)
        step >>= 1
    return np.minimum(np.where(x > hi, len(k), pos), len(k))


@pytest.mark.parametrize("NB", [256, 4096])
@pytest.mark.parametrize("shape", ["uniform_float_bits", "clustered", "full_int_range", "two_keys", "one_key"])
def test_rank_is_exact_and_two_level_search_finds_it(shape, NB):
    rng = np.random.default_rng(hash((shape, NB)) % (1 << 32))
    if shape == "uniform_float_bits":
        keys = rng.random(8000).astype(np.float32).view(np.int32)
    elif shape == "clustered":
        keys = np.concatenate([(np.float32(0.25) + rng.integers(0, 3000, 4000).astype(np.float32) * np.float32(2 ** -22)).view(np.int32),
                               np.array([np.float32(-3e38), np.float32(3e38)]).view(np.int32)])
    elif shape == "full_int_range":
        keys = rng.integers(-2 ** 31, 2 ** 31, 5000).astype(np.int32)
    elif shape == "two_keys":
        keys = np.array([-5, 7], np.int32)
    else:
        keys = np.array([123456], np.int32)
    k, lo, hi, shift, starts, P = build(keys, NB)
    x = np.concatenate([k, k - 1, k + 1, rng.integers(-2 ** 31, 2 ** 31, 20000),
                        np.array([-2 ** 31, 2 ** 31 - 1, 0, -1, 1])]).astype(np.int64)
    x = np.clip(x, -2 ** 31, 2 ** 31 - 1)
    r = np.searchsorted(k, x, side="right")                     # r(x) = #{t_k <= x}
    # claim 1: every node test is decided by the rank alone
    for j in rng.integers(0, len(k), 50):
        assert np.array_equal(x < k[j], r < j + 1)
    # claim 2: the bucket + short binary search computes exactly r(x)
    assert np.array_equal(search(x, k, lo, hi, shift, starts, P, NB), r)
    assert P <= 2 * max(1, np.bincount(((k - lo) >> shift).astype(np.int64)).max())
