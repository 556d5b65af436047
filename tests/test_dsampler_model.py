"""The device sampler's ALGORITHM (tools/proto_dsampler.py: candidate compaction, tile cuts at a user's end and at repeats,
shuffle targets with prefix-bound refinement, the permutation by sort + pointer chase) against the streams the reference
produced (tests/golden/sampler_*.npz) -- on CPU, at several tile sizes, so that the kernels' logic is checked before a GPU
sees it.  The kernels themselves are tested in tests/test_gpu_dsampler.py."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import proto_dsampler as P  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _state(seed):
    rs = np.random.RandomState(seed)
    st = rs.get_state()
    return st[1], st[2]


def _next_randint30(key, pos, used):
    """np.random.randint(2**30) after `used` words of the stream"""
    raw = P.raw_words(key, pos, used + 64)
    k = used
    while True:  # mask 2**30 - 1, max 2**30 - 1: the first word is taken
        v = int(raw[k]) & (2 ** 30 - 1)
        return v


@pytest.mark.parametrize("tile", [5, 64, 1024])
@pytest.mark.parametrize("name", ["toy", "syn200", "dense", "pow2"])
def test_tile_algorithm_reproduces_reference_streams(name, tile):
    g = np.load(os.path.join(GOLDEN, "sampler_%s.npz" % name))
    key, pos = _state(int(g["seed"]))
    args = (g["keys"], g["ptr"], g["items"], int(g["item_nums"]), int(g["neg_ratio"]))
    (u, i, j), used = P.sample("pairwise", key, pos, *args, TILE=tile)
    assert np.array_equal(u, g["pw_u"]) and np.array_equal(i, g["pw_i"]) and np.array_equal(j, g["pw_j"])
    assert _next_randint30(key, pos, used) == int(g["pw_next"])
    (u, i, y), used = P.sample("pointwise", key, pos, *args, TILE=tile)
    assert np.array_equal(u, g["pt_u"]) and np.array_equal(i, g["pt_i"]) and np.array_equal(y, g["pt_y"])
    assert _next_randint30(key, pos, used) == int(g["pt_next"])
    (u, i, ng), used = P.sample("cml", key, pos, *args, TILE=tile)
    assert np.array_equal(u, g["cml_u"]) and np.array_equal(i, g["cml_i"]) and np.array_equal(ng, g["cml_neg"])
    assert _next_randint30(key, pos, used) == int(g["cml_next"])
