"""MT19937 jump-ahead (csrc/mt_jump.cpp): the polynomial jump that gives the device sampler's chains their start blocks,
applied on the host (crec_debug_mt_jump) and compared with numpy's own generator run the long way."""
import numpy as np

from cleverrec_amd import _lib


def _state(seed):
    st = np.random.RandomState(seed).get_state()
    return st[1], st[2]


def test_jump_ahead_on_the_host_matches_plain_generation():
    """the polynomial jump the device chains use (crec_debug_mt_jump), against numpy's own generator"""
    lib = _lib.load()
    key, _ = _state(99)
    bg = np.random.MT19937()
    st = bg.state
    st["state"]["key"], st["state"]["pos"] = key, 624
    bg.state = st
    raw = bg.random_raw(128 * 128 * 624)  # tempered words of blocks 1 .. 16384

    def temper(y):
        y = y.astype(np.uint64)
        y ^= y >> 11
        y ^= (y << 7) & 0x9d2c5680
        y ^= (y << 15) & 0xefc60000
        y ^= y >> 18
        return (y & 0xffffffff).astype(np.uint32)
    for c in (1, 2, 77, 128):
        out = np.empty(624, np.uint32)
        assert lib.crec_debug_mt_jump(_lib._np(np.ascontiguousarray(key)), c, _lib._np(out)) == 0
        b = c * 128
        assert np.array_equal(temper(out), raw[(b - 1) * 624:b * 624].astype(np.uint32))
