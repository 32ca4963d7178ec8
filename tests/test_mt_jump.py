"""The MT19937 jump polynomials the device-side legacy noise generator ships
(autompc_amd/data/mt19937_jump.npz, tools/mt_jump.py) against numpy's own generator: the window
s * J words further on must equal the correlation of g_s with the stream (CPU, no GPU)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _raw_stream(key, nwords):
    x = [int(v) for v in key]
    for k in range(nwords):
        y = (x[k] & 0x80000000) | (x[k + 1] & 0x7fffffff)
        x.append(x[k + 397] ^ (y >> 1) ^ (0x9908b0df if (y & 1) else 0))
    return np.array(x[624:], dtype=np.uint64)


def _temper(v):
    v = v ^ (v >> 11)
    v = v ^ ((v << 7) & 0x9d2c5680)
    v = v ^ ((v << 15) & 0xefc60000)
    return (v ^ (v >> 18)) & 0xffffffff


_TABLES = np.load(os.path.join(ROOT, "autompc_amd", "data", "mt19937_jump.npz"))


@pytest.mark.parametrize("jb", [int(j) for j in _TABLES["jumps"]])      # one table per segment length
def test_jump_polynomials_reproduce_numpys_stream(jb):
    polys = _TABLES["polys_%d" % jb]
    assert polys.shape[1] == 624 and jb >= 34
    J = jb * 624
    rs = np.random.RandomState(2024)
    key = rs.get_state()[1]
    x = _raw_stream(key, 19937 + 624)                     # x[0] = the first word numpy would output
    for s in (1, 2, polys.shape[0]):
        bits = np.unpackbits(polys[s - 1].view(np.uint8), bitorder="little")[:19937].astype(bool)
        idx = np.nonzero(bits)[0]
        win = np.array([np.bitwise_xor.reduce(x[idx + p]) for p in range(5)], dtype=np.uint64)
        ref = np.random.RandomState(1)
        ref.set_state(("MT19937", key, 624))
        ref.randint(0, 2 ** 32, size=s * J, dtype=np.uint64)          # skip s*J outputs
        out = ref.randint(0, 2 ** 32, size=5, dtype=np.uint64)
        np.testing.assert_array_equal(_temper(win), out)
