"""autompc_amd._npstate: zero-copy view of numpy's global legacy generator (MT19937 key, position,
cached Gaussian) -- the state the device-side draw of the reference's noise starts from and hands
back (mppi.py:16-24, :126)."""
import numpy as np
import pytest

from autompc_amd import _npstate


def test_view_tracks_numpy_and_writes_are_numpys_state():
    ls = _npstate.get()
    if ls is None:
        pytest.skip("numpy's RandomState layout not recognised: the public get_state/set_state path is used")
    np.random.seed(123)
    for k in (0, 1, 5, 700, 3):
        np.random.normal(size=k)
        st = np.random.get_state()
        np.testing.assert_array_equal(ls.key[:624], st[1])
        assert int(ls.key[624]) == st[2] and ls.has_gauss.value == st[3]
        if st[3]:
            assert ls.gauss.value == st[4]
    # writing through the view == set_state: the following draws continue identically
    np.random.seed(7)
    np.random.normal(size=11)                       # odd count: a value sits in the cache
    st = np.random.get_state()
    ref = np.random.normal(size=9)
    np.random.seed(99)                              # scramble
    with ls.lock:
        ls.key[:624] = st[1]
        ls.key[624] = st[2]
        ls.has_gauss.value, ls.gauss.value = st[3], st[4]
    np.testing.assert_array_equal(np.random.normal(size=9), ref)
    assert ls.current()
