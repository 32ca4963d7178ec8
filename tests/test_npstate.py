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


def test_recognition_reports_what_it_found_and_never_raises():
    """the direct view relies on numpy's private RandomState layout.  On every numpy this
    suite runs under, `get()` either proves the layout on a private instance and on the global one, or returns
    None (MPPI then goes through get_state / set_state: tests/test_gpu_legacy_noise.py forces that route on the
    device) -- it must never raise, and a recognised layout must survive seeding and draws.  The numpy version is
    part of the assertion message so a bump that loses the fast path shows in the log instead of skipping silently."""
    ls = _npstate.get()
    assert ls is None or ls.current(), "numpy %s: a stale view was handed out" % np.__version__
    if ls is None:
        import warnings
        warnings.warn("numpy %s: RandomState layout not recognised -- MPPI(noise='numpy') uses get_state()/set_state() "
                      "(slower per call, same results)" % np.__version__)


def test_a_layout_that_does_not_match_is_refused_not_trusted(monkeypatch):
    """What happens on a numpy whose RandomState keeps its Gaussian cache elsewhere: the scan of the PRIVATE probe
    instance finds nothing, nothing is ever written through raw addresses of the global generator, `get()` warns
    once and returns None from then on -- and the global stream is untouched by the attempt."""
    import ctypes
    monkeypatch.setattr(_npstate, "_state", None)
    monkeypatch.setattr(_npstate, "_failed", False)
    monkeypatch.setattr(ctypes, "string_at", lambda addr, size: bytes(size))      # a layout with no cache in sight
    np.random.seed(5)
    np.random.normal(size=3)                                  # (a value sits in the cache)
    before = np.random.get_state()
    with pytest.warns(RuntimeWarning, match="in-place access"):
        assert _npstate.get() is None
    assert _npstate.get() is None                             # sticky: no second probe, no second warning
    after = np.random.get_state()
    np.testing.assert_array_equal(before[1], after[1])
    assert before[2:] == after[2:]
