"""Zero-copy access to numpy's GLOBAL legacy generator (the stream the reference's MPPI draws
from: np.random.normal, autompc/control/mppi.py:16-24, :126).

``np.random.get_state()`` / ``set_state()`` convert the 624-word MT19937 key to and from Python
objects: ~50-70 us each, more than the device needs to make the draw itself.  The state lives in two
C structs: ``mt19937_state {uint32 key[624]; int pos;}`` -- whose address numpy publishes
(``bit_generator.ctypes.state_address``) -- and the legacy Gaussian cache ``aug_bitgen_t {bitgen_t*;
int has_gauss; double gauss;}`` inside the ``RandomState`` object, which it does not.  The cache is
located once -- on a PRIVATE ``RandomState``, never on the global one -- by writing a recognisable
value through ``set_state`` and scanning the object's memory for it, PROVEN there by writing through
the located fields and reading back through ``get_state``; the offset is then applied to the global
instance (same class, same size) and checked read-only against its ``get_state()``.
If anything does not check out (another numpy layout, a replaced global generator) ``get()`` returns
None and callers fall back to get_state / set_state.  Access is serialised with numpy's own lock.
"""
import ctypes
import struct
import warnings

import numpy as np


def _cache_offset():
    """Byte offset of ``aug_bitgen_t.gauss`` inside a RandomState object (``has_gauss`` is the int
    8 bytes before it), found and PROVEN on a private ``np.random.RandomState()`` -- the global
    generator is never written to while probing: ``np.random.get_state()`` / ``set_state()`` do not
    take the bit generator's lock, so a probe value installed there could be observed, or the
    restore clobbered, by another thread."""
    rs = np.random.RandomState(12345)
    if type(rs._bit_generator).__name__ != "MT19937":
        raise RuntimeError("legacy generator is not MT19937")
    saved = rs.get_state()
    probe = 0.8414709848078965
    rs.set_state(("MT19937", saved[1], saved[2], 1, probe))
    size = int(rs.__sizeof__())
    raw = ctypes.string_at(id(rs), size)
    pat = struct.pack("<d", probe)
    hits = [o for o in range(8, size - 7, 8) if raw[o:o + 8] == pat and struct.unpack_from("<i", raw, o - 8)[0] == 1]
    if len(hits) != 1:
        raise RuntimeError("legacy Gaussian cache not found in the RandomState object")
    off = hits[0]
    has_gauss = ctypes.c_int.from_address(id(rs) + off - 8)
    gauss = ctypes.c_double.from_address(id(rs) + off)
    # proof: what we write is what numpy reads, and the other way round
    for hg, gv in ((0, 0.0), (1, -2.5)):
        has_gauss.value, gauss.value = hg, gv
        st = rs.get_state()
        if st[3] != hg or st[4] != gv:
            raise RuntimeError("located fields are not the cache")
    rs.set_state(("MT19937", saved[1], saved[2], 0, 0.0))
    if has_gauss.value != 0:
        raise RuntimeError("located fields are not the cache")
    # and the key: state_address shows the generator's words
    key = np.ctypeslib.as_array(ctypes.cast(int(rs._bit_generator.ctypes.state_address),
                                            ctypes.POINTER(ctypes.c_uint32)), shape=(625,))
    st = rs.get_state()
    if not (np.array_equal(key[:624], st[1]) and int(key[624]) == int(st[2])):
        raise RuntimeError("state_address does not show the generator's key")
    return off, type(rs), size


class LegacyState:
    def __init__(self):
        off, cls, size = _cache_offset()
        rs = np.random.mtrand._rand
        bg = rs._bit_generator
        if type(bg).__name__ != "MT19937":
            raise RuntimeError("global legacy generator is not MT19937")
        if type(rs) is not cls or int(rs.__sizeof__()) != size:
            raise RuntimeError("global generator is not a plain RandomState")
        self.rs, self.bg, self.lock = rs, bg, bg.lock
        addr = int(bg.ctypes.state_address)
        self.key_ptr = ctypes.cast(addr, ctypes.POINTER(ctypes.c_uint32))
        self.key = np.ctypeslib.as_array(self.key_ptr, shape=(625,))     # [624] is `pos`
        self.has_gauss = ctypes.c_int.from_address(id(rs) + off - 8)
        self.gauss = ctypes.c_double.from_address(id(rs) + off)
        # read-only validation on the global instance: memory and get_state() agree
        st = rs.get_state()
        if not (np.array_equal(self.key[:624], st[1]) and int(self.key[624]) == int(st[2])
                and int(self.has_gauss.value) == int(st[3])
                and (int(st[3]) == 0 or float(self.gauss.value) == float(st[4]))):
            raise RuntimeError("the global generator's memory does not match get_state()")

    def current(self):
        """True while numpy's global generator is still the object this accessor was built on."""
        return np.random.mtrand._rand is self.rs and self.rs._bit_generator is self.bg


_state = None
_failed = False


def get():
    """The accessor, or None when numpy's internals are not laid out as expected (one warning; the
    callers then use the public get_state() / set_state(), ~100 us slower per control step)."""
    global _state, _failed
    if _failed:
        return None
    if _state is None or not _state.current():
        try:
            _state = LegacyState()
        except Exception as e:             # noqa: BLE001 -- any surprise: use the public API instead
            _state, _failed = None, True
            warnings.warn("autompc_amd: in-place access to numpy's global generator is unavailable (%s); "
                          "MPPI(noise='numpy') goes through np.random.get_state()/set_state()" % (e,),
                          RuntimeWarning, stacklevel=2)
            return None
    return _state
