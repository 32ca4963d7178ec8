"""Zero-copy access to numpy's GLOBAL legacy generator (the stream the reference's MPPI draws
from: np.random.normal, autompc/control/mppi.py:16-24, :126).

``np.random.get_state()`` / ``set_state()`` convert the 624-word MT19937 key to and from Python
objects: ~50-70 us each, more than the device needs to make the draw itself.  The state lives in two
C structs: ``mt19937_state {uint32 key[624]; int pos;}`` -- whose address numpy publishes
(``bit_generator.ctypes.state_address``) -- and the legacy Gaussian cache ``aug_bitgen_t {bitgen_t*;
int has_gauss; double gauss;}`` inside the ``RandomState`` object, which it does not.  The cache is
located once by writing a recognisable value through ``set_state`` and scanning the object's memory
for it, then PROVEN by writing through the located fields and reading back through ``get_state``.
If anything does not check out (another numpy layout, a replaced global generator) ``get()`` returns
None and callers fall back to get_state / set_state.  Access is serialised with numpy's own lock.
"""
import ctypes
import struct

import numpy as np


class LegacyState:
    def __init__(self):
        rs = np.random.mtrand._rand
        bg = rs._bit_generator
        if type(bg).__name__ != "MT19937":
            raise RuntimeError("global legacy generator is not MT19937")
        self.rs, self.bg, self.lock = rs, bg, bg.lock
        addr = int(bg.ctypes.state_address)
        self.key_ptr = ctypes.cast(addr, ctypes.POINTER(ctypes.c_uint32))
        self.key = np.ctypeslib.as_array(self.key_ptr, shape=(625,))     # [624] is `pos`
        with self.lock:
            saved = rs.get_state()
            try:
                self._locate(saved)
            finally:
                rs.set_state(saved)

    def _locate(self, saved):
        rs = self.rs
        if not (np.array_equal(self.key[:624], saved[1]) and int(self.key[624]) == int(saved[2])):
            raise RuntimeError("state_address does not show the generator's key")
        probe = 0.8414709848078965
        rs.set_state(("MT19937", saved[1], saved[2], 1, probe))
        size = int(rs.__sizeof__())
        raw = ctypes.string_at(id(rs), size)
        pat = struct.pack("<d", probe)
        hits = [o for o in range(8, size - 7, 8) if raw[o:o + 8] == pat and struct.unpack_from("<i", raw, o - 8)[0] == 1]
        if len(hits) != 1:
            raise RuntimeError("legacy Gaussian cache not found in the RandomState object")
        self.has_gauss = ctypes.c_int.from_address(id(rs) + hits[0] - 8)
        self.gauss = ctypes.c_double.from_address(id(rs) + hits[0])
        # proof: what we write is what numpy reads, and the other way round
        self.has_gauss.value, self.gauss.value = 0, 0.0
        st = rs.get_state()
        if st[3] != 0 or st[4] != 0.0:
            raise RuntimeError("located fields are not the cache")
        self.has_gauss.value, self.gauss.value = 1, -2.5
        st = rs.get_state()
        if st[3] != 1 or st[4] != -2.5:
            raise RuntimeError("located fields are not the cache")
        rs.set_state(("MT19937", saved[1], saved[2], 0, 0.0))
        if self.has_gauss.value != 0:
            raise RuntimeError("located fields are not the cache")

    def current(self):
        """True while numpy's global generator is still the object this accessor was built on."""
        return np.random.mtrand._rand is self.rs and self.rs._bit_generator is self.bg


_state = None
_failed = False


def get():
    """The accessor, or None when numpy's internals are not laid out as expected."""
    global _state, _failed
    if _failed:
        return None
    if _state is None or not _state.current():
        try:
            _state = LegacyState()
        except Exception:                  # noqa: BLE001 -- any surprise: use the public API instead
            _state, _failed = None, True
            return None
    return _state
