#!/usr/bin/env python3
"""Jump-ahead polynomials of MT19937 for the block-parallel legacy-stream generator
(autompc_amd/csrc/legacy_rng_kernels.hpp).

MT19937's word recurrence x[k+624] = x[k+397] ^ twist(x[k], x[k+1]) is linear over GF(2); with
phi(t) its characteristic polynomial (degree 19937), the state J word-steps ahead is g(A) s with
g = t^J mod phi, and because A^i s is just the window of the stream i words further on,
    x[J + p] = XOR over { i : g_i = 1 } of x[i + p]           (p = 0 .. 623)
-- a correlation of the bit vector g with the first 19937 + 624 words of the stream, which every
segment of the stream can evaluate independently.  This script computes phi with Berlekamp-Massey
from a generated bit sequence and then g_s = t^(s*J) mod phi for s = 1 .. S (J = 64 and 256 blocks of 624
words: a segment is what one workgroup regenerates sequentially, ~150 us; the chain runs several control steps ahead of the draws), and stores them as uint32 words in autompc_amd/data/mt19937_jump.npz.  Pure integer
arithmetic (polynomials are Python ints, bit j = coefficient of t^j); takes a few seconds.
The result is checked against numpy's own generator before it is written.
"""
import os
import sys

import numpy as np

N, M, DEG = 624, 397, 19937
JUMPS = (64, 256)     # segment lengths in blocks: one table each (short chains / few jump evaluations)
S_MAX = 150


def raw_stream(key, nwords):
    """Untempered words following the window `key` (key itself excluded)."""
    x = [int(v) for v in key]
    out = []
    for k in range(nwords):
        y = (x[k] & 0x80000000) | (x[k + 1] & 0x7fffffff)
        v = x[k + M] ^ (y >> 1) ^ (0x9908b0df if (y & 1) else 0)
        x.append(v)
        out.append(v)
    return out


def berlekamp_massey(bits):
    """Connection polynomial C (int, bit i = c_i, c_0 = 1) and linear complexity L of a GF(2)
    sequence.  The sequence is kept reversed in one int so that the discrepancy
    sum_i c_i s_{k-i} is popcount(C & (r >> shift)) & 1."""
    n = len(bits)
    # reversed sequence: bit (n-1-i) = bits[i]  ->  window of s_{k}, s_{k-1}, .. is a right shift
    r = 0
    for i, b in enumerate(bits):
        if b:
            r |= 1 << (n - 1 - i)
    C, B, L, m = 1, 1, 0, 1
    for k in range(n):
        # s_{k-i} sits at bit (n-1-k+i) of r: shifting right by (n-1-k) puts s_{k-i} at bit i
        win = r >> (n - 1 - k)
        d = bin(C & win).count("1") & 1
        if d:
            T = C
            C ^= B << m
            if 2 * L <= k:
                L, B, m = k + 1 - L, T, 1
            else:
                m += 1
        else:
            m += 1
    return C, L


def clmul(a, b):
    r = 0
    while b:
        low = b & -b
        r ^= a << (low.bit_length() - 1)
        b ^= low
    return r


def polymod(p, phi, deg):
    while p.bit_length() - 1 >= deg:
        p ^= phi << (p.bit_length() - 1 - deg)
    return p


def main():
    rs = np.random.RandomState(12345)
    key = rs.get_state()[1]
    bits = [v & 1 for v in raw_stream(key, 2 * DEG + 64)]
    C, L = berlekamp_massey(bits)
    assert L == DEG, "linear complexity %d" % L
    # characteristic polynomial: phi_j = c_{L-j}
    phi = 0
    for i in range(L + 1):
        if (C >> i) & 1:
            phi |= 1 << (L - i)
    assert (phi >> DEG) & 1 and phi & 1

    def mulmod(a, b):
        return polymod(clmul(a, b), phi, DEG)

    def powmod_t(e):
        result, base = 1, 2          # base = t
        while e:
            if e & 1:
                result = mulmod(result, base)
            base = mulmod(base, base)
            e >>= 1
        return result

    def table(jump_blocks):
        """[S_MAX, 624] words of g_s = t^(s * jump_blocks * 624) mod phi, s = 1 .. S_MAX."""
        J = jump_blocks * N
        g1 = powmod_t(J)
        polys, g = [], 1
        for s in range(1, S_MAX + 1):
            g = mulmod(g, g1)
            polys.append(g)
        # check against numpy's generator: window at word offset 624 + s*J from the correlation
        rs = np.random.RandomState(777)
        key = rs.get_state()[1]
        need = N + DEG + N
        x = raw_stream(key, need)               # x[0] = stream word 624 relative to `key`
        for s in (1, 3):
            gs = polys[s - 1]
            win = [0] * 4
            for i in range(DEG):
                if (gs >> i) & 1:
                    for p in range(4):
                        win[p] ^= x[i + p]
            chk = np.random.RandomState(777)
            chk.set_state(("MT19937", key, N))
            raw = chk.randint(0, 2 ** 32, size=s * J + 8, dtype=np.uint64)   # tempered outputs
            y = win[:4]
            t = []
            for v in y:
                v ^= v >> 11; v ^= (v << 7) & 0x9d2c5680; v ^= (v << 15) & 0xefc60000; v ^= v >> 18
                t.append(v & 0xffffffff)
            assert t == [int(a) for a in raw[s * J:s * J + 4]], "jump polynomial %d does not reproduce numpy's stream" % s
        arr = np.zeros((S_MAX, N), dtype=np.uint32)
        for s, gs in enumerate(polys):
            for w in range(N):
                arr[s, w] = (gs >> (32 * w)) & 0xffffffff
        return arr
    tables = {j: table(j) for j in JUMPS}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "autompc_amd", "data",
                       "mt19937_jump.npz")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, jumps=np.array(sorted(tables)), **{"polys_%d" % j: a for j, a in tables.items()})
    print("wrote %s  (%d polynomials each for jumps of %s blocks)" % (out, S_MAX, sorted(tables)))


if __name__ == "__main__":
    main()
