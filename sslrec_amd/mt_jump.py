"""Jump-ahead polynomials of MT19937 (host side of csrc/mt19937.hip's many-workgroup replay).

One step of the generator -- the window of 624 consecutive words x[k..k+623] moving on by one word -- is a linear map F
of the window's bits over GF(2).  For a jump of J words write x^J = q(x) phi(x) + g(x) with phi the characteristic
polynomial of the generator (degree 19937): on windows the generator itself has produced, phi(F) = 0, so

    window_J = F^J window_0 = g(F) window_0 = XOR over the set bits i of g of window_i,        i <= 19936,

i.e. the state J words ahead is an XOR of windows of the NEXT 19937 + 623 words only, whatever J is (Haramoto, Matsumoto,
Nishimura, Panneton, L'Ecuyer: "Efficient jump ahead for F2-linear random number generators", 2008).  The device does the
XOR (mt_poly_apply_kernel); this module supplies phi and the g's: plain Python integers as GF(2) polynomials (bit i = the
coefficient of x^i), about a second of arithmetic once per process.

phi is not typed in: it is the minimal polynomial of any output bit stream of the generator (Berlekamp-Massey over
2 x 19937 bits of numpy's MT19937, which is the same generator as torch's CPU one), checked to have degree 19937.
"""
import functools

import numpy as np

DEGREE = 19937
N = 624
_SPREAD = {ord('0'): '00', ord('1'): '01'}


def _berlekamp_massey(bits):
    """connection polynomial C (C_0 = 1) and length L of the shortest LFSR producing `bits` (s[n] = XOR_i C_i s[n-i])"""
    c, b, length, m = 1, 1, 0, 1
    win = 0                                      # bit i = s[n - i]
    for n, s in enumerate(bits):
        win = (win << 1) | int(s)
        d = (c & win).bit_count() & 1
        if d == 0:
            m += 1
        elif 2 * length <= n:
            c, b = c ^ (b << m), c
            length = n + 1 - length
            m = 1
        else:
            c ^= b << m
            m += 1
    return c, length


@functools.lru_cache(maxsize=None)
def charpoly():
    """phi(x), degree 19937, as an int"""
    raw = np.random.MT19937(20240925).random_raw(2 * DEGREE + 64)
    c, length = _berlekamp_massey((raw & 1).astype(np.uint8).tolist())
    if length != DEGREE:
        raise RuntimeError('MT19937: minimal polynomial of degree %d found, 19937 expected' % length)
    # the characteristic polynomial is the reciprocal of the connection polynomial
    return _reverse(c, DEGREE)


def _reverse(p, degree):
    return int(bin(p)[2:].zfill(degree + 1)[::-1], 2)


def _reduce(a, phi=None):
    phi = charpoly() if phi is None else phi
    while True:
        top = a.bit_length() - 1
        if top < DEGREE:
            return a
        a ^= phi << (top - DEGREE)


def _square(a):
    return _reduce(int(bin(a)[2:].translate(_SPREAD), 2))


def _mul(a, b):
    acc = 0
    if a.bit_count() < b.bit_count():
        a, b = b, a
    while b:
        low = b & -b
        acc ^= a << (low.bit_length() - 1)
        b ^= low
    return _reduce(acc)


@functools.lru_cache(maxsize=None)
def xpow(e):
    """x^e mod phi"""
    if e < DEGREE:
        return 1 << e
    r = 1
    for bit in bin(e)[2:]:
        r = _square(r)
        if bit == '1':
            r = _reduce(r << 1)
    return r


def poly_words(g):
    """the 624 uint32 words the device reads: bit i of the polynomial = bit (i & 31) of word (i >> 5)"""
    return np.frombuffer(g.to_bytes(N * 4, 'little'), dtype=np.uint32).copy()


@functools.lru_cache(maxsize=None)
def two_level_table(stretch_blocks, fan1, fan2):
    """[fan1 - 1 + fan2 - 1, 624] words: x^(624 S j) for j = 1..fan1-1, then x^(624 S fan1 k) for k = 1..fan2-1 -- worker
    fan1 k + j of a large draw starts at block S (fan1 k + j): one level-2 jump from the base state, then one level-1 jump"""
    base = xpow(N * stretch_blocks)
    rows, p = [], base
    for _ in range(1, fan1):
        rows.append(p)
        p = _mul(p, base)
    base2 = p                                    # x^(624 S fan1)
    for _ in range(1, fan2):
        rows.append(p)
        p = _mul(p, base2)
    return np.stack([poly_words(g) for g in rows])


# ---- a plain restatement for the tests ------------------------------------------------------------------------------
def raw_stream(block, n_words):
    """x[0 .. n_words) from the block x[0..623] (untempered words, numpy, vectorised 227 at a time)"""
    x = np.zeros(max(n_words, N) + 227, dtype=np.uint32)
    x[:N] = block
    k = N
    while k < n_words:
        m = min(227, x.size - k)
        y = (x[k - 624:k - 624 + m] & np.uint32(0x80000000)) | (x[k - 623:k - 623 + m] & np.uint32(0x7fffffff))
        x[k:k + m] = x[k - 227:k - 227 + m] ^ (y >> np.uint32(1)) ^ np.where(y & np.uint32(1), np.uint32(0x9908b0df), np.uint32(0))
        k += m
    return x[:max(n_words, N)]


def project_to_image(block):
    """the block with the low 31 bits of word 0 replaced by what the recurrence implies (a no-op on every block the
    generator produced; a seeded block differs only in bits no future output depends on)"""
    b = np.array(block, dtype=np.uint32, copy=True)
    t = int(b[623] ^ b[396])
    lsb = t >> 31
    t ^= 0x9908b0df if lsb else 0
    b[0] = np.uint32((int(b[0]) & 0x80000000) | ((t & 0x3fffffff) << 1) | lsb)
    return b


def apply_poly(block, g):
    """g(F) block on the host (tests): XOR of the windows at the set bits of g"""
    x = raw_stream(project_to_image(block), DEGREE + N)
    acc = np.zeros(N, dtype=np.uint32)
    i = 0
    while g:
        if g & 1:
            acc ^= x[i:i + N]
        g >>= 1
        i += 1
    return acc
