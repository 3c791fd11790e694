"""
Host-side algebra that turns polynomial surface descriptions into the monomial terms
``sum c_ij x^i y^j`` the device evaluates (``xypoly_eval`` in csrc/prt_device.h):

* Zernike shapes (reference: surface_shape.py:927-1160).  The reference's polynomials are the
  un-normalised ``Z_n^m = R_n^|m|(rho) cos(m phi)`` (m >= 0) / ``sin(|m| phi)`` (m < 0) with the
  Fringe or ANSI single index; ``rho^|m| cos/sin(|m| phi) = Re/Im (x + i y)^|m|`` and
  ``rho^2 = x^2 + y^2`` make every one of them a polynomial with INTEGER coefficients in the
  normalised coordinates, computed here exactly.
* polynomials seen from a translated frame (``LinearCombination`` of shapes whose frames are
  shifted against each other): ``p(x - dx, y - dy)`` by binomial expansion.
"""
import math
from fractions import Fraction


def fringe_nm(j):
    """Fringe index (1-based) -> (n, m)  (surface_shape.py:1108-1115)"""
    root = int(math.ceil(math.sqrt(j)))
    next_sq = root * root
    m_plus_n = 2 * root - 2
    m = int(math.ceil((next_sq - j) / 2.0))
    n = m_plus_n - m
    if (next_sq - j) % 2 == 1:
        m = -m
    return (n, m)


def ansi_nm(j):
    """ANSI / OSA index (coefficients start at 1 for index 0) -> (n, m)  (:1129-1134)"""
    j0 = j - 1
    n = int(math.floor((-1.0 + math.sqrt(1.0 + 8.0 * j0)) * 0.5))
    m = n - 2 * j0 + n * (n + 1)
    return (n, -m)


INDEXING = {"fringe": fringe_nm, "ansi": ansi_nm}


def _mul(p, q):
    out = {}
    for ((a, b), u) in p.items():
        for ((c, d), v) in q.items():
            key = (a + c, b + d)
            out[key] = out.get(key, 0) + u * v
    return out


def _pow(p, e):
    out = {(0, 0): 1}
    for _ in range(e):
        out = _mul(out, p)
    return out


def zernike_monomials(n, m):
    """{(i, j): integer coefficient} of Z_n^m(x, y) on the unit disk"""
    omega = abs(m)
    if (n - omega) % 2 != 0 or omega > n:
        return {}
    # angular part: Re / Im of (x + i y)^omega
    ang = {}
    for q in range(omega + 1):
        # C(omega, q) x^(omega-q) (i y)^q ; i^q = 1, i, -1, -i
        c = math.comb(omega, q)
        phase = q % 4
        if m >= 0 and phase in (0, 2):
            ang[(omega - q, q)] = ang.get((omega - q, q), 0) + (c if phase == 0 else -c)
        if m < 0 and phase in (1, 3):
            ang[(omega - q, q)] = ang.get((omega - q, q), 0) + (c if phase == 1 else -c)
    r2 = {(2, 0): 1, (0, 2): 1}
    out = {}
    for l in range((n - omega) // 2 + 1):
        # radial coefficient (-1)^l (n-l)! / (l! ((n+omega)/2 - l)! ((n-omega)/2 - l)!)  (:1005-1006)
        rc = Fraction((-1) ** l * math.factorial(n - l),
                      math.factorial(l) * math.factorial((n + omega) // 2 - l)
                      * math.factorial((n - omega) // 2 - l))
        assert rc.denominator == 1
        term = _mul(ang, _pow(r2, (n - omega) // 2 - l))
        for (key, v) in term.items():
            out[key] = out.get(key, 0) + int(rc) * v
    return {k: v for (k, v) in out.items() if v != 0}


def zernike_terms(indexing, normradius, coefficients):
    """physical-coordinate monomial terms {(i, j): c} of sum_j Z_j(x / R, y / R) val_j"""
    to_nm = INDEXING[indexing]
    out = {}
    for (num, val) in enumerate(coefficients):
        if val == 0.0:
            continue
        (n, m) = to_nm(num + 1)
        for ((i, j), c) in zernike_monomials(n, m).items():
            out[(i, j)] = out.get((i, j), 0.0) + float(val) * c / float(normradius) ** (i + j)
    return out


def xy_terms(normradius, terms):
    """XYPolynomials coefficients (i, j, c) -> physical monomial terms (surface_shape.py:785-793)"""
    out = {}
    for (i, j, c) in terms:
        out[(int(i), int(j))] = out.get((int(i), int(j)), 0.0) + float(c) / float(normradius) ** (int(i) + int(j))
    return out


def shifted(terms, dx, dy):
    """terms of p(x - dx, y - dy)"""
    if dx == 0.0 and dy == 0.0:
        return dict(terms)
    out = {}
    for ((i, j), c) in terms.items():
        for a in range(i + 1):
            fa = math.comb(i, a) * (-dx) ** (i - a)
            for b in range(j + 1):
                fb = math.comb(j, b) * (-dy) ** (j - b)
                out[(a, b)] = out.get((a, b), 0.0) + c * fa * fb
    return out


def rotated(terms, rot):
    """terms of p(xs, ys) with (xs, ys) = rot^T (x, y), rot = [[r00, r01], [r10, r11]] (the 2 x 2 block of a rotation
    about z that takes part coordinates to combination coordinates): every monomial xs^i ys^j becomes a
    homogeneous polynomial of the same degree, so the set of degrees -- and the dense triangle the device
    evaluates -- stays what it was"""
    ((r00, r01), (r10, r11)) = rot
    if r00 == 1.0 and r11 == 1.0 and r01 == 0.0 and r10 == 0.0:
        return dict(terms)
    xs = {(1, 0): r00, (0, 1): r10}          # xs = r00 x + r10 y
    ys = {(1, 0): r01, (0, 1): r11}          # ys = r01 x + r11 y
    out = {}
    powers_x = {0: {(0, 0): 1.0}}
    powers_y = {0: {(0, 0): 1.0}}

    def power(cache, base, e):
        if e not in cache:
            cache[e] = _mul_float(power(cache, base, e - 1), base)
        return cache[e]
    for ((i, j), c) in terms.items():
        for (key, v) in _mul_float(power(powers_x, xs, i), power(powers_y, ys, j)).items():
            out[key] = out.get(key, 0.0) + c * v
    return out


def _mul_float(p, q):
    out = {}
    for ((a, b), u) in p.items():
        for ((c, d), v) in q.items():
            out[(a + c, b + d)] = out.get((a + c, b + d), 0.0) + u * v
    return out


def add_scaled(acc, terms, factor):
    for (key, c) in terms.items():
        acc[key] = acc.get(key, 0.0) + factor * c
    return acc
