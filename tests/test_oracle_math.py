"""Portable (CPU == GPU bit-identical) sin/cos/pow of the oracle: correctly rounded against
mpmath, and within 1 ulp of the host libm the reference uses."""
import ctypes as C

import mpmath as mp
import numpy as np

dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))


def test_sincos_correctly_rounded(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(11)
    a = np.concatenate([rng.random(200000) * (2.0 * 3.141592653589793),
                        np.array([0.0, 1e-300, 1e-20, np.pi / 2, np.pi, 3 * np.pi / 2,
                                  np.nextafter(2 * np.pi, 0), 0.7853981633974483, 2.356194490192345])])
    s = np.empty_like(a); c = np.empty_like(a); s2 = np.empty_like(a); c2 = np.empty_like(a)
    L.oracle_port_sincos(dp(a), dp(s), dp(c), a.size)
    L.oracle_libm_sincos(dp(a), dp(s2), dp(c2), a.size)
    assert np.all(np.abs(s - s2) <= np.spacing(np.abs(s2)))
    assert np.all(np.abs(c - c2) <= np.spacing(np.abs(c2)))
    assert (s != s2).mean() < 0.01 and (c != c2).mean() < 0.01
    mp.mp.prec = 300
    idx = np.concatenate([rng.integers(0, a.size, 1500), np.arange(a.size - 9, a.size)])
    for i in idx:
        assert float(mp.sin(mp.mpf(a[i]))) == s[i]
        assert float(mp.cos(mp.mpf(a[i]))) == c[i]


def test_sincos_fast_path_is_the_double_double_path(oracle):
    """port_sincos = fast path + rounding test, else the double-double path (round 5).  Wherever the test passes the fast result IS
    the double-double result; the fall-back is rare; h + corr is within 2^-64.9 |h| of the exact value -- the bound DERIVED term by
    term in oracle/tor_oracle.c / csrc/tor_math.hpp, a quarter of the rounding test's window 2^-63 |h| -- on EVERY argument of the
    set (round 6; round 5 sampled 4000 of them): signed zeros, the table's nodes, the half-interval boundaries and the neighbourhoods
    of the multiples of pi / 2 included.  The reference for that is the double-double evaluation (hi + lo), itself checked against
    mpmath (2^-73.7 at worst, at the ends of its pi / 4 reduction interval; 2^-84 typically: 500 times finer than the bound)."""
    L = oracle.lib()
    rng = np.random.default_rng(21)
    n = 1_500_000
    a = np.concatenate([rng.random(n) * (2.0 * 3.141592653589793),
                        (rng.integers(0, 257, 200000) * (np.pi / 128) + rng.standard_normal(200000) * 1e-9).clip(0, 6.2831853),
                        ((rng.integers(0, 256, 200000) + 0.5) * (np.pi / 128) + rng.standard_normal(200000) * 1e-12).clip(0, 6.2831853),
                        10.0 ** rng.uniform(-300, -3, 50000),
                        # the doubles around the multiples of pi / 2 (where a result crosses zero and |r| gets as small as it can)
                        np.concatenate([np.nextafter(np.full(41, k * np.pi / 2), np.inf) + np.arange(-20, 21) * np.spacing(k * np.pi / 2) for k in (1, 2, 3)]),
                        np.nextafter(2 * np.pi, 0) - np.arange(0, 40) * np.spacing(2 * np.pi),
                        np.array([0.0, 1e-300, 1e-20, np.pi / 2, np.pi, 3 * np.pi / 2, np.nextafter(2 * np.pi, 0)])])
    s = np.empty_like(a); c = np.empty_like(a); s2 = np.empty_like(a); c2 = np.empty_like(a); sf = np.empty_like(a); cf = np.empty_like(a)
    ok = np.zeros(a.size, dtype=np.int32); parts = np.zeros((a.size, 4))
    L.oracle_port_sincos(dp(a), dp(s), dp(c), a.size)
    L.oracle_port_sincos_slow(dp(a), dp(s2), dp(c2), a.size)
    L.oracle_port_sincos_fast(dp(a), dp(sf), dp(cf), ok.ctypes.data_as(C.POINTER(C.c_int32)), dp(parts), a.size)
    assert np.array_equal(s, s2) and np.array_equal(c, c2)
    assert np.array_equal(np.signbit(s), np.signbit(s2)) and np.array_equal(np.signbit(c), np.signbit(c2))
    good = ok.astype(bool)
    assert np.array_equal(sf[good], s2[good]) and np.array_equal(cf[good], c2[good])
    assert 0.001 < 1.0 - good.mean() < 0.008        # ~1 call in 300 falls back (window 2^-63 |h|)
    # |h + corr - exact| / |h| on every argument, against the double-double results (hi + lo) of the slow path
    lo = np.zeros((a.size, 2))
    L.oracle_port_sincos_slow_dd(dp(a), dp(s2), dp(c2), dp(lo), a.size)
    inside = (a >= 0.0) & (a < 6.2890625)            # the fast path's domain (parts are written there)
    worst = 0.0
    for h, co, hi, l in ((parts[:, 0], parts[:, 1], s2, lo[:, 0]), (parts[:, 2], parts[:, 3], c2, lo[:, 1])):
        m = inside & (h != 0.0)
        # h and hi are neighbouring doubles at most: h - hi is exact; the rest is far below their ulp
        err = np.abs((h[m] - hi[m]) + (co[m] - l[m])) / np.abs(h[m])
        worst = max(worst, float(err.max()))
        assert np.all(h[inside & (h == 0.0)] + co[inside & (h == 0.0)] == hi[inside & (h == 0.0)])
    assert worst < 2.0 ** -64.9, np.log2(worst)
    # ... and the double-double reference itself against mpmath
    mp.mp.prec = 200
    dd_worst = 0.0
    for i in rng.integers(0, a.size, 3000):
        x = mp.mpf(float(a[i]))
        for hi, l, e in ((s2[i], lo[i, 0], mp.sin(x)), (c2[i], lo[i, 1], mp.cos(x))):
            if e != 0:
                dd_worst = max(dd_worst, float(abs((mp.mpf(float(hi)) + mp.mpf(float(l)) - e) / e)))
    assert dd_worst < 2.0 ** -73, np.log2(dd_worst)


def test_pow5_and_gamma_pow_correctly_rounded(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(12)
    x = rng.random(100000) * 2.0
    y = np.empty_like(x)
    L.oracle_port_pow5(dp(x), dp(y), x.size)
    mp.mp.prec = 300
    for i in rng.integers(0, x.size, 1000):
        assert float(mp.mpf(x[i]) ** 5) == y[i]
    g = 1.0 / float(np.float32(2.2))
    x = np.concatenate([rng.random(50000), rng.random(50000) * 1e-6, np.array([0.0, 1.0, 2.0 ** -48, 3.0])])
    y = np.empty_like(x); y2 = np.empty_like(x)
    L.oracle_port_pow(dp(x), g, dp(y), x.size)
    L.oracle_libm_pow(dp(x), g, dp(y2), x.size)
    assert np.all(np.abs(y - y2) <= np.spacing(y2))
    for i in rng.integers(0, x.size, 600):
        if x[i] > 0:
            assert float(mp.mpf(x[i]) ** mp.mpf(g)) == y[i]
    assert y[-4] == 0.0 and y[-3] == 1.0


def test_dd_constant_tables_match_mpmath():
    """The tables pasted into oracle/tor_oracle.c and csrc/tor_math.hpp are what
    tools/gen_dd_constants.py prints."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_dd_constants.py")],
                         capture_output=True, text=True, check=True).stdout
    pairs = re.findall(r"\{ (\S+), (\S+) \}", out)
    # (the table of the fast sin / cos path lives in a generated include, one copy per side: `gen_dd_constants.py --table`)
    c_src = open(os.path.join(root, "oracle", "tor_oracle.c")).read() + open(os.path.join(root, "oracle", "tor_sincos_table.inc")).read()
    h_src = (open(os.path.join(root, "trace-of-radiance_amd", "csrc", "tor_math.hpp")).read()
             + open(os.path.join(root, "trace-of-radiance_amd", "csrc", "tor_sincos_table.inc")).read())
    assert open(os.path.join(root, "oracle", "tor_sincos_table.inc")).read() == \
        open(os.path.join(root, "trace-of-radiance_amd", "csrc", "tor_sincos_table.inc")).read()
    assert len(pairs) > 60
    for hi, lo in pairs:
        if hi in ("0x1.62e42fefa39efp-1", "0x1.45f306dc9c883p-1", "0x1.71547652b82fep+0"):
            assert hi in c_src and hi in h_src
            continue
        assert f"{hi}, {lo}" in c_src, (hi, lo)
        assert f"{hi}, {lo}" in h_src, (hi, lo)
