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
    the double-double result; the fall-back is rare; h + corr is within 2^-65 of the exact value (the test's 2^-64 is not tight);
    signed zeros and the table's nodes and interval boundaries included."""
    L = oracle.lib()
    rng = np.random.default_rng(21)
    n = 1_500_000
    a = np.concatenate([rng.random(n) * (2.0 * 3.141592653589793),
                        (rng.integers(0, 257, 200000) * (np.pi / 128) + rng.standard_normal(200000) * 1e-9).clip(0, 6.2831853),
                        ((rng.integers(0, 256, 200000) + 0.5) * (np.pi / 128) + rng.standard_normal(200000) * 1e-12).clip(0, 6.2831853),
                        10.0 ** rng.uniform(-300, -3, 50000),
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
    assert 0.0005 < 1.0 - good.mean() < 0.004       # ~1 call in 600 falls back
    mp.mp.prec = 160
    worst = 0.0
    for i in rng.integers(0, a.size, 4000):
        x = mp.mpf(float(a[i]))
        for h, co, e in ((parts[i, 0], parts[i, 1], mp.sin(x)), (parts[i, 2], parts[i, 3], mp.cos(x))):
            if h != 0.0:
                worst = max(worst, float(abs((mp.mpf(float(h)) + mp.mpf(float(co)) - e) / mp.mpf(float(h)))))
    assert worst < 2.0 ** -65, worst


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
