"""Generate the double-double constants used by the portable (CPU==GPU bit-identical)
sin/cos/pow of this repo. Prints C initialisers; run once, paste into
oracle/tor_oracle.c and trace-of-radiance_amd/csrc/tor_math.hpp (both files carry the
same tables; tests/test_oracle_math.py re-derives them with mpmath and compares)."""
import mpmath as mp
mp.mp.prec = 400

def dd(x):
    hi = float(x)
    lo = float(x - mp.mpf(hi))
    return hi, lo

def fmt(v):
    return float.hex(v)

def show(name, vals):
    print(f"/* {name} */")
    for i, x in vals:
        hi, lo = dd(x)
        print(f"  {{ {fmt(hi)}, {fmt(lo)} }}, /* {i} */")

if __name__ == "__main__":
    # sin: S_k = (-1)^k/(2k+1)!  k=1..10 ; cos: C_k = (-1)^k/(2k)! k=1..11
    show("SIN_S", [(k, mp.mpf((-1) ** k) / mp.factorial(2 * k + 1)) for k in range(1, 11)])
    show("COS_C", [(k, mp.mpf((-1) ** k) / mp.factorial(2 * k)) for k in range(1, 12)])
    show("INV_ODD 1/(2k+1)", [(k, mp.mpf(1) / (2 * k + 1)) for k in range(0, 22)])
    show("INV_FACT 1/k!", [(k, mp.mpf(1) / mp.factorial(k)) for k in range(0, 17)])
    show("LN2", [(0, mp.log(2))])
    show("TWO_OVER_PI", [(0, 2 / mp.pi)])
    show("INV_LN2", [(0, 1 / mp.log(2))])
    # pi/2 in 4 pieces, first three with 33 significant bits (fdlibm split)
    p = mp.pi / 2
    pieces = []
    rem = p
    for i in range(3):
        f = float(rem)
        import struct
        b = struct.unpack("<Q", struct.pack("<d", f))[0]
        b &= ~((1 << 20) - 1)            # keep 33 bits (52-20=32 explicit + implicit)
        f = struct.unpack("<d", struct.pack("<Q", b))[0]
        pieces.append(f)
        rem = rem - mp.mpf(f)
    pieces.append(float(rem))
    print("/* PIO2 pieces */", ", ".join(fmt(x) for x in pieces))
    print("/* check */", [repr(x) for x in pieces])
