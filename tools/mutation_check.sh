#!/bin/bash
# Mutation check of the conservative screens (tor_screen.hpp): builds the library with every margin set to zero
# (-DTOR_SCREEN_MUTATE) into a scratch directory and runs the host tests of tests/test_screen.py against it.  They must FAIL:
# that is the evidence that the adversarial pairs sit on the decision boundary and that the margins are what keeps them.
# usage (here, no GPU needed): bash tools/mutation_check.sh  -> profiles/r5_mutation_check.txt
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/profiles/r5_mutation_check.txt}
D=$(mktemp -d /tmp/tor_mut_XXXX)
FLAGS="-O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -munsafe-fp-atomics -Wall -Wno-unused-function -pthread"
for lvl in 1 2; do
  mkdir -p $D/$lvl
  make -C $R/trace-of-radiance_amd/csrc OUT=$D/$lvl CXXFLAGS="$FLAGS -DTOR_SCREEN_MUTATE=$lvl" > $D/build$lvl.log 2>&1 || { tail -20 $D/build$lvl.log; exit 2; }
done
cd $R
{
  echo "# tools/mutation_check.sh: the host tests of the conservative screens against builds whose margins are set to zero"
  echo "# expected: FAILURES / misses (a test that still passes would not be testing the boundary); the product build passes all of them"
  echo "## -DTOR_SCREEN_MUTATE=1 (every margin of the first form, the second form and the plane screen): tests/test_screen.py"
  TOR_AB_LIB=$D/1/libtor_mi355x.so python -m pytest tests/test_screen.py -q -p no:cacheprovider 2>&1 | grep -E "^FAILED|passed|failed" | sed 's/ - .*//'
  echo "## -DTOR_SCREEN_MUTATE=2 (only the plane screen's threshold: R^2 w2 without the 2^-40 and the 2^-45 B^2): needed pairs dropped by stage one alone"
  TOR_AB_LIB=$D/2/libtor_mi355x.so python - <<'PY'
import importlib, sys
import numpy as np
sys.path.insert(0, "tests")
from test_filter32 import _unit
tor = importlib.import_module("trace-of-radiance_amd")
rng = np.random.default_rng(21)
n = 300_000
# rays whose ground track is tangent to the sphere's ground circle at the equator point, nudged by ulps (tests/test_screen.py)
c0 = np.column_stack([rng.uniform(-11, 11, n), rng.choice([0.2, 0.7, -3.0], n), rng.uniform(-11, 11, n)])
r = rng.choice([0.2, 1.0, 7.5], size=n)
ang = rng.uniform(0, 2 * np.pi, n)
nrm = np.column_stack([np.cos(ang), np.zeros(n), np.sin(ang)]); tang = np.column_stack([-np.sin(ang), np.zeros(n), np.cos(ang)])
slope = rng.choice([0.0, 1e-3, 0.5, 3.0, 1e3], size=(n, 1)) * rng.choice([-1.0, 1.0], size=(n, 1))
dirn = tang + slope * np.array([0.0, 1.0, 0.0])
target = c0 + nrm * r[:, None]
o = target - dirn / np.linalg.norm(dirn, axis=1, keepdims=True) * rng.uniform(0.5, 30.0, (n, 1))
d = (target - o) * (1.0 + rng.integers(-4, 5, size=(n, 3)) * 2.0 ** -52)
fm = rng.uniform(-0.5, 1.5, n)
for name, dc, mv in (("statics (xkind 10 / 11)", np.zeros((n, 3)), np.zeros(n, dtype=np.int32)),
                     ("movers along y (12 / 14)", np.column_stack([np.zeros(n), rng.uniform(-.5, .5, n), np.zeros(n)]), np.ones(n, dtype=np.int32)),
                     ("movers in general position (13)", np.column_stack([rng.uniform(-.7, .7, n), rng.uniform(-.5, .5, n), rng.uniform(-.7, .7, n)]), np.ones(n, dtype=np.int32))):
    keep, need = tor.selftest_screen2(o, d, c0 - dc * fm[:, None], dc, mv, fm * mv, r * r, 2)
    print(f"   {name}: {int(np.count_nonzero((need != 0) & (keep == 0)))} of {int(np.count_nonzero(need))} needed pairs dropped")
PY
} | tee $OUT
rm -rf $D
