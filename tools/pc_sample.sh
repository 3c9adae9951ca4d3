#!/bin/bash
# PC sampling of one launch of integrate_kernel (rocprofv3 --pc-sampling-beta-enabled, stochastic, hardware-based): WHERE the waves'
# issue cycles go, by instruction class and by source line (a -gline-tables-only build of the library: `make -C csrc lines`, same code).
# usage: tools/pc_sample.sh <tag> <pmc-child spec: W,H,spp,depth,seeding,arith,accel,0,1,1,frame> [interval, power of two cycles]
# A sampling run that does not come back is killed after 300 s.
set -u
TAG=$1; SPEC=$2; INT=${3:-1048576}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pcs_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
LIBG=$R/trace-of-radiance_amd/lib/libtor_mi355x_g.so
[ -f "$LIBG" ] && export TOR_AB_LIB=$LIBG
rm -rf /tmp/pcs_$TAG
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit cycles --pc-sampling-method stochastic --pc-sampling-interval $INT \
  --kernel-trace --output-format csv -d /tmp/pcs_$TAG -o pcs -- python $R/bench.py --pmc-child $SPEC > "$OUT/run.log" 2>&1
echo "rocprofv3 rc=$?" >> "$OUT/run.log"
ls -la /tmp/pcs_$TAG/* >> "$OUT/run.log" 2>&1
python $R/tools/pc_sample_summary.py /tmp/pcs_$TAG "$OUT/summary.txt" >> "$OUT/run.log" 2>&1
tail -5 "$OUT/run.log"
