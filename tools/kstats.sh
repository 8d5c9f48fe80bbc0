#!/bin/bash
# Per-kernel time of any command on the GPU box:  bash tools/kstats.sh <tag> <program> [args…]
# (rocprofv3 --kernel-trace --stats, csv; the summary lands in gpurun_out/<tag>/ and is printed)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- "$@" > $OUT/run.log 2>&1 || { echo "run failed"; tail -20 $OUT/run.log; exit 1; }
F=$(find $OUT -name "*_kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:10.1f} us  total {float(r['TotalDurationNs'])/1e6:9.3f} ms")
PY
grep -v "^W2026\|amdgpu.ids" $OUT/run.log | tail -12
