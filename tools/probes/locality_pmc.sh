#!/bin/bash
# Bytes that leave the L2s per product (FETCH_SIZE, gfx950: x 2 for 16-byte-per-lane reads) for a shuffled structured matrix
# with and without the inspector's locality order:   bash tools/probes/locality_pmc.sh "band1k/1M community/1M"
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for tag in $1; do
  for mode in natural plain scheduled; do
    OUT=$R/gpurun_out/locality_pmc/$(echo $tag | tr / _)_$mode
    mkdir -p $OUT
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/FETCH_SIZE -- python3 $R/tools/bench_locality_order.py --one $tag --mode $mode > $OUT/run.log 2>&1 || { echo "pass failed"; tail -5 $OUT/run.log; exit 1; }
    python3 - "$OUT" "$tag" "$mode" <<'PY'
import csv, glob, re, sys
out, tag, mode = sys.argv[1:4]
alg = int(re.search(r"algorithmic bytes per product (\d+)", open(out + "/run.log").read()).group(1))
f = glob.glob(out + "/FETCH_SIZE/**/*_counter_collection.csv", recursive=True)[0]
s, names = 0.0, set()
for r in csv.DictReader(open(f)):
    if "spmm_" in r["Kernel_Name"] and "sched_" not in r["Kernel_Name"]:
        s += float(r["Counter_Value"])
        names.add(re.search(r"spmm_\w+", r["Kernel_Name"]).group(0))
fetch = 2 * (s / 5) * 1024  # five products; KB
print(f"{tag:<16} {mode:<10} read beyond the L2s per product {fetch / 1e9:8.3f} GB = {fetch / alg:.3f} x algorithmic ({alg / 1e9:.2f} GB)   kernels: {', '.join(sorted(names))}")
PY
  done
done
