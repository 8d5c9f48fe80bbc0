#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of AUTO's plan on one HBM-regime shape (two separate --pmc passes, kernel trace only):
#   bash tools/probes/regime_pmc.sh <M=K> <N> <per_row>      -> gpurun_out/regime_pmc/<M>_<N>_<d>/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/regime_pmc/$1_$2_$3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/tools/probes/regime_one.py $1 $2 $3 > $OUT/$c.log 2>&1 || { echo "$c pass failed"; tail -5 $OUT/$c.log; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
alg = int(re.search(r"alg_bytes (\d+)", open(out + "/FETCH_SIZE.log").read()).group(1))
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + f"/{c}/**/*_counter_collection.csv", recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if "spmm" in r["Kernel_Name"]:
            per.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    tot[c] = sum(sum(v) for v in per.values()) / 4  # four products
    print(c, {k[-60:]: round(sum(v) / 4) for k, v in per.items()})
traffic = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024  # gfx950: FETCH_SIZE x 2 for 16-byte-per-lane reads (KB units)
print(f"{out.split('/')[-1]}: traffic per product {traffic / 1e9:.2f} GB = {traffic / alg:.3f} x algorithmic ({alg / 1e9:.2f} GB)")
PY
