#!/bin/bash
# Bytes that leave the L2s (FETCH_SIZE, gfx950: x 2 for 16-byte-per-lane reads; WRITE_SIZE) per product for pinned plans on one L2-regime
# shape — what the column panels are for:   bash tools/probes/l2_pmc.sh M K N per_row "variant variant …" [pattern]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in $5; do
  OUT=$R/gpurun_out/l2_pmc/$1_$2_$3_$4_${6:-uniform}_v$v
  mkdir -p $OUT
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/tools/probes/l2_pmc_one.py $1 $2 $3 $4 $v ${6:-uniform} > $OUT/$c.log 2>&1 || { echo "$c pass failed"; tail -5 $OUT/$c.log; exit 1; }
  done
  python3 - "$OUT" "$v" <<'PY'
import csv, glob, sys, re
out, v = sys.argv[1], sys.argv[2]
alg = int(re.search(r"alg_bytes (\d+)", open(out + "/FETCH_SIZE.log").read()).group(1))
tot, names = {}, set()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + f"/{c}/**/*_counter_collection.csv", recursive=True)[0]
    s = 0.0
    for r in csv.DictReader(open(f)):
        if "spmm" in r["Kernel_Name"]:
            s += float(r["Counter_Value"])
            names.add(re.search(r"spmm_\w+", r["Kernel_Name"]).group(0))
    tot[c] = s / 4  # four products
fetch, write = 2 * tot["FETCH_SIZE"] * 1024, tot["WRITE_SIZE"] * 1024
print(f"{out.split('/')[-1]:<36} plan {v:>2} ({', '.join(sorted(names))}): beyond the L2s per product: read {fetch / 1e9:7.3f} GB + written {write / 1e9:6.3f} GB = "
      f"{(fetch + write) / alg:.3f} x algorithmic ({alg / 1e9:.3f} GB)")
PY
done
