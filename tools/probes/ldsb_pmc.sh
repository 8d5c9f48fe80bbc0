set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/a -- python3 $R/tools/probes/ldsb_pmc.py > $OUT/a.log 2>&1 || { echo "pmc a failed"; tail -5 $OUT/a.log; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/b -- python3 $R/tools/probes/ldsb_pmc.py > $OUT/b.log 2>&1 || { echo "pmc b failed"; tail -5 $OUT/b.log; }
python3 - <<PY
import csv,glob,collections
for tag in ("a","b"):
    fs=glob.glob("$OUT/%s/**/*counter_collection.csv"%tag,recursive=True)
    if not fs: print(tag,"no csv"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"]
        if "ldsq" in k or "ldsb" in k:
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,d in acc.items():
        print(tag,k)
        for c,v in d.items():
            print("    %-24s"%c," ".join("%.3g"%x for x in v))
PY
