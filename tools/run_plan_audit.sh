R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/final5b
python bench.py > gpurun_out/final5b/bench_c3.json 2> gpurun_out/final5b/c3.err
python bench.py --workload c2 > gpurun_out/final5b/bench_c2.json 2> gpurun_out/final5b/c2.err
: > gpurun_out/final5b/plans.log
for a in "131072 512 64 0.1" "131072 512 64 0.02" "131072 512 64 0.008" "131072 512 64 0.005" "65536 128 256 0.25" "65536 128 256 0.1" "65536 128 256 0.05" "65536 256 128 0.2" "65536 256 128 0.03" "32768 512 128 0.1" "32768 512 128 0.01" "16384 2048 64 0.05" "16384 1024 64 0.05" "16384 512 256 0.1" "200000 300 64 0.05" "4096 4096 256 0.05" "16384 4096 256 0.1" "16384 16384 256 0.1" "8192 8192 1024 0.01"; do timeout -k 10 200 python tools/bench_plans.py $a 2>&1 | grep "^M " >> gpurun_out/final5b/plans.log; done
python3 - <<PY
import json
for w in ("c3","c2"):
    r=json.load(open("gpurun_out/final5b/bench_%s.json"%w)); print(w, r["ms_per_step"], r["roofline"]["frac"], r["roofline"].get("traffic"))
PY
tail -4 gpurun_out/final5b/plans.log
