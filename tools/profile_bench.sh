#!/bin/bash
# Runs on the GPU box (via gpurun): bench.py, then rocprofv3 kernel-trace stats and the two PMC
# passes of the same command; everything lands under gpurun_out/<tag>/.
#   bash tools/profile_bench.sh <tag> [bench args…]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-prof}; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python3 $R/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --no-cpu-baseline --no-live-pmc "$@" > $OUT/stats.log 2>&1 || { echo "stats run failed"; tail -5 $OUT/stats.log; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc "$@" > $OUT/pmc_fetch.log 2>&1 || { echo "fetch run failed"; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc "$@" > $OUT/pmc_write.log 2>&1 || { echo "write run failed"; exit 1; }
# third pass: which share of the L2s' fabric requests went to DRAM (the rest: Infinity Cache / other agents)
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum --output-format csv -d $OUT/pmc_dram -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc "$@" > $OUT/pmc_dram.log 2>&1 || { echo "dram-counter run failed (counters unavailable?)"; tail -3 $OUT/pmc_dram.log; }
find $OUT -name "*_kernel_stats.csv" | head -1 | xargs head -5
