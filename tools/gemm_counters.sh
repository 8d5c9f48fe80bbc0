#!/bin/bash
# On the GPU box: SQ counters of the BERT products (tools/gemm_probe.py: 5 launches of q.kT, 5 of probs.V) in two
# rocprofv3 --pmc passes -> gpurun_out/<tag>/pmc{1,2}; summarised by tools/collect_gemm_counters.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-gemm_pmc}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 $R/tools/gemm_probe.py > $OUT/pmc1.log 2>&1 || { echo "pmc1 failed"; tail -5 $OUT/pmc1.log; exit 1; }
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY --output-format csv -d $OUT/pmc2 -- python3 $R/tools/gemm_probe.py > $OUT/pmc2.log 2>&1 || { echo "pmc2 failed"; tail -5 $OUT/pmc2.log; exit 1; }
echo done
