set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/logs
cd $R
timeout -k 10 500 python tools/bench_attn_csr.py > gpurun_out/logs/attention_csr.log 2>&1 || echo "attn_csr failed"
: > gpurun_out/logs/attention_csr_bwd.log
for S in 512 1024 2048; do
  MI_SEQ=$S MI_KEPT=0.25,0.1,0.05 timeout -k 10 300 python tools/bench_attn_csr_bwd.py >> gpurun_out/logs/attention_csr_bwd.log 2>&1 || echo "bwd $S failed"
done
timeout -k 10 300 python tools/probes/ldsb_timing.py > gpurun_out/logs/ldsb_phase_stamps.log 2>&1 || echo "stamps failed"
tail -3 gpurun_out/logs/attention_csr.log; grep 'kept\|# tools' gpurun_out/logs/attention_csr_bwd.log
