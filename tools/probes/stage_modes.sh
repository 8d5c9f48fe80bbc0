# Developer probe (GPU box): the staged heavy / long-row kernel under its measurement overrides (MI_LONG_ROWS_FORM = staged | wave
# for the listed rows, MI_STAGE_COLS = 64 | 128 for the heavy slots); a kernel timeline of the arxiv-like matrix with one row
# beyond the long-row threshold.    bash tools/probes/stage_modes.sh [OUTDIR]
set -e
O=${1:-gpurun_out/stage_modes}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MI_AUTO_SCHEDULE=0
rocprofv3 --kernel-trace --output-format csv -d $O/tr_arxiv -- python tools/probes/skew_trace.py arxiv 0 > $O/arxiv0.log 2>&1
python tools/probes/ktimeline.py $O/tr_arxiv 6 > $O/arxiv0_timeline.log
rm -f $O/modes.log
for shape in "arxiv 0" "arxiv 8000" "c3skew 0" "products 0" "reddit 0"; do
  for mode in default staged wave; do
    case $mode in
      default) unset MI_LONG_ROWS_FORM;;
      *) export MI_LONG_ROWS_FORM=$mode;;
    esac
    echo "== $shape $mode" >> $O/modes.log
    python tools/probes/skew_trace.py $shape 2>&1 | tail -2 >> $O/modes.log
  done
done
cat $O/arxiv0_timeline.log $O/modes.log
