set -e
O=gpurun_out/r6st10
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for clip in 0 8000; do
rocprofv3 --kernel-trace --output-format csv -d $O/tr_$clip -- python tools/probes/skew_trace.py arxiv $clip > $O/arxiv$clip.log 2>&1
python - $O/tr_$clip > $O/timeline_$clip.log <<'PY'
import csv, sys
from pathlib import Path
f = next(Path(sys.argv[1]).rglob("*kernel_trace.csv"))
rows = list(csv.DictReader(open(f)))
# the plain loop = 10 products before the scheduled loop's 11; print dispatches 60..100 from the end
rows = rows[-110:-60]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(f"{r['Kernel_Name'][:60]:60s} q{r.get('Queue_Id','?'):>2s} {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} -> {(int(r['End_Timestamp'])-t0)/1e3:9.1f}")
PY
grep -E "plain|scheduled" $O/arxiv$clip.log
done
