#!/bin/bash
# per-kernel time of the bench loop with the extrapolated start (rocprofv3 kernel trace + stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-gs}
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o $tag -- python3 bench.py --steps ${2:-20} --warmup 5 --no-cpu-baseline > gpurun_out/$tag.json 2> gpurun_out/$tag.err
echo "rc=$?"
f=$(ls gpurun_out/$tag/*/${tag}_kernel_stats.csv gpurun_out/$tag/${tag}_kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    if float(r["AverageNs"]) > 1e6: continue
    print("   %-60s calls %6s  avg %9.1f us  total %9.1f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
