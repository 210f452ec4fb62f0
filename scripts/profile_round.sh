#!/bin/bash
# Round profile: kernel stats of the default bench + HBM traffic counters of the SpMV (separate PMC passes).
# usage: scripts/profile_round.sh <tag>     (writes under gpurun_out/; copy the summaries into profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_${tag}_bench.json 2> gpurun_out/prof_${tag}_bench.err
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcf_$tag -- python3 scripts/dev_perf.py 512 2 > gpurun_out/pmcf_$tag.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcw_$tag -- python3 scripts/dev_perf.py 512 2 > gpurun_out/pmcw_$tag.log 2>&1
echo "write pass done"
ls gpurun_out/prof_$tag gpurun_out/pmcf_$tag/* gpurun_out/pmcw_$tag/*
