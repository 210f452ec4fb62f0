#!/bin/bash
# Round-3 profile of the headline command (run on the GPU box, writes under gpurun_out/; scripts/profile_summary.py turns
# it into profiles/r03_spmv_profile.json, which bench.py reads):
#   1. plain bench line
#   2. rocprofv3 --kernel-trace --stats of THE SAME command (per-kernel durations; the stats' average is what a reader
#      recomputes the roofline fraction from)
#   3. PMC passes, one counter set per run, kernel trace only (FETCH_SIZE, WRITE_SIZE, L2 hit / miss, L1 -> L2 requests, TA busy)
#      on the same loop (scripts/dev_perf.py 512 4: the bench's problem and time loop without the JSON bookkeeping)
# usage: scripts/profile_round3.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/prof_${tag}_bench_plain.json 2> gpurun_out/prof_${tag}_bench_plain.err
echo "plain bench rc=$?" > gpurun_out/prof_${tag}_progress.txt
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_${tag}_bench_under_rocprofv3.json 2> gpurun_out/prof_${tag}_bench_under_rocprofv3.err
echo "stats pass rc=$?" >> gpurun_out/prof_${tag}_progress.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 scripts/dev_perf.py 512 4 > gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "pmc pass $i ($set) rc=$?" >> gpurun_out/prof_${tag}_progress.txt
done
python3 scripts/profile_summary.py $tag > gpurun_out/prof_${tag}_summary.log 2>&1
echo "summary rc=$?" >> gpurun_out/prof_${tag}_progress.txt
cat gpurun_out/prof_${tag}_progress.txt
