#!/bin/bash
# Round-2 profile: kernel stats of the default bench + HBM-side traffic of the SpMV launches (separate PMC passes, kernel
# trace only) + L2 / L1 / TA counters.  usage: scripts/profile_round2.sh <tag>  (writes under gpurun_out/; the summaries
# that are judged are copied into profiles/ by hand)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
python3 bench.py --steps 20 --warmup 3 > gpurun_out/prof_${tag}_bench_plain.json 2> gpurun_out/prof_${tag}_bench_plain.err
echo "plain bench done" >> gpurun_out/prof_${tag}_progress.txt
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_${tag}_bench.json 2> gpurun_out/prof_${tag}_bench.err
echo "stats pass rc=$?" >> gpurun_out/prof_${tag}_progress.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 scripts/dev_perf.py 512 4 > gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "pmc pass $i ($set) rc=$?" >> gpurun_out/prof_${tag}_progress.txt
done
python3 scripts/pmc_traffic_json.py gpurun_out/pmc_${tag} $(ls gpurun_out/prof_$tag/*/*kernel_trace.csv gpurun_out/prof_$tag/*kernel_trace.csv 2>/dev/null | head -1) > gpurun_out/prof_${tag}_traffic.json 2> gpurun_out/prof_${tag}_traffic.err
cat gpurun_out/prof_${tag}_progress.txt
cat gpurun_out/prof_${tag}_traffic.json
