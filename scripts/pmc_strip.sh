#!/bin/bash
# FETCH_SIZE of the SpMV launches under variants of the work-item order: usage  scripts/pmc_strip.sh "ENV=VAL ..." tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in $1; do export $kv; done
tag=$2
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 scripts/dev_perf.py 512 4 > gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "$tag pass $i rc=$?"
done
python3 scripts/pmc_traffic_json.py gpurun_out/pmc_${tag} > gpurun_out/pmc_${tag}.json
