#!/bin/bash
# Memory-side counters of the slice SpMV launched back to back by scripts/spmv_probe.py (mode 0; the last launches are the
# "cold" ones: rotating vectors).  Separate passes, kernel-trace only.  usage: pmc_probe.sh <tag> [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
for kv in "$@"; do export "$kv"; done
out=gpurun_out/pmcp_$tag
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d ${out}_$i -- python3 scripts/spmv_probe.py 512 12 0 > ${out}_$i.log 2>&1
  echo "pass $i rc=$?" >> gpurun_out/pmcp_${tag}_progress.txt
done
python3 - "$out" <<'PY' > gpurun_out/pmcp_$tag.txt 2>&1
import csv, glob, collections, sys
out = sys.argv[1]
for i in (1, 2, 3, 4):
    for f in sorted(glob.glob(f"{out}_{i}/*/*counter_collection.csv")):
        rows = [r for r in csv.DictReader(open(f)) if "k_spmv_s" in r["Kernel_Name"]]
        by = collections.defaultdict(list)
        for r in rows:
            by[r["Counter_Name"]].append((int(r["Start_Timestamp"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for c, v in by.items():
            v.sort()
            cold = v[-10:]          # the last launches: rotating vectors
            warm = v[-25:-16]       # before them: the same vectors every launch
            med = lambda a: sorted(a)[len(a) // 2]
            print(f"{c:38s} cold {med([x[1] for x in cold]):14.0f} (dur {med([x[2] for x in cold]) / 1e3:6.1f} us)   warm {med([x[1] for x in warm]):14.0f} (dur {med([x[2] for x in warm]) / 1e3:6.1f} us)  n={len(v)}")
PY
cat gpurun_out/pmcp_$tag.txt
