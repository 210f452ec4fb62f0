#!/bin/bash
n=${1:-4096}
slabs=${2:-3}
run() { echo "== $1"; shift; env "$@" timeout -k 10 240 python3 scripts/moving_bench.py $n $slabs BE 2>&1 | tail -c 900; echo; }
run "pool off" PG_ASYNC_ALLOC=-1
run "limit 200 + drain" PG_POOL_LIMIT_MB=200
run "limit 300 + drain" PG_POOL_LIMIT_MB=300
run "all pool (limit 1024)"
run "all pool + poison" PG_ALLOC_POISON=1
run "all pool + stream sync before every hipMallocAsync" PG_ALLOC_SYNC=8
run "all pool + stream sync before every hipFreeAsync" PG_ALLOC_SYNC=16
run "all pool again"
