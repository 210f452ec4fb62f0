#!/bin/bash
n=${1:-4096}
slabs=${2:-3}
run() { echo "== $1"; shift; env "$@" timeout -k 10 240 python3 scripts/moving_bench.py $n $slabs BE 2>&1 | tail -c 1500; echo; }
run "all pool, live-block overlap check" PG_ALLOC_SYNC=32
run "limit 200 + drain, overlap check" PG_POOL_LIMIT_MB=200 PG_ALLOC_SYNC=32
