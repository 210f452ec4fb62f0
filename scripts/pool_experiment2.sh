#!/bin/bash
n=${1:-3072}
slabs=${2:-5}
run() { echo "== $1"; shift; env "$@" timeout -k 10 240 python3 scripts/moving_bench.py $n $slabs BE 2>&1 | tail -c 900; echo; }
run "mixed (limit 64)" PG_POOL_LIMIT_MB=64
run "mixed (limit 64) + poison" PG_POOL_LIMIT_MB=64 PG_ALLOC_POISON=1
run "mixed (limit 64) + device sync before hipFree" PG_POOL_LIMIT_MB=64 PG_ALLOC_SYNC=1
run "mixed (limit 64) + stream sync before hipMalloc" PG_POOL_LIMIT_MB=64 PG_ALLOC_SYNC=2
run "mixed (limit 64) + both" PG_POOL_LIMIT_MB=64 PG_ALLOC_SYNC=3
run "mixed (limit 16)" PG_POOL_LIMIT_MB=16
run "mixed (limit 256)" PG_POOL_LIMIT_MB=256
