#!/bin/bash
n=${1:-3072}
slabs=${2:-5}
run() { echo "== $1"; shift; env "$@" timeout -k 10 240 python3 scripts/moving_bench.py $n $slabs BE 2>&1 | tail -c 900; echo; }
run "default (pool < 1 GiB, drain before plain hipMalloc)"
run "limit 64 MB, drain" PG_POOL_LIMIT_MB=64
run "limit 64 MB, NO drain (round 2's configuration)" PG_POOL_LIMIT_MB=64 PG_ALLOC_SYNC=4
run "limit 256 MB, NO drain" PG_POOL_LIMIT_MB=256 PG_ALLOC_SYNC=4
run "pool off" PG_ASYNC_ALLOC=-1
