#!/bin/bash
# One-off experiment behind DESIGN.md "stream-ordered pool": the moving solver's slabs at n^2 with the pool off / limited /
# unlimited / unlimited + poisoned allocations; every line carries a checksum of the last state (state_l2).
n=${1:-3072}
slabs=${2:-5}
run() { echo "== $1"; shift; env "$@" timeout -k 10 240 python3 scripts/moving_bench.py $n $slabs BE 2>&1 | tail -c 900; echo; }
run "pool off" PG_ASYNC_ALLOC=-1
run "pool, blocks < 64 MB (default)" PG_POOL_LIMIT_MB=64
run "pool, no limit" PG_POOL_LIMIT_MB=0
run "pool, no limit, poisoned allocations" PG_POOL_LIMIT_MB=0 PG_ALLOC_POISON=1
run "pool off, poisoned allocations" PG_ASYNC_ALLOC=-1 PG_ALLOC_POISON=1
