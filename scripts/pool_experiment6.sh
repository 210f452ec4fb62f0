#!/bin/bash
# Allocator check of the moving path (DESIGN.md "device memory"): the same slabs with the library's block cache (default
# inside the moving path's entry points), without it, with poisoned allocations, and with the cache limited to small blocks
# (a mixed run); state_l2 is a checksum of the last state and must not depend on the allocator.
n=${1:-3072}
slabs=${2:-5}
run() { echo "== $1"; shift; env "$@" timeout -k 10 240 python3 scripts/moving_bench.py $n $slabs BE 2>&1 | tail -c 900; echo; }
run "block cache (default)"
run "cache off" PG_ASYNC_ALLOC=-1
run "block cache, poisoned allocations" PG_ALLOC_POISON=1
run "block cache for blocks < 64 MB only" PG_POOL_LIMIT_MB=64
run "block cache everywhere" PG_ASYNC_ALLOC=1
