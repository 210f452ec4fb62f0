#!/bin/bash
# Diagnostics: rebuild pg_spmv.hip with an ablation macro ON THE GPU BOX and time the bare kernel (outputs wrong on purpose).
for a in ${1:-1 2 3 4 0}; do
  echo "== PG_SPMV_ABLATE=$a"
  touch penguin/jl_amd/csrc/pg_spmv.hip
  PG_HIPCC_FLAGS="-DPG_SPMV_ABLATE=$a" python -m penguin.jl_amd.build 2>&1 | tail -1
  timeout -k 5 120 python scripts/spmv_time_only.py 512 2>&1 | tail -1
done
