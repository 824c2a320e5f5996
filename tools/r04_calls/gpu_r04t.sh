#!/bin/bash
# call t: evaluation top-k with the item table streamed from a tile-major copy (1 KB loads) and three register sets
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04t
timeout 500 python -m pytest tests -q -m gpu -x -k "eval or topk or predict or c_abi" 2>&1 | tail -4
for spec in "tiled3::3" "tiled2::2" "rowmajor3:tools/variants/eval_row_major.so:3" "rowmajor2:tools/variants/eval_row_major.so:2" "nocand3:tools/variants/eval_no_cand.so:3"; do
  IFS=: read tag lib bufs <<< "$spec"
  if [ -n "$lib" ]; then export SSLREC_HIP_LIBRARY="$PWD/$lib"; else unset SSLREC_HIP_LIBRARY; fi
  SSLREC_EVAL_BUFS=$bufs timeout 200 python tools/eval_variants.py $tag 2>&1 | tail -1 | tee -a gpurun_out/r04t/eval_variants.jsonl
done
