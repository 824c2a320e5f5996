#!/bin/bash
# call h: evaluation top-k on fp16-plane score tiles (h3), the hardened reductions' stress test, the bundled-compact test
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "eval or topk or predict or metric or round5 or sharded_path or trainer or epoch" 2>&1 | tail -16 > $O/pytest_tail.txt; tail -8 $O/pytest_tail.txt
for prec in h3 fp32; do
  SSLREC_EVAL_PRECISION=$prec timeout 300 python tools/eval_profile.py > $O/eval_$prec.json 2> $O/eval_$prec.err || echo "eval $prec failed"
  tail -c 1200 $O/eval_$prec.json; echo
done
