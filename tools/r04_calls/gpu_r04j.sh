#!/bin/bash
O=gpurun_out/r04j; mkdir -p $O
R=$(pwd)
timeout 100 python tools/eval_variants.py two-chains | tee -a $O/eval_variants.jsonl
SSLREC_HIP_LIBRARY=$R/tools/variants/libsslrec_nocand.so timeout 100 python tools/eval_variants.py two-chains-no-candidates | tee -a $O/eval_variants.jsonl
timeout 200 python -m pytest tests -x -q -m gpu -k "evaluation or topk" 2>&1 | tail -2
