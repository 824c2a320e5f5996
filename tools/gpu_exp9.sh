#!/bin/bash
OUT=gpurun_out/r01j; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --tb=short -k "spmm or propagate or tiny or amazon or yelp or revalued" > $OUT/test.log 2>&1; echo "== pytest exit $?"; tail -3 $OUT/test.log
for w in 4096 5120 8192; do for f in 0 4096; do SSLREC_SPMM_STREAMS=$w python tools/spmm_sweep.py --only amazon-book --order degree --fold $f 2>&1 | grep graph | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($w, d['fold'], round(d['us'],1), round(d['gather_GBs']))"; done; done
for d in 32 128 256; do SSLREC_SPMM_STREAMS=5120 python tools/spmm_sweep.py --only amazon-book --order degree --d $d 2>&1 | grep graph | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('d', $d, round(d['us'],1), round(d['gather_GBs']))"; done
SSLREC_SPMM_STREAMS=5120 python tools/spmm_sweep.py --only yelp --order degree 2>&1 | grep graph | cut -c100-300
