#!/bin/bash
# round 6, call m: the default bench line on the round's last code, with its wall time
O=gpurun_out/r06m; mkdir -p $O
S=$(date +%s); timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc $? wall $(( $(date +%s) - S )) s"; cut -c1-300 $O/bench_line.json
