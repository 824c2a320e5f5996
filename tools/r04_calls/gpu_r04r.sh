#!/bin/bash
# call r: what the event records of the roofline measurement cost the eager step
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 240 python tools/eager_overhead.py 2>&1 | tail -12
