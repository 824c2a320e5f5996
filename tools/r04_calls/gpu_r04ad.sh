#!/bin/bash
# call ad: the shared backward chain of SimGCL's views in every form (fused first layer, loop form, row-sharded, feature-sliced)
cd "$GRAFT_REPO_ROOT"
timeout 700 python -m pytest tests -q -m gpu -x -k "simgcl or views or sliced or sharded" 2>&1 | tail -4
