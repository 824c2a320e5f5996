#!/bin/bash
# call ah: the lighter two thirds of the GPU suite on the round's last commit (what the remaining budget allows)
cd "$GRAFT_REPO_ROOT"
timeout 140 python -m pytest tests -q -m gpu -x -k "not amazon and not yelp and not traj and not cfg5 and not bench and not config_lines and not rccl and not full_size" 2>&1 | tail -3
