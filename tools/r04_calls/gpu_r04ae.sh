#!/bin/bash
# call ae: reg_loss held to the exact (fp64) sum at 2e-6 in the whole-step tests
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests -q -m gpu -x -k "training_step_matches_reference or lightgcl_step_matches or amazon_book_size_matches" 2>&1 | tail -4
