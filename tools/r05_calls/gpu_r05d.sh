#!/bin/bash
# call d: the fp16 two-plane InfoNCE mode (h3): parity tests in every mode, errors / times against fp64, the SGL / SimGCL golden steps and
# trajectories with h3 as the process default; durations of the eight-process tests with the host threads divided among the ranks
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05d; mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu -k "infonce" -x 2>&1 | tail -15 > $O/pytest_infonce.txt; tail -4 $O/pytest_infonce.txt
INFONCE_MODES=x6,h3,fp32 timeout 300 python tools/infonce_modes.py $O/infonce_modes.json 2>&1 | tail -8
SSLREC_INFONCE_PRECISION=h3 timeout 900 python -m pytest tests -q -m gpu -k "golden or traj or whole_training_step or simgcl or sgl or smoke" 2>&1 | tail -25 > $O/pytest_h3_default.txt; tail -12 $O/pytest_h3_default.txt
timeout 900 python -m pytest tests -q -m gpu --durations=8 -k "(two_ranks_on_one_gpu and 8) or (feature_sliced_ranks_on_one_gpu and 8) or (feature_sliced_lightgcl_ranks_on_one_gpu and 8-)" 2>&1 | tail -20 > $O/pytest_world8.txt; tail -14 $O/pytest_world8.txt
