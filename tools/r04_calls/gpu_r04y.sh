#!/bin/bash
# call y: what loops of dependent MFMAs sustain (fp32 32x32x2, bf16 32x32x16) by waves per SIMD, chains, and vector work in between
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04y
timeout 120 ./tools/micro/mfma_peak | tee gpurun_out/r04y/mfma_peak.json
