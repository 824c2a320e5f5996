// Stream format shared by the column-swept kernel (spmm_swept.hip) and the row-bundled kernel (spmm_bundle.hip): a wave's
// stream is a sequence of 64-dword blocks of S steps; dword g*LPG + j of a block is the entry of (step j, lane group g),
// so one coalesced dword load per array fetches S steps and a DPP permute hands every lane its lane group's entry.
#pragma once
#include "common.h"

// steps per 64-dword metadata block
// (= min(16, lanes per lane group); D = 16 / 8 are the feature-sliced widths of sslrec_amd/shard.py: a GPU holds d / P columns)
template <int D> struct SweptFmt { static constexpr int S = (D == 8) ? 2 : (D == 16) ? 4 : (D == 32) ? 8 : 16; };

// entry of step J of the block for THIS lane's lane group, from the wave's coalesced dword V (see the layout above)
template <int D, int J>
__device__ __forceinline__ int sw_bcast(int v) {
    if constexpr (D == 16) {      // lane groups of 4 = DPP quads: quad_perm [J, J, J, J]
        return __builtin_amdgcn_update_dpp(0, v, J * 0x55, 0xF, 0xF, false);
    } else if constexpr (D == 8) {       // two lane groups of 2 per quad: quad_perm [J, J, 2 + J, 2 + J]
        return __builtin_amdgcn_update_dpp(0, v, J | (J << 2) | ((2 + J) << 4) | ((2 + J) << 6), 0xF, 0xF, false);
    } else if constexpr (D == 32) {      // a 16-lane row holds two lane groups of 8: lanes 0-7 take lane J, lanes 8-15 lane 8+J
        const int lo = __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xF, 0x3, false);
        return __builtin_amdgcn_update_dpp(lo, v, 0x150 + 8 + J, 0xF, 0xC, false);
    } else {
        return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xF, 0xF, false);      // row_newbcast:J
    }
}

