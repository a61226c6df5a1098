// Shared by the two translation units of the fused mono / stereo recurrence (smst_vocoder.hip: the tile form and its launchers;
// smst_vocoder_cont.hip: the continuous wavefront across the tiles of a call): the geometry of the line-aligned producers' LDS
// buffers, the ring constants, the cross-lane helpers of the recurrence wave.
#pragma once
#include "smst_recurrence.h"

namespace smst {

template <int CH, int L>
struct AlignGeom {
	static constexpr int RING = 32;                   // bins per (row, array): two lines
	static constexpr int ROWLEN = 2*CH*RING;          // float2 per row: CH input buffers, then CH previous-input buffers
	static constexpr int XLEN = CH*16;                // the row above the wave's first row: 16 bins per channel
	static constexpr int PER_PRODUCER = 8*ROWLEN + XLEN;
	static constexpr int LOADS = CH;                  // (4 rows x 2*CH arrays x 8 pieces) / 64 lanes
};

constexpr int kVocBlockSteps = 8, kVocBlocks = 3, kVocBlocksStaged = 2, kVocStagedProducers = 8, kVocOutBlocks = 4; // (kVocWaves: smst_recurrence.h)
constexpr int kVocOutBlocksAligned = 3; // lag 8: a row's 16-bin line lies in exactly two result blocks
// results ring: [block][step][channel][kVocOutPitch] -- 66, not 64: the writer reads a row's values of steps 2 apart in adjacent
// lane groups, and 2*CH*64 float2 is a multiple of the 32 banks (an 8-way conflict on every writer read with the first layout)
constexpr int kVocOutPitch = 66;

__device__ __forceinline__ float2 selectPair(bool pick, float2 a, float2 b) { return make_float2(pick ? a.x : b.x, pick ? a.y : b.y); }
__device__ __forceinline__ float2 fromLaneBelow(float2 v, float2 lane0) { // lane k receives lane k-1's v; lane 0 keeps its `lane0`
	// DPP wave_shr:1 without bound_ctrl: a lane with no source lane keeps the old value of the destination register
	return make_float2(__int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane0.x), __float_as_int(v.x), 0x138, 0xf, 0xf, false)),
	                   __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane0.y), __float_as_int(v.y), 0x138, 0xf, 0xf, false)));
}

} // namespace smst
