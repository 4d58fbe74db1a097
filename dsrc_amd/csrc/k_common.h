// Wave / workgroup primitives shared by every kernel of the block-compression path.
// gfx950: 64-lane wavefronts; a workgroup of DSRC_WG threads owns one FASTQ block.
#pragma once
#include <hip/hip_runtime.h>
#include "dsrc_types.h"

#define WG DSRC_WG
#define WAVES DSRC_WAVES

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ u32 wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ u32 min_u32(u32 a, u32 b) { return a < b ? a : b; }
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// acc + the number of lanes below this one whose bit is set in a wave-uniform mask (v_mbcnt_lo / v_mbcnt_hi: the mask stays in
// scalar registers -- a prefix sum over one-bit values without the six dependent DPP steps of a scan)
__device__ __forceinline__ u32 wave_count_below(u64 mask, u32 acc)
{
#ifdef DSRC_EMU_BUILD
	return acc + (u32)__builtin_popcountll(mask & lanemask_lt());
#else
	return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, acc));
#endif
}
// this lane's bit of a wave-uniform mask as a condition: the mask becomes the execution mask, no vector instruction
__device__ __forceinline__ bool wave_lane_in(u64 mask)
{
#ifdef DSRC_EMU_BUILD
	return ((mask >> lane_id()) & 1ull) != 0;
#else
	return __builtin_amdgcn_inverse_ballot_w64(mask);
#endif
}

__device__ __forceinline__ u32 wave_incl_scan(u32 v)
{
	const u32 l = lane_id();
	for (u32 d = 1; d < 64; d <<= 1)
	{
		u32 t = __shfl_up(v, d);
		if (l >= d) v += t;
	}
	return v;
}

// the same with DPP row shifts / row broadcasts (gfx9: v_add with a dpp modifier, ~2 cycles a step) instead of six
// ds_bpermute round trips through the LDS crossbar
__device__ __forceinline__ u32 wave_incl_scan_dpp(u32 v)
{
#ifdef DSRC_EMU_BUILD
	return wave_incl_scan(v);
#else
	int x = (int)v;
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);      // row_shr:1
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);      // row_shr:2
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);      // row_shr:4
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);      // row_shr:8
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
	return (u32)x;
#endif
}

// lane i takes lane i - 1's value, lane 0 `first`: one DPP move (gfx9 wave_shr:1; lanes without a source keep the old value)
// instead of a ds_bpermute round trip
__device__ __forceinline__ u32 wave_shr1(u32 v, u32 first)
{
#ifdef DSRC_EMU_BUILD
	const u32 t = __shfl_up(v, 1);
	return lane_id() == 0 ? first : t;
#else
	return (u32)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xf, 0xf, false);
#endif
}

__device__ __forceinline__ u32 wave_max(u32 v)
{
	for (u32 d = 32; d >= 1; d >>= 1) { u32 t = __shfl_xor(v, (int)d); v = t > v ? t : v; }
	return v;
}

// Pointers into LDS keep their address space so that LDS-DMA destinations are plain 32-bit constants
#define LDS_AS __attribute__((address_space(3)))
#define GLOBAL_AS __attribute__((address_space(1)))

// LDS DMA (gfx950 global_load_lds_dwordx4): every active lane fetches 16 bytes from sbase (wave-uniform) + voff
// (its own 32-bit byte offset) and they land at lds_dst (wave-uniform) + 16 * lane without passing through registers
__device__ __forceinline__ void lds_dma16(const u8* sbase, u32 voff, LDS_AS void* lds_dst)
{
	__builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(sbase + voff), lds_dst, 16, 0, 0);
}

__device__ __forceinline__ const u8* uniform_ptr(const void* p)
{
	const u64 v = (u64)p;
	return (const u8*)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)v));
}

// orders one wave's LDS traffic in program order (no instruction: the lanes of a wave run in lock-step and the LDS
// executes a wave's instructions in order)
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// every LDS-DMA request this wave has issued so far has landed: s_waitcnt vmcnt(0) (gfx9 encoding: vmcnt in bits
// 3:0 and 15:14, expcnt/lgkmcnt left at their maxima), followed by a wave-scope fence (no instruction; it keeps the
// compiler from moving LDS reads above the wait).  The compiler's own tracking of LDS DMA is not relied upon.
__device__ __forceinline__ void lds_dma_wait()
{
	__builtin_amdgcn_s_waitcnt(0x0F70);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ u32 wave_sum(u32 v)
{
	for (u32 d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, (int)d);
	return v;
}

// exclusive scan over the workgroup (must be called by every thread); *total = sum of all v
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32* total)
{
	__shared__ u32 s_w[16];          // up to 1024 threads whatever DSRC_WG is (k_sort may use a larger workgroup than the other kernels)
	const u32 inc = wave_incl_scan(v);
	if (lane_id() == 63) s_w[wave_id()] = inc;
	__syncthreads();
	u32 base = 0, tot = 0;
	for (u32 i = 0; i < (blockDim.x >> 6); ++i)
	{
		const u32 x = s_w[i];
		if (i < wave_id()) base += x;
		tot += x;
	}
	__syncthreads();
	*total = tot;
	return base + inc - v;
}

// ---- compressed-stream staging ---------------------------------------------------------
// A stream is staged as u32 words holding the MSB-first bit stream "logically big-endian":
// stream byte k lives at ((u8*)words)[k ^ 3]; k_assemble undoes that when it copies the
// four streams of a block into place.  Bit writers OR into zero-initialised words.
__device__ __forceinline__ void put_bits(u32* words, u64 bitpos, u32 code, u32 len)
{
	if (len == 0) return;
	if (len < 32) code &= (1u << len) - 1u;
	const u64 w = bitpos >> 5;
	const u32 sh = (u32)bitpos & 31u;
	const u64 v = (u64)code << (64u - len - sh);
	atomicOr(&words[w], (u32)(v >> 32));
	const u32 lo = (u32)v;
	if (lo) atomicOr(&words[w + 1], lo);
}

__device__ __forceinline__ void put_byte(u32* words, u64 k, u32 v) { ((u8*)words)[k ^ 3ull] = (u8)v; }
__device__ __forceinline__ void put_be32(u32* words, u64 k, u32 v)
{
	put_byte(words, k, v >> 24); put_byte(words, k + 1, v >> 16); put_byte(words, k + 2, v >> 8); put_byte(words, k + 3, v);
}

// core::bit_length (src/utils.h:181-189)
__device__ __forceinline__ u32 bit_length32(u64 x)
{
	for (u32 i = 0; i < 32; ++i)
		if (x < (1ull << i)) return i;
	return 64;
}

__device__ __forceinline__ u32 ilog2_floor(u32 x) { u32 r = 0; while (x > 1) { x >>= 1; ++r; } return r; }
