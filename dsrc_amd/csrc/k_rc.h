// Order-k context modelling + range coding (the 83 % of the reference's CPU time at -d3 -q2):
//   TDnaRCOrderModeler            src/DnaModelerRCO.h:27-132
//   TQualityModelExt / encoders   src/QualityEncoder.h:24-367
//   TSymbolCoderRC                src/SymbolCoderRC.h:23-93
//   RangeEncoder                  src/RangeCoder.h:51-84
//
// The reference walks one 3.3 M-symbol chain per stream: table row -> (freq, cum, total) ->
// range update.  Context ids depend on the INPUT symbols only, never on coder state, so the
// chain is cut in three data-parallel stages and one short serial one:
//   k_sort    : context id of every symbol (pure function of <= order+2 previous symbols), computed on the fly by
//               the first pass of a stable LSD radix sort of (ctx, sym, t) by ctx -> each context's history is contiguous
//   k_replay  : one lane replays one context's history on a private counter row in LDS and
//               emits (total, cum, freq) for every symbol, scattered back to stream order
//   k_rc      : the integer range-coder recurrence, one LANE per stream (RC_LANES = 32 streams per wave); each
//               stream's records are contiguous, the wave pulls them through LDS 768 B at a time (LDS DMA)
// No adaptive table ever exists in HBM (the reference clears 2-64 MiB per block).
#pragma once
#include "k_common.h"

#define ELEM_T_BITS 32
#define ELEM_SYM_SHIFT 32
#define ELEM_CTX_SHIFT 40

// what the range coder consumes per symbol (written by k_replay in stream order, one contiguous array per chain)
struct RcRec { u32 m_lo, mf, cum; };     // what the coder works on (LDS rows): m = ceil(2^48/total): m_lo = m & 0xFFFFFFFF, mf = (m >> 32) << 16 | freq
typedef u64 RcPack;                      // what k_model leaves per symbol, 8 aligned bytes: freq | cum << 16 | total << 32 (total < 2^16) | (t mod 8192) << 48;
                                         // also the unit the record arrays are carved in (8 bytes of address space per symbol)

// ---- what k_rc reads: six bytes per symbol (round 6) --------------------------------------------------------------------------
// 64 consecutive records of a stream are one 384-byte CHUNK: 64 dwords freq | cum << 16, then 64 u16 totals -- a loader wave of
// k_rc takes a chunk with one dword and one ushort load per lane.  A stream's chunks lie back to back in its array (the ONLY thing of a
// stream that stays in HBM until the range coder has run: 40 MB per 8 MiB block at -d3 -q2, 53 MB with the 8-byte records of rounds
// 1-5).  k_model's 8-byte records (48 bits + t mod 8192) live in the element slice, one launch group at a time; k_place puts a time
// bin of them into stream order and writes its 128 chunks.  k_replay (and k_model without time bins) write chunks directly.
#define RC6_TB 13
#define RC6_CHUNK_BYTES 384u
__device__ __forceinline__ u32 rc6_chunk_off(u32 chunk) { return chunk * RC6_CHUNK_BYTES; }
__device__ __forceinline__ void rc6_store(RcPack* chain, u32 t, u32 f, u32 cum, u32 tot)
{
	u8* p = (u8*)chain + rc6_chunk_off(t >> 6);
	((u32*)p)[t & 63u] = f | (cum << 16);
	((u16*)(p + 256))[t & 63u] = (u16)tot;
}
__device__ __forceinline__ RcPack rc6_load(const RcPack* chain, u32 t)      // freq | cum << 16 | total << 32
{
	const u8* p = (const u8*)chain + rc6_chunk_off(t >> 6);
	return (u64)((const u32*)p)[t & 63u] | ((u64)((const u16*)(p + 256))[t & 63u] << 32);
}

struct CtxJob     // one (block, stream)
{
	u64 src_off;        // byte offset of the symbol stream (q_stream / d_stream)
	u64 elems;          // u64 index of sort buffer A
	u64 elems_b;        // u64 index of sort buffer B
	u64 trip;           // RcPack index of this chain's first record
	u32 n;              // symbols
	u32 blk;
	u32 alpha_bits;     // log2(alphabet)
	u32 order;          // symbol order (DNA: Order; quality: SymbolOrder)
	u32 rescale_shift;  // quality: pctx = qp >> rescale_shift
	u32 translate;      // quality: 1 = dense rank via q_sym (lossless), 0 = raw value (lossy)
	u32 key_bits;
	u32 passes, dbits;
	u32 sorted_in_b;    // where the sorted elements end up
	u32 out_byte0;      // first byte of the range-coder output inside the staging stream
	u32 out_cap;        // bytes
	u64 out_words;      // u32 index of the staging stream
	u32 is_dna;
	u32 scheme;         // scheme byte of the stream prologue
	u32 n_alpha;        // alphabet size (replay template selector)
	u32 qlen;           // quality: read length if every read of the block has the same one (then qp_stream does not exist), else 0
	u32 qm_lo, qm_hi;   // ceil(2^48 / qlen) for exact_div
	u32 jid;            // index of the job in the batch's job array (per-job words of the `bk` pool, fallback lists)
	// bucketed path (k_bucket.h); bk_on = 0: the stream goes through k_sort / k_replay
	u32 bk_on, bk_binned;
	u32 bk_hb, bk_lb;   // bucket digit bits / key bits sorted in LDS (bk_hb + bk_lb = key_bits)
	u32 bk_mul, bk_kmask; // key = (ctx * bk_mul) & bk_kmask
	u32 bk_big;         // a bucket of at least this many elements looks for its windows inside one tile's run first (BK_BIG; tests lower it)
	u32 bk_fb;          // bk index of the fallback list of the stream's launch group: count, then job ids
	u32 bk_cnt;         // index (u16 units) of the stream's per-tile bucket counts / per-bin bucket offsets (k_part, k_binoff)
	u32 bk_limit;       // largest bucket a wave of k_model may walk (BK_LIMIT; tests lower it)
	u32 bk_narrow_bins; // streams of up to this many tiles keep their tile table in k_model's LDS (BK_MAX_BINS; tests lower it)
};

typedef u64 __attribute__((aligned(1))) u64_unaligned;
typedef u32 __attribute__((aligned(1))) u32_unaligned;

// floor(n / d) for n < 2^32, d <= 2^16 with m = ceil(2^48 / d) (same identity as rc_div)
__device__ __forceinline__ u32 exact_div(u32 n, u32 m_lo, u32 m_hi)
{
	const u64 p = (u64)n * m_hi + __umulhi(n, m_lo);
	return (u32)(p >> 16);
}

// position context of quality symbol t: floor(j * 128 / len) >> shift, j = position inside the read.  With reads of one
// length j = t mod len is a closed form of t and no qp_stream is needed; otherwise k_prep_write stored floor(j*128/len).
__device__ __forceinline__ u32 qua_pctx(const CtxJob& j, const u8* qp, u32 t)
{
	if (j.qlen == 0) return (u32)qp[t] >> j.rescale_shift;
	const u32 pos = t - exact_div(t, j.qm_lo, j.qm_hi) * j.qlen;
	return exact_div(pos * 128u, j.qm_lo, j.qm_hi) >> j.rescale_shift;
}

// ---- DNA context: hash of the previous `order` symbols, carried across records --------------
// (TDnaRCOrderModeler::UpdateHash, src/DnaModelerRCO.h:121-131).  Element of symbol t:
// ctx << 40 | sym << 32 | t.  Symbols before the start of the stream do not enter the hash.

// eight 2-bit symbols, one per byte (byte i -> bits 2i..2i+1)
__device__ __forceinline__ u32 pack2x8(u64 x)
{
	x &= 0x0303030303030303ull;
	x = (x | (x >> 6)) & 0x000F000F000F000Full;
	x = (x | (x >> 12)) & 0x000000FF000000FFull;
	return (u32)(x | (x >> 24)) & 0xFFFFu;
}

// hash of the symbols before t (newest in the lowest bits) and the symbol itself; *sym is the raw stream byte
__device__ __forceinline__ u32 dna_hash(const CtxJob& j, const u8* s, u32 t, u32* sym)
{
	const u32 ab = j.alpha_bits, order = j.order;
	u32 h = 0;
	if (t >= 15 && ab == 2 && order <= 15)
	{	// s[t-15 .. t] in two unaligned 8-byte loads; byte-reversed so that s[t-1] comes first
		const u64 hi = *(const u64_unaligned*)(s + t - 7), lo = *(const u64_unaligned*)(s + t - 15);
		*sym = (u32)(hi >> 56);
		h = pack2x8(__builtin_bswap64(hi) >> 8);                 // s[t-1] .. s[t-7] -> bits 0..13
		if (order > 7) h |= pack2x8(__builtin_bswap64(lo)) << 14;  // s[t-8] .. s[t-15] -> bits 14..29
	}
	else
	{
		const u32 k0 = t < order ? t : order;
		for (u32 k = k0; k >= 1; --k) h = (h << ab) | s[t - k];
		*sym = s[t];
	}
	return h & (u32)((1ull << (ab * order)) - 1ull);
}

__device__ __forceinline__ u64 ctx_elem_dna(const CtxJob& j, const u8* s, u32 t, bool* bad)
{
	u32 sym;
	const u32 h = dna_hash(j, s, t, &sym);
	const u32 n_alpha = 1u << j.alpha_bits;
	if (sym >= n_alpha) *bad = true;                        // reference UB
	return ((u64)h << ELEM_CTX_SHIFT) | ((u64)(sym & (n_alpha - 1)) << ELEM_SYM_SHIFT) | t;
}

// ---- quality context (TQualityModelBase::UpdateHash, src/QualityEncoder.h:77-94) -------------
// Before coding symbol t the hash slots are: k < order/2 : raw s[t-1-k];
// k >= order/2 : floor((s[t-1-k] + s[t-2-k]) / 2)  (order 1: slot 0 is raw).  Symbols before the
// start of the block read as 0.  ctx = (slots << alpha_bits) | position_context  (<= 21 bits).
// rank: 256-entry LDS table (dense rank of a raw quality value, or identity for the lossy model).
__device__ __forceinline__ u64 ctx_elem_qua(const CtxJob& j, const u8* s, const u8* qp, const u8* rank, u32 t)
{
	const u32 ab = j.alpha_bits, order = j.order, half = order / 2;
	u32 v[8];                                               // v[k] = rank of s[t-1-k], k <= order (<= 6)
	u32 cur;
	if (t >= 7)
	{
		const u64 w = *(const u64_unaligned*)(s + t - 7);     // s[t-7 .. t]
		cur = (u32)(w >> 56);
#pragma unroll
		for (u32 k = 0; k < 7; ++k) v[k] = k <= order ? rank[(u32)(w >> (8 * (6 - k))) & 0xFFu] : 0;
	}
	else
	{
		cur = s[t];
#pragma unroll
		for (u32 k = 0; k < 7; ++k) v[k] = (k <= order && t >= k + 1) ? rank[s[t - 1 - k]] : 0;
	}
	v[7] = 0;
	u32 h = 0;
#pragma unroll
	for (u32 k = 6; k >= 1; --k)
		if (k <= order)
		{
			const u32 slot = k - 1;
			const u32 x = (slot < half || order == 1) ? v[slot] : ((v[slot] + v[slot + 1]) >> 1);
			h = (h << ab) | x;
		}
	const u32 pctx = qua_pctx(j, qp, t);
	const u32 ctx = (h << ab) | pctx;
	const u32 sym = rank[cur] & ((1u << ab) - 1u);
	return ((u64)ctx << ELEM_CTX_SHIFT) | ((u64)sym << ELEM_SYM_SHIFT) | t;
}

// lowest `dbits` bits of the context of symbol t (the first sort digit) without building the element
__device__ __forceinline__ u32 ctx_digit0(const CtxJob& j, const u8* s, const u8* qp, const u8* rank, u32 t, u32 dmask, bool* bad)
{
	if (j.is_dna)
	{
		u32 sym;
		const u32 h = dna_hash(j, s, t, &sym);
		if (sym >= (1u << j.alpha_bits)) *bad = true;
		return h & dmask;
	}
	if (2 * j.alpha_bits >= j.dbits)
	{	// position context + slot 0, which is always the raw previous symbol
		const u32 v0 = t >= 1 ? rank[s[t - 1]] : 0u;
		return ((v0 << j.alpha_bits) | qua_pctx(j, qp, t)) & dmask;
	}
	return (u32)(ctx_elem_qua(j, s, qp, rank, t) >> ELEM_CTX_SHIFT) & dmask;
}

// ---- stable LSD radix sort by ctx; one workgroup owns one stream ------------------------------
// Pass 0 computes the elements from the symbol stream on the fly (no context array is ever written); every pass
// builds the histogram of the NEXT digit while it scatters, so a pass reads its input exactly once.
// Per pass: exclusive scan of the digit histogram -> tiles of WG*SORT_ITEMS elements.  A wave owns a contiguous
// run of 64*SORT_ITEMS elements of the tile and walks it 64 at a time: equal-digit lanes are found with one
// ballot per digit bit, the wave's private counter row in LDS gives the running rank, and a per-digit
// scan over the waves turns the rows into global offsets once per tile.
#ifndef SORT_DIGIT_BITS
#define SORT_DIGIT_BITS 10     // 20-bit quality contexts (32 symbols, order 3) in two passes instead of three: k_sort 73.5 -> 64.7 ms per 512 blocks
#endif
#define SORT_MAX_BINS (1 << SORT_DIGIT_BITS)
#ifndef SORT_ITEMS
#define SORT_ITEMS 8
#endif
#ifndef SORT_WG
#define SORT_WG WG
#endif
#define SORT_WAVES (SORT_WG / 64)
#ifndef SORT_OCC
#define SORT_OCC 4          // waves per SIMD the register allocation aims at
#endif
// PROBE != 0: timing experiments only (tools/variant_bench.sh, -DDSRC_SORT_PROBE), the output is not a sort.  Every probe writes
// tile position p to dst[tile + p]; 1 no digit-0 histogram, 2 no ranking, 4 no per-tile scan, 8 nothing leaves the registers,
// 16 no global store, 32 elements made up instead of loaded / computed, 64 no next-digit histogram
template <u32 PROBE, bool ATOMIC>
__global__ void __launch_bounds__(SORT_WG) __attribute__((amdgpu_waves_per_eu(SORT_OCC, SORT_OCC))) k_sort(const CtxJob* jobs, u64* pool, const u8* d_stream, const u8* q_stream, const u8* qp_stream, BlkState* st, const u32* fb)
{	// fb == nullptr: workgroup i sorts jobs[i].  Otherwise fb = {count, job ids ...} (the streams k_part handed back, k_bucket.h) and
	// the workgroups of the launch share the list.
	__shared__ u32 s_base[SORT_MAX_BINS];                 // where the next element of a digit goes (index into dst)
	__shared__ u32 s_next[SORT_MAX_BINS];
	__shared__ u32 s_delta[SORT_MAX_BINS];                // tile position p of an element with digit d -> dst index s_delta[d] + p
	__shared__ u16 s_cnt[SORT_WAVES][SORT_MAX_BINS];      // per tile a wave ranks 64*SORT_ITEMS elements: 16 bits are plenty
	__shared__ u16 s_off[SORT_WAVES][SORT_MAX_BINS];      // tile position of a wave's first element of a digit
	__shared__ u64 s_tile[SORT_WG * SORT_ITEMS];          // the tile in digit order.  102 KB in all: one workgroup per CU, next to a k_rc workgroup's 53 KB
	__shared__ u32 s_ws[SORT_WAVES];
	__shared__ u8 s_rank[256];
	const u32 n_f = fb ? fb[0] : gridDim.x;
	for (u32 f_i = blockIdx.x; f_i < n_f; f_i += gridDim.x)
	{
	const CtxJob j = jobs[fb ? fb[1 + f_i] : f_i];
	const u32 n = j.n, bins = 1u << j.dbits;
	const u32 wv = wave_id(), lane = lane_id();
	constexpr u32 nw = SORT_WAVES;                           // always launched with SORT_WG threads
	constexpr u32 tile_elems = SORT_WG * SORT_ITEMS;
	const u8* sym_src = (j.is_dna ? d_stream : q_stream) + j.src_off;
	const u8* qp = qp_stream + j.src_off;

	for (u32 i = threadIdx.x; i < 256; i += blockDim.x) s_rank[i] = (!j.is_dna && j.translate) ? st[j.blk].q_sym[i] : (u8)i;
	for (u32 i = threadIdx.x; i < bins; i += blockDim.x) s_base[i] = 0;
	__syncthreads();
	if (!(PROBE & 1))
	{	// histogram of digit 0
		bool bad = false;
		u32 i_from = 0;
		if (j.is_dna && j.alpha_bits == 2 && j.dbits <= 10 && n >= 16)
		{	// 2-bit bases: the digits of four consecutive symbols t..t+3 come out of ONE 8-byte window s[t-5 .. t+2]
			// (packed newest-last, symbol t+i is the 10-bit field at bit 6-2i); the histogram does not care about order
			const u32 dmask = (bins - 1) & (u32)((1ull << (2 * j.order)) - 1ull);
			const u32 n4 = (n - 8) / 4;                                   // groups starting at t = 8, 12, ...
			for (u32 g = threadIdx.x; g < n4; g += blockDim.x)
			{
				const u32 t = 8 + 4 * g;
				const u64 w = *(const u64_unaligned*)(sym_src + t - 5);
				if ((w & 0xFCFCFCFCFCFCFCFCull) || sym_src[t + 3] >= 4) bad = true;
				const u32 R = pack2x8(__builtin_bswap64(w));                 // byte b of w at bits 2*(7-b)
#pragma unroll
				for (u32 k = 0; k < 4; ++k) atomicAdd(&s_base[(R >> (6 - 2 * k)) & dmask], 1u);
			}
			// symbols 0..7 and the tail go the general way
			for (u32 i = threadIdx.x; i < 8; i += blockDim.x) atomicAdd(&s_base[ctx_digit0(j, sym_src, qp, s_rank, i, bins - 1, &bad)], 1u);
			i_from = 8 + 4 * n4;
		}
		if (!j.is_dna && j.qlen && 2 * j.alpha_bits >= j.dbits && n >= 16)
		{	// quality with reads of one length: digit = (rank of s[t-1] << ab | position context) -- four consecutive
			// symbols share one 4-byte window s[t-1 .. t+2] and one division for the position inside the read
			const u32 dmask = bins - 1;
			const u32 n4 = (n - 1) / 4;                                   // groups starting at t = 1, 5, ...
			for (u32 g = threadIdx.x; g < n4; g += blockDim.x)
			{
				const u32 t = 1 + 4 * g;
				const u32 w = *(const u32_unaligned*)(sym_src + t - 1);
				u32 pos = t - exact_div(t, j.qm_lo, j.qm_hi) * j.qlen;
#pragma unroll
				for (u32 k = 0; k < 4; ++k)
				{
					const u32 v0 = s_rank[(w >> (8 * k)) & 0xFFu];
					const u32 pc = exact_div(pos * 128u, j.qm_lo, j.qm_hi) >> j.rescale_shift;
					atomicAdd(&s_base[((v0 << j.alpha_bits) | pc) & dmask], 1u);
					pos = pos + 1 == j.qlen ? 0u : pos + 1;
				}
			}
			if (threadIdx.x == 0) atomicAdd(&s_base[ctx_digit0(j, sym_src, qp, s_rank, 0, bins - 1, &bad)], 1u);
			i_from = 1 + 4 * n4;
		}
		for (u32 i = i_from + threadIdx.x; i < n; i += blockDim.x)
		{
			atomicAdd(&s_base[ctx_digit0(j, sym_src, qp, s_rank, i, bins - 1, &bad)], 1u);
		}
		if (bad) atomicOr(&st[j.blk].err, (u32)DSRC_ERR_REF_UB);
	}
	__syncthreads();

	for (u32 pass = 0; pass < j.passes; ++pass)
	{
		// sources/destinations alternate so that the last pass lands in the buffer k_replay reads (sorted_in_b)
		const bool to_b = ((j.passes - 1 - pass) & 1u) ? !j.sorted_in_b : (bool)j.sorted_in_b;
		const u64* src = pool + (to_b ? j.elems : j.elems_b);
		u64* dst = pool + (to_b ? j.elems_b : j.elems);
		const u32 shift = ELEM_CTX_SHIFT + pass * j.dbits;
		const bool more = pass + 1 < j.passes;

		{	// exclusive scan of the histogram
			u32 carry = 0;
			for (u32 b0 = 0; b0 < bins; b0 += blockDim.x)
			{
				const u32 i = b0 + threadIdx.x;
				const u32 v = i < bins ? s_base[i] : 0;
				u32 tot;
				const u32 ex = block_excl_scan(v, &tot);
				if (i < bins) s_base[i] = carry + ex;
				carry += tot;
			}
		}
		for (u32 i = threadIdx.x; i < bins; i += blockDim.x) s_next[i] = 0;
		for (u32 i = threadIdx.x; i < SORT_WAVES * SORT_MAX_BINS; i += blockDim.x) (&s_cnt[0][0])[i] = 0;
		__syncthreads();

		for (u32 tile = 0; tile < n; tile += tile_elems)
		{
			u64 el[SORT_ITEMS]; u32 rk[SORT_ITEMS];
			const u32 wbase = tile + wv * 64 * SORT_ITEMS;
			bool bad = false;
#pragma unroll
			for (u32 k = 0; k < SORT_ITEMS; ++k)
			{
				const u32 i = wbase + k * 64 + lane;
				el[k] = 0;
				if (PROBE & 32) el[k] = ((u64)(i * 2654435761u) << 32) | i;
				else if (i < n) el[k] = pass ? src[i] : (j.is_dna ? ctx_elem_dna(j, sym_src, i, &bad) : ctx_elem_qua(j, sym_src, qp, s_rank, i));
			}
#pragma unroll
			for (u32 k = 0; k < SORT_ITEMS; ++k)
			{
				if (PROBE & 2) { rk[k] = 0; continue; }
				const u32 i = wbase + k * 64 + lane;
				const bool valid = i < n;
				const u32 d = (u32)(el[k] >> shift) & (bins - 1);
				if (ATOMIC)
				{	// one LDS atomic per element on the wave's packed counter pair: the LDS serialises the lanes of one instruction
					// that meet in a word in ascending lane order (k_lds_order_test checks exactly that on the device before this
					// variant is ever launched), and a wave's instructions in program order -- which is the stable order
					const u32 sh = (d & 1u) * 16u;
					u32 old = 0;
					if (valid) old = atomicAdd(&((u32*)s_cnt[wv])[d >> 1], 1u << sh);
					rk[k] = (old >> sh) & 0xFFFFu;
#ifdef DSRC_EMU_BUILD
					(void)__ballot(true);                                     // the emulator runs lanes one after the other: keep them in step per k
#endif
					continue;
				}
				u64 peers = __ballot(valid);
#pragma unroll
				for (u32 b = 0; b < SORT_DIGIT_BITS; ++b)                   // digit bits above dbits are zero for every lane: no effect
				{
					const u64 m = __ballot((d >> b) & 1u);
					peers &= ((d >> b) & 1u) ? m : ~m;
				}
				const u32 before = valid ? s_cnt[wv][d] : 0;          // this wave's earlier elements with the same digit
				const u32 r = (u32)__popcll(peers & lanemask_lt());
				rk[k] = before + r;
				const u64 sync = __ballot(true);                       // every lane has read its counter before a leader bumps it
				if (valid && r == 0 && sync) s_cnt[wv][d] = (u16)(before + (u32)__popcll(peers));
			}
			__syncthreads();
			// the tile's digit counts -> tile positions (exclusive scan over digits, then over the waves of a digit)
			for (u32 d0 = 0, carry = 0; d0 < bins && !(PROBE & 4); d0 += SORT_WG)              // one round with 1024 threads
			{
				const u32 dd = d0 + threadIdx.x;
				u32 c[SORT_WAVES], tot = 0;
#pragma unroll
				for (u32 w = 0; w < SORT_WAVES; ++w) { c[w] = dd < bins ? s_cnt[w][dd] : 0u; tot += c[w]; }
				const u32 inc = wave_incl_scan_dpp(tot);
				if (lane == 63) s_ws[wv] = inc;
				__syncthreads();
				u32 run = carry + inc - tot;
				const u32 nws = (bins - d0 + 63) / 64 < SORT_WAVES ? (bins - d0 + 63) / 64 : SORT_WAVES;   // waves that hold digits
				for (u32 i = 0; i < nws; ++i) { const u32 x = s_ws[i]; run += i < wv ? x : 0u; carry += x; }
				if (dd < bins)
				{
					const u32 g = s_base[dd];
					s_delta[dd] = g - run; s_base[dd] = g + tot;
#pragma unroll
					for (u32 w = 0; w < SORT_WAVES; ++w) { s_off[w][dd] = (u16)run; run += c[w]; s_cnt[w][dd] = 0; }
				}
				if (d0 + SORT_WG < bins) __syncthreads();
			}
			__syncthreads();
#pragma unroll
			for (u32 k = 0; k < SORT_ITEMS; ++k)
			{
				const u32 i = wbase + k * 64 + lane;
				if (PROBE) { if (i < n && !(PROBE & 8)) s_tile[(wv * SORT_ITEMS + k) * 64 + lane + (rk[k] & 1u)] = el[k]; }
				else if (i < n) s_tile[(u32)s_off[wv][(u32)(el[k] >> shift) & (bins - 1)] + rk[k]] = el[k];
			}
			__syncthreads();
			// out in tile order: the elements of a digit are neighbours here and in dst, so a wave's store covers a few
			// contiguous runs instead of 64 separate 8-byte targets
			const u32 tile_n = n - tile < tile_elems ? n - tile : tile_elems;
#pragma unroll
			for (u32 k = 0; k < SORT_ITEMS; ++k)
			{
				const u32 p = k * SORT_WG + threadIdx.x;
				if (p < tile_n && !(PROBE & 8))
				{
					const u64 e = s_tile[p];
					if (!PROBE) dst[s_delta[(u32)(e >> shift) & (bins - 1)] + p] = e;
					else if (!(PROBE & 16)) dst[tile + p + (s_delta[(u32)(e >> shift) & (bins - 1)] & 0u)] = e;
					if (more && !(PROBE & 64)) atomicAdd(&s_next[(u32)(e >> (shift + j.dbits)) & (bins - 1)], 1u);
				}
			}
			// no barrier here: the next tile touches s_delta / s_tile only behind its own first two barriers
		}
		__syncthreads();                                    // the last tile's s_next updates
		for (u32 i = threadIdx.x; i < bins; i += blockDim.x) s_base[i] = s_next[i];
		__syncthreads();
	}
	}
}

// ceil(2^48 / d) for 2 <= d < 2^16 without a 64-bit integer division (a ~100-instruction sequence on this machine), as the low 52
// bits of a double.  y = 1 / d to within 2^-52 of it either way (the hardware's estimate and two Newton steps; the IEEE division --
// v_div_scale x 2, v_rcp, 6 fma, v_div_fmas, v_div_fixup -- was a third of a loader wave's f64 instructions in k_rcs).  With
// K = 2^48 * (1 - 2^-50) the product y * K = Q * (1 - e), Q = 2^48 / d, 0.6 * 2^-50 < e < 1.4 * 2^-50: below Q by less than
// 0.35 / d.  Q is an integer (d a power of two) or at least 1 / d away from one, so floor(y * K) = ceil(Q) - 1 either way, and
// t = y * K + 0.5 (one fma: the product is not rounded on its own; t's rounding is below 0.04 / d) lies strictly between
// ceil(Q) - 0.5 + 0.6 / d and ceil(Q) + 0.5 - 0.9 / d: t + 2^52 rounds to 2^52 + ceil(Q), whose mantissa field is the result.
// Round 6 took the integer apart with two conversions, a multiplication and an fma and added one unless d was a power of two: 19
// instructions per record against 10 with the packing, and k_rcs's period waited for the loader waves.  k_selftest tries every d on the device.
__device__ __forceinline__ u64 recip48_bits(u32 d)
{
#ifdef DSRC_EMU_BUILD
	const double y = 1.0 / (double)d;
#else
	const double x = (double)d;
	double y = __builtin_amdgcn_rcp(x);
	y = __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
	y = __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
#endif
	const double t = __builtin_fma(y, (1.0 - 0x1p-50) * 0x1p48, 0.5);
	const double r = t + 0x1p52;
	u64 bits;
	__builtin_memcpy(&bits, &r, 8);
	return bits;                                               // 0x433 << 52 | ceil(2^48 / d)
}
__device__ __forceinline__ u64 recip48(u32 d) { return recip48_bits(d) & 0xFFFFFFFFFFFFFull; }
// the two words wave R multiplies by: the top 32 bits of m << 16 and its low 32 bits (m < 2^47; the alignbit drops the exponent field)
__device__ __forceinline__ void recip48_ab(u32 d, u32* a, u32* b)
{
	const u64 bits = recip48_bits(d);
	*a = __builtin_amdgcn_alignbit((u32)(bits >> 32), (u32)bits, 16);
	*b = (u32)bits << 16;
}

// Is the order in which the LDS applies the lanes of ONE ds_add_rtn instruction to one word the lane order?  k_sort<.., true>
// relies on it.  Every wave tries `rounds` digit patterns (1 .. 512 distinct digits, full and ragged execution masks) and compares
// what the atomics return with the rank computed from ballots.  *bad != 0: the host keeps the ballot variant.
__global__ void __launch_bounds__(256) k_lds_order_test(u32* bad, u32 rounds)
{
	__shared__ u32 s_c[4][SORT_MAX_BINS / 2];
	const u32 wv = wave_id(), lane = lane_id();
	u32 wrong = 0;
	for (u32 round = 0; round < rounds; ++round)
	{
		for (u32 i = lane; i < SORT_MAX_BINS / 2; i += 64) s_c[wv][i] = 0;
		wave_fence();
		const u32 mask = (2u << (round % SORT_DIGIT_BITS)) - 1u;        // 2 .. SORT_MAX_BINS bins
		u32 x = (blockIdx.x * 4 + wv) * 0x9E3779B9u + round * 0x85EBCA6Bu + lane * 0xC2B2AE35u;
		x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
		const u32 d = (round & 64) ? (x & mask & ~1u) | (lane & 1u) : x & mask;     // also patterns where neighbours share a word
		const bool valid = (round & 128) ? ((x >> 20) & 3u) != 0 : true;
		u64 peers = __ballot(valid);
#pragma unroll
		for (u32 b = 0; b < SORT_DIGIT_BITS; ++b) { const u64 m = __ballot((d >> b) & 1u); peers &= ((d >> b) & 1u) ? m : ~m; }
		const u32 expect = (u32)__popcll(peers & lanemask_lt());
		const u32 sh = (d & 1u) * 16u;
		u32 old = 0;
		if (valid) old = atomicAdd(&s_c[wv][d >> 1], 1u << sh);
		if (valid && ((old >> sh) & 0xFFFFu) != expect) wrong = 1;
		wave_fence();
	}
	if (wrong) atomicOr(bad, 1u);
}

// The same question for the 64-bit form (ds_add_rtn_u64, four 16-bit fields per word): k_model (k_bucket.h) codes a symbol with one
// such atomic per radix-4 level of its counter row and needs what comes back to be the state as of the lanes below.  bad bit 1.
__global__ void __launch_bounds__(256) k_lds_order_test64(u32* bad, u32 rounds)
{
	__shared__ unsigned long long s_c[4][SORT_MAX_BINS / 4];
	const u32 wv = wave_id(), lane = lane_id();
	u32 wrong = 0;
	for (u32 round = 0; round < rounds; ++round)
	{
		for (u32 i = lane; i < SORT_MAX_BINS / 4; i += 64) s_c[wv][i] = 0;
		wave_fence();
		const u32 mask = (2u << (round % SORT_DIGIT_BITS)) - 1u;
		u32 x = (blockIdx.x * 4 + wv) * 0x9E3779B9u + round * 0x85EBCA6Bu + lane * 0xC2B2AE35u;
		x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
		const u32 d = (round & 64) ? (x & mask & ~3u) | (lane & 3u) : x & mask;     // also patterns where neighbours share a word
		const bool valid = (round & 128) ? ((x >> 20) & 3u) != 0 : true;
		u64 peers = __ballot(valid);
#pragma unroll
		for (u32 b = 0; b < SORT_DIGIT_BITS; ++b) { const u64 m = __ballot((d >> b) & 1u); peers &= ((d >> b) & 1u) ? m : ~m; }
		const u32 expect = (u32)__popcll(peers & lanemask_lt());
		const u32 sh = (d & 3u) * 16u;
		unsigned long long old = 0;
		if (valid) old = atomicAdd(&s_c[wv][d >> 2], 1ull << sh);
		if (valid && ((u32)(old >> sh) & 0xFFFFu) != expect) wrong = 1;
		wave_fence();
	}
	if (wrong) atomicOr(bad, 2u);
}

// device self-test (dsrcgpu_selftest): recip48 against the integer division for every divisor, and rc_div against
// the hardware division on a spread of numerators
__device__ __forceinline__ u32 rc_div(u32 range, u32 m_lo, u32 m_hi);
__global__ void __launch_bounds__(256) k_selftest(u32* bad)
{
	const u32 d = blockIdx.x * blockDim.x + threadIdx.x + 2;
	if (d >= 65536) return;
	const u64 m = recip48(d);
	u32 wrong = m != ((1ull << 48) + d - 1) / d ? 1u : 0u;
	for (u32 k = 0; k < 64; ++k)
	{
		const u32 n = k < 32 ? (0xFFFFFFFFu >> k) : (u32)((u64)d * (k * 2654435761u | 1u) - (k & 1u));
		if (rc_div(n, (u32)m, (u32)(m >> 32)) != n / d) wrong = 1;
	}
	if (wrong) atomicAdd(bad, 1u);
}

// ---- model replay ---------------------------------------------------------------------------
// After the sort every context's history is a contiguous *segment*.  One wave walks a range of
// the sorted array 64 elements at a time and replays all segments inside the window at once:
// inside one adaptive epoch (between two Rescale() calls, src/SymbolCoderRC.h:69-90) the counter
// row is   stats[v] = base[v] + 2 * (#earlier v in the epoch)   so for the symbol in lane i
//     freq  = base[s] + 2 * #{earlier lanes of my segment with the same symbol}
//     cum   = cumbase[s] + 2 * #{earlier lanes of my segment with a smaller symbol}
//     total = T0 + 2 * (symbols coded so far in the epoch)
// which are popcounts of per-symbol ballots masked to the lane's segment.  (The records are scattered into
// the chain's own 8 B x n array: with few chains in flight that array stays in the memory-side cache, so the
// 8-byte pieces merge into full lines before they reach HBM -- the launch uses many waves per chain.)  Only the segment that
// is still open at the end of a window carries state (lane v keeps base[v] / cnt[v]); a window is
// cut short where that segment hits its rescale point.  A wave replays exactly its slice of the array (see
// k_replay_seams for segments that cross slices), so any number of waves can work on one stream.
#define REPLAY_WG 256

template <int N> struct ReplayRow
{
	// lane v holds entry v (and v + 64 when N == 128)
	u32 a, b;
	__device__ __forceinline__ u32 get(u32 sym) const
	{
		const u32 lo = __shfl(a, (int)(sym & 63u));
		if (N <= 64) return lo;
		const u32 hi = __shfl(b, (int)(sym & 63u));
		return sym < 64 ? lo : hi;
	}
};

// value of lane 63 in every lane (v_readlane: no trip through the LDS crossbar)
__device__ __forceinline__ u32 wave_last(u32 v)
{
#ifdef DSRC_EMU_BUILD
	return __shfl(v, 63);
#else
	return (u32)__builtin_amdgcn_readlane((int)v, 63);
#endif
}

template <int N> __device__ __forceinline__ void replay_prefix(const ReplayRow<N>& r, ReplayRow<N>& ex, u32* total)
{
	const u32 ia = wave_incl_scan_dpp(r.a);
	const u32 ta = wave_last(ia);
	ex.a = ia - r.a;
	u32 tb = 0; ex.b = 0;
	if (N > 64) { const u32 ib = wave_incl_scan_dpp(r.b); tb = wave_last(ib); ex.b = ta + ib - r.b; }
	*total = ta + tb;
}

// Ranges: the sorted array of a stream is cut into parts * REPLAY_WG/64 ranges of `per` elements (a multiple of 64,
// at least 256); wave w replays exactly range w.  A segment that crosses a range boundary needs the row state at the
// boundary: k_replay_seams computes it with a cheap walk (counts only: no ranks, no records) by the wave whose range holds
// the segment's head, and leaves it in the sort buffer that is no longer needed (REPLAY_SEAM_WORDS u32 per boundary, which
// always fits: per >= 256 elements of 8 bytes per boundary).  So a context that holds most of a stream's symbols is
// replayed by all the stream's waves, not by one.
// Which (part, stream) a k_replay / k_replay_seams workgroup works on: all parts of a stream are neighbours in the grid, so
// that few streams are in flight at a time and their scattered record lines stay in the memory-side cache (512 parts: 71 ms,
// 32 parts: 164 ms per 512 DNA streams).  Measured and rejected: one stream per XCD (workgroup id -> stream 8 * (id / 8 / parts)
// + id % 8, eight streams in flight): 7.8 / 10.0 instead of 6.7 / 6.7 ms per 128 quality / DNA streams.  Where the time goes
// (one instance, 128 streams): records not stored 2.6 ms, stored in sorted order 2.7 ms, scattered into a 192 KB window
// 3.2 ms, scattered by t 6.7 ms -- the scatter to stream order is two thirds of k_replay.
__device__ __forceinline__ bool replay_slot(u32 parts, u32 cnt, u32* part, u32* stream)
{
	*stream = blockIdx.x / parts;
	*part = blockIdx.x % parts;
	return *stream < cnt;
}

#define REPLAY_SEAM_WORDS 260u
__device__ __forceinline__ u32 replay_per(u32 n, u32 n_ranges)
{
	const u32 per = ((n + n_ranges - 1) / n_ranges + 63u) & ~63u;
	return per < 256u ? 256u : per;
}

// Ranges bend to the segments: the boundary between two ranges is nominally b = w * per, but if a segment starts in
// [b, b + 64) the ranges meet at that segment's head -- the wave before replays the few elements up to it, nobody needs a row
// state for b.  Only a segment that runs through the whole window (a long one) is cut at b itself, with the state from
// k_replay_seams.  Both neighbours and k_replay_seams call this with the same b and get the same answer.  0 < b < n.
__device__ __forceinline__ u32 replay_snap(const u64* src, u32 b, u32 n, bool* inside)
{
	const u32 idx = b + lane_id(), i = idx < n ? idx : n - 1;
	const bool head = idx < n && (src[i] >> ELEM_CTX_SHIFT) != (src[i - 1] >> ELEM_CTX_SHIFT);
	const u64 m = __ballot(head);
	*inside = m == 0;
	return m ? b + (u32)__ffsll((long long)m) - 1u : b;
}

template <int N>
__device__ __forceinline__ void replay_seams_part(const CtxJob& j, u64* pool, u32 parts, u32 part)
{
	constexpr int BITS = N <= 4 ? 2 : N <= 8 ? 3 : N <= 16 ? 4 : N <= 32 ? 5 : N <= 64 ? 6 : 7;
	const u64* src = pool + (j.sorted_in_b ? j.elems_b : j.elems);
	u32* seams = (u32*)(pool + (j.sorted_in_b ? j.elems : j.elems_b));
	const u32 n = j.n;
	const u32 lane = lane_id();
	const u32 limit = (1u << 16) - 2u * N;
	const u32 per = replay_per(n, parts * (REPLAY_WG / 64));
	const u64 r_lo64 = (u64)(part * (REPLAY_WG / 64) + wave_id()) * per;
	if (r_lo64 + per >= n) return;                          // no boundary after this range
	const u32 r_lo = (u32)r_lo64, hi = r_lo + per;
	{ bool inside; (void)replay_snap(src, hi, n, &inside); if (!inside) return; }      // nothing long crosses my upper boundary
	// head of the segment that crosses it, if it lies in my range (otherwise the owner of that head walks through here)
	u32 h = 0; bool found = false;
	for (u32 we = hi; we > r_lo && !found; we -= 64)
	{
		const u32 i = we - 64 + lane;
		const bool hd = i == 0 || (src[i] >> ELEM_CTX_SHIFT) != (src[i - 1] >> ELEM_CTX_SHIFT);
		const u64 m = __ballot(hd);
		if (m) { h = we - 64 + 63u - (u32)__clzll((long long)m); found = true; }
	}
	if (!found) return;
	const u64 segctx = src[h] >> ELEM_CTX_SHIFT;
	u32 base_a = lane < (u32)N ? 1u : 0u, base_b = N > 64 ? 1u : 0u, cnt_a = 0, cnt_b = 0;
	u32 T0 = N, epoch_cnt = 0, epoch_left = (limit - N + 1) / 2;
	u32 pos = h, next_b = hi;
	u64 el_cur = src[pos + lane < n ? pos + lane : n - 1];
	for (;;)
	{
		const u32 idx = pos + lane;
		const u64 el = el_cur;
		// windows are almost always full: ask for the next 64 elements before this window's length is known
		// (unconditional, clamped address: with a predicated load the compiler waits for it right here: 5.9 -> 4.3 ms;
		// a second window of look-ahead through rotating registers was slower again)
		const u64 el_spec = src[idx + 64 < n ? idx + 64 : n - 1];
		const bool same = idx < n && (el >> ELEM_CTX_SHIFT) == segctx;
		const u64 m_not = __ballot(!same);
		const u32 seg_rem = m_not ? (u32)__ffsll((long long)m_not) - 1 : 64u;
		u32 tile_len = seg_rem;
		if (epoch_left < tile_len) tile_len = epoch_left;
		if (next_b - pos < tile_len) tile_len = next_b - pos;
		const bool seg_ends = tile_len == seg_rem && seg_rem < 64;
		const u64 tmask = tile_len >= 64 ? ~0ull : ((1ull << tile_len) - 1ull);
		const u32 sym = (u32)(el >> ELEM_SYM_SHIFT) & 0xFFu;
		// lane c counts the lanes of the window that carry symbol c (and c + 64)
		u64 ma = tmask, mb = tmask;
#pragma unroll
		for (int k = BITS - 1; k >= 0; --k)
		{
			const u64 m = __ballot(lane < tile_len && ((sym >> k) & 1u));
			if (k == 6) { ma &= ~m; mb &= m; }
			else { const u64 sel = ((lane >> k) & 1u) ? m : ~m; ma &= sel; mb &= sel; }
		}
		if (lane < (u32)N) cnt_a += (u32)__popcll(ma);
		if (N > 64) cnt_b += (u32)__popcll(mb);
		epoch_cnt += tile_len; epoch_left -= tile_len; pos += tile_len;
		if (epoch_left == 0)
		{	// Rescale(), as in k_replay
			u32 x = base_a + 2 * cnt_a; base_a = lane < (u32)N ? x - (x >> 1) : 0u;
			if (N > 64) { x = base_b + 2 * cnt_b; base_b = x - (x >> 1); }
			cnt_a = cnt_b = 0;
			T0 = wave_sum(base_a + base_b);
			epoch_cnt = 0; epoch_left = (limit - T0 + 1) / 2;
		}
		if (seg_ends || pos >= n) break;
		el_cur = tile_len == 64 ? el_spec : src[pos + lane < n ? pos + lane : n - 1];
		if (pos == next_b)
		{
			u32* st = seams + (u64)(pos / per) * REPLAY_SEAM_WORDS;
			if (lane == 0) { st[0] = T0; st[1] = epoch_cnt; st[2] = epoch_left; st[3] = 0x5EA35EA3u; }
			// lanes >= N hold zeros by construction and the second halves only exist for N = 128: nothing else is written or read
			if (lane < (u32)N) { st[4 + lane] = base_a; st[132 + lane] = cnt_a; }
			if (N > 64) { st[68 + lane] = base_b; st[196 + lane] = cnt_b; }
			next_b += per;
		}
	}
}

// Launch forms of k_replay_seams / k_replay: fb == nullptr: workgroup id -> (stream, part) of jobs[0 .. n_streams) (replay_slot);
// otherwise fb = {count, job ids ...} (k_bucket.h: the streams k_part handed back) and gridDim.x / parts streams are in flight at a time.
template <int N>
__global__ void __launch_bounds__(REPLAY_WG) k_replay_seams(const CtxJob* jobs, u64* pool, u32 parts, u32 n_streams, const u32* fb)
{
	if (!fb)
	{
		u32 part, stream;
		if (replay_slot(parts, n_streams, &part, &stream)) replay_seams_part<N>(jobs[stream], pool, parts, part);
		return;
	}
	const u32 n_f = fb[0], part = blockIdx.x % parts;
	for (u32 f = blockIdx.x / parts; f < n_f; f += gridDim.x / parts) replay_seams_part<N>(jobs[fb[1 + f]], pool, parts, part);
}

// PROBE != 0: timing experiments only (-DDSRC_SORT_PROBE): 1 no record store, 2 records stored in sorted order (no scatter),
// 8 / 16 scatter inside an 8 MB / 128 KB window
template <int N, int PROBE>
__device__ __forceinline__ void replay_part(const CtxJob& j, const u64* pool, RcPack* rec_pool, u32 parts, u32 part, u32 (*s_tail)[128])
{
	constexpr int BITS = N <= 4 ? 2 : N <= 8 ? 3 : N <= 16 ? 4 : N <= 32 ? 5 : N <= 64 ? 6 : 7;
	const u64* src = pool + (j.sorted_in_b ? j.elems_b : j.elems);
	RcPack* recs = rec_pool + j.trip;
	const u32 n = j.n;
	const u32 lane = lane_id();
	const u32 limit = (1u << 16) - 2u * N;                   // MaxAccumulatedValue (src/SymbolCoderRC.h:67)
	const u32 per = replay_per(n, parts * (REPLAY_WG / 64));
	const u64 r_lo = (u64)(part * (REPLAY_WG / 64) + wave_id()) * per;
	if (r_lo >= n) return;
	u32 hi = (u32)(r_lo + per < n ? r_lo + per : n);
	u32 pos = (u32)r_lo;
	// both ends bend to the next segment head if there is one within 64 elements (replay_snap); a range that still starts
	// inside a segment takes the row state k_replay_seams left for its lower boundary
	bool mid = false;
	if (pos > 0) pos = replay_snap(src, pos, n, &mid);
	if (hi < n) { bool in_hi; hi = replay_snap(src, hi, n, &in_hi); }

	ReplayRow<N> base, cnt, cumbase, cntpre;
	base.a = base.b = 1; cnt.a = cnt.b = 0; cumbase.a = cumbase.b = 0; cntpre.a = cntpre.b = 0;
	bool open = false;                      // a segment continues from the previous window
	u32 T0 = N, epoch_cnt = 0, epoch_left = 0;
	const u32 E0 = (limit - N + 1) / 2;     // symbols a fresh row codes before its first rescale
	if (mid)
	{
		const u32* st = (const u32*)(pool + (j.sorted_in_b ? j.elems : j.elems_b)) + (u64)(pos / per) * REPLAY_SEAM_WORDS;
		T0 = st[0]; epoch_cnt = st[1]; epoch_left = st[2];
		base.a = lane < (u32)N ? st[4 + lane] : 0u; cnt.a = lane < (u32)N ? st[132 + lane] : 0u;
		base.b = N > 64 ? st[68 + lane] : 0u; cnt.b = N > 64 ? st[196 + lane] : 0u;
		u32 t; replay_prefix<N>(base, cumbase, &t); replay_prefix<N>(cnt, cntpre, &t);
		open = true;
	}

	// software pipeline: the next window is requested as soon as this window's length is known
	u64 el_cur = pos + lane < n ? src[pos + lane] : 0;
	u64 prev_ctx = pos > 0 ? (src[pos - 1] >> ELEM_CTX_SHIFT) : ~0ull;      // context of the element before the window
	for (;;)
	{
		if (pos >= n) break;
		const u32 idx = pos + lane;
		const bool valid = idx < n;
		const u64 el = el_cur;
		const u64 ctx = el >> ELEM_CTX_SHIFT;
		u64 pctx = __shfl_up(ctx, 1);
		if (lane == 0) pctx = prev_ctx;
		const bool head = valid && ctx != pctx;
		const u64 hm_all = __ballot(head);
		u32 tile_len = (u32)__popcll(__ballot(valid));
		bool last = false;
		if (pos + tile_len >= hi) { tile_len = hi - pos; last = true; }      // every wave stops at the end of its range
		if (!(hm_all & 1ull) && !open) { /* cannot happen: windows start on a head unless a segment is open */ }
		bool rescale_after = false;
		const bool cont = open && !(hm_all & 1ull);          // lane 0 continues the open segment
		if (cont)
		{
			u32 c = hm_all ? (u32)__ffsll((long long)hm_all) - 1 : 64u;
			if (c > tile_len) c = tile_len;
			if (c > epoch_left) { tile_len = epoch_left; rescale_after = true; last = false; }
			else if (c == epoch_left && c == tile_len && !last) { /* boundary falls on the window end: rescale lazily below */ }
		}
		const u32 npos = pos + tile_len;
		const u64 el_next = npos + lane < n ? src[npos + lane] : 0;
		const u64 tmask = tile_len >= 64 ? ~0ull : ((1ull << tile_len) - 1ull);
		const bool active = lane < tile_len;
		const u64 hm = hm_all & tmask;
		const u64 heads_le = hm & (lanemask_lt() | (1ull << lane));
		const u32 seg_start = heads_le ? 63u - (u32)__clzll((long long)heads_le) : 0u;
		const bool in_cont = cont && heads_le == 0;
		const u64 segmask_lt = lanemask_lt() & ~((1ull << seg_start) - 1ull);
		const u32 sym = (u32)(el >> ELEM_SYM_SHIFT) & 0xFFu;
		// the segment that stays open after this window = the one containing lane tile_len-1
		const u32 last_start = hm ? 63u - (u32)__clzll((long long)hm) : 0u;
		const bool last_is_cont = cont && hm == 0;
		const u64 lastmask = tmask & ~((1ull << last_start) - 1ull);

		// lanes of the window whose symbol equals mine (EQ) / is smaller than mine (LT): one ballot per symbol BIT,
		// refined from the most significant bit down (radix compare) -- log2(N) steps instead of N
		u64 EQ = tmask, LT = 0;
#pragma unroll
		for (int k = BITS - 1; k >= 0; --k)
		{
			const bool bit = (sym >> k) & 1u;
			const u64 m = __ballot(active && bit);
			const u64 sb = bit ? ~0ull : 0ull;
			LT |= EQ & ~m & sb;
			EQ &= ~(m ^ sb);
		}
		const u32 same = (u32)__popcll(EQ & segmask_lt), less = (u32)__popcll(LT & segmask_lt);
		// counts carried by the segment that stays open: the last lane of each symbol inside it publishes that
		// symbol's count to the lane that holds the symbol's counter
		u32* tl = s_tail[wave_id()];
		tl[lane] = 0; if (N > 64) tl[lane + 64] = 0;
		wave_fence();
		const u64 eq_last = EQ & lastmask;
		if (active && ((lastmask >> lane) & 1ull) && (eq_last >> lane) == 1ull) tl[sym] = (u32)__popcll(eq_last);
		wave_fence();
		const u32 add_a = tl[lane], add_b = (N > 64) ? tl[lane + 64] : 0u;
		const u32 in_seg = (u32)__popcll(segmask_lt & tmask);
		u32 f, cum, tot;
		{
			ReplayRow<N> st, cs;                                   // the row and its prefix as they stand at the start of the window
			st.a = base.a + 2 * cnt.a; st.b = base.b + 2 * cnt.b; cs.a = cumbase.a + 2 * cntpre.a; cs.b = cumbase.b + 2 * cntpre.b;
			const u32 b0 = st.get(sym), cb = cs.get(sym);
			if (in_cont) { f = b0 + 2 * same; cum = cb + 2 * less; tot = T0 + 2 * (epoch_cnt + in_seg); }
			else { f = 1 + 2 * same; cum = sym + 2 * less; tot = N + 2 * in_seg; }
		}
		if (active)
		{	// what k_rc needs per symbol: freq, cum, total -- one aligned 8-byte record (one 32-byte sector per store; the 12-byte
			// record with the reciprocal took 1.25, and a stream's array was half as large again)
			const RcPack rr = (u64)f | ((u64)cum << 16) | ((u64)tot << 32);
			if (PROBE & 8) recs[(u32)el & 0xFFFFFu] = rr;            // scatter inside 8 MB
			else if (PROBE & 16) recs[(u32)el & 0x3FFFu] = rr;      // scatter inside 128 KB
			else if (!(PROBE & 3)) rc6_store(recs, (u32)el, f, cum, tot);   // (a non-temporal store here: 169 instead of 55 ms per 512 blocks -- the lines do merge in the caches)
			else if (PROBE & 2) recs[idx] = rr;
			else if (rr == ~0ull) recs[idx] = rr;                  // keeps the computation alive, never true
		}

		// carry the segment that is open at the end of the window
		if (tile_len > 0)
		{
			if (last_is_cont) { cnt.a += add_a; cnt.b += add_b; epoch_cnt += tile_len; epoch_left -= tile_len; }
			else
			{
				base.a = (lane < (u32)N) ? 1u : 0u; base.b = (N > 64) ? 1u : 0u;
				cnt.a = add_a; cnt.b = add_b;
				T0 = N; epoch_cnt = tile_len - last_start; epoch_left = E0 - epoch_cnt;
				u32 t; replay_prefix<N>(base, cumbase, &t);
				open = true;
			}
		}
		if (rescale_after || (open && epoch_left == 0))
		{	// Rescale(): stats[i] -= stats[i] >> 1 on stats = base + 2*cnt
			u32 x = base.a + 2 * cnt.a; base.a = (lane < (u32)N) ? x - (x >> 1) : 0u;
			if (N > 64) { x = base.b + 2 * cnt.b; base.b = x - (x >> 1); }
			cnt.a = cnt.b = 0;
			replay_prefix<N>(base, cumbase, &T0);
			epoch_cnt = 0; epoch_left = (limit - T0 + 1) / 2;
		}
		{ u32 t; replay_prefix<N>(cnt, cntpre, &t); }
		if (tile_len) prev_ctx = __shfl(ctx, (int)(tile_len - 1));
		el_cur = el_next;
		pos = npos;
		if (last) break;
	}
}

template <int N, int PROBE = 0>
__global__ void __launch_bounds__(REPLAY_WG) k_replay(const CtxJob* jobs, const u64* pool, RcPack* rec_pool, u32 parts, u32 n_streams, const u32* fb)
{
	__shared__ u32 s_tail[REPLAY_WG / 64][128];
	if (!fb)
	{
		u32 part, stream;
		if (replay_slot(parts, n_streams, &part, &stream)) replay_part<N, PROBE>(jobs[stream], pool, rec_pool, parts, part, s_tail);
		return;
	}
	const u32 n_f = fb[0], part = blockIdx.x % parts;
	for (u32 f = blockIdx.x / parts; f < n_f; f += gridDim.x / parts) replay_part<N, PROBE>(jobs[fb[1 + f]], pool, rec_pool, parts, part, s_tail);
}

// ---- range coder: one lane = one stream ---------------------------------------------------------
// The only serial part of the path.  Per symbol the dependent chain is
//   range -> floor(range / total) -> * freq -> renormalise          low -> low + r*cum -> renormalise
// and on the GPU its cost is instruction issue of ONE wave: 71 clocks per symbol for both recurrences in one lane (tools/rc_lab.hip),
// 43 for the range alone.  `low` never feeds back into `range` -- except through the carry clamp of RangeEncoder::EncodeFrequency
// (src/RangeCoder.h:64-74), which needs bits 24..47 of low to be all ones when a byte leaves (once in 2^24 symbols) and a carry on top of that.  So (round 6) the
// recurrence is cut in two waves of one workgroup, k_rcs:
//   * wave R (lane = stream): range alone.  r = floor(range / total) is a multiply by the 64-bit reciprocal the loaders prepared
//     (rc_div64), range' = r * freq renormalised by clz; it leaves r and the bytes-leaving count per symbol in LDS;
//   * wave L (lane = stream), one 64-symbol chunk behind: low' = (low + r * cum) << 8 k, the symbol's code (top three bytes of low
//     and how many of them leave) in place of r -- and the clamp's pre-condition as a running maximum, checked per 16 symbols; when
//     it shows, the group is walked again with the reference's loop, and a clamp that really fires puts the stream on the REDO
//     list: k_rc below (both recurrences in one lane, replay of a group on the spot) codes it again from the start, after the
//     batch's state read-back -- about one stream in 10^4;
//   * loader waves: records HBM -> registers (RC_DEPTH chunks ahead) -> the R and L rows in LDS one chunk ahead; codes -> bytes
//     two chunks behind (prefix sum of the byte counts of a row's 64 codes, up to three byte stores per lane).
// They meet at one barrier per chunk.  Every wave of R and L keeps its SIMD to itself (RC_SPARE_SIMD).
struct RcChain
{
	u64 trip;          // RcPack index of the chain's first record (a multiple of 2: 16-byte aligned)
	u64 out_words;     // u32 index of the staging stream
	u32 n;
	u32 out_byte0, out_cap;
	u32 blk, is_dna;
	u32 force_exact;   // tests: take the reference-loop path for every group (same bytes by construction); 2: report a clamp for every stream (the redo path)
	u32 jid;           // the stream's job: bk[jid] != 0 = handed back by the bucketed front end
	u32 bk_on;         // ... by a kernel of the device (k_part / k_model), i.e. its records are not there yet when k_rcs runs: skipped, coded by the redo launch
};

#define RC_GROUP 16                    // symbols per register group / clamp check
#ifndef RC_LANES
#define RC_LANES 32                    // chains per wave (lanes beyond idle): half a wave keeps the LDS of a workgroup at ~100 KB
#endif
#define RC_CHUNK 64                    // symbols per chain per LDS chunk
#define RC_ROW_U4 49                   // LDS row pitch in 16-byte units: 48 of data + 1 so that a 16-lane ds_read_b128 pass covers all 64 banks
#ifndef RC_LOADERS
#define RC_LOADERS 8                   // loader waves per workgroup (each feeds RC_LANES / RC_LOADERS rows)
#endif
// Waves of a workgroup go round the CU's four SIMDs in order.  A serial wave keeps its SIMD to itself when the waves that would
// share it leave at once (RC_SPARE_SIMD) -- it is bound by the VALU issue of its SIMD, every cycle a loader spends there is added
// to the chain of dependent steps.  k_rc: wave 0 codes, waves 4, 8, .. leave; k_rcs: waves 0 (R) and 1 (L), waves 4, 5, 8, 9, .. leave.
#ifndef RC_SPARE_SIMD
#define RC_SPARE_SIMD 1
#endif
#if RC_SPARE_SIMD
#define RC_WG_WAVES (1 + RC_LOADERS + (RC_LOADERS + 2) / 3)
#define RCS_WG_WAVES (2 * RC_LOADERS)               // waves 0 (R), 1 (L), then of every four the two on SIMDs 2 and 3
#else
#define RC_WG_WAVES (1 + RC_LOADERS)
#define RCS_WG_WAVES (2 + RC_LOADERS)
#endif
#ifndef RC_DEPTH
#define RC_DEPTH 8                     // register sets of a loader wave: a chunk is requested RC_DEPTH - 1 chunk periods before it is converted (even)
#endif
#define RC_OVERREAD ((RC_DEPTH + 4) * RC_CHUNK)      // (k_rcs: RCS_DEPTH + 2 chunks)     // records the loaders may touch past the longest chain of a wave (arena slack)
#define RC_XB 64                       // per-lane byte buffer of the exact path (LDS)

typedef u32 __attribute__((vector_size(16))) U4;   // one 16-byte LDS / global access

// magic = ceil(2^48 / d): floor(n * magic / 2^48) == floor(n / d) for every n < 2^32, d <= 2^16 (the error term
// n*e/2^48 < 2^-16 cannot carry the fraction (<= 1 - 1/d) over an integer)
__device__ __forceinline__ u32 rc_div(u32 range, u32 m_lo, u32 m_hi)
{
	const u64 p = (u64)range * m_hi + __umulhi(range, m_lo);
	return __builtin_amdgcn_alignbit((u32)(p >> 32), (u32)p, 16);          // (u32)(p >> 16) as one opaque 32-bit value
}
// the same quotient with the magic shifted up by 16 (a = bits 63..32, b = bits 31..0 of magic << 16): the top dword of the product,
// no shift (k_rcs, wave R)
__device__ __forceinline__ u32 rc_div64(u32 range, u32 a, u32 b)
{
	const u64 p = (u64)range * a + __umulhi(range, b);
	return (u32)(p >> 32);
}

struct RcState { u64 low; u32 range; };

// fast step: returns the symbol's code = top three bytes of low (bits 31..8) | 8 * bytes leaving (0..2: range' >= 2^8
// because range >= 2^24 and total <= 2^16); `flag` collects the clamp pre-condition as a running maximum of bits 8..39 of low
// (its top half is the maximum of bits 24..39: one alignbit per symbol, no mask; the count stays multiplied by 8: no shift)
__device__ __forceinline__ u32 rc_step_fast(RcState& s, const RcRec& e, u32& flag)
{
	const u32 r = rc_div(s.range, e.m_lo, e.mf >> 16);
	const u64 low = s.low + (u64)r * e.cum;                                // r*cum <= range < 2^32: identical to the reference's 32-bit product
	const u32 range = r * (e.mf & 0xFFFFu);
	__builtin_assume(range != 0);
	const u32 lz = (u32)__builtin_clz(range);
	const u32 k8 = lz & 0x18u;                                             // 8 * bytes leaving
	const u32 z = __builtin_amdgcn_alignbit((u32)(low >> 32), (u32)low, 8);     // bits 24..39 all ones <=> z >> 16 == 0xFFFF
	flag = z > flag ? z : flag;
	s.low = low << k8;
	s.range = range << k8;
	return ((u32)(low >> 32) & 0xFFFFFF00u) | k8;
}

// exact step: RangeEncoder::EncodeFrequency, verbatim; bytes go to the lane's LDS buffer
__device__ inline void rc_step_exact(RcState& s, const RcRec& e, u8* xb, u32& nb)
{
	const u32 r = rc_div(s.range, e.m_lo, e.mf >> 16);
	u64 low = s.low + (u64)r * e.cum;
	u32 range = r * (e.mf & 0xFFFFu);
	// a valid record (freq >= 1, total <= 2^16) leaves range >= 2^8: at most two bytes go out.  The bound keeps a record that is
	// not one (a front-end bug) from spinning here for ever; the stream is then wrong and the block fails its check instead.
	for (u32 guard = 0; range <= 0x00FFFFFFu && guard < 8; ++guard)
	{
		if ((low ^ (low + range)) & 0xFF00000000000000ull) { const u32 rr = (u32)low; range = (rr | 0x00FFFFFFu) - rr; }
		if (nb < RC_XB) xb[nb] = (u8)(low >> 56);
		++nb;
		low <<= 8; range <<= 8;
	}
	s.low = low; s.range = range;
}

// one register group: 16 records = 48 dwords = 12 x 16 bytes, as they lie in the chain's row
struct RcRegs { U4 q[3 * RC_GROUP / 4]; };

__device__ __forceinline__ RcRec rc_rec(const RcRegs& g, u32 i)
{
	const u32* d = (const u32*)g.q;
	RcRec e; e.m_lo = d[3 * i]; e.mf = d[3 * i + 1]; e.cum = d[3 * i + 2];
	return e;
}

// codes of one group -> codes[0..15].  `row` = the chain's LDS row the group was read from (still intact).
__device__ __forceinline__ void rc_group(RcState& s, LDS_AS u32* codes, const RcRegs& g, const LDS_AS U4* row, u32 grp, u8* xb, u32* err, u32 force_exact)
{
	const RcState snap = s;
	u32 zmax = force_exact ? 0xFFFF0000u : 0u;
	u32 c[RC_GROUP];
#pragma unroll
	for (u32 i = 0; i < RC_GROUP; ++i) c[i] = rc_step_fast(s, rc_rec(g, i), zmax);
	if ((zmax >> 16) == 0xFFFFu)
	{	// the reference's loop on the same 16 records; its bytes are dealt out three per code slot, in order
		s = snap;
		const LDS_AS u32* d = (const LDS_AS u32*)row + 3 * RC_GROUP * grp;
		u32 nb = 0;
		for (u32 i = 0; i < RC_GROUP; ++i)
		{
			RcRec e; e.m_lo = d[3 * i]; e.mf = d[3 * i + 1]; e.cum = d[3 * i + 2];
			rc_step_exact(s, e, xb, nb);
		}
		if (nb > 3 * RC_GROUP) atomicOr(err, (u32)DSRC_ERR_OUT_OVERFLOW);
#pragma unroll
		for (u32 i = 0; i < RC_GROUP; ++i)
		{
			const u32 have = nb > 3 * i ? nb - 3 * i : 0u, take = have < 3 ? have : 3u;
			c[i] = ((u32)xb[3 * i] << 24) | ((u32)xb[3 * i + 1] << 16) | ((u32)xb[3 * i + 2] << 8) | (take << 3);
		}
	}
#pragma unroll
	for (u32 i = 0; i < RC_GROUP / 4; ++i)
	{
		const U4 v = {c[4 * i], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3]};
		((LDS_AS U4*)codes)[i] = v;
	}
}

__device__ __forceinline__ void rc_load_group(RcRegs& g, const LDS_AS U4* row, u32 grp)
{
#pragma unroll
	for (u32 i = 0; i < 3 * RC_GROUP / 4; ++i) g.q[i] = row[grp * (3 * RC_GROUP / 4) + i];
}

__device__ __forceinline__ RcRec rc_unpack(RcPack v)
{
	const u64 m = recip48((u32)(v >> 32));
	RcRec e; e.m_lo = (u32)m; e.mf = ((u32)(m >> 32) << 16) | ((u32)v & 0xFFFFu); e.cum = ((u32)v >> 16) & 0xFFFFu;
	return e;
}

// Loader wave `lw` of RC_LOADERS feeds rows lw, lw + RC_LOADERS, ...: lane l fetches record l of the row's chunk `chunk` -- one
// dword (freq | cum << 16) and one ushort (total) of the 384-byte chunk (rc6_chunk_off).  Requests run RC_DEPTH chunks ahead.
#define RC_ROWS_PER_LOADER (RC_LANES / RC_LOADERS)
struct RcFetch { u32 fc, tot; };
template <int ROWS> struct RcRowBases { const u8* p[ROWS]; };       // the arrays of a loader's rows (wave-uniform: scalar registers)
template <int ROWS> __device__ __forceinline__ void rc_fetch(RcFetch* r, const RcRowBases<ROWS>& rb, u32 chunk, u32 lw, u32 n_live)
{
	// the row's array in scalar registers + a 32-bit offset per lane, the same for every row (the address as a 64-bit sum per lane and
	// row: two v_lshl_add_u64 per row and period on waves that k_rcs's period waits for)
	const u32 o4 = rc6_chunk_off(chunk) + 4u * lane_id(), o2 = rc6_chunk_off(chunk) + 256u + 2u * lane_id();
#pragma unroll
	for (u32 k = 0; k < (u32)ROWS; ++k)
	{
		// Every row is requested, also the ones a partial workgroup (the last of a launch) does not have: rc_rows_setup gives those the
		// workgroup's first chain's array, nobody uses what comes back.  Skipping them (`live ? load : 0`, a wave-uniform condition
		// known only at run time) made the compiler put every request of the partial workgroup under a branch with its own wait: with
		// 9 ... 31 chains in it the launch took 97-101 ms instead of 77 (660 or 720 streams; 900 leave 4, which hid it).
		const GLOBAL_AS u8* sp = (const GLOBAL_AS u8*)rb.p[k];
		// global, not flat: a flat access orders itself against the LDS traffic
		r[k].fc = *(const GLOBAL_AS u32*)(sp + o4);
		r[k].tot = (u32)*(const GLOBAL_AS u16*)(sp + o2);
	}
}

// ... and turns them into the coder's 12-byte records (reciprocal of the total) in the rows of `buf`;
// a 3-dword stride over the lanes touches every LDS bank once
__device__ __forceinline__ void rc_convert(LDS_AS U4* buf, const RcFetch* r, u32 lw, u32 n_live)
{
#pragma unroll
	for (u32 k = 0; k < RC_ROWS_PER_LOADER; ++k)
	{
		const u32 j = lw + k * RC_LOADERS;
		if (j < n_live)
		{
			const u64 m = recip48(r[k].tot);
			LDS_AS u32* d = (LDS_AS u32*)(buf + j * RC_ROW_U4) + 3u * lane_id();
			d[0] = (u32)m; d[1] = ((u32)(m >> 32) << 16) | (r[k].fc & 0xFFFFu); d[2] = r[k].fc >> 16;
		}
	}
}

// one 64-symbol chunk of the coder wave: the chain's records are in `cur` (landed), r0 holds its first group
__device__ __forceinline__ void rc_chunk(RcState& s, LDS_AS u32* c, RcRegs& r0, RcRegs& r1, const LDS_AS U4* row, u32 t0, u32 n,
										 u8* xb, u32* err, u32 fx)
{	// c: the chain's row of the chunk's code buffer (64 codes)
	rc_load_group(r1, row, 1);
	if (t0 + 1 * RC_GROUP <= n) rc_group(s, c, r0, row, 0, xb, err, fx);
	rc_load_group(r0, row, 2);
	if (t0 + 2 * RC_GROUP <= n) rc_group(s, c + RC_GROUP, r1, row, 1, xb, err, fx);
	rc_load_group(r1, row, 3);
	if (t0 + 3 * RC_GROUP <= n) rc_group(s, c + 2 * RC_GROUP, r0, row, 2, xb, err, fx);
	if (t0 + 4 * RC_GROUP <= n) rc_group(s, c + 3 * RC_GROUP, r1, row, 3, xb, err, fx);
}

// The codes of a chunk go from the coder to the loader waves through LDS (round 4; before, they went to HBM over the consumed
// records and a second kernel, k_rc_emit, turned them into bytes: 4 B written + 4 B read per symbol and a launch on the critical
// path of every batch).  Row pitch 68 words: a 16-lane ds_write_b128 pass covers all 64 banks.
#define RC_CODE_PITCH 68
// What a loader wave knows about its rows (wave-uniform values, taken once from the lanes of its own copy of the chains:
// scalar registers -- nothing the byte stores need comes from memory, a load here would wait behind the record fetches in flight).
template <int ROWS> struct RcEmitRows { GLOBAL_AS u8* out[ROWS]; u32 limit[ROWS], n_full[ROWS], pos[ROWS]; };

// Loader wave `lw`, rows lw, lw + RC_LOADERS, ...: the 64 codes of chunk `chunk` of each row -> bytes at the row's running position.
// Lane l takes code l (up to three bytes).  A code's byte count is two bits: where a lane's bytes go is a count of the lanes below it
// in two ballots (four v_mbcnt), the row's total two scalar popcounts, and which lanes store comes from the same masks -- a scan over
// the counts was eight DPP steps with their wait states, a readlane and three compares per row on waves that k_rcs's period waits for.
template <int ROWS> __device__ __forceinline__ void rc_emit_chunk(const LDS_AS u32* codes, RcEmitRows<ROWS>& R, u32 chunk, u32 lw, u32 n_live, u32* over)
{
	const u32 lane = lane_id();
	const u32 t0 = chunk * RC_CHUNK;
	u32 v[ROWS];
#pragma unroll
	for (u32 k = 0; k < (u32)ROWS; ++k)
	{
		const u32 j = lw + k * RC_LOADERS;
		v[k] = codes[(j < n_live ? j : 0u) * RC_CODE_PITCH + lane];
	}
#pragma unroll
	for (u32 k = 0; k < (u32)ROWS; ++k)
	{
		const u32 j = lw + k * RC_LOADERS;
		// the row's symbols in this chunk are its first n_ok (all scalar: n_full is a multiple of 16)
		const u32 n_ok = j < n_live && R.n_full[k] > t0 ? (R.n_full[k] - t0 < RC_CHUNK ? R.n_full[k] - t0 : (u32)RC_CHUNK) : 0u;
		const u64 okm = n_ok >= 64u ? ~0ull : (1ull << n_ok) - 1ull;
		const u64 b0 = __ballot((v[k] & 8u) != 0u) & okm, b1 = __ballot((v[k] & 16u) != 0u) & okm;      // the code's low byte is 8 * bytes
		const u32 total = (u32)__popcll(b0) + 2u * (u32)__popcll(b1);
		const u32 at = wave_count_below(b0, R.pos[k] + 2u * wave_count_below(b1, 0u));
		if (R.pos[k] + total > R.limit[k]) *over |= 1u << k;
		else
		{
			GLOBAL_AS u8* out = R.out[k];
			if (wave_lane_in(b0 | b1)) out[at] = (u8)(v[k] >> 24);
			if (wave_lane_in(b1)) out[at + 1] = (u8)(v[k] >> 16);
			if (wave_lane_in(b0 & b1)) out[at + 2] = (u8)(v[k] >> 8);
		}
		R.pos[k] += total;
	}
}

// what a loader wave takes once from the lanes of its own copy of the chains (scalar registers)
template <int ROWS> __device__ __forceinline__ void rc_rows_setup(RcRowBases<ROWS>& rb, RcEmitRows<ROWS>& R, const RcChain& c, u32 n_full, const RcPack* rec_pool, u32* word_pool, u32 lw, u32 n_live)
{
#pragma unroll
	for (u32 k = 0; k < (u32)ROWS; ++k)
	{
		const u32 j = lw + k * RC_LOADERS < n_live ? lw + k * RC_LOADERS : 0u;
		rb.p[k] = uniform_ptr(rec_pool + __shfl(c.trip, (int)j));
		R.out[k] = (GLOBAL_AS u8*)uniform_ptr(word_pool + __shfl(c.out_words, (int)j));
		R.limit[k] = (u32)__builtin_amdgcn_readfirstlane((int)__shfl(c.out_byte0 + c.out_cap, (int)j));
		R.n_full[k] = (u32)__builtin_amdgcn_readfirstlane((int)__shfl(n_full, (int)j));
		R.pos[k] = (u32)__builtin_amdgcn_readfirstlane((int)__shfl(c.out_byte0, (int)j));
	}
}

// the last n mod 16 symbols and RangeEncoder::End, behind the bytes of the chunks
__device__ __forceinline__ void rc_finish(RcState s, const RcChain& c, const RcPack* rec_pool, u32* word_pool, BlkState* st, u32 n_full, u32 pos, bool report, u8* xb)
{
	const RcPack* p = rec_pool + c.trip;
	u8* out = (u8*)(word_pool + c.out_words);
	const u32 limit = c.out_byte0 + c.out_cap;
	u32* err = &st[c.blk].err;
	u32 nb = 0;
	for (u32 t = n_full; t < c.n; ++t) { const RcRec e = rc_unpack(rc6_load(p, t)); rc_step_exact(s, e, xb, nb); }
	if (nb > RC_XB) { if (report) atomicOr(err, (u32)DSRC_ERR_OUT_OVERFLOW); nb = 0; }
	if (pos + nb + 8 > limit) { if (report) atomicOr(err, (u32)DSRC_ERR_OUT_OVERFLOW); }
	else
	{
		for (u32 k = 0; k < nb; ++k) out[pos + k] = xb[k];
		for (u32 k = 0; k < 8; ++k) { out[pos + nb + k] = (u8)(s.low >> 56); s.low <<= 8; }
	}
	pos += nb + 8;
	if (c.is_dna) st[c.blk].dna_bytes = pos; else st[c.blk].qua_bytes = pos;
}

// ---- k_rc: both recurrences in one lane ------------------------------------------------------------------------------------------
// Since round 6 the kernel of the REDO list (`list`: count, then chain ids): streams in which the carry clamp fired under k_rcs and
// streams the bucketed front end handed back on the device (their records come from k_sort / k_replay after the batch's state
// read-back).  With list = nullptr it codes chains [0, n_chains) -- the path the hooks build can force for a whole batch.
// A workgroup is 1 + RC_LOADERS waves.  Wave 0 codes (one lane = one chain, RC_LANES chains); the others fetch the records, RC_DEPTH
// chunks ahead, and write the coder's records into the LDS rows one chunk ahead, so that the coder's instruction stream is the
// arithmetic and nothing else.  They meet at one barrier per 64-symbol chunk: a loader arrives when its rows of the chunk after the
// current one are written, the coder when it has finished the current one -- after the barrier the loaders may overwrite the buffer
// the coder has just left.  The carry clamp's pre-condition (bits 24..39 of low all ones) is accumulated branch-free and checked
// once per 16 symbols: if it shows, the group is replayed from a snapshot with the reference's loop, verbatim.
// FULL: every lane below RC_LANES has a chain; the last workgroup of a launch may be partial and then must not request
// rows it does not have (one launch, two instantiations of the body: a wave only ever fetches the code of its own).
template <bool FULL>
__device__ __forceinline__ void rc_workgroup(const RcChain* chains, u32 n_chains, const u32* list, RcPack* rec_pool, u32* word_pool, BlkState* st, LDS_AS U4* buf_a, LDS_AS U4* buf_b, u8* s_xb,
												 LDS_AS u32* code_a, LDS_AS u32* code_b, LDS_AS u32* s_pos)
{
	__builtin_amdgcn_s_setprio(3);                                             // the serial waves win issue arbitration against co-resident data-parallel waves
	const u32 first_chain = blockIdx.x * RC_LANES;
	const u32 lane = lane_id(), id = first_chain + lane;
	const bool loader = wave_id() >= 1;
#if RC_SPARE_SIMD
	// Waves that leave before the first barrier: on gfx9 (GCN / CDNA) s_barrier releases when every wave of the workgroup that has NOT
	// yet terminated has arrived -- a wave that ends drops out of the count (ISA: "s_barrier: ... waves that have ended are not
	// waited for").  The HIP programming model does not promise that, so the form is tied to the architecture it was measured on;
	// any other target has to build with RC_SPARE_SIMD=0 (all loaders stay, the coder shares its SIMD: 188 instead of 118 ms).
#if !defined(DSRC_EMU_BUILD) && defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "RC_SPARE_SIMD relies on the gfx9 barrier counting only live waves: build other targets with -DRC_SPARE_SIMD=0"
#endif
	if (loader && (wave_id() & 3u) == 0) return;                               // before the first barrier: the hardware counts the waves still alive
	const u32 loader_id = wave_id() - 1u - (wave_id() >> 2);
	if (loader && loader_id >= RC_LOADERS) return;
#else
	const u32 loader_id = wave_id() - 1u;
#endif
	const bool have = lane < RC_LANES && (FULL || id < n_chains);
	const u32 cid = have ? id : n_chains - 1;                                  // idle lanes shadow a real chain's values and code nothing
	const RcChain c = chains[list ? list[1 + cid] : cid];
	const u32 n_live = FULL ? (u32)RC_LANES : n_chains - first_chain;
	const u32 n = have ? c.n : 0;
	const u32 n_full = n & ~(u32)(RC_GROUP - 1);
	const u32 wave_full = (u32)__builtin_amdgcn_readfirstlane((int)wave_max(n_full));      // same value in both waves

	if (loader)
	{
		if (!wave_full) return;
		const u32 lw = loader_id;
		RcRowBases<RC_ROWS_PER_LOADER> rb; RcEmitRows<RC_ROWS_PER_LOADER> R; u32 over = 0;
		rc_rows_setup(rb, R, c, n_full, rec_pool, word_pool, lw, n_live);
		// RC_DEPTH register sets: a chunk is requested RC_DEPTH - 1 chunk periods before it is converted -- with other instances'
		// traffic on the memory system a load can take many microseconds, and the coder waits for the slowest of a chunk's 32
		// rows (round 2, four sets of two loaders: k_rc 130 ms alone, 175 ms next to three other instances' front ends)
		static_assert(RC_DEPTH % 2 == 0 && RC_DEPTH >= 2, "buffer parity must be static");
		RcFetch r[RC_DEPTH][RC_ROWS_PER_LOADER];
#pragma unroll
		for (u32 d = 0; d < RC_DEPTH; ++d) rc_fetch(r[d], rb, d, lw, n_live);
		rc_convert(buf_a, r[0], lw, n_live);
		__syncthreads();                                                       // chunk 0 is there
		// one barrier per chunk the coder works through: it takes them in pairs (buf_a, buf_b)
		const u32 n_sync = 2u * ((wave_full + 2 * RC_CHUNK - 1) / (2 * RC_CHUNK));
		for (u32 q = 0; q < n_sync; q += RC_DEPTH)
		{
#pragma unroll
			for (u32 k = 0; k < RC_DEPTH; ++k)
			{	// the coder is in chunk q + k (buf_a / code_a for even k): its set is free for chunk q + k + RC_DEPTH, the next chunk goes
				// into the other record buffer, and the codes of the chunk before it (in the other code buffer) become bytes
				if (q + k >= n_sync) break;
				rc_fetch(r[k], rb, q + k + RC_DEPTH, lw, n_live);
				rc_convert((k & 1u) ? buf_a : buf_b, r[(k + 1) % RC_DEPTH], lw, n_live);
				if (q + k >= 1) rc_emit_chunk((k & 1u) ? code_a : code_b, R, q + k - 1, lw, n_live, &over);
				__syncthreads();
			}
		}
		rc_emit_chunk(code_b, R, n_sync - 1, lw, n_live, &over);     // n_sync is even: the last chunk's codes are in code_b
#pragma unroll
		for (u32 k = 0; k < RC_ROWS_PER_LOADER; ++k)
			if (lw + k * RC_LOADERS < n_live && lane_id() == 0) s_pos[lw + k * RC_LOADERS] = R.pos[k] | (((over >> k) & 1u) << 31);
		__syncthreads();                                                       // the coder takes the positions for the chains' tails
		return;
	}

	u8* xb = s_xb + lane * RC_XB;
	u32* err = &st[c.blk].err;
	RcState s;
	s.low = 0; s.range = 0xFFFFFFFFu;
	if (wave_full)
	{
		const u32 rowi = lane < RC_LANES ? lane : RC_LANES - 1;               // idle lanes read a valid row and code nothing
		const LDS_AS U4* row_a = buf_a + rowi * RC_ROW_U4;
		const LDS_AS U4* row_b = buf_b + rowi * RC_ROW_U4;
		LDS_AS u32* crow_a = code_a + rowi * RC_CODE_PITCH;
		LDS_AS u32* crow_b = code_b + rowi * RC_CODE_PITCH;
		RcRegs r0, r1;
		__syncthreads();
		rc_load_group(r0, row_a, 0);
		for (u32 t0 = 0; t0 < wave_full; t0 += 2 * RC_CHUNK)
		{
			rc_chunk(s, crow_a, r0, r1, row_a, t0, n, xb, err, c.force_exact);
			__syncthreads();
			rc_load_group(r0, row_b, 0);
			rc_chunk(s, crow_b, r0, r1, row_b, t0 + RC_CHUNK, n, xb, err, c.force_exact);
			__syncthreads();
			rc_load_group(r0, row_a, 0);
		}
		__syncthreads();                                                       // the loaders have turned the last chunk's codes into bytes
	}
	if (!have) return;
	u32 pos = c.out_byte0;
	if (wave_full) { const u32 v = s_pos[lane]; pos = v & 0x7FFFFFFFu; if (v >> 31) atomicOr(err, (u32)DSRC_ERR_OUT_OVERFLOW); }
	rc_finish(s, c, rec_pool, word_pool, st, n_full, pos, true, xb);
}

__global__ void __launch_bounds__(64 * RC_WG_WAVES) k_rc(const RcChain* chains, u32 n_chains, const u32* list, RcPack* rec_pool, u32* word_pool, BlkState* st)
{
	__shared__ U4 s_a[RC_LANES * RC_ROW_U4];
	__shared__ U4 s_b[RC_LANES * RC_ROW_U4];
	__shared__ U4 s_ca[RC_LANES * RC_CODE_PITCH / 4];
	__shared__ U4 s_cb[RC_LANES * RC_CODE_PITCH / 4];
	__shared__ u32 s_pos[RC_LANES];
	__shared__ u8 s_xb[64 * RC_XB];
	if (list) n_chains = list[0];
	if (blockIdx.x * RC_LANES >= n_chains) return;
	if (blockIdx.x * RC_LANES + RC_LANES <= n_chains) rc_workgroup<true>(chains, n_chains, list, rec_pool, word_pool, st, (LDS_AS U4*)s_a, (LDS_AS U4*)s_b, s_xb, (LDS_AS u32*)s_ca, (LDS_AS u32*)s_cb, (LDS_AS u32*)s_pos);
	else rc_workgroup<false>(chains, n_chains, list, rec_pool, word_pool, st, (LDS_AS U4*)s_a, (LDS_AS U4*)s_b, s_xb, (LDS_AS u32*)s_ca, (LDS_AS u32*)s_cb, (LDS_AS u32*)s_pos);
}

// ---- k_rcs: the two recurrences on two waves ---------------------------------------------------------------------------------------
// LDS of a workgroup: R rows (a, b, freq: 12 B per symbol) x 2, L rows (freq | cum << 16) x 2, R's words x 2, code rows x 2 -- the
// chunk of period p is converted in period p - 1, coded by R in p, by L in p + 1 and turned into bytes in p + 2.
#define RCS_RROW_U4 (LANES * RC_ROW_U4)
#define RCS_LROW_U4 (LANES * RC_CODE_PITCH / 4)
#define RCS_KROW_U4 ((LANES + 1) * RC_CODE_PITCH / 4)      // + a row for the idle lanes of R and L to write to
#ifndef RCS_PROBE
#define RCS_PROBE 0                    // experiments only (wrong output): 1 no byte emission, 2 wave L idle, 4 wave R idle, 8 no conversion
#endif
// the serial waves ask for the next symbols' words BEFORE they code the ones they have: left to itself the instruction scheduler moves
// the reads down to just ahead of their first use, and the LDS latency (next to eight loader waves) shows once per half group
// (wave R busy 3170 -> 3030 clocks per 64-symbol period, k_rcs<32> 77.2 -> 76.3 ms)
#ifdef DSRC_EMU_BUILD
#define RCS_SCHED_FENCE
#else
#define RCS_SCHED_FENCE __builtin_amdgcn_sched_barrier(0);
#endif
#define RCS_DEPTH 6                    // register sets of a loader wave: a chunk is requested RCS_DEPTH - 1 chunk periods before it is converted

// wave R, one symbol: r | (bytes leaving) << 30.  r < 2^30: a row's total is at least the alphabet size (>= 4).
__device__ __forceinline__ u32 rcs_step_r(u32& range_, u32 a, u32 b, u32 freq)
{
	const u32 r = rc_div64(range_, a, b);
	const u32 range = r * freq;
	__builtin_assume(range != 0);
	const u32 k8 = (u32)__builtin_clz(range) & 0x18u;
	range_ = range << k8;
	return r | (k8 << 27);
}

// wave L, one symbol: the code rc_step_fast gives.  The clamp (src/RangeCoder.h:67-71) tests the top byte of low against that of
// low + range with range < 2^24, for the first byte that leaves on low itself (a carry through bits 24..55), for the second on
// low << 8 (bits 16..47 of low): either needs bits 24..47 of low to be all ones.  `flag`: running maximum of bits 16..47 -- its top
// 24 bits are the maximum of bits 24..47 (one alignbit per symbol, no mask); once in 2^24 symbols.
__device__ __forceinline__ u32 rcs_step_l(u64& low_, u32 rk, u32 fc, u32& flag)
{
	const u32 r = rk & 0x3FFFFFFFu, k8 = (rk >> 27) & 0x18u;
	const u64 low = low_ + (u64)r * (fc >> 16);
	const u32 z = __builtin_amdgcn_alignbit((u32)(low >> 32), (u32)low, 16);     // bits 16..47
	flag = z > flag ? z : flag;
	low_ = low << k8;
	return ((u32)(low >> 32) & 0xFFFFFF00u) | k8;
}

// wave L, a chunk in which the pre-condition showed: the reference's loop on low over the chain's `count` symbols of the chunk (r and
// the byte counts are R's); true if the clamp fires (R's range is then not the reference's from that symbol on: the stream goes to
// the redo list).  Not inlined: rare, and the serial wave's loop stays short.
__device__ __attribute__((noinline)) bool rcs_chunk_clamps(u64 low, const LDS_AS u32* rk, const LDS_AS u32* fc, u32 count)
{
#pragma unroll 1
	for (u32 i = 0; i < count; ++i)
	{
		const u32 r = rk[i] & 0x3FFFFFFFu, fci = fc[i];
		low += (u64)r * (fci >> 16);
		u32 range = r * (fci & 0xFFFFu);
#pragma unroll 1
		for (u32 guard = 0; range <= 0x00FFFFFFu && guard < 8; ++guard)
		{
			if ((low ^ (low + range)) & 0xFF00000000000000ull) return true;
			low <<= 8; range <<= 8;
		}
	}
	return false;
}

// Wave L, a chunk q in which the clamp fires (about once in four 450-block batches): R's range is not the reference's from that symbol
// on -- and R is already a chunk further.  The lane walks chunk q again from the state at its start with RangeEncoder::EncodeFrequency
// verbatim (records straight from the stream's array in HBM: nothing of it is left in LDS), deals the bytes out three per code slot
// (the emitter only concatenates them), then codes chunk q + 1 the way R does from the range that is right: its words go to a row of
// their own (`fixrow`: L reads them in place of R's in the next period), the range after them to R, which takes it at the start of
// chunk q + 2.  res: [0,1] low after chunk q, [2] range after chunk q, [3] range after chunk q + 1, [4] 0 = done, 1 = the bytes do
// not fit the chunk's slots (the stream goes to the redo list instead).  Not inlined: rare, and the serial wave's loop stays short.
#define RCS_RXB 200
#define RCS_FIX_SLOTS 8                 // recoveries a workgroup can have under way at a time
__device__ __attribute__((noinline)) void rcs_recover(const RcPack* chain, u32 q, u64 low, u32 range, u32 n_full, LDS_AS u32* crow, LDS_AS u32* fixrow, u8* xb, LDS_AS u32* res)
{
	const u32 t0 = q * RC_CHUNK;
	const u32 cnt = n_full - t0 < RC_CHUNK ? n_full - t0 : RC_CHUNK;
	u32 nb = 0;
#pragma unroll 1
	for (u32 i = 0; i < cnt; ++i)
	{
		const RcRec e = rc_unpack(rc6_load(chain, t0 + i));
		const u32 r = rc_div(range, e.m_lo, e.mf >> 16);
		low += (u64)r * e.cum;
		range = r * (e.mf & 0xFFFFu);
#pragma unroll 1
		for (u32 guard = 0; range <= 0x00FFFFFFu && guard < 8; ++guard)
		{
			if ((low ^ (low + range)) & 0xFF00000000000000ull) { const u32 rr = (u32)low; range = (rr | 0x00FFFFFFu) - rr; }
			if (nb < RCS_RXB) xb[nb] = (u8)(low >> 56);
			++nb;
			low <<= 8; range <<= 8;
		}
	}
	res[0] = (u32)low; res[1] = (u32)(low >> 32); res[2] = range;
	if (nb > 3 * cnt || nb > RCS_RXB) { res[4] = 1; return; }
#pragma unroll 1
	for (u32 i = 0; i < RC_CHUNK; ++i)
	{
		const u32 have = nb > 3 * i ? nb - 3 * i : 0u, take = have < 3 ? have : 3u;
		u32 code = take << 3;
		if (take >= 1) code |= (u32)xb[3 * i] << 24;
		if (take >= 2) code |= (u32)xb[3 * i + 1] << 16;
		if (take >= 3) code |= (u32)xb[3 * i + 2] << 8;
		crow[i] = code;
	}
	const u32 t1 = t0 + RC_CHUNK;
	const u32 cnt1 = n_full > t1 ? (n_full - t1 < RC_CHUNK ? n_full - t1 : RC_CHUNK) : 0u;
#pragma unroll 1
	for (u32 i = 0; i < RC_CHUNK; ++i)
	{
		u32 w = 0;
		if (i < cnt1)
		{
			const RcPack v = rc6_load(chain, t1 + i);
			const u64 m = recip48((u32)(v >> 32));
			w = rcs_step_r(range, (u32)(m >> 16), (u32)m << 16, (u32)v & 0xFFFFu);
		}
		fixrow[i] = w;
	}
	res[3] = range; res[4] = 0;
}

struct RcsLds { LDS_AS U4* r; LDS_AS U4* l; LDS_AS U4* k; LDS_AS U4* c; u8* xb; LDS_AS u32* pos; LDS_AS u32* range; LDS_AS u32* rstart; LDS_AS u32* fix; LDS_AS u32* fixrow; LDS_AS u32* res; u8* rxb; LDS_AS u32* owner; };

template <bool FULL, int LANES>
__device__ __forceinline__ void rcs_workgroup(const RcChain* chains, u32 n_chains, RcPack* rec_pool, u32* word_pool, BlkState* st, const u32* bk, u32* redo, const RcsLds S)
{
	LDS_AS U4* const s_r = S.r; LDS_AS U4* const s_l = S.l; LDS_AS U4* const s_k = S.k; LDS_AS U4* const s_c = S.c;
	u8* const s_xb = S.xb; LDS_AS u32* const s_pos = S.pos; LDS_AS u32* const s_range = S.range;
	__builtin_amdgcn_s_setprio(3);
	constexpr int ROWS = LANES / RC_LOADERS;
	static_assert(ROWS * RC_LOADERS == LANES, "every loader wave feeds the same number of rows");
	const u32 first_chain = blockIdx.x * LANES;
	const u32 lane = lane_id(), id = first_chain + lane, w = wave_id();
	const bool loader = w >= 2;
#if RC_SPARE_SIMD
#if !defined(DSRC_EMU_BUILD) && defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "RC_SPARE_SIMD relies on the gfx9 barrier counting only live waves: build other targets with -DRC_SPARE_SIMD=0"
#endif
	if (loader && (w & 3u) < 2u) return;                                       // the waves that would share R's and L's SIMDs leave before the first barrier
	const u32 loader_id = (w >> 2) * 2u + (w & 1u);
#else
	const u32 loader_id = w - 2u;
#endif
	const bool have0 = lane < (u32)LANES && id < n_chains;
	const RcChain c = chains[have0 ? id : n_chains - 1];                       // idle lanes shadow a real chain's values and code nothing
	// a stream the device handed back to k_sort / k_replay has no records yet: the redo launch codes it
	const bool have = have0 && !(c.bk_on && bk && bk[c.jid]);
	// The loader waves feed all LANES rows, also the ones a partial workgroup (the last of a launch) has no chain for: such a row shadows
	// the launch's last chain (above), its lane codes nothing (n = 0), no byte of it is written (n_full = 0).  A second instantiation
	// of this body that knew its rows only at run time cost the partial workgroup -- and so the launch -- 10-25 ms (its masks and byte
	// positions no longer stayed in scalar registers): 660 / 720 streams 101 / 97 ms against 77 for 800 or 1024.
	const u32 n_live = (u32)LANES;
	(void)FULL;
	const u32 n = have ? c.n : 0;
	const u32 n_full = n & ~(u32)(RC_GROUP - 1);
	const u32 wave_full = (u32)__builtin_amdgcn_readfirstlane((int)wave_max(n_full));      // same value in every wave
	const u32 n_chunks = (wave_full + RC_CHUNK - 1) / RC_CHUNK;

	if (loader)
	{
		if (!wave_full) return;
		const u32 lw = loader_id;
		RcRowBases<ROWS> rb; RcEmitRows<ROWS> R; u32 over = 0;
		rc_rows_setup(rb, R, c, n_full, rec_pool, word_pool, lw, n_live);
		RcFetch r[RCS_DEPTH][ROWS];
#pragma unroll
		for (u32 d = 0; d < RCS_DEPTH; ++d) rc_fetch(r[d], rb, d, lw, n_live);
		// a chunk's R rows and L rows are in buffers (chunk & 1).  The R rows are written a period before R codes the chunk; the L rows
		// -- a copy of the fetched dwords, read by L a period after R -- in R's period, just before the chunk's register set is refilled
		// (written with the R rows they needed three buffers: 8.7 KB of the workgroup's LDS, which with the rest kept a k_part workgroup
		// from sharing the CU)
		auto lrows = [&](const RcFetch* f, u32 l2)
		{
			LDS_AS u32* lbuf = (LDS_AS u32*)(s_l + l2 * RCS_LROW_U4);
#pragma unroll
			for (u32 k = 0; k < (u32)ROWS; ++k)
			{
				const u32 j = lw + k * RC_LOADERS;
				if (j < n_live) lbuf[j * RC_CODE_PITCH + lane_id()] = f[k].fc;
			}
		};
		auto convert = [&](const RcFetch* f, u32 r2)
		{
			LDS_AS U4* rbuf = s_r + r2 * RCS_RROW_U4;
#pragma unroll
			for (u32 k = 0; k < (u32)ROWS; ++k)
			{
				const u32 j = lw + k * RC_LOADERS;
				if (j < n_live)
				{
					u32 ma, mb;
					recip48_ab(f[k].tot, &ma, &mb);
					LDS_AS u32* d = (LDS_AS u32*)(rbuf + j * RC_ROW_U4) + 3u * lane_id();
					d[0] = ma; d[1] = mb; d[2] = f[k].fc & 0xFFFFu;
				}
			}
		};
		convert(r[0], 0);
		__syncthreads();                                                       // chunk 0 is there
		// periods 0 .. n_chunks + 1: R is in chunk p, L in p - 1, the bytes of p - 2 are written, p + 1 is converted.  Unrolled over
		// the register sets: every set and buffer is known statically (a rolled loop that picks the sets with a switch makes the
		// compiler wait for every request in flight, vmcnt(0), once per period: 144 instead of 100 ms)
		static_assert(RCS_DEPTH % 6 == 0, "buffer indices must be static in the unrolled loop");
		const u32 n_per = n_chunks + 2u;
		for (u32 q = 0; q < n_per; q += RCS_DEPTH)
		{
#pragma unroll
			for (u32 k = 0; k < RCS_DEPTH; ++k)
			{
				const u32 p = q + k;                                           // p mod 2 = k mod 2, p mod 3 = k mod 3
				if (p >= n_per) break;
				lrows(r[k], k & 1u);
				rc_fetch(r[k], rb, p + RCS_DEPTH, lw, n_live);
				if (!(RCS_PROBE & 8)) convert(r[(k + 1) % RCS_DEPTH], (k + 1) & 1u);
				if (p >= 2u && !(RCS_PROBE & 1)) rc_emit_chunk((const LDS_AS u32*)(s_c + (k & 1u) * RCS_KROW_U4), R, p - 2u, lw, n_live, &over);      // (p - 2) mod 2
				__syncthreads();
			}
		}
#pragma unroll
		for (u32 k = 0; k < (u32)ROWS; ++k)
			if (lw + k * RC_LOADERS < n_live && lane_id() == 0) s_pos[lw + k * RC_LOADERS] = R.pos[k] | (((over >> k) & 1u) << 31);
		__syncthreads();                                                       // L takes the positions for the chains' tails
		return;
	}

	// R and L have NO lane-dependent control flow in their loops (a guard per group of 16 made the compiler sink the LDS reads under it
	// and cost wave L 25 of 76 clocks per symbol): every lane codes every chunk of the wave's longest chain -- past its own chain's
	// end on whatever lies there -- and keeps the state it had at the end of its own last full group (a select per group).  Idle lanes
	// (RC_LANES .. 63) read the last row and write to a spare one.
	const u32 rowi = lane < (u32)LANES ? lane : (u32)LANES - 1u;
	const u32 rowo = lane < (u32)LANES ? lane : (u32)LANES;
	if (w == 0)
	{	// ---- wave R
		if (!wave_full) return;
		u32 range = 0xFFFFFFFFu, range_end = 0xFFFFFFFFu;
		__syncthreads();
		for (u32 p = 0; p < n_chunks + 2u; ++p)
		{
			if (p < n_chunks && !(RCS_PROBE & 4))
			{
				{	// L's correction for this chunk, if any (a clamp two chunks back: rcs_recover); the range this chunk starts from, for L
					const u32 fp = S.fix[2 * lane], fr = S.fix[2 * lane + 1];
					range = fp == p ? fr : range;
					S.rstart[(p & 1u) * 64u + lane] = range;
				}
				const LDS_AS U4* row = s_r + (p & 1u) * RCS_RROW_U4 + rowi * RC_ROW_U4;
				LDS_AS U4* out = s_k + (p & 1u) * RCS_KROW_U4 + rowo * (RC_CODE_PITCH / 4);
				// eight symbols (six 16-byte reads) at a time, the next eight on their way: two sets of 24 registers
				constexpr u32 H = RC_GROUP / 2, HQ = 3 * H / 4, NH = RC_CHUNK / H;
				U4 ha[HQ], hb[HQ];
#pragma unroll
				for (u32 i = 0; i < HQ; ++i) ha[i] = row[i];
#pragma unroll
				for (u32 hh = 0; hh < NH; ++hh)
				{
					U4* cur = (hh & 1u) ? hb : ha; U4* nxt = (hh & 1u) ? ha : hb;
					if (hh + 1 < NH)
					{
#pragma unroll
						for (u32 i = 0; i < HQ; ++i) nxt[i] = row[(hh + 1) * HQ + i];
					}
					RCS_SCHED_FENCE
					const u32* d = (const u32*)cur;
					u32 v[H];
#pragma unroll
					for (u32 i = 0; i < H; ++i) v[i] = rcs_step_r(range, d[3 * i], d[3 * i + 1], d[3 * i + 2]);
#pragma unroll
					for (u32 i = 0; i < H / 4; ++i) { const U4 x = {v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]}; out[hh * (H / 4) + i] = x; }
					if (hh & 1u) range_end = p * RC_CHUNK + (hh / 2 + 1) * RC_GROUP == n_full ? range : range_end;      // the chain's last full group ends here
				}
			}
			__syncthreads();
		}
		s_range[lane] = range_end;
		__syncthreads();
		return;
	}

	// ---- wave L
	u8* xb = s_xb + lane * RC_XB;
	RcState s;
	s.low = 0; s.range = 0xFFFFFFFFu;
	bool dead = false;
	if (wave_full)
	{
		u64 low = 0, low_end = 0;
		u32 range_fix_end = 0;                                                 // the range at the chain's last full group, if a recovery passed over it
		bool fix_mine = false; u32 fix_start = 0, slot = 0;                    // this lane's words of the chunk come from fix row `slot`; the range that chunk starts from
		__syncthreads();
		for (u32 p = 0; p < n_chunks + 2u; ++p)
		{
			if (p >= 1u && p <= n_chunks && !(RCS_PROBE & 2))
			{
				const u32 q = p - 1u;
				const LDS_AS U4* fcrow = s_l + (q & 1u) * RCS_LROW_U4 + rowi * (RC_CODE_PITCH / 4);
				const LDS_AS U4* krow = fix_mine ? (const LDS_AS U4*)(S.fixrow + slot * RC_CODE_PITCH) : s_k + (q & 1u) * RCS_KROW_U4 + rowi * (RC_CODE_PITCH / 4);
				LDS_AS U4* crow = s_c + (q & 1u) * RCS_KROW_U4 + rowo * (RC_CODE_PITCH / 4);
				// a group's words are requested while the group before it is coded (two sets of registers)
				constexpr u32 GQ = RC_GROUP / 4, NG = RC_CHUNK / RC_GROUP;
				U4 ka[GQ], fa[GQ], kb[GQ], fb[GQ];
#pragma unroll
				for (u32 i = 0; i < GQ; ++i) { ka[i] = krow[i]; fa[i] = fcrow[i]; }
				const u64 snap = low;
				u32 flag = c.force_exact ? 0xFFFFFF00u : 0u;
#pragma unroll
				for (u32 g = 0; g < NG; ++g)
				{
					U4* kq = (g & 1u) ? kb : ka; U4* fq = (g & 1u) ? fb : fa;
					U4* kn = (g & 1u) ? ka : kb; U4* fn = (g & 1u) ? fa : fb;
					if (g + 1 < NG)
					{
#pragma unroll
						for (u32 i = 0; i < GQ; ++i) { kn[i] = krow[(g + 1) * GQ + i]; fn[i] = fcrow[(g + 1) * GQ + i]; }
					}
					RCS_SCHED_FENCE
					const u32* rk = (const u32*)kq; const u32* fc = (const u32*)fq;
					u32 v[RC_GROUP];
#pragma unroll
					for (u32 i = 0; i < RC_GROUP; ++i) v[i] = rcs_step_l(low, rk[i], fc[i], flag);
#pragma unroll
					for (u32 i = 0; i < GQ; ++i) { const U4 x = {v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]}; crow[g * GQ + i] = x; }
					low_end = q * RC_CHUNK + (g + 1) * RC_GROUP == n_full ? low : low_end;
				}
				// the clamp's pre-condition somewhere in the chunk, in some lane's own symbols (once in 2^24 symbols; beyond a chain's end
				// the words are whatever lay there): a branch of the whole wave, once per chunk -- R's words and the L row are still there
				const bool look = !(RCS_PROBE & 16) && (flag >> 8) == 0xFFFFFFu && !dead && q * RC_CHUNK < n_full;
				const bool had_fix = fix_mine;
				bool recovered = false;
				if (__ballot(look))
				{
					const u32 count = n_full - q * RC_CHUNK < RC_CHUNK ? n_full - q * RC_CHUNK : RC_CHUNK;
					// tests: 2 = every stream reports a clamp in its first chunk and goes to the redo list, 3 = a recovery every few chunks (the walk with
					// the reference's loop gives the bytes the fast path gave: nothing changes but the road)
					const bool forced = c.force_exact == 2u || (c.force_exact == 3u && (q + lane) % 5u == 2u);
					if (look && (forced || rcs_chunk_clamps(snap, (const LDS_AS u32*)krow, (const LDS_AS u32*)fcrow, count)))
					{
						// RCS_FIX_SLOTS recoveries at a time per workgroup (a fix row each; with skewed symbols -- four-level qualities -- low keeps its
						// ones for many chunks and a lane holds its row as long): a lane that finds none free, and anything rcs_recover cannot
						// express, goes to the redo list -- k_rc codes the stream again (what this launch still writes for it is overwritten)
						bool mine = had_fix;
						u32 why = c.force_exact == 2u ? 3u : 1u;                   // (the list's top bits, for DSRC_GPU_DEBUG: 1 no fix row free, 2 rcs_recover refused, 3 forced)
						if (!mine && c.force_exact != 2u)
							for (u32 k = 0; k < RCS_FIX_SLOTS && !mine; ++k)
								if (atomicCAS((u32*)(S.owner + k), 0u, lane + 1u) == 0u) { mine = true; slot = k; }
						if (mine)
						{
							const u32 r0 = had_fix ? fix_start : S.rstart[(q & 1u) * 64u + lane];
							LDS_AS u32* res = S.res + slot * 8u;
							rcs_recover(rec_pool + c.trip, q, snap, r0, n_full, (LDS_AS u32*)crow, S.fixrow + slot * RC_CODE_PITCH, S.rxb + slot * (RCS_RXB + 8), res);
							if (res[4] == 0u)
							{
								low = (u64)res[0] | ((u64)res[1] << 32);
								const u32 rq = res[2], rq1 = res[3], end_q = (q + 1u) * RC_CHUNK;
								if (n_full <= end_q) { low_end = low; range_fix_end = rq; }
								else if (n_full <= end_q + RC_CHUNK) range_fix_end = rq1;
								fix_mine = true; fix_start = rq; recovered = true;
								S.fix[2 * lane] = q + 2u; S.fix[2 * lane + 1] = rq1;       // R takes it at the start of chunk q + 2
							}
							else { mine = false; why = 2u; if (!had_fix) S.owner[slot] = 0u; }
						}
						if (!mine)
						{
							dead = true;
							if (!(RCS_PROBE & 64)) { const u32 at = atomicAdd(&redo[0], 1u); redo[1 + at] = id | (why << 28); }
						}
					}
				}
				// the fix row has been read; unless a recovery has just filled it again for the next chunk, it is free
				if (had_fix && !recovered) { fix_mine = false; S.owner[slot] = 0u; }
			}
			__syncthreads();
		}
		__syncthreads();                                                       // the loaders have turned the last chunk's codes into bytes
		s.low = low_end; s.range = range_fix_end ? range_fix_end : s_range[lane];
	}
	if (!have || dead) return;
	u32 pos = c.out_byte0;
	if (wave_full) { const u32 v = s_pos[lane]; pos = v & 0x7FFFFFFFu; if (v >> 31) atomicOr(&st[c.blk].err, (u32)DSRC_ERR_OUT_OVERFLOW); }
	rc_finish(s, c, rec_pool, word_pool, st, n_full, pos, true, xb);
}

template <int LANES> __global__ void __launch_bounds__(64 * RCS_WG_WAVES) k_rcs(const RcChain* chains, u32 n_chains, RcPack* rec_pool, u32* word_pool, BlkState* st, const u32* bk, u32* redo)
{
	if (RCS_PROBE & 128) return;                                               // (experiments: the front end alone)
	__shared__ U4 s_r[2 * RCS_RROW_U4];
	__shared__ U4 s_l[2 * RCS_LROW_U4];
	__shared__ U4 s_k[2 * RCS_KROW_U4];
	__shared__ U4 s_c[2 * RCS_KROW_U4];
	__shared__ u32 s_pos[LANES];
	__shared__ u32 s_range[64];
	__shared__ u32 s_rstart[2 * 64];
	__shared__ u32 s_fix[2 * 64];
	__shared__ u32 s_fixrow[RCS_FIX_SLOTS * RC_CODE_PITCH];
	__shared__ u32 s_res[RCS_FIX_SLOTS * 8];
	__shared__ u8 s_rxb[RCS_FIX_SLOTS * (RCS_RXB + 8)];
	__shared__ u32 s_owner[RCS_FIX_SLOTS];
	if (threadIdx.x < 128) s_fix[threadIdx.x] = 0xFFFFFFFFu;                  // no chunk has this number
	if (threadIdx.x < RCS_FIX_SLOTS) s_owner[threadIdx.x] = 0;
	RcsLds S;
	S.r = (LDS_AS U4*)s_r; S.l = (LDS_AS U4*)s_l; S.k = (LDS_AS U4*)s_k; S.c = (LDS_AS U4*)s_c; S.xb = (u8*)s_r;         /* the tails' byte buffers: the R rows are free by then */ S.pos = (LDS_AS u32*)s_pos; S.range = (LDS_AS u32*)s_range;
	S.rstart = (LDS_AS u32*)s_rstart; S.fix = (LDS_AS u32*)s_fix; S.fixrow = (LDS_AS u32*)s_fixrow; S.res = (LDS_AS u32*)s_res; S.rxb = s_rxb; S.owner = (LDS_AS u32*)s_owner;
	rcs_workgroup<true, LANES>(chains, n_chains, rec_pool, word_pool, st, bk, redo, S);
}

// ---- device self-test of the split coder (dsrcgpu_selftest) --------------------------------------------------------------------------
// Groups of 16 symbols coded three ways from the same state: RangeEncoder::EncodeFrequency verbatim (division by the hardware, the
// clamp where it fires), wave R's steps + wave L's steps, and L's check of a group (rcs_group_clamps).  The states are made to
// sit where the clamp lives -- bits 24..55 (or 16..47) of low all ones, a range that does or does not carry -- which no FASTQ file of
// a test-suite ever reaches.  Counted as mismatches: a clamp the pre-condition flag did not announce, a check that disagrees with the
// reference, bytes or state that differ in a group without a clamp.  *hits counts the groups in which the reference clamped.
__global__ void __launch_bounds__(256) k_selftest_rcs(u32* bad, u32* hits)
{
	const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
	__shared__ u32 s_rk[256][RC_GROUP + 1], s_fc[256][RC_GROUP + 1];
	u64 x = 0x9E3779B97F4A7C15ull * (tid + 1);
	auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (u32)(x >> 21); };
	u32 wrong = 0, clamps = 0;
	for (u32 round = 0; round < 8; ++round)
	{
		// state: a normalised range, low near the clamp in one of three ways
		u32 range0 = (rnd() | 0x01000000u);
		u64 low0 = ((u64)rnd() << 32) | rnd();
		const u32 kind = rnd() % 4u;
		if (kind == 0) low0 |= 0x00FFFFFFFF000000ull;                     // first byte: a carry through bits 24..55
		else if (kind == 1) low0 |= 0x0000FFFFFFFF0000ull;                // second byte: bits 16..47
		else if (kind == 2) low0 |= 0x0000FFFFFF000000ull;                // the pre-condition alone
		u32 tot[RC_GROUP], frq[RC_GROUP], cum[RC_GROUP];
		for (u32 i = 0; i < RC_GROUP; ++i)
		{
			const u32 shape = rnd() % 4u;
			tot[i] = shape == 0 ? 4u + rnd() % 60u : shape == 1 ? 65000u + rnd() % 500u : 4u + rnd() % 65000u;
			frq[i] = shape == 3 ? 1u : 1u + rnd() % tot[i];
			cum[i] = (i < 2 && kind < 3 && (rnd() & 1u)) ? 0u : rnd() % (tot[i] - frq[i] + 1u);       // cum = 0 keeps low where it was put
		}
		// the reference
		u64 low = low0; u32 range = range0; bool clamped = false; u8 ref[3 * RC_GROUP]; u32 nref = 0;
		for (u32 i = 0; i < RC_GROUP; ++i)
		{
			range /= tot[i]; low += (u64)(range * cum[i]); range *= frq[i];
			while (range <= 0x00FFFFFFu)
			{
				if ((low ^ (low + range)) & 0xFF00000000000000ull) { const u32 r = (u32)low; range = (r | 0x00FFFFFFu) - r; clamped = true; }
				if (nref < 3 * RC_GROUP) ref[nref] = (u8)(low >> 56);
				++nref; low <<= 8; range <<= 8;
			}
		}
		// R, then L
		u32 rr = range0, flag = 0; u64 ll = low0; u8 got[3 * RC_GROUP]; u32 ngot = 0;
		for (u32 i = 0; i < RC_GROUP; ++i)
		{
			const u64 m = recip48(tot[i]);
			const u32 rk = rcs_step_r(rr, (u32)(m >> 16), (u32)m << 16, frq[i]);
			const u32 fc = frq[i] | (cum[i] << 16);
			s_rk[threadIdx.x][i] = rk; s_fc[threadIdx.x][i] = fc;
			const u32 code = rcs_step_l(ll, rk, fc, flag);
			for (u32 k = 0; k < ((code >> 3) & 3u); ++k) if (ngot < 3 * RC_GROUP) got[ngot++] = (u8)(code >> (24 - 8 * k));
		}
		const bool announced = (flag >> 8) == 0xFFFFFFu;
		const bool check = rcs_chunk_clamps(low0, (const LDS_AS u32*)s_rk[threadIdx.x], (const LDS_AS u32*)s_fc[threadIdx.x], RC_GROUP);
		if (clamped) { ++clamps; if (!announced || !check) wrong = 1; }
		else
		{
			if (check) wrong = 1;
			if (ngot != nref || ll != low || rr != range) wrong = 1;
			for (u32 k = 0; k < ngot && k < nref; ++k) if (got[k] != ref[k]) wrong = 1;
		}
	}
	if (wrong) atomicAdd(bad, 1u);
	if (clamps) atomicAdd(hits, clamps);
}

// ... and rcs_recover (what wave L does when the clamp fires) against the reference's loop on chunks made the same way: the bytes it
// deals out over the chunk's code slots, the state after the chunk, and the words and the range it hands on for the next chunk.
// scratch: 1 KB per thread (two chunks of records in k_rc's layout).
__global__ void __launch_bounds__(64) k_selftest_rcv(u32* bad, u32* hits, u8* scratch)
{
	const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
	__shared__ u32 s_crow[64][RC_CHUNK + 1], s_fixrow[64][RC_CHUNK + 1], s_res[64][8];
	__shared__ u8 s_rxb[64][RCS_RXB + 8];
	RcPack* chain = (RcPack*)(scratch + (size_t)tid * 1024u);
	u64 x = 0xD1B54A32D192ED03ull * (tid + 1);
	auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (u32)(x >> 21); };
	u32 wrong = 0, clamps = 0;
	for (u32 round = 0; round < 4; ++round)
	{
		const u32 range0 = rnd() | 0x01000000u;
		u64 low0 = ((u64)rnd() << 32) | rnd();
		const u32 kind = rnd() % 3u;
		if (kind == 0) low0 |= 0x00FFFFFFFF000000ull; else if (kind == 1) low0 |= 0x0000FFFFFFFF0000ull;
		const u32 n_full = round == 3 ? 48u + 16u * (rnd() % 5u) : 2 * RC_CHUNK;      // a chain that ends inside the two chunks, too
		for (u32 i = 0; i < 2 * RC_CHUNK; ++i)
		{
			const u32 shape = rnd() % 4u;
			const u32 tot = shape == 0 ? 4u + rnd() % 60u : shape == 1 ? 65000u + rnd() % 500u : 4u + rnd() % 65000u;
			const u32 frq = shape == 3 ? 1u : 1u + rnd() % tot;
			const u32 cum = (i < 3 && (rnd() & 1u)) ? 0u : rnd() % (tot - frq + 1u);
			rc6_store(chain, i, frq, cum, tot);
		}
		// the reference over the first chunk's symbols of the chain
		const u32 cnt = n_full < RC_CHUNK ? n_full : RC_CHUNK;
		u64 low = low0; u32 range = range0; bool clamped = false; u8 ref[RCS_RXB]; u32 nref = 0;
		for (u32 i = 0; i < cnt; ++i)
		{
			const RcPack v = rc6_load(chain, i);
			range /= (u32)(v >> 32); low += (u64)(range * (((u32)v >> 16) & 0xFFFFu)); range *= (u32)v & 0xFFFFu;
			while (range <= 0x00FFFFFFu)
			{
				if ((low ^ (low + range)) & 0xFF00000000000000ull) { const u32 r = (u32)low; range = (r | 0x00FFFFFFu) - r; clamped = true; }
				if (nref < RCS_RXB) ref[nref] = (u8)(low >> 56);
				++nref; low <<= 8; range <<= 8;
			}
		}
		rcs_recover(chain, 0, low0, range0, n_full, (LDS_AS u32*)s_crow[threadIdx.x], (LDS_AS u32*)s_fixrow[threadIdx.x], s_rxb[threadIdx.x], (LDS_AS u32*)s_res[threadIdx.x]);
		const u32* res = s_res[threadIdx.x];
		if (clamped) ++clamps;
		if (res[4] == 0)
		{
			if ((((u64)res[1] << 32) | res[0]) != low || res[2] != range) wrong = 1;
			u32 ngot = 0;
			for (u32 i = 0; i < RC_CHUNK; ++i)
			{
				const u32 code = s_crow[threadIdx.x][i];
				for (u32 k = 0; k < ((code >> 3) & 3u); ++k) { if (ngot >= nref || ref[ngot] != (u8)(code >> (24 - 8 * k))) wrong = 1; ++ngot; }
				if (i >= cnt && ((code >> 3) & 3u)) wrong = 1;                       // no bytes in the slots behind the chain's end
			}
			if (ngot != nref) wrong = 1;
			u32 rr = range;
			for (u32 i = 0; i < RC_CHUNK; ++i)
			{
				u32 want = 0;
				if (RC_CHUNK + i < n_full) { const RcPack v = rc6_load(chain, RC_CHUNK + i); const u64 m = recip48((u32)(v >> 32)); want = rcs_step_r(rr, (u32)(m >> 16), (u32)m << 16, (u32)v & 0xFFFFu); }
				if (s_fixrow[threadIdx.x][i] != want) wrong = 1;
			}
			if (res[3] != rr) wrong = 1;
		}
		else if (nref <= 3 * cnt && nref <= RCS_RXB) wrong = 1;                       // refused although the bytes fit
	}
	if (wrong) atomicAdd(bad, 1u);
	if (clamps) atomicAdd(hits, clamps);
}

// ---- stream prologues ---------------------------------------------------------------------------
// quality, lossless order model: scheme byte + 256-bit presence map of the raw quality values
// (IQualityModelerProxy::Encode, src/QualityModelerProxy.h:48-58; TTranslationalQualityEncoder::Store,
// src/QualityEncoder.h:332-342); lossy: nothing.  DNA: scheme byte (src/DnaModelerProxy.h:50-60).
__global__ void __launch_bounds__(64) k_rc_headers(const CtxJob* jobs, u32 n_jobs, const BlkState* st, u32* word_pool)
{
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_jobs) return;
	const CtxJob j = jobs[i];
	u32* out = word_pool + j.out_words;
	u8* ob = (u8*)out;                      // range-coder streams are staged as plain bytes
	if (j.is_dna) { ob[0] = (u8)j.scheme; return; }
	if (!j.translate) return;
	ob[0] = (u8)j.scheme;
	const BlkState* S = &st[j.blk];
	for (u32 k = 0; k < 32; ++k)
	{
		u32 v = 0;
		for (u32 b = 0; b < 8; ++b) v = (v << 1) | (S->q_sym[8 * k + b] != 255 ? 1u : 0u);
		ob[1 + k] = (u8)v;
	}
}

// blocks whose DNA stream is empty: a single scheme byte 255 (SchemeNone)
__global__ void __launch_bounds__(64) k_dna_none(const BlkDesc* desc, BlkState* st, u32* word_pool, u32 n_blocks)
{
	const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_blocks || desc[b].d_scheme != 255) return;
	put_byte(word_pool + desc[b].dna_out, 0, 255);
	st[b].dna_bytes = 1;
}

__global__ void __launch_bounds__(64) k_init_state(BlkState* st, u32 n_blocks)
{
	const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b < n_blocks) { st[b].first_bad = 0xFFFFFFFFu; st[b].min_len = 0xFFFFFFFFu; }      // (min_len: k_prep_stats in several parts folds with atomicMin)
}
