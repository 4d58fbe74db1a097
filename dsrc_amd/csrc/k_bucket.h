// Bucketed context path (round 4): the order-k model statistics of a stream without a sort -- 4-byte elements grouped tile by
// tile, counter rows in LDS -- and records that leave in runs instead of one 32-byte sector per 8-byte record.
//   TDnaRCOrderModeler::UpdateHash / EncodeSymbol    src/DnaModelerRCO.h:94-131
//   TQualityModelBase::UpdateHash, TQualityModelExt  src/QualityEncoder.h:77-94,126-151
//   TSymbolCoderRC::EncodeSymbol/Accumulate/Rescale   src/SymbolCoderRC.h:35-90
//
// k_sort + k_replay (k_rc.h) sort a stream's (context, symbol, t) elements by context with two LSD passes through HBM, replay
// the sorted array and scatter one 8-byte record per symbol to stream order.  Here:
//   k_part   : every tile of BK_BIN = 8192 consecutive symbols is grouped, in place, by the top `hb` bits of a *mixed* context key
//              (key = ctx * odd constant mod 2^K: a bijection, so equal keys <=> equal contexts, and the top bits spread hot
//              neighbourhoods of the context space over all buckets): <= 1024 buckets, stable inside a tile; one workgroup per
//              tile, no pass over the stream before it; the element is 4 bytes (low key bits, symbol, t mod 8192);
//   k_binoff : per tile the exclusive scan of the elements per bucket: where a bucket's elements lie inside the tile;
//   k_model  : one WAVE per bucket: the adaptive counter rows of the bucket's contexts live in LDS (the reference's model table,
//              1/1024 of it at a time) and every symbol is coded on its row with one returning LDS atomic per trie level, in
//              stream order: the wave reads its bucket tile by tile and writes the record of an element over the element's
//              index in the stream's record array (the record carries the low bits of t);
//   k_place  : one workgroup per (stream, tile = time bin): the bin's 8 K records are put in stream order through LDS, in place.
// Per symbol that is 4 B written (elements), 4 B read + 8 B written (model), 8 B + 8 B (place), all in runs, against
// 8 + 16 + 8 B in runs plus one 32-byte sector per record before -- and no sort at all: the LSD passes, the
// segmented-scan replay and its seams are replaced by log2(N) LDS atomics per symbol.
//
// A wave walks its bucket serially, so a bucket must not grow without bound: a stream with a bucket beyond BK_LIMIT elements (a
// context that holds a sixth of a 3 M-symbol stream or more), or with more contexts in one bucket than the wave has rows, is handed
// to k_sort / k_replay_seams / k_replay, whose range-splitting replay is made for exactly that: k_model appends it to
// the fallback list of its launch group (k_part: the streams too short or too long for the path).  Either way the records k_rc
// reads are the same.
#pragma once
#include <type_traits>
#include "k_rc.h"

#define BK_MAX_HB 10                   // bucket digit: <= 1024 buckets (k_part's LDS is k_sort's)
#define BK_MAX_LB 11                   // key bits left inside a bucket (k_model's key -> row map)
#define BK_LIMIT 524288                // largest bucket one wave is allowed to walk (8 K windows, a few ms)
#define BK_BIG 16384                   // a bucket from which k_model looks for windows inside one tile's run first
#ifndef BK_TB
#define BK_TB 13                       // log2(records per time bin): 48 bits of record + the low bits of t fit the 8-byte slot; k_place holds a bin in LDS
#endif                                 // (64 KB: with 14 bits and 128 KB a k_place workgroup needs a CU nearly to itself -- 1.5 ms alone, 6.9 ms next to three other instances)
#define BK_BIN (1u << BK_TB)
static_assert(BK_TB == RC6_TB, "a time bin is the unit of k_rc's record layout (rc6_chunk_off)");
#ifndef BK_MAX_BINS
#define BK_MAX_BINS 512                // streams of up to 4 M symbols keep their tiles' offsets and counts of a bucket in k_model's LDS;
#endif
#define BK_MAX_BINS_WIDE 65536         // longer ones (-b64, -b256: 27 M / 107 M symbols per stream) read them from the count table 64 tiles at a time
#define BK_HASH_MUL 0x9E3779B1u

// the `bk` pool (u32): [0 .. NJ) fallback flag per job | fallback lists, one per launch group: count, then job ids; zeroed per
// batch.  `bcnt` (u16): per job and tile the elements per bucket (k_part), turned by k_binoff into every bucket's offset inside the tile.

__device__ __forceinline__ u64 bk_rekey(const CtxJob& j, u64 el)
{
	const u32 key = ((u32)(el >> ELEM_CTX_SHIFT) * j.bk_mul) & j.bk_kmask;
	return (el & ((1ull << ELEM_CTX_SHIFT) - 1ull)) | ((u64)key << ELEM_CTX_SHIFT);
}

__device__ __forceinline__ void bk_fallback(const CtxJob& j, u32* bk)
{
	if (atomicExch(&bk[j.jid], 1u) != 0u) return;
	const u32 k = atomicAdd(&bk[j.bk_fb], 1u);
	bk[j.bk_fb + 1 + k] = j.jid;
}

// Elements of EIGHT consecutive symbols t0 .. t0+7 (t0 >= 16, t0 + 7 < n): their contexts overlap in all but one symbol, so the
// windows are loaded (and, for qualities, translated to ranks) once -- 2 table look-ups per symbol instead of order + 2, three
// 8-byte loads per eight symbols instead of two per symbol.  Same values as ctx_elem_dna / ctx_elem_qua followed by bk_rekey.
__device__ __forceinline__ bool bk_fast8(const CtxJob& j) { return j.is_dna ? (j.alpha_bits == 2 && j.order <= 9) : j.order <= 4; }

// the windows of eight consecutive symbols: DNA s[t0-9 ..], s[t0-1 ..], s[t0 ..]; quality s[t0-8 ..], s[t0 ..] (8 bytes each)
struct BkWin { u64 a, b, c; };
__device__ __forceinline__ BkWin bk_load8(const CtxJob& j, const u8* s, u32 t0)
{
	BkWin w;
	if (j.is_dna) { w.a = *(const u64_unaligned*)(s + t0 - 9); w.b = *(const u64_unaligned*)(s + t0 - 1); w.c = *(const u64_unaligned*)(s + t0); }
	else { w.a = *(const u64_unaligned*)(s + t0 - 8); w.b = 0; w.c = *(const u64_unaligned*)(s + t0); }
	return w;
}

__device__ __forceinline__ void bk_elems8_dna(const CtxJob& j, const BkWin& w, u32 t0, u64* el, bool* bad)
{
	const u64 wa = w.a, wb = w.b, wc = w.c;
	if ((wa | wc) & 0xFCFCFCFCFCFCFCFCull) *bad = true;
	const u32 P = (pack2x8(__builtin_bswap64(wa)) << 16) | pack2x8(__builtin_bswap64(wb));      // s[t0-9] on top, s[t0+6] at the bottom
	const u32 cmask = (u32)((1ull << (2 * j.order)) - 1ull);
#pragma unroll
	for (u32 i = 0; i < 8; ++i)
	{
		const u32 ctx = (P >> (14 - 2 * i)) & cmask;
		const u32 key = (ctx * j.bk_mul) & j.bk_kmask;
		el[i] = ((u64)key << ELEM_CTX_SHIFT) | ((u64)((u32)(wc >> (8 * i)) & 3u) << ELEM_SYM_SHIFT) | (t0 + i);
	}
}

__device__ __forceinline__ void bk_elems8_qua(const CtxJob& j, const BkWin& w, const u8* qp, const u8* rank, u32 t0, u64* el)
{
	const u32 ab = j.alpha_bits, order = j.order, half = order / 2;
	const u64 w0 = w.a, w1 = w.c;
	u32 r[16];                                              // r[m] = rank of s[t0 - 8 + m]
#pragma unroll
	for (u32 m = 0; m < 8; ++m) { r[m] = rank[(u32)(w0 >> (8 * m)) & 0xFFu]; r[8 + m] = rank[(u32)(w1 >> (8 * m)) & 0xFFu]; }
	u32 pos = j.qlen ? t0 - exact_div(t0, j.qm_lo, j.qm_hi) * j.qlen : 0u;
#pragma unroll
	for (u32 i = 0; i < 8; ++i)
	{	// v[k] = rank of s[t-1-k] = r[7 + i - k]
		u32 h = 0;
#pragma unroll
		for (u32 k = 4; k >= 1; --k)
			if (k <= order)
			{
				const u32 slot = k - 1;
				const u32 x = (slot < half || order == 1) ? r[7 + i - slot] : ((r[7 + i - slot] + r[6 + i - slot]) >> 1);
				h = (h << ab) | x;
			}
		const u32 pctx = (j.qlen ? exact_div(pos * 128u, j.qm_lo, j.qm_hi) : (u32)qp[t0 + i]) >> j.rescale_shift;
		const u32 key = ((((h << ab) | pctx)) * j.bk_mul) & j.bk_kmask;
		el[i] = ((u64)key << ELEM_CTX_SHIFT) | ((u64)(r[8 + i] & ((1u << ab) - 1u)) << ELEM_SYM_SHIFT) | (t0 + i);
		pos = pos + 1 == j.qlen ? 0u : pos + 1;
	}
}

// ---- k_part: a tile's elements grouped by the top digit of the mixed key ---------------------------------------------------------------
// Grid: x = tile (= time bin: BK_BIN consecutive symbols), y = stream of the slice.  The workgroup makes the tile's elements, ranks
// them by bucket (stable; as in k_sort<.., true>: one LDS atomic per element on the wave's packed counter pair -- the path is only
// taken on devices that passed k_lds_order_test) and writes them, grouped by bucket, over the tile's own place of the element
// array: one contiguous 32 KB piece per workgroup, no pass over the stream before it and nothing carried from tile to tile.  The
// element is 4 bytes -- the bucket is where it lies, the tile is where it lies: what is left is the low key bits, the symbol and
// the 13 low bits of t.  Per tile the elements per bucket go to `bcnt` (k_binoff turns them into the buckets' offsets inside the tile,
// which k_model reads the elements by and writes the records by).
#ifndef PART_ITEMS
#define PART_ITEMS 16                  // elements per thread (a multiple of 8): the workgroup is BK_BIN / PART_ITEMS = 512 threads with 48 KB of LDS.
                                       // Next to other instances' kernels a workgroup of 1024 threads and 66 KB waits long for a CU with that much free at once
                                       // (k_part of 128 streams: 1.4 ms alone, 5.5 ms in the 4-instance bench; with 512 threads 1.0 and 2.9 ms)
#endif
#define PART_WG (BK_BIN / PART_ITEMS)
#ifndef PART_PASSES
#define PART_PASSES 1                  // the tile leaves through LDS in this many pieces (two of 16 KB: 32 KB of LDS per workgroup instead of 48 -- no faster, alone or in the bench)
#endif
#define PART_WAVES (PART_WG / 64)
#define BK_EL_SYM_SHIFT BK_TB
#define BK_EL_KEY_SHIFT (BK_TB + 7)
static_assert(BK_EL_KEY_SHIFT + BK_MAX_LB <= 32, "low key bits, symbol and time inside the tile in 32 bits");
__device__ __forceinline__ u32 bk_keysym(u64 e) { return ((u32)(e >> (ELEM_SYM_SHIFT + 1)) & ~0x7Fu) | ((u32)(e >> ELEM_SYM_SHIFT) & 0x7Fu); }      // mixed key << 7 | symbol
static_assert(ELEM_CTX_SHIFT == ELEM_SYM_SHIFT + 8, "bk_keysym");

__global__ void __launch_bounds__(PART_WG) k_part(const CtxJob* jobs, u64* pool, const u8* d_stream, const u8* q_stream, const u8* qp_stream, BlkState* st, u32* bk, u16* bcnt)
{
	__shared__ u16 s_cnt[PART_WAVES][SORT_MAX_BINS];           // per wave: elements per bucket, then the wave's offset of the bucket inside the tile
	__shared__ u32 s_tile[BK_BIN / PART_PASSES];                // the tile in bucket order, PART_PASSES pieces one after the other
	__shared__ u32 s_ws[PART_WAVES];
	__shared__ u8 s_rank[256];
	constexpr u32 tile_elems = PART_WG * PART_ITEMS;
	const CtxJob j = jobs[blockIdx.y];
	if (!j.bk_on) { if (blockIdx.x == 0 && threadIdx.x == 0) bk_fallback(j, bk); return; }
	const u32 tile = blockIdx.x * tile_elems;
	if (tile >= j.n) return;
	const u32 n = j.n, bins = 1u << j.bk_hb, lb = j.bk_lb;
	const u32 wv = wave_id(), lane = lane_id();
	const u8* sym_src = (j.is_dna ? d_stream : q_stream) + j.src_off;
	const u8* qp = qp_stream + j.src_off;

	for (u32 i = threadIdx.x; i < 256; i += blockDim.x) s_rank[i] = (!j.is_dna && j.translate) ? st[j.blk].q_sym[i] : (u8)i;
	for (u32 i = threadIdx.x; i < PART_WAVES * SORT_MAX_BINS / 2; i += blockDim.x) ((u32*)&s_cnt[0][0])[i] = 0;
	__syncthreads();

	// ks[k]: mixed key << 7 | symbol of element wbase + 64 k + lane
	u32 ks[PART_ITEMS], rk[PART_ITEMS];
	const u32 wbase = tile + wv * 64 * PART_ITEMS;
	bool bad = false;
	if (bk_fast8(j) && tile > 0 && tile + tile_elems + 8 <= n)
	{	// an inner tile: every lane makes the elements of eight consecutive symbols, and the wave's 512 elements change places
		// through its strip of s_tile (unused until the tile is ranked) so that lane l holds elements l, l + 64, ...: the order
		// the ranking needs
		static_assert(PART_ITEMS % 8 == 0, "eight consecutive symbols per lane and round");
		static_assert(PART_WAVES * 512 <= BK_BIN / PART_PASSES, "a strip of 512 elements per wave");
		u32* strip = s_tile + wv * 512;
		BkWin w = bk_load8(j, sym_src, wbase + 8 * lane);
#pragma unroll
		for (u32 h = 0; h < PART_ITEMS / 8; ++h)
		{	// 512 consecutive elements of the wave per round (the next round's windows are on their way)
			u64 e8[8];
			const u32 t0 = wbase + 512 * h + 8 * lane;
			const BkWin wc = w;
			if (h + 1 < PART_ITEMS / 8) w = bk_load8(j, sym_src, t0 + 512);
			if (j.is_dna) bk_elems8_dna(j, wc, t0, e8, &bad); else bk_elems8_qua(j, wc, qp, s_rank, t0, e8);
			if (h) wave_fence();
#pragma unroll
			for (u32 k = 0; k < 8; ++k) strip[8 * lane + k] = bk_keysym(e8[k]);
			wave_fence();
#pragma unroll
			for (u32 k = 0; k < 8; ++k) ks[8 * h + k] = strip[64 * k + lane];
		}
	}
	else
	{
#pragma unroll
		for (u32 k = 0; k < PART_ITEMS; ++k)
		{
			const u32 i = wbase + k * 64 + lane;
			ks[k] = 0;
			if (i < n)
			{
				const u64 e = bk_rekey(j, j.is_dna ? ctx_elem_dna(j, sym_src, i, &bad) : ctx_elem_qua(j, sym_src, qp, s_rank, i));
				ks[k] = bk_keysym(e);
			}
		}
	}
	if (bad) atomicOr(&st[j.blk].err, (u32)DSRC_ERR_REF_UB);
	const u32 shift = 7 + lb;
#pragma unroll
	for (u32 k = 0; k < PART_ITEMS; ++k)
	{
		const u32 i = wbase + k * 64 + lane;
		const u32 d = ks[k] >> shift;
		const u32 sh = (d & 1u) * 16u;
		u32 old = 0;
		if (i < n) old = atomicAdd(&((u32*)s_cnt[wv])[d >> 1], 1u << sh);
		rk[k] = (old >> sh) & 0xFFFFu;
#ifdef DSRC_EMU_BUILD
		(void)__ballot(true);                                     // the emulator runs lanes one after the other: keep them in step per k
#endif
	}
	__syncthreads();                                               // the strips are read, the counts are complete
	for (u32 d0 = 0, carry = 0; d0 < bins; d0 += PART_WG)
	{
		const u32 dd = d0 + threadIdx.x;
		u32 c[PART_WAVES], tot = 0;
#pragma unroll
		for (u32 w = 0; w < PART_WAVES; ++w) { c[w] = dd < bins ? s_cnt[w][dd] : 0u; tot += c[w]; }
		const u32 inc = wave_incl_scan_dpp(tot);
		if (lane == 63) s_ws[wv] = inc;
		__syncthreads();
		u32 run = carry + inc - tot;
		const u32 nws = (bins - d0 + 63) / 64 < PART_WAVES ? (bins - d0 + 63) / 64 : PART_WAVES;
		for (u32 i = 0; i < nws; ++i) { const u32 x = s_ws[i]; run += i < wv ? x : 0u; carry += x; }
		if (dd < bins)
		{
			bcnt[(u64)j.bk_cnt + (u64)blockIdx.x * bins + dd] = (u16)tot;      // the tile's elements per bucket (k_binoff)
#pragma unroll
			for (u32 w = 0; w < PART_WAVES; ++w) { s_cnt[w][dd] = (u16)run; run += c[w]; }
		}
		if (d0 + PART_WG < bins) __syncthreads();
	}
	__syncthreads();
	const u32 lbmask = (1u << lb) - 1u;
	u32* dst = (u32*)(pool + j.elems) + tile;
	const u32 tile_n = n - tile < tile_elems ? n - tile : tile_elems;
	constexpr u32 PIECE = BK_BIN / PART_PASSES;
	// where every element goes inside the tile (the ranks' registers are free now)
#pragma unroll
	for (u32 k = 0; k < PART_ITEMS; ++k) rk[k] += (u32)s_cnt[wv][ks[k] >> shift];
#pragma unroll
	for (u32 p = 0; p < PART_PASSES; ++p)
	{
#pragma unroll
		for (u32 k = 0; k < PART_ITEMS; ++k)
		{
			const u32 i = wbase + k * 64 + lane;
			const u32 at = rk[k] - p * PIECE;
			if (i < n && at < PIECE)
				s_tile[at] = (((ks[k] >> 7) & lbmask) << BK_EL_KEY_SHIFT) | ((ks[k] & 0x7Fu) << BK_EL_SYM_SHIFT) | (i & (BK_BIN - 1u));
		}
		__syncthreads();
		if (tile_n == tile_elems)
		{
#pragma unroll
			for (u32 k = 0; k < PIECE / 2 / PART_WG; ++k) ((u64*)(dst + p * PIECE))[k * PART_WG + threadIdx.x] = ((const u64*)s_tile)[k * PART_WG + threadIdx.x];      // the array is 8-byte aligned
		}
		else for (u32 q = threadIdx.x; q < PIECE && p * PIECE + q < tile_n; q += PART_WG) dst[p * PIECE + q] = s_tile[q];
		if (p + 1 < PART_PASSES) __syncthreads();
	}
}

// ---- k_model: a bucket's model statistics on counter rows in LDS -------------------------------------------------------------------
// One WAVE per bucket, no barriers.  The bucket's elements arrive in stream order (k_part is stable); 64 at a time, lane i takes
// element i.  A context met for the first time gets the next free counter row of the wave's LDS (key -> row through a map).  A row
// is the N counters of TSymbolCoderRC<N> kept as a radix-4 trie: a 64-bit word per node with the four 16-bit sums of the counters
// under its children (MdRow).  Coding symbol s then is ONE returning LDS atomic per level -- add 2 to the field s falls in -- and what
// comes back gives everything the range coder needs as of this element: total = the four fields of the root, cum = the fields
// below s's own at every level, freq = its own field at the last level (2-bit bases: one atomic per symbol, 32 quality values: three).
// The LDS applies the lanes of one atomic instruction that meet in a word in lane order (k_lds_order_test / k_lds_order_test64
// measure exactly that before this path is ever used), and a wave's instructions in program order: lane order is stream order.
// Rescale() (a row's total reaches 2^16 - 2N, src/SymbolCoderRC.h:67-90) needs the exact serial order: in buckets large enough
// for it (> 32 K symbols) a window in which some lane's row could get there is coded one element at a time, and the lane whose row
// is due halves its counters and rebuilds the sums first.
// The records leave grouped by time bin: the elements of a bin are consecutive (the bucket is in stream order) and go, as a run,
// to the place k_binoff worked out for this bucket inside the bin's region of the record array.
// waves per workgroup: the LDS a wave needs decides how many waves a CU holds, in steps of a workgroup (measured: 32-symbol rows of
// 8 KB 2.88 ms with two waves against 3.20 with four; 4-symbol rows of 4 KB 2.11 with four against 2.38 with two)
#define MD_WAVES_FOR(ROW_BYTES) ((ROW_BYTES) <= 4096 ? 4 : 2)
#define MD_ROW_BYTES 8192              // counter rows per wave: 64 rows of a 32-symbol alphabet
#define MD_NONE 0xFFFFFFFFu

// A row: the N counters of TSymbolCoderRC<N> as a radix-4 trie of 64-bit words with four 16-bit sums each (sums of the counters
// under each of the four children), plus a last level of 32-bit pair words when log2(N) is odd.
template <int N> struct MdRow
{
	static constexpr int B = N <= 4 ? 2 : N <= 8 ? 3 : N <= 16 ? 4 : N <= 32 ? 5 : N <= 64 ? 6 : 7;
	static constexpr int L4 = B / 2;                          // radix-4 levels
	static constexpr bool R2 = (B & 1) != 0;
	static constexpr u32 W64 = ((1u << (2 * L4)) - 1u) / 3u;  // 64-bit words of the radix-4 levels: 1 + 4 + 16
	static constexpr u32 STRIDE = 2 * W64 + (R2 ? N / 2 : 0); // row pitch in u32 (a multiple of 2: rows stay 8-byte aligned)
};

// every counter 1
template <int N> __device__ __forceinline__ u32 md_init_word(u32 x)
{
	typedef MdRow<N> G;
	if (x >= 2 * G::W64) return 0x00010001u;                 // pair words: two counters
	const u32 w = x >> 1;                                    // 64-bit word: level l holds words (4^l - 1) / 3 ..; a field covers N / 4^(l+1) counters
	const u32 l = w >= 5 ? 2u : w >= 1 ? 1u : 0u;
	const u32 v = (u32)N >> (2 * (l + 1));
	return v | (v << 16);
}

// the row's total (the root's four fields)
template <int N> __device__ __forceinline__ u32 md_total(const u32* row)
{
	const u64 v = *(const unsigned long long*)row;
	return ((u32)v & 0xFFFFu) + ((u32)v >> 16) + ((u32)(v >> 32) & 0xFFFFu) + (u32)(v >> 48);
}

// TSymbolCoderRC::Rescale (src/SymbolCoderRC.h:80-90): every counter x becomes x - (x >> 1); the sums above them are rebuilt.
// One lane, rarely (a context needs ~32 K symbols to get here).
template <int N> __device__ __forceinline__ void md_rescale(u32* row)
{
	typedef MdRow<N> G;
	unsigned long long* r64 = (unsigned long long*)row;
	if (G::R2)
		for (u32 x = 0; x < (u32)N / 2; ++x)
		{
			const u32 v = row[2 * G::W64 + x];
			u32 a = v & 0xFFFFu, b = v >> 16;
			a -= a >> 1; b -= b >> 1;
			row[2 * G::W64 + x] = a | (b << 16);
		}
	for (int l = G::L4 - 1; l >= 0; --l)
	{
		const u32 base = ((1u << (2 * l)) - 1u) / 3u, nw = 1u << (2 * l);
		for (u32 w = 0; w < nw; ++w)
		{
			u64 v = 0;
			if (!G::R2 && l == G::L4 - 1)
			{	// the fields are the counters themselves
				const u64 o = r64[base + w];
#pragma unroll
				for (u32 q = 0; q < 4; ++q) { u32 x = (u32)(o >> (16 * q)) & 0xFFFFu; x -= x >> 1; v |= (u64)x << (16 * q); }
			}
			else
			{
#pragma unroll
				for (u32 q = 0; q < 4; ++q)
				{
					const u32 c = 4 * w + q;
					u32 sum;
					if (l == G::L4 - 1) { const u32 p = row[2 * G::W64 + c]; sum = (p & 0xFFFFu) + (p >> 16); }
					else { const u64 o = r64[((1u << (2 * (l + 1))) - 1u) / 3u + c]; sum = ((u32)o & 0xFFFFu) + ((u32)o >> 16) + ((u32)(o >> 32) & 0xFFFFu) + (u32)(o >> 48); }
					v |= (u64)sum << (16 * q);
				}
			}
			r64[base + w] = v;
		}
	}
}

// one symbol on its row: returns freq | cum << 16 | total << 32
template <int N> __device__ __forceinline__ u64 md_code(u32* row, u32 sym)
{
	typedef MdRow<N> G;
	unsigned long long* r64 = (unsigned long long*)row;
	u32 cum = 0, tot = 0, f = 0;
#pragma unroll
	for (int l = 0; l < G::L4; ++l)
	{
		const u32 q = (sym >> (G::B - 2 * l - 2)) & 3u;
		const u32 wi = ((1u << (2 * l)) - 1u) / 3u + (l ? sym >> (G::B - 2 * l) : 0u);
		const u64 old = atomicAdd(&r64[wi], 2ull << (16 * q));
		const u64 below = old & ((1ull << (16 * q)) - 1ull);
		cum += ((u32)below & 0xFFFFu) + ((u32)below >> 16) + ((u32)(below >> 32) & 0xFFFFu);
		if (l == 0) tot = ((u32)old & 0xFFFFu) + ((u32)old >> 16) + ((u32)(old >> 32) & 0xFFFFu) + (u32)(old >> 48);
		f = (u32)(old >> (16 * q)) & 0xFFFFu;
	}
	if (G::R2)
	{
		const u32 old = atomicAdd(&row[2 * G::W64 + (sym >> 1)], (sym & 1u) ? (2u << 16) : 2u);
		cum += (sym & 1u) ? (old & 0xFFFFu) : 0u;
		f = (sym & 1u) ? (old >> 16) : (old & 0xFFFFu);
	}
	return (u64)f | ((u64)cum << 16) | ((u64)tot << 32);
}

// ---- k_binoff: where a bucket's elements (and records) of a tile lie ------------------------------------------------------------------
// k_part left the number of elements per (tile, bucket).  Per tile the exclusive scan over the buckets gives every bucket's place
// inside the tile, written over the counts.  Grid: x = tile, y = stream of the slice; 256 threads, four buckets each.
__global__ void __launch_bounds__(256) k_binoff(const CtxJob* jobs, u16* bcnt, const u32* bk)
{
	const CtxJob j = jobs[blockIdx.y];
	const u32 bin = blockIdx.x;
	if (!j.bk_on || bk[j.jid] || (bin << BK_TB) >= j.n) return;
	const u32 bins = 1u << j.bk_hb;
	u16* row = bcnt + j.bk_cnt + (u64)bin * bins;
	u32 c[4], mine = 0;
#pragma unroll
	for (u32 k = 0; k < 4; ++k)
	{
		const u32 b = 4 * threadIdx.x + k;
		c[k] = b < bins ? (u32)row[b] : 0u;
		mine += c[k];
	}
	u32 tot;
	u32 run = block_excl_scan(mine, &tot);
#pragma unroll
	for (u32 k = 0; k < 4; ++k)
	{
		const u32 b = 4 * threadIdx.x + k;
		if (b < bins) row[b] = (u16)run;
		run += c[k];
	}
}

template <bool SMALL> struct MdMapT { typedef u8 T; static constexpr u32 NONE = 0xFFu, CLAIM = 0x80u; };
template <> struct MdMapT<false> { typedef u16 T; static constexpr u32 NONE = 0xFFFFu, CLAIM = 0x8000u; };

// Grid: x = bucket / MD_WAVES, y = stream of the launch group (one alphabet size).  MAPBITS = 0: every stream of the group has at
// most ROWS keys per bucket, key k owns row k; otherwise MAPBITS >= the group's largest bk_lb and rows are handed out on first use.
// ROW_BYTES: LDS per wave for the rows (the fewer, the more buckets a CU works on at a time).
#ifndef MD_AHEAD
#define MD_AHEAD 3
#endif
#ifndef MD_ROTATE_ALL
#define MD_ROTATE_ALL 0
#endif
//      MD_AHEAD:                     // windows of a bucket requested ahead of the one being coded
template <int N, int MAPBITS, int ROW_BYTES>
__global__ void __launch_bounds__(64 * MD_WAVES_FOR(ROW_BYTES)) k_model(const CtxJob* jobs, const u64* pool, RcPack* rec_pool, u32* bk, const u16* bcnt, RcPack* nowhere)
{
	constexpr u32 STRIDE = MdRow<N>::STRIDE;
	constexpr u32 ROWS = ROW_BYTES / (4 * STRIDE);
	constexpr u32 MD_WAVES = MD_WAVES_FOR(ROW_BYTES);
	typedef MdMapT<(ROWS <= 127)> Map;
	typedef typename Map::T map_t;
	__shared__ map_t s_map[MD_WAVES][1 << MAPBITS];
	__shared__ unsigned long long s_rows[MD_WAVES][ROW_BYTES / 8];
	__shared__ u16 s_off[MD_WAVES][BK_MAX_BINS];
	__shared__ u16 s_num[MD_WAVES][BK_MAX_BINS];
	__shared__ u8 s_head[MD_WAVES][64];
	constexpr u32 limit = (1u << 16) - 2u * N;              // MaxAccumulatedValue (src/SymbolCoderRC.h:67): Rescale() before a symbol is coded on a row that has reached it
	const CtxJob j = jobs[blockIdx.y];
	const u32 w = wave_id(), lane = lane_id();
	// Neighbouring buckets' runs share 128-byte lines of the element and record arrays, and consecutive workgroups go round the eight
	// XCDs (each with its own L2): workgroup x takes the buckets of place (x mod 8) * gridDim.x / 8 + x / 8, so that an XCD works
	// on one contiguous eighth of the buckets (element fetch of the 32-symbol model: 4.4 x what it reads without this)
#ifndef MD_XCD_MAP
#define MD_XCD_MAP 1
#endif
	u32 bx = blockIdx.x;
	// ... which eighth, goes round with the stream: the same contexts are hot in every stream of a launch (four-level qualities: ten
	// buckets of 75-170 k symbols at the same ten places of all 128 streams), and with the same eighth on the same XCD for every
	// stream their ~1300 long walks met on the SIMDs of the two or three XCDs those places fall on (k_model<16>: 14.0 ms per launch)
	if (MD_XCD_MAP && gridDim.x % 8u == 0) bx = (((bx & 7u) + blockIdx.y) & 7u) * (gridDim.x >> 3) + (bx >> 3);
	const u32 bucket = bx * MD_WAVES + w;
	const u32 buckets = 1u << j.bk_hb;
	if (!j.bk_on || bucket >= buckets || bk[j.jid]) return;
	const u32* src = (const u32*)(pool + j.elems);
	// the 8-byte records of the bucketed path go to the stream's second element buffer (only k_sort uses that one), k_place reads them
	// there; without time bins (tests) the record goes into the stream's array of chunks at once
	RcPack* recs = j.bk_binned ? (RcPack*)pool + j.elems_b : rec_pool + j.trip;
	const u32 keys = 1u << j.bk_lb;
	const bool binned = j.bk_binned != 0;
	map_t* map = s_map[w]; u32* rows = (u32*)s_rows[w]; u16* off = s_off[w]; u16* num = s_num[w]; u8* head = s_head[w];
	const u32 n_bins = (j.n + BK_BIN - 1) >> BK_TB;

	// where this bucket's elements lie inside every tile (k_binoff: the offsets of the tile's buckets), and how many.  A stream of more
	// tiles than the LDS arrays hold (`wide`: the reference's -m1 / -m2 buffer sizes; rounds 4-5 sent those through k_sort / k_replay,
	// 560 of a batch's 900 ms at -b64) only counts here and reads a group's 64 tiles again when it gets there
	const bool wide = n_bins > j.bk_narrow_bins;
	auto tile_info = [&](u32 b, u32* o_out) -> u32
	{
		const u16* row = bcnt + (u64)j.bk_cnt + (u64)b * buckets;
		const u32 o = row[bucket];
		const u32 tile_n = j.n - (b << BK_TB) < BK_BIN ? j.n - (b << BK_TB) : BK_BIN;
		*o_out = o;
		return (bucket + 1 < buckets ? (u32)row[bucket + 1] : tile_n) - o;
	};
	u32 nb = 0;
	for (u32 b = lane; b < n_bins; b += 64)
	{
		u32 o;
		const u32 c = tile_info(b, &o);
		if (!wide) { off[b] = (u16)o; num[b] = (u16)c; }
		nb += c;
	}
	head[lane] = 0xFFu;
	for (u32 d = 32; d >= 1; d >>= 1) nb += __shfl_xor(nb, (int)d);
	if (!nb) return;
	if (nb > j.bk_limit)
	{	// a bucket one wave should not walk: the stream goes through k_sort / k_replay (their launches follow this one)
		if (lane == 0) bk_fallback(j, bk);
		return;
	}
	const bool may_rescale = (u32)N + 2u * nb + 128u >= (1u << 16) - 2u * N;
	wave_fence();

	// The windows of a bucket: its elements tile by tile (stream order), 64 tiles (a group) at a time: lane k looks at tile g_bin + k,
	// whose elements take the positions [g_f, g_inc) of the group.  A window is the next 64 positions: the lanes whose tile starts
	// (or goes on) inside it leave their number at head[position], and a position finds its tile at the nearest head at or below
	// it.  The last window of a group is short.  -> index into the element array (= into the record array), or MD_NONE
	u32 g_bin = 0, g_pos = 0, g_tot = 0, left = nb;        // wave-uniform
	u32 g_inc = 0, g_f = 0, g_off = 0;                    // (g_off: wide streams -- the offset of tile g_bin + lane)
	auto load_group = [&]()
	{
		const u32 b = g_bin + lane;
		u32 c = 0;
		if (b < n_bins) { if (wide) c = tile_info(b, &g_off); else c = (u32)num[b]; }
		g_inc = wave_incl_scan_dpp(c); g_f = g_inc - c;
		g_tot = wave_last(g_inc); g_pos = 0;
	};
	load_group();
	// A large bucket (a hot context: round 5 measured 75-170 k symbols with one or two rows, 2700 cycles per window, half of them the
	// bookkeeping below) has hundreds of elements per tile, so most of its windows lie inside ONE tile's run: the tile that holds the
	// group position g_pos is found with one ballot and the window is 64 consecutive elements of it.
	const bool big = nb >= j.bk_big;
	// (s_setprio(3) for such a wave -- the launch's long pole, bound by instruction issue -- was measured: 36.2 / 35.3 against 35.9 / 35.4 GB/s)
	auto next_window = [&]() -> u32
	{
		if (!left) return MD_NONE;
		while (g_pos >= g_tot && g_bin < n_bins) { g_bin += 64; load_group(); }
		if (g_pos >= g_tot) { left = 0; return MD_NONE; }          // (the counts add up to nb: not reached)
		if (big)
		{
			const u64 in = __ballot(g_f <= g_pos && g_pos < g_inc);              // exactly one lane: the runs tile [0, g_tot)
			const u32 k0 = (u32)__ffsll((long long)in) - 1u;
			const u32 e0 = (u32)__builtin_amdgcn_readlane((int)g_inc, (int)k0);
			if (e0 - g_pos >= 64u)
			{
				const u32 f0 = (u32)__builtin_amdgcn_readlane((int)g_f, (int)k0);
				const u32 bin = g_bin + k0;
				const u32 o0 = wide ? (u32)__builtin_amdgcn_readlane((int)g_off, (int)k0) : (u32)off[bin & (BK_MAX_BINS - 1u)];
				const u32 idx = (bin << BK_TB) + o0 + (g_pos - f0) + lane;
				g_pos += 64u; left -= 64u;
				return idx;
			}
		}
		const int rel = (int)(g_f - g_pos);
		if (g_inc > g_f && g_inc > g_pos && rel < 64) head[rel > 0 ? rel : 0] = (u8)lane;
		wave_fence();
		const u32 hd = head[lane];
		wave_fence();
		head[lane] = 0xFFu;
		const u64 hm = __ballot(hd != 0xFFu);
		const u64 le = hm & (lanemask_lt() | (1ull << lane));
		const u32 hl = le ? 63u - (u32)__clzll((long long)le) : 0u;
		const u32 k = (u32)__shfl((int)hd, (int)hl) & 63u;
		const u32 k0 = (u32)__builtin_amdgcn_readfirstlane((int)hd) & 63u;         // position 0 always has a head
		const u32 f0 = (u32)__builtin_amdgcn_readlane((int)g_f, (int)k0);
		const u32 r = lane - hl + (hl == 0 ? g_pos - f0 : 0u);
		const u32 avail = g_tot - g_pos < 64 ? g_tot - g_pos : 64u;
		const u32 bin = g_bin + k;
		const u32 ok = wide ? (u32)__shfl((int)g_off, (int)k) : (u32)off[bin & (BK_MAX_BINS - 1u)];
		const u32 idx = lane < avail ? (bin << BK_TB) + ok + r : MD_NONE;
		g_pos += avail; left -= avail;
		return idx;
	};

	// the first windows are on their way while the rows are set up
	u32 elq[MD_AHEAD], ixq[MD_AHEAD];                       // the elements of the next windows and where they lie
#pragma unroll
	for (u32 k = 0; k < MD_AHEAD; ++k) { ixq[k] = next_window(); elq[k] = src[ixq[k] != MD_NONE ? ixq[k] : 0u]; }
	if (MAPBITS) for (u32 i = lane; i < keys; i += 64) map[i] = (map_t)Map::NONE;
	for (u32 i = lane; i < (MAPBITS ? ROWS : keys) * STRIDE; i += 64) rows[i] = md_init_word<N>(i % STRIDE);      // every counter 1
	wave_fence();

	u32 n_rows = 0;
	// the walk in two instantiations: the rescale test costs registers and instructions that only buckets of > 32 K symbols need
#ifdef MD_PROFILE
	u64 pf[6] = {0, 0, 0, 0, 0, 0}; u32 pf_n = 0;
#define MD_T(k, dep) { u64 t_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_) : "s"(dep) : "memory"); pf[k] += t_ - pf_t; pf_t = t_; }
#else
#define MD_T(k, dep)
#endif
	auto walk = [&](auto resc_tag)
	{
	constexpr bool RESC = decltype(resc_tag)::value;
	u32 coded = 0, dry = 0;
	// one window: the elements of slot (el_slot, ix_slot) are coded and the slot takes the window MD_AHEAD further on.  The slots keep
	// their registers (the loop below goes round them): moving a requested element to another register would wait for it -- and with
	// it for every request made since (round 5: `s_waitcnt vmcnt(0)` at the head of the loop, a memory round trip per window)
	auto step = [&](u32& el_slot, u32& ix_slot) -> bool
	{
#ifdef MD_PROFILE
		u64 pf_t; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(pf_t) :: "memory"); ++pf_n;
#endif
		const u32 el = el_slot, ix = ix_slot;
		const bool valid = ix != MD_NONE;
		const u32 n_valid = (u32)__popcll(__ballot(valid));
		coded += n_valid;
		dry = (n_valid || left) ? 0u : dry + 1u;               // (nothing left and nothing in flight: the counts did not add up -- not reached)
		const u32 ix_new = next_window();
		// (requests and stores are made by every lane, those without an element from / to a place that does not matter: a request
		// or store under a branch cannot be counted on by the wait for an older request, which then waits for everything)
		const u32 el_new = src[ix_new != MD_NONE ? ix_new : 0u];
		MD_T(0, __builtin_amdgcn_readfirstlane((int)ix_new))                    // window bookkeeping, next load issued
		MD_T(1, __builtin_amdgcn_readfirstlane((int)el))                        // the wait for this window's elements

		const u32 key = el >> BK_EL_KEY_SHIFT, sym = (el >> BK_EL_SYM_SHIFT) & (u32)(N - 1);
		u32 rid = key;
		if (MAPBITS)
		{
			rid = valid ? (u32)map[key] : 0u;
			const bool need = valid && rid == Map::NONE;
			if (__ballot(need))
			{	// contexts met for the first time: one lane of each (whichever store the LDS applies last) takes the next row
				if (need) map[key] = (map_t)(Map::CLAIM | lane);
				wave_fence();
				const bool leader = need && (u32)map[key] == (Map::CLAIM | lane);
				const u64 lm = __ballot(leader);
				const u32 mine = n_rows + (u32)__popcll(lm & lanemask_lt());
				n_rows += (u32)__popcll(lm);
				if (n_rows > ROWS)
				{	// more contexts in this bucket than rows: the stream goes through k_sort / k_replay (their launches follow this one)
					if (lane == 0) bk_fallback(j, bk);
					return false;
				}
				wave_fence();
				if (leader) map[key] = (map_t)mine;
				wave_fence();
				if (need) rid = (u32)map[key];
			}
		}
		MD_T(2, __builtin_amdgcn_readfirstlane((int)rid))                       // key -> row
		u64 rec = 0;
		u32* row = rows + rid * STRIDE;
		// Only a bucket with > 32 K symbols can bring a row to its rescale point.  There, a window in which some lane's row could get
		// that far (64 symbols add 128) is coded one element at a time, and the lane whose row is due halves it first.
		const bool near = RESC && valid && md_total<N>(row) + 128u >= limit;
		if (RESC && __ballot(near))
		{
			const u64 vm = __ballot(valid);
			for (u32 l = 0; l < 64 && ((vm >> l) & 1ull); ++l)
			{
				if (lane == l)
				{
					if (md_total<N>(row) >= limit) md_rescale<N>(row);
					rec = md_code<N>(row, sym);
				}
				wave_fence();
			}
		}
		else if (valid) rec = md_code<N>(row, sym);
#ifdef DSRC_EMU_BUILD
		(void)__ballot(true);                                         // the emulator runs lanes one after the other: keep them in step per window
#endif
		MD_T(3, __builtin_amdgcn_readfirstlane((int)(u32)rec))                  // the symbols on their rows
		// the record takes the element's place (k_place puts the tile in stream order), or goes where it belongs at once
		if (binned)
		{
			RcPack* to = recs + ix;
			if (!valid) to = nowhere + lane;
			*to = rec | ((u64)(el & (BK_BIN - 1u)) << 48);
		}
		else
		{	// (tests: no time bins) the record goes where k_rc reads it at once: rc6_store's two pieces
			const u32 t = (ix & ~(BK_BIN - 1u)) | (el & (BK_BIN - 1u));
			u8* chunk = (u8*)recs + rc6_chunk_off(t >> 6);
			u32* p32 = valid ? (u32*)chunk + (t & 63u) : (u32*)(nowhere + lane);
			u16* p16 = valid ? (u16*)(chunk + 256) + (t & 63u) : (u16*)(nowhere + lane) + 2;
			*p32 = (u32)rec; *p16 = (u16)(rec >> 32);
		}
		el_slot = el_new; ix_slot = ix_new;
		MD_T(4, 0)                                                               // the records stored (issue)
		return coded < nb && dry <= MD_AHEAD;
	};
	if (MD_ROTATE_ALL || MAPBITS != 0)
		for (bool go = nb != 0; go;)
		{
#pragma unroll
			for (u32 k = 0; k < MD_AHEAD; ++k) if (go) go = step(elq[k], ixq[k]);
		}
	else
		// (rows owned by their keys, eight-byte rows, four waves to a workgroup: enough waves per SIMD to wait behind, and the loop
		// in one copy is the faster one -- k_model<4, 0, 4096> 2.03 ms against 2.29 with the slots going round)
		for (bool go = nb != 0; go;)
		{
			u32 e = elq[0], i = ixq[0];
			go = step(e, i);
#pragma unroll
			for (u32 k = 0; k + 1 < MD_AHEAD; ++k) { elq[k] = elq[k + 1]; ixq[k] = ixq[k + 1]; }
			elq[MD_AHEAD - 1] = e; ixq[MD_AHEAD - 1] = i;
		}
	};
	if (may_rescale) walk(std::true_type()); else walk(std::false_type());
#ifdef MD_PROFILE
	if (lane == 0 && nb >= 65536u && (bucket & 3u) == 0)
		printf("MD N=%d nb=%u rows=%u windows=%u cycles/window: book %u wait %u map %u code %u store %u\n", N, nb, n_rows, pf_n,
			(u32)(pf[0] / pf_n), (u32)(pf[1] / pf_n), (u32)(pf[2] / pf_n), (u32)(pf[3] / pf_n), (u32)(pf[4] / pf_n));
#endif
}

// ---- k_place: a time bin's records into stream order ------------------------------------------------------------------------
// Grid: x = time bin, y = stream of the slice.  In: the bin's 8-byte records in bucket order (k_model; in the stream's second element
// buffer of the slice), t mod 8192 in their top 16 bits.  Out: the bin's 128 chunks of k_rc's six-byte layout (k_rc.h: 64 dwords
// freq | cum << 16, then 64 u16 totals per chunk) in the stream's array of chunks.
#ifndef PLACE_WG
#define PLACE_WG 512
#endif
#ifndef PLACE_PASSES
#define PLACE_PASSES 2                 // the bin's records wait in registers (16 per thread) and go through LDS half a bin at a time: 32 KB, so that a
#endif                                 // workgroup finds room on a CU next to a k_rc workgroup and other instances' kernels (64 KB: 1.4 ms alone, 2.5 - 3 ms in the bench)
__global__ void __launch_bounds__(PLACE_WG) k_place(const CtxJob* jobs, const u64* pool, RcPack* rec_pool, const u32* bk)
{
	constexpr u32 PER = BK_BIN / PLACE_WG, HALF = BK_BIN / PLACE_PASSES;
	__shared__ u64 s_rec[HALF];
	const CtxJob j = jobs[blockIdx.y];
	if (!j.bk_on || !j.bk_binned || bk[j.jid]) return;
	const u32 bin = blockIdx.x;
	const u32 t0 = bin << BK_TB;
	if (t0 >= j.n) return;
	const u32 cnt = j.n - t0 < BK_BIN ? j.n - t0 : BK_BIN;
	const RcPack* r = (const RcPack*)pool + j.elems_b + t0;
	u8* out = (u8*)(rec_pool + j.trip) + (size_t)(t0 >> 6) * RC6_CHUNK_BYTES;      // the bin's 128 chunks
	u64 v[PER];
#pragma unroll
	for (u32 k = 0; k < PER; ++k) v[k] = threadIdx.x + k * PLACE_WG < cnt ? r[threadIdx.x + k * PLACE_WG] : ~0ull;      // no record: a time no pass takes
#pragma unroll
	for (u32 p = 0; p < PLACE_PASSES; ++p)
	{
		const u32 lo = p * HALF;
#pragma unroll
		for (u32 k = 0; k < PER; ++k)
		{
			const u32 t = (u32)(v[k] >> 48) - lo;
			if (t < HALF) s_rec[t] = v[k] & 0xFFFFFFFFFFFFull;
		}
		__syncthreads();                                               // (first pass: every record of the bin is in some thread's registers by now)
		for (u32 i = threadIdx.x; i < HALF && lo + i < cnt; i += PLACE_WG)
		{
			const u32 t = lo + i;
			const u64 v6 = s_rec[i];
			u8* c = out + (t >> 6) * RC6_CHUNK_BYTES;
			((u32*)c)[t & 63u] = (u32)v6;
			((u16*)(c + 256))[t & 63u] = (u16)(v6 >> 32);
		}
		if (p + 1 < PLACE_PASSES) __syncthreads();
	}
}
