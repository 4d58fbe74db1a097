// Bucketed context path (round 4): the order-k model statistics of a stream in ONE pass over 8-byte elements in HBM
// instead of two, and records that leave in runs instead of one 32-byte sector per 8-byte record.
//   TDnaRCOrderModeler::UpdateHash / EncodeSymbol    src/DnaModelerRCO.h:94-131
//   TQualityModelBase::UpdateHash, TQualityModelExt  src/QualityEncoder.h:77-94,126-151
//   TSymbolCoderRC::EncodeSymbol/Accumulate/Rescale   src/SymbolCoderRC.h:35-90
//
// k_sort + k_replay (k_rc.h) sort a stream's (context, symbol, t) elements by context with two LSD passes through HBM, replay
// the sorted array and scatter one 8-byte record per symbol to stream order.  Here:
//   k_part   : ONE stable partition of the elements by the top `hb` bits of a *mixed* context key (key = ctx * odd constant mod
//              2^K: a bijection, so equal keys <=> equal contexts, and the top bits spread hot neighbourhoods of the context space
//              over all buckets) into <= 1024 buckets of a few thousand elements -- what pass 0 of k_sort does, on another digit;
//   k_finish : one workgroup per bucket: the bucket is sorted by the remaining key bits IN LDS (stable counting sort: per-wave
//              counters, one LDS atomic per element), replayed there with the segmented wave scan of k_replay, and its records
//              leave grouped by time bin (t >> 14): a bucket's ~16 records of a bin are written as one run into the bin's
//              region of the stream's record array, wherever the bin's fill counter says (the record carries t & 16383);
//   k_place  : one workgroup per (stream, time bin): the bin's 16 K records are put in stream order through LDS, in place.
// Per symbol that is 8 B written + 8 B read (partition), 8 B + 8 B (finish -> bins), 8 B + 8 B (place), all in runs, against
// 8 + 16 + 8 B in runs plus one 32-byte sector per record before.
//
// Buckets that do not fit the LDS tile (BK_CAP elements) are sorted in rounds of key ranges (the bucket is re-read once per
// round); a single context with more than BK_CAP symbols is replayed chunk by chunk by one wave that carries the row.  A stream
// with a bucket beyond BK_LIMIT elements (a context that holds a large share of the stream: four-level qualities, poly-A reads) is
// handed to k_sort / k_replay_seams / k_replay, whose range-splitting replay is made for exactly that: k_part appends it to the
// fallback list of its launch group.  Either way the records k_rc reads are the same.
#pragma once
#include "k_rc.h"

#define BK_MAX_HB 10                   // bucket digit: <= 1024 buckets (k_part's LDS is k_sort's)
#define BK_MAX_LB 11                   // key bits sorted in LDS
#define BK_WG 256
#define BK_WAVES (BK_WG / 64)
#ifndef BK_CAP
#define BK_CAP 4096                    // elements per LDS round
#endif
#define BK_LIMIT (8 * BK_CAP)          // largest bucket the rounds are allowed to chew through (a bucket is re-read per round)
#define BK_TB 14                       // log2(records per time bin): 48 bits of record + 14 bits of t fit the 8-byte slot, a bin fits LDS
#define BK_BIN (1u << BK_TB)
#define BK_MAX_BINS 1024               // streams of up to 16 M symbols
#define BK_HASH_MUL 0x9E3779B1u

// the `bk` pool (u32): [0 .. NJ) fallback flag per job | fallback lists, one per launch group: count, then job ids |
// bin fill counters | bucket offsets (2^hb + 1 per job).  Everything up to the bucket offsets is zeroed per batch.

__device__ __forceinline__ u64 bk_rekey(const CtxJob& j, u64 el)
{
	const u32 key = ((u32)(el >> ELEM_CTX_SHIFT) * j.bk_mul) & j.bk_kmask;
	return (el & ((1ull << ELEM_CTX_SHIFT) - 1ull)) | ((u64)key << ELEM_CTX_SHIFT);
}

__device__ __forceinline__ void bk_fallback(const CtxJob& j, u32* bk)
{
	const u32 k = atomicAdd(&bk[j.bk_fb], 1u);
	bk[j.bk_fb + 1 + k] = j.jid;
	bk[j.jid] = 1u;
}

// ---- k_part: stable partition by the top digit of the mixed key ------------------------------------------------------------------
// One workgroup per stream, tiles of SORT_WG * SORT_ITEMS elements, ranking as in k_sort (one LDS atomic per element on the
// wave's packed counter pair, or ballots where the device failed k_lds_order_test).
template <bool ATOMIC>
__global__ void __launch_bounds__(SORT_WG) __attribute__((amdgpu_waves_per_eu(SORT_OCC, SORT_OCC))) k_part(const CtxJob* jobs, u64* pool, const u8* d_stream, const u8* q_stream, const u8* qp_stream, BlkState* st, u32* bk)
{
	__shared__ u32 s_base[SORT_MAX_BINS];
	__shared__ u32 s_delta[SORT_MAX_BINS];
	__shared__ u16 s_cnt[SORT_WAVES][SORT_MAX_BINS];
	__shared__ u16 s_off[SORT_WAVES][SORT_MAX_BINS];
	__shared__ u64 s_tile[SORT_WG * SORT_ITEMS];
	__shared__ u32 s_ws[SORT_WAVES];
	__shared__ u8 s_rank[256];
	__shared__ u32 s_max;
	const CtxJob j = jobs[blockIdx.x];
	if (!j.bk_on) { if (threadIdx.x == 0) bk_fallback(j, bk); return; }
	const u32 n = j.n, bins = 1u << j.bk_hb, lb = j.bk_lb;
	const u32 wv = wave_id(), lane = lane_id();
	constexpr u32 tile_elems = SORT_WG * SORT_ITEMS;
	const u8* sym_src = (j.is_dna ? d_stream : q_stream) + j.src_off;
	const u8* qp = qp_stream + j.src_off;

	for (u32 i = threadIdx.x; i < 256; i += blockDim.x) s_rank[i] = (!j.is_dna && j.translate) ? st[j.blk].q_sym[i] : (u8)i;
	for (u32 i = threadIdx.x; i < bins; i += blockDim.x) s_base[i] = 0;
	if (threadIdx.x == 0) s_max = 0;
	__syncthreads();
	{	// bucket sizes
		bool bad = false;
		u32 i_from = 0;
		if (j.is_dna && j.alpha_bits == 2 && j.order <= 9 && n >= 32)
		{	// 2-bit bases: the contexts of four consecutive symbols t..t+3 are 2*order-bit fields of the twelve symbols s[t-9 .. t+2],
			// which come out of two 8-byte windows
			const u32 cmask = (u32)((1ull << (2 * j.order)) - 1ull);
			const u32 n4 = (n - 12) / 4;                                  // groups starting at t = 12, 16, ...
			for (u32 g = threadIdx.x; g < n4; g += blockDim.x)
			{
				const u32 t = 12 + 4 * g;
				const u64 wa = *(const u64_unaligned*)(sym_src + t - 9), wb = *(const u64_unaligned*)(sym_src + t - 5);
				if (((wa | wb) & 0xFCFCFCFCFCFCFCFCull) || sym_src[t + 3] >= 4) bad = true;
				const u32 R = (pack2x8(__builtin_bswap64(wa)) << 8) | (pack2x8(__builtin_bswap64(wb)) & 0xFFu);     // s[t-9] on top, s[t+2] at the bottom
#pragma unroll
				for (u32 k = 0; k < 4; ++k)
				{
					const u32 ctx = (R >> (6 - 2 * k)) & cmask;
					atomicAdd(&s_base[((ctx * j.bk_mul) & j.bk_kmask) >> lb], 1u);
				}
			}
			i_from = 12 + 4 * n4;
			for (u32 i = threadIdx.x; i < 12; i += blockDim.x)
				atomicAdd(&s_base[(u32)(bk_rekey(j, ctx_elem_dna(j, sym_src, i, &bad)) >> (ELEM_CTX_SHIFT + lb))], 1u);
		}
		for (u32 i = i_from + threadIdx.x; i < n; i += blockDim.x)
		{
			const u64 el = j.is_dna ? ctx_elem_dna(j, sym_src, i, &bad) : ctx_elem_qua(j, sym_src, qp, s_rank, i);
			atomicAdd(&s_base[(u32)(bk_rekey(j, el) >> (ELEM_CTX_SHIFT + lb))], 1u);
		}
		if (bad) atomicOr(&st[j.blk].err, (u32)DSRC_ERR_REF_UB);
	}
	__syncthreads();
	{	// exclusive scan -> bucket offsets (k_finish reads them), largest bucket
		u32 carry = 0;
		for (u32 b0 = 0; b0 < bins; b0 += blockDim.x)
		{
			const u32 i = b0 + threadIdx.x;
			const u32 v = i < bins ? s_base[i] : 0;
			if (v) atomicMax(&s_max, v);
			u32 tot;
			const u32 ex = block_excl_scan(v, &tot);
			if (i < bins) { s_base[i] = carry + ex; bk[j.bk_boff + i] = carry + ex; }
			carry += tot;
		}
		if (threadIdx.x == 0) bk[j.bk_boff + bins] = n;
	}
	for (u32 i = threadIdx.x; i < SORT_WAVES * SORT_MAX_BINS; i += blockDim.x) (&s_cnt[0][0])[i] = 0;
	__syncthreads();
	if (s_max > BK_LIMIT) { if (threadIdx.x == 0) bk_fallback(j, bk); return; }

	u64* dst = pool + j.elems;
	const u32 shift = ELEM_CTX_SHIFT + lb;
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
			if (i < n) el[k] = bk_rekey(j, j.is_dna ? ctx_elem_dna(j, sym_src, i, &bad) : ctx_elem_qua(j, sym_src, qp, s_rank, i));
		}
#pragma unroll
		for (u32 k = 0; k < SORT_ITEMS; ++k)
		{
			const u32 i = wbase + k * 64 + lane;
			const bool valid = i < n;
			const u32 d = (u32)(el[k] >> shift);
			if (ATOMIC)
			{
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
			for (u32 b = 0; b < SORT_DIGIT_BITS; ++b)
			{
				const u64 m = __ballot((d >> b) & 1u);
				peers &= ((d >> b) & 1u) ? m : ~m;
			}
			const u32 before = valid ? s_cnt[wv][d] : 0;
			const u32 r = (u32)__popcll(peers & lanemask_lt());
			rk[k] = before + r;
			const u64 sync = __ballot(true);
			if (valid && r == 0 && sync) s_cnt[wv][d] = (u16)(before + (u32)__popcll(peers));
		}
		__syncthreads();
		for (u32 d0 = 0, carry = 0; d0 < bins; d0 += SORT_WG)
		{
			const u32 dd = d0 + threadIdx.x;
			u32 c[SORT_WAVES], tot = 0;
#pragma unroll
			for (u32 w = 0; w < SORT_WAVES; ++w) { c[w] = dd < bins ? s_cnt[w][dd] : 0u; tot += c[w]; }
			const u32 inc = wave_incl_scan_dpp(tot);
			if (lane == 63) s_ws[wv] = inc;
			__syncthreads();
			u32 run = carry + inc - tot;
			const u32 nws = (bins - d0 + 63) / 64 < SORT_WAVES ? (bins - d0 + 63) / 64 : SORT_WAVES;
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
			if (i < n) s_tile[(u32)s_off[wv][(u32)(el[k] >> shift)] + rk[k]] = el[k];
		}
		__syncthreads();
		const u32 tile_n = n - tile < tile_elems ? n - tile : tile_elems;
#pragma unroll
		for (u32 k = 0; k < SORT_ITEMS; ++k)
		{
			const u32 p = k * SORT_WG + threadIdx.x;
			if (p < tile_n)
			{
				const u64 e = s_tile[p];
				dst[s_delta[(u32)(e >> shift)] + p] = e;
			}
		}
		if (bad) atomicOr(&st[j.blk].err, (u32)DSRC_ERR_REF_UB);
		// no barrier here: the next tile touches s_delta / s_tile only behind its own first two barriers
	}
}

// ---- k_finish ---------------------------------------------------------------------------------------------------------------------
// the open segment's row, carried from window to window (and from chunk to chunk of a context that is replayed in chunks)
template <int N> struct BkRow
{
	ReplayRow<N> base, cnt, cumbase, cntpre;
	bool open; u32 T0, epoch_cnt, epoch_left; u64 prev_ctx;
	__device__ __forceinline__ void reset()
	{
		base.a = base.b = 1; cnt.a = cnt.b = 0; cumbase.a = cumbase.b = 0; cntpre.a = cntpre.b = 0;
		open = false; T0 = N; epoch_cnt = 0; epoch_left = 0; prev_ctx = ~0ull;
	}
};

// k_replay's window loop on a sorted tile in LDS: s[0 .. n) sorted by key, this wave replays [pos, hi); pos is a segment head
// unless R.open (then s[pos] continues the segment R carries).  The record of s[i] is written over s[i] together with the low
// time bits: freq | cum << 16 | total << 32 | (t & (BK_BIN - 1)) << 48, and the time bin goes to s_bin[i].
template <int N>
__device__ __forceinline__ void bk_replay_range(u64* s, u16* s_bin, u32 pos, const u32 hi, const u32 n, BkRow<N>& R, u32* tl)
{
	constexpr int BITS = N <= 4 ? 2 : N <= 8 ? 3 : N <= 16 ? 4 : N <= 32 ? 5 : N <= 64 ? 6 : 7;
	const u32 lane = lane_id();
	const u32 limit = (1u << 16) - 2u * N;                   // MaxAccumulatedValue (src/SymbolCoderRC.h:67)
	const u32 E0 = (limit - N + 1) / 2;                      // symbols a fresh row codes before its first rescale
	if (pos >= hi) return;
	u64 el_cur = pos + lane < n ? s[pos + lane] : 0;
	for (;;)
	{
		const u32 idx = pos + lane;
		const bool valid = idx < n;
		const u64 el = el_cur;
		const u64 ctx = el >> ELEM_CTX_SHIFT;
		u64 pctx = __shfl_up(ctx, 1);
		if (lane == 0) pctx = R.prev_ctx;
		const bool head = valid && ctx != pctx;
		const u64 hm_all = __ballot(head);
		u32 tile_len = (u32)__popcll(__ballot(valid));
		bool last = false;
		if (pos + tile_len >= hi) { tile_len = hi - pos; last = true; }
		bool rescale_after = false;
		const bool cont = R.open && !(hm_all & 1ull);            // lane 0 continues the open segment
		if (cont)
		{
			u32 c = hm_all ? (u32)__ffsll((long long)hm_all) - 1 : 64u;
			if (c > tile_len) c = tile_len;
			if (c > R.epoch_left) { tile_len = R.epoch_left; rescale_after = true; last = false; }
		}
		const u32 npos = pos + tile_len;
		const u64 el_next = npos + lane < n ? s[npos + lane] : 0;
		const u64 tmask = tile_len >= 64 ? ~0ull : ((1ull << tile_len) - 1ull);
		const bool active = lane < tile_len;
		const u64 hm = hm_all & tmask;
		const u64 heads_le = hm & (lanemask_lt() | (1ull << lane));
		const u32 seg_start = heads_le ? 63u - (u32)__clzll((long long)heads_le) : 0u;
		const bool in_cont = cont && heads_le == 0;
		const u64 segmask_lt = lanemask_lt() & ~((1ull << seg_start) - 1ull);
		const u32 sym = (u32)(el >> ELEM_SYM_SHIFT) & 0xFFu;
		const u32 last_start = hm ? 63u - (u32)__clzll((long long)hm) : 0u;
		const bool last_is_cont = cont && hm == 0;
		const u64 lastmask = tmask & ~((1ull << last_start) - 1ull);

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
		tl[lane] = 0; if (N > 64) tl[lane + 64] = 0;
		wave_fence();
		const u64 eq_last = EQ & lastmask;
		if (active && ((lastmask >> lane) & 1ull) && (eq_last >> lane) == 1ull) tl[sym] = (u32)__popcll(eq_last);
		wave_fence();
		const u32 add_a = tl[lane], add_b = (N > 64) ? tl[lane + 64] : 0u;
		const u32 in_seg = (u32)__popcll(segmask_lt & tmask);
		u32 f, cum, tot;
		{
			ReplayRow<N> st, cs;
			st.a = R.base.a + 2 * R.cnt.a; st.b = R.base.b + 2 * R.cnt.b; cs.a = R.cumbase.a + 2 * R.cntpre.a; cs.b = R.cumbase.b + 2 * R.cntpre.b;
			const u32 b0 = st.get(sym), cb = cs.get(sym);
			if (in_cont) { f = b0 + 2 * same; cum = cb + 2 * less; tot = R.T0 + 2 * (R.epoch_cnt + in_seg); }
			else { f = 1 + 2 * same; cum = sym + 2 * less; tot = N + 2 * in_seg; }
		}
		if (active)
		{
			const u32 t = (u32)el;
			s[idx] = (u64)f | ((u64)cum << 16) | ((u64)tot << 32) | ((u64)(t & (BK_BIN - 1u)) << 48);
			s_bin[idx] = (u16)(t >> BK_TB);
		}
		if (tile_len > 0)
		{
			if (last_is_cont) { R.cnt.a += add_a; R.cnt.b += add_b; R.epoch_cnt += tile_len; R.epoch_left -= tile_len; }
			else
			{
				R.base.a = (lane < (u32)N) ? 1u : 0u; R.base.b = (N > 64) ? 1u : 0u;
				R.cnt.a = add_a; R.cnt.b = add_b;
				R.T0 = N; R.epoch_cnt = tile_len - last_start; R.epoch_left = E0 - R.epoch_cnt;
				u32 t; replay_prefix<N>(R.base, R.cumbase, &t);
				R.open = true;
			}
		}
		if (rescale_after || (R.open && R.epoch_left == 0))
		{	// Rescale(): stats[i] -= stats[i] >> 1 on stats = base + 2*cnt
			u32 x = R.base.a + 2 * R.cnt.a; R.base.a = (lane < (u32)N) ? x - (x >> 1) : 0u;
			if (N > 64) { x = R.base.b + 2 * R.cnt.b; R.base.b = x - (x >> 1); }
			R.cnt.a = R.cnt.b = 0;
			replay_prefix<N>(R.base, R.cumbase, &R.T0);
			R.epoch_cnt = 0; R.epoch_left = (limit - R.T0 + 1) / 2;
		}
		{ u32 t; replay_prefix<N>(R.cnt, R.cntpre, &t); }
		if (tile_len) R.prev_ctx = __shfl(ctx, (int)(tile_len - 1));
		el_cur = el_next;
		pos = npos;
		if (last) break;
	}
}

// position of the first segment head at or after p in the sorted tile s[0 .. n) (n if there is none); 0 < p
__device__ __forceinline__ u32 bk_next_head(const u64* s, u32 p, u32 n)
{
	for (; p < n; p += 64)
	{
		const u32 i = p + lane_id();
		const bool head = i < n && (s[i] >> ELEM_CTX_SHIFT) != (s[i - 1] >> ELEM_CTX_SHIFT);
		const u64 m = __ballot(head);
		if (m) return p + (u32)__ffsll((long long)m) - 1u;
	}
	return n;
}

// Grid: x = bucket, y = stream of the launch group (one alphabet size).  j.bk_binned: the records leave grouped by time bin for
// k_place; otherwise they are scattered to stream order from here (DSRC_GPU_BUCKETS_BINNED=0: measurements).
template <int N, bool ATOMIC>
__global__ void __launch_bounds__(BK_WG) k_finish(const CtxJob* jobs, const u64* pool, RcPack* rec_pool, u32* bk)
{
	__shared__ u64 s_el[BK_CAP];
	__shared__ u16 s_bin[BK_CAP];
	__shared__ u32 s_cnt[BK_WAVES][(1 << BK_MAX_LB) / 2];     // per wave and key, two 16-bit fields per word: counts, then tile positions
	__shared__ u16 s_pre[(1 << BK_MAX_LB) + 2];               // buckets larger than the tile: exclusive prefix of the key totals
	__shared__ u32 s_tail[BK_WAVES][128];
	__shared__ u32 s_bnd[BK_WAVES + 1];
	__shared__ u32 s_hist[BK_MAX_BINS], s_gbase[BK_MAX_BINS];
	const CtxJob j = jobs[blockIdx.y];
	const u32 bucket = blockIdx.x;
	if (!j.bk_on || bucket >= (1u << j.bk_hb) || bk[j.jid]) return;
	const u32 lo = bk[j.bk_boff + bucket], nb = bk[j.bk_boff + bucket + 1] - lo;
	if (!nb) return;
	const u64* src = pool + j.elems + lo;
	RcPack* recs = rec_pool + j.trip;
	const u32 keys = 1u << j.bk_lb, kmask = keys - 1u;
	const u32 words = keys >= 2 ? keys / 2 : 1u;               // counter words per wave
	const u32 w = wave_id(), lane = lane_id();
	const u32 q = (nb + BK_WG - 1) / BK_WG * 64;               // the wave's piece of the bucket: a multiple of 64 elements
	const u32 w_lo = w * q, w_hi = w_lo + q < nb ? w_lo + q : nb;
	const u32 n_bins = (j.n + BK_BIN - 1) >> BK_TB;
	const bool BINNED = j.bk_binned != 0;

	const bool multi = nb > BK_CAP;
	if (multi)
	{	// key totals over the whole bucket -> s_pre
		u32* tot = &s_cnt[0][0];                                // 2048 words
		for (u32 i = threadIdx.x; i < keys; i += BK_WG) tot[i] = 0;
		__syncthreads();
		for (u32 i = threadIdx.x; i < nb; i += BK_WG) atomicAdd(&tot[(u32)(src[i] >> ELEM_CTX_SHIFT) & kmask], 1u);
		__syncthreads();
		u32 carry = 0;
		for (u32 k0 = 0; k0 < keys; k0 += BK_WG)
		{
			const u32 i = k0 + threadIdx.x;
			const u32 v = i < keys ? tot[i] : 0;
			u32 t;
			const u32 ex = block_excl_scan(v, &t);
			if (i < keys) s_pre[i] = (u16)(carry + ex);
			carry += t;
		}
		if (threadIdx.x == 0) s_pre[keys] = (u16)nb;            // nb <= BK_LIMIT < 2^16
		__syncthreads();
	}

	BkRow<N> R; R.reset();
	u32 a = 0, b = keys, p0 = 0;
	for (;;)
	{
		u32 round_total = nb;
		if (multi)
		{
			if (p0 == 0)
			{	// keys [a, b): as many as fit the tile; a key that does not fit on its own is replayed in chunks (b = a + 1)
				const u32 base = s_pre[a];
				u32 l = a + 1, h = keys;                            // largest b in [a + 1, keys] with s_pre[b] - base <= BK_CAP, if any
				while (l < h) { const u32 m = (l + h + 1) / 2; if ((u32)s_pre[m] - base <= BK_CAP) l = m; else h = m - 1; }
				b = l;
			}
			round_total = (u32)s_pre[b] - (u32)s_pre[a];
		}
		const bool chunked = round_total > BK_CAP;              // one key, replayed BK_CAP symbols at a time by wave 0
		const u32 n_r = round_total - p0 < BK_CAP ? round_total - p0 : BK_CAP;

		if (round_total)
		{
			for (u32 i = threadIdx.x; i < BK_WAVES * ((1 << BK_MAX_LB) / 2); i += BK_WG) (&s_cnt[0][0])[i] = 0;
			__syncthreads();
			// counts per wave and key
			for (u32 i0 = w_lo; i0 < w_hi; i0 += 64)
			{
				const u32 i = i0 + lane;
				const bool valid = i < w_hi;
				const u32 d = valid ? (u32)(src[i] >> ELEM_CTX_SHIFT) & kmask : 0u;
				if (valid && d >= a && d < b) atomicAdd(&s_cnt[w][d >> 1], 1u << ((d & 1u) * 16u));
			}
			__syncthreads();
			{	// counts -> tile positions: key-major, then wave (the stable order)
				const u32 wpt = words >= BK_WG ? words / BK_WG : 1u;   // counter words per thread
				const u32 x0 = threadIdx.x * wpt;
				u32 mine = 0;
				if (x0 < words)
					for (u32 x = x0; x < x0 + wpt; ++x)
#pragma unroll
						for (u32 ww = 0; ww < BK_WAVES; ++ww) { const u32 v = s_cnt[ww][x]; mine += (v & 0xFFFFu) + (v >> 16); }
				u32 tot;
				u32 run = block_excl_scan(mine, &tot);
				if (x0 < words)
					for (u32 x = x0; x < x0 + wpt; ++x)
					{
						u32 c[BK_WAVES];
#pragma unroll
						for (u32 ww = 0; ww < BK_WAVES; ++ww) c[ww] = s_cnt[ww][x];
						u32 lo_pos[BK_WAVES];
#pragma unroll
						for (u32 ww = 0; ww < BK_WAVES; ++ww) { lo_pos[ww] = run; run += c[ww] & 0xFFFFu; }
#pragma unroll
						for (u32 ww = 0; ww < BK_WAVES; ++ww) { s_cnt[ww][x] = lo_pos[ww] | (run << 16); run += c[ww] >> 16; }
					}
			}
			__syncthreads();
			// elements -> the tile
			for (u32 i0 = w_lo; i0 < w_hi; i0 += 64)
			{
				const u32 i = i0 + lane;
				const bool valid = i < w_hi;
				const u64 el = valid ? src[i] : 0ull;
				const u32 d = (u32)(el >> ELEM_CTX_SHIFT) & kmask;
				const bool in = valid && d >= a && d < b;
				const u32 sh = (d & 1u) * 16u;
				u32 pos;
				if (ATOMIC)
				{
					u32 old = 0;
					if (in) old = atomicAdd(&s_cnt[w][d >> 1], 1u << sh);
					pos = (old >> sh) & 0xFFFFu;
#ifdef DSRC_EMU_BUILD
					(void)__ballot(true);
#endif
				}
				else
				{
					u64 peers = __ballot(in);
#pragma unroll
					for (u32 bb = 0; bb < BK_MAX_LB; ++bb)
					{
						const u64 m = __ballot((d >> bb) & 1u);
						peers &= ((d >> bb) & 1u) ? m : ~m;
					}
					const u32 before = in ? (s_cnt[w][d >> 1] >> sh) & 0xFFFFu : 0u;
					const u32 r = (u32)__popcll(peers & lanemask_lt());
					pos = before + r;
					const u64 sync = __ballot(true);
					if (in && r == 0 && sync) atomicAdd(&s_cnt[w][d >> 1], (u32)__popcll(peers) << sh);
				}
				if (in && pos >= p0 && pos - p0 < BK_CAP) s_el[pos - p0] = el;
			}
			__syncthreads();
			// ranges of the waves: nominal quarter boundaries moved up to the next segment head
			if (chunked) { if (threadIdx.x == 0) { s_bnd[0] = 0; s_bnd[1] = n_r; s_bnd[2] = n_r; s_bnd[3] = n_r; s_bnd[4] = n_r; } }
			else
			{
				const u32 step = (n_r + BK_WAVES * 64 - 1) / (BK_WAVES * 64) * 64;
				const u32 nom = w * step;
				const u32 at = w == 0 ? 0u : (nom < n_r ? bk_next_head(s_el, nom, n_r) : n_r);
				if (lane == 0) { s_bnd[w] = at; if (w == 0) s_bnd[BK_WAVES] = n_r; }
			}
			__syncthreads();
			{
				const u32 r_lo = s_bnd[w], r_hi = s_bnd[w + 1];
				if (!(chunked && p0 > 0)) R.reset();                  // a later chunk of a key continues the row wave 0 carries
				bk_replay_range<N>(s_el, s_bin, r_lo, r_hi, n_r, R, s_tail[w]);
			}
			__syncthreads();
			if (BINNED)
			{	// the tile's records, grouped by time bin: a run per bin at the position the bin's fill counter gives
				for (u32 i = threadIdx.x; i < n_bins; i += BK_WG) s_hist[i] = 0;
				__syncthreads();
				for (u32 i = threadIdx.x; i < n_r; i += BK_WG) atomicAdd(&s_hist[s_bin[i]], 1u);
				__syncthreads();
				for (u32 i = threadIdx.x; i < n_bins; i += BK_WG)
				{
					const u32 c = s_hist[i];
					if (c) s_gbase[i] = atomicAdd(&bk[j.bk_fill + i], c);
					s_hist[i] = 0;
				}
				__syncthreads();
				for (u32 i = threadIdx.x; i < n_r; i += BK_WG)
				{
					const u32 bin = s_bin[i];
					const u32 r = atomicAdd(&s_hist[bin], 1u);
					recs[(bin << BK_TB) + s_gbase[bin] + r] = s_el[i];
				}
			}
			else
			{
				for (u32 i = threadIdx.x; i < n_r; i += BK_WG)
				{
					const u64 v = s_el[i];
					recs[((u32)s_bin[i] << BK_TB) | (u32)(v >> 48)] = v & 0xFFFFFFFFFFFFull;
				}
			}
			__syncthreads();
		}
		if (!multi) break;
		if (p0 + n_r < round_total) p0 += n_r;
		else { a = b; p0 = 0; if (a >= keys) break; }
	}
}

// ---- k_place: a time bin's records into stream order, in place ------------------------------------------------------------------------
// Grid: x = time bin, y = stream of the slice.
#define PLACE_WG 1024
__global__ void __launch_bounds__(PLACE_WG) k_place(const CtxJob* jobs, RcPack* rec_pool, const u32* bk)
{
	__shared__ u64 s_rec[BK_BIN];
	const CtxJob j = jobs[blockIdx.y];
	if (!j.bk_on || !j.bk_binned || bk[j.jid]) return;
	const u32 bin = blockIdx.x;
	const u32 t0 = bin << BK_TB;
	if (t0 >= j.n) return;
	const u32 cnt = j.n - t0 < BK_BIN ? j.n - t0 : BK_BIN;
	RcPack* r = rec_pool + j.trip + t0;
	for (u32 i = threadIdx.x; i < cnt; i += PLACE_WG)
	{
		const u64 v = r[i];
		s_rec[(u32)(v >> 48)] = v & 0xFFFFFFFFFFFFull;
	}
	__syncthreads();
	for (u32 i = threadIdx.x; i < cnt; i += PLACE_WG) r[i] = s_rec[i];
}
