// Order-context quality streams, FOUR BLOCKS PER WAVE: TQualityOrderModeler::Decode + T*QualityEncoder::Decode +
// TQualityModelExt::DecodeSymbol (src/QualityOrderModeler.h:49-65, src/QualityEncoder.h:77-143,248-263,306-326),
// TSymbolCoderRC<N>::DecodeSymbol (src/SymbolCoderRC.h:50-91), RangeDecoder (src/RangeCoder.h:90-142).
//
// Why this shape (measured): with the wave on ONE stream (k_dec_qrc) a symbol costs ~200 instructions, a wave
// issues one instruction per four cycles, and the pass takes 1.1 s + 0.7 s x (waves per SIMD) whether the model tables are 64 MB or
// 64 KB per block -- the decoder is bound by instruction issue, not by memory.  Nearly all of those instructions are the coder's
// and the context's bookkeeping, the same for every stream.  So a wave takes four streams, one per ROW of 16 lanes: the
// bookkeeping is ordinary per-lane code that the four rows share instruction by instruction, a model row (N inclusive counts) is
// spread over its 16 lanes (N / 16 counts per lane), and what has to cross lanes stays inside a row: the total is a DPP
// rotate-and-max, the symbol index a rotate-and-add of per-lane counts of `count x r <= buffer`, the two counts around the index
// two shuffles.  Everything is executed by all 64 lanes in wave-uniform control flow; a row whose stream has ended idles.
#pragma once
#include "k_dec_q0.h"

// ---- operations inside a row of 16 lanes -------------------------------------------------------------------------------------
template <u32 K> __device__ __forceinline__ u32 q4_ror(u32 v)          // rotate by K lanes inside the row
{
#if defined(DSRC_EMU_BUILD) || defined(DSRC_Q4_NO_DPP)
	const u32 lane = lane_id();
	return (u32)__shfl((int)v, (int)((lane & 48u) | ((lane + K) & 15u)));
#else
	return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x120 + K, 0xf, 0xf, false);      // row_ror:K
#endif
}
__device__ __forceinline__ u32 q4_shr1(u32 v)                           // lane i of a row gets lane i - 1's value, lane 0 gets 0
{
#if defined(DSRC_EMU_BUILD) || defined(DSRC_Q4_NO_DPP)
	const u32 lane = lane_id();
	const u32 p = (u32)__shfl((int)v, (int)(lane ? lane - 1 : 0));
	return (lane & 15u) ? p : 0u;
#else
	u32 r = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);                 // row_shr:1, out-of-row reads 0
	asm volatile("" : "+v"(r));            // keep it a v_mov_b32_dpp: folded into the subtraction that follows it gave wrong rows on gfx950 (hipcc 7.2)
	return r;
#endif
}
__device__ __forceinline__ u32 q4_max(u32 v)
{
	u32 t;
	t = q4_ror<8>(v); v = v > t ? v : t;
	t = q4_ror<4>(v); v = v > t ? v : t;
	t = q4_ror<2>(v); v = v > t ? v : t;
	t = q4_ror<1>(v); v = v > t ? v : t;
	return v;
}
__device__ __forceinline__ u32 q4_sum(u32 v)
{
	v += q4_ror<8>(v); v += q4_ror<4>(v); v += q4_ror<2>(v); v += q4_ror<1>(v);
	return v;
}
// inclusive prefix sum over the lanes of a row
__device__ __forceinline__ u32 q4_scan(u32 v)
{
#if defined(DSRC_EMU_BUILD) || defined(DSRC_Q4_NO_DPP)
	const u32 lane = lane_id(), l = lane & 15u;
	for (u32 dd = 1; dd < 16; dd <<= 1) { const u32 t = (u32)__shfl((int)v, (int)(lane >= dd ? lane - dd : 0)); if (l >= dd) v += t; }
	return v;
#else
	int x = (int)v;
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);      // row_shr:1
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);      // row_shr:2
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);      // row_shr:4
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);      // row_shr:8
	return (u32)x;
#endif
}


// this lane's E counts of a row: 2 E bytes at `p`
template <u32 E> __device__ __forceinline__ void q4_load(const u8* p, u32 (&raw)[(E + 1) / 2])
{
	if (E == 1) raw[0] = *(const u16*)p;
	else if (E == 2) raw[0] = *(const u32*)p;
	else if (E == 4) { const uint2 t = *(const uint2*)p; raw[0] = t.x; raw[1] = t.y; }
	else { const uint4 t = *(const uint4*)p; raw[0] = t.x; raw[1] = t.y; raw[2] = t.z; raw[3] = t.w; }
}
template <u32 E> __device__ __forceinline__ void q4_store(u8* p, const u32 (&cur)[E])
{
	if (E == 1) *(u16*)p = (u16)cur[0];
	else if (E == 2) *(u32*)p = cur[0] | (cur[1] << 16);
	else if (E == 4) *(uint2*)p = make_uint2(cur[0] | (cur[1] << 16), cur[2] | (cur[3] << 16));
	else *(uint4*)p = make_uint4(cur[0] | (cur[1] << 16), cur[2] | (cur[3] << 16), cur[4 % E] | (cur[5 % E] << 16), cur[6 % E] | (cur[7 % E] << 16));
}

template <u32 N>
__global__ void __launch_bounds__(64) k_dec_qrc4(const u8* in, const DecDesc* desc, DecState* st, const DecTab* tabs, u32 n_tabs, RecPools rp, u8* out,
												  u32* tables, DecParams prm, u32 scheme)
{
	constexpr u32 E = N >= 16 ? N / 16 : 1;                   // counts per lane
	constexpr u32 RAW = (E + 1) / 2;
	constexpr u32 LIM = (1u << 16) - 2 * N;
	constexpr u32 abits = N == 8 ? 3u : N == 16 ? 4u : N == 32 ? 5u : N == 64 ? 6u : 7u;
	__shared__ u8 s_sym[4][256];
	__shared__ u32 s_par[4][4];
	const u32 lane = lane_id(), g = lane >> 4, l = lane & 15u, gbase = lane & 48u;
	const u32 ti = blockIdx.x * 4 + g;
	const bool has = ti < n_tabs;
	const DecTab tb = tabs[has ? ti : 0];
	const u32 b = tb.block;
	DecState* S = &st[b];
	bool active = has && S->err == 0;
	const DecDesc d = desc[b];
	BitSrc s; s.p = in + d.in_off; s.size = d.in_size; s.err = 0; s.bit = (u64)S->qua_pos * 8;
	if (l == 0)
	{
		for (u32 i = 0; i < 256; ++i) s_sym[g][i] = 255;
		if (active && !prm.lossy)
		{
			(void)bs_byte(s);                             // scheme byte: the tag kernel has left it in q_scheme
			bs_align(s);
			u32 n = 0;
			for (u32 i = 0; i < 256; ++i) if (bs_bit(s)) s_sym[g][n++] = (u8)i;
			bs_align(s);
		}
		s_par[g][0] = (u32)s.bit; s_par[g][1] = (u32)(s.bit >> 32); s_par[g][2] = s.err;
	}
	__syncthreads();
	s.bit = ((u64)s_par[g][1] << 32) | s_par[g][0];
	u32 err = s_par[g][2];
	QrcScheme qs = {0u, 0u, 0u, 0u};
	if (!qrc_scheme(prm.quality_order, prm.lossy, scheme, &qs) || qs.n != N) err |= DEC_ERR_FORMAT;
	const u32 cnt = prm.lossy ? 8u : S->q_cnt;
	if (active && (cnt == 0 || cnt > N || (!prm.lossy && S->q_scheme != scheme))) err |= DEC_ERR_FORMAT;
	const u32 ord = qs.ord, rescale = qs.rescale, rshift = dec_int_log2(qs.rescale), lossy = prm.lossy;
	const bool translate = qs.translate != 0;
	if ((((u64)1 << (abits * ord)) * rescale * N / 2) > tb.words) err |= DEC_ERR_POOL;
	const u32 sym_mask = N - 1;
	const u32 bits_lo = (ord / 2) * abits, bits_hi = (ord / 2 + 1) * abits;
	const u32 hash_mask = (1u << (ord * abits)) - 1u;
	const u32 swap_mask = ((1u << bits_lo) - 1u) | ~((1u << bits_hi) - 1u);
	u8* tbytes = (u8*)(tables + tb.off);
	u8* text = out + d.out_off;
	const u32 voff = l * 2 * E;                              // this lane's bytes inside a row
	const u32 gi0 = l * E;                                   // index of its first count
	const bool live = gi0 < N;

	const u64 g0 = d.rec_base;
	const u32 n_recs = S->n_recs;
	u32 k = 0, d_total = 0, ql = 0;
	active = active && !err;
	if (active)
		for (; k < n_recs; ++k)
		{	// records without a quality line code nothing
			ql = rp.len[g0 + k];
			if (ql) break;
			if (l == 0) { rp.kept[g0 + k] = 0; rp.d_off[g0 + k] = d_total; }
		}
	// RangeDecoder::Start (src/RangeCoder.h:97-106): eight bytes; buffer < range on every stream an encoder writes, so the first
	// four are zero and the buffer is 32 bits wide from here on (anything else is refused)
	LWin win;
	u32 buffer = 0, range = 0xFFFFFFFFu; u64 low = 0;
	{
		const u64 first = lw_start(win, s);
		if (active && (first >> 32)) err |= DEC_ERR_FORMAT;
		buffer = (u32)first;
	}
	active = active && !err && k < n_recs;
	u32 qoff = active ? rp.qual_off[g0 + k] : 0u;
	u32 hash = 0, sym_buf = 0, j = 0, pctx = 0, rem = 0, ncount = 0, pk = 0, ri = 0, max_idx = 0;
	u32 upd[E], nxt[RAW];
#pragma unroll
	for (u32 e = 0; e < E; ++e) upd[e] = live ? gi0 + e + 1 : 0u;     // row 0 of a fresh table is 1, 2, .., N
#pragma unroll
	for (u32 e = 0; e < RAW; ++e) nxt[e] = 0;
	bool same = true;
	u32 end_pos = lw_pos(win);                                // where the stream ends (a stream without symbols: behind the coder's eight bytes)
	double nf = dec_div_prep(range);
	u32 pn, rem2, nb, hpre, base_next;
#define Q4_PREP() do { \
		pn = 0; rem2 = 0; \
		if (j + 1 != ql) { pn = pctx; rem2 = rem + rescale; while (rem2 >= ql && ql) { rem2 -= ql; ++pn; } } \
		const u32 h2_ = hash << abits; \
		nb = (h2_ >> bits_lo) & sym_mask; \
		hpre = (h2_ & swap_mask) | (((nb + sym_buf) >> 1) << bits_lo); \
		base_next = ((hpre & hash_mask) << rshift) + pn; } while (0)
	Q4_PREP();
	dec_vm_drain();
	while (__any(active))
	{
		// ---- the row --------------------------------------------------------------------------------------------------------
		u32 cur[E];
#pragma unroll
		for (u32 e = 0; e < E; ++e) cur[e] = same ? upd[e] : (nxt[e / 2] >> (16 * (e & 1))) & 0xFFFFu;
		const u32 total = q4_max(cur[E - 1]);
		const u32 r = dec_div(nf, total ? total : 1u);
		// ---- the symbol: how many counts have count * r <= buffer (count * r <= total * r <= range: 32 bits) -----------------------
		u32 c_le = 0;
#pragma unroll
		for (u32 e = 0; e < E; ++e) c_le += (live && cur[e] * r <= buffer) ? 1u : 0u;
		u32 idx = q4_sum(c_le);
		if (idx >= N) { if (active) err |= DEC_ERR_FORMAT; idx = N - 1; }      // buffer >= total * r: the reference walks off the row
		// ---- request the next row ------------------------------------------------------------------------------------------------
		const u32 ri_next = base_next + (idx << rshift);
		const u32 wl_pos = win.nx + 8;
		u64 wl = 0;
		if (active)
		{
			if (live) q4_load<E>(tbytes + ((ri_next << (abits + 1)) + voff), nxt);
			wl = lw_load(win.p, win.size, wl_pos);
		}
		// ---- the counts around the index: two shuffles inside the row ---------------------------------------------------------------
		u32 hi, lo;
		{
			const u32 im = idx ? idx - 1 : 0u;
			u32 sh = cur[0], sl = cur[0];
#pragma unroll
			for (u32 e = 1; e < E; ++e) { if ((idx % E) == e) sh = cur[e]; if ((im % E) == e) sl = cur[e]; }
			hi = (u32)__shfl((int)sh, (int)(gbase + idx / E));
			lo = (u32)__shfl((int)sl, (int)(gbase + im / E));
			if (idx == 0) lo = 0;
		}
		const u32 rr = lo * r;
		buffer -= rr; low += rr;
		range = r * (hi - lo);
		if (range == 0) { if (active) err |= DEC_ERR_FORMAT; range = 0xFFFFFFFFu; }
		{
			// bytes to shift in: 0 .. 3; one shift does it unless the carry clamp can fire (it needs bits 39..24 of `low` all set,
			// src/RangeCoder.h:126-130) or the window holds fewer bytes than that
			const u32 nb8 = range <= 0x00FFFFFFu ? (u32)__clz((int)range) >> 3 : 0u;
			const bool fast = (((u32)(low >> 24)) & 0xFFFFu) != 0xFFFFu && win.left >= nb8;
			if (fast)
			{
				const u32 sh = nb8 * 8;
				range <<= sh; low <<= sh;
				buffer = (buffer << sh) | (u32)(((u64)(u32)(win.w0 >> 32) << sh) >> 32);
				win.w0 <<= sh; win.left -= nb8;
			}
			else if (active)
			{	// RangeDecoder::DecodeFrequency's loop as written (src/RangeCoder.h:122-135)
				while (range <= 0x00FFFFFFu)
				{
					if ((low ^ (low + range)) & 0xFF00000000000000ull)
					{
						const u32 l32 = (u32)low;
						range = (l32 | 0x00FFFFFFu) - l32;
					}
					buffer = (buffer << 8) + lw_byte(win);
					low <<= 8; range <<= 8;
					if (range == 0) { err |= DEC_ERR_FORMAT; range = 0xFFFFFFFFu; break; }
				}
			}
		}
		if (win.left == 0 && win.wp_pos == win.nx + 8)
		{	// the window ran dry exactly: the next 8 bytes are in registers already (lw_byte would do the same inside its loop)
			win.w0 = win.w1; win.w1 = win.wp; win.nx += 8; win.left = 8;
		}
		nf = dec_div_prep(range);
		// ---- the row: +2 on the symbol = +2 on every inclusive count from it on; Rescale() now instead of at the next visit ------------
#pragma unroll
		for (u32 e = 0; e < E; ++e) cur[e] += (gi0 + e >= idx) ? 2u : 0u;
		if (__any(active && total + 2 >= LIM))
		{
			const u32 below = q4_shr1(cur[E - 1]);
			u32 c[E], acc = 0;
#pragma unroll
			for (u32 e = 0; e < E; ++e)
			{
				u32 x = live ? cur[e] - (e ? cur[e - 1] : below) : 0u;
				x -= x >> 1;
				acc += x; c[e] = acc;
			}
			const u32 before = q4_scan(acc) - acc;
			if (total + 2 >= LIM)
#pragma unroll
				for (u32 e = 0; e < E; ++e) cur[e] = before + c[e];
		}
#pragma unroll
		for (u32 e = 0; e < E; ++e) upd[e] = cur[e];
		if (active && live) q4_store<E>(tbytes + ((ri << (abits + 1)) + voff), cur);
		// ---- the symbol's character; four at a time go out ---------------------------------------------------------------------------
		const u32 qv = translate ? (u32)s_sym[g][idx] : idx;
		if (active) max_idx = max_idx > idx ? max_idx : idx;
		pk |= qv << (8 * (j & 3u));
		ncount += q_special(qv, lossy) ? 1u : 0u;
		++j;
		if (active && l == 0)
		{
			u8* q = text + qoff;
			if ((j & 3u) == 0) *(dec_u32_unaligned*)(q + j - 4) = pk;
			else if (j == ql) for (u32 t = 0; t < (j & 3u); ++t) q[(j & ~3u) + t] = (u8)(pk >> (8 * t));
		}
		if ((j & 3u) == 0) pk = 0;
		same = ri_next == ri;
		ri = ri_next;
		hash = hpre | idx; sym_buf = nb; pctx = pn; rem = rem2;
		if (err) active = false;
		if (active && j == ql)
		{
			if (l == 0) { rp.kept[g0 + k] = (u16)(ql - ncount); rp.d_off[g0 + k] = d_total; }
			d_total += ql - ncount;
			for (++k; k < n_recs; ++k)
			{
				ql = rp.len[g0 + k];
				if (ql) break;
				if (l == 0) { rp.kept[g0 + k] = 0; rp.d_off[g0 + k] = d_total; }
			}
			if (k == n_recs) { active = false; end_pos = lw_pos(win); }          // a row whose stream has ended keeps executing: its results are taken here
			else { qoff = rp.qual_off[g0 + k]; j = 0; ncount = 0; pk = 0; }
		}
		Q4_PREP();
		win.wp = wl; win.wp_pos = wl_pos;
	}
#undef Q4_PREP
	if (has && l == 0 && S->err == 0)
	{
		if (!err && max_idx >= cnt) err |= DEC_ERR_FORMAT;               // a symbol the block's alphabet does not have: no encoder writes it
		const u32 end = end_pos;
		if (end > s.size) err |= DEC_ERR_TRUNC;
		S->d_total = d_total;
		if (!err) S->dna_pos = end;
		S->err |= err;
	}
}
