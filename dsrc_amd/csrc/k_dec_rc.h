// The range-decoded levels of the decompressor (-q1/-q2 quality, -d1..-d3 DNA), laid out for LATENCY.
// Replaces TQualityOrderModeler::Decode + T*QualityEncoder::Decode + TQualityModelExt::DecodeSymbol
// (src/QualityOrderModeler.h:49-65, src/QualityEncoder.h:77-143,248-263,306-326), TDnaRCOrderModeler::Decode
// (src/DnaModelerRCO.h:62-79), TSymbolCoderRC<N>::DecodeSymbol (src/SymbolCoderRC.h:50-91) and RangeDecoder
// (src/RangeCoder.h:90-142).
//
// A decoded stream advances one symbol per DEPENDENT row read: the address of the next model row is a function of the symbol
// just decoded.  The rate of a pass is (chains in flight) / (time per symbol), so everything here is about what sits between
// "row arrives" and "next row requested":
//   * the model table is PRIVATE to the decoder, so its layout is ours: rows are dense (hash * rescale + pctx: every row is
//     reachable), sized per block from the block's scheme, cleared by a streaming kernel -- and a quality row holds INCLUSIVE
//     CUMULATIVE counts, so neither a prefix scan nor the row total has to be computed when it arrives;
//   * the symbol is found without dividing by the range: incl[i] * r > buffer  <=>  incl[i] > floor(buffer / r), one 64-bit
//     multiply per lane and a ballot; r = floor(range / total) is one f64 reciprocal with a Newton step, exact (dec_div);
//   * Rescale() is applied when a row is WRITTEN (the reference applies it at the next visit; nothing else reads the row in
//     between), so a row is ready when it arrives;
//   * the next row is requested as soon as the symbol index is known; the coder's state, the counter update, the output byte and
//     the context bookkeeping of the next symbol all run in the shadow of that request.
// DNA (4 or 8 counters per row) is decoded ONE LANE PER BLOCK, 64 blocks per wave: the N candidate rows of the next symbol are
// consecutive and are requested before this symbol is decoded.  (Touching the line of the 16 candidates of the symbol after that
// one symbol earlier does not work on this machine: a wave's vector loads return in order, so the younger request that hits
// could not be consumed before the older one that misses.)
#pragma once
#include "k_dec.h"

// every vector-memory request of this wave has completed: s_waitcnt vmcnt(0) (gfx9 encoding, expcnt / lgkmcnt left at their maxima)
__device__ __forceinline__ void dec_vm_drain() { __builtin_amdgcn_s_waitcnt(0x0F70); }

#ifndef DEC_SWIZZLE
#define DEC_SWIZZLE 0
#endif
#ifndef DEC_QRC_LEAN
#define DEC_QRC_LEAN 1          // alphabets of <= 64 symbols take qrc_decode_lean (round 5); 0: the loop of rounds 3-4 for every alphabet
#endif

struct DecTab            // one model table of one block in the table region (host -> device)
{
	u64 off;             // u32 words from the start of the region
	u64 words;           // u32 words
	u32 block;           // block index in the batch
	u32 n;               // quality: alphabet size of the cumulative fill pattern (row = 1, 2, .., n); 0: all counters 1 (DNA, serial decoder)
};

// ---- streaming clear of the model tables of one round --------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_dec_fill(u32* tables, const DecTab* tabs)
{
	const DecTab t = tabs[blockIdx.y];
	uint4* dst = (uint4*)(tables + t.off);
	const u64 n16 = t.words / 4;                             // table sizes are multiples of 16 bytes
	const u32 n = t.n;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x)
	{
		uint4 v;
		if (n == 0) { v.x = v.y = v.z = v.w = 0x00010001u; }
		else
		{	// eight u16 elements e0 .. e0 + 7 of a row of n (n >= 8, a power of two): value = index in the row + 1
			const u32 e0 = (u32)((i * 8) & (u64)(n - 1)) + 1;
			v.x = e0 | ((e0 + 1) << 16); v.y = (e0 + 2) | ((e0 + 3) << 16); v.z = (e0 + 4) | ((e0 + 5) << 16); v.w = (e0 + 6) | ((e0 + 7) << 16);
		}
		dst[i] = v;
	}
}

// ---- exact floor(n / d) for n < 2^32, 1 <= d < 2^16 ------------------------------------------------------------------------
// x1 = one Newton step on the hardware reciprocal (relative error e0 <= 2^-22 -> e1 <= 2^-43); nf = n * (1 + 2^-40), prepared
// when n becomes known (off the critical path).  nf * x1 = (n / d) * (1 + 2^-40) * (1 + e1) lies in [n / d, n / d * (1 + 2^-39)]:
// it cannot fall below n / d (the bias outweighs e1 and the roundings of 2^-53), and it cannot reach the next integer, which is
// at least 1 / d away while n / d * 2^-39 < 2^-7 / d.  So the truncation is the quotient.  Checked for every d on the device by
// dsrcgpu_selftest (k_selftest_dec), like the encoder's reciprocal.
__device__ __forceinline__ double dec_rcp(double x)
{
#ifdef DSRC_EMU_BUILD
	return 1.0 / x;
#else
	return __builtin_amdgcn_rcp(x);
#endif
}
__device__ __forceinline__ double dec_div_prep(u32 n) { return (double)n * (1.0 + 0x1p-40); }
__device__ __forceinline__ u32 dec_div(double nf, u32 d)
{
	const double df = (double)d;
	double x = dec_rcp(df);
	x = __builtin_fma(__builtin_fma(-df, x, 1.0), x, x);
	return (u32)(nf * x);
}

__global__ void __launch_bounds__(256) k_selftest_dec(u32* bad)
{
	const u32 d = blockIdx.x * blockDim.x + threadIdx.x + 1;
	if (d >= 65536) return;
	u32 wrong = 0;
	for (u32 k = 0; k < 96; ++k)
	{
		// multiples of d, the values just below them, and a spread of others; the largest quotients included
		const u32 q = k < 32 ? (0xFFFFFFFFu / d) >> k : (u32)((u64)(k * 2654435761u) % ((u64)(0xFFFFFFFFu / d) + 1));
		const u64 base = (u64)q * d;
		const u32 ns[3] = {(u32)base, base ? (u32)(base - 1) : 0u, (u32)(base + d - 1 < 0xFFFFFFFFull ? base + d - 1 : 0xFFFFFFFFull)};
		for (u32 j = 0; j < 3; ++j)
			if (dec_div(dec_div_prep(ns[j]), d) != ns[j] / d) wrong = 1;
	}
	if (wrong) atomicAdd(bad, 1u);
}

// ---- the coder's bytes for a wave-uniform stream: scalar loads -----------------------------------------------------------------
// The block is read-only while the pass runs, so its bytes can come through the scalar cache (s_load_dword: constant address
// space), whose loads return out of order and on their own counter -- a vector load here would have to be waited for together
// with the row request that is in flight (a wave's vector loads return in order).  Two dwords are kept requested ahead; the
// index is clamped to the dword that holds the block's last byte, so nothing outside the block's dwords is touched.
#ifdef DSRC_EMU_BUILD
#define CONST_AS
#else
#define CONST_AS __attribute__((address_space(4)))
#endif
struct UWin { const CONST_AS u32* p4; u32 cw, left, q0, q1, nextw, lastw, origin; };

__device__ __forceinline__ u32 uw_bswap(u32 x) { return __builtin_bswap32(x); }
__device__ __forceinline__ void uw_start(UWin& w, const BitSrc& s)
{
	const u64 a = (u64)s.p, x = a + (s.bit >> 3);
	const u64 w0 = x & ~3ull;
	const u32 mis = (u32)(x & 3ull);
	w.p4 = (const CONST_AS u32*)w0;
	w.lastw = (u32)((((a + s.size - 1) & ~3ull) - w0) >> 2);
	w.origin = (u32)(s.bit >> 3) - mis;                             // block position of byte 0 of dword 0
	w.cw = uw_bswap(w.p4[0]) << (8 * mis); w.left = 4 - mis;
	w.q0 = w.p4[1 < w.lastw ? 1 : w.lastw]; w.q1 = w.p4[2 < w.lastw ? 2 : w.lastw];
	w.nextw = 3;
}
__device__ __forceinline__ u32 uw_byte(UWin& w)
{
	if (w.left == 0)
	{
		w.cw = uw_bswap(w.q0); w.q0 = w.q1;
		w.q1 = w.p4[w.nextw < w.lastw ? w.nextw : w.lastw];
		++w.nextw; w.left = 4;
	}
	const u32 b = w.cw >> 24;
	w.cw <<= 8; --w.left;
	return b;
}
// bytes of the block consumed so far, as a position inside the block
__device__ __forceinline__ u64 uw_pos(const UWin& w) { return (u64)w.origin + (u64)(w.nextw - 2) * 4 - w.left; }

// ---- quality: the wave on one stream ----------------------------------------------------------------------------------------
__device__ __forceinline__ u32 qrc_readlane(u32 v, u32 l)
{
	return (u32)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane((int)l));
}

// lane i gets lane i - 1's value, lane 0 gets 0 (wave_shr:1).  The result only ever feeds v_readlane here: a DPP move that the
// compiler folds into a following SUBTRACTION came out wrong on gfx950 (round 3's four-streams-per-wave experiment, removed in round 4: git show 36bc838)
__device__ __forceinline__ u32 qrc_up1(u32 v)
{
#ifdef DSRC_EMU_BUILD
	u32 p = __shfl_up(v, 1); if (lane_id() == 0) p = 0; return p;
#else
	u32 r = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
	asm volatile("" : "+v"(r));
	return r;
#endif
}

// lane (l mod 64) of `old` becomes the wave-uniform v.  (No builtin for v_writelane in this compiler; the lane select goes through M0:
// a VOP3 instruction of gfx9 reads one SGPR besides it.)
__device__ __forceinline__ u32 qrc_writelane(u32 v, u32 l, u32 old)
{
#ifdef DSRC_EMU_BUILD
	return lane_id() == (l & 63u) ? v : old;
#else
	asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(v), "s"(l) : "m0");
	return old;
#endif
}

// element i of a row held one (N <= 64) or two (N == 128: low half = even element) per lane
template <u32 CPL> __device__ __forceinline__ u32 qrc_elem(u32 cur, u32 i)
{
	if (CPL == 1) return qrc_readlane(cur, i);
	const u32 w = qrc_readlane(cur, i >> 1);
	return (i & 1u) ? w >> 16 : w & 0xFFFFu;
}

template <u32 N>
__device__ __forceinline__ void qrc_decode(BitSrc& s, u32* table, u32 ord, u32 rescale, u32 cnt, u32 swz_seed, bool translate, const u8* sym_tab, u32 lossy,
										   const DecDesc& d, DecState* S, RecPools rp, u8* text)
{
	constexpr u32 CPL = N > 64 ? 2u : 1u;
	constexpr u32 LANES = N / CPL;
	constexpr u32 LIM = (1u << 16) - 2 * N;
	const u32 lane = lane_id();
	const bool live = lane < LANES;
	const u32 abits = dec_int_log2(N);
	const u32 sym_mask = N - 1;
	const u32 bits_lo = (ord / 2) * abits, bits_hi = (ord / 2 + 1) * abits;
	const u32 hash_mask = (1u << (ord * abits)) - 1u;                   // ord * abits <= 21 for every scheme
	const u32 swap_mask = ((1u << bits_lo) - 1u) | ~((1u << bits_hi) - 1u);
	u16* tab16 = (u16*)table;
	// Where a row lives: its number XOR a per-block constant.  Blocks of one file have the same hot contexts, and tables that are laid
	// out alike put them on the same memory channels: 2400 waves then queue on a few of them (measured: 2.8 instead of 1.7 s per
	// pass).  The XOR is a bijection on the 2^k rows; which row is which is nobody's business but this decoder's.
	const u32 swz = DEC_SWIZZLE ? swz_seed & (((hash_mask + 1u) * rescale) - 1u) : 0u;
	// symbol value of counter i, in the lane that holds counter i (two per lane when N == 128)
	u32 tr_v = 0;
	if (translate) tr_v = CPL == 1 ? (live ? (u32)sym_tab[lane] : 0u) : ((u32)sym_tab[2 * lane] | ((u32)sym_tab[2 * lane + 1] << 16));
	else if (CPL == 1) tr_v = lane;

	const u64 g0 = d.rec_base;
	const u32 n_recs = S->n_recs;
	u32 k = 0, d_total = 0, err = 0;
	u32 ql = 0;
	for (; k < n_recs; ++k)
	{	// records without a quality line code nothing
		ql = rp.len[g0 + k];
		if (ql) break;
		if (lane == 0) { rp.kept[g0 + k] = 0; rp.d_off[g0 + k] = d_total; }
	}
	// RangeDecoder::Start (src/RangeCoder.h:97-106): eight bytes into the buffer
	struct { u64 low, buffer; u32 range; } rd;
	UWin win; uw_start(win, s);
	rd.low = 0; rd.range = 0xFFFFFFFFu; rd.buffer = 0;
	for (u32 i = 0; i < 8; ++i) rd.buffer = (rd.buffer << 8) | uw_byte(win);
	if (k < n_recs)
	{
		u8* q = text + rp.qual_off[g0 + k];
		u32 hash = 0, sym_buf = 0, j = 0, pctx = 0, rem = 0, ncount = 0, mine = 0;
		u32 ri = 0;                                                   // row of the symbol being decoded: (hash & mask) * rescale + pctx
		u32 max_idx = 0, min_r = 0xFFFFFFFFu;
		// this lane's element(s) of that row; row 0 of a fresh table is 1, 2, .., N (no load: a request pending at the loop's entry would
		// make the compiler's wait at the top of the loop cover the stores of every later iteration as well)
		u32 cur = !live ? 0u : CPL == 1 ? lane + 1 : (2 * lane + 1) | ((2 * lane + 2) << 16);
		double nf = dec_div_prep(rd.range);
		dec_vm_drain();
		// Where the row of the symbol AFTER the one being decoded is, up to that symbol itself: position context, and the hash with
		// its low slot still empty.  It is prepared one symbol ahead (in the shadow of the previous request), so that between
		// "row arrives" and "next row requested" there is only the symbol search.
		// The loop is cut into SEGMENTS so that the body of a symbol carries no test of where in the record it is: a segment ends at
		// a multiple of 64 symbols (they are stored), one symbol before the record's end (the row after the LAST symbol is the next
		// record's first: position context 0 -- QRC_LAST() turns the prepared row into that one) and at the record's end.
		u32 pn, rem2, nb, hpre, base_next;
		// the position context advances by rescale / ql per symbol: quotient and remainder per record, no loop per symbol
		u32 step_q = rescale / ql, step_r = rescale - step_q * ql;
#define QRC_STEP() do { step_q = rescale / ql; step_r = rescale - step_q * ql; } while (0)
#define QRC_POS() do { rem2 = rem + step_r; const u32 c_ = rem2 >= ql ? 1u : 0u; pn = pctx + step_q + c_; rem2 -= c_ ? ql : 0u; } while (0)
#define QRC_PREP() do { \
			QRC_POS(); \
			const u32 h2_ = hash << abits; \
			nb = (h2_ >> bits_lo) & sym_mask; \
			hpre = (h2_ & swap_mask) | (((nb + sym_buf) >> 1) << bits_lo); \
			base_next = (hpre & hash_mask) * rescale + pn; } while (0)
#define QRC_LAST() do { base_next -= pn; pn = 0; rem2 = 0; } while (0)
		QRC_PREP();
		for (;;)
		{
			u32 stop = (j | 63u) + 1u;
			if (j + 1 < ql) { if (stop > ql - 1) stop = ql - 1; }
			else { stop = ql; QRC_LAST(); }
			do
			{
			// ---- the row has arrived: symbol index ---------------------------------------------------------------------
			const u32 total = qrc_elem<CPL>(cur, N - 1);
			const u32 r = dec_div(nf, total);
			min_r = min_r < r ? min_r : r;                               // r == 0 (range < total): no valid stream; tested once at the end
			u32 idx;
			{
				const u32 e_hi = CPL == 1 ? cur : cur >> 16, e_lo = cur & 0xFFFFu;
				// no `live &&`: a lane past the row holds 0, a copy of one of the row's counts (it asks for the row like the others), that
				// copy + 2, or the row's total (after Rescale()) -- never more than the total in lane LANES - 1, so whenever such a lane
				// answers yes, lane LANES - 1 does too, and the lowest yes is a lane of the row
				u64 m = __ballot((u64)e_hi * r > rd.buffer);
				u64 m0 = CPL == 2 ? __ballot((u64)e_lo * r > rd.buffer) : 0ull;
				if (m == 0)
				{	// buffer >= total * r: not a stream the encoder writes; the reference compares with the TRUNCATED quotient
					const u32 cul = div_u64_u32(rd.buffer, r ? r : 1u);
					m = __ballot(live && e_hi > cul);
					if (CPL == 2) m0 = __ballot(live && e_lo > cul);
				}
				if (m == 0) { err |= DEC_ERR_FORMAT; idx = N - 1; }          // the reference walks off the row here
				else
				{
					const u32 l = (u32)__ffsll((long long)m) - 1u;
					idx = CPL == 1 ? l : 2 * l + (((m0 >> l) & 1ull) ? 0u : 1u);
				}
				max_idx = max_idx > idx ? max_idx : idx;                    // a symbol the block's alphabet does not have: no encoder writes it; tested once at the end
			}
			// ---- request the next row -----------------------------------------------------------------------------------
			const u32 ri_next = base_next + idx * rescale;
			u32 nxt = 0;
			// every lane asks (lanes past the row for an element of the row again: the same line, and their copy is never used): no
			// change of the execution mask around the request
			nxt = CPL == 1 ? (u32)tab16[(u64)(ri_next ^ swz) * N + (lane & (LANES - 1))] : table[(u64)(ri_next ^ swz) * (N / 2) + (lane & (LANES - 1))];

			// ---- in its shadow: coder state ------------------------------------------------------------------------------
			u32 hi, lo;
			if (CPL == 1)
			{	// the count below the symbol's: the row shifted up by one lane (lane 0 gets 0), read at the same lane
				hi = (u32)__builtin_amdgcn_readlane((int)cur, (int)idx);
				lo = (u32)__builtin_amdgcn_readlane((int)qrc_up1(cur), (int)idx);
			}
			else { hi = qrc_elem<CPL>(cur, idx); lo = idx ? qrc_elem<CPL>(cur, idx - 1) : 0u; }
			const u32 f = hi - lo;
			const u32 rr = lo * r;                                     // uint32 product
			rd.buffer -= rr; rd.low += rr;
			rd.range = r * f;
			// (if + do-while with the hint: the compiler keeps the renormalisation out of the straight path; as a plain while loop the
			// symbol that shifts nothing in -- two of three -- still took three jumps through the loop's header)
			if (__builtin_expect(rd.range <= 0x00FFFFFFu, 0))
			do
			{
				if ((rd.low ^ (rd.low + rd.range)) & 0xFF00000000000000ull)
				{
					const u32 l32 = (u32)rd.low;
					rd.range = (l32 | 0x00FFFFFFu) - l32;
				}
				rd.buffer = (rd.buffer << 8) + uw_byte(win);
				rd.low <<= 8; rd.range <<= 8;
				if (rd.range == 0) { err |= DEC_ERR_FORMAT; rd.range = 0xFFFFFFFFu; break; }
			} while (rd.range <= 0x00FFFFFFu);
			nf = dec_div_prep(rd.range);
			// ---- the row: +2 on the symbol = +2 on every cumulative count from it on; Rescale() now instead of at the next visit
			{
				const u32 il = CPL == 1 ? idx : idx >> 1;
				if (CPL == 1) { if (lane >= il) cur += 2; }
				else if (lane > il) cur += 0x00020002u;
				else if (lane == il) cur += (idx & 1u) ? 0x00020000u : 0x00020002u;
				if (total + 2 >= LIM)
				{
					if (CPL == 1)
					{
						u32 p = __shfl_up(cur, 1); if (lane == 0) p = 0;
						u32 c = live ? cur - p : 0u;
						c -= c >> 1;
						cur = dec_wave_scan(c);
					}
					else
					{
						const u32 e0 = cur & 0xFFFFu, e1 = cur >> 16;
						u32 p = __shfl_up(e1, 1); if (lane == 0) p = 0;
						u32 c0 = e0 - p, c1 = e1 - e0;
						c0 -= c0 >> 1; c1 -= c1 >> 1;
						const u32 inc = dec_wave_scan(c0 + c1);
						cur = (inc - c1) | (inc << 16);
					}
					if (live) { if (CPL == 1) tab16[(u64)(ri ^ swz) * N + lane] = (u16)cur; else table[(u64)(ri ^ swz) * (N / 2) + lane] = cur; }
				}
				else if (live && lane >= il) { if (CPL == 1) tab16[(u64)(ri ^ swz) * N + lane] = (u16)cur; else table[(u64)(ri ^ swz) * (N / 2) + lane] = cur; }
			}
			// ---- the symbol: lane (j mod 64) keeps it until 64 are together or the record ends ----------------------------
			if (CPL == 1)
			{	// the index; 64 of them are translated, tested for "base lives in the quality stream" and stored at once (below)
				if (lane == (j & 63u)) mine = idx;
			}
			else
			{
				u32 qv = idx;
				if (translate) qv = qrc_elem<CPL>(tr_v, idx);
				if (lane == (j & 63u)) mine = qv;
				ncount += q_special(qv, lossy) ? 1u : 0u;
			}
			// a row that is visited twice in a row was requested before it was written
			if (ri_next == ri) nxt = cur;
			cur = nxt; ri = ri_next;
			hash = hpre | idx; sym_buf = nb; pctx = pn; rem = rem2;
			QRC_PREP();
			} while (++j < stop);
			// ---- the segment's end ------------------------------------------------------------------------------------------
			if ((j & 63u) == 0 || j == ql)
			{
				const u32 base = (j - 1) & ~63u;
				const bool in = lane < j - base;
				if (CPL == 1)
				{
					const u32 qv = (u32)__shfl((int)tr_v, (int)(mine & 63u));
					if (in) q[base + lane] = (u8)qv;
					ncount += (u32)__popcll(__ballot(in && q_special(qv, lossy)));
				}
				else if (in) q[base + lane] = (u8)mine;
			}
			if (j == ql)
			{
				if (lane == 0) { rp.kept[g0 + k] = (u16)(ql - ncount); rp.d_off[g0 + k] = d_total; }
				d_total += ql - ncount;
				for (++k; k < n_recs; ++k)
				{
					ql = rp.len[g0 + k];
					if (ql) break;
					if (lane == 0) { rp.kept[g0 + k] = 0; rp.d_off[g0 + k] = d_total; }
				}
				if (k == n_recs) break;
				q = text + rp.qual_off[g0 + k];
				j = 0; ncount = 0;
				// the row prepared at the end of the body was for position 1 of a record of the OLD length; the first symbol's own row
				// (position 0) is the one QRC_LAST() made, so only what follows it is prepared again, with the new length
				pctx = 0; rem = 0;
				QRC_STEP();
				QRC_PREP();
			}
			if (err) break;
		}
#undef QRC_PREP
#undef QRC_LAST
#undef QRC_POS
#undef QRC_STEP
		if (min_r == 0 || max_idx >= cnt) err |= DEC_ERR_FORMAT;
	}
	s.bit = uw_pos(win) * 8;
	if (s.bit > (u64)s.size * 8) s.err |= DEC_ERR_TRUNC;
	s.err |= err;
	if (lane == 0) S->d_total = d_total;
}

// ---- quality, the lean loop (round 5) -------------------------------------------------------------------------------------------
// The loop above is what the compiler makes of the straightforward formulation: ~110 instructions per symbol, two thirds of them
// scalar control flow and 64-bit scalar arithmetic, and a wave issues one instruction per ~4.6 cycles -- the stage saturates at
// ~12 GB/s however many blocks are in flight.  This one is written for instruction count, N <= 64 (one counter per lane):
//   * 32-bit coder value: on every stream an encoder writes, buffer - low < range <= 2^32 (the first four bytes of the stream are
//     zero); anything else is refused (DEC_ERR_FORMAT) instead of being decoded the reference's undefined way.  Products are
//     32 bits (count * r <= total * r <= range), the search is one v_mul_lo + one compare;
//   * a symbol whose cumulative counts never exceed the value (no encoder writes that) ends the block with DEC_ERR_FORMAT;
//   * the coder's bytes come from a 64-bit window in scalar registers (FWin): a renormalisation of one or two bytes is a handful
//     of shifts, decided from clz(range) -- unless the carry clamp of src/RangeCoder.h:122-129 can fire (bits 24..39 of low all
//     set), which takes the reference's loop byte by byte;
//   * every lane holds a counter of the row (lane l the one numbered l mod N) and keeps it up to date, so the row is stored by all
//     lanes without touching the execution mask, at the byte offset it was loaded from (one VGPR, SGPR base);
//   * the position contexts of 64 consecutive positions are computed by the 64 lanes at once, once per 64 symbols; a symbol reads
//     its own with one v_readlane instead of stepping quotient and remainder in scalar registers.
// Same segments as above (to the next multiple of 64, to the symbol before the record's last, the last one).
// "this wave-uniform value lives in a scalar register from here on": the compiler knows which values are uniform, but it keeps them
// in vector registers once a vector-only instruction (a byte swap, an alignbit it has matched, a conversion) has touched them, and
// everything computed from them follows.  An empty asm with an SGPR constraint makes it move the value back (v_readfirstlane) there.
#define DEC_SGPR(x) ((x) = (u32)__builtin_amdgcn_readfirstlane((int)(x)))
// ... and the same for a value that IS in a scalar register, to keep the compiler from matching a vector-only pattern across it
#ifdef DSRC_EMU_BUILD
#define DEC_OPAQUE_S(x) ((void)0)
#else
#define DEC_OPAQUE_S(x) asm volatile("" : "+s"(x))
#endif
struct FWin { const CONST_AS u32* p4; u64 wq; u32 wbits, q0, q1, nextw, lastw, origin; };
// the byte swap is a vector instruction (v_perm_b32); its result is wave-uniform and goes back to a scalar register, or everything
// downstream of the window -- the coder's value, the search -- ends up in vector registers with it
__device__ __forceinline__ u32 fw_bswap(u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)__builtin_bswap32(x)); }

__device__ __forceinline__ void fw_refill(FWin& w)
{	// 32 more bits behind the wbits (< 32... <= 32) that are left
	w.wq |= (u64)fw_bswap(w.q0) << (32u - w.wbits);
	w.wbits += 32u;
	w.q0 = w.q1;
	w.q1 = w.p4[w.nextw < w.lastw ? w.nextw : w.lastw];
	++w.nextw;
}
__device__ __forceinline__ void fw_start(FWin& w, const BitSrc& s)
{
	const u64 a = (u64)s.p, x = a + (s.bit >> 3);
	const u64 w0 = x & ~3ull;
	const u32 mis = (u32)(x & 3ull);
	w.p4 = (const CONST_AS u32*)w0;
	w.lastw = (u32)((((a + s.size - 1) & ~3ull) - w0) >> 2);
	w.origin = (u32)(s.bit >> 3) - mis;                             // block position of byte 0 of dword 0
	w.wq = (u64)(fw_bswap(w.p4[0]) << (8 * mis)) << 32; w.wbits = 32u - 8u * mis;
	w.q0 = w.p4[1 < w.lastw ? 1 : w.lastw]; w.q1 = w.p4[2 < w.lastw ? 2 : w.lastw];
	w.nextw = 3;
	fw_refill(w);
}
// the next nbits (8 or 16) of the stream; at least 32 are in the window before and after
__device__ __forceinline__ u32 fw_take(FWin& w, u32 nbits)
{
	const u32 v = (u32)(w.wq >> (64u - nbits));
	w.wq <<= nbits; w.wbits -= nbits;
	if (__builtin_expect(w.wbits < 32u, 0)) fw_refill(w);
	return v;
}
// bytes of the block consumed so far, as a position inside the block: (nextw - 2) dwords have been folded into the window
__device__ __forceinline__ u64 fw_pos(const FWin& w) { return (u64)w.origin + (u64)(w.nextw - 2) * 4 - w.wbits / 8; }

template <u32 N>
__device__ __forceinline__ void qrc_decode_lean(BitSrc& s, u32* table, u32 ord, u32 rescale, u32 cnt, bool translate, const u8* sym_tab, u32 lossy,
												const DecDesc& d, DecState* S, RecPools rp, u8* text)
{
	static_assert(N <= 64, "one counter per lane");
	constexpr u32 LIM = (1u << 16) - 2 * N;
	const u32 lane = lane_id();
	const u32 lanemod = lane & (N - 1);
	const u32 abits = dec_int_log2(N);
	const u32 rs_shift = dec_int_log2(rescale);                         // rescale is 8 or N: a power of two
	const u32 bits_lo = (ord / 2) * abits, bits_hi = (ord / 2 + 1) * abits;
	const u32 hash_mask = (1u << (ord * abits)) - 1u;                   // ord * abits <= 21 for every scheme
	const u32 swap_mask = (((1u << bits_lo) - 1u) | ~((1u << bits_hi) - 1u)) & hash_mask;
	const u32 row_shift = abits + 1;                                    // a row is N counters of two bytes
	const u32 lane2 = lanemod * 2u;
	u8* tabb = (u8*)table;
	const u32 tr_v = translate ? (u32)sym_tab[lanemod] : lanemod;       // symbol value of counter i, in lane i

	const u64 g0 = d.rec_base;
	const u32 n_recs = S->n_recs;
	u32 k = 0, d_total = 0, err = 0;
	u32 ql = 0;
	for (; k < n_recs; ++k)
	{	// records without a quality line code nothing
		ql = rp.len[g0 + k];
		if (ql) break;
		if (lane == 0) { rp.kept[g0 + k] = 0; rp.d_off[g0 + k] = d_total; }
	}
	// RangeDecoder::Start (src/RangeCoder.h:97-106): eight bytes into the buffer, of which the first four are zero
	FWin win; fw_start(win, s);
	u32 buf, range = 0xFFFFFFFFu; u64 low = 0;
	{
		const u32 b0 = fw_take(win, 16), b1 = fw_take(win, 16), b2 = fw_take(win, 16), b3 = fw_take(win, 16);
		if (k < n_recs && (b0 | b1)) err |= DEC_ERR_FORMAT;
		buf = (b2 << 16) | b3;
	}
	if (k < n_recs && !err)
	{
		u8* q = text + rp.qual_off[g0 + k];
		u32 hpre = 0, sym_buf = 0, j = 0, ncount = 0, mine = 0, max_idx = 0;
		u64 all_m = ~0ull;
		u32 ri = 0, off_cur = lane2;                                  // row of the symbol being decoded and this lane's place in it
		u32 cur = lanemod + 1;                                        // row 0 of a fresh table is 1, 2, .., N
		double nf = dec_div_prep(range);
		dec_vm_drain();
		// position contexts floor(p * rescale / ql) of the positions p = g + 2 + lane, g = the group of 64 the symbol being decoded is
		// in: the symbol at position j prepares the row of position j + 2 (its own row and the next one's base are known by then)
		u32 v_pn = 0, pn = 0, base_next = 0;
#define QL_GROUP(g_) do { v_pn = dec_div(dec_div_prep(((g_) + 2u + lane) << rs_shift), ql); } while (0)
#define QL_PN(j_) ((u32)__builtin_amdgcn_readlane((int)v_pn, (int)((j_) & 63u)))
		// the row base of the symbol after the next one, up to that symbol's predecessor: hash slots from hpre | idx and sym_buf
#define QL_PREP(idx_, pn_) do { \
			const u32 h2_ = (hpre | (idx_)) << abits; \
			const u32 nb_ = (h2_ >> bits_lo) & (N - 1); \
			hpre = (h2_ & swap_mask) | (((nb_ + sym_buf) >> 1) << bits_lo); \
			sym_buf = nb_; \
			pn = (pn_); \
			base_next = (hpre << rs_shift) + pn; } while (0)
		// before the first symbol: its own row is row 0 (hash 0, position 0); the second symbol's base has an empty hash and position 1
		QL_GROUP(0u - 2u + 0u);                                         // lane l: position l (group "-2": positions 0 .. 63)
		pn = (u32)__builtin_amdgcn_readfirstlane(__builtin_amdgcn_readlane((int)v_pn, 1));
		if (ql == 1) pn = 0;                                          // the second symbol is the next record's first
		base_next = pn;
		QL_GROUP(0u);
		for (;;)
		{
			u32 stop = (j | 63u) + 1u;
			if (j + 1 < ql) { if (stop > ql - 1) stop = ql - 1; }
			else { stop = ql; base_next -= pn; pn = 0; }              // the row after the LAST symbol is the next record's first: position 0
			DEC_SGPR(buf); DEC_SGPR(base_next); DEC_SGPR(pn); DEC_SGPR(ri); DEC_SGPR(hpre); DEC_SGPR(sym_buf);
			// one symbol; the segment takes two per turn of its loop, so that what one symbol hands to the next (row, offset, context)
			// changes registers by renaming instead of by copies at the loop's back edge (the compiler ignores `#pragma unroll` here)
			auto symbol = [&](const u32 j) __attribute__((always_inline))
			{
				// ---- the row has arrived: symbol index -----------------------------------------------------------------
				const u32 total = (u32)__builtin_amdgcn_readlane((int)cur, N - 1);
				const u32 r = dec_div(nf, total);
				const u32 r_s = (u32)__builtin_amdgcn_readfirstlane((int)r);
				const u64 m = __ballot(cur * r > buf);
				// value >= total * r (no lane of the row says yes): no encoder writes that.  The loop has no exit in its middle (the
				// compiler keeps wave-uniform state that lives across such an exit in vector registers): the symbol becomes the row's
				// last one -- every access stays inside the row and the table -- and the lanes' answers are AND-ed up; bit N - 1 of
				// the result is tested once per segment.
				all_m &= m;
				const u32 idx = (u32)__ffsll((long long)(m | (1ull << (N - 1)))) - 1u;
				// ---- request the next row ------------------------------------------------------------------------------
				const u32 ri_next = base_next + (idx << rs_shift);
				const u32 off_next = (ri_next << row_shift) + lane2;
				u32 nxt = (u32)*(const u16*)(tabb + off_next);
				// ---- in its shadow: coder state ------------------------------------------------------------------------
				const u32 hi = (u32)__builtin_amdgcn_readlane((int)cur, (int)idx);
				const u32 lo = (u32)__builtin_amdgcn_readlane((int)qrc_up1(cur), (int)idx);
				const u32 rr = lo * r_s;
				buf -= rr; low += rr;
				range = r_s * (hi - lo);
				// RangeDecoder::DecodeFrequency's loop as written (src/RangeCoder.h:122-135): one or two bytes (range >= 2^8: r >= 2^8,
				// freq >= 1), one symbol in three.  (A two-byte step decided from clz(range) with the loop as its fallback made the
				// compiler shuffle a dozen registers where the two roads meet: more instructions than the loop itself.)
				while (range <= 0x00FFFFFFu)
				{
					if (__builtin_expect(((low ^ (low + range)) & 0xFF00000000000000ull) != 0, 0))
					{	// the carry clamp; a range it leaves empty is no encoder's
						const u32 l32 = (u32)low;
						range = (l32 | 0x00FFFFFFu) - l32;
						if (range == 0) { all_m = 0; range = 0x00FFFFFFu; }
					}
					u32 byte = fw_take(win, 8);
					DEC_OPAQUE_S(byte);                                   // (or the compiler fuses shift and or into a v_alignbit, and the value follows it)
					buf = (buf << 8) | byte;
					low <<= 8; range <<= 8;
				}
				nf = dec_div_prep(range);
				// ---- the row: +2 on the symbol = +2 on every cumulative count from it on; Rescale() now instead of at the next visit
				u32 upd = cur + (lanemod >= idx ? 2u : 0u);
				if (__builtin_expect(total >= LIM - 2, 0))
				{
					u32 p = __shfl_up(upd, 1); if (lanemod == 0) p = 0;
					u32 c = upd - p;                                    // (lanes past the row repeat it: their scan restarts at their own counter 0)
					c -= c >> 1;
					u32 inc = dec_wave_scan(c);
					if (N < 64) { const u32 tot2 = (u32)__builtin_amdgcn_readlane((int)inc, N - 1); inc -= tot2 * (lane / N); }
					upd = inc;
				}
				*(u16*)(tabb + off_cur) = (u16)upd;
				// ---- the symbol: lane (j mod 64) keeps it until 64 are together or the record ends ----------------------
				mine = qrc_writelane(idx, j, mine);
				max_idx = max_idx > idx ? max_idx : idx;                // a symbol the block's alphabet does not have: tested once at the end
				// a row that is visited twice in a row was requested before it was written
				if (ri_next == ri) nxt = upd;
				cur = nxt; ri = ri_next; off_cur = off_next;
				QL_PREP(idx, QL_PN(j));
			};
			for (; j + 2 <= stop; j += 2) { symbol(j); symbol(j + 1); }
			if (j < stop) { symbol(j); ++j; }
			if (!((all_m >> (N - 1)) & 1ull)) err |= DEC_ERR_FORMAT;
			if (err) break;
			// ---- the segment's end ------------------------------------------------------------------------------------------
			if ((j & 63u) == 0 || j == ql)
			{
				const u32 base = (j - 1) & ~63u;
				const bool in = lane < j - base;
				const u32 qv = (u32)__shfl((int)tr_v, (int)(mine & 63u));
				if (in) q[base + lane] = (u8)qv;
				ncount += (u32)__popcll(__ballot(in && q_special(qv, lossy)));
				if (j < ql) QL_GROUP(j);
			}
			if (j == ql)
			{
				if (lane == 0) { rp.kept[g0 + k] = (u16)(ql - ncount); rp.d_off[g0 + k] = d_total; }
				d_total += ql - ncount;
				for (++k; k < n_recs; ++k)
				{
					ql = rp.len[g0 + k];
					if (ql) break;
					if (lane == 0) { rp.kept[g0 + k] = 0; rp.d_off[g0 + k] = d_total; }
				}
				if (k == n_recs) break;
				q = text + rp.qual_off[g0 + k];
				j = 0; ncount = 0;
				// the base prepared by the record's last symbol was for a position behind the record's end; the first symbol's own row
				// (position 0) is the one made above, so only what follows it is prepared again, with the new length: position 1
				QL_GROUP(0u - 2u + 0u);
				{
					u32 p1 = (u32)__builtin_amdgcn_readfirstlane(__builtin_amdgcn_readlane((int)v_pn, 1));
					if (ql == 1) p1 = 0;
					base_next += p1 - pn; pn = p1;
				}
				QL_GROUP(0u);
			}
		}
#undef QL_GROUP
#undef QL_PN
#undef QL_PREP
		if (max_idx >= cnt) err |= DEC_ERR_FORMAT;
	}
	s.bit = fw_pos(win) * 8;
	if (s.bit > (u64)s.size * 8) s.err |= DEC_ERR_TRUNC;
	s.err |= err;
	if (lane == 0) S->d_total = d_total;
}

// the scheme byte and the alphabet of an order-context quality stream: IQualityModelerProxy::Decode (src/QualityModelerProxy.h:59-69;
// the lossy order proxy has no scheme byte, :156-159), TTranslationalQualityEncoder::Read (src/QualityEncoder.h:344-357)
struct QrcScheme { u32 n, ord, rescale, translate; };
__device__ __forceinline__ bool qrc_scheme(u32 quality_order, u32 lossy, u32 scheme_byte, QrcScheme* q)
{
	if (lossy) { q->n = 8; q->ord = quality_order; q->rescale = 8; q->translate = 0; return true; }
	if (scheme_byte > 7) return false;
	const u32 sc = scheme_byte & 3u;
	q->n = 16u << sc;
	q->ord = quality_order == 1 ? (sc == 0 ? 3u : sc == 1 ? 2u : 1u) : (4u - sc);
	q->rescale = scheme_byte < 4 ? 8u : q->n;
	q->translate = 1;
	return true;
}

// wave per block of the round; tabs[blockIdx.x] names the block and its table.  One kernel per alphabet size (the host launches the
// sizes a round contains; a wave whose block has another size leaves at once): the five decoders in one kernel cost the common
// ones registers they do not need.
template <u32 NSEL>
__global__ void __launch_bounds__(64) k_dec_qrc(const u8* in, const DecDesc* desc, DecState* st, const DecTab* tabs, RecPools rp, u8* out, u32* tables, DecParams prm)
{
	__shared__ u8 s_sym[256];
	__shared__ u32 s_par[4];
	const DecTab tb = tabs[blockIdx.x];
	const u32 b = tb.block;
	DecState* S = &st[b];
	if (S->err) return;                                   // wave-uniform
	{
		QrcScheme q0;
		if (qrc_scheme(prm.quality_order, prm.lossy, S->q_scheme, &q0) && q0.n != NSEL) return;      // another launch's
	}
	const DecDesc d = desc[b];
	BitSrc s; s.p = in + d.in_off; s.size = d.in_size; s.err = 0; s.bit = (u64)S->qua_pos * 8;
	if (threadIdx.x == 0)
	{
		for (u32 i = 0; i < 256; ++i) s_sym[i] = 255;
		if (!prm.lossy)
		{
			(void)bs_byte(s);                             // scheme byte: k_dec_tags has left it in q_scheme
			bs_align(s);
			u32 cnt = 0;
			for (u32 i = 0; i < 256; ++i) if (bs_bit(s)) s_sym[cnt++] = (u8)i;
			bs_align(s);
		}
		s_par[0] = (u32)s.bit; s_par[1] = (u32)(s.bit >> 32); s_par[2] = s.err;
	}
	__syncthreads();
	s.bit = ((u64)s_par[1] << 32) | s_par[0]; s.err = s_par[2];
	QrcScheme qs = {0u, 0u, 0u, 0u};
	if (!qrc_scheme(prm.quality_order, prm.lossy, S->q_scheme, &qs)) s.err |= DEC_ERR_FORMAT;
	const u32 cnt = prm.lossy ? 8u : S->q_cnt;
	if (cnt == 0 || cnt > qs.n) s.err |= DEC_ERR_FORMAT;
	if (!s.err)
	{
		const u32 ab = dec_int_log2(qs.n);
		const u64 words = ((u64)1 << (ab * qs.ord)) * qs.rescale * qs.n / 2;
		if (words > tb.words) s.err |= DEC_ERR_POOL;
	}
	const u32 swz_seed = (b + 1u) * 0x9E3779B1u >> 7;
	if (!s.err)
	{
		u32* table = tables + tb.off;
		u8* text = out + d.out_off;
#if DEC_QRC_LEAN
		if constexpr (NSEL <= 64) qrc_decode_lean<NSEL>(s, table, qs.ord, qs.rescale, cnt, qs.translate != 0, s_sym, prm.lossy, d, S, rp, text);
		else
#endif
		qrc_decode<NSEL>(s, table, qs.ord, qs.rescale, cnt, swz_seed, qs.translate != 0, s_sym, prm.lossy, d, S, rp, text);
	}
	if (threadIdx.x == 0)
	{
		if (!s.err) S->dna_pos = bs_pos(s);
		if (s.err) atomicOr(&S->err, s.err);                  // (a verifying pass runs k_dec_dnarc on the same DecState at the same time)
	}
}

// ---- verification of blocks this library has just written: where the DNA stream starts and how many symbols it holds is known
// from the compressing pass, so its chain need not wait for the quality chain to find out (run_decode launches both at once).
struct DecHint { u32 dna_pos, d_total, d_scheme, pad; };
__global__ void __launch_bounds__(64) k_dec_hint(DecState* st, const DecHint* hint, u32 n_blocks)
{
	const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_blocks) return;
	st[b].dna_pos = hint[b].dna_pos; st[b].d_total = hint[b].d_total; st[b].d_scheme = hint[b].d_scheme;
}

// ---- the scheme byte of the DNA stream (IDnaModelerProxy::Decode, src/DnaModelerProxy.h:61-71): thread per block --------------
// With `hint` (a verifying pass whose DNA chains started from the compressing pass's figures): what the quality stage has parsed out
// of the block's BYTES since -- where the DNA stream starts, how many bases it holds -- and the scheme byte found there must be what
// the DNA stage was told, or the block is not the one that was meant to be written.
__global__ void __launch_bounds__(64) k_dec_dhead(const u8* in, const DecDesc* desc, DecState* st, DecParams prm, const DecHint* hint)
{
	const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= prm.n_blocks) return;
	DecState* S = &st[b];
	if (S->err) return;
	const DecDesc d = desc[b];
	if (S->dna_pos >= d.in_size) { S->err |= DEC_ERR_TRUNC; return; }
	const u32 sch = in[d.in_off + S->dna_pos];
	S->d_scheme = sch;
	if (sch != 255 && sch > 1) S->err |= DEC_ERR_FORMAT;
	if (hint && (hint[b].dna_pos != S->dna_pos || hint[b].d_total != S->d_total || hint[b].d_scheme != sch)) S->err |= DEC_ERR_FORMAT;
}

// ---- the coder's bytes for one lane's stream ------------------------------------------------------------------------------------
// A wave's vector loads return in order, so a window load that is waited for in the middle of a symbol would also wait for the
// candidate rows requested just before it -- and with 64 unsynchronised streams some lane refills its window at almost every
// symbol.  So the refill only moves registers: every symbol requests the 8 bytes behind the window together with the candidate
// rows (`wl`, mostly the same cached address), and the refill takes the copy that arrived one symbol earlier (`wp`).
struct LWin { const u8* p; u32 size; u64 w0, w1, wp; u32 left, nx, wp_pos; };

// 8 bytes of the block at byte position pos, the first one in the top bits; bytes behind the end read as zero, nothing
// outside the block is touched (size >= 16)
__device__ __forceinline__ u64 lw_load(const u8* p, u32 size, u32 pos)
{
	const u32 at = pos < size ? pos : size;
	const u32 over = at + 8 > size ? at + 8 - size : 0u;
	const u64 v = __builtin_bswap64(*(const dec_u64_unaligned*)(p + (at - over)));
	return (v << (4 * over)) << (4 * over);                    // over <= 8; no select, so that the load stays unconditional
}
__device__ __forceinline__ u64 lw_start(LWin& w, const BitSrc& s)          // returns the coder's first 8 bytes
{
	const u32 at = (u32)(s.bit >> 3);
	w.p = s.p; w.size = s.size;
	w.w0 = lw_load(w.p, w.size, at + 8); w.w1 = lw_load(w.p, w.size, at + 16); w.nx = at + 16; w.left = 8;
	w.wp_pos = w.nx + 8; w.wp = lw_load(w.p, w.size, w.wp_pos);
	return lw_load(w.p, w.size, at);
}
__device__ __forceinline__ u32 lw_byte(LWin& w)
{
	if (w.left == 0)
	{
		w.w0 = w.w1;
		if (w.wp_pos != w.nx + 8) { w.wp_pos = w.nx + 8; w.wp = lw_load(w.p, w.size, w.wp_pos); }      // two refills within two symbols
		w.w1 = w.wp; w.nx += 8; w.left = 8;
	}
	const u32 b = (u32)(w.w0 >> 56);
	w.w0 <<= 8; --w.left;
	return b;
}
__device__ __forceinline__ u32 lw_pos(const LWin& w) { return w.nx - w.left; }

#ifndef DNA_WL_EVERY
#define DNA_WL_EVERY 2u
#endif
// ---- DNA, order-k range coder: one LANE per block ----------------------------------------------------------------------------
template <u32 N>
__global__ void __launch_bounds__(64) k_dec_dnarc(const u8* in, const DecDesc* desc, DecState* st, const DecTab* tabs, u32 n_tabs,
												   u8* d_stream, u32* tables, DecParams prm)
{
	constexpr u32 W = N / 2;                                  // dwords per row
	constexpr u32 LIM = (1u << 16) - 2 * N;
	constexpr u32 abits = N == 8 ? 3u : 2u;
	// a few dozen waves with one long dependent chain each: when another pass's quality stage fills the SIMDs, they go first
	__builtin_amdgcn_s_setprio(3);
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_tabs) return;
	const DecTab tb = tabs[i];
	const u32 b = tb.block;
	DecState* S = &st[b];
	if (S->err) return;
	const DecDesc d = desc[b];
	const u32 ord = N == 8 ? (prm.dna_order < 7u ? prm.dna_order : 7u) : prm.dna_order;
	const u32 mask = (1u << (abits * ord)) - 1u;              // <= 21 bits
	BitSrc s; s.p = in + d.in_off; s.size = d.in_size; s.err = 0; s.bit = ((u64)S->dna_pos + 1) * 8;      // behind the scheme byte
	if ((((u64)mask + 1) * W) > tb.words) { atomicOr(&S->err, (u32)DEC_ERR_POOL); return; }
	u32* tab = tables + tb.off;
	u8* dst = d_stream + d.d_base;
	const u32 total = S->d_total;
	// RangeDecoder::Start (src/RangeCoder.h:97-106): eight bytes; buffer < range holds on every stream an encoder writes, so the
	// first four are zero and the buffer is 32 bits wide from here on (anything else is refused).  The step is a chain of ~150
	// dependent instructions that 64 streams share: it is written for instruction count (32-bit products: count * r <= total * r
	// <= range; one shift per renormalisation unless the carry clamp can fire or the window runs dry).
	LWin win;
	u32 buffer, range = 0xFFFFFFFFu, err = 0; u64 low = 0;
	{
		const u64 first = lw_start(win, s);
		if (total && (first >> 32)) err |= DEC_ERR_FORMAT;
		buffer = (u32)first;
	}
	double nf = dec_div_prep(range);
	u32 hash = 0;
	u32 cur[W];                                               // row 0 of a fresh table
#pragma unroll
	for (u32 w = 0; w < W; ++w) cur[w] = 0x00010001u;
	u64 pack = 0;
	dec_vm_drain();                                           // nothing pending at the loop's entry: the waits inside it then count only its own requests
	// One symbol.  touch_new / touch_old: see below; two variables taken in turns by the two calls of the loop body, because a
	// copy from one to the other would have to wait for the data.
	auto step = [&](const u32 t, u32& touch_new, const u32 touch_old) __attribute__((always_inline))
	{
		// the N candidate rows of the next symbol are consecutive and known before this symbol is
		const u32 nbase = (hash << abits) & mask;
		const uint2* cp = (const uint2*)(tab + (u64)nbase * W);
		uint2 cand[N * W / 2];
#pragma unroll
		for (u32 w = 0; w < N * W / 2; ++w) cand[w] = cp[w];
		// the 8 bytes behind the window: requested every DNA_WL_EVERY-th symbol (t is the same in all lanes: a uniform branch).  A
		// symbol consumes 0-3 bytes, typically a quarter of one, so the copy is there long before the window has moved 8 bytes on;
		// when it is not (wp_pos stale), the refill below does not happen and the reference's loop fetches its bytes itself.  Every
		// symbol asking cost 0.4 s of a 2400-block pass (64 lanes = 64 lines and pages per request, in front of the look-ahead
		// touch); measured at -d3 -q0, 64 / 2400 blocks: every symbol 1.69 / 3.47 s, every 2nd 1.66 / 3.11, 4th 1.88 / 3.24, 8th 1.80 / 3.18.
		const bool wl_now = (t & (DNA_WL_EVERY - 1u)) == 0;
		u32 wl_pos = 0; u64 wl = 0;
		if (wl_now) { wl_pos = win.nx + 8; wl = lw_load(win.p, win.size, wl_pos); }
		// ... and the N * N candidate rows of the symbol after that are one 128-byte line (N = 4): touching it now, one symbol
		// before its rows are requested, takes a symbol's time off that request's latency.  The value is not used; it is
		// "consumed" at the end of the NEXT symbol, when the wave's in-order returns have long delivered it.  (An LDS-DMA
		// request instead, which needs no register, makes the compiler wait with vmcnt(0) everywhere; hidden from it in inline
		// assembly it makes every wait that counts younger requests one too strict.)
		if (N == 4) touch_new = tab[(u64)((nbase << abits) & mask) * W];
		u32 c[N], a[N];
#pragma unroll
		for (u32 k = 0; k < N; ++k) c[k] = (cur[k / 2] >> (16 * (k & 1))) & 0xFFFFu;
		a[0] = c[0];
#pragma unroll
		for (u32 k = 1; k < N; ++k) a[k] = a[k - 1] + c[k];
		const u32 T = a[N - 1];
		const u32 r = dec_div(nf, T);
		if (r == 0 || buffer >= T * r) err |= DEC_ERR_FORMAT;          // range < total, or the reference walks off the row
		u32 idx = 0, rr = 0, f = c[0];
#pragma unroll
		for (u32 k = 1; k < N; ++k)
		{
			const u32 sk = a[k - 1] * r;
			if (buffer >= sk) { idx = k; rr = sk; f = c[k]; }
		}
		buffer -= rr; low += rr;
		range = r * f;
		if (range == 0) { err |= DEC_ERR_FORMAT; range = 0xFFFFFFFFu; }
		{
			const u32 nb8 = range <= 0x00FFFFFFu ? (u32)__clz((int)range) >> 3 : 0u;      // bytes to shift in: 0 .. 3
			if ((((u32)(low >> 24)) & 0xFFFFu) != 0xFFFFu && win.left >= nb8)
			{
				const u32 sh = nb8 * 8;
				range <<= sh; low <<= sh;
				buffer = (buffer << sh) | (u32)(((u64)(u32)(win.w0 >> 32) << sh) >> 32);
				win.w0 <<= sh; win.left -= nb8;
			}
			else
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
			if (win.left == 0 && win.wp_pos == win.nx + 8) { win.w0 = win.w1; win.w1 = win.wp; win.nx += 8; win.left = 8; }
		}
		nf = dec_div_prep(range);
		// the row: +2, Rescale() now instead of at the next visit
#pragma unroll
		for (u32 w = 0; w < W; ++w) cur[w] += (idx >> 1) == w ? (2u << (16 * (idx & 1u))) : 0u;
		if (T + 2 >= LIM)
		{
#pragma unroll
			for (u32 w = 0; w < W; ++w)
			{
				const u32 x0 = cur[w] & 0xFFFFu, x1 = cur[w] >> 16;
				cur[w] = (x0 - (x0 >> 1)) | ((x1 - (x1 >> 1)) << 16);
			}
		}
		{
			u32* row = tab + (u64)hash * W;
			if (W == 2) *(uint2*)row = make_uint2(cur[0], cur[1]);
			else *(uint4*)row = make_uint4(cur[0], cur[1], cur[W - 2], cur[W - 1]);
		}
		pack |= (u64)idx << (8 * (t & 7u));
		if ((t & 7u) == 7u) { *(u64*)(dst + (t & ~7u)) = pack; pack = 0; }
		const u32 nh = nbase | idx;
		if (nh != hash)
		{	// candidate row idx (a row visited twice in a row is the one in registers)
			constexpr u32 R2 = W / 2 > 0 ? W / 2 : 1;               // uint2 per row: 1 (N = 4), 2 (N = 8)
			uint2 sel[R2];
#pragma unroll
			for (u32 w = 0; w < R2; ++w) sel[w] = cand[w];
#pragma unroll
			for (u32 k = 1; k < N; ++k)
				if (idx == k)
				{
#pragma unroll
					for (u32 w = 0; w < R2; ++w) sel[w] = cand[k * R2 + w];
				}
#pragma unroll
			for (u32 w = 0; w < R2; ++w) { cur[2 * w] = sel[w].x; cur[2 * w + 1] = sel[w].y; }
		}
		hash = nh;
		if (wl_now) { win.wp = wl; win.wp_pos = wl_pos; }
#ifndef DSRC_EMU_BUILD
		if (N == 4) asm volatile("" :: "v"(touch_old));
#else
		(void)touch_old;
#endif
	};
	{
		u32 ta = 0, tb = 0, t = 0;
		for (; t + 1 < total && !err; t += 2) { step(t, ta, tb); step(t + 1, tb, ta); }
		if (t < total && !err) step(t, ta, tb);
	}
	if (total & 7u)
	{	// d_base is 64-byte aligned and the stream's allocation is padded: the last, partial group is stored whole
		*(u64*)(dst + (total & ~7u)) = pack;
	}
	const u32 end = lw_pos(win);
	if (end > s.size) err |= DEC_ERR_TRUNC;
	S->end_pos = end;
	if (err) atomicOr(&S->err, err);                          // (a verifying pass runs k_dec_qrc on the same DecState at the same time)
}

// ---- DNA of the -d0 level and blocks without a DNA stream: wave per block -----------------------------------------------------
// DnaModelerBasicB2::Decode (src/DnaModelerBasicB2.h:48-60): symbol t is bits 2t, 2t+1, unpacked by the whole wave;
// DnaModelerHuffman::Decode (src/DnaModelerHuffman.cpp:75-113): lane 0 walks the tree.
__global__ void __launch_bounds__(64) k_dec_dna0(const u8* in, const DecDesc* desc, DecState* st, u32* pool, u8* d_stream, DecParams prm)
{
	__shared__ u8 s_sym[32];
	const u32 b = blockIdx.x;
	DecState* S = &st[b];
	if (S->err) return;
	const u32 d_scheme = S->d_scheme;
	if (prm.dna_order > 0 && d_scheme != 255) return;          // k_dec_dnarc's
	const DecDesc d = desc[b];
	BitSrc s; s.p = in + d.in_off; s.size = d.in_size; s.err = 0; s.bit = ((u64)S->dna_pos + 1) * 8;
	const u32 total = S->d_total;
	u8* dst = d_stream + d.d_base;
	if (d_scheme == 0)
	{
		BitSrc t = s;
		const u64 bit0 = s.bit;
		for (u32 t0 = threadIdx.x * 16u; t0 < total; t0 += blockDim.x * 16u)
		{
			t.bit = bit0 + 2ull * t0;
			const u32 w = bs_peek32(t);
			const u32 cnt = total - t0 < 16u ? total - t0 : 16u;
			if (cnt == 16u)
			{	// sixteen bytes as two aligned 8-byte stores (d_base and t0 are multiples of 16)
				u64 lo = 0, hi = 0;
#pragma unroll
				for (u32 k = 0; k < 8; ++k) { lo |= (u64)((w >> (30 - 2 * k)) & 3u) << (8 * k); hi |= (u64)((w >> (14 - 2 * k)) & 3u) << (8 * k); }
				((u64*)(dst + t0))[0] = lo; ((u64*)(dst + t0))[1] = hi;
			}
			else for (u32 k = 0; k < cnt; ++k) dst[t0 + k] = (u8)((w >> (30 - 2 * k)) & 3u);
		}
	}
	if (threadIdx.x != 0) return;
	if (d_scheme == 0) { bs_skip(s, total); bs_skip(s, total); bs_align(s); }
	else if (d_scheme == 1)
	{
		NodePool np; np.w = pool + d.qnode_off; np.cap = d.qnode_cap; np.top = 0;
		u32 n = 0;
		for (u32 i = 0; i < 20; ++i) s_sym[i] = 255;
		for (u32 i = 0; i < 20; ++i) if (bs_bit(s)) s_sym[n++] = (u8)i;
		const u32 tr = huff_load(s, np);
		for (u32 t = 0; t < total && !s.err; ++t) { const u32 x = huff_sym(s, np.w + tr); dst[t] = x < 20 ? s_sym[x] : 255; }
		bs_align(s);
	}
	// d_scheme == 255: no DNA stream at all, the block ends behind the scheme byte
	S->end_pos = bs_pos(s);
	S->err |= s.err;
}
