// The -q0 position schemes on the fast path: QualityPositionModelerPlain / Truncated ::Decode
// (src/QualityPositionModeler.cpp:39-103,189-220,291-337) with HuffmanEncoder::LoadTree / Decode (src/huffman.cpp:225-262).
//
// k_dec_qhuff (k_dec.h) walks one tree node per code bit: ~6 dependent LDS reads per symbol, 46 KB of LDS per block (three blocks
// per CU).  A stream of variable-length codes is one chain per block whatever is done, so the rate of the pass is (blocks
// resident) / (time per symbol); this kernel works on both:
//   * per read position a 64-entry table indexed by the next 6 bits gives the quality character and the code length in ONE read (codes longer than 6
//     bits -- symbols rarer than 1/64 -- leave the table with the tree node they have reached and finish bit by bit);
//   * the trees are kept as two bytes per node (children < 128: alphabets of up to 128 symbols), 12 KB instead of 46;
//     with the tables (19 KB) a block takes 32 KB of LDS: five blocks per CU;
//   * the code bits come through the scalar cache (SWin, k_dec_tags.h), four characters leave per store.
// The RLE scheme (QualityRLEModeler::Decode, src/QualityRLEModeler.cpp:48-113,380-486 -- what binned qualities of current instruments
// get at -q0) takes the same route: per previous symbol a 64-entry table for the next quality symbol and one for the run length
// (the length byte itself in the entry), trees of longer codes stay in the global pool.
// Blocks it does not cover (reads longer than 152, alphabets over 128 symbols / 76 for RLE, more than 5888 tree nodes) are left
// to k_dec_qhuff: `q_done` in the block's state says which kernel has decoded the stream.
#pragma once
#include "k_dec_tags.h"

typedef u32 __attribute__((aligned(1))) dec_u32_unaligned;

#define Q0_MAXL 152u          // with Q0_NODES: 31.8 KB of LDS per block, five blocks per CU
#define Q0_NODES 5888u

// the RLE scheme with the wave's tables: called by every lane of the block's wave after the header has been parsed
__device__ __forceinline__ void q0_rle(BitSrc& s, NodePool& np, const u32* s_par, u16* s_fast, const u8* s_sym, const u8* s_ls,
									   const DecDesc& d, DecState* S, RecPools rp, u8* text, u32 lossy)
{
	const u32 dir = s_par[2], run_len = s_par[3], qn = s_par[4], ln = s_par[5];
	// entry [2 * 64 * prev + 64 * which + next 6 bits]: value | code length << 8 (value: which = 0 the next symbol's index, which = 1
	// the length byte), 0x8000 | node for a longer code, 0xFFFF for a code no encoder writes (an index outside the alphabet)
	for (u32 e = threadIdx.x; e < qn * 128; e += blockDim.x)
	{
		const u32 tree = __hip_atomic_load(np.w + dir + (e >> 6), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), which = (e >> 6) & 1u, bits = e & 63u;
		u32 node = 0, val = 0;
		for (u32 k = 1; k <= 6; ++k)
		{
			const u32 t = __hip_atomic_load(np.w + tree + node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const u32 child = ((bits >> (6 - k)) & 1u) ? t >> 16 : t & 0xFFFFu;
			if (child & 0x8000u)
			{
				const u32 x = child & 0x7FFFu;
				val = which == 0 ? (x < qn ? (x | (k << 8)) : 0xFFFFu) : (x < ln ? ((u32)s_ls[x] | (k << 8)) : 0xFFFFu);
				break;
			}
			node = child;
			if (k == 6) val = 0x8000u | node;
		}
		s_fast[e] = (u16)val;
	}
	__syncthreads();
	if (threadIdx.x != 0) return;
	s.bit = ((u64)s_par[8] << 32) | s_par[6];
	SWin w; sw_init(w, s);
	u32 err = 0;
	// one code: through the table, longer ones on through the tree in the pool (u32 nodes: left | right << 16, 0x8000 | leaf)
	auto code = [&](u32 prev, u32 which) -> u32
	{
		sw_refill(w);
		const u32 e = s_fast[prev * 128 + which * 64 + (u32)(w.w >> 58)];
		if (!(e & 0x8000u)) { const u32 len = (e >> 8) & 7u; w.w <<= len; w.n -= len; return e & 0xFFu; }
		if (e == 0xFFFFu) { err |= DEC_ERR_FORMAT; return 0; }
		const u32* T = np.w + np.w[dir + 2 * prev + which];
		u32 node = e & 0x7FFFu;
		w.w <<= 6; w.n -= 6;
		for (u32 guard = 0; guard < 600; ++guard)
		{
			if (w.n == 0) sw_refill(w);
			const u32 t = T[node];
			const u32 child = (w.w >> 63) ? t >> 16 : t & 0xFFFFu;
			w.w <<= 1; w.n -= 1;
			if (child & 0x8000u)
			{
				const u32 x = child & 0x7FFFu;
				if (x >= (which ? ln : qn)) { err |= DEC_ERR_FORMAT; return 0; }
				return which ? (u32)s_ls[x] : x;
			}
			node = child;
		}
		err |= DEC_ERR_FORMAT;
		return 0;
	};
	const u32 n_recs = S->n_recs;
	const u64 g0 = d.rec_base;
	u32 cur_len = 0, idx = 0, cur_q = 0, prev = 0, d_total = 0, special = 0;
	u32 nql = n_recs ? (u32)rp.len[g0] : 0u, nqo = n_recs ? rp.qual_off[g0] : 0u;       // one record ahead
	for (u32 k = 0; k < n_recs && !err; ++k)
	{
		const u64 g = g0 + k;
		const u32 ql = nql; u8* q = text + nqo;
		if (k + 1 < n_recs) { nql = rp.len[g + 1]; nqo = rp.qual_off[g + 1]; }
		u32 ncount = 0;
		for (u32 j = 0; j < ql && !err; )
		{
			if (cur_len == 0)
			{
				if (idx >= run_len) { err |= DEC_ERR_FORMAT; break; }
				prev = code(prev, 0);
				cur_q = s_sym[prev]; special = q_special(cur_q, lossy) ? 1u : 0u;
				cur_len = code(prev, 1) + 1;
				idx++;
			}
			const u32 m = cur_len < ql - j ? cur_len : ql - j;
			for (u32 t = 0; t < m; ++t) q[j + t] = (u8)cur_q;
			ncount += special * m; j += m; cur_len -= m;
		}
		rp.kept[g] = (u16)(ql - ncount); rp.d_off[g] = d_total; d_total += ql - ncount;
	}
	// runs the records did not consume are still read by the reference (DecodeRuns comes first)
	for (; idx < run_len && !err; ++idx) { prev = code(prev, 0); (void)code(prev, 1); }
	S->d_total = d_total;
	sw_finish(w, s);
	s.err |= err;
	bs_align(s);
	S->dna_pos = bs_pos(s);
	S->q_done = 1;
	S->err |= s.err;
}

__global__ void __launch_bounds__(64) k_dec_qpos(const u8* in, const DecDesc* desc, DecState* st, RecPools rp, u8* out, u32* pool, DecParams prm)
{
	__shared__ u16 s_fast[Q0_MAXL * 64];
	__shared__ u16 s_cn[Q0_NODES];
	__shared__ u16 s_cdir[Q0_MAXL];
	__shared__ u8 s_sym[256];
	__shared__ u8 s_ls[256];
	__shared__ u32 s_par[10];
	const u32 b = blockIdx.x;
	DecState* S = &st[b];
	if (S->err) return;                                   // wave-uniform
	const DecDesc d = desc[b];
	u8* text = out + d.out_off;
	BitSrc s; s.p = in + d.in_off; s.size = d.in_size; s.err = 0; s.bit = (u64)S->qua_pos * 8;
	NodePool np; np.w = pool + d.qnode_off; np.cap = d.qnode_cap; np.top = 0;
	const u32 lossy = prm.lossy;
	if (threadIdx.x == 0)
	{
		u32 dir = 0, maxl = 0, n = 0, first = 0, ok = 0;
		const u32 q_scheme = bs_byte(s);                      // validated by the tag kernel
		if (q_scheme <= 1)
		{	// IQualityPositionModeler::Decode: statistics, symbols, one tree per position (src/QualityPositionModeler.cpp:39-103)
			bs_align(s);
			maxl = bs_word(s);
			for (u32 i = 0; i < 256; ++i) if (bs_bit(s)) s_sym[n++] = (u8)i;
			if (maxl >= 1 && maxl <= Q0_MAXL && n <= 128 && !s.err)
			{
				dir = pool_take(np, maxl, &s.err);
				first = np.top;
				ok = 1;
				for (u32 i = 0; i < maxl && !s.err; ++i) { const u32 t = huff_load(s, np); np.w[dir + i] = t; }
				if (s.err || np.top - first > Q0_NODES) ok = 0;     // a malformed tree: k_dec_qhuff reports it
			}
		}
		else if (q_scheme == 2)
		{	// QualityRLEModeler::Decode: number of runs, the two alphabets, per quality symbol a tree of the next symbol and one of the lengths
			const u32 run_len = bs_word(s);
			u32 ln = 0;
			for (u32 i = 0; i < 256; ++i) if (bs_bit(s)) s_sym[n++] = (u8)i;
			for (u32 i = 0; i < 256; ++i) if (bs_bit(s)) s_ls[ln++] = (u8)i;
			bs_align(s);
			if (n > 1 && n * 128 <= Q0_MAXL * 64 && ln >= 1 && run_len >= 1 && !s.err)
			{
				dir = pool_take(np, 2 * n, &s.err);
				ok = 2;
				for (u32 i = 0; i < n && !s.err; ++i) { const u32 a = huff_load(s, np); np.w[dir + 2 * i] = a; const u32 c = huff_load(s, np); np.w[dir + 2 * i + 1] = c; }
				bs_align(s);
				if (s.err) ok = 0;
			}
			maxl = run_len; first = ln;
		}
		s_par[0] = ok; s_par[1] = np.top; s_par[2] = dir; s_par[3] = maxl; s_par[4] = n; s_par[5] = first;
		s_par[6] = (u32)s.bit; s_par[7] = q_scheme; s_par[8] = (u32)(s.bit >> 32);
	}
	__syncthreads();
	if (!s_par[0]) return;                                // k_dec_qhuff's
	if (s_par[0] == 2)
	{
		q0_rle(s, np, s_par, s_fast, s_sym, s_ls, d, S, rp, text, lossy);
		return;
	}
	const u32 top = s_par[1], dir = s_par[2], maxl = s_par[3], n = s_par[4], first = s_par[5], q_scheme = s_par[7];
	// the trees, two bytes per node (a child is an internal node of the same tree, < 128, or 0x80 | symbol); the nodes were written by
	// lane 0 a moment ago: read past this CU's vector cache
	for (u32 i = threadIdx.x; i < top - first; i += blockDim.x)
	{
		const u32 t = __hip_atomic_load(np.w + first + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const u32 l = t & 0xFFFFu, r = t >> 16;
		s_cn[i] = (u16)(((l & 0x8000u) ? 0x80u | (l & 0x7Fu) : l & 0x7Fu) | (((r & 0x8000u) ? 0x80u | (r & 0x7Fu) : r & 0x7Fu) << 8));
	}
	for (u32 i = threadIdx.x; i < maxl; i += blockDim.x)
		s_cdir[i] = (u16)(__hip_atomic_load(np.w + dir + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - first);
	__syncthreads();
	// per position and 6-bit prefix: character | length << 8 | special << 11, or 0x8000 | the node reached after 6 bits
	for (u32 e = threadIdx.x; e < maxl * 64; e += blockDim.x)
	{
		const u32 base = s_cdir[e >> 6], bits = e & 63u;
		u32 node = 0, val = 0;
		for (u32 k = 1; k <= 6; ++k)
		{
			const u32 c = s_cn[base + node];
			const u32 child = ((bits >> (6 - k)) & 1u) ? c >> 8 : c & 0xFFu;
			if (child & 0x80u)
			{	// the CHARACTER, the code length and whether the base lives in the quality stream: one read per symbol
				const u32 x = child & 0x7Fu, qv = x < n ? (u32)s_sym[x] : 255u;
				val = qv | (k << 8) | (q_special(qv, lossy) ? 0x800u : 0u);
				break;
			}
			node = child;
			if (k == 6) val = 0x8000u | node;
		}
		s_fast[e] = (u16)val;
	}
	__syncthreads();
	if (threadIdx.x != 0) return;

	s.bit = ((u64)s_par[8] << 32) | s_par[6];
	SWin w; sw_init(w, s);
	const bool truncated = q_scheme == 1;
	const u32 max_bits = dec_bit_length(maxl);
	const u32 variable = truncated ? sw_bits(w, 1) : 0u;
	const u32 hash_sym = lossy ? 1u : 2u;                    // HashSymbolQuantized / HashSymbolNormal
	u32 d_total = 0, err = 0;
	const u32 n_recs = S->n_recs;
	const u64 g0 = d.rec_base;
	u32 nql = n_recs ? (u32)rp.len[g0] : 0u, nqo = n_recs ? rp.qual_off[g0] : 0u;       // one record ahead
	for (u32 k = 0; k < n_recs && !err; ++k)
	{
		const u64 g = g0 + k;
		const u32 ql = nql; u8* q = text + nqo;
		if (k + 1 < n_recs) { nql = rp.len[g + 1]; nqo = rp.qual_off[g + 1]; }
		u32 th = ql, ncount = 0;
		if (truncated && sw_bits(w, 1)) th = sw_bits(w, variable ? dec_bit_length(ql) : max_bits);
		if (th > ql || th > maxl) { err |= DEC_ERR_FORMAT; break; }
		// one symbol: table entry for the next 6 bits; a code longer than that goes on bit by bit from the node it has reached
		auto one = [&](u32 j) -> u32
		{
			sw_refill(w);
			u32 e = s_fast[j * 64 + (u32)(w.w >> 58)];
			if (!(e & 0x8000u)) { const u32 len = (e >> 8) & 7u; w.w <<= len; w.n -= len; return e; }
			u32 node = e & 0x7FFFu;
			w.w <<= 6; w.n -= 6;
			const u32 base = s_cdir[j];
			u32 x = 0xFFFFu;
			for (u32 guard = 0; guard < 130; ++guard)
			{
				if (w.n == 0) sw_refill(w);
				const u32 c = s_cn[base + node];
				const u32 child = (w.w >> 63) ? c >> 8 : c & 0xFFu;
				w.w <<= 1; w.n -= 1;
				if (child & 0x80u) { x = child & 0x7Fu; break; }
				node = child;
			}
			if (x == 0xFFFFu) { err |= DEC_ERR_FORMAT; x = 0; }
			const u32 qv = x < n ? (u32)s_sym[x] : 255u;
			return qv | (q_special(qv, lossy) ? 0x800u : 0u);
		};
		u32 j = 0;
		for (; j + 4 <= th; j += 4)
		{	// four characters per store
			const u32 e0 = one(j), e1 = one(j + 1), e2 = one(j + 2), e3 = one(j + 3);
			*(dec_u32_unaligned*)(q + j) = (e0 & 0xFFu) | ((e1 & 0xFFu) << 8) | ((e2 & 0xFFu) << 16) | (e3 << 24);
			ncount += ((e0 >> 11) & 1u) + ((e1 >> 11) & 1u) + ((e2 >> 11) & 1u) + ((e3 >> 11) & 1u);
		}
		for (; j < th; ++j) { const u32 e = one(j); q[j] = (u8)e; ncount += (e >> 11) & 1u; }
		for (; j < ql; ++j) q[j] = (u8)hash_sym;               // the truncated tail (never a base that lives in the quality stream)
		rp.kept[g] = (u16)(ql - ncount); rp.d_off[g] = d_total; d_total += ql - ncount;
	}
	S->d_total = d_total;
	sw_finish(w, s);
	s.err |= err;
	bs_align(s);
	S->dna_pos = bs_pos(s);
	S->q_done = 1;
	S->err |= s.err;
}
