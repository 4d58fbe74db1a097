// Static-Huffman side of the codec:
//   HuffmanEncoder::Complete / StoreTree     src/huffman.cpp:94-221   (huff_build / huff_store, one lane per tree)
//   DnaModelerBasicB2 / DnaModelerHuffman    src/DnaModelerBasicB2.h:34-46, src/DnaModelerHuffman.cpp:21-73
//   QualityPositionModelerPlain/Truncated    src/QualityPositionModeler.cpp:24-287
//   QualityRLEModeler                        src/QualityRLEModeler.cpp:121-373
// Variable-length codes are placed by prefix-scanning code lengths into absolute bit offsets and
// OR-ing each code into the zero-initialised staging words (put_bits).
#pragma once
#include "k_common.h"
#include "k_parse.h"

// ---------------------------------------------------------------------------------------------
// Huffman construction, serial per lane.  ws = 10*n_eff u32 of private workspace in HBM:
// [heap_sym n][heap_freq n][left 2n][right 2n][code 2n][len 2n].  The reference's heap comparator is
// a strict total order (lowest frequency, then lowest id; merged node i gets id n+i), so selecting
// the minimum reproduces its tree (src/huffman.h:67-70).
// ---------------------------------------------------------------------------------------------
struct HuffView { u32 n; i32 root; u32* code; u32* len; i32* left; i32* right; };

__device__ __forceinline__ u32 huff_ws_words(u32 n) { const u32 m = n < 2 ? 2 : n; return 10 * m; }

__device__ inline u32 hsel_min(const u32* hs, const u32* hf, u32 n)
{
	u32 m = 0;
	for (u32 i = 1; i < n; ++i)
		if (hf[i] < hf[m] || (hf[i] == hf[m] && hs[i] < hs[m])) m = i;
	return m;
}

// freq(i) is read through a strided pointer so that histogram rows/columns can be used in place
__device__ inline HuffView huff_build(const u32* freqs, u32 stride, u32 n_in, u32* ws, u32* err)
{
	u32 n = n_in;
	if (n < 2) { n = 2; atomicOr(err, (u32)DSRC_ERR_REF_UB); }     // reference reads a stale slot (Appendix B.4)
	u32* hs = ws; u32* hf = ws + n;
	HuffView h;
	h.n = n; h.left = (i32*)(ws + 2 * n); h.right = (i32*)(ws + 4 * n); h.code = ws + 6 * n; h.len = ws + 8 * n;
	for (u32 i = 0; i < n; ++i) { hs[i] = i; hf[i] = i < n_in ? freqs[(u64)i * stride] : 0; }
	for (u32 i = 0; i < 2 * n - 1; ++i) { h.code[i] = 0; h.len[i] = 0; h.left[i] = i < n ? -1 : 0; h.right[i] = i < n ? -1 : 0; }

	u32 hsz = n;
	bool special = false; u32 spl_s = 0, spl_f = 0, spr_s = 0, spr_f = 0;
	if (hsz == 2)
	{
		const u32 top = hsel_min(hs, hf, 2);
		if (hf[top] == 0)
		{	// (*) special case: patched in place, the original top stays the first popped (src/huffman.cpp:124-131)
			hf[top] = 1; if (hf[1 - top] == 0) hf[1 - top] = 1;
			special = true; spl_s = hs[top]; spl_f = hf[top]; spr_s = hs[1 - top]; spr_f = hf[1 - top];
		}
	}
	else
	{
		while (hsz > 2)
		{
			const u32 m = hsel_min(hs, hf, hsz);
			if (hf[m] != 0) break;
			--hsz; hs[m] = hs[hsz]; hf[m] = hf[hsz];
		}
	}
	const u32 present = hsz;
	for (u32 i = 0; i + 1 < present; ++i)
	{
		u32 ls, lf, rs, rf;
		if (special) { ls = spl_s; lf = spl_f; rs = spr_s; rf = spr_f; hsz = 0; }
		else
		{
			u32 m = hsel_min(hs, hf, hsz); ls = hs[m]; lf = hf[m]; --hsz; hs[m] = hs[hsz]; hf[m] = hf[hsz];
			m = hsel_min(hs, hf, hsz);     rs = hs[m]; rf = hf[m]; --hsz; hs[m] = hs[hsz]; hf[m] = hf[hsz];
		}
		hs[hsz] = n + i; hf[hsz] = lf + rf; ++hsz;
		h.left[n + i] = (i32)ls; h.right[n + i] = (i32)rs;
	}
	for (i32 i = (i32)(n + present) - 2; i >= (i32)n; --i)
	{
		const i32 l = h.left[i], r = h.right[i];
		h.len[l] = h.len[i] + 1; h.code[l] = h.code[i] << 1;
		h.len[r] = h.len[i] + 1; h.code[r] = (h.code[i] << 1) | 1u;
	}
	h.root = (i32)(n + present) - 2;
	for (u32 i = 0; i < n; ++i)
		if (h.len[i] > 31) atomicOr(err, (u32)DSRC_ERR_CODE_TOO_LONG);
	return h;
}

// serial MSB-first byte sink for the small dictionary records written by a single lane
struct ByteSink { u8* p; u32 pos; u32 acc; u32 nb; };
__device__ __forceinline__ void bs_bits(ByteSink* s, u32 v, u32 n)
{
	for (i32 k = (i32)n - 1; k >= 0; --k)
	{
		s->acc = (s->acc << 1) | ((v >> k) & 1u);
		if (++s->nb == 8) { s->p[s->pos++] = (u8)s->acc; s->acc = 0; s->nb = 0; }
	}
}
__device__ __forceinline__ void bs_flush(ByteSink* s) { if (s->nb) { s->p[s->pos++] = (u8)(s->acc << (8 - s->nb)); s->acc = 0; s->nb = 0; } }
__device__ __forceinline__ void bs_byte(ByteSink* s, u32 v) { s->p[s->pos++] = (u8)v; }
__device__ __forceinline__ void bs_be32(ByteSink* s, u32 v) { bs_byte(s, v >> 24); bs_byte(s, v >> 16); bs_byte(s, v >> 8); bs_byte(s, v); }

__device__ __forceinline__ u32 huff_tree_cap(u32 n) { const u32 m = n < 2 ? 2 : n; return (16 + (2 * m + m * 10) / 8 + 8 + 3) & ~3u; }

// HuffmanEncoder::StoreTree (src/huffman.cpp:177-221): [BE32 memSize][BE32 root][BE32 n][u8 min_len][pre-order bits][pad].
// `stack` = n words of workspace (may alias the dead heap area of ws).  Returns the record size in bytes.
__device__ inline u32 huff_store(const HuffView& h, u8* dst, u32* stack)
{
	ByteSink s; s.p = dst; s.pos = 4; s.acc = 0; s.nb = 0;
	u32 bits_per_id = ilog2_floor(h.n);
	if (h.n & (h.n - 1)) bits_per_id++;
	u32 min_len = h.n;
	for (u32 i = 0; i < h.n; ++i)
		if (h.len[i] < min_len && h.len[i] > 0) min_len = h.len[i];
	bs_be32(&s, (u32)h.root); bs_be32(&s, h.n); bs_byte(&s, min_len);
	u32 sp = 0;
	stack[sp++] = (u32)h.root;
	while (sp)
	{
		const i32 id = (i32)stack[--sp];
		if (h.left[id] == -1) { bs_bits(&s, 1, 1); bs_bits(&s, (u32)id, bits_per_id); }
		else { bs_bits(&s, 0, 1); stack[sp++] = (u32)h.right[id]; stack[sp++] = (u32)h.left[id]; }
	}
	bs_flush(&s);
	const u32 mem = s.pos;
	dst[0] = (u8)(mem >> 24); dst[1] = (u8)(mem >> 16); dst[2] = (u8)(mem >> 8); dst[3] = (u8)mem;
	return mem;
}

// copy a serially written dictionary record into the staging stream (byte k -> words[k^3])
__device__ __forceinline__ void stage_bytes(u32* words, u64 at, const u8* src, u32 n)
{
	for (u32 i = 0; i < n; ++i) put_byte(words, at + i, src[i]);
}

// ---------------------------------------------------------------------------------------------
// per-block scratch for the level-0 quality / DNA paths (u32 words, zeroed before use):
//   [0 .. hist_words)            histograms
//   codes  : (code,len) tables
//   trees  : serialized trees, fixed slot size
//   ws     : Huffman workspaces
// Offsets are computed on the host from the block statistics (QuaPlan).
// ---------------------------------------------------------------------------------------------
struct QuaPlan
{
	u64 scr;              // u32 index of the block's scratch
	u64 run_start;        // RLE: u32 index of the run-start array (q_total + 2 words, not zeroed)
	u32 hist_words;       // Plain/Trunc: max_len*q_count ; RLE: see k_qrle_*
	u32 code_off, len_off, tree_off, ws_off;   // u32 offsets inside the scratch
	u32 tree_slot;        // bytes per serialized tree slot
	u32 ws_slot;          // words per Huffman workspace
	u32 n_trees;
	u32 aux_off;          // per-record u32 (bit offsets)
	u32 scheme;
	u32 blk;
	u32 lf_off;           // RLE: run-length rank table (256 words)
	u32 pad0, pad1, pad2;
};

// ---- DNA level 0 --------------------------------------------------------------------------------
__global__ void __launch_bounds__(WG) k_dna_b2(const BlkDesc* desc, BlkState* st, const u8* d_stream, u32* word_pool)
{
	const u32 b = blockIdx.y;
	const BlkDesc d = desc[b];
	BlkState* S = &st[b];
	if (d.d_scheme != 0) return;
	const u8* s = d_stream + d.d_base;
	u32* out = word_pool + d.dna_out;
	const u32 n = S->d_total, nbytes = (n + 3) / 4;
	for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < nbytes; k += gridDim.x * blockDim.x)
	{
		u32 v = 0;
		for (u32 i = 0; i < 4; ++i)
		{
			const u32 t = 4 * k + i;
			v = (v << 2) | (t < n ? (s[t] & 3u) : 0u);
		}
		put_byte(out, 1 + k, v);
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) { put_byte(out, 0, 0); S->dna_bytes = 1 + nbytes; }
}

__global__ void __launch_bounds__(WG) k_dna_huff(const BlkDesc* desc, BlkState* st, const u8* d_stream, u32* word_pool, u32* scr_pool, const QuaPlan* plans)
{
	__shared__ u32 s_code[20], s_len[20];
	__shared__ u32 s_hdr;
	const u32 b = blockIdx.x;
	const BlkDesc d = desc[b];
	BlkState* S = &st[b];
	if (d.d_scheme != 1) return;
	const QuaPlan pl = plans[b];
	u32* scr = scr_pool + pl.scr;
	u32* out = word_pool + d.dna_out;
	if (threadIdx.x == 0)
	{
		// DnaModelerHuffman::ProcessStats inserts symbolFreqs[symbols[i]], i < symbolCount (Appendix B.2);
		// index 255 lands on qualityStats.symbolFreqs[229] in the reference's object layout.
		u32* fr = scr + pl.hist_words;          // 20 words of temp
		for (u32 i = 0; i < S->d_count; ++i)
		{
			const u32 x = S->d_sym[i];
			fr[i] = x == 255 ? S->q_freq[229] : S->d_freq[x];
		}
		HuffView h = huff_build(fr, 1, S->d_count, scr + pl.ws_off, &S->err);
		for (u32 i = 0; i < 20; ++i) { s_code[i] = i < h.n ? h.code[i] : 0; s_len[i] = i < h.n ? h.len[i] : 0; }
		u8* tr = (u8*)(scr + pl.tree_off);
		const u32 tb = huff_store(h, tr, scr + pl.ws_off);
		put_byte(out, 0, 1);
		u32 pres = 0;
		for (u32 i = 0; i < 20; ++i) pres = (pres << 1) | (S->d_sym[i] != 255 ? 1u : 0u);
		pres <<= 4;                              // 20 bits + pad -> 3 bytes
		put_byte(out, 1, pres >> 16); put_byte(out, 2, pres >> 8); put_byte(out, 3, pres);
		stage_bytes(out, 4, tr, tb);
		s_hdr = 4 + tb;
	}
	__syncthreads();
	const u8* s = d_stream + d.d_base;
	const u32 n = S->d_total;
	u64 bitpos = (u64)s_hdr * 8;
	for (u32 base = 0; base < n; base += blockDim.x)
	{
		const u32 t = base + threadIdx.x;
		u32 code = 0, len = 0;
		if (t < n) { const u32 x = S->d_sym[s[t]]; code = s_code[x < 20 ? x : 0]; len = s_len[x < 20 ? x : 0]; }
		u32 tot;
		const u32 off = block_excl_scan(len, &tot);
		if (t < n) put_bits(out, bitpos + off, code, len);
		bitpos += tot;
	}
	if (threadIdx.x == 0) S->dna_bytes = (u32)((bitpos + 7) / 8);
}

// ---- quality, Plain (scheme 0) / Truncated (scheme 1) --------------------------------------------
#define QPOS_LDS_WORDS 16384

__global__ void __launch_bounds__(WG) k_qpos_hist(const BlkDesc* desc, const BlkState* st, RecPools rp, const u8* q_stream, u32* scr_pool, const QuaPlan* plans)
{
	__shared__ u32 s_hist[QPOS_LDS_WORDS];
	__shared__ u8 s_rank[256];
	const u32 b = blockIdx.x;
	const BlkState* S = &st[b];
	if (plans[b].scheme > 1) return;
	const BlkDesc d = desc[b];
	const QuaPlan pl = plans[b];
	u32* gh = scr_pool + pl.scr;
	const u32 nsym = S->q_count, words = pl.hist_words;
	const bool use_lds = words <= QPOS_LDS_WORDS;
	u32* hist = use_lds ? s_hist : gh;
	for (u32 i = threadIdx.x; i < 256; i += blockDim.x) s_rank[i] = S->q_sym[i];
	if (use_lds) for (u32 i = threadIdx.x; i < words; i += blockDim.x) s_hist[i] = 0;
	__syncthreads();
	const u8* qs = q_stream + d.q_base;
	const bool trunc = plans[b].scheme == 1;
	const u32 lane = lane_id();
	for (u32 r = wave_id(); r < S->n_recs; r += (blockDim.x >> 6))
	{
		const u64 g = (u64)d.rec_base + r;
		const u32 n = trunc ? rp.trunc[g] : rp.len[g];
		const u8* q = qs + rp.q_off[g];
		for (u32 j = lane; j < n; j += 64) atomicAdd(&hist[j * nsym + s_rank[q[j]]], 1u);
	}
	__syncthreads();
	if (use_lds) for (u32 i = threadIdx.x; i < words; i += blockDim.x) gh[i] = s_hist[i];
}

// one lane per position tree
__global__ void __launch_bounds__(64) k_qpos_trees(BlkState* st, u32* scr_pool, const QuaPlan* plans)
{
	const u32 b = blockIdx.y;
	BlkState* S = &st[b];
	if (plans[b].scheme > 1) return;
	const QuaPlan pl = plans[b];
	const u32 pos = blockIdx.x * blockDim.x + threadIdx.x;
	if (pos >= S->max_len) return;
	u32* scr = scr_pool + pl.scr;
	const u32 nsym = S->q_count;
	u32* ws = scr + pl.ws_off + (u64)pos * pl.ws_slot;
	HuffView h = huff_build(scr + (u64)pos * nsym, 1, nsym, ws, &S->err);
	for (u32 i = 0; i < nsym; ++i)
	{
		scr[pl.code_off + (u64)pos * nsym + i] = h.code[i];
		scr[pl.len_off + (u64)pos * nsym + i] = h.len[i];
	}
	u8* tr = (u8*)(scr + pl.tree_off) + (u64)pos * pl.tree_slot;
	const u32 tb = huff_store(h, tr + 4, ws);
	*(u32*)tr = tb;
}

// Code tables of the first positions are held in LDS as (len << 32 | code); positions beyond QPOS_TAB / nsym fall
// back to the global tables.  Codes are written 64 symbols at a time: the wave ORs them into a private LDS strip at their
// bit offsets, then stores whole words -- only the first and last word of a strip can be shared with a neighbour and go
// out as atomicOr (before: two global atomics and two global table reads per symbol; 75 -> 32 ms per 512 blocks).
#define QPOS_TAB 8192u
#define QPOS_K 3u

__global__ void __launch_bounds__(WG) k_qpos_emit(const BlkDesc* desc, BlkState* st, RecPools rp, const u8* q_stream, u32* word_pool, u32* scr_pool, const QuaPlan* plans)
{
	__shared__ u8 s_rank[256];
	__shared__ u32 s_hdr;
	__shared__ u64 s_tab[QPOS_TAB];
	__shared__ u32 s_stage[WG / 64][64 * QPOS_K + 4];
	const u32 b = blockIdx.x;
	BlkState* S = &st[b];
	if (plans[b].scheme > 1) return;
	const BlkDesc d = desc[b];
	const QuaPlan pl = plans[b];
	u32* scr = scr_pool + pl.scr;
	u32* out = word_pool + d.qua_out;
	const u32 nsym = S->q_count, maxl = S->max_len, n_recs = S->n_recs;
	const bool trunc = plans[b].scheme == 1;
	const bool variable = S->min_len != S->max_len;
	const u32 max_bits = bit_length32(maxl);
	const u32* clen = scr + pl.len_off;
	const u32* ccode = scr + pl.code_off;
	const u32 p_fit = nsym ? (QPOS_TAB / nsym < maxl ? QPOS_TAB / nsym : maxl) : 0;
	for (u32 i = threadIdx.x; i < 256; i += blockDim.x) s_rank[i] = S->q_sym[i];
	for (u32 i = threadIdx.x; i < p_fit * nsym; i += blockDim.x) s_tab[i] = ((u64)clen[i] << 32) | ccode[i];
	if (threadIdx.x == 0)
	{
		put_byte(out, 0, plans[b].scheme);
		put_be32(out, 1, maxl);
		for (u32 k = 0; k < 32; ++k)
		{
			u32 v = 0;
			for (u32 i = 0; i < 8; ++i) v = (v << 1) | (S->q_sym[8 * k + i] != 255 ? 1u : 0u);
			put_byte(out, 5 + k, v);
		}
		u32 at = 37;
		for (u32 p = 0; p < maxl; ++p)
		{
			const u8* tr = (const u8*)(scr + pl.tree_off) + (u64)p * pl.tree_slot;
			const u32 tb = *(const u32*)tr;
			stage_bytes(out, at, tr + 4, tb);
			at += tb;
		}
		s_hdr = at;
	}
	__syncthreads();
	u32* rbits = scr + pl.aux_off;
	const u8* qs = q_stream + d.q_base;
	const u32 lane = lane_id();
	auto entry = [&](u32 j, u32 qv) -> u64
	{
		const u64 ix = (u64)j * nsym + s_rank[qv];
		return j < p_fit ? s_tab[ix] : (((u64)clen[ix] << 32) | ccode[ix]);
	};
	// pass A: bits per record
	for (u32 r = wave_id(); r < n_recs; r += (blockDim.x >> 6))
	{
		const u64 g = (u64)d.rec_base + r;
		const u32 ql = rp.len[g], tl = rp.trunc[g];
		const u32 n = trunc ? tl : ql;
		const u8* q = qs + rp.q_off[g];
		u32 bits = 0;
		for (u32 j = lane; j < n; j += 64) bits += (u32)(entry(j, q[j]) >> 32);
		bits = wave_sum(bits);
		if (trunc) bits += 1 + (ql != tl ? (variable ? bit_length32(ql) : max_bits) : 0);
		if (lane == 0) rbits[r] = bits;
	}
	__syncthreads();
	// pass B: exclusive scan -> absolute bit offsets
	u64 carry = (u64)s_hdr * 8 + (trunc ? 1 : 0);
	__shared__ u64 s_total;
	{
		u64 run = carry;
		for (u32 base = 0; base < n_recs; base += blockDim.x)
		{
			const u32 r = base + threadIdx.x;
			const u32 v = r < n_recs ? rbits[r] : 0;
			u32 tot;
			const u32 ex = block_excl_scan(v, &tot);
			// offsets relative to `carry` fit in 32 bits for blocks < 512 MiB of codes
			if (r < n_recs) rbits[r] = (u32)(run - carry) + ex;
			run += tot;
		}
		if (threadIdx.x == 0) s_total = run;
	}
	__syncthreads();
	if (threadIdx.x == 0 && trunc) put_bits(out, (u64)s_hdr * 8, variable ? 1u : 0u, 1);
	// pass C: write
	u32* stg = s_stage[wave_id()];
	for (u32 r = wave_id(); r < n_recs; r += (blockDim.x >> 6))
	{
		const u64 g = (u64)d.rec_base + r;
		const u32 ql = rp.len[g], tl = rp.trunc[g];
		const u32 n = trunc ? tl : ql;
		const u8* q = qs + rp.q_off[g];
		u64 at = carry + rbits[r];
		if (trunc)
		{
			if (lane == 0)
			{
				put_bits(out, at, ql != tl ? 1u : 0u, 1);
				if (ql != tl) put_bits(out, at + 1, tl, variable ? bit_length32(ql) : max_bits);
			}
			at += 1 + (ql != tl ? (variable ? bit_length32(ql) : max_bits) : 0);
		}
		for (u32 j0 = 0; j0 < n; j0 += 64 * QPOS_K)
		{
			// QPOS_K consecutive symbols per lane: one scan, one strip per 192 symbols (a 150-base read is one strip)
			u32 code[QPOS_K], len[QPOS_K], lsum = 0;
#pragma unroll
			for (u32 k = 0; k < QPOS_K; ++k)
			{
				const u32 j = j0 + lane * QPOS_K + k;
				code[k] = 0; len[k] = 0;
				if (j < n) { const u64 e = entry(j, q[j]); code[k] = (u32)e; len[k] = (u32)(e >> 32); }
				lsum += len[k];
			}
			const u32 inc = wave_incl_scan(lsum);
			const u32 T = __shfl(inc, 63);
			if (T)
			{
				const u32 sh0 = (u32)at & 31u;
				const u64 w0 = at >> 5;
				const u32 nw = (sh0 + T + 31u) >> 5;             // <= 64 * QPOS_K + 1 words for codes of <= 32 bits
				for (u32 i = lane; i < nw + 1; i += 64) stg[i] = 0;
				wave_fence();
				u32 rel = sh0 + inc - lsum;
#pragma unroll
				for (u32 k = 0; k < QPOS_K; ++k)
				{
					if (len[k])
					{
						const u32 c = len[k] < 32 ? code[k] & ((1u << len[k]) - 1u) : code[k];
						const u64 v = (u64)c << (64u - len[k] - (rel & 31u));
						atomicOr(&stg[rel >> 5], (u32)(v >> 32));
						if ((u32)v) atomicOr(&stg[(rel >> 5) + 1], (u32)v);
						rel += len[k];
					}
				}
				wave_fence();
				for (u32 i = lane; i < nw; i += 64)
				{
					const u32 v = stg[i];
					if (v) { if (i == 0 || i == nw - 1) atomicOr(&out[w0 + i], v); else out[w0 + i] = v; }
				}
				wave_fence();
			}
			at += T;
		}
	}
	if (threadIdx.x == 0) S->qua_bytes = (u32)((s_total + 7) / 8);
}

// ---- quality, RLE (scheme 2) ------------------------------------------------------------------------
// runs cross record boundaries; a run holds at most 255 symbols (length byte 0..254).
#define QRUN_ITEMS 4u
__global__ void __launch_bounds__(WG) k_qrle_runs(const BlkDesc* desc, BlkState* st, const u8* q_stream, u32* scr_pool, const QuaPlan* plans)
{
	__shared__ u32 s_wmax[WAVES];
	const u32 b = blockIdx.x;
	BlkState* S = &st[b];
	if (plans[b].scheme != 2) return;
	const BlkDesc d = desc[b];
	const QuaPlan pl = plans[b];
	u32* scr = scr_pool + pl.scr;
	u32* run_start = scr_pool + pl.run_start;
	const u8* q = q_stream + d.q_base;
	const u32 n = S->q_total;
	u32 carry_head = 0, carry_runs = 0;
	// QRUN_ITEMS consecutive symbols per thread (one 4-byte load; the quality streams are 64-byte aligned and padded)
	for (u32 base = 0; base < n; base += blockDim.x * QRUN_ITEMS)
	{
		const u32 t0 = base + threadIdx.x * QRUN_ITEMS;
		const u32 w = t0 < n ? *(const u32*)(q + t0) : 0u;
		u32 prevb = (t0 > 0 && t0 < n) ? q[t0 - 1] : 256u;
		u32 hv = 0, hmask = 0;                      // last head position (+1) among my symbols; which of them are heads
#pragma unroll
		for (u32 i = 0; i < QRUN_ITEMS; ++i)
		{
			const u32 t = t0 + i, bq = (w >> (8 * i)) & 255u;
			if (t < n && (t == 0 || bq != prevb)) { hv = t + 1; hmask |= 1u << i; }
			prevb = bq;
		}
		// inclusive max-scan of head positions (+1 so that 0 means "none")
		u32 v = hv;
		for (u32 dd = 1; dd < 64; dd <<= 1) { const u32 o = __shfl_up(v, dd); if (lane_id() >= dd && o > v) v = o; }
		if (lane_id() == 63) s_wmax[wave_id()] = v;
		__syncthreads();
		u32 pre = carry_head, all = carry_head;
		for (u32 wv = 0; wv < (blockDim.x >> 6); ++wv) { const u32 x = s_wmax[wv]; if (wv < wave_id() && x > pre) pre = x; if (x > all) all = x; }
		u32 cur = __shfl_up(v, 1);                  // last head before my first symbol
		if (lane_id() == 0) cur = 0;
		if (pre > cur) cur = pre;
		u32 cnt = 0, where[QRUN_ITEMS];
#pragma unroll
		for (u32 i = 0; i < QRUN_ITEMS; ++i)
		{
			const u32 t = t0 + i;
			if ((hmask >> i) & 1u) cur = t + 1;       // cur-1 = start of the maximal run containing t
			where[i] = 0;
			if (t < n && ((t - (cur - 1)) % 255u == 0)) { where[cnt] = t; ++cnt; }
		}
		u32 tot;
		const u32 ex = block_excl_scan(cnt, &tot);
		for (u32 i = 0; i < cnt; ++i) run_start[carry_runs + ex + i] = where[i];
		carry_runs += tot; carry_head = all;
	}
	if (threadIdx.x == 0) { S->q_runs = carry_runs; run_start[carry_runs] = n; }
}

#define QRLE_LDS_HIST 8192u
// histograms: lf[256] (run-length symbols), then qF[prev][q] and lF[q][l] over dense ranks
__global__ void __launch_bounds__(WG) k_qrle_hist(const BlkDesc* desc, BlkState* st, const u8* q_stream, u32* scr_pool, const QuaPlan* plans)
{
	__shared__ u32 s_lf[256];
	__shared__ u8 s_lrank[256];
	__shared__ u8 s_qrank[256];
	__shared__ u32 s_ln;
	const u32 b = blockIdx.x;
	BlkState* S = &st[b];
	if (plans[b].scheme != 2) return;
	const BlkDesc d = desc[b];
	const QuaPlan pl = plans[b];
	u32* scr = scr_pool + pl.scr;
	const u32* run_start = scr_pool + pl.run_start;
	const u8* q = q_stream + d.q_base;
	__shared__ u32 s_qf[256];
	__shared__ u32 s_qn;
	const u32 R = S->q_runs;
	for (u32 i = threadIdx.x; i < 256; i += blockDim.x) { s_lf[i] = 0; s_qf[i] = 0; }
	__syncthreads();
	for (u32 k = threadIdx.x; k < R; k += blockDim.x) { atomicAdd(&s_lf[run_start[k + 1] - run_start[k] - 1], 1u); atomicAdd(&s_qf[q[run_start[k]]], 1u); }
	__syncthreads();
	if (threadIdx.x == 0)
	{
		u32 ln = 0, qc = 0;
		for (u32 i = 0; i < 256; ++i) s_lrank[i] = s_lf[i] ? (u8)ln++ : (u8)255;
		for (u32 i = 0; i < 256; ++i) s_qrank[i] = s_qf[i] ? (u8)qc++ : (u8)255;      // EncodeRecords' own symbol set (src/QualityRLEModeler.cpp:142-205)
		s_ln = ln; s_qn = qc;
		S->scratch[0] = ln; S->rle_qn = qc;
	}
	__syncthreads();
	for (u32 i = threadIdx.x; i < 256; i += blockDim.x) S->rle_qsym[i] = s_qrank[i];
	const u32 qn = s_qn;
	u32* lr = scr + pl.lf_off;                  // 256 words: run-length rank table for later kernels
	for (u32 i = threadIdx.x; i < 256; i += blockDim.x) lr[i] = s_lrank[i];
	if (qn <= 1) return;
	const u32 ln = s_ln;
	u32* qF = scr;                              // [qn][qn]
	u32* lF = scr + (u64)qn * qn;               // [qn][ln]
	// the counters are few (4 x 4 + 4 x ln for four-level qualities) and every thread of the workgroup hits them:
	// count in LDS and add the totals to the (zeroed) global tables once; global atomics only when they do not fit
	__shared__ u32 s_h[QRLE_LDS_HIST];
	const u32 words = qn * qn + qn * ln;
	const bool in_lds = words <= QRLE_LDS_HIST;
	if (in_lds) { for (u32 i = threadIdx.x; i < words; i += blockDim.x) s_h[i] = 0; __syncthreads(); }
	for (u32 k = threadIdx.x; k < R; k += blockDim.x)
	{
		const u32 rs = run_start[k];
		const u32 qs = s_qrank[q[rs]];
		const u32 prev = k ? s_qrank[q[rs - 1]] : 0;          // the run before k ends right before k's first symbol
		const u32 l = s_lrank[run_start[k + 1] - rs - 1];
		if (in_lds) { atomicAdd(&s_h[prev * qn + qs], 1u); atomicAdd(&s_h[qn * qn + qs * ln + l], 1u); }
		else { atomicAdd(&qF[(u64)prev * qn + qs], 1u); atomicAdd(&lF[(u64)qs * ln + l], 1u); }
	}
	if (in_lds)
	{
		__syncthreads();
		for (u32 i = threadIdx.x; i < words; i += blockDim.x) if (s_h[i]) scr[i] = s_h[i];
	}
}

// 2*qn trees: tree 2i = q | prev = i ; tree 2i+1 = len | q = i
__global__ void __launch_bounds__(64) k_qrle_trees(BlkState* st, u32* scr_pool, const QuaPlan* plans)
{
	const u32 b = blockIdx.y;
	BlkState* S = &st[b];
	if (plans[b].scheme != 2 || S->rle_qn <= 1) return;
	const QuaPlan pl = plans[b];
	const u32 qn = S->rle_qn, ln = S->scratch[0];
	const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= 2 * qn) return;
	u32* scr = scr_pool + pl.scr;
	const u32 i = k >> 1;
	const bool is_len = k & 1;
	const u32 nsym = is_len ? ln : qn;
	const u32* fr = is_len ? scr + (u64)qn * qn + (u64)i * ln : scr + (u64)i * qn;
	u32* ws = scr + pl.ws_off + (u64)k * pl.ws_slot;
	HuffView h = huff_build(fr, 1, nsym, ws, &S->err);
	// code tables: q codes [qn][qn] at code_off, len codes [qn][256] after them
	const u32 stride = is_len ? 256u : qn;
	const u64 cbase = is_len ? (u64)qn * qn + (u64)i * 256 : (u64)i * qn;
	for (u32 x = 0; x < nsym; ++x) { scr[pl.code_off + cbase + x] = h.code[x]; scr[pl.len_off + cbase + x] = h.len[x]; }
	(void)stride;
	u8* tr = (u8*)(scr + pl.tree_off) + (u64)k * pl.tree_slot;
	const u32 tb = huff_store(h, tr + 4, ws);
	*(u32*)tr = tb;
}

#define QRLE_LDS_TAB 4096u
#define QRLE_ITEMS 4u
__global__ void __launch_bounds__(WG) k_qrle_emit(const BlkDesc* desc, BlkState* st, const u8* q_stream, u32* word_pool, u32* scr_pool, const QuaPlan* plans)
{
	__shared__ u8 s_qrank[256];
	__shared__ u32 s_hdr;
	__shared__ u64 s_ct[QRLE_LDS_TAB];
	const u32 b = blockIdx.x;
	BlkState* S = &st[b];
	if (plans[b].scheme != 2) return;
	const BlkDesc d = desc[b];
	const QuaPlan pl = plans[b];
	u32* scr = scr_pool + pl.scr;
	u32* out = word_pool + d.qua_out;
	const u32* run_start = scr_pool + pl.run_start;
	const u32* lrank = scr + pl.lf_off;
	const u8* q = q_stream + d.q_base;
	const u32 R = S->q_runs, qn = S->rle_qn, ln = S->scratch[0];
	for (u32 i = threadIdx.x; i < 256; i += blockDim.x) s_qrank[i] = S->rle_qsym[i];
	if (threadIdx.x == 0)
	{
		put_byte(out, 0, 2);
		put_be32(out, 1, R);
		for (u32 k = 0; k < 32; ++k)
		{
			u32 v = 0, w = 0;
			for (u32 i = 0; i < 8; ++i) { v = (v << 1) | (S->rle_qsym[8 * k + i] != 255 ? 1u : 0u); w = (w << 1) | (lrank[8 * k + i] != 255 ? 1u : 0u); }
			put_byte(out, 5 + k, v); put_byte(out, 37 + k, w);
		}
		u32 at = 69;
		if (qn > 1)
		{
			for (u32 k = 0; k < 2 * qn; ++k)
			{
				const u8* tr = (const u8*)(scr + pl.tree_off) + (u64)k * pl.tree_slot;
				const u32 tb = *(const u32*)tr;
				stage_bytes(out, at, tr + 4, tb);
				at += tb;
			}
		}
		else if (ln > 1)
		{
			put_byte(out, at, lrank[run_start[1] - run_start[0] - 1]);
			at += 1;
		}
		s_hdr = at;
	}
	__syncthreads();
	u64 bitpos = (u64)s_hdr * 8;
	if (qn > 1)
	{
		const u32* ccode = scr + pl.code_off; const u32* clen = scr + pl.len_off;
		// code tables in LDS when they fit (q | prev: [qn][qn], then len | q: [qn][256]), as (len << 32 | code)
		const u32 tabw = qn * qn + qn * 256u;
		const bool in_lds = tabw <= QRLE_LDS_TAB;
		if (in_lds) { for (u32 i = threadIdx.x; i < tabw; i += blockDim.x) s_ct[i] = ((u64)clen[i] << 32) | ccode[i]; }
		__syncthreads();
		// QRLE_ITEMS consecutive runs per thread: one workgroup scan per 4096 runs, and the thread's codes leave through a
		// 64-bit accumulator (one or two word updates instead of two per run)
		for (u32 base = 0; base < R; base += blockDim.x * QRLE_ITEMS)
		{
			const u32 k0 = base + threadIdx.x * QRLE_ITEMS;
			u32 cc[2 * QRLE_ITEMS], ll[2 * QRLE_ITEMS], sum = 0;
#pragma unroll
			for (u32 i = 0; i < QRLE_ITEMS; ++i)
			{
				const u32 k = k0 + i;
				cc[2 * i] = cc[2 * i + 1] = ll[2 * i] = ll[2 * i + 1] = 0;
				if (k < R)
				{
					const u32 rs = run_start[k];
					const u32 qs = s_qrank[q[rs]];
					const u32 prev = k ? s_qrank[q[rs - 1]] : 0;
					const u32 l = lrank[run_start[k + 1] - rs - 1];
					const u32 i1 = prev * qn + qs, i2 = qn * qn + qs * 256u + l;
					const u64 e1 = in_lds ? s_ct[i1] : (((u64)clen[i1] << 32) | ccode[i1]);
					const u64 e2 = in_lds ? s_ct[i2] : (((u64)clen[i2] << 32) | ccode[i2]);
					cc[2 * i] = (u32)e1; ll[2 * i] = (u32)(e1 >> 32); cc[2 * i + 1] = (u32)e2; ll[2 * i + 1] = (u32)(e2 >> 32);
				}
				sum += ll[2 * i] + ll[2 * i + 1];
			}
			u32 tot;
			const u32 off = block_excl_scan(sum, &tot);
			u64 at = bitpos + off, buf = 0; u32 nb = 0;
			auto flush = [&]()
			{
				if (!nb) return;
				const u32 h = nb < 32 ? nb : 32;
				put_bits(out, at, (u32)(buf >> (64 - h)), h);
				if (nb > 32) put_bits(out, at + 32, (u32)(buf >> (64 - nb)), nb - 32);
				at += nb; nb = 0; buf = 0;
			};
#pragma unroll
			for (u32 i = 0; i < 2 * QRLE_ITEMS; ++i)
			{
				const u32 len = ll[i];
				if (!len) continue;
				if (nb + len > 64) flush();
				const u32 c = len < 32 ? cc[i] & ((1u << len) - 1u) : cc[i];
				buf |= (u64)c << (64 - nb - len); nb += len;
			}
			flush();
			bitpos += tot;
		}
	}
	if (threadIdx.x == 0) S->qua_bytes = (u32)((bitpos + 7) / 8);
}
