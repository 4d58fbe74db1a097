// Order-k context modelling + range coding (the 83 % of the reference's CPU time at -d3 -q2):
//   TDnaRCOrderModeler            src/DnaModelerRCO.h:27-132
//   TQualityModelExt / encoders   src/QualityEncoder.h:24-367
//   TSymbolCoderRC                src/SymbolCoderRC.h:23-93
//   RangeEncoder                  src/RangeCoder.h:51-84
//
// The reference walks one 3.3 M-symbol chain per stream: table row -> (freq, cum, total) ->
// range update.  Context ids depend on the INPUT symbols only, never on coder state, so the
// chain is cut in three data-parallel stages and one short serial one:
//   k_ctx_*   : context id of every symbol (pure function of <= order+2 previous symbols)
//   k_sort    : stable LSD radix sort of (ctx, sym, t) by ctx -> each context's history is contiguous
//   k_replay  : one lane replays one context's history on a private counter row in LDS and
//               emits (total, cum, freq) for every symbol, scattered back to stream order
//   k_rc      : the integer range-coder recurrence, one LANE per stream (64 streams per wave);
//               triples are lane-interleaved so that the wave's loads are one coalesced 512 B row
// No adaptive table ever exists in HBM (the reference clears 2-64 MiB per block).
#pragma once
#include "k_common.h"

#define ELEM_T_BITS 32
#define ELEM_SYM_SHIFT 32
#define ELEM_CTX_SHIFT 40

struct CtxJob     // one (block, stream)
{
	u64 src_off;        // byte offset of the symbol stream (q_stream / d_stream)
	u64 elems;          // u64 index of sort buffer A
	u64 elems_b;        // u64 index of sort buffer B
	u64 trip;           // u64 index of this chain's first triple (group base + lane)
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
	u32 pad;
};

// ---- DNA context: hash of the previous `order` symbols, carried across records --------------
__global__ void __launch_bounds__(WG) k_ctx_dna(const CtxJob* jobs, const u8* d_stream, u64* pool, BlkState* st)
{
	const CtxJob j = jobs[blockIdx.y];
	const u8* s = d_stream + j.src_off;
	u64* e = pool + j.elems;
	const u32 ab = j.alpha_bits, n_alpha = 1u << ab;
	const u64 mask = (1ull << (ab * j.order)) - 1ull;
	bool bad = false;
	for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < j.n; t += gridDim.x * blockDim.x)
	{
		u64 h = 0;
		const u32 k0 = t < j.order ? t : j.order;
		for (u32 k = k0; k >= 1; --k) h = (h << ab) | s[t - k];
		h &= mask;
		const u32 sym = s[t];
		if (sym >= n_alpha) bad = true;                     // reference UB
		e[t] = (h << ELEM_CTX_SHIFT) | ((u64)(sym & (n_alpha - 1)) << ELEM_SYM_SHIFT) | t;
	}
	if (bad) atomicOr(&st[j.blk].err, (u32)DSRC_ERR_REF_UB);
}

// ---- quality context (TQualityModelBase::UpdateHash, src/QualityEncoder.h:77-94) -------------
// Before coding symbol t the hash slots are: k < order/2 : raw s[t-1-k];
// k >= order/2 : floor((s[t-1-k] + s[t-2-k]) / 2)  (order 1: slot 0 is raw).  Symbols before the
// start of the block read as 0.  ctx = (slots << alpha_bits) | position_context.
__global__ void __launch_bounds__(WG) k_ctx_qua(const CtxJob* jobs, const u8* q_stream, const u8* qp_stream, u64* pool, const BlkState* st)
{
	__shared__ u8 s_rank[256];
	const CtxJob j = jobs[blockIdx.y];
	for (u32 i = threadIdx.x; i < 256; i += blockDim.x) s_rank[i] = j.translate ? st[j.blk].q_sym[i] : (u8)i;
	__syncthreads();
	const u8* s = q_stream + j.src_off;
	const u8* qp = qp_stream + j.src_off;
	u64* e = pool + j.elems;
	const u32 ab = j.alpha_bits, half = j.order / 2;
	for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < j.n; t += gridDim.x * blockDim.x)
	{
		u32 v[8];                                           // v[k] = rank of s[t-1-k], k <= order (<= 6) + 1
		for (u32 k = 0; k <= j.order; ++k) v[k] = (t >= k + 1) ? s_rank[s[t - 1 - k]] : 0;
		u64 h = 0;
		for (u32 k = j.order; k >= 1; --k)
		{
			const u32 slot = k - 1;
			const u32 x = (slot < half || j.order == 1) ? v[slot] : ((v[slot] + v[slot + 1]) >> 1);
			h = (h << ab) | x;
		}
		const u32 pctx = qp[t] >> j.rescale_shift;
		const u64 ctx = (h << ab) | pctx;
		const u32 sym = s_rank[s[t]] & ((1u << ab) - 1u);
		e[t] = (ctx << ELEM_CTX_SHIFT) | ((u64)sym << ELEM_SYM_SHIFT) | t;
	}
}

// ---- stable LSD radix sort by ctx; one workgroup owns one stream ------------------------------
#define SORT_MAX_BINS 256
__global__ void __launch_bounds__(WG) k_sort(const CtxJob* jobs, u64* pool)
{
	__shared__ u32 s_base[SORT_MAX_BINS];
	__shared__ u32 s_cnt[WAVES][SORT_MAX_BINS];
	__shared__ u32 s_off[WAVES][SORT_MAX_BINS];
	const CtxJob j = jobs[blockIdx.x];
	const u32 n = j.n, bins = 1u << j.dbits;
	const u32 wv = wave_id(), nw = blockDim.x >> 6;

	for (u32 pass = 0; pass < j.passes; ++pass)
	{
		const u64* src = pool + ((pass & 1) ? j.elems_b : j.elems);
		u64* dst = pool + ((pass & 1) ? j.elems : j.elems_b);
		const u32 shift = ELEM_CTX_SHIFT + pass * j.dbits;

		for (u32 i = threadIdx.x; i < bins; i += blockDim.x) s_base[i] = 0;
		for (u32 i = threadIdx.x; i < WAVES * SORT_MAX_BINS; i += blockDim.x) (&s_cnt[0][0])[i] = 0;
		__syncthreads();
		for (u32 i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&s_base[(u32)(src[i] >> shift) & (bins - 1)], 1u);
		__syncthreads();
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
		__syncthreads();

		for (u32 tile = 0; tile < n; tile += blockDim.x)
		{
			const u32 i = tile + threadIdx.x;
			const bool valid = i < n;
			const u64 el = valid ? src[i] : 0;
			const u32 d = (u32)(el >> shift) & (bins - 1);
			u64 peers = __ballot(valid);
			for (u32 b = 0; b < j.dbits; ++b)
			{
				const u64 m = __ballot((d >> b) & 1u);
				peers &= ((d >> b) & 1u) ? m : ~m;
			}
			const u32 rank = (u32)__popcll(peers & lanemask_lt());
			if (valid && rank == 0) s_cnt[wv][d] = (u32)__popcll(peers);
			__syncthreads();
			for (u32 dd = threadIdx.x; dd < bins; dd += blockDim.x)
			{
				u32 run = s_base[dd];
				for (u32 w = 0; w < nw; ++w)
				{
					const u32 c = s_cnt[w][dd];
					s_cnt[w][dd] = 0;
					s_off[w][dd] = run;
					run += c;
				}
				s_base[dd] = run;
			}
			__syncthreads();
			if (valid) dst[s_off[wv][d] + rank] = el;
			__syncthreads();
		}
	}
}

// ---- model replay ---------------------------------------------------------------------------
#define REPLAY_WG 256
#define REPLAY_TILE (REPLAY_WG * 8)

template <int N>
__global__ void __launch_bounds__(REPLAY_WG) k_replay(const CtxJob* jobs, const u64* pool, u64* trip_pool)
{
	__shared__ u16 s_row[N][REPLAY_WG];
	__shared__ u32 s_heads[REPLAY_TILE];
	__shared__ u32 s_nheads;
	const CtxJob j = jobs[blockIdx.x];
	const u64* src = pool + (j.sorted_in_b ? j.elems_b : j.elems);
	u64* trip = trip_pool + j.trip;
	const u32 n = j.n, tid = threadIdx.x;
	const u32 limit = (1u << 16) - 2u * N;                   // MaxAccumulatedValue (src/SymbolCoderRC.h:67)

	for (u32 tile = 0; tile < n; tile += REPLAY_TILE)
	{
		if (tid == 0) s_nheads = 0;
		__syncthreads();
		for (u32 k = 0; k < 8; ++k)
		{
			const u32 i = tile + k * REPLAY_WG + tid;
			if (i < n && (i == 0 || (src[i] >> ELEM_CTX_SHIFT) != (src[i - 1] >> ELEM_CTX_SHIFT)))
				s_heads[atomicAdd(&s_nheads, 1u)] = i;
		}
		__syncthreads();
		const u32 nh = s_nheads;
		for (u32 h = tid; h < nh; h += REPLAY_WG)
		{
			u32 i = s_heads[h];
			const u64 ctx = src[i] >> ELEM_CTX_SHIFT;
			for (int k = 0; k < N; ++k) s_row[k][tid] = 1;
			u32 total = N;
			for (;;)
			{
				const u64 el = src[i];
				const u32 sym = (u32)(el >> ELEM_SYM_SHIFT) & 0xFFu;
				const u32 t = (u32)el;
				if (total >= limit)                          // Rescale (src/SymbolCoderRC.h:69-90)
				{
					total = 0;
					for (int k = 0; k < N; ++k) { u32 x = s_row[k][tid]; x -= x >> 1; s_row[k][tid] = (u16)x; total += x; }
				}
				u32 cum = 0;
				for (u32 k = 0; k < sym; ++k) cum += s_row[k][tid];
				const u32 f = s_row[sym][tid];
				trip[(u64)t * 64] = ((u64)total << 32) | ((u64)cum << 16) | f;
				s_row[sym][tid] = (u16)(f + 2);
				total += 2;
				++i;
				if (i >= n || (src[i] >> ELEM_CTX_SHIFT) != ctx) break;
			}
		}
		__syncthreads();
	}
}

// ---- range coder: one lane = one stream ---------------------------------------------------------
struct RcChain
{
	u64 trip;          // u64 index of the chain's first triple (stride 64)
	u64 out_words;     // u32 index of the staging stream
	u32 n;
	u32 out_byte0, out_cap;
	u32 blk, is_dna;
	u32 pad;
};

__global__ void __launch_bounds__(64) k_rc(const RcChain* chains, u32 n_chains, const u64* trip_pool, u32* word_pool, BlkState* st)
{
	const u32 id = blockIdx.x * 64 + threadIdx.x;
	const bool live = id < n_chains;
	RcChain c;
	if (live) c = chains[id]; else { c.n = 0; c.trip = 0; c.out_words = 0; c.out_byte0 = 0; c.out_cap = 0; c.blk = 0; c.is_dna = 0; }
	const u32 nmax = wave_max(c.n);
	const u64* trip = trip_pool + c.trip;
	u32* out = word_pool + c.out_words;
	u64 low = 0; u32 range = 0xFFFFFFFFu;
	u32 pos = c.out_byte0;
	const u32 cap = c.out_byte0 + c.out_cap;
	bool ovf = false;
	for (u32 t = 0; t < nmax; ++t)
	{
		if (t < c.n)
		{
			const u64 e = trip[(u64)t * 64];
			const u32 f = (u32)e & 0xFFFFu, cum = (u32)(e >> 16) & 0xFFFFu, tot = (u32)(e >> 32);
			range /= tot;
			low += (u32)(range * cum);
			range *= f;
			while (range <= 0x00FFFFFFu)
			{
				if ((low ^ (low + range)) & 0xFF00000000000000ull)
				{
					const u32 r = (u32)low;
					range = (r | 0x00FFFFFFu) - r;
				}
				if (pos < cap) put_byte(out, pos, (u32)(low >> 56)); else ovf = true;
				++pos;
				low <<= 8; range <<= 8;
			}
		}
	}
	if (live)
	{
		for (u32 k = 0; k < 8; ++k)
		{
			if (pos < cap) put_byte(out, pos, (u32)(low >> 56)); else ovf = true;
			++pos; low <<= 8;
		}
		if (c.is_dna) st[c.blk].dna_bytes = pos; else st[c.blk].qua_bytes = pos;
		if (ovf) atomicOr(&st[c.blk].err, (u32)DSRC_ERR_OUT_OVERFLOW);
	}
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
	if (j.is_dna) { put_byte(out, 0, j.scheme); return; }
	if (!j.translate) return;
	put_byte(out, 0, j.scheme);
	const BlkState* S = &st[j.blk];
	for (u32 k = 0; k < 32; ++k)
	{
		u32 v = 0;
		for (u32 b = 0; b < 8; ++b) v = (v << 1) | (S->q_sym[8 * k + b] != 255 ? 1u : 0u);
		put_byte(out, 1 + k, v);
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
	if (b < n_blocks) st[b].first_bad = 0xFFFFFFFFu;
}
