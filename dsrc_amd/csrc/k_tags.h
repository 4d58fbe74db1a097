// Read-id ("tag") stream:
//   TagAnalyzer::Initialize/Update/FinalizeFieldsStats   src/TagModeler.cpp:159-551
//   TagTokenizerEncoder (field dictionary + payload)      src/TagModeler.cpp:556-884
//   TagRawEncoder (mixed field layout fallback)           src/TagModeler.cpp:1217-1284
//   per-record variable-length bits                       src/BlockCompressor.cpp:458-488
//
// One workgroup owns one block.  The reference's record-by-record state machines are restated
// as reductions / scans over the records:
//   * field tokenisation is per record (one lane walks one title against record 0's separators);
//   * const / numeric / min / max are reductions; record-0 double counting is a +1;
//   * value- and delta-RLE run lists are maximal runs cut every 256 records: run heads by a
//     max-scan, run ends by a suffix min-scan;
//   * payload bits are placed by a prefix scan of per-record bit counts.
#pragma once
#include "k_common.h"
#include "k_parse.h"
#include "k_huff.h"

struct TagPlan     // host -> device after the statistics are known
{
	u64 scr;               // u32 index of the block's tag scratch (zeroed)
	u64 val;               // u32 index: values [num_slot][n_recs]
	u64 rl;                // u16 index: chunk info [2*num_slot + {0:value,1:delta}][n_recs]  (len+1 at chunk starts, else 0)
	u32 rbits_off;         // u32 offset in scratch: per-record payload bit offsets
	u32 hdr_off;           // u32 offset in scratch: dictionary staging bytes (unused)
	u32 raw_hist_off;      // 128-bin title histogram + raw tree workspace
	u32 pad;
};

__device__ __forceinline__ bool tag_is_sep(u32 c)
{
	return c == ' ' || c == '.' || c == '_' || c == ',' || c == '=' || c == ':' || c == '/' || c == '-' || c == '#' || c == 0;
}

// sequential byte access through 8-byte windows (titles are walked once, left to right, by one lane)
typedef u64 __attribute__((aligned(1))) u64_unaligned;
struct TitleReader
{	// 32 bytes per refill: the lanes of a wave read 64 different lines, and with a thousand titles per workgroup in flight a line
	// does not survive in the caches from one 8-byte refill to the next (round 4 PMC: k_tag_scan + k_tag_emit fetched 27 GB per 512
	// blocks for 0.8 GB of titles); four loads issued together find it there
	const u8* p; u32 lim; u64 w[4]; u32 base;
	__device__ __forceinline__ void init(const u8* p_, u32 lim_) { p = p_; lim = lim_; base = 0xFFFFFFE0u; w[0] = w[1] = w[2] = w[3] = 0; }
	__device__ __forceinline__ u32 get(u32 k)
	{
		if (k - base >= 32u)
		{
			base = k & ~31u;
			if (base + 32u <= lim)
			{
#pragma unroll
				for (u32 i = 0; i < 4; ++i) w[i] = *(const u64_unaligned*)(p + base + 8 * i);
			}
			else
			{
#pragma unroll
				for (u32 i = 0; i < 4; ++i)
				{
					w[i] = 0;
					if (base + 8 * i + 8 <= lim) w[i] = *(const u64_unaligned*)(p + base + 8 * i);
					else for (u32 c = 0; base + 8 * i + c < lim && c < 8; ++c) w[i] |= (u64)p[base + 8 * i + c] << (8 * c);
				}
			}
		}
		const u32 o = k - base;
		const u64 lo = (o & 8u) ? w[1] : w[0], hi = (o & 8u) ? w[3] : w[2];
		return (u32)(((o & 16u) ? hi : lo) >> (8 * (o & 7u))) & 0xFFu;
	}
};

// core::is_num (src/utils.h:163-175)
__device__ __forceinline__ bool tag_is_num(const u8* s, u32 len, u32* val)
{
	u32 v = 0, i;
	for (i = 0; i < len; ++i)
	{
		const u32 c = s[i];
		if (c < '0' || c > '9') break;
		v = v * 10u + (c - '0');
	}
	*val = v;
	return i == len && (len == 1 || (len == 0 ? true : s[0] != '0'));
}

// ---- -f: title field filter (FastqParserExt::ReadNextRecord, src/FastqParser.cpp:198-251) -----------------------
// One lane per record rewrites its title IN PLACE, as the reference does: a field ends at a separator (or NUL) or at
// the end of the title and is kept, including that end byte, iff its 1-based number is set in the mask.  The end
// byte of the last field is the line terminator, so a kept last field makes the title one byte longer than the
// text before it was; the unsigned "bytes cut" total then goes down by one (no assert in the release build).
// The chunk text is modified (BlockCompressor::Store destroys its input as well).
__global__ void __launch_bounds__(WG) k_tag_filter(u8* in, const BlkDesc* desc, BlkState* st, RecPools rp, DsrcParams prm)
{
	const u32 b = blockIdx.y;
	const BlkDesc d = desc[b];
	BlkState* S = &st[b];
	u32 n_cand = (S->n_term + 1 + 3) / 4;
	if (n_cand > d.rec_cap) n_cand = d.rec_cap;
	const u32 n_recs = S->first_bad < n_cand ? S->first_bad : n_cand;
	const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_recs) return;
	const u64 g = (u64)d.rec_base + r;
	u8* t = in + d.in_off + rp.title_off[g];
	const u32 tl = rp.title_len[g];
	u32 field_no = 0, begin = 0, bp = 0;
	bool ub = tl > 512;                                      // the reference's scratch buffer
	// pieces move towards the front only (bp <= begin), so the copy can be done in place, front to back
	for (u32 i = 0; i <= tl; ++i)
	{
		if (i != tl && !(tag_is_sep(t[i]) || t[i] == 0)) continue;
		++field_no;
		if (field_no >= 31) ub = true;                       // BIT(x) is a 32-bit shift in the reference
		else if (prm.tag_flags & (1u << field_no))
		{
			const u32 end = (i == tl && (u64)rp.title_off[g] + tl >= d.in_size) ? i : i + 1;     // never touch bytes past the chunk
			for (u32 k = begin; k < end; ++k) t[bp + (k - begin)] = t[k];
			bp += i + 1 - begin;
		}
		begin = i + 1;
	}
	if (bp > 512) ub = true;
	if (ub) { atomicOr(&S->err, (u32)DSRC_ERR_REF_UB); return; }
	rp.title_len[g] = (u16)bp;
	const i32 cut = (i32)tl - (i32)bp;
	if (cut) atomicAdd(&S->title_cut, cut);
}

// A title that swallowed its one-byte terminator ends where the sequence line begins, and the tag tokenizer reads
// the byte after a title as the last field's separator -- after ProcessForward has turned the sequence into base
// indices in place (every base becomes its index, kept ones are then compacted to the front).  Reproduce that byte:
// index of the first kept base, or of base 0 if none is kept.  Runs after every kernel that reads the sequence text.
__global__ void __launch_bounds__(WG) k_tag_poke(u8* in, const BlkDesc* desc, const BlkState* st, RecPools rp, DsrcParams prm)
{
	const u32 b = blockIdx.y;
	const BlkDesc d = desc[b];
	const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= st[b].n_recs) return;
	const u64 g = (u64)d.rec_base + r;
	const u32 so = rp.seq_off[g], len = rp.len[g];
	const u32 tend = rp.title_off[g] + rp.title_len[g];
	if ((!prm.record_layout && tend != so) || len == 0) return;             // the title does not reach the sequence line
	u8* p = in + d.in_off;
	u32 first = 255;
	for (u32 j = 0; j < len && first == 255; ++j)
	{
		u32 sidx; bool keep;
		transform_base(p[so + j], p[rp.qual_off[g] + j], prm.quality_offset, prm.lossy, &sidx, &keep);
		if (keep) first = sidx;
	}
	// record layout (BlockCompressorExt::InsertNewRecord): tag and sequence lie back to back, so the same byte is what
	// follows the title; in our text layout that place is the title's line terminator
	p[prm.record_layout ? tend : so] = (u8)(first != 255 ? first : dna_index(p[so]));
}

// ---- record 0 -> field template -----------------------------------------------------------------
__global__ void __launch_bounds__(64) k_tag_template(const u8* in, const BlkDesc* desc, BlkState* st, RecPools rp, u32 n_blocks)
{
	const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_blocks) return;
	BlkState* S = &st[b];
	const BlkDesc d = desc[b];
	if (S->n_recs == 0) { S->n_fields = 0; S->n_num0 = 0; return; }
	const u8* t = in + d.in_off + rp.title_off[d.rec_base];
	const u32 tl = rp.title_len[d.rec_base];
	u32 nf = 0, start = 0, nnum = 0;
	for (u32 i = 0; i <= tl; ++i)
	{
		if (!tag_is_sep(t[i]) && i != tl) continue;
		if (nf >= DSRC_MAX_FIELDS) { atomicOr(&S->err, (u32)DSRC_ERR_TOO_MANY_FLD); break; }
		TagField* f = &S->fld[nf];
		f->start0 = start; f->len0 = i - start; f->sep = t[i];
		u32 v;
		const bool isn = tag_is_num(t + start, i - start, &v);
		f->isnum0 = isn ? 1 : 0;
		f->num_slot = isn ? (u8)nnum++ : 0;
		f->keep_double = nf >= d.fields_keep_from ? 1 : 0;
		f->min_len = f->max_len = i - start;
		f->not_const = 0; f->not_lenconst = 0; f->not_numeric = isn ? 0 : 1;
		f->min_value = isn ? (i32)v : (1 << 30); f->max_value = isn ? (i32)v : -(1 << 30);
		f->min_delta = 1 << 30; f->max_delta = -(1 << 30);
		f->runs_val = f->last_val_len = f->runs_delta = f->last_delta_len = 0;
		start = i + 1; nf++;
	}
	S->n_fields = nf; S->n_num0 = nnum;
	S->mixed = 0; S->first_mixed = 0xFFFFFFFFu;
	S->min_title = 0xFFFFFFFFu; S->max_title = 0;
}

// ---- all records vs the template (UpdateFieldsStats) ----------------------------------------------
// Grid: x = part, y = block.  In a batch of few, large blocks (-b64, -b256) several workgroups take a block's records (every gridDim.x-th
// group of blockDim.x); what they find is minima, maxima and flags, which meet in the block's state through atomics (k_tag_template
// has set their starting values) -- no workgroup has to come last.
__global__ void __launch_bounds__(WG) k_tag_scan(const u8* in, const BlkDesc* desc, BlkState* st, RecPools rp, u32* val_pool, const TagPlan* plans)
{
	__shared__ u32 s_minlen[DSRC_MAX_FIELDS], s_maxlen[DSRC_MAX_FIELDS];
	__shared__ u32 s_nc[DSRC_MAX_FIELDS], s_nlc[DSRC_MAX_FIELDS], s_nn[DSRC_MAX_FIELDS];
	__shared__ i32 s_minv[DSRC_MAX_FIELDS], s_maxv[DSRC_MAX_FIELDS];
	__shared__ u8 s_sep[DSRC_MAX_FIELDS], s_isnum[DSRC_MAX_FIELDS], s_slot[DSRC_MAX_FIELDS];
	__shared__ u32 s_start0[DSRC_MAX_FIELDS], s_len0[DSRC_MAX_FIELDS];
	__shared__ u32 s_tmin, s_tmax, s_fmix;
	__shared__ u8 s_t0[256];                // head of record 0's title (the field template's text)
	const u32 b = blockIdx.y, parts = gridDim.x;
	BlkState* S = &st[b];
	const BlkDesc d = desc[b];
	const u32 nf = S->n_fields, n = S->n_recs;
	if (n == 0) return;
	for (u32 i = threadIdx.x; i < nf; i += blockDim.x)
	{
		const TagField* f = &S->fld[i];
		s_minlen[i] = f->min_len; s_maxlen[i] = f->max_len; s_nc[i] = 0; s_nlc[i] = 0; s_nn[i] = f->not_numeric;
		s_minv[i] = f->min_value; s_maxv[i] = f->max_value;
		s_sep[i] = f->sep; s_isnum[i] = f->isnum0; s_slot[i] = f->num_slot; s_start0[i] = f->start0; s_len0[i] = f->len0;
	}
	if (threadIdx.x == 0) { s_tmin = 0xFFFFFFFFu; s_tmax = 0; s_fmix = 0xFFFFFFFFu; }
	__syncthreads();
	const u8* base = in + d.in_off;
	const u8* t0 = base + rp.title_off[d.rec_base];
	u32* val_arr = val_pool + plans[b].val;
	for (u32 i = threadIdx.x; i < 256; i += blockDim.x) s_t0[i] = i < rp.title_len[d.rec_base] ? t0[i] : 0;
	__syncthreads();
	for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += blockDim.x * parts)
	{
		const u64 g = (u64)d.rec_base + r;
		const u32 toff = rp.title_off[g];
		const u32 tl = rp.title_len[g];
		atomicMin(&s_tmin, tl); atomicMax(&s_tmax, tl);
		TitleReader tr; tr.init(base + toff, d.in_size - toff);
		u32 c = 0, start = 0, k;
		// one pass: per field accumulate "all digits" / value / first char / equality with record 0
		bool numok = true, same = true; u32 val = 0, first = 0;
		for (k = 0; k <= tl && c < nf; ++k)
		{
			const u32 ch = k < tl ? tr.get(k) : 0;
			if (k < tl && ch != s_sep[c])
			{
				const u32 x = k - start;
				if (x == 0) first = ch;
				if (ch < '0' || ch > '9') numok = false;
				val = val * 10u + (ch - '0');
				if (x >= s_len0[c] || ch != (x + s_start0[c] < 256u ? (u32)s_t0[s_start0[c] + x] : (u32)t0[s_start0[c] + x])) same = false;
				continue;
			}
			const u32 L = k - start;
			if (L > s_maxlen[c]) atomicMax(&s_maxlen[c], L);
			if (L < s_minlen[c]) atomicMin(&s_minlen[c], L);
			if (L != s_len0[c]) { s_nlc[c] = 1; same = false; }
			if (!same) s_nc[c] = 1;
			if (s_isnum[c])
			{
				const bool isn = numok && (L == 1 || L == 0 || first != '0');         // core::is_num
				if (isn)
				{
					const u32 v = L ? val : 0;
					atomicMin(&s_minv[c], (i32)v); atomicMax(&s_maxv[c], (i32)v);
					val_arr[(u64)s_slot[c] * n + r] = v;
				}
				else { s_nn[c] = 1; val_arr[(u64)s_slot[c] * n + r] = 0; }
			}
			start = k + 1; c++;
			numok = true; same = true; val = 0; first = 0;
		}
		if (c != nf || k != tl + 1) atomicMin(&s_fmix, r);
	}
	__syncthreads();
	if (parts > 1)
	{
		for (u32 i = threadIdx.x; i < nf; i += blockDim.x)
		{
			TagField* f = &S->fld[i];
			atomicMin(&f->min_len, s_minlen[i]); atomicMax(&f->max_len, s_maxlen[i]);
			if (s_nc[i]) atomicOr(&f->not_const, 1u);
			if (s_nlc[i]) atomicOr(&f->not_lenconst, 1u);
			if (s_nn[i]) atomicOr(&f->not_numeric, 1u);
			atomicMin(&f->min_value, s_minv[i]); atomicMax(&f->max_value, s_maxv[i]);
		}
		if (threadIdx.x == 0)
		{
			atomicMin(&S->min_title, s_tmin); atomicMax(&S->max_title, s_tmax);
			if (s_fmix != 0xFFFFFFFFu) { atomicMin(&S->first_mixed, s_fmix); atomicOr(&S->mixed, 1u); atomicOr(&S->flags, 4u); }
		}
		return;
	}
	for (u32 i = threadIdx.x; i < nf; i += blockDim.x)
	{
		TagField* f = &S->fld[i];
		f->min_len = s_minlen[i]; f->max_len = s_maxlen[i]; f->not_const = s_nc[i]; f->not_lenconst = s_nlc[i]; f->not_numeric = s_nn[i];
		f->min_value = s_minv[i]; f->max_value = s_maxv[i];
	}
	if (threadIdx.x == 0)
	{
		S->min_title = s_tmin; S->max_title = s_tmax;
		S->first_mixed = s_fmix; S->mixed = s_fmix != 0xFFFFFFFFu ? 1u : 0u;
		if (S->mixed) S->flags |= 4u;                          // FLAG_MIXED_FIELD_FORMATTING
	}
}

// ---- numeric fields: deltas and chunked runs (UpdateNumericField, src/TagModeler.cpp:341-459) ------
// seq(i), i in [i0, n): values (i0 = 0) or deltas (i0 = 1).  A chunk = at most 256 equal consecutive
// elements; out[i] = chunk_length (1..256) at chunk starts, 0 elsewhere.
__device__ inline void tag_chunk_runs(const u32* v, u32 n, bool delta, u16* out, u32* n_chunks, u32* last_len)
{
	__shared__ u32 s_w[WAVES];
	__shared__ u32 s_cnt, s_last;
	const u32 i0 = delta ? 1u : 0u;
	if (threadIdx.x == 0) { s_cnt = 0; s_last = 0; }
	__syncthreads();
	// forward: run starts by max-scan, chunk starts every 256
	u32 carry = 0;                                          // (last head index + 1) so far
	for (u32 base = i0; base < n; base += blockDim.x)
	{
		const u32 i = base + threadIdx.x;
		const bool valid = i < n;
		u32 cur = 0, prv = 0;
		if (valid)
		{
			cur = delta ? v[i] - v[i - 1] : v[i];
			if (i > i0) prv = delta ? v[i - 1] - v[i - 2] : v[i - 1];
		}
		const bool head = valid && (i == i0 || cur != prv);
		u32 m = head ? i + 1 : 0;
		for (u32 dd = 1; dd < 64; dd <<= 1) { const u32 o = __shfl_up(m, dd); if (lane_id() >= dd && o > m) m = o; }
		if (lane_id() == 63) s_w[wave_id()] = m;
		__syncthreads();
		u32 pre = carry, all = carry;
		for (u32 w = 0; w < (blockDim.x >> 6); ++w) { const u32 x = s_w[w]; if (w < wave_id() && x > pre) pre = x; if (x > all) all = x; }
		if (pre > m) m = pre;
		const bool chunk = valid && (((i - (m - 1)) & 255u) == 0);
		if (valid) out[i] = chunk ? 1 : 0;
		if (chunk) { atomicAdd(&s_cnt, 1u); atomicMax(&s_last, i + 1); }
		carry = all;
		__syncthreads();
	}
	// backward: next chunk start by suffix min-scan (mirrored index)
	const u32 cnt_n = n - i0;
	u32 carry_next = n;                                      // nearest chunk start to the right so far
	for (u32 base = 0; base < cnt_n; base += blockDim.x)
	{
		const u32 k = base + threadIdx.x;                    // mirrored
		const bool valid = k < cnt_n;
		const u32 i = valid ? n - 1 - k : 0;
		const bool chunk = valid && out[i] != 0;
		u32 m = chunk ? i : 0xFFFFFFFFu;                     // exclusive suffix-min: shift by one lane first
		u32 ex = __shfl_up(m, 1); if (lane_id() == 0) ex = 0xFFFFFFFFu;
		for (u32 dd = 1; dd < 64; dd <<= 1) { const u32 o = __shfl_up(ex, dd); if (lane_id() >= dd && o < ex) ex = o; }
		u32 incl = ex < m ? ex : m;
		if (lane_id() == 63) s_w[wave_id()] = incl;
		__syncthreads();
		u32 pre = carry_next, all = carry_next;
		for (u32 w = 0; w < (blockDim.x >> 6); ++w) { const u32 x = s_w[w]; if (w < wave_id() && x < pre) pre = x; if (x < all) all = x; }
		const u32 nxt = ex < pre ? ex : pre;                 // nearest chunk start strictly right of i
		if (chunk) out[i] = (u16)(nxt - i);
		carry_next = all;
		__syncthreads();
	}
	*n_chunks = s_cnt;
	*last_len = s_cnt ? (n - (s_last - 1)) - 1 : 0;
	__syncthreads();
}

__global__ void __launch_bounds__(WG) k_tag_numeric(const BlkDesc* desc, BlkState* st, u32* val_pool, u16* rl_pool, const TagPlan* plans)
{
	__shared__ i32 s_mind, s_maxd;
	const u32 b = blockIdx.x;
	BlkState* S = &st[b];
	const u32 nf = S->n_fields, n = S->n_recs;
	if (n == 0 || S->mixed) return;
	const TagPlan pl = plans[b];
	for (u32 fi = 0; fi < nf; ++fi)
	{
		TagField* f = &S->fld[fi];
		if (!f->isnum0 || f->not_numeric || !f->not_const) continue;     // uniform over the workgroup
		const u32* v = val_pool + pl.val + (u64)f->num_slot * n;
		u16* rv = rl_pool + pl.rl + (u64)(2 * f->num_slot) * n;
		u16* rd = rv + n;
		if (threadIdx.x == 0) { s_mind = 0x7FFFFFFF; s_maxd = (i32)0x80000000; }
		__syncthreads();
		i32 mn = 0x7FFFFFFF, mx = (i32)0x80000000;
		for (u32 r = 1 + threadIdx.x; r < n; r += blockDim.x)
		{
			const i32 dlt = (i32)(v[r] - v[r - 1]);
			mn = dlt < mn ? dlt : mn; mx = dlt > mx ? dlt : mx;
		}
		atomicMin(&s_mind, mn); atomicMax(&s_maxd, mx);
		__syncthreads();
		u32 rc, ll;
		tag_chunk_runs(v, n, false, rv, &rc, &ll);
		if (threadIdx.x == 0) { f->runs_val = rc; f->last_val_len = ll; }
		tag_chunk_runs(v, n, true, rd, &rc, &ll);
		if (threadIdx.x == 0)
		{
			f->runs_delta = rc; f->last_delta_len = ll;
			if (n >= 2) { f->min_delta = s_mind; f->max_delta = s_maxd; }
		}
		__syncthreads();
	}
}

// ---- FinalizeFieldsStats (src/TagModeler.cpp:461-551), one lane per field ---------------------------
__global__ void __launch_bounds__(64) k_tag_finalize(BlkState* st)
{
	BlkState* S = &st[blockIdx.x];
	const u32 n = S->n_recs;
	const u32 fi = threadIdx.x;
	if (n == 0 || fi >= S->n_fields) return;
	TagField* f = &S->fld[fi];
	f->is_constant = f->not_const ? 0 : 1;
	f->is_len_constant = f->not_lenconst ? 0 : 1;
	f->is_numeric = (f->isnum0 && !f->not_numeric) ? 1 : 0;
	f->is_string = (!f->is_constant && !f->is_numeric) ? 1 : 0;
	f->scheme = NS_NONE; f->var_stat_encode = 0; f->bits_num = f->bits_value = f->bits_len = 0;
	if (S->mixed) return;
	if (!f->is_numeric)
	{
		if (!f->is_constant) f->bits_len = bit_length32(f->max_len - f->min_len);
		return;
	}
	if (f->is_constant) return;                               // stored verbatim in the dictionary
	const i32 dv = (i32)((u32)f->max_value - (u32)f->min_value);
	const i32 dd = (i32)((u32)f->max_delta - (u32)f->min_delta);
	bool delta_coding; i32 diff;
	if (dv < dd) { delta_coding = false; diff = dv; } else { delta_coding = true; diff = dd; }
	// rle.run_len after FinalizeFieldsStats = chunks - 1 (+1 if the last chunk has repeats)
	const u32 run_val = f->runs_val - 1 + (f->last_val_len > 0 ? 1u : 0u);
	const bool try_val = (float)n / (float)run_val > 1.25f;
	bool delta_const = false, try_delta = false;
	if (delta_coding)
	{
		delta_const = diff == 0;
		if (!delta_const)
		{
			const u32 run_d = f->runs_delta ? f->runs_delta - 1 + (f->last_delta_len > 0 ? 1u : 0u) : 0u;
			try_delta = (float)n / (float)run_d > 1.25f;
		}
	}
	if (delta_coding && delta_const) f->scheme = NS_DELTA_CONST;
	else if (delta_coding && try_delta) f->scheme = NS_DELTA_RLE;
	else if (try_val) f->scheme = NS_VALUE_RLE;
	else if (delta_coding) { f->scheme = NS_DELTA_VAR; f->var_stat_encode = ((u32)dd + 1u <= 512u) ? 1 : 0; }
	else { f->scheme = NS_VALUE_VAR; f->var_stat_encode = ((u32)dv + 1u <= 512u) ? 1 : 0; }
	f->bits_num = bit_length32((u64)(i64)diff);
	f->bits_value = bit_length32((u64)(i64)dv);
}

// per-field resources inside the tag scratch (host fills these after k_tag_finalize)
struct TagFieldRes
{
	u32 hist_off;      // string: [129][256] ; numeric var: [512]
	u32 code_off, len_off;   // string: [129][256] each ; numeric: [512] each
	u32 tree_off;      // bytes: slots of tree_slot bytes, first word = size
	u32 tree_slot;
	u32 ws_off, ws_slot;
	u32 ham_off;       // bytes (as u32 words, 1 = equal to record 0 at that position in every record)
};

// ---- histograms for Huffman-coded fields -----------------------------------------------------------
__global__ void __launch_bounds__(WG) k_tag_hist(const u8* in, const BlkDesc* desc, BlkState* st, RecPools rp, u32* val_pool,
												 u32* scr_pool, const TagPlan* plans, const TagFieldRes* res_all)
{
	const u32 b = blockIdx.x;
	BlkState* S = &st[b];
	const u32 nf = S->n_fields, n = S->n_recs;
	if (n == 0 || S->mixed) return;
	const BlkDesc d = desc[b];
	const TagPlan pl = plans[b];
	const TagFieldRes* res = res_all + (u64)b * DSRC_MAX_FIELDS;
	u32* scr = scr_pool + pl.scr;
	const u8* base = in + d.in_off;
	const u8* t0 = base + rp.title_off[d.rec_base];
	bool any_string = false;
	for (u32 fi = 0; fi < nf; ++fi)
	{
		const TagField* f = &S->fld[fi];
		if (f->is_string) any_string = true;
		if (f->is_numeric && !f->is_constant && f->var_stat_encode)
		{
			const u32* v = val_pool + pl.val + (u64)f->num_slot * n;
			u32* h = scr + res[fi].hist_off;
			if (f->scheme == NS_VALUE_VAR)
			{
				for (u32 r = threadIdx.x; r < n; r += blockDim.x) atomicAdd(&h[v[r] - (u32)f->min_value], 1u);
				if (threadIdx.x == 0 && fi >= d.fields_keep_from) atomicAdd(&h[v[0] - (u32)f->min_value], 1u);   // record 0 counted twice
			}
			else
				for (u32 r = 1 + threadIdx.x; r < n; r += blockDim.x) atomicAdd(&h[v[r] - v[r - 1] - (u32)f->min_delta], 1u);
		}
	}
	if (!any_string) return;
	for (u32 r = threadIdx.x; r < n; r += blockDim.x)
	{
		const u64 g = (u64)d.rec_base + r;
		const u8* t = base + rp.title_off[g];
		const u32 tl = rp.title_len[g];
		u32 c = 0, start = 0;
		for (u32 k = 0; k <= tl && c < nf; ++k)
		{
			const TagField* f = &S->fld[c];
			if (k < tl && t[k] != f->sep) continue;
			if (f->is_string)
			{
				const u32 L = k - start;
				u32* h = scr + res[c].hist_off;
				u32* ham = scr + res[c].ham_off;
				for (u32 x = 0; x < L; ++x)
				{
					const u32 ch = t[start + x];
					atomicAdd(&h[(x < 128 ? x : 128u) * 256u + ch], 1u);
					if (x < f->len0 && ch != t0[f->start0 + x]) ham[x] = 1;      // 1 = differs somewhere
				}
			}
			start = k + 1; c++;
		}
	}
}

// one lane per tree.  Tree index space per block: field fi, sub-index j (string: position 0..128; numeric: 0)
__global__ void __launch_bounds__(64) k_tag_trees(BlkState* st, u32* scr_pool, const TagPlan* plans, const TagFieldRes* res_all)
{
	const u32 b = blockIdx.y;
	BlkState* S = &st[b];
	if (S->n_recs == 0 || S->mixed) return;
	const u32 id = blockIdx.x * blockDim.x + threadIdx.x;
	const u32 fi = id / DSRC_MAX_STRF, j = id % DSRC_MAX_STRF;
	if (fi >= S->n_fields) return;
	const TagField* f = &S->fld[fi];
	const TagFieldRes rs = res_all[(u64)b * DSRC_MAX_FIELDS + fi];
	u32* scr = scr_pool + plans[b].scr;
	if (f->is_string)
	{
		const u32 lim = f->max_len < 128 ? f->max_len : 128;
		const u32* ham = scr + rs.ham_off;
		const bool need = (j < lim && (j >= f->len0 || ham[j])) || (j == 128 && f->max_len >= 128);
		u8* tr = (u8*)scr + rs.tree_off + (u64)j * rs.tree_slot;
		if (!need) { *(u32*)tr = 0; return; }
		u32* ws = scr + rs.ws_off + (u64)j * rs.ws_slot;
		HuffView h = huff_build(scr + rs.hist_off + (u64)j * 256, 1, 256, ws, &S->err);
		for (u32 x = 0; x < 256; ++x) { scr[rs.code_off + (u64)j * 256 + x] = h.code[x]; scr[rs.len_off + (u64)j * 256 + x] = h.len[x]; }
		*(u32*)tr = huff_store(h, tr + 4, ws);
	}
	else if (f->is_numeric && !f->is_constant && f->var_stat_encode && j == 0)
	{
		const u32 nsym = (f->scheme == NS_DELTA_VAR ? (u32)f->max_delta - (u32)f->min_delta : (u32)f->max_value - (u32)f->min_value) + 1u;
		u32* ws = scr + rs.ws_off;
		HuffView h = huff_build(scr + rs.hist_off, 1, nsym, ws, &S->err);
		for (u32 x = 0; x < nsym; ++x) { scr[rs.code_off + x] = h.code[x]; scr[rs.len_off + x] = h.len[x]; }
		u8* tr = (u8*)scr + rs.tree_off;
		*(u32*)tr = huff_store(h, tr + 4, ws);
	}
}

// what the per-record payload writer needs of a field, staged in LDS (BlkState::fld lives in HBM)
struct TagFieldLite
{
	u32 min_value, min_delta, min_len, len0;
	u32 bits_num, bits_value, bits_len;
	u32 code_off, len_off, ham_off;
	u8 sep, is_constant, is_numeric, is_len_constant, scheme, var_stat_encode, num_slot, pad;
};

__device__ __forceinline__ void tag_lite_load(TagFieldLite* dst, const BlkState* S, const TagFieldRes* res, u32 nf)
{
	for (u32 i = threadIdx.x; i < nf; i += blockDim.x)
	{
		const TagField* f = &S->fld[i];
		TagFieldLite l;
		l.min_value = (u32)f->min_value; l.min_delta = (u32)f->min_delta; l.min_len = f->min_len; l.len0 = f->len0;
		l.bits_num = f->bits_num; l.bits_value = f->bits_value; l.bits_len = f->bits_len;
		l.code_off = res[i].code_off; l.len_off = res[i].len_off; l.ham_off = res[i].ham_off;
		l.sep = f->sep; l.is_constant = f->is_constant; l.is_numeric = f->is_numeric; l.is_len_constant = f->is_len_constant;
		l.scheme = f->scheme; l.var_stat_encode = f->var_stat_encode; l.num_slot = f->num_slot; l.pad = 0;
		dst[i] = l;
	}
}

// bits of one numeric field of record r (StoreNumericField, src/TagModeler.cpp:753-874); emits when out != 0
__device__ __forceinline__ u32 tag_numeric_bits(const TagFieldLite& f, const u32* scr, const u32* v, const u16* rv, const u16* rd,
												u32 r, u32* out, u64 at)
{
	const u32 cur = v[r];
	if (r == 0)
	{
		u32 bits = f.bits_value;
		if (out) put_bits(out, at, cur - f.min_value, f.bits_value);
		if (f.scheme == NS_VALUE_RLE) { if (out) put_bits(out, at + bits, (u32)rv[0] - 1u, 8); bits += 8; }
		return bits;
	}
	switch (f.scheme)
	{
	case NS_DELTA_RLE:
		if (rd[r])
		{
			if (out) { put_bits(out, at, cur - v[r - 1] - f.min_delta, f.bits_num); put_bits(out, at + f.bits_num, (u32)rd[r] - 1u, 8); }
			return f.bits_num + 8;
		}
		return 0;
	case NS_VALUE_RLE:
		if (rv[r])
		{
			if (out) { put_bits(out, at, cur - f.min_value, f.bits_value); put_bits(out, at + f.bits_value, (u32)rv[r] - 1u, 8); }
			return f.bits_value + 8;
		}
		return 0;
	case NS_DELTA_VAR:
	case NS_VALUE_VAR:
	{
		const u32 x = f.scheme == NS_DELTA_VAR ? cur - v[r - 1] - f.min_delta : cur - f.min_value;
		if (f.var_stat_encode)
		{
			const u32 len = scr[f.len_off + x];
			if (out) put_bits(out, at, scr[f.code_off + x], len);
			return len;
		}
		if (out) put_bits(out, at, x, f.bits_num);
		return f.bits_num;
	}
	default: return 0;   // DeltaConst
	}
}

// payload bits of record r: walks the title once; emits when out != 0
__device__ inline u32 tag_record_bits(const TagFieldLite* fl, u32 nf, const u32* scr, const u8* t, u32 tlim, u32 tl, u32 n, u32 r,
									  const u32* val, const u16* rl, u32 len_bits, u32 qlen_minus_min, u32* out, u64 at0)
{
	TitleReader tr; tr.init(t, tlim);
	u32 c = 0, start = 0; u64 at = at0;
	for (u32 k = 0; k <= tl && c < nf; ++k)
	{
		const TagFieldLite& f = fl[c];
		const u32 ch = k < tl ? tr.get(k) : 0;
		if (k < tl && ch != f.sep) continue;
		if (!f.is_constant)
		{
			if (f.is_numeric)
			{
				const u32* v = val + (u64)f.num_slot * n;
				const u16* rv = rl + (u64)(2 * f.num_slot) * n;
				at += tag_numeric_bits(f, scr, v, rv, rv + n, r, out, at);
			}
			else
			{
				const u32 L = k - start;
				if (!f.is_len_constant) { if (out) put_bits(out, at, L - f.min_len, f.bits_len); at += f.bits_len; }
				const u32* ham = scr + f.ham_off;
				for (u32 x = 0; x < L; ++x)
				{
					if (x >= f.len0 || ham[x])
					{
						const u32 ix = (x < 128 ? x : 128u) * 256u + t[start + x];
						const u32 len = scr[f.len_off + ix];
						if (out) put_bits(out, at, scr[f.code_off + ix], len);
						at += len;
					}
				}
			}
		}
		start = k + 1; c++;
	}
	if (len_bits) { if (out) put_bits(out, at, qlen_minus_min, len_bits); at += len_bits; }
	return (u32)(at - at0);
}

// ---- dictionary + payload (StoreFields :569-693, EncodeNextFields :695-751) -----------------------------
__global__ void __launch_bounds__(WG) k_tag_emit(const u8* in, const BlkDesc* desc, BlkState* st, RecPools rp, u32* val_pool, u16* rl_pool,
												 u32* scr_pool, u32* word_pool, const TagPlan* plans, const TagFieldRes* res_all)
{
	__shared__ u32 s_hdr;
	__shared__ u64 s_total;
	__shared__ TagFieldLite s_fl[DSRC_MAX_FIELDS];
	const u32 b = blockIdx.x;
	BlkState* S = &st[b];
	const u32 nf = S->n_fields, n = S->n_recs;
	if (n == 0 || S->mixed) return;
	const BlkDesc d = desc[b];
	const TagPlan pl = plans[b];
	const TagFieldRes* res = res_all + (u64)b * DSRC_MAX_FIELDS;
	u32* scr = scr_pool + pl.scr;
	u32* out = word_pool + d.tag_out;
	tag_lite_load(s_fl, S, res, nf);
	const u8* base = in + d.in_off;
	const u8* t0 = base + rp.title_off[d.rec_base];
	if (threadIdx.x == 0)
	{
		u32 at = 0;
		put_byte(out, at++, nf & 0xFF);
		for (u32 fi = 0; fi < nf; ++fi)
		{
			const TagField* f = &S->fld[fi];
			put_byte(out, at++, f->sep);
			put_byte(out, at++, f->is_constant);
			if (f->is_constant)
			{
				put_be32(out, at, f->len0); at += 4;
				for (u32 x = 0; x < f->len0; ++x) put_byte(out, at++, t0[f->start0 + x]);
				continue;
			}
			put_byte(out, at++, f->is_numeric);
			if (f->is_numeric)
			{
				put_byte(out, at++, f->scheme);
				put_be32(out, at, (u32)f->min_value); at += 4; put_be32(out, at, (u32)f->max_value); at += 4;
				if (f->scheme == NS_DELTA_CONST || f->scheme == NS_DELTA_RLE || f->scheme == NS_DELTA_VAR)
				{
					put_be32(out, at, (u32)f->min_delta); at += 4; put_be32(out, at, (u32)f->max_delta); at += 4;
				}
				if (f->scheme == NS_DELTA_VAR || f->scheme == NS_VALUE_VAR)
				{
					put_byte(out, at++, f->var_stat_encode);
					if (f->var_stat_encode)
					{
						const u8* tr = (const u8*)scr + res[fi].tree_off;
						const u32 tb = *(const u32*)tr;
						stage_bytes(out, at, tr + 4, tb); at += tb;
					}
				}
				continue;
			}
			put_byte(out, at++, f->is_len_constant);
			put_be32(out, at, f->len0); at += 4; put_be32(out, at, f->max_len); at += 4; put_be32(out, at, f->min_len); at += 4;
			for (u32 x = 0; x < f->len0; ++x) put_byte(out, at++, t0[f->start0 + x]);
			const u32* ham = scr + res[fi].ham_off;
			for (u32 x = 0; x < f->len0; x += 8)
			{
				u32 v = 0;
				for (u32 i = 0; i < 8; ++i) v = (v << 1) | ((x + i < f->len0 && !ham[x + i]) ? 1u : 0u);
				put_byte(out, at++, v);
			}
			for (u32 j = 0; j < DSRC_MAX_STRF; ++j)
			{
				const u8* tr = (const u8*)scr + res[fi].tree_off + (u64)j * res[fi].tree_slot;
				const u32 tb = *(const u32*)tr;
				if (tb) { stage_bytes(out, at, tr + 4, tb); at += tb; }
			}
		}
		s_hdr = at;
		S->tag_hdr_bytes = at;
	}
	__syncthreads();
	const u32 len_bits = bit_length32((u64)(u16)((u16)S->max_len - (u16)S->min_len));
	const u32 minq = (u16)(S->min_len - S->cs_reduced);     // colour space: rp.len is the shortened length (k_cs_reduce)
	const u32* val = val_pool + pl.val;
	const u16* rl = rl_pool + pl.rl;
	u32* rbits = scr + pl.rbits_off;
	for (u32 r = threadIdx.x; r < n; r += blockDim.x)
	{
		const u64 g = (u64)d.rec_base + r;
		rbits[r] = tag_record_bits(s_fl, nf, scr, base + rp.title_off[g], d.in_size - rp.title_off[g], rp.title_len[g], n, r, val, rl, len_bits, rp.len[g] - minq, 0, 0);
	}
	__syncthreads();
	const u64 carry = (u64)s_hdr * 8;
	{
		u64 run = carry;
		for (u32 b0 = 0; b0 < n; b0 += blockDim.x)
		{
			const u32 r = b0 + threadIdx.x;
			const u32 v = r < n ? rbits[r] : 0;
			u32 tot;
			const u32 ex = block_excl_scan(v, &tot);
			if (r < n) rbits[r] = (u32)(run - carry) + ex;
			run += tot;
		}
		if (threadIdx.x == 0) s_total = run;
	}
	__syncthreads();
	for (u32 r = threadIdx.x; r < n; r += blockDim.x)
	{
		const u64 g = (u64)d.rec_base + r;
		tag_record_bits(s_fl, nf, scr, base + rp.title_off[g], d.in_size - rp.title_off[g], rp.title_len[g], n, r, val, rl, len_bits, rp.len[g] - minq, out, carry + rbits[r]);
	}
	if (threadIdx.x == 0) S->tag_bytes = (u32)((s_total + 7) / 8);
}

// ---- raw fallback (TagRawEncoder, src/TagModeler.cpp:1217-1284) ------------------------------------------
__global__ void __launch_bounds__(WG) k_tag_raw(const u8* in, const BlkDesc* desc, BlkState* st, RecPools rp, u32* scr_pool, u32* word_pool, const TagPlan* plans)
{
	__shared__ u32 s_hist[128];
	__shared__ u32 s_code[128], s_len[128];
	__shared__ u32 s_hdr;
	__shared__ u64 s_total;
	const u32 b = blockIdx.x;
	BlkState* S = &st[b];
	const u32 n = S->n_recs;
	if (n == 0 || !S->mixed) return;
	const BlkDesc d = desc[b];
	const TagPlan pl = plans[b];
	u32* scr = scr_pool + pl.scr;
	u32* out = word_pool + d.tag_out;
	const u8* base = in + d.in_off;
	if (threadIdx.x < 128) s_hist[threadIdx.x] = 0;
	__syncthreads();
	// symbolFreqs: every title char, record 0 twice, minus the unvisited tail of the first mixed record (Appendix B.7/B.10)
	for (u32 r = threadIdx.x; r < n; r += blockDim.x)
	{
		const u64 g = (u64)d.rec_base + r;
		const u8* t = base + rp.title_off[g];
		const u32 tl = rp.title_len[g];
		u32 upto = tl;
		if (r == S->first_mixed)
		{
			u32 c = 0, k;
			for (k = 0; k <= tl && c < S->n_fields; ++k)
			{
				if (k < tl && t[k] != S->fld[c].sep) continue;
				c++;
			}
			upto = k < tl ? k : tl;
		}
		for (u32 x = 0; x < upto; ++x) atomicAdd(&s_hist[t[x] & 127u], 1u);
		if (r == 0) for (u32 x = 0; x < tl; ++x) atomicAdd(&s_hist[t[x] & 127u], 1u);
	}
	__syncthreads();
	const u32 tl_bits = bit_length32(S->max_title - S->min_title);
	if (threadIdx.x == 0)
	{
		u32* fr = scr + pl.raw_hist_off;          // compacted frequencies
		u32 ns = 0; u32 rank[128];
		for (u32 i = 0; i < 128; ++i) { rank[i] = 0xFFFFFFFFu; if (s_hist[i]) { rank[i] = ns; fr[ns++] = s_hist[i]; } }
		u32* ws = fr + 128;
		HuffView h = huff_build(fr, 1, ns, ws, &S->err);
		for (u32 i = 0; i < 128; ++i) { s_code[i] = rank[i] != 0xFFFFFFFFu ? h.code[rank[i]] : 0; s_len[i] = rank[i] != 0xFFFFFFFFu ? h.len[rank[i]] : 0; }
		u32 at = 0;
		put_be32(out, at, S->min_title); at += 4; put_be32(out, at, S->max_title); at += 4;
		for (u32 k = 0; k < 16; ++k)
		{
			u32 v = 0;
			for (u32 i = 0; i < 8; ++i) v = (v << 1) | (s_hist[8 * k + i] ? 1u : 0u);
			put_byte(out, at++, v);
		}
		u8* tr = (u8*)(ws + huff_ws_words(128));
		const u32 tb = huff_store(h, tr, ws);
		stage_bytes(out, at, tr, tb); at += tb;
		s_hdr = at; S->tag_hdr_bytes = at;
	}
	__syncthreads();
	const u32 len_bits = bit_length32((u64)(u16)((u16)S->max_len - (u16)S->min_len));
	const u32 minq = (u16)(S->min_len - S->cs_reduced);     // colour space: rp.len is the shortened length (k_cs_reduce)
	u32* rbits = scr + pl.rbits_off;
	for (u32 r = threadIdx.x; r < n; r += blockDim.x)
	{
		const u64 g = (u64)d.rec_base + r;
		const u8* t = base + rp.title_off[g];
		const u32 tl = rp.title_len[g];
		u32 bits = tl_bits + len_bits;
		for (u32 x = 0; x < tl; ++x) bits += s_len[t[x] & 127u];
		rbits[r] = bits;
	}
	__syncthreads();
	const u64 carry = (u64)s_hdr * 8;
	{
		u64 run = carry;
		for (u32 b0 = 0; b0 < n; b0 += blockDim.x)
		{
			const u32 r = b0 + threadIdx.x;
			const u32 v = r < n ? rbits[r] : 0;
			u32 tot;
			const u32 ex = block_excl_scan(v, &tot);
			if (r < n) rbits[r] = (u32)(run - carry) + ex;
			run += tot;
		}
		if (threadIdx.x == 0) s_total = run;
	}
	__syncthreads();
	for (u32 r = threadIdx.x; r < n; r += blockDim.x)
	{
		const u64 g = (u64)d.rec_base + r;
		const u8* t = base + rp.title_off[g];
		const u32 tl = rp.title_len[g];
		u64 at = carry + rbits[r];
		if (tl_bits) { put_bits(out, at, tl - S->min_title, tl_bits); at += tl_bits; }
		for (u32 x = 0; x < tl; ++x) { const u32 c = t[x] & 127u; put_bits(out, at, s_code[c], s_len[c]); at += s_len[c]; }
		if (len_bits) put_bits(out, at, rp.len[g] - minq, len_bits);
	}
	if (threadIdx.x == 0) S->tag_bytes = (u32)((s_total + 7) / 8);
}
