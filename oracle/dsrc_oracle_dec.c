/*
 * TEST INFRASTRUCTURE ONLY -- the parity oracle, decoding half (see dsrc_oracle.h).
 *
 * Plain-C restatement of the DSRC 2 block decompressor: one block in -> the FASTQ text of the
 * chunk out.  Reference: BlockCompressor::Read / ReadRecords / ReadMetaData / ReadTags
 * (src/BlockCompressor.cpp:262-356,491-573), VerifyChecksum (:576-594), the modelers' Decode
 * methods, RangeDecoder (src/RangeCoder.h:90-142), HuffmanEncoder::LoadTree/Decode
 * (src/huffman.cpp:225-291, src/huffman.h:110-177) and ProcessBackward
 * (src/RecordsProcessor.cpp:269-315,410-454).  The architecture is our own (a pure bit source,
 * a flat record table, four stream decoders); the behaviour is the reference's.
 *
 * Parity: PINNED against oracle/_ref (ref_decompress_block, the unmodified reference) by
 * tests/test_oracle_decode.py and against the block vectors under tests/golden.
 */
#define _GNU_SOURCE
#include "dsrc_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;

#define MINV(a, b) ((a) <= (b) ? (a) : (b))

/* ------------------------------------------------------------------------
 * bit source: BitMemoryReader (src/BitMemory.h:29-213).  GetBit/Get2Bits/GetBits all take bits
 * MSB-first out of one byte-wide buffer; GetByte/GetWord/GetBytes read memory[position] directly
 * and leave the buffered bits alone, FlushInputWordBuffer drops them.
 * ---------------------------------------------------------------------- */
typedef struct
{
	const u8* mem; u64 size, pos;
	u32 buf, nb;
	int err;                 /* read past the end of the block: the stream is corrupt (reference: ASSERT / UB) */
} br_t;

static u32 br_byte(br_t* r)
{
	if (r->pos >= r->size) { r->err = 1; return 0; }
	return r->mem[r->pos++];
}
static u32 br_bit(br_t* r)
{
	if (r->nb == 0) { r->buf = br_byte(r); r->nb = 8; }
	return (r->buf >> --r->nb) & 1;
}
static u32 br_bits(br_t* r, u32 n)      /* GetBits: n == 0 reads nothing (src/BitMemory.h:93-123) */
{
	u32 v = 0;
	while (n--) v = (v << 1) | br_bit(r);
	return v;
}
static u32 br_word(br_t* r) { u32 v = br_byte(r); v = (v << 8) | br_byte(r); v = (v << 8) | br_byte(r); return (v << 8) | br_byte(r); }
static void br_align(br_t* r) { r->nb = 0; }

static u32 bit_length(u64 x)             /* src/utils.h:181-189 */
{
	for (u32 i = 0; i < 32; ++i)
		if (x < (1ull << i)) return i;
	return 64;
}
static u32 int_log2(u32 x) { u32 r = 0; for (u64 t = 2; t <= x; t *= 2) ++r; return r; }

static __thread int g_dec_ub;            /* the block drives the reference decoder into undefined behaviour */

/* ------------------------------------------------------------------------
 * Huffman: LoadTree + Decode (src/huffman.cpp:225-262, src/huffman.h:110-177).  The stored tree
 * is a preorder walk (0 = internal, 1 + id = leaf); decoding a symbol is a walk from the root,
 * which is what GetBits(min_len) + DecodeFast + Decode(bit)... amounts to for any stream the
 * encoder can produce (min_len never exceeds the shortest code).
 * child values: >= 0 internal node, < 0 leaf holding symbol ~v.
 * ---------------------------------------------------------------------- */
typedef struct { u32 n_sym, n_nodes, cap; i32* kid; } hdec_t;     /* kid[2*i], kid[2*i+1] */

static i32 hdec_parse(hdec_t* h, br_t* r, u32 bits_per_id, u32 depth)
{
	if (r->err || depth > 1100) { r->err = 1; return -1; }
	if (br_bit(r)) return ~(i32)br_bits(r, bits_per_id);
	if (h->n_nodes >= h->cap) { r->err = 1; return -1; }
	const u32 id = h->n_nodes++;
	const i32 l = hdec_parse(h, r, bits_per_id, depth + 1);
	const i32 rr = hdec_parse(h, r, bits_per_id, depth + 1);
	h->kid[2 * id] = l; h->kid[2 * id + 1] = rr;
	return (i32)id;
}

static void hdec_load(hdec_t* h, br_t* r)
{
	br_align(r);
	const u64 begin = r->pos;
	const u32 mem_size = br_word(r);
	const u32 root_id = br_word(r);
	const u32 n = br_word(r);
	(void)br_byte(r);                                  /* min_len: only sizes the reference's speed-up table */
	memset(h, 0, sizeof(*h));
	if (n < 2 || n >= 1024 || root_id < n || root_id > 2 * n - 2) { r->err = 1; return; }
	u32 bits_per_id = int_log2(n);
	if (n & (n - 1)) bits_per_id++;
	h->n_sym = n; h->cap = root_id - n + 1;            /* internal nodes: root_id - n + 1 (ids root..1 in the reference) */
	h->kid = (i32*)calloc(2 * (size_t)h->cap + 2, sizeof(i32));
	const i32 root = hdec_parse(h, r, bits_per_id, 0);
	if (root != 0) r->err = 1;                         /* a leaf as root leaves the reference's tree[] uninitialised */
	br_align(r);
	if (begin + mem_size != r->pos) r->err = 1;        /* ASSERT(memBegin + memSize == Position()) */
}

static void hdec_free(hdec_t* h) { free(h->kid); h->kid = NULL; }

static u32 hdec_sym(const hdec_t* h, br_t* r)
{
	i32 v = 0;
	if (!h->kid) { r->err = 1; return 0; }
	for (u32 guard = 0; guard < 64; ++guard)
	{
		v = h->kid[2 * v + (i32)br_bit(r)];
		if (v < 0) return (u32)~v;
		if (r->err) break;
	}
	r->err = 1;
	return 0;
}

/* ------------------------------------------------------------------------
 * Range decoder + adaptive rows: RangeDecoder (src/RangeCoder.h:90-142),
 * TSymbolCoderRC<N>::DecodeSymbol (src/SymbolCoderRC.h:50-64,66-91)
 * ---------------------------------------------------------------------- */
typedef struct { u64 low, buffer; u32 range; br_t* r; } rd_t;

static void rd_start(rd_t* d, br_t* r)
{
	d->r = r; d->buffer = 0;
	for (u32 i = 1; i <= 8; ++i) d->buffer |= (u64)br_byte(r) << (64 - i * 8);
	d->low = 0; d->range = 0xFFFFFFFFu;
}

static u32 row_decode(u16* row, u32 n, rd_t* d)
{
	u32 acc = 0;
	for (u32 i = 0; i < n; ++i) acc += row[i];
	if (acc >= (1u << 16) - n * 2)
	{
		acc = 0;
		for (u32 i = 0; i < n; ++i) { row[i] -= row[i] >> 1; acc += row[i]; }
	}
	d->range /= acc;
	if (d->range == 0) { d->r->err = 1; return 0; }
	const u32 cul = (u32)(d->buffer / d->range);          /* Freq is uint32: the quotient is truncated */
	u32 idx = 0, hi = 0;
	for (;;)
	{
		hi += row[idx];
		if (hi > cul) break;
		if (++idx == n) { d->r->err = 1; return 0; }      /* the reference walks off the row here */
	}
	hi -= row[idx];
	{
		const u32 rr = hi * d->range;                     /* uint32 product */
		d->buffer -= rr; d->low += rr;
		d->range *= row[idx];
		while (d->range <= 0x00FFFFFFu)
		{
			if ((d->low ^ (d->low + d->range)) & 0xFF00000000000000ull)
			{
				const u32 lo = (u32)d->low;
				d->range = (lo | 0x00FFFFFFu) - lo;
			}
			d->buffer = (d->buffer << 8) + br_byte(d->r);
			d->low <<= 8; d->range <<= 8;
			if (d->r->err) return 0;
		}
	}
	row[idx] += 2;
	return idx;
}

/* ------------------------------------------------------------------------
 * records
 * ---------------------------------------------------------------------- */
typedef struct
{
	u64 title, seq, qual;          /* offsets into the output text */
	u32 title_len, seq_len, qual_len;
} drec_t;

typedef struct
{
	const orc_config* cfg;
	br_t* r;
	u8* out; u64 cap, pos;
	int overflow;
	drec_t* recs; u32 n_recs;
	u32 max_qlen, min_qlen, flags, chunk_size;
	int cs_const; u8 cs_seq_begin, cs_qua_begin;
	u32 crc_tag, crc_seq, crc_qua;         /* stored in the block */
} dblk_t;

static void put(dblk_t* b, u8 c)
{
	if (b->pos < b->cap) b->out[b->pos] = c; else b->overflow = 1;
	b->pos++;
}

/* ------------------------------------------------------------------------
 * tags: TagTokenizerDecoder (src/TagModeler.cpp:887-1211), TagRawDecoder (:1287-1343)
 * ---------------------------------------------------------------------- */
enum { NS_NONE = 0, NS_VALUE_VAR, NS_VALUE_RLE, NS_DELTA_VAR, NS_DELTA_RLE, NS_DELTA_CONST };

typedef struct
{
	u8 sep; int is_constant, is_numeric, is_len_constant;
	u32 len, max_len, min_len;
	u8* data; u8* ham;
	u8 scheme; int var_stat;
	i32 min_value, max_value, min_delta, max_delta;
	u32 bits_value, bits_num, bits_len;
	hdec_t global; int has_global;
	hdec_t local[129]; u8 has_local[129];
	u32 rle_len; u32 rle_sym;
	u32 prev;
} dfield_t;

/* core::to_string (src/utils.h:69-97); values >= 10^9 overflow `power` in the reference (do-not-test zone) */
static void put_number(dblk_t* b, u32 value)
{
	char tmp[16]; u32 n = 0;
	if (value >= 1000000000u) g_dec_ub = 1;
	if (value == 0) { put(b, '0'); return; }
	while (value) { tmp[n++] = (char)('0' + value % 10); value /= 10; }
	while (n) put(b, (u8)tmp[--n]);
}

static void tags_read_fields(br_t* r, dfield_t** out_f, u32* out_n)
{
	const u32 nf = br_byte(r);
	dfield_t* F = (dfield_t*)calloc(nf ? nf : 1, sizeof(dfield_t));
	*out_f = F; *out_n = nf;
	for (u32 i = 0; i < nf && !r->err; ++i)
	{
		dfield_t* f = &F[i];
		f->sep = (u8)br_byte(r);
		f->is_constant = br_byte(r) != 0;
		if (f->is_constant)
		{
			f->len = br_word(r);
			if (f->len >= (1u << 16)) { r->err = 1; return; }
			f->data = (u8*)malloc(f->len + 1);
			for (u32 j = 0; j < f->len; ++j) f->data[j] = (u8)br_byte(r);
			continue;
		}
		f->is_numeric = br_byte(r) != 0;
		if (f->is_numeric)
		{
			f->scheme = (u8)br_byte(r);
			f->min_value = (i32)br_word(r); f->max_value = (i32)br_word(r);
			f->bits_value = bit_length((u64)(int64_t)(i32)((u32)f->max_value - (u32)f->min_value));
			f->bits_num = 0;
			switch (f->scheme)
			{
			case NS_DELTA_CONST: case NS_DELTA_RLE: case NS_DELTA_VAR:
				f->min_delta = (i32)br_word(r); f->max_delta = (i32)br_word(r);
				f->bits_num = bit_length((u64)(int64_t)(i32)((u32)f->max_delta - (u32)f->min_delta));
				if (f->scheme == NS_DELTA_VAR)
				{
					f->var_stat = (int)br_byte(r);
					if (f->var_stat) { hdec_load(&f->global, r); f->has_global = 1; }
				}
				break;
			case NS_VALUE_RLE:
				f->bits_num = f->bits_value;
				break;
			case NS_VALUE_VAR:
				f->bits_num = f->bits_value;
				f->var_stat = (int)br_byte(r);
				if (f->var_stat) { hdec_load(&f->global, r); f->has_global = 1; }
				break;
			default:
				r->err = 1; return;
			}
			continue;
		}
		f->is_len_constant = br_byte(r) != 0;
		f->len = br_word(r); f->max_len = br_word(r); f->min_len = br_word(r);
		if (f->len >= (1u << 16) || f->max_len >= (1u << 16) || f->min_len > f->max_len) { r->err = 1; return; }
		f->bits_len = bit_length(f->max_len - f->min_len);
		f->data = (u8*)malloc(f->len + 1); f->ham = (u8*)malloc(f->len + 1);
		for (u32 j = 0; j < f->len; ++j) f->data[j] = (u8)br_byte(r);
		for (u32 j = 0; j < f->len; ++j) f->ham[j] = (u8)br_bit(r);
		br_align(r);
		for (u32 j = 0; j < MINV(f->max_len, 128u); ++j)
			if (j >= f->len || !f->ham[j]) { hdec_load(&f->local[j], r); f->has_local[j] = 1; }
		if (f->max_len >= 128) { hdec_load(&f->local[128], r); f->has_local[128] = 1; }
	}
}

/* TagTokenizerDecoder::ReadNumericField (src/TagModeler.cpp:1098-1205) */
static u32 tags_read_numeric(br_t* r, dfield_t* f, u32 rec_counter)
{
	u32 v = 0;
	if (rec_counter == 0)
	{
		v = br_bits(r, f->bits_value);
		if (f->scheme == NS_VALUE_RLE) { f->rle_len = br_bits(r, 8); f->rle_sym = v; }
		return v + (u32)f->min_value;
	}
	switch (f->scheme)
	{
	case NS_DELTA_CONST:
		return f->prev + (u32)f->min_delta;
	case NS_DELTA_RLE:
		if (rec_counter == 1 || f->rle_len == 0)
		{
			v = br_bits(r, f->bits_num); f->rle_sym = v; f->rle_len = br_bits(r, 8);
		}
		else { f->rle_len--; v = f->rle_sym; }
		return v + f->prev + (u32)f->min_delta;
	case NS_VALUE_VAR: case NS_DELTA_VAR:
		v = f->has_global ? hdec_sym(&f->global, r) : br_bits(r, f->bits_num);
		return f->scheme == NS_DELTA_VAR ? v + f->prev + (u32)f->min_delta : v + (u32)f->min_value;
	case NS_VALUE_RLE:
		if (f->rle_len == 0) { v = br_bits(r, f->bits_num); f->rle_sym = v; f->rle_len = br_bits(r, 8); }
		else { f->rle_len--; v = f->rle_sym; }
		return v + (u32)f->min_value;
	default:
		r->err = 1; return 0;
	}
}

/* ReadTags (src/BlockCompressor.cpp:491-573): titles are decoded straight into the text buffer, and the
 * positions of the sequence / plus / quality lines are fixed as soon as the record's length is known */
static void tags_decode(dblk_t* b)
{
	br_t* r = b->r;
	const int mixed = (b->flags & 4) != 0;
	const u32 len_bits = bit_length((u64)(b->max_qlen - b->min_qlen));
	const int cs_delta = b->cfg->color_space && (b->flags & 1);
	dfield_t* F = NULL; u32 nf = 0;
	hdec_t raw; u8 raw_sym[128]; u32 raw_n = 0, min_title = 0, max_title = 0, tl_bits = 0;
	memset(&raw, 0, sizeof(raw));

	if (!mixed) tags_read_fields(r, &F, &nf);
	else
	{
		min_title = br_word(r); max_title = br_word(r);
		tl_bits = bit_length((u64)(max_title - min_title));
		for (u32 i = 0; i < 128; ++i) if (br_bit(r)) raw_sym[raw_n++] = (u8)i;
		hdec_load(&raw, r);
	}

	for (u32 i = 0; i < b->n_recs && !r->err && !b->overflow; ++i)
	{
		drec_t* rec = &b->recs[i];
		rec->title = b->pos;
		if (!mixed)
		{
			for (u32 j = 0; j < nf; ++j)
			{
				dfield_t* f = &F[j];
				if (f->is_constant) { for (u32 k = 0; k < f->len; ++k) put(b, f->data[k]); put(b, f->sep); continue; }
				if (f->is_numeric)
				{
					const u32 v = tags_read_numeric(r, f, i);
					put_number(b, v); f->prev = v; put(b, f->sep);
					continue;
				}
				const u32 fl = f->is_len_constant ? f->len : br_bits(r, f->bits_len) + f->min_len;
				for (u32 k = 0; k < fl; ++k)
				{
					if (k < f->len && f->ham[k]) put(b, f->data[k]);
					else put(b, (u8)hdec_sym(&f->local[MINV(k, 128u)], r));
				}
				put(b, f->sep);
			}
			if (nf == 0) { r->err = 1; break; }
			b->pos--;                                     /* the last separator is not part of the title */
			rec->title_len = (u32)(b->pos - rec->title);
		}
		else
		{
			rec->title_len = tl_bits ? br_bits(r, tl_bits) + min_title : max_title;
			for (u32 k = 0; k < rec->title_len; ++k)
			{
				const u32 s = hdec_sym(&raw, r);
				put(b, s < raw_n ? raw_sym[s] : 255);
			}
		}
		put(b, '\n');
		rec->qual_len = len_bits ? br_bits(r, len_bits) + b->min_qlen : b->max_qlen;
		rec->seq_len = rec->qual_len;
		rec->seq = b->pos; b->pos += rec->seq_len;
		if (cs_delta) { rec->seq++; b->pos++; }
		put(b, '\n');
		put(b, '+');
		if (b->cfg->plus_repetition)
			for (u32 k = 1; k < rec->title_len; ++k) put(b, rec->title + k < b->cap ? b->out[rec->title + k] : 0);
		put(b, '\n');
		rec->qual = b->pos; b->pos += rec->qual_len;
		if (cs_delta) { rec->qual++; b->pos++; }
		put(b, '\n');
		if (b->pos > b->cap) b->overflow = 1;
	}
	br_align(r);                                          /* FinishDecoding */

	for (u32 j = 0; j < nf; ++j)
	{
		dfield_t* f = &F[j];
		free(f->data); free(f->ham);
		if (f->has_global) hdec_free(&f->global);
		for (u32 k = 0; k < 129; ++k) if (f->has_local[k]) hdec_free(&f->local[k]);
	}
	free(F);
	if (mixed) hdec_free(&raw);
}

/* ------------------------------------------------------------------------
 * quality: QualityNormalModelerProxy / QualityOrderModelerProxy* ::Decode
 * (src/QualityModelerProxy.h:59-69,156-159), position modelers
 * (src/QualityPositionModeler.cpp:39-103,189-220,291-337), RLE modeler
 * (src/QualityRLEModeler.cpp:48-113,380-486), order modelers (src/QualityOrderModeler.h:49-65,
 * src/QualityEncoder.h:77-94,134-143,248-263,306-357)
 * ---------------------------------------------------------------------- */
static int is_special(const dblk_t* b, u32 q) { return b->cfg->lossy ? q == 0 : q >= 128; }

static void qua_position_decode(dblk_t* b, int truncated)
{
	br_t* r = b->r;
	br_align(r);
	const u32 maxl = br_word(r);
	u8 sym[256]; u32 n = 0;
	for (u32 i = 0; i < 256; ++i) if (br_bit(r)) sym[n++] = (u8)i;
	if (maxl > 65535 || r->err) { r->err = 1; return; }
	hdec_t* trees = (hdec_t*)calloc(maxl ? maxl : 1, sizeof(hdec_t));
	for (u32 i = 0; i < maxl && !r->err; ++i) hdec_load(&trees[i], r);
	const u32 max_bits = bit_length(maxl);
	const int variable = truncated ? (int)br_bit(r) : 0;
	const u8 hash = b->cfg->lossy ? 1 : 2;                /* HashSymbolQuantized / HashSymbolNormal */
	for (u32 k = 0; k < b->n_recs && !r->err; ++k)
	{
		drec_t* rec = &b->recs[k];
		u32 th = rec->qual_len, ncount = 0;
		if (truncated && br_bit(r)) th = br_bits(r, variable ? bit_length(rec->qual_len) : max_bits);
		if (th > rec->qual_len || th > maxl) { r->err = 1; break; }
		for (u32 j = 0; j < th; ++j)
		{
			const u32 s = hdec_sym(&trees[j], r);
			const u8 q = s < n ? sym[s] : 255;
			b->out[rec->qual + j] = q;
			ncount += is_special(b, q);
		}
		for (u32 j = th; j < rec->qual_len; ++j) b->out[rec->qual + j] = hash;
		rec->seq_len = rec->qual_len - ncount;
	}
	br_align(r);
	for (u32 i = 0; i < maxl; ++i) hdec_free(&trees[i]);
	free(trees);
}

static void qua_rle_decode(dblk_t* b)
{
	br_t* r = b->r;
	const u32 run_len = br_word(r);                        /* ReadStatsData: no alignment before it */
	u8 qs[256], ls[256]; u32 qn = 0, ln = 0;
	memset(qs, 255, sizeof(qs)); memset(ls, 255, sizeof(ls));
	for (u32 i = 0; i < 256; ++i) if (br_bit(r)) qs[qn++] = (u8)i;
	for (u32 i = 0; i < 256; ++i) if (br_bit(r)) ls[ln++] = (u8)i;
	br_align(r);
	if (qn == 0 || ln == 0 || run_len == 0 || r->err) { r->err = 1; return; }
	u8* sym_run = (u8*)malloc((size_t)run_len + 1); u8* len_run = (u8*)malloc((size_t)run_len + 1);
	if (qn > 1)
	{
		hdec_t* qt = (hdec_t*)calloc(qn, sizeof(hdec_t)); hdec_t* lt = (hdec_t*)calloc(qn, sizeof(hdec_t));
		for (u32 i = 0; i < qn && !r->err; ++i) { hdec_load(&qt[i], r); hdec_load(&lt[i], r); }
		br_align(r);
		u32 prev = 0;
		for (u32 i = 0; i < run_len && !r->err; ++i)
		{
			u32 s = hdec_sym(&qt[prev], r);
			if (s >= qn) { r->err = 1; break; }
			sym_run[i] = qs[s]; prev = s;
			s = hdec_sym(&lt[prev], r);
			if (s >= ln) { r->err = 1; break; }
			len_run[i] = ls[s];
		}
		for (u32 i = 0; i < qn; ++i) { hdec_free(&qt[i]); hdec_free(&lt[i]); }
		free(qt); free(lt);
	}
	else
	{
		/* one quality value in the whole block: every run has the first length, the last run the other one */
		br_align(r);
		u8 l_begin, l_end;
		if (ln > 1)
		{
			br_align(r);
			l_begin = ls[br_byte(r) & 255];
			l_end = ls[0];
			if (l_end == l_begin) l_end = ls[1];
		}
		else { l_begin = ls[0]; l_end = l_begin; }
		memset(sym_run, qs[0], run_len); memset(len_run, l_begin, run_len);
		len_run[run_len - 1] = l_end;
	}
	/* DecodeRecords (src/QualityRLEModeler.cpp:80-113): runs cross record boundaries */
	u32 cur_len = 0, idx = 0; u8 cur_q = 0;
	for (u32 k = 0; k < b->n_recs && !r->err; ++k)
	{
		drec_t* rec = &b->recs[k];
		u32 ncount = 0;
		for (u32 j = 0; j < rec->qual_len; ++j)
		{
			if (cur_len == 0)
			{
				if (idx >= run_len) { r->err = 1; break; }
				cur_q = sym_run[idx]; cur_len = (u32)len_run[idx] + 1; idx++;
			}
			b->out[rec->qual + j] = cur_q; --cur_len;
			ncount += is_special(b, cur_q);
		}
		rec->seq_len = rec->qual_len - ncount;
	}
	br_align(r);
	free(sym_run); free(len_run);
}

static void qua_order_decode(dblk_t* b, u32 n, u32 ord, u32 rescale, const u8* translate)
{
	br_t* r = b->r;
	const u32 abits = int_log2(n);
	const u64 models = 1ull << (abits * (ord + 1));
	u16* tab = (u16*)malloc(models * n * sizeof(u16));
	for (u64 i = 0; i < models * n; ++i) tab[i] = 1;
	const u64 sym_mask = ((u64)1 << abits) - 1;
	const u32 bits_lo = (ord / 2) * abits, bits_hi = (ord / 2 + 1) * abits;
	const u64 lo_mask = bits_lo ? (((u64)1 << bits_lo) - 1) : 0;
	const u64 hi_mask = ((u64)1 << bits_hi) - 1;
	const u64 swap_mask = lo_mask | ~hi_mask;
	const u64 hash_mask = ((u64)1 << (ord * abits)) - 1;
	u64 hash = 0, sym_buf = 0;
	rd_t d; rd_start(&d, r);
	for (u32 k = 0; k < b->n_recs && !r->err; ++k)
	{
		drec_t* rec = &b->recs[k];
		u32 ncount = 0;
		for (u32 j = 0; j < rec->qual_len && !r->err; ++j)
		{
			const u32 pctx = j * rescale / rec->qual_len;
			const u64 h = ((hash & hash_mask) << abits) | pctx;
			const u32 c = row_decode(tab + h * n, n, &d);
			const u8 q = translate ? translate[c] : (u8)c;
			b->out[rec->qual + j] = q;
			ncount += is_special(b, q);
			hash <<= abits;
			const u64 next_buf = (hash >> bits_lo) & sym_mask;
			const u64 swp = (next_buf + sym_buf) / 2;
			hash &= swap_mask; hash |= swp << bits_lo; hash |= c;
			sym_buf = next_buf;
		}
		rec->seq_len = rec->qual_len - ncount;
	}
	free(tab);
}

static void qua_decode(dblk_t* b)
{
	br_t* r = b->r;
	const u32 qo = b->cfg->quality_order;
	if (qo > 0 && b->cfg->lossy) { qua_order_decode(b, 8, qo, 8, NULL); return; }
	const u32 scheme = br_byte(r);
	if (scheme == 255) return;                           /* SchemeNone: never written for quality */
	if (qo == 0)
	{
		if (scheme == 2) qua_rle_decode(b);
		else if (scheme <= 1) qua_position_decode(b, scheme == 1);
		else r->err = 1;
		return;
	}
	static const u32 N_[8] = {16, 32, 64, 128, 16, 32, 64, 128};
	static const u32 ORD1[4] = {3, 2, 1, 1}, ORD2[4] = {4, 3, 2, 1};
	if (scheme > 7) { r->err = 1; return; }
	const u32 n = N_[scheme], ord = (qo == 1) ? ORD1[scheme & 3] : ORD2[scheme & 3];
	/* TTranslationalQualityEncoder::Read (src/QualityEncoder.h:344-357) */
	u8 sym[256]; u32 cnt = 0;
	memset(sym, 255, sizeof(sym));
	br_align(r);
	for (u32 i = 0; i < 256; ++i) if (br_bit(r)) sym[cnt++] = (u8)i;
	br_align(r);
	qua_order_decode(b, n, ord, scheme < 4 ? 8 : n, sym);
}

/* ------------------------------------------------------------------------
 * DNA: IDnaModelerProxy::Decode (src/DnaModelerProxy.h:61-71), DnaModelerBasicB2::Decode
 * (src/DnaModelerBasicB2.h:48-60), DnaModelerHuffman::Decode (src/DnaModelerHuffman.cpp:75-113),
 * TDnaRCOrderModeler::Decode (src/DnaModelerRCO.h:62-79)
 * ---------------------------------------------------------------------- */
static void dna_decode(dblk_t* b)
{
	br_t* r = b->r;
	const u32 order = b->cfg->dna_order;
	const u32 scheme = br_byte(r);
	if (scheme == 255) return;
	if (order == 0)
	{
		if (scheme == 0)
		{
			for (u32 k = 0; k < b->n_recs; ++k)
				for (u32 j = 0; j < b->recs[k].seq_len; ++j) b->out[b->recs[k].seq + j] = (u8)br_bits(r, 2);
			br_align(r);
			return;
		}
		if (scheme != 1) { r->err = 1; return; }
		u8 sym[20]; u32 n = 0;
		memset(sym, 255, sizeof(sym));
		for (u32 i = 0; i < 20; ++i) if (br_bit(r)) sym[n++] = (u8)i;
		hdec_t h; hdec_load(&h, r);
		for (u32 k = 0; k < b->n_recs && !r->err; ++k)
			for (u32 j = 0; j < b->recs[k].seq_len; ++j)
			{
				const u32 s = hdec_sym(&h, r);
				b->out[b->recs[k].seq + j] = s < 20 ? sym[s] : 255;
			}
		br_align(r);
		hdec_free(&h);
		return;
	}
	if (scheme > 1) { r->err = 1; return; }
	const u32 n = scheme ? 8 : 4, abits = scheme ? 3 : 2;
	const u32 ord = scheme ? MINV(order, 7u) : order;
	const u64 models = 1ull << (abits * ord);
	u16* tab = (u16*)malloc(models * n * sizeof(u16));
	for (u64 i = 0; i < models * n; ++i) tab[i] = 1;
	u64 hash = 0; const u64 mask = models - 1;
	rd_t d; rd_start(&d, r);
	for (u32 k = 0; k < b->n_recs && !r->err; ++k)
		for (u32 j = 0; j < b->recs[k].seq_len && !r->err; ++j)
		{
			const u32 s = row_decode(tab + hash * n, n, &d);
			b->out[b->recs[k].seq + j] = (u8)s;
			hash = ((hash << abits) | s) & mask;
		}
	free(tab);
}

/* ------------------------------------------------------------------------
 * ProcessBackward (src/RecordsProcessor.cpp:269-315,410-454) + colour space
 * (ProcessRecordToColorSpace, :60-101)
 * ---------------------------------------------------------------------- */
static const char DNA_ORDER[] = "AGCTNRWSKMDVHBYXU.-";   /* src/RecordsProcessor.cpp:186-206; index 19 stays 255 */

static void records_backward(dblk_t* b)
{
	static const u8 lossy_q[8] = {0, 6, 15, 22, 27, 33, 37, 40};
	u8 from_idx[256];
	memset(from_idx, 255, sizeof(from_idx));
	for (u32 i = 0; DNA_ORDER[i]; ++i) from_idx[i] = (u8)DNA_ORDER[i];
	const u32 off = b->cfg->quality_offset;
	const int lossy = b->cfg->lossy;
	for (u32 k = 0; k < b->n_recs; ++k)
	{
		drec_t* rec = &b->recs[k];
		u8* seq = b->out + rec->seq; u8* qua = b->out + rec->qual;
		i32 si = (i32)rec->seq_len - 1;
		for (i32 i = (i32)rec->qual_len - 1; i >= 0; --i)
		{
			u32 q = qua[i], sv;
			if (!lossy)
			{
				if (q >= 128) { sv = (q - 128 + 16) / 8 + 3 - 1; q &= 7; }
				else sv = si >= 0 ? seq[si--] : (g_dec_ub = 1, 0u);
				if (sv >= 20) g_dec_ub = 1;
				seq[i] = from_idx[sv]; qua[i] = (u8)(off + q);
			}
			else
			{
				if (q == 0) sv = 4;
				else sv = si >= 0 ? seq[si--] : (g_dec_ub = 1, 0u);
				if (sv >= 20 || q >= 8) g_dec_ub = 1;
				seq[i] = from_idx[sv]; qua[i] = (u8)(off + lossy_q[q & 7]);
			}
		}
		rec->seq_len = rec->qual_len;

		if (b->cfg->color_space)
		{
			/* without a constant primer the reference looks record 0's *character* up in the index table and adds the
			 * quality offset twice: undefined (and never a round trip) */
			if (!b->cs_const) { g_dec_ub = 1; continue; }
			static const char deltas[] = "NNACGT" "NNCATG" "NNGTAC" "NNTGCA";
			const u8 seq0 = from_idx[b->cs_seq_begin < 20 ? b->cs_seq_begin : 19];
			const u8 qua0 = (u8)(b->cs_qua_begin + off);
			rec->seq--; rec->qual--; rec->seq_len++; rec->qual_len++;
			seq = b->out + rec->seq; qua = b->out + rec->qual;
			u8 symbol = seq0;
			seq[0] = seq0; qua[0] = qua0;
			const char* m = deltas;
			for (u32 i = 1; i < rec->seq_len; ++i)
			{
				switch (symbol) { case 'A': m = deltas; break; case 'C': m = deltas + 6; break; case 'G': m = deltas + 12; break; case 'T': m = deltas + 18; break; default: break; }
				symbol = seq[i];
				u32 c = 0; while (c < 6 && (u8)m[c] != symbol) c++;
				seq[i] = (u8)(c + '.');
			}
		}
	}
}

/* ------------------------------------------------------------------------
 * block
 * ---------------------------------------------------------------------- */
int orc_decompress_block(const orc_config* cfg, const uint8_t* in, uint64_t size, uint8_t* out, uint64_t cap,
						 uint64_t* out_size, uint32_t stored_crc[3], uint32_t actual_crc[3])
{
	br_t r; memset(&r, 0, sizeof(r));
	r.mem = in; r.size = size;
	dblk_t b; memset(&b, 0, sizeof(b));
	b.cfg = cfg; b.r = &r; b.out = out; b.cap = cap;
	g_dec_ub = 0;
	*out_size = 0;

	/* ReadMetaData (src/BlockCompressor.cpp:300-356) */
	b.n_recs = br_word(&r); b.max_qlen = br_word(&r); b.flags = br_word(&r); b.chunk_size = br_word(&r);
	b.min_qlen = (b.flags & 2) ? br_word(&r) : b.max_qlen;
	if (cfg->color_space)
	{
		b.cs_const = (b.flags & 1) != 0;
		if (b.cs_const) { b.cs_seq_begin = (u8)br_byte(&r); b.cs_qua_begin = (u8)br_byte(&r); }
	}
	if (cfg->calc_crc32)
	{
		if (!cfg->tag_preserve_flags) b.crc_tag = br_word(&r);
		b.crc_seq = br_word(&r);
		if (!cfg->lossy) b.crc_qua = br_word(&r);
	}
	br_align(&r);
	if (r.err || b.n_recs == 0 || b.n_recs > size * 8 || b.max_qlen > 65535 || b.min_qlen > b.max_qlen) return ORC_E_INPUT;

	b.recs = (drec_t*)calloc(b.n_recs, sizeof(drec_t));
	tags_decode(&b);
	if (!r.err && !b.overflow) qua_decode(&b);
	if (!r.err && !b.overflow) dna_decode(&b);
	int rc = ORC_OK;
	if (b.overflow) rc = ORC_E_CAP;
	else if (r.err) rc = ORC_E_INPUT;
	if (rc == ORC_OK)
	{
		records_backward(&b);
		if (g_dec_ub) rc = ORC_E_UNSUPPORTED;
	}
	if (rc == ORC_OK)
	{
		/* the reference hands back chunkSize + 1 bytes (src/BlockCompressor.cpp:279-281); the text it laid out
		 * has that length whenever the block came from FASTQ text through Store() -- except with -f on files that
		 * repeat the title on the plus line, where the tail of its buffer is stale memory.  We return what was laid out. */
		*out_size = b.pos;
		u32 ct = 0xFFFFFFFFu, cs = 0xFFFFFFFFu, cq = 0xFFFFFFFFu;
		if (actual_crc)
		{
			for (u32 k = 0; k < b.n_recs; ++k)
			{
				const drec_t* rec = &b.recs[k];
				ct = orc_crc32_update(ct, out + rec->title, rec->title_len);
				cs = orc_crc32_update(cs, out + rec->seq, rec->seq_len);
				cq = orc_crc32_update(cq, out + rec->qual, rec->qual_len);
			}
			actual_crc[0] = ct ^ 0xFFFFFFFFu; actual_crc[1] = cs ^ 0xFFFFFFFFu; actual_crc[2] = cq ^ 0xFFFFFFFFu;
		}
		if (stored_crc) { stored_crc[0] = b.crc_tag; stored_crc[1] = b.crc_seq; stored_crc[2] = b.crc_qua; }
	}
	free(b.recs);
	return rc;
}

/* BlockCompressor::VerifyChecksum (src/BlockCompressor.cpp:576-594): decode and compare the checksums the
 * settings enable.  Returns 1 = valid, 0 = mismatch, < 0 = error. */
int orc_verify_block(const orc_config* cfg, const uint8_t* in, uint64_t size, uint64_t text_cap)
{
	u8* tmp = (u8*)malloc(text_cap ? text_cap : 1);
	u64 n = 0; u32 st[3], ac[3];
	int rc = orc_decompress_block(cfg, in, size, tmp, text_cap, &n, st, ac);
	free(tmp);
	if (rc != ORC_OK) return rc;
	int ok = 1;
	if (!cfg->tag_preserve_flags) ok &= st[0] == ac[0];
	ok &= st[1] == ac[1];
	if (!cfg->lossy) ok &= st[2] == ac[2];
	return ok;
}
