/*
 * TEST INFRASTRUCTURE ONLY -- the parity oracle (see dsrc_oracle.h).
 *
 * Plain-C restatement of the DSRC 2 block compressor.  Every function cites the
 * reference file:line it follows (paths relative to /root/reference/).  The
 * architecture is our own (one flat record index + four independent stream
 * encoders over a pure MSB-first bit sink); the *behaviour* is the reference's,
 * quirks included.
 *
 * Parity: pinned against oracle/_ref (the unmodified reference) by
 * tests/test_oracle_vs_ref.py and against the vectors in tests/golden.
 */
#define _GNU_SOURCE
#define _FILE_OFFSET_BITS 64
#include "dsrc_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;

#define MINV(a, b) ((a) <= (b) ? (a) : (b))
#define MAXV(a, b) ((a) >= (b) ? (a) : (b))

/* ------------------------------------------------------------------------
 * bit sink: src/BitMemory.h:215-476.  All call sites keep direct byte writes
 * aligned, so the output is a pure MSB-first bit stream.
 * ---------------------------------------------------------------------- */
typedef struct
{
	u8* buf;
	u64 cap, pos;
	u64 acc;
	u32 nbits;
} bw_t;

static void bw_init(bw_t* w, u64 cap)
{
	w->cap = cap < 64 ? 64 : cap;
	w->buf = (u8*)malloc(w->cap);
	w->pos = 0; w->acc = 0; w->nbits = 0;
}

static void bw_free(bw_t* w) { free(w->buf); w->buf = NULL; }

static void bw_emit(bw_t* w, u8 b)
{
	if (w->pos >= w->cap)
	{
		w->cap += w->cap >> 1;
		w->buf = (u8*)realloc(w->buf, w->cap);
	}
	w->buf[w->pos++] = b;
}

/* PutBits (src/BitMemory.h:318-338); n == 0 is a no-op (Appendix B.11) */
static void bw_bits(bw_t* w, u32 v, u32 n)
{
	if (n == 0) return;
	if (n < 32) v &= ((u32)1 << n) - 1;
	w->acc = (w->acc << n) | v;
	w->nbits += n;
	while (w->nbits >= 8)
	{
		bw_emit(w, (u8)(w->acc >> (w->nbits - 8)));
		w->nbits -= 8;
	}
}

static void bw_bit(bw_t* w, u32 b) { bw_bits(w, b & 1, 1); }
static void bw_byte(bw_t* w, u32 b) { bw_bits(w, b & 0xFF, 8); }
static void bw_word(bw_t* w, u32 v) { bw_bits(w, v >> 16, 16); bw_bits(w, v & 0xFFFF, 16); }
static void bw_dword(bw_t* w, u64 v) { bw_word(w, (u32)(v >> 32)); bw_word(w, (u32)v); }

/* FlushPartialWordBuffer (src/BitMemory.h:394-409): zero-pad to the next byte */
static void bw_flush(bw_t* w)
{
	if (w->nbits)
	{
		bw_emit(w, (u8)(w->acc << (8 - w->nbits)));
		w->nbits = 0;
	}
	w->acc = 0;
}

/* ------------------------------------------------------------------------
 * utils: src/utils.h:138-190
 * ---------------------------------------------------------------------- */
static u32 bit_length(u64 x)
{
	for (u32 i = 0; i < 32; ++i)
		if (x < (1ull << i)) return i;
	return 64;
}

static u32 int_log2(u32 x)
{
	u32 r = 0;
	for (u64 t = 2; t <= x; t *= 2) ++r;
	return r;
}

static int is_num(const u8* s, u32 len, u32* val)
{
	u32 v = 0, i;
	for (i = 0; i < len; ++i)
	{
		if (s[i] < '0' || s[i] > '9') break;
		v = v * 10 + (u32)(s[i] - '0');
	}
	*val = v;
	return i == len && (len == 1 || s[0] != '0');
}

static u32 to_num(const u8* s, u32 len)
{
	u32 r = 0;
	for (u32 i = 0; i < len; ++i) r = r * 10 + (u32)(s[i] - '0');
	return r;
}

/* ------------------------------------------------------------------------
 * CRC-32: src/Crc32.h:24-104 (poly 0xEDB88320, init/final 0xFFFFFFFF)
 * ---------------------------------------------------------------------- */
static u32 crc_table[256];
static int crc_ready = 0;

static void crc_init(void)
{
	if (crc_ready) return;
	for (u32 i = 0; i < 256; ++i)
	{
		u32 h = i;
		for (int j = 0; j < 8; ++j) h = (h & 1) ? (0xEDB88320u ^ (h >> 1)) : (h >> 1);
		crc_table[i] = h;
	}
	crc_ready = 1;
}

static u32 crc_update(u32 crc, const u8* p, u32 n)
{
	for (u32 i = 0; i < n; ++i) crc = (crc >> 8) ^ crc_table[(p[i] ^ crc) & 0xFF];
	return crc;
}

/* raw register update (no initial / final inversion), for running checksums over many pieces */
uint32_t orc_crc32_update(uint32_t state, const uint8_t* p, uint32_t n)
{
	crc_init();
	return crc_update(state, p, n);
}

uint32_t orc_crc32(const uint8_t* p, uint32_t n)
{
	crc_init();
	return crc_update(0xFFFFFFFFu, p, n) ^ 0xFFFFFFFFu;
}

/* ------------------------------------------------------------------------
 * Huffman: src/huffman.h:26-122, src/huffman.cpp:94-221
 * The heap comparator is a strict total order (lowest frequency first, ties ->
 * lowest id), so a plain "select the minimum" gives the same tree.
 * ---------------------------------------------------------------------- */
#define HUF_MAX 512

typedef struct
{
	u32 n;                       /* n_symbols */
	i32 root;
	u32 code[2 * HUF_MAX], len[2 * HUF_MAX];
	i32 left[2 * HUF_MAX], right[2 * HUF_MAX];
} huff_t;

typedef struct { u32 sym, freq; } hfreq_t;

/* set when the input drives the reference into undefined behaviour:
 * the caller then reports ORC_E_UNSUPPORTED instead of pretending to know the reference's bytes */
static __thread int g_ref_ub;
static __thread int g_chunk_size_set = 0; static __thread u64 g_chunk_size_value = 0;

static int hf_less(const hfreq_t* a, const hfreq_t* b)   /* a pops before b */
{
	return a->freq < b->freq || (a->freq == b->freq && a->sym < b->sym);
}

static u32 hf_min(const hfreq_t* h, u32 n)
{
	u32 m = 0;
	for (u32 i = 1; i < n; ++i)
		if (hf_less(&h[i], &h[m])) m = i;
	return m;
}

/* HuffmanEncoder::Restart/Insert/Complete (src/huffman.cpp:94-174) */
static void huff_build(huff_t* h, const u32* freqs, u32 n_in)
{
	hfreq_t heap[HUF_MAX + 1];
	u32 n = n_in;
	for (u32 i = 0; i < n; ++i) { heap[i].sym = i; heap[i].freq = freqs[i]; }
	if (n < 2)          /* Appendix B.4: reference reads a stale slot here; do-not-test zone */
	{
		g_ref_ub = 1;
		heap[1].sym = 1; heap[1].freq = 0;
		if (n == 0) { heap[0].sym = 0; heap[0].freq = 0; }
		n = 2;
	}
	h->n = n;
	for (u32 i = 0; i < 2 * n - 1; ++i)
	{
		h->code[i] = 0; h->len[i] = 0;
		h->left[i] = (i < n) ? -1 : 0; h->right[i] = (i < n) ? -1 : 0;
	}

	u32 hs = n;
	int special = 0;
	hfreq_t sp_left = {0, 0}, sp_right = {0, 0};
	if (hs == 2)
	{
		u32 top = hf_min(heap, 2);
		if (heap[top].freq == 0)
		{
			/* (*) special case: the heap is patched in place and NOT re-heapified,
			 * so the original top stays the first one popped (src/huffman.cpp:124-131) */
			heap[top].freq = 1;
			if (heap[1 - top].freq == 0) heap[1 - top].freq = 1;
			special = 1;
			sp_left = heap[top]; sp_right = heap[1 - top];
		}
	}
	else
	{
		while (hs > 2)
		{
			u32 m = hf_min(heap, hs);
			if (heap[m].freq != 0) break;
			heap[m] = heap[--hs];
		}
	}

	u32 present = hs;
	for (u32 i = 0; i + 1 < present; ++i)
	{
		hfreq_t l, r;
		if (special) { l = sp_left; r = sp_right; hs = 0; }
		else
		{
			u32 m = hf_min(heap, hs); l = heap[m]; heap[m] = heap[--hs];
			m = hf_min(heap, hs);     r = heap[m]; heap[m] = heap[--hs];
		}
		heap[hs].sym = n + i; heap[hs].freq = l.freq + r.freq; hs++;
		h->left[n + i] = (i32)l.sym; h->right[n + i] = (i32)r.sym;
	}
	for (i32 i = (i32)(n + present) - 2; i >= (i32)n; --i)
	{
		h->len[h->left[i]] = h->len[i] + 1;  h->code[h->left[i]] = h->code[i] << 1;
		h->len[h->right[i]] = h->len[i] + 1; h->code[h->right[i]] = (h->code[i] << 1) | 1;
	}
	h->root = (i32)(n + present) - 2;
}

static void huff_store_node(const huff_t* h, bw_t* w, i32 id, u32 bits_per_id)
{
	if (h->left[id] == -1)
	{
		bw_bit(w, 1);
		bw_bits(w, (u32)id, bits_per_id);
	}
	else
	{
		bw_bit(w, 0);
		huff_store_node(h, w, h->left[id], bits_per_id);
		huff_store_node(h, w, h->right[id], bits_per_id);
	}
}

/* HuffmanEncoder::StoreTree (src/huffman.cpp:177-221), SURVEY A.5 */
static void huff_store(const huff_t* h, bw_t* w)
{
	bw_flush(w);
	u64 size_pos = w->pos;
	bw_word(w, 0);
	u32 bits_per_id = int_log2(h->n);
	if (h->n & (h->n - 1)) bits_per_id++;
	u32 min_len = h->n;
	for (u32 i = 0; i < h->n; ++i)
		if (h->len[i] < min_len && h->len[i] > 0) min_len = h->len[i];
	bw_word(w, (u32)h->root);
	bw_word(w, h->n);
	bw_byte(w, min_len);
	huff_store_node(h, w, h->root, bits_per_id);
	bw_flush(w);
	u32 mem = (u32)(w->pos - size_pos);
	w->buf[size_pos] = (u8)(mem >> 24); w->buf[size_pos + 1] = (u8)(mem >> 16);
	w->buf[size_pos + 2] = (u8)(mem >> 8); w->buf[size_pos + 3] = (u8)mem;
}

/* ------------------------------------------------------------------------
 * Range coder + adaptive frequency rows: src/RangeCoder.h:51-84,
 * src/SymbolCoderRC.h:23-93.  SURVEY A.6.
 * ---------------------------------------------------------------------- */
typedef struct { u64 low; u32 range; bw_t* w; } rc_t;

static void rc_start(rc_t* rc, bw_t* w) { rc->low = 0; rc->range = 0xFFFFFFFFu; rc->w = w; }

static void rc_encode(rc_t* rc, u32 f, u32 c, u32 t)
{
	rc->range /= t;
	rc->low += (u32)(rc->range * c);          /* Freq is uint32: the product wraps */
	rc->range *= f;
	while (rc->range <= 0x00FFFFFFu)
	{
		if ((rc->low ^ (rc->low + rc->range)) & 0xFF00000000000000ull)
		{
			u32 r = (u32)rc->low;
			rc->range = (r | 0x00FFFFFFu) - r;
		}
		bw_byte(rc->w, (u32)(rc->low >> 56));
		rc->low <<= 8; rc->range <<= 8;
	}
}

static void rc_end(rc_t* rc)
{
	for (int i = 0; i < 8; ++i) { bw_byte(rc->w, (u32)(rc->low >> 56)); rc->low <<= 8; }
}

/* TSymbolCoderRC<N>::EncodeSymbol on a row of N uint16 counters */
static void row_encode(u16* row, u32 n, rc_t* rc, u32 sym)
{
	u32 acc = 0;
	for (u32 i = 0; i < n; ++i) acc += row[i];
	if (acc >= (1u << 16) - n * 2)
	{
		acc = 0;
		for (u32 i = 0; i < n; ++i) { row[i] -= row[i] >> 1; acc += row[i]; }
	}
	u32 lo = 0;
	for (u32 i = 0; i < sym; ++i) lo += row[i];
	rc_encode(rc, row[sym], lo, acc);
	row[sym] += 2;
}

/* ------------------------------------------------------------------------
 * records: src/Fastq.h:31-62, src/FastqParser.h:40-135, src/FastqParser.cpp:140-164
 * ---------------------------------------------------------------------- */
typedef struct
{
	u32 title, seq, qual;       /* offsets into the (private, mutable) chunk copy */
	u16 title_len, seq_len, qual_len, trunc_len;
} rec_t;

typedef struct
{
	u8* mem; u64 size, pos, skipped;
} parser_t;

static int p_getc(parser_t* p) { return p->pos == p->size ? -1 : p->mem[p->pos++]; }
static int p_peek(parser_t* p) { return p->pos == p->size ? -1 : p->mem[p->pos]; }

static u32 p_skip_line(parser_t* p)
{
	u32 len = 0;
	for (;;)
	{
		int c = p_getc(p);
		if (c == -1) break;
		if (c != '\n' && c != '\r') len++;
		else
		{
			if (c == '\r' && p_peek(p) == '\n') { p->pos++; p->skipped++; }
			break;
		}
	}
	return len;
}

static int p_next_record(parser_t* p, rec_t* r)
{
	if (p->pos == p->size) return 0;
	r->title = (u32)p->pos; r->title_len = (u16)p_skip_line(p);
	if (r->title_len == 0 || p->mem[r->title] != '@') return 0;
	r->seq = (u32)p->pos; r->seq_len = (u16)p_skip_line(p);
	u32 plus = p_skip_line(p);
	r->qual = (u32)p->pos; r->qual_len = (u16)p_skip_line(p);
	r->trunc_len = 0;
	return plus > 0 && r->seq_len == r->qual_len;
}

/* FastqParserExt::ReadNextRecord (src/FastqParser.cpp:198-251): with -f the title is rewritten in place to the fields
 * whose number (1-based) is set in the mask; a field ends at one of " ._,=:/-#" or NUL or at the end of the title and
 * is copied INCLUDING that end byte -- for the last field that is the line terminator, which thereby becomes part
 * of the title.  Titles longer than the reference's 512-byte scratch are undefined behaviour there. */
static int p_next_record_ext(parser_t* p, rec_t* r, u64 flags, u64* cut)
{
	if (p->pos == p->size) return 0;
	r->title = (u32)p->pos; r->title_len = (u16)p_skip_line(p);
	if (r->title_len == 0 || p->mem[r->title] != '@') return 0;
	{
		static const char seps[10] = " ._,=:/-#";          /* the tenth byte is the NUL */
		u8 buf[1024];
		u8* t = p->mem + r->title;
		const u32 tl = r->title_len;
		u32 field_no = 0, begin = 0, bp = 0;
		if (tl > 512) g_ref_ub = 1;
		for (u32 i = 0; i <= tl; ++i)
		{
			if (i != tl && !memchr(seps, t[i], 10)) continue;
			field_no++;
			if (field_no < 31 && (flags & (1u << field_no)))       /* BIT(x) is a 32-bit int shift in the reference */
			{
				if (bp + (i + 1 - begin) <= sizeof(buf)) memcpy(buf + bp, t + begin, i + 1 - begin);
				bp += i + 1 - begin;
			}
			if (field_no >= 31) g_ref_ub = 1;
			begin = i + 1;
		}
		if (bp > 512) g_ref_ub = 1;                              /* overruns the reference's scratch */
		/* with the last field kept bp may be tl + 1 (title + terminator): the release build has no assert and the
		 * unsigned subtraction wraps, i.e. the cut total goes DOWN by one */
		*cut += (u64)tl - (u64)bp;
		if (bp > 0 && bp <= sizeof(buf)) memcpy(t, buf, bp);
		r->title_len = (u16)bp;
	}
	r->seq = (u32)p->pos; r->seq_len = (u16)p_skip_line(p);
	u32 plus = p_skip_line(p);
	r->qual = (u32)p->pos; r->qual_len = (u16)p_skip_line(p);
	r->trunc_len = 0;
	return plus > 0 && r->seq_len == r->qual_len;
}

/* ------------------------------------------------------------------------
 * stats: src/Stats.h:44-101
 * ---------------------------------------------------------------------- */
typedef struct
{
	u32 d_count, d_freq[20]; u8 d_sym[20];
	u32 q_count, q_freq[256]; u8 q_sym[256];
	u32 min_len, max_len, raw_len, th_len, rle_len;
} stats_t;

typedef struct
{
	const orc_config* cfg;
	u8* mem; u64 size;
	rec_t* recs; u64 n_recs, recs_cap;
	u64 chunk_size;
	u64 raw[4];
	stats_t st;
	u32 crc_tag, crc_seq, crc_qual, crc_flags;
	u16 min_qlen, max_qlen; u32 flags;
	int cs_const; u8 cs_seq_begin, cs_qua_begin;      /* ColorSpaceStats (src/Stats.h:23-42) / ChunkHeader::csSeqBegin, csQuaBegin */
	void* tags;
	u32 fields_cap;
} block_t;

static const char DNA_ORDER[] = "AGCTNRWSKMDVHBYXU.-";   /* src/RecordsProcessor.cpp:186-206 */

/* ParseRecords (src/BlockCompressor.cpp:112-137) */
static void block_parse(block_t* b)
{
	parser_t p = { b->mem, b->size, 0, 0 };
	u64 cut = 0;
	b->recs_cap = 8 * 1024; b->recs = (rec_t*)malloc(b->recs_cap * sizeof(rec_t));
	b->n_recs = 0;
	memset(b->raw, 0, sizeof(b->raw));
	while (p.pos < p.size)
	{
		rec_t r;
		if (!(b->cfg->tag_preserve_flags ? p_next_record_ext(&p, &r, b->cfg->tag_preserve_flags, &cut) : p_next_record(&p, &r))) break;
		if (b->n_recs + 1 >= b->recs_cap)
		{
			b->recs_cap *= 2; b->recs = (rec_t*)realloc(b->recs, b->recs_cap * sizeof(rec_t));
		}
		b->recs[b->n_recs++] = r;
		b->raw[1] += r.title_len; b->raw[2] += r.seq_len; b->raw[3] += r.qual_len;
	}
	b->chunk_size = b->size - cut - p.skipped;
	if (g_chunk_size_set) b->chunk_size = g_chunk_size_value;     /* record-level API: see orc_compress_records_file */
}

/* PreprocessRecords: IRecordsProcessor::ProcessForward + Lossless/Lossy
 * ProcessForward + FinalizeStats (src/RecordsProcessor.cpp:112-154,209-267,344-408) */
static void block_preprocess(block_t* b)
{
	u8 dna_to_idx[256]; u8 lossy_bin[256];
	memset(dna_to_idx, 255, sizeof(dna_to_idx));
	for (u32 i = 0; DNA_ORDER[i]; ++i) dna_to_idx[(u8)DNA_ORDER[i]] = (u8)i;
	{
		static const u32 ranges[] = {0, 2, 10, 20, 25, 30, 35, 40, 64};
		memset(lossy_bin, 255, sizeof(lossy_bin));
		for (u32 i = 0; i < 8; ++i)
			for (u32 j = ranges[i]; j < ranges[i + 1]; ++j) lossy_bin[j] = (u8)i;
	}
	stats_t* s = &b->st;
	memset(s, 0, sizeof(*s));
	memset(s->d_sym, 255, sizeof(s->d_sym)); memset(s->q_sym, 255, sizeof(s->q_sym));
	s->min_len = 0xFFFFFFFFu;
	const u32 off = b->cfg->quality_offset;
	const int lossy = b->cfg->lossy;

	u32 c_tag = 0xFFFFFFFFu, c_seq = 0xFFFFFFFFu, c_qual = 0xFFFFFFFFu;
	crc_init();

	for (u64 k = 0; k < b->n_recs; ++k)
	{
		rec_t* r = &b->recs[k];
		u8* seq = b->mem + r->seq; u8* qua = b->mem + r->qual;
		if (b->crc_flags & 1) c_tag = crc_update(c_tag, b->mem + r->title, r->title_len);
		if (b->crc_flags & 2) c_seq = crc_update(c_seq, seq, r->seq_len);
		if (b->crc_flags & 4) c_qual = crc_update(c_qual, qua, r->qual_len);

		if (b->cfg->color_space)
		{
			/* IRecordsProcessor::ProcessRecordFromColorSpace + ProcessFromColorSpace (src/RecordsProcessor.cpp:25-58,
			 * src/RecordsProcessor.h:92-105): colours '.','/','0'..'3' -> bases through the transition matrix of the
			 * last base that was one of ACGT (matrix A before any); then the begin-symbol statistics on the raw chars */
			static const char deltas[] = "NNACGT" "NNCATG" "NNGTAC" "NNTGCA";
			const char* m = deltas;
			u8 sym = seq[0];
			for (u32 i = 1; i < r->seq_len; ++i)
			{
				switch (sym) { case 'A': m = deltas; break; case 'C': m = deltas + 6; break; case 'G': m = deltas + 12; break; case 'T': m = deltas + 18; break; default: break; }
				const u32 c = (u32)seq[i] - '.';
				if (c > 5) { g_ref_ub = 1; break; }             /* reads outside the 24-entry table */
				sym = (u8)m[c];
				seq[i] = sym;
			}
			if (k == 0) { b->cs_const = 1; b->cs_seq_begin = seq[0]; b->cs_qua_begin = qua[0]; }
			b->cs_const &= (b->cs_seq_begin == seq[0]);
		}
		u32 kept = 0, th = 0; u8 prev = 255;
		const u32 n = r->seq_len;
		for (u32 i = 0; i < n; ++i)
		{
			u8 sidx = dna_to_idx[seq[i]];
			u8 q;
			int keep;
			if (!lossy)
			{
				q = (u8)(qua[i] - off);
				if (sidx > 3 && q < 7)
				{
					q = (u8)(q + (u8)(128 + (((u32)sidx - 3 + 1) << 3) - 16));
					keep = 0;
				}
				else keep = 1;
			}
			else
			{
				q = lossy_bin[(u8)(qua[i] - off) & 255];   /* 64-entry LUT in the reference (Appendix B.6) */
				if (sidx >= 4) { q = 0; keep = 0; }
				else { if (q == 0) q = 1; keep = 1; }
			}
			seq[i] = sidx;                                       /* every base becomes its index in place, kept ones are then compacted */
			if (keep) { seq[kept++] = sidx; s->d_freq[sidx < 20 ? sidx : 19]++; }
			qua[i] = q;
			s->q_freq[q]++;
			if (q != prev) s->rle_len++;
			if (q != 2) th = i;
			prev = q;
		}
		r->seq_len = (u16)kept;
		r->trunc_len = (u16)(th + (r->qual_len > 0));
		if (prev == 2 && s->rle_len > 0) s->rle_len--;
		s->raw_len += r->qual_len;
		s->th_len += th;
		s->min_len = MINV(s->min_len, (u32)r->qual_len);
		s->max_len = MAXV(s->max_len, (u32)r->qual_len);
	}
	for (u32 i = 0; i < 20; ++i) if (s->d_freq[i]) s->d_sym[i] = (u8)s->d_count++;
	for (u32 i = 0; i < 256; ++i) if (s->q_freq[i]) s->q_sym[i] = (u8)s->q_count++;
	b->crc_tag = c_tag ^ 0xFFFFFFFFu; b->crc_seq = c_seq ^ 0xFFFFFFFFu; b->crc_qual = c_qual ^ 0xFFFFFFFFu;
}

/* ------------------------------------------------------------------------
 * DNA streams: src/DnaModelerProxy.h:29-229, DnaModelerBasicB2.h:34-46,
 * DnaModelerHuffman.cpp:21-73, DnaModelerRCO.h:27-132
 * ---------------------------------------------------------------------- */
static void dna_store(block_t* b, bw_t* w)
{
	const stats_t* s = &b->st;
	const u32 order = b->cfg->dna_order;
	if (s->d_count == 0) { bw_byte(w, 255); return; }

	if (order == 0)
	{
		if (s->d_count <= 4)
		{
			bw_byte(w, 0);
			for (u64 k = 0; k < b->n_recs; ++k)
			{
				const rec_t* r = &b->recs[k];
				for (u32 j = 0; j < r->seq_len; ++j) bw_bits(w, b->mem[r->seq + j] & 3, 2);
			}
			bw_flush(w);
			return;
		}
		bw_byte(w, 1);
		/* DnaModelerHuffman::ProcessStats inserts symbolFreqs[symbols[i]] for i < symbolCount
		 * (Appendix B.2): correct only for prefix-closed alphabets; index 255 reads
		 * qualityStats.symbolFreqs[229] in the reference's object layout. */
		u32 freqs[20];
		for (u32 i = 0; i < s->d_count; ++i)
		{
			u8 x = s->d_sym[i];
			freqs[i] = (x == 255) ? s->q_freq[229] : s->d_freq[x];
		}
		huff_t* h = (huff_t*)malloc(sizeof(huff_t));
		huff_build(h, freqs, s->d_count);
		for (u32 i = 0; i < 20; ++i) bw_bit(w, s->d_sym[i] != 255);
		bw_flush(w);
		huff_store(h, w);
		for (u64 k = 0; k < b->n_recs; ++k)
		{
			const rec_t* r = &b->recs[k];
			for (u32 j = 0; j < r->seq_len; ++j)
			{
				u8 x = s->d_sym[b->mem[r->seq + j]];
				bw_bits(w, h->code[x], h->len[x]);
			}
		}
		bw_flush(w);
		free(h);
		return;
	}

	/* order-k context + range coder */
	u32 scheme = (s->d_count <= 4) ? 0 : 1;
	u32 n = scheme ? 8 : 4, abits = scheme ? 3 : 2;
	u32 ord = scheme ? MINV(order, 7u) : order;
	bw_byte(w, scheme);
	u64 models = 1ull << (abits * ord);
	u16* tab = (u16*)malloc(models * n * sizeof(u16));
	for (u64 i = 0; i < models * n; ++i) tab[i] = 1;
	u64 hash = 0, mask = models - 1;
	rc_t rc; rc_start(&rc, w);
	for (u64 k = 0; k < b->n_recs; ++k)
	{
		const rec_t* r = &b->recs[k];
		for (u32 j = 0; j < r->seq_len; ++j)
		{
			u32 sym = b->mem[r->seq + j];
			if (sym >= n) g_ref_ub = 1;                          /* reference UB (Appendix B.3) */
			row_encode(tab + hash * n, n, &rc, sym & (n - 1));
			hash = ((hash << abits) | sym) & mask;
		}
	}
	rc_end(&rc);
	free(tab);
}

/* ------------------------------------------------------------------------
 * quality, level 0: src/QualityModelerProxy.h:27-128,
 * src/QualityPositionModeler.cpp:24-287, src/QualityRLEModeler.cpp:24-373
 * ---------------------------------------------------------------------- */
static void qua_position_store(block_t* b, bw_t* w, int truncated)
{
	const stats_t* s = &b->st;
	const u32 maxl = s->max_len, nsym = s->q_count;
	u32* hist = (u32*)calloc((size_t)maxl * nsym + 1, sizeof(u32));
	for (u64 k = 0; k < b->n_recs; ++k)
	{
		const rec_t* r = &b->recs[k];
		u32 n = truncated ? r->trunc_len : r->qual_len;
		for (u32 j = 0; j < n; ++j) hist[(size_t)j * nsym + s->q_sym[b->mem[r->qual + j]]]++;
	}
	huff_t* trees = (huff_t*)malloc(sizeof(huff_t) * (maxl ? maxl : 1));
	for (u32 i = 0; i < maxl; ++i) huff_build(&trees[i], hist + (size_t)i * nsym, nsym);

	bw_flush(w);
	bw_word(w, maxl);
	for (u32 i = 0; i < 256; ++i) bw_bit(w, s->q_sym[i] != 255);
	for (u32 i = 0; i < maxl; ++i) huff_store(&trees[i], w);

	if (truncated)
	{
		const int variable = s->min_len != s->max_len;
		const u32 max_bits = bit_length(maxl);
		bw_bit(w, (u32)variable);
		for (u64 k = 0; k < b->n_recs; ++k)
		{
			const rec_t* r = &b->recs[k];
			bw_bit(w, r->qual_len != r->trunc_len);
			if (r->qual_len != r->trunc_len)
				bw_bits(w, r->trunc_len, variable ? bit_length(r->qual_len) : max_bits);
			for (u32 j = 0; j < r->trunc_len; ++j)
			{
				u32 q = s->q_sym[b->mem[r->qual + j]];
				bw_bits(w, trees[j].code[q], trees[j].len[q]);
			}
		}
	}
	else
	{
		for (u64 k = 0; k < b->n_recs; ++k)
		{
			const rec_t* r = &b->recs[k];
			for (u32 j = 0; j < r->qual_len; ++j)
			{
				u32 q = s->q_sym[b->mem[r->qual + j]];
				bw_bits(w, trees[j].code[q], trees[j].len[q]);
			}
		}
	}
	bw_flush(w);
	free(trees); free(hist);
}

static void qua_rle_store(block_t* b, bw_t* w)
{
	const stats_t* s = &b->st;
	u64 cap = (u64)s->raw_len + 2;
	u8* sym_run = (u8*)malloc(cap); u8* len_run = (u8*)malloc(cap);
	u32 qf[256], lf[256];
	memset(qf, 0, sizeof(qf)); memset(lf, 0, sizeof(lf));
	u32 run_len = 0; u8 prev = 255, cur = 0;
	/* EncodeRecords (src/QualityRLEModeler.cpp:142-205): runs cross record boundaries, cap 254 */
	for (u64 k = 0; k < b->n_recs; ++k)
	{
		const rec_t* r = &b->recs[k];
		for (u32 j = 0; j < r->qual_len; ++j)
		{
			u8 q = b->mem[r->qual + j];
			if (q == prev && cur < 254) cur++;
			else
			{
				if (prev != 255)
				{
					sym_run[run_len] = prev; len_run[run_len++] = cur;
					qf[prev]++; lf[cur]++;
				}
				cur = 0; prev = q;
			}
		}
	}
	sym_run[run_len] = prev; len_run[run_len++] = cur;
	qf[prev]++; lf[cur]++;

	u8 qs[256], ls[256]; u32 qn = 0, ln = 0;
	memset(qs, 255, sizeof(qs)); memset(ls, 255, sizeof(ls));
	for (u32 i = 0; i < 256; ++i)
	{
		if (qf[i]) qs[i] = (u8)qn++;
		if (lf[i]) ls[i] = (u8)ln++;
	}

	huff_t* qt = NULL; huff_t* lt = NULL;
	if (qn > 1)
	{
		u32* qF = (u32*)calloc((size_t)qn * qn, sizeof(u32));
		u32* lF = (u32*)calloc((size_t)qn * ln, sizeof(u32));
		u32 p = 0;
		for (u32 i = 0; i < run_len; ++i)
		{
			u32 q = qs[sym_run[i]], l = ls[len_run[i]];
			qF[(size_t)p * qn + q]++; lF[(size_t)q * ln + l]++;
			p = q;
		}
		qt = (huff_t*)malloc(sizeof(huff_t) * qn); lt = (huff_t*)malloc(sizeof(huff_t) * qn);
		for (u32 i = 0; i < qn; ++i)
		{
			huff_build(&qt[i], qF + (size_t)i * qn, qn);
			huff_build(&lt[i], lF + (size_t)i * ln, ln);
		}
		free(qF); free(lF);
	}

	bw_flush(w);
	bw_word(w, run_len);
	for (u32 i = 0; i < 256; ++i) bw_bit(w, qs[i] != 255);
	for (u32 i = 0; i < 256; ++i) bw_bit(w, ls[i] != 255);
	if (qn > 1)
	{
		for (u32 i = 0; i < qn; ++i) { huff_store(&qt[i], w); huff_store(&lt[i], w); }
		u32 p = 0;
		for (u32 i = 0; i < run_len; ++i)
		{
			u32 q = qs[sym_run[i]], l = ls[len_run[i]];
			bw_bits(w, qt[p].code[q], qt[p].len[q]);
			bw_bits(w, lt[q].code[l], lt[q].len[l]);
			p = q;
		}
	}
	else if (ln > 1)
	{
		bw_flush(w);
		bw_byte(w, ls[len_run[0]]);
	}
	bw_flush(w);
	free(qt); free(lt); free(sym_run); free(len_run);
}

/* ------------------------------------------------------------------------
 * quality, levels 1-2: src/QualityModelerProxy.h:130-293, QualityOrderModeler.h,
 * QualityEncoder.h:24-367.  n = alphabet, ord = SymbolOrder, rescale = pctx scale.
 * ---------------------------------------------------------------------- */
static void qua_order_encode(block_t* b, bw_t* w, u32 n, u32 ord, u32 rescale, const u8* translate)
{
	u32 abits = int_log2(n);
	u64 models = 1ull << (abits * (ord + 1));
	u16* tab = (u16*)malloc(models * n * sizeof(u16));
	for (u64 i = 0; i < models * n; ++i) tab[i] = 1;

	const u64 sym_mask = ((u64)1 << abits) - 1;
	const u32 bits_lo = (ord / 2) * abits, bits_hi = (ord / 2 + 1) * abits;
	const u64 lo_mask = bits_lo ? (((u64)1 << bits_lo) - 1) : 0;
	const u64 hi_mask = ((u64)1 << bits_hi) - 1;
	const u64 swap_mask = lo_mask | ~hi_mask;
	const u64 hash_mask = ((u64)1 << (ord * abits)) - 1;
	u64 hash = 0, sym_buf = 0;

	rc_t rc; rc_start(&rc, w);
	for (u64 k = 0; k < b->n_recs; ++k)
	{
		const rec_t* r = &b->recs[k];
		for (u32 j = 0; j < r->qual_len; ++j)
		{
			u32 q = b->mem[r->qual + j];
			u32 sym = translate ? translate[q] : q;
			u32 pctx = j * rescale / r->qual_len;
			u64 h = ((hash & hash_mask) << abits) | pctx;
			row_encode(tab + h * n, n, &rc, sym & (n - 1));
			/* TQualityModelBase::UpdateHash (src/QualityEncoder.h:77-89) */
			hash <<= abits;
			u64 next_buf = (hash >> bits_lo) & sym_mask;
			u64 swp = (next_buf + sym_buf) / 2;
			hash &= swap_mask;
			hash |= swp << bits_lo;
			hash |= sym;
			sym_buf = next_buf;
		}
	}
	rc_end(&rc);
	free(tab);
}

static int qua_store(block_t* b, bw_t* w)
{
	const stats_t* s = &b->st;
	const u32 qo = b->cfg->quality_order;
	if (qo == 0)
	{
		/* QualityNormalModelerProxy::SelectSchemeId (src/QualityModelerProxy.h:113-122) */
		u32 scheme;
		if ((float)s->th_len / (float)s->rle_len > 1.25f) scheme = 2;
		else if ((float)s->raw_len / (float)s->th_len > 1.10f) scheme = 1;
		else scheme = 0;
		bw_byte(w, scheme);
		if (scheme == 2) qua_rle_store(b, w);
		else qua_position_store(b, w, scheme == 1);
		return ORC_OK;
	}
	if (b->cfg->lossy)
	{
		/* QualityOrderModelerProxyLossy: no scheme byte, no bitmap (src/QualityModelerProxy.h:151-154) */
		qua_order_encode(b, w, 8, qo, 8, NULL);
		return ORC_OK;
	}
	/* QualityOrderModelerProxyLossless::SelectSchemeId (src/QualityModelerProxy.h:257-283) */
	u32 scheme = 255;
	for (u32 i = 0; i < 8; ++i)
		if ((16u << i) >= s->q_count) { scheme = i; break; }
	if (scheme != 255 && qo == 2)
	{
		double ratio = (double)s->raw_len / (double)s->rle_len;
		if (s->max_len == s->min_len && ratio > 1.175) scheme += 4;
	}
	bw_byte(w, scheme);
	static const u32 N_[8] = {16, 32, 64, 128, 16, 32, 64, 128};
	static const u32 ORD1[4] = {3, 2, 1, 1}, ORD2[4] = {4, 3, 2, 1};
	if (scheme > 7 || (scheme > 3 && s->q_count > 128)) return ORC_E_UNSUPPORTED;   /* Appendix B.12 */
	u32 n = N_[scheme];
	u32 ord = (qo == 1) ? ORD1[scheme & 3] : ORD2[scheme & 3];
	u32 rescale = scheme < 4 ? 8 : n;
	/* TTranslationalQualityEncoder::Store (src/QualityEncoder.h:332-342): 32-byte presence map */
	bw_flush(w);
	for (u32 i = 0; i < 256; ++i) bw_bit(w, s->q_sym[i] != 255);
	qua_order_encode(b, w, n, ord, rescale, s->q_sym);
	return ORC_OK;
}

/* ------------------------------------------------------------------------
 * tags: src/TagModeler.h:31-145, src/TagModeler.cpp:159-884,1217-1284
 * ---------------------------------------------------------------------- */
#define VM_CAP 2048
typedef struct { i32 key[VM_CAP]; i32 val[VM_CAP]; u8 used[VM_CAP]; u32 size; } vmap_t;   /* std::map<int32,int32> */

static void vm_clear(vmap_t* m) { memset(m->used, 0, sizeof(m->used)); m->size = 0; }
static i32* vm_at(vmap_t* m, i32 k)
{
	u32 h = ((u32)k * 2654435761u) & (VM_CAP - 1);
	while (m->used[h] && m->key[h] != k) h = (h + 1) & (VM_CAP - 1);
	if (!m->used[h]) { m->used[h] = 1; m->key[h] = k; m->val[h] = 0; m->size++; }
	return &m->val[h];
}
static i32 vm_get(const vmap_t* m, i32 k)
{
	u32 h = ((u32)k * 2654435761u) & (VM_CAP - 1);
	while (m->used[h] && m->key[h] != k) h = (h + 1) & (VM_CAP - 1);
	return m->used[h] ? m->val[h] : 0;
}

typedef struct { i32 cur_sym; u32 cur_len, run_len; u8* lens; u32 n_lens, cap; } rle_t;
static void rle_push(rle_t* r, u32 v)
{
	if (r->n_lens == r->cap) { r->cap = r->cap ? r->cap * 2 : 256; r->lens = (u8*)realloc(r->lens, r->cap); }
	r->lens[r->n_lens++] = (u8)v;
}

enum { NS_NONE = 0, NS_VALUE_VAR, NS_VALUE_RLE, NS_DELTA_VAR, NS_DELTA_RLE, NS_DELTA_CONST };

typedef struct
{
	u32 len, min_len, max_len; u8 sep;
	int is_constant, is_len_constant, is_numeric;
	i32 min_value, max_value, min_delta, max_delta;
	u32 bits_num, bits_value, bits_len;
	int is_delta_coding, try_rle_val, try_rle_delta, is_delta_const, var_stat_encode;
	u8 scheme;
	rle_t rle_val, rle_delta;
	u8* data; u8* ham;
	vmap_t* num_values; vmap_t* delta_values;
	u32 (*chars)[256];           /* [129][256]: vector<map<char,uint32>> */
	huff_t* huf_global; huff_t* huf_local[129];
} field_t;

typedef struct
{
	field_t* f; u32 nf;
	u32 min_title, max_title;
	u32 sym_freq[256];           /* 128 in the reference; titles are 7-bit */
	int mixed;
	i32* prev; u32 rec_counter;
} tags_t;

static int is_sep(u8 c)
{
	return c == ' ' || c == '.' || c == '_' || c == ',' || c == '=' || c == ':' || c == '/' || c == '-' || c == '#' || c == 0;
}

/* TagAnalyzer::InitializeFieldsStats (src/TagModeler.cpp:159-222) */
static void tags_init(tags_t* t, const u8* title, u32 tl, u32* fields_cap)
{
	memset(t, 0, sizeof(*t));
	t->min_title = 0xFFFFFFFFu;
	t->f = (field_t*)calloc(tl + 2, sizeof(field_t));
	u32 start = 0;
	for (u32 i = 0; i <= tl; ++i)
	{
		t->sym_freq[title[i]] += (i != tl);
		if (!is_sep(title[i]) && i != tl) continue;
		field_t* f = &t->f[t->nf++];
		f->len = i - start; f->min_len = f->max_len = f->len;
		f->data = (u8*)malloc(f->len + 1); memcpy(f->data, title + start, f->len); f->data[f->len] = 0;
		f->sep = title[i];
		f->is_constant = 1; f->is_len_constant = 1;
		f->min_value = 1 << 30; f->max_value = -(1 << 30); f->min_delta = 1 << 30; f->max_delta = -(1 << 30);
		u32 v;
		f->is_numeric = is_num(f->data, f->len, &v);
		f->ham = (u8*)malloc(f->len + 1); memset(f->ham, 1, f->len + 1);
		f->num_values = (vmap_t*)malloc(sizeof(vmap_t)); vm_clear(f->num_values);
		f->delta_values = (vmap_t*)malloc(sizeof(vmap_t)); vm_clear(f->delta_values);
		f->chars = (u32(*)[256])calloc(129, sizeof(u32[256]));
		if (f->is_numeric)
		{
			f->min_value = f->max_value = (i32)v;
			(*vm_at(f->num_values, (i32)v))++;
		}
		start = i + 1;
	}
	/* std::vector<Field>::push_back reallocations copy Fields through Field(const Field&)
	 * (src/TagModeler.cpp:63-135), which drops num_values; the vector keeps its capacity
	 * across blocks of one BlockCompressor (TagStats::Reset only clear()s).  So record 0's
	 * extra num_values count survives only for fields pushed at/after the last reallocation. */
	{
		u32 cap = *fields_cap; i32 last = -1;
		for (u32 i = 0; i < t->nf; ++i)
			if (i == cap) { last = (i32)i; cap = cap ? cap * 2 : 1; }
		for (i32 i = 0; i < last; ++i) vm_clear(t->f[i].num_values);
		*fields_cap = cap;
	}
	t->prev = (i32*)calloc(t->nf + 1, sizeof(i32));
	t->rec_counter = 0;
}

/* TagAnalyzer::UpdateNumericField (src/TagModeler.cpp:341-459) */
static void tags_update_numeric(tags_t* t, field_t* f, i32 cur, i32 prev)
{
	if (cur < f->min_value) f->min_value = cur;
	else if (cur > f->max_value) f->max_value = cur;

	if (t->rec_counter > 0)
	{
		if (f->rle_val.cur_sym != cur)
		{
			f->rle_val.run_len++; f->rle_val.cur_sym = cur;
			rle_push(&f->rle_val, f->rle_val.cur_len); f->rle_val.cur_len = 0;
		}
		else
		{
			f->rle_val.cur_len++;
			if (f->rle_val.cur_len > 255) { rle_push(&f->rle_val, 255); f->rle_val.cur_len = 0; f->rle_val.run_len++; }
		}
		if (f->num_values->size)
		{
			(*vm_at(f->num_values, cur))++;
			if (f->num_values->size > 512) vm_clear(f->num_values);
		}
	}
	else
	{
		f->rle_val.cur_sym = cur; f->rle_val.cur_len = 0; f->rle_val.run_len = 0; f->rle_val.n_lens = 0;
		(*vm_at(f->num_values, cur))++;
	}

	if (t->rec_counter >= 1)
	{
		i32 d = (i32)((u32)cur - (u32)prev);
		if (t->rec_counter > 1)
		{
			if (d > f->max_delta) f->max_delta = d;
			else if (d < f->min_delta) f->min_delta = d;
			if (f->rle_delta.cur_sym != d)
			{
				f->rle_delta.run_len++; f->rle_delta.cur_sym = d;
				rle_push(&f->rle_delta, f->rle_delta.cur_len); f->rle_delta.cur_len = 0;
			}
			else
			{
				f->rle_delta.cur_len++;
				if (f->rle_delta.cur_len > 255) { rle_push(&f->rle_delta, 255); f->rle_delta.cur_len = 0; f->rle_delta.run_len++; }
			}
			if (f->delta_values->size)
			{
				(*vm_at(f->delta_values, d))++;
				if (f->delta_values->size > 512) vm_clear(f->delta_values);
			}
		}
		else
		{
			f->max_delta = f->min_delta = d;
			f->rle_delta.cur_sym = d; f->rle_delta.cur_len = 0; f->rle_delta.run_len = 0; f->rle_delta.n_lens = 0;
			(*vm_at(f->delta_values, d))++;
		}
	}
}

/* TagAnalyzer::UpdateFieldsStats (src/TagModeler.cpp:224-339) */
static void tags_update(tags_t* t, const u8* title, u32 tl)
{
	t->min_title = MINV(t->min_title, tl); t->max_title = MAXV(t->max_title, tl);
	if (t->mixed)
	{
		for (u32 i = 0; i < tl; ++i) t->sym_freq[title[i]]++;
		return;
	}
	u32 c = 0, start = 0, k;
	for (k = 0; k <= tl && c < t->nf; ++k)
	{
		t->sym_freq[title[k]] += (k != tl);
		if (title[k] != t->f[c].sep && k < tl) continue;
		field_t* f = &t->f[c];
		u32 L = k - start;
		if (L > f->max_len) f->max_len = L;
		else if (L < f->min_len) f->min_len = L;
		u32 cl = MINV(128u, L);
		for (u32 x = 0; x < cl; ++x) f->chars[x][title[start + x]]++;
		for (u32 x = 128; x < L; ++x) f->chars[128][title[start + x]]++;
		if (f->is_constant)
		{
			if (L != f->len) f->is_constant = 0;
			else f->is_constant = memcmp(f->data, title + start, f->len) == 0;
		}
		if (f->is_len_constant) f->is_len_constant = f->len == L;
		if (f->is_numeric)
		{
			u32 v;
			f->is_numeric = is_num(title + start, L, &v);
			if (f->is_numeric)
			{
				tags_update_numeric(t, f, (i32)v, t->prev[c]);
				t->prev[c] = (i32)v;
			}
		}
		if (!f->is_constant)
			for (u32 p = 0; p < L && p < f->len; ++p) f->ham[p] &= (f->data[p] == title[p + start]);
		start = k + 1; c++;
	}
	if (c != t->nf || k != tl + 1) t->mixed = 1;
	t->rec_counter++;
}

/* TagAnalyzer::FinalizeFieldsStats (src/TagModeler.cpp:461-551) */
static void tags_finalize(tags_t* t)
{
	if (t->mixed) return;
	for (u32 i = 0; i < t->nf; ++i)
	{
		field_t* f = &t->f[i];
		if (!f->is_numeric)
		{
			if (!f->is_constant) f->bits_len = bit_length(f->max_len - f->min_len);
			continue;
		}
		i32 diff;
		i32 dv = (i32)((u32)f->max_value - (u32)f->min_value);
		i32 dd = (i32)((u32)f->max_delta - (u32)f->min_delta);
		if (dv < dd) { f->is_delta_coding = 0; diff = dv; }
		else { f->is_delta_coding = 1; diff = dd; }

		rle_push(&f->rle_val, f->rle_val.cur_len);
		if (f->rle_val.cur_len > 0) { f->rle_val.cur_len = 0; f->rle_val.run_len++; }
		if ((float)t->rec_counter / (float)f->rle_val.run_len > 1.25f) f->try_rle_val = 1;

		if (f->is_delta_coding)
		{
			f->is_delta_const = diff == 0;
			if (!f->is_delta_const)
			{
				rle_push(&f->rle_delta, f->rle_delta.cur_len);
				if (f->rle_delta.cur_len > 0) { f->rle_delta.cur_len = 0; f->rle_delta.run_len++; }
				if ((float)t->rec_counter / (float)f->rle_delta.run_len > 1.25f) f->try_rle_delta = 1;
			}
		}
		if (f->is_delta_coding && f->is_delta_const) f->scheme = NS_DELTA_CONST;
		else if (f->is_delta_coding && f->try_rle_delta) f->scheme = NS_DELTA_RLE;
		else if (f->try_rle_val) f->scheme = NS_VALUE_RLE;
		else if (f->is_delta_coding)
		{
			f->scheme = NS_DELTA_VAR;
			u32 d = (u32)dd + 1;
			f->var_stat_encode = d <= 512 && f->delta_values->size;
		}
		else
		{
			f->scheme = NS_VALUE_VAR;
			u32 d = (u32)dv + 1;
			f->var_stat_encode = d <= 512 && f->num_values->size;
		}
		f->bits_num = bit_length((u64)(int64_t)diff);
		f->bits_value = bit_length((u64)(int64_t)dv);
	}
}

/* TagTokenizerEncoder::StoreFields (src/TagModeler.cpp:569-693), SURVEY A.2 */
static void tags_store_fields(tags_t* t, bw_t* w)
{
	bw_byte(w, t->nf & 0xFF);
	for (u32 i = 0; i < t->nf; ++i)
	{
		field_t* f = &t->f[i];
		bw_byte(w, f->sep);
		bw_byte(w, (u32)f->is_constant);
		if (f->is_constant)
		{
			bw_word(w, f->len);
			for (u32 j = 0; j < f->len; ++j) bw_byte(w, f->data[j]);
			continue;
		}
		bw_byte(w, (u32)f->is_numeric);
		if (f->is_numeric)
		{
			bw_byte(w, f->scheme);
			bw_word(w, (u32)f->min_value); bw_word(w, (u32)f->max_value);
			if (f->scheme == NS_DELTA_CONST || f->scheme == NS_DELTA_RLE || f->scheme == NS_DELTA_VAR)
			{
				bw_word(w, (u32)f->min_delta); bw_word(w, (u32)f->max_delta);
			}
			if (f->scheme == NS_DELTA_VAR || f->scheme == NS_VALUE_VAR)
			{
				bw_byte(w, (u32)f->var_stat_encode);
				if (f->var_stat_encode)
				{
					const int delta = f->scheme == NS_DELTA_VAR;
					u32 d = delta ? (u32)f->max_delta - (u32)f->min_delta : (u32)f->max_value - (u32)f->min_value;
					d++;
					u32 freqs[HUF_MAX];
					for (u32 j = 0; j < d; ++j)
						freqs[j] = (u32)(delta ? vm_get(f->delta_values, f->min_delta + (i32)j)
											   : vm_get(f->num_values, f->min_value + (i32)j));
					f->huf_global = (huff_t*)malloc(sizeof(huff_t));
					huff_build(f->huf_global, freqs, d);
					huff_store(f->huf_global, w);
				}
			}
			continue;
		}
		bw_byte(w, (u32)f->is_len_constant);
		bw_word(w, f->len); bw_word(w, f->max_len); bw_word(w, f->min_len);
		for (u32 j = 0; j < f->len; ++j) bw_byte(w, f->data[j]);
		for (u32 j = 0; j < f->len; ++j) bw_bit(w, f->ham[j]);
		bw_flush(w);
		for (u32 j = 0; j < MINV(f->max_len, 128u); ++j)
		{
			if (j >= f->len || !f->ham[j])
			{
				f->huf_local[j] = (huff_t*)malloc(sizeof(huff_t));
				huff_build(f->huf_local[j], f->chars[j], 256);
				huff_store(f->huf_local[j], w);
			}
		}
		if (f->max_len >= 128)
		{
			/* max_len == 128 exactly indexes past the resized vector in the reference (UB); we read zeros */
			f->huf_local[128] = (huff_t*)malloc(sizeof(huff_t));
			huff_build(f->huf_local[128], f->chars[128], 256);
			huff_store(f->huf_local[128], w);
		}
	}
}

/* TagTokenizerEncoder::StoreNumericField (src/TagModeler.cpp:753-874) */
static void tags_store_numeric(tags_t* t, bw_t* w, field_t* f, i32 cur, i32 prev)
{
	if (t->rec_counter == 0)
	{
		i32 dval = (i32)((u32)cur - (u32)f->min_value);
		bw_bits(w, (u32)dval, f->bits_value);
		if (f->scheme == NS_VALUE_RLE)
		{
			f->rle_val.run_len = 0; f->rle_val.cur_len = f->rle_val.lens[0]; f->rle_val.cur_sym = dval;
			bw_bits(w, f->rle_val.cur_len, 8);
		}
		return;
	}
	switch (f->scheme)
	{
	case NS_DELTA_CONST: break;
	case NS_DELTA_RLE:
	{
		i32 dval = (i32)((u32)cur - (u32)prev - (u32)f->min_delta);
		if (t->rec_counter == 1 || f->rle_delta.cur_len == 0)
		{
			if (t->rec_counter == 1) f->rle_delta.run_len = 0; else f->rle_delta.run_len++;
			f->rle_delta.cur_len = f->rle_delta.lens[f->rle_delta.run_len];
			f->rle_delta.cur_sym = dval;
			bw_bits(w, (u32)dval, f->bits_num);
			bw_bits(w, f->rle_delta.cur_len, 8);
		}
		else f->rle_delta.cur_len--;
		break;
	}
	case NS_DELTA_VAR:
	{
		i32 v = (i32)((u32)cur - (u32)prev - (u32)f->min_delta);
		if (f->huf_global) bw_bits(w, f->huf_global->code[v], f->huf_global->len[v]);
		else bw_bits(w, (u32)v, f->bits_num);
		break;
	}
	case NS_VALUE_RLE:
	{
		i32 dval = (i32)((u32)cur - (u32)f->min_value);
		if (f->rle_val.cur_len == 0)
		{
			f->rle_val.run_len++;
			f->rle_val.cur_len = f->rle_val.lens[f->rle_val.run_len];
			f->rle_val.cur_sym = dval;
			bw_bits(w, (u32)dval, f->bits_value);
			bw_bits(w, f->rle_val.cur_len, 8);
		}
		else f->rle_val.cur_len--;
		break;
	}
	case NS_VALUE_VAR:
	{
		i32 v = (i32)((u32)cur - (u32)f->min_value);
		if (f->huf_global) bw_bits(w, f->huf_global->code[v], f->huf_global->len[v]);
		else bw_bits(w, (u32)v, f->bits_num);
		break;
	}
	default: break;
	}
}

/* TagTokenizerEncoder::EncodeNextFields (src/TagModeler.cpp:695-751) */
static void tags_encode_record(tags_t* t, bw_t* w, const u8* title, u32 tl)
{
	u32 c = 0, start = 0;
	for (u32 k = 0; k <= tl; ++k)
	{
		if (c >= t->nf) break;                  /* cannot happen for non-mixed blocks */
		field_t* f = &t->f[c];
		if (title[k] != f->sep && k < tl) continue;
		if (f->is_constant) { start = k + 1; c++; continue; }
		if (f->is_numeric)
		{
			i32 v = (i32)to_num(title + start, k - start);
			tags_store_numeric(t, w, f, v, t->prev[c]);
			t->prev[c] = v; start = k + 1; c++;
			continue;
		}
		u32 L = k - start;
		if (!f->is_len_constant) bw_bits(w, L - f->min_len, f->bits_len);
		for (u32 j = 0; j < L; ++j)
		{
			if (j >= f->len || !f->ham[j])
			{
				u8 ch = title[start + j];
				const huff_t* h = f->huf_local[MINV(j, 128u)];
				bw_bits(w, h->code[ch], h->len[ch]);
			}
		}
		start = k + 1; c++;
	}
	t->rec_counter++;
}

static void tags_free(tags_t* t)
{
	for (u32 i = 0; i < t->nf; ++i)
	{
		field_t* f = &t->f[i];
		free(f->data); free(f->ham); free(f->num_values); free(f->delta_values); free(f->chars);
		free(f->rle_val.lens); free(f->rle_delta.lens); free(f->huf_global);
		for (u32 j = 0; j < 129; ++j) free(f->huf_local[j]);
	}
	free(t->f); free(t->prev);
}

/* AnalyzeTags (src/BlockCompressor.cpp:359-400) */
static void tags_analyze(block_t* b)
{
	tags_t* t = (tags_t*)malloc(sizeof(tags_t));
	const rec_t* r0 = &b->recs[0];
	tags_init(t, b->mem + r0->title, r0->title_len, &b->fields_cap);
	for (u64 k = 0; k < b->n_recs; ++k) tags_update(t, b->mem + b->recs[k].title, b->recs[k].title_len);
	tags_finalize(t);
	if (t->mixed) b->flags |= 4;
	b->tags = t;
}

/* StoreTags (src/BlockCompressor.cpp:458-488) */
static void tags_emit(block_t* b, bw_t* w)
{
	tags_t* t = (tags_t*)b->tags;
	const u32 len_bits = bit_length((u64)(u16)(b->max_qlen - b->min_qlen));
	if (!t->mixed)
	{
		tags_store_fields(t, w);
		t->rec_counter = 0;
		memset(t->prev, 0, sizeof(i32) * (t->nf + 1));
		for (u64 k = 0; k < b->n_recs; ++k)
		{
			const rec_t* r = &b->recs[k];
			tags_encode_record(t, w, b->mem + r->title, r->title_len);
			if (len_bits) bw_bits(w, (u32)(r->qual_len - b->min_qlen), len_bits);
		}
		bw_flush(w);
	}
	else
	{
		/* TagRawEncoder (src/TagModeler.cpp:1217-1284) */
		u32 tl_bits = bit_length(t->max_title - t->min_title);
		bw_word(w, t->min_title); bw_word(w, t->max_title);
		i32 syms[128]; u32 freqs[128], n = 0;
		for (u32 i = 0; i < 128; ++i)
		{
			syms[i] = -1;
			if (t->sym_freq[i] > 0) { syms[i] = (i32)n; freqs[n++] = t->sym_freq[i]; }
		}
		huff_t* h = (huff_t*)malloc(sizeof(huff_t));
		huff_build(h, freqs, n);
		for (u32 i = 0; i < 128; ++i) bw_bit(w, syms[i] != -1);
		bw_flush(w);
		huff_store(h, w);
		for (u64 k = 0; k < b->n_recs; ++k)
		{
			const rec_t* r = &b->recs[k];
			if (tl_bits) bw_bits(w, r->title_len - t->min_title, tl_bits);
			for (u32 i = 0; i < r->title_len; ++i)
			{
				i32 x = syms[b->mem[r->title + i] & 127];
				if (x >= 0) bw_bits(w, h->code[x], h->len[x]);   /* x < 0 is reference UB (Appendix B.10) */
			}
			if (len_bits) bw_bits(w, (u32)(r->qual_len - b->min_qlen), len_bits);
		}
		bw_flush(w);
		free(h);
	}
	tags_free(t);
	free(t);
}

/* ------------------------------------------------------------------------
 * block = meta . tags . quality . dna (src/BlockCompressor.cpp:208-259,403-443)
 * ---------------------------------------------------------------------- */
static int block_run(const orc_config* cfg, const u8* in, u64 size, bw_t* w, u64 raw[4], u64 comp[4], block_t* keep, u32* fields_cap)
{
	block_t b;
	memset(&b, 0, sizeof(b));
	b.cfg = cfg;
	b.fields_cap = fields_cap ? *fields_cap : 0;
	b.mem = (u8*)malloc(size + 16);
	memcpy(b.mem, in, size);
	memset(b.mem + size, '\n', 16);
	b.size = size;
	if (cfg->calc_crc32)
		b.crc_flags = (cfg->tag_preserve_flags ? 0 : 1) | 2 | (cfg->lossy ? 0 : 4);      /* src/BlockCompressor.cpp:84-93 */

	block_parse(&b);
	if (b.n_recs == 0) { free(b.mem); free(b.recs); return ORC_E_INPUT; }
	block_preprocess(&b);
	if (keep) { *keep = b; return ORC_OK; }
	if (g_chunk_size_set)
	{
		/* BlockCompressorExt::InsertNewRecord stores tag, sequence and quality back to back, so the byte the tag
		 * tokenizer takes for the last field's separator (title[titleLen]) is sequence[0] as ProcessForward left it. */
		for (u64 i = 0; i < b.n_recs; ++i) b.mem[b.recs[i].title + b.recs[i].title_len] = b.mem[b.recs[i].seq];
	}

	/* AnalyzeMetaData (src/BlockCompressor.cpp:184-205) */
	b.max_qlen = (u16)b.st.max_len; b.min_qlen = (u16)b.st.min_len;
	if (b.max_qlen != b.min_qlen) b.flags |= 2;
	const int cs_reduce = cfg->color_space && b.cs_const;
	if (cs_reduce)
	{
		/* src/BlockCompressor.cpp:190-199: the begin symbols are read AFTER ProcessForward, i.e. base index and
		 * quality - offset (or its hoisted code) of record 0 */
		b.flags |= 1;
		b.cs_seq_begin = b.mem[b.recs[0].seq]; b.cs_qua_begin = b.mem[b.recs[0].qual];
		b.max_qlen = (u16)(b.max_qlen - 1); b.min_qlen = (u16)(b.min_qlen - 1);
	}
	tags_analyze(&b);
	if (cs_reduce)
	{
		/* AnalyzeTags, "2nd pass" (src/BlockCompressor.cpp:380-393): every record loses its first (kept) base and its
		 * first quality; the statistics above were taken with them */
		for (u64 k = 0; k < b.n_recs; ++k)
		{
			rec_t* r = &b.recs[k];
			if (r->seq_len < 1 || r->qual_len < 2) g_ref_ub = 1;    /* ASSERTs of the reference; lengths wrap in a release build */
			r->seq++; r->qual++; r->qual_len--; r->seq_len--;
			if (r->trunc_len > 0) r->trunc_len--;
		}
	}
	if (fields_cap) *fields_cap = b.fields_cap;

	u64 pos = w->pos;
	bw_word(w, (u32)b.n_recs); bw_word(w, b.max_qlen); bw_word(w, b.flags); bw_word(w, (u32)b.chunk_size);
	if (b.flags & 2) bw_word(w, b.min_qlen);
	if (cfg->color_space && (b.flags & 1)) { bw_byte(w, b.cs_seq_begin); bw_byte(w, b.cs_qua_begin); }   /* src/BlockCompressor.cpp:415-422 */
	if (cfg->calc_crc32)
	{
		if (b.crc_flags & 1) bw_word(w, b.crc_tag);
		bw_word(w, b.crc_seq);
		if (!cfg->lossy) bw_word(w, b.crc_qual);
	}
	bw_flush(w);
	comp[0] = w->pos - pos; pos = w->pos;

	tags_emit(&b, w);
	comp[1] = w->pos - pos; pos = w->pos;

	int rcode = qua_store(&b, w);
	comp[3] = w->pos - pos; pos = w->pos;

	dna_store(&b, w);
	comp[2] = w->pos - pos;

	for (int i = 0; i < 4; ++i) raw[i] = b.raw[i];
	free(b.mem); free(b.recs);
	return rcode;
}

int orc_compress_block_state(const orc_config* cfg, uint32_t* fields_cap, const uint8_t* in, uint64_t size,
							 uint8_t* out, uint64_t cap, uint64_t* out_size, uint64_t raw[4], uint64_t comp[4])
{
	bw_t w; bw_init(&w, size / 2 + 4096);
	g_ref_ub = 0;
	int rc = block_run(cfg, in, size, &w, raw, comp, NULL, fields_cap);
	if (rc == ORC_OK && g_ref_ub) rc = ORC_E_UNSUPPORTED;
	bw_flush(&w);
	*out_size = w.pos;
	if (rc == ORC_OK)
	{
		if (w.pos > cap) rc = ORC_E_CAP;
		else memcpy(out, w.buf, w.pos);
	}
	bw_free(&w);
	return rc;
}

int orc_compress_block(const orc_config* cfg, const uint8_t* in, uint64_t size,
					   uint8_t* out, uint64_t cap, uint64_t* out_size, uint64_t raw[4], uint64_t comp[4])
{
	uint32_t fresh = 0;
	return orc_compress_block_state(cfg, &fresh, in, size, out, cap, out_size, raw, comp);
}

int orc_block_stats(const orc_config* cfg, const uint8_t* in, uint64_t size,
					uint32_t* d_out, uint32_t* q_out, uint64_t* recs, uint64_t* chunk_size, uint64_t raw[4])
{
	block_t b; u64 comp[4];
	int rc = block_run(cfg, in, size, NULL, raw, comp, &b, NULL);
	if (rc != ORC_OK) return rc;
	d_out[0] = b.st.d_count; for (int i = 0; i < 20; ++i) d_out[1 + i] = b.st.d_freq[i];
	q_out[0] = b.st.q_count; q_out[1] = b.st.min_len; q_out[2] = b.st.max_len;
	q_out[3] = b.st.raw_len; q_out[4] = b.st.th_len; q_out[5] = b.st.rle_len;
	for (int i = 0; i < 256; ++i) q_out[6 + i] = b.st.q_freq[i];
	*recs = b.n_recs; *chunk_size = b.chunk_size;
	for (int i = 0; i < 4; ++i) raw[i] = b.raw[i];
	free(b.mem); free(b.recs);
	return ORC_OK;
}

/* ------------------------------------------------------------------------
 * FastqParser::Analyze (src/FastqParser.cpp:27-138)
 * ---------------------------------------------------------------------- */
int orc_analyze(const uint8_t* in, uint64_t size, int estimate, uint32_t* qoff, int32_t* plus_rep, int32_t* cs)
{
	parser_t p = { (u8*)in, size, 0, 0 };
	u8 minq = 255, maxq = 0;
	int color = 0, plus = 0;
	u32 n = 0;
	while (p.pos < p.size)
	{
		const u8* title = in + p.pos; u32 tl = p_skip_line(&p);
		if (tl == 0 || title[0] != '@') break;
		const u8* seq = in + p.pos; u32 sl = p_skip_line(&p);
		if (sl == 0) break;
		const u8* pl = in + p.pos; int prep = p_skip_line(&p) > 1;
		if (pl[0] != '+') break;
		if (estimate)
		{
			const u8* q = in + p.pos; u32 ql = p_skip_line(&p);
			for (u32 i = 0; i < ql; ++i) { minq = MINV(minq, q[i]); maxq = MAXV(maxq, q[i]); }
		}
		else if (p_skip_line(&p) == 0) break;
		int cenc = (seq[1] >= '0' && seq[1] <= '3') || seq[1] == '.';
		if (n != 0)
		{
			if (color != cenc) return -1;
			if (color && seq[0] >= '0' && seq[0] <= '3') return -1;
			if (plus != prep) return -1;
		}
		else { plus = prep; color = cenc; }
		n++;
	}
	*plus_rep = plus; *cs = color;
	if (estimate)
	{
		if (maxq <= 74) { if (minq >= 33) *qoff = 33; }
		else if (maxq <= 105) { if (minq >= 64) *qoff = 64; else if (minq >= 59) *qoff = 59; }
		if (*qoff == 0) { if (minq >= 33) *qoff = 33; else return -1; }
	}
	return n > 1 ? 0 : -1;
}

/* ------------------------------------------------------------------------
 * chunk cutter: IFastqStreamReader::ReadNextChunk / GetNextRecordPos / SkipToEol
 * (src/FastqStream.cpp:18-98, src/FastqStream.h:74-89) over an in-memory file
 * ---------------------------------------------------------------------- */
static void skip_to_eol(const u8* d, u64* pos, u64 size, int* crlf)
{
	while (*pos < size && d[*pos] != '\n' && d[*pos] != '\r') ++*pos;
	if (*pos < size && d[*pos] == '\r')
		if (*pos + 1 < size && d[*pos + 1] == '\n') { *crlf = 1; ++*pos; }
}

static u64 next_record_pos(const u8* d, u64 pos, u64 size, int* crlf)
{
	skip_to_eol(d, &pos, size, crlf); ++pos;
	while (pos < size && d[pos] != '@') { skip_to_eol(d, &pos, size, crlf); ++pos; }
	u64 pos0 = pos;
	skip_to_eol(d, &pos, size, crlf); ++pos;
	if (pos < size && d[pos] == '@') return pos;
	skip_to_eol(d, &pos, size, crlf); ++pos;
	return pos0;
}

int64_t orc_cut_chunks(const uint8_t* file, uint64_t file_size, uint64_t buf_size,
					   uint64_t* starts, uint64_t* sizes, uint64_t cap)
{
	u64 start = 0, carry = 0, rd = 0;   /* rd = file bytes consumed by Read() so far */
	int eof = 0, crlf = 0;
	int64_t n = 0;
	while (!eof)
	{
		u64 to_read = buf_size - carry;
		u64 r = MINV(to_read, file_size - rd);
		u64 size = carry;
		if (r > 0)
		{
			rd += r;
			if (r == to_read)
			{
				const u8* d = file + start;
				u64 end = next_record_pos(d, buf_size - 8192, buf_size, &crlf);
				size = end - 1; if (crlf) size -= 1;
				if ((u64)n < cap) { starts[n] = start; sizes[n] = size; }
				n++;
				carry = buf_size - end; start += end;
				continue;
			}
			size = carry + r - 1; if (crlf) size -= 1;
			eof = 1;
		}
		else eof = 1;
		if ((u64)n < cap) { starts[n] = start; sizes[n] = size; }
		n++;
	}
	return n;
}

/* ------------------------------------------------------------------------
 * archive: DsrcFileWriter::WriteFileHeader/WriteFileFooter (src/DsrcFile.cpp:112-170)
 * ---------------------------------------------------------------------- */
static void be32(u8* p, u32 v) { p[0] = (u8)(v >> 24); p[1] = (u8)(v >> 16); p[2] = (u8)(v >> 8); p[3] = (u8)v; }
static void be64(u8* p, u64 v) { be32(p, (u32)(v >> 32)); be32(p + 4, (u32)v); }

uint64_t orc_archive_header(uint8_t out[40], uint64_t footer_offset, uint32_t footer_size, uint64_t block_count)
{
	out[0] = 0xAA; out[1] = 2; out[2] = 0; out[3] = 2;
	be32(out + 4, footer_size); be64(out + 8, footer_offset); be64(out + 16, 0); be64(out + 24, block_count);
	memset(out + 32, 0xAA, 8);
	return 40;
}

uint64_t orc_archive_footer(uint8_t* out, const uint32_t* block_sizes, uint64_t block_count, const orc_config* cfg)
{
	u64 p = 0;
	out[p++] = 0xCC;
	memcpy(out + p, block_sizes, block_count * 4); p += block_count * 4;   /* host-endian memcpy in the reference */
	out[p++] = (u8)((cfg->color_space ? 2 : 0) | (cfg->plus_repetition ? 1 : 0));
	out[p++] = (u8)cfg->quality_offset;
	out[p++] = (u8)((cfg->lossy ? 1 : 0) | (cfg->calc_crc32 ? 2 : 0));
	out[p++] = (u8)cfg->dna_order; out[p++] = (u8)cfg->quality_order;
	be64(out + p, cfg->tag_preserve_flags); p += 8;
	return p;
}

int orc_compress_file(const char* in_path, const char* out_path, uint32_t dna_level, uint32_t quality_level,
					  int lossy, int crc, uint32_t qoff, uint32_t buf_mb)
{
	FILE* fi = fopen(in_path, "rb");
	if (!fi) return ORC_E_IO;
	fseeko(fi, 0, SEEK_END); u64 fsz = (u64)ftello(fi); fseeko(fi, 0, SEEK_SET);
	u8* file = (u8*)malloc(fsz + 16);
	if (fread(file, 1, fsz, fi) != fsz) { fclose(fi); free(file); return ORC_E_IO; }
	fclose(fi);
	memset(file + fsz, 0, 16);

	u64 buf = (u64)buf_mb << 20;
	u64 cap = fsz / (buf > 8192 ? buf - 8192 : 1) + 8;
	u64* starts = (u64*)malloc(cap * 8); u64* sizes = (u64*)malloc(cap * 8);
	int64_t n = orc_cut_chunks(file, fsz, buf, starts, sizes, cap);

	orc_config cfg; memset(&cfg, 0, sizeof(cfg));
	cfg.dna_order = dna_level * 3;
	cfg.quality_order = lossy ? quality_level * 3 : quality_level;   /* src/DsrcOperator.h:74-90 */
	cfg.lossy = lossy; cfg.calc_crc32 = crc; cfg.quality_offset = qoff;
	int rc = orc_analyze(file + starts[0], sizes[0], qoff == 0, &cfg.quality_offset, &cfg.plus_repetition, &cfg.color_space);
	if (rc != 0) { free(file); free(starts); free(sizes); return ORC_E_INPUT; }

	FILE* fo = fopen(out_path, "wb");
	if (!fo) { free(file); free(starts); free(sizes); return ORC_E_IO; }
	u8 hdr[40]; memset(hdr, 0, 40); fwrite(hdr, 1, 40, fo);
	u32* bsz = (u32*)malloc((size_t)n * 4);
	u64 off = 40;
	u32 fields_cap = 0;                       /* one BlockCompressor for the whole file (-t1 behaviour) */
	for (int64_t i = 0; i < n && rc == 0; ++i)
	{
		u64 ocap = sizes[i] + (1 << 16), osz = 0, raw[4], comp[4];
		u8* out = (u8*)malloc(ocap);
		rc = orc_compress_block_state(&cfg, &fields_cap, file + starts[i], sizes[i], out, ocap, &osz, raw, comp);
		if (rc == 0) { fwrite(out, 1, osz, fo); bsz[i] = (u32)osz; off += osz; }
		free(out);
	}
	if (rc == 0)
	{
		u8* foot = (u8*)malloc((size_t)n * 4 + 32);
		u64 fs = orc_archive_footer(foot, bsz, (u64)n, &cfg);
		fwrite(foot, 1, fs, fo);
		orc_archive_header(hdr, off, (u32)fs, (u64)n);
		fseeko(fo, 0, SEEK_SET); fwrite(hdr, 1, 40, fo);
		free(foot);
	}
	fclose(fo);
	free(bsz); free(file); free(starts); free(sizes);
	return rc;
}

/* ------------------------------------------------------------------------
 * Record-level API: wrap::DsrcArchive::StartCompress / WriteNextRecord / FinishCompress over
 * wrap::BlockCompressorExt (src/DsrcArchive.cpp:33-47,100-150,217-224, src/BlockCompressorExt.cpp:20-46,65-127),
 * fed by wrap::FastqFile::ReadNextRecord (src/FastqFile.cpp:66-92: strings up to '\n', an empty string ends the file).
 *  - a chunk is flushed once the title+sequence+quality bytes held exceed bufferMB << 20 (checked after every record);
 *  - chunkHeader.chunkSize grows by (tag+1)+(seq+1)+(plus+1)+(qual+1) per record and is NOT cleared by Reset()
 *    (src/BlockCompressor.cpp:105-109), so block k stores the running total over blocks 0..k (mod 2^32);
 *  - settings: dnaOrder = 3*level, qualityOrder = 3*level also when lossless (src/DsrcArchive.cpp:41-42), no CRC,
 *    no field filter, dataset type straight from the setters (no Analyze);
 *  - Flush = PreprocessRecords + AnalyzeRecords + StoreRecords on one persistent BlockCompressor.
 * ---------------------------------------------------------------------- */
int orc_compress_records_block(const orc_config* cfg, uint32_t* fields_cap, uint32_t chunk_size, const uint8_t* in, uint64_t size,
							   uint8_t* out, uint64_t cap, uint64_t* out_size, uint64_t raw[4], uint64_t comp[4])
{
	g_chunk_size_set = 1; g_chunk_size_value = chunk_size;
	const int rc = orc_compress_block_state(cfg, fields_cap, in, size, out, cap, out_size, raw, comp);
	g_chunk_size_set = 0;
	return rc;
}

int orc_compress_records_file(const char* in_path, const char* out_path, uint32_t dna_level, uint32_t quality_level,
							  int lossy, uint32_t qoff, uint32_t buf_mb, int plus_rep)
{
	FILE* fi = fopen(in_path, "rb");
	if (!fi) return ORC_E_IO;
	fseeko(fi, 0, SEEK_END); u64 fsz = (u64)ftello(fi); fseeko(fi, 0, SEEK_SET);
	u8* file = (u8*)malloc(fsz + 16);
	if (fread(file, 1, fsz, fi) != fsz) { fclose(fi); free(file); return ORC_E_IO; }
	fclose(fi);
	orc_config cfg; memset(&cfg, 0, sizeof(cfg));
	cfg.dna_order = dna_level * 3; cfg.quality_order = quality_level * 3;
	cfg.lossy = lossy; cfg.quality_offset = qoff; cfg.plus_repetition = plus_rep;
	FILE* fo = fopen(out_path, "wb");
	if (!fo) { free(file); return ORC_E_IO; }
	u8 hdr[40]; memset(hdr, 0, 40); fwrite(hdr, 1, 40, fo);
	u64 bcap = 64, nb = 0; u32* bsz = (u32*)malloc(bcap * 4);
	u64 off = 40, total = 0, payload = 0, chunk_start = 0, pos = 0;
	u32 fields_cap = 0;
	int rc = 0, done = 0;
	const u64 buf = (u64)buf_mb << 20;
	while (!done && rc == 0)
	{
		/* one record = four strings; ReadNextRecord fails on the first empty one */
		u64 len[4], p = pos; int ok = 1;
		for (int k = 0; k < 4 && ok; ++k)
		{
			u64 e = p; while (e < fsz && file[e] != '\n') e++;
			len[k] = e - p; p = e < fsz ? e + 1 : e;
			if (len[k] == 0) ok = 0;
		}
		int flush = 0;
		if (ok)
		{
			pos = p; payload += len[0] + len[1] + len[3]; total += len[0] + len[1] + len[2] + len[3] + 4;
			flush = payload > buf;
		}
		else { done = 1; flush = payload > 0; }
		if (flush)
		{
			/* the chunk as FASTQ text without its last newline = what Store() would be given */
			u64 csz = pos - chunk_start; if (csz && file[chunk_start + csz - 1] == '\n') csz--;
			u64 ocap = csz + (1 << 16), osz = 0, raw[4], comp[4];
			u8* out = (u8*)malloc(ocap);
			rc = orc_compress_records_block(&cfg, &fields_cap, (u32)total, file + chunk_start, csz, out, ocap, &osz, raw, comp);
			if (rc == 0)
			{
				fwrite(out, 1, osz, fo);
				if (nb == bcap) { bcap *= 2; bsz = (u32*)realloc(bsz, bcap * 4); }
				bsz[nb++] = (u32)osz; off += osz;
			}
			free(out);
			chunk_start = pos; payload = 0;
		}
	}
	if (rc == 0)
	{
		u8* foot = (u8*)malloc((size_t)nb * 4 + 32);
		u64 fs = orc_archive_footer(foot, bsz, nb, &cfg);
		fwrite(foot, 1, fs, fo);
		orc_archive_header(hdr, off, (u32)fs, nb);
		fseeko(fo, 0, SEEK_SET); fwrite(hdr, 1, 40, fo);
		free(foot);
	}
	fclose(fo); free(bsz); free(file);
	return rc;
}

/* ------------------------------------------------------------------------
 * primitive probes
 * ---------------------------------------------------------------------- */
uint64_t orc_bitwriter_script(const uint32_t* ops, uint32_t nops, uint8_t* out, uint64_t cap)
{
	bw_t w; bw_init(&w, 256);
	for (u32 i = 0; i < nops; ++i)
	{
		u32 k = ops[3 * i], a = ops[3 * i + 1], b = ops[3 * i + 2];
		switch (k)
		{
		case 0: bw_bit(&w, a); break;
		case 1: bw_bits(&w, a & 3, 2); break;
		case 2: bw_bits(&w, a, b); break;
		case 3: bw_byte(&w, a); break;
		case 4: bw_word(&w, a); break;
		case 5: bw_flush(&w); break;
		}
	}
	bw_flush(&w);
	u64 n = w.pos;
	memcpy(out, w.buf, MINV(n, cap));
	bw_free(&w);
	return n;
}

uint64_t orc_huffman(const uint32_t* freqs, uint32_t n, uint32_t* codes, uint32_t* lens, uint8_t* tree, uint64_t cap)
{
	huff_t* h = (huff_t*)malloc(sizeof(huff_t));
	huff_build(h, freqs, n);
	for (u32 i = 0; i < n; ++i) { codes[i] = h->code[i]; lens[i] = h->len[i]; }
	bw_t w; bw_init(&w, 256);
	huff_store(h, &w);
	bw_flush(&w);
	u64 sz = w.pos;
	memcpy(tree, w.buf, MINV(sz, cap));
	bw_free(&w); free(h);
	return sz;
}

uint64_t orc_rc_script(const uint32_t* fct, uint32_t n, uint8_t* out, uint64_t cap)
{
	bw_t w; bw_init(&w, 256);
	rc_t rc; rc_start(&rc, &w);
	for (u32 i = 0; i < n; ++i) rc_encode(&rc, fct[3 * i], fct[3 * i + 1], fct[3 * i + 2]);
	rc_end(&rc);
	bw_flush(&w);
	u64 sz = w.pos;
	memcpy(out, w.buf, MINV(sz, cap));
	bw_free(&w);
	return sz;
}

uint64_t orc_rc_adaptive4(const uint8_t* syms, uint32_t n, uint8_t* out, uint64_t cap)
{
	bw_t w; bw_init(&w, 256);
	rc_t rc; rc_start(&rc, &w);
	u16 row[4] = {1, 1, 1, 1};
	for (u32 i = 0; i < n; ++i) row_encode(row, 4, &rc, syms[i]);
	rc_end(&rc);
	bw_flush(&w);
	u64 sz = w.pos;
	memcpy(out, w.buf, MINV(sz, cap));
	bw_free(&w);
	return sz;
}
