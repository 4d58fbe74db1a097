// Block decompression: one DSRC block in -> the FASTQ text of its chunk out.
// Replaces BlockCompressor::Read / ReadRecords / ReadMetaData / ReadTags / ReadQuality / ReadDNA / VerifyChecksum
// (src/BlockCompressor.cpp:262-356,491-594), the modelers' Decode methods, RangeDecoder (src/RangeCoder.h:90-142),
// HuffmanEncoder::LoadTree/Decode (src/huffman.cpp:225-262, src/huffman.h:110-177) and ProcessBackward
// (src/RecordsProcessor.cpp:269-315,410-454).
//
// The three sub-streams of a block are not delimited: where the quality stream starts is known when the last title
// has been decoded, where the DNA stream starts when the last quality has.  Inside a stream every symbol depends on the
// ones before it (variable-length codes; adaptive contexts made of decoded symbols), so a block is three serial chains
// back to back and the parallelism is ACROSS blocks: one wavefront owns one block.  Lane 0 walks the chain; the other
// lanes clear model tables and, in the kernels that follow, everything that is per record once the symbols exist
// (N re-insertion, index -> character, line layout, colour space, CRC) is data-parallel again.
#pragma once
#include "k_common.h"
#include "k_parse.h"
#include "k_block.h"

typedef u64 __attribute__((aligned(1))) dec_u64_unaligned;

enum
{
	DEC_ERR_TRUNC    = 1u << 0,   // the decoder ran off the end of the block (the reference reads stale memory there)
	DEC_ERR_FORMAT   = 1u << 1,   // impossible header / tree / scheme id
	DEC_ERR_TEXT     = 1u << 2,   // decoded text does not fit the chunk size the block declares
	DEC_ERR_REF_UB   = 1u << 3,   // defined bytes, undefined behaviour in the reference's decoder (colour space without a constant primer, ...)
	DEC_ERR_POOL     = 1u << 4,   // scratch pool too small (host sizing bug)
};

#define DEC_STACK_SLACK 1024u      // node-pool words kept free behind the trees: parse stack of huff_load

struct DecParams
{
	u32 dna_order, quality_order, lossy, crc, quality_offset;
	u32 n_blocks, tag_flags, plus_rep, color_space;
	u32 table_words;           // u32 words per model-table slot
	u32 serial_quality;        // 1: the order-context quality decoder runs on one lane (DSRC_GPU_DEC_SERIAL; tests, comparison)
};

struct DecDesc          // host -> device
{
	u64 in_off;  u32 in_size;  u32 out_cap;
	u64 out_off;
	u32 rec_base, rec_cap;
	u64 d_base;             // decoded base indices, compact (bytes)
	u64 node_off;           // u32 words: Huffman nodes + tree directories of the tag stream
	u64 qnode_off;          // u32 words: trees of the quality / DNA stream
	u32 node_cap, qnode_cap;
	u64 fld_off;            // DecField[n_fields] (bytes)
};

struct DecState         // device -> host, and device scratch between the stages
{
	u32 err;
	u32 n_recs, max_qlen, min_qlen, flags, chunk_size;
	u32 cs_seq_begin, cs_qua_begin;
	u32 crc_stored[3], crc_actual[3];      // tag, sequence, quality
	u32 tag_pos, qua_pos, dna_pos, end_pos;        // byte positions inside the block
	u32 tag_nodes, n_fields;                // what the tag header needs from the node pool
	u32 text_bytes;                         // laid-out text
	u32 q_total, d_total;
	u32 q_scheme, d_scheme;
	u32 q_cnt;                              // order-context quality: symbols present (the alphabet's presence bitmap)
	u32 q_done;                             // -q0: k_dec_qpos (k_dec_q0.h) has decoded the quality stream, k_dec_qhuff leaves the block alone
	u32 pad[1];
};

struct DecField
{
	u8  sep, is_constant, is_numeric, is_len_constant, scheme, var_stat, has_global, pad0;
	u32 len, max_len, min_len;
	u32 data_pos;           // byte position of the constant text / string template inside the block
	u32 ham_bit;            // bit position of the hamming mask
	i32 min_value, max_value, min_delta, max_delta;
	u32 bits_value, bits_num, bits_len;
	u32 global_tree;        // node-pool index of the numeric dictionary tree
	u32 local_dir;          // node-pool index of 129 tree indices (0xFFFFFFFF = no tree at that position)
	u32 rle_len, rle_sym, prev;
};

// ---- bit source: BitMemoryReader (src/BitMemory.h:29-213) ---------------------------------------------------------------
// Every call site of the reference reads whole bytes only at byte boundaries, so the reader is a bit position into the
// block; FlushInputWordBuffer = round up to a byte.  Bits past the end read as zero and raise DEC_ERR_TRUNC.
struct BitSrc
{
	const u8* p; u32 size; u32 err;
	u64 bit;
};

__device__ __forceinline__ u32 bs_peek32(const BitSrc& s)
{
	const u64 by = s.bit >> 3; const u32 sh = (u32)s.bit & 7u;
	u64 v;
	if (by + 8 <= s.size) v = __builtin_bswap64(*(const dec_u64_unaligned*)(s.p + by));
	else
	{
		v = 0;
		for (u32 k = 0; k < 8; ++k) v = (v << 8) | (by + k < s.size ? (u64)s.p[by + k] : 0ull);
	}
	return (u32)((v << sh) >> 32);
}
__device__ __forceinline__ void bs_skip(BitSrc& s, u32 n)
{
	s.bit += n;
	if (s.bit > (u64)s.size * 8) s.err |= DEC_ERR_TRUNC;
}
__device__ __forceinline__ u32 bs_bits(BitSrc& s, u32 n)      // n <= 32; n == 0 reads nothing (src/BitMemory.h:93-123)
{
	if (n == 0) return 0;
	const u32 v = bs_peek32(s) >> (32u - n);
	bs_skip(s, n);
	return v;
}
__device__ __forceinline__ u32 bs_bit(BitSrc& s) { return bs_bits(s, 1); }
__device__ __forceinline__ void bs_align(BitSrc& s) { s.bit = (s.bit + 7) & ~7ull; }
__device__ __forceinline__ u32 bs_byte(BitSrc& s) { return bs_bits(s, 8); }
__device__ __forceinline__ u32 bs_word(BitSrc& s) { return bs_bits(s, 32); }
__device__ __forceinline__ u32 bs_pos(const BitSrc& s) { return (u32)(s.bit >> 3); }

__device__ __forceinline__ u32 dec_bit_length(u64 x)           // core::bit_length (src/utils.h:181-189)
{
	for (u32 i = 0; i < 32; ++i)
		if (x < (1ull << i)) return i;
	return 64;
}
__device__ __forceinline__ u32 dec_int_log2(u32 x) { u32 r = 0; for (u64 t = 2; t <= x; t *= 2) ++r; return r; }

// ---- Huffman trees --------------------------------------------------------------------------------------------------
// A stored tree is a preorder walk: 0 = internal node, 1 + id = leaf (HuffmanEncoder::StoreTree, src/huffman.cpp:177-221).
// It is unfolded into one u32 per internal node: low half = left child, high half = right child; a child is the index
// of an internal node of the same tree or 0x8000 | symbol.  Decoding a symbol walks from node 0, which is what
// GetBits(min_len) + DecodeFast + Decode(bit).. does for every stream the encoder can write.
struct NodePool { u32* w; u32 cap; u32 top; };

__device__ __forceinline__ u32 pool_take(NodePool& np, u32 n, u32* err)
{
	const u32 at = np.top;
	if ((u64)at + n + DEC_STACK_SLACK > np.cap) { *err |= DEC_ERR_POOL; return 0; }
	np.top += n;
	return at;
}

// size of the tree at the byte-aligned position `pos` without parsing it: LoadTree's memSize word
__device__ __forceinline__ u32 huff_peek(const BitSrc& s, u32 pos, u32* n_int)
{
	BitSrc t = s; t.bit = (u64)pos * 8;
	const u32 mem_size = bs_word(t), root_id = bs_word(t), n = bs_word(t);
	*n_int = (n >= 2 && n < 1024 && root_id >= n && root_id <= 2 * n - 2) ? root_id - n + 1 : 0;
	return mem_size;
}

// parses the tree at the (aligned) read position into the pool and returns its pool index
__device__ __forceinline__ u32 huff_load(BitSrc& s, NodePool& np)
{
	bs_align(s);
	const u32 begin = bs_pos(s);
	const u32 mem_size = bs_word(s), root_id = bs_word(s), n = bs_word(s);
	(void)bs_byte(s);                                            // min_len: only sizes the reference's speed-up table
	if (n < 2 || n >= 1024 || root_id < n || root_id > 2 * n - 2) { s.err |= DEC_ERR_FORMAT; return 0; }
	u32 bits_per_id = dec_int_log2(n);
	if (n & (n - 1)) bits_per_id++;
	const u32 n_int = root_id - n + 1;
	const u32 base = pool_take(np, n_int, &s.err);
	if (s.err & DEC_ERR_POOL) return 0;
	u32* T = np.w + base;
	u32* stack = np.w + np.top;                                 // nodes waiting for their right child (depth <= n_int <= slack)
	u32 sp = 0, made = 0, cur = 0, side = 0;
	if (bs_bit(s)) { s.err |= DEC_ERR_FORMAT; return base; }    // a leaf as root leaves the reference's tree uninitialised
	made = 1; T[0] = 0;
	for (u32 guard = 0; guard < 2 * n_int + 2; ++guard)
	{
		u32 child;
		if (bs_bit(s)) child = 0x8000u | (bs_bits(s, bits_per_id) & 0x7FFFu);
		else
		{
			if (made >= n_int) { s.err |= DEC_ERR_FORMAT; return base; }
			child = made++; T[child] = 0;
		}
		if (side == 0)
		{
			T[cur] |= child;
			if (child & 0x8000u) side = 1;
			else { stack[sp++] = cur; cur = child; }
		}
		else
		{
			T[cur] |= child << 16;
			if (!(child & 0x8000u)) { cur = child; side = 0; }
			else
			{
				if (sp == 0) { cur = 0xFFFFFFFFu; break; }
				cur = stack[--sp];
			}
		}
	}
	if (cur != 0xFFFFFFFFu || made != n_int) s.err |= DEC_ERR_FORMAT;
	bs_align(s);
	if (begin + mem_size != bs_pos(s)) s.err |= DEC_ERR_FORMAT;   // ASSERT(memBegin + memSize == Position())
	return base;
}

__device__ __forceinline__ u32 huff_sym(BitSrc& s, const u32* T)
{
	u32 node = 0;
	for (u32 round = 0; round < 2; ++round)
	{
		u32 w = bs_peek32(s);
		for (u32 k = 1; k <= 32; ++k)
		{
			const u32 t = T[node];
			const u32 child = (w >> 31) ? (t >> 16) : (t & 0xFFFFu);
			w <<= 1;
			if (child & 0x8000u) { bs_skip(s, k); return child & 0x7FFFu; }
			node = child;
		}
		bs_skip(s, 32);
	}
	s.err |= DEC_ERR_FORMAT;
	return 0;
}

// floor(n / d) for the range decoder: n < 2^53 on every valid stream (buffer < range x total), so one f64 division is at
// most one off and two integer checks make it exact; larger numerators (corrupt data) take the integer division
__device__ __forceinline__ u32 div_u64_u32(u64 n, u32 d)
{
	if (n >> 52) return (u32)(n / d);
	u64 q = (u64)((double)n / (double)d);
	if (q * d > n) --q;
	else if ((q + 1) * d <= n) ++q;
	return (u32)q;
}

// ---- range decoder + adaptive rows: RangeDecoder (src/RangeCoder.h:90-142), TSymbolCoderRC<N>::DecodeSymbol
// (src/SymbolCoderRC.h:50-91) -------------------------------------------------------------------------------------------
// The coder's byte source is a 16-byte look-ahead window over the block (w0 = the 8 bytes being consumed, w1 = the next 8,
// requested when w0 is taken into use), so a renormalisation step costs a shift, not a dependent memory access.
struct RangeDec { u64 low, buffer; u32 range; u64 w0, w1; u32 left; u64 next; };

__device__ __forceinline__ u64 rd_window(const BitSrc& s, u64 byte_pos)          // 8 bytes at byte_pos, first byte in the top bits
{
	BitSrc t = s; t.bit = byte_pos * 8;
	const u64 hi = bs_peek32(t);
	t.bit += 32;
	return (hi << 32) | (u64)bs_peek32(t);
}
__device__ __forceinline__ u32 rd_byte(RangeDec& d, BitSrc& s)
{
	if (d.left == 0) { d.w0 = d.w1; d.w1 = rd_window(s, d.next + 16); d.next += 8; d.left = 8; }     // next = first byte of w0
	const u32 b = (u32)(d.w0 >> 56);
	d.w0 <<= 8; --d.left;
	return b;
}
// hands the stream position back to the bit source (bytes consumed so far)
__device__ __forceinline__ void rd_finish(const RangeDec& d, BitSrc& s)
{
	s.bit = (d.next + 8 - d.left) * 8;
	if (s.bit > (u64)s.size * 8) s.err |= DEC_ERR_TRUNC;
}

__device__ __forceinline__ void rd_start(RangeDec& d, BitSrc& s)
{
	const u64 at = s.bit >> 3;                                  // byte aligned here
	d.buffer = rd_window(s, at);
	d.next = at + 8; d.w0 = rd_window(s, d.next); d.w1 = rd_window(s, d.next + 8); d.left = 8;
	d.low = 0; d.range = 0xFFFFFFFFu;
}

template <u32 N>
__device__ __forceinline__ u32 rd_symbol(RangeDec& d, BitSrc& s, u16* row)
{
	u32 acc = 0;
	for (u32 i = 0; i < N; ++i) acc += row[i];
	if (acc >= (1u << 16) - N * 2)
	{
		acc = 0;
		for (u32 i = 0; i < N; ++i) { const u16 v = row[i]; const u16 nv = (u16)(v - (v >> 1)); row[i] = nv; acc += nv; }
	}
	d.range /= acc;
	if (d.range == 0) { s.err |= DEC_ERR_FORMAT; d.range = 1; }
	const u32 cul = div_u64_u32(d.buffer, d.range);               // Freq is uint32: the quotient is truncated
	u32 idx = 0, hi = 0;
	for (;;)
	{
		hi += row[idx];
		if (hi > cul) break;
		if (++idx == N) { s.err |= DEC_ERR_FORMAT; idx = N - 1; break; }     // the reference walks off the row here
	}
	const u32 f = row[idx];
	hi -= f;
	const u32 rr = hi * d.range;                                   // uint32 product
	d.buffer -= rr; d.low += rr;
	d.range *= f;
	while (d.range <= 0x00FFFFFFu)
	{
		if ((d.low ^ (d.low + d.range)) & 0xFF00000000000000ull)
		{
			const u32 lo = (u32)d.low;
			d.range = (lo | 0x00FFFFFFu) - lo;
		}
		d.buffer = (d.buffer << 8) + rd_byte(d, s);
		d.low <<= 8; d.range <<= 8;
		if (d.range == 0) { s.err |= DEC_ERR_FORMAT; d.range = 0xFFFFFFFFu; break; }
	}
	row[idx] = (u16)(f + 2);
	return idx;
}

// ---- stage 1: meta stream + size of the tag header (thread per block) -----------------------------------------------------
// ReadMetaData (src/BlockCompressor.cpp:300-356).  The tag header is walked once without building anything -- trees are
// skipped through their memSize word -- to learn how many tree nodes and fields the block needs (TagTokenizerDecoder::
// ReadFields, src/TagModeler.cpp:896-1017).
__global__ void __launch_bounds__(64) k_dec_meta(const u8* in, const DecDesc* desc, DecState* st, DecParams prm)
{
	const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= prm.n_blocks) return;
	const DecDesc d = desc[b];
	DecState S; memset(&S, 0, sizeof(S));
	BitSrc s; s.p = in + d.in_off; s.size = d.in_size; s.err = 0; s.bit = 0;
	S.n_recs = bs_word(s); S.max_qlen = bs_word(s); S.flags = bs_word(s); S.chunk_size = bs_word(s);
	S.min_qlen = (S.flags & 2u) ? bs_word(s) : S.max_qlen;
	if (prm.color_space)
	{
		if (S.flags & 1u) { S.cs_seq_begin = bs_byte(s); S.cs_qua_begin = bs_byte(s); }
		else s.err |= DEC_ERR_REF_UB;        // ProcessBackward looks a character up in the index table (src/RecordsProcessor.cpp:297-313)
	}
	if (prm.crc)
	{
		if (!prm.tag_flags) S.crc_stored[0] = bs_word(s);
		S.crc_stored[1] = bs_word(s);
		if (!prm.lossy) S.crc_stored[2] = bs_word(s);
	}
	bs_align(s);
	S.tag_pos = bs_pos(s);
	if (S.n_recs == 0 || S.n_recs > d.in_size * 8u || S.max_qlen > 65535u || S.min_qlen > S.max_qlen || (S.flags & ~7u)) s.err |= DEC_ERR_FORMAT;
	u32 nodes = 0, nf = 0;
	if (!s.err)
	{
		if (S.flags & 4u)
		{	// TagRawDecoder::StartDecoding (src/TagModeler.cpp:1287-1316): two words, 128 presence bits, one tree
			u32 ni; (void)huff_peek(s, S.tag_pos + 8 + 16, &ni); nodes = ni;
		}
		else
		{
			nf = bs_byte(s);
			if (nf == 0) s.err |= DEC_ERR_FORMAT;
			for (u32 i = 0; i < nf && !s.err; ++i)
			{
				(void)bs_byte(s);
				if (bs_byte(s)) { const u32 len = bs_word(s); if (len >= (1u << 16)) { s.err |= DEC_ERR_FORMAT; break; } bs_skip(s, len * 8); continue; }
				if (bs_byte(s))
				{
					const u32 scheme = bs_byte(s);
					bs_skip(s, 64);
					if (scheme == NS_DELTA_CONST || scheme == NS_DELTA_RLE || scheme == NS_DELTA_VAR) bs_skip(s, 64);
					else if (scheme != NS_VALUE_RLE && scheme != NS_VALUE_VAR) { s.err |= DEC_ERR_FORMAT; break; }
					if (scheme == NS_DELTA_VAR || scheme == NS_VALUE_VAR)
						if (bs_byte(s)) { u32 ni; const u32 ms = huff_peek(s, bs_pos(s), &ni); nodes += ni; if (ms < 13 || ni == 0) { s.err |= DEC_ERR_FORMAT; break; } bs_skip(s, ms * 8); }
					continue;
				}
				(void)bs_byte(s);
				const u32 len = bs_word(s), max_len = bs_word(s), min_len = bs_word(s);
				if (len >= (1u << 16) || max_len >= (1u << 16) || min_len > max_len) { s.err |= DEC_ERR_FORMAT; break; }
				bs_skip(s, len * 8);
				const u64 ham = s.bit;
				bs_skip(s, len); bs_align(s);
				nodes += 129;
				const u32 upto = max_len < 128u ? max_len : 128u;
				for (u32 j = 0; j < upto + (max_len >= 128u ? 1u : 0u) && !s.err; ++j)
				{
					const u32 k = j < upto ? j : 128u;
					bool has = k == 128u;
					if (!has)
					{
						BitSrc t = s; t.bit = ham + k;
						has = k >= len || !bs_bit(t);
					}
					if (!has) continue;
					u32 ni; const u32 ms = huff_peek(s, bs_pos(s), &ni); nodes += ni;
					if (ms < 13 || ni == 0) { s.err |= DEC_ERR_FORMAT; break; }
					bs_skip(s, ms * 8);
				}
			}
		}
	}
	S.tag_nodes = nodes; S.n_fields = nf;
	S.err = s.err;
	st[b] = S;
}

// ---- stage 2: tag stream (wave per block, lane 0 walks) -------------------------------------------------------------------
__device__ __forceinline__ void text_put(u8* out, u32 cap, u32& pos, u32 c, u32* err)
{
	if (pos < cap) out[pos] = (u8)c; else *err |= DEC_ERR_TEXT;
	pos++;
}

// core::to_string (src/utils.h:69-97); values >= 10^9 overflow `power` in the reference
__device__ __forceinline__ void text_put_number(u8* out, u32 cap, u32& pos, u32 v, u32* err)
{
	if (v >= 1000000000u) *err |= DEC_ERR_REF_UB;
	u32 digits = 1;
	for (u32 t = v; t >= 10; t /= 10) ++digits;
	const u32 end = pos + digits;
	for (u32 k = 0; k < digits; ++k)
	{
		const u32 at = end - 1 - k;
		if (at < cap) out[at] = (u8)('0' + v % 10); else *err |= DEC_ERR_TEXT;
		v /= 10;
	}
	pos = end;
}

// TagTokenizerDecoder::ReadNumericField (src/TagModeler.cpp:1098-1205)
__device__ __forceinline__ u32 tag_numeric(BitSrc& s, DecField& f, const u32* pool, u32 rec)
{
	u32 v;
	if (rec == 0)
	{
		v = bs_bits(s, f.bits_value);
		if (f.scheme == NS_VALUE_RLE) { f.rle_len = bs_bits(s, 8); f.rle_sym = v; }
		return v + (u32)f.min_value;
	}
	switch (f.scheme)
	{
	case NS_DELTA_CONST:
		return f.prev + (u32)f.min_delta;
	case NS_DELTA_RLE:
		if (rec == 1 || f.rle_len == 0) { v = bs_bits(s, f.bits_num); f.rle_sym = v; f.rle_len = bs_bits(s, 8); }
		else { f.rle_len--; v = f.rle_sym; }
		return v + f.prev + (u32)f.min_delta;
	case NS_VALUE_VAR: case NS_DELTA_VAR:
		v = f.has_global ? huff_sym(s, pool + f.global_tree) : bs_bits(s, f.bits_num);
		return f.scheme == NS_DELTA_VAR ? v + f.prev + (u32)f.min_delta : v + (u32)f.min_value;
	case NS_VALUE_RLE:
		if (f.rle_len == 0) { v = bs_bits(s, f.bits_num); f.rle_sym = v; f.rle_len = bs_bits(s, 8); }
		else { f.rle_len--; v = f.rle_sym; }
		return v + (u32)f.min_value;
	default:
		s.err |= DEC_ERR_FORMAT; return 0;
	}
}

// ReadTags (src/BlockCompressor.cpp:491-573) with TagTokenizerDecoder / TagRawDecoder (src/TagModeler.cpp:887-1343):
// titles are decoded straight into the text at their final position; the positions of the other three lines of the
// record follow from its length.  The separators ('\n', '+', the repeated title) are written by k_dec_layout.
struct TagHead { u32 nf, raw_tree, min_title, max_title, tl_bits, raw_n, raw_map; };

// the header of the tag stream: field descriptors and Huffman trees (TagTokenizerDecoder::ReadFields, src/TagModeler.cpp:896-1017;
// TagRawDecoder::StartDecoding, :1287-1316); one lane
__device__ __forceinline__ void tags_header(BitSrc& s, NodePool& np, DecField* F, bool mixed, TagHead& H)
{
	H.nf = 0; H.raw_tree = 0; H.min_title = 0; H.max_title = 0; H.tl_bits = 0; H.raw_n = 0; H.raw_map = 0;
	if (!mixed)
	{
		H.nf = bs_byte(s);
		for (u32 i = 0; i < H.nf && !s.err; ++i)
		{
			DecField f; memset(&f, 0, sizeof(f));
			f.sep = (u8)bs_byte(s);
			f.is_constant = bs_byte(s) != 0;
			if (f.is_constant)
			{
				f.len = bs_word(s); f.data_pos = bs_pos(s); bs_skip(s, f.len * 8);
				F[i] = f; continue;
			}
			f.is_numeric = bs_byte(s) != 0;
			if (f.is_numeric)
			{
				f.scheme = (u8)bs_byte(s);
				f.min_value = (i32)bs_word(s); f.max_value = (i32)bs_word(s);
				f.bits_value = dec_bit_length((u64)(i64)(i32)((u32)f.max_value - (u32)f.min_value));
				if (f.scheme == NS_DELTA_CONST || f.scheme == NS_DELTA_RLE || f.scheme == NS_DELTA_VAR)
				{
					f.min_delta = (i32)bs_word(s); f.max_delta = (i32)bs_word(s);
					f.bits_num = dec_bit_length((u64)(i64)(i32)((u32)f.max_delta - (u32)f.min_delta));
				}
				else f.bits_num = f.bits_value;
				if (f.scheme == NS_DELTA_VAR || f.scheme == NS_VALUE_VAR)
				{
					f.var_stat = (u8)bs_byte(s);
					if (f.var_stat) { f.global_tree = huff_load(s, np); f.has_global = 1; }
				}
				F[i] = f; continue;
			}
			f.is_len_constant = bs_byte(s) != 0;
			f.len = bs_word(s); f.max_len = bs_word(s); f.min_len = bs_word(s);
			f.bits_len = dec_bit_length((u64)(f.max_len - f.min_len));
			f.data_pos = bs_pos(s); bs_skip(s, f.len * 8);
			f.ham_bit = (u32)s.bit; bs_skip(s, f.len); bs_align(s);
			f.local_dir = pool_take(np, 129, &s.err);
			if (s.err) break;
			u32* dir = np.w + f.local_dir;
			for (u32 j = 0; j < 129; ++j) dir[j] = 0xFFFFFFFFu;
			const u32 upto = f.max_len < 128u ? f.max_len : 128u;
			for (u32 j = 0; j < upto && !s.err; ++j)
			{
				BitSrc t = s; t.bit = (u64)f.ham_bit + j;
				if (j >= f.len || !bs_bit(t)) dir[j] = huff_load(s, np);
			}
			if (f.max_len >= 128u && !s.err) dir[128] = huff_load(s, np);
			F[i] = f;
		}
	}
	else
	{
		H.min_title = bs_word(s); H.max_title = bs_word(s);
		H.tl_bits = dec_bit_length((u64)(H.max_title - H.min_title));
		H.raw_map = pool_take(np, 128, &s.err);
		for (u32 i = 0; i < 128 && !s.err; ++i) if (bs_bit(s)) np.w[H.raw_map + H.raw_n++] = i;
		if (!s.err) H.raw_tree = huff_load(s, np);
	}

}

// the records of the tag stream, one lane walking the bit stream (the reference's own loop; the fallback of k_dec_tags_wave and
// what DSRC_GPU_DEC_SERIAL runs)
__device__ __forceinline__ void tags_records_serial(BitSrc& s, NodePool& np, DecField* F, const TagHead& H, bool mixed, DecState* S, const DecDesc& d,
													RecPools rp, u8* text, const DecParams& prm, u32* pos_out, u32* q_total_out)
{
	const u32 cap = d.out_cap;
	const u32 len_bits = dec_bit_length((u64)(S->max_qlen - S->min_qlen));
	const u32 cs_delta = (prm.color_space && (S->flags & 1u)) ? 1u : 0u;
	const u64 r0 = d.rec_base;
	u32 pos = 0, q_total = 0;
	for (u32 i = 0; i < S->n_recs && !s.err; ++i)
	{
		const u32 t0 = pos;
		u32 tl;
		if (!mixed)
		{
			for (u32 j = 0; j < H.nf; ++j)
			{
				DecField& f = F[j];
				if (f.is_constant)
				{
					const u8* src = s.p + f.data_pos;
					for (u32 k = 0; k < f.len; ++k) text_put(text, cap, pos, src[k], &s.err);
				}
				else if (f.is_numeric)
				{
					const u32 v = tag_numeric(s, f, np.w, i);
					text_put_number(text, cap, pos, v, &s.err);
					f.prev = v;
				}
				else
				{
					const u32 fl = f.is_len_constant ? f.len : bs_bits(s, f.bits_len) + f.min_len;
					const u8* tpl = s.p + f.data_pos;
					const u32* dir = np.w + f.local_dir;
					for (u32 k = 0; k < fl && !s.err; ++k)
					{
						bool fixed = false;
						if (k < f.len) { BitSrc t = s; t.bit = (u64)f.ham_bit + k; fixed = bs_bit(t) != 0; }
						if (fixed) text_put(text, cap, pos, tpl[k], &s.err);
						else
						{
							const u32 tr = dir[k < 128u ? k : 128u];
							if (tr == 0xFFFFFFFFu) { s.err |= DEC_ERR_FORMAT; break; }
							text_put(text, cap, pos, huff_sym(s, np.w + tr), &s.err);
						}
					}
				}
				text_put(text, cap, pos, f.sep, &s.err);
			}
			pos--;                                                // the last separator is not part of the title
			tl = pos - t0;
		}
		else
		{
			tl = H.tl_bits ? bs_bits(s, H.tl_bits) + H.min_title : H.max_title;
			for (u32 k = 0; k < tl && !s.err; ++k)
			{
				const u32 x = huff_sym(s, np.w + H.raw_tree);
				text_put(text, cap, pos, x < H.raw_n ? np.w[H.raw_map + x] : 255u, &s.err);
			}
		}
		pos++;                                                    // '\n'
		const u32 ql = len_bits ? bs_bits(s, len_bits) + S->min_qlen : S->max_qlen;
		if (tl > 65535u || ql > 65535u) { s.err |= DEC_ERR_FORMAT; break; }
		const u64 g = r0 + i;
		rp.title_off[g] = t0; rp.title_len[g] = (u16)tl; rp.len[g] = (u16)ql;
		rp.seq_off[g] = pos + cs_delta; pos += ql + cs_delta + 1;           // sequence line + '\n'
		pos += 1 + (prm.plus_rep ? tl - 1 : 0u) + 1;                         // '+' [title] '\n'
		rp.qual_off[g] = pos + cs_delta; pos += ql + cs_delta + 1;
		q_total += ql;
		if (pos > cap) s.err |= DEC_ERR_TEXT;
	}
	*pos_out = pos; *q_total_out = q_total;
}

// what follows the records: where the quality stream starts, and the head of it that the host sizes the model table from
__device__ __forceinline__ void tags_finish(BitSrc& s, DecState* S, const DecParams& prm, u32 pos, u32 q_total)
{
	bs_align(s);
	S->qua_pos = bs_pos(s); S->text_bytes = pos; S->q_total = q_total;
	// the scheme byte of the quality stream (IQualityModelerProxy::Decode, src/QualityModelerProxy.h:59-69): the host sizes the
	// model table of an order-context scheme from it
	if (!s.err && !(prm.quality_order > 0 && prm.lossy))
	{
		BitSrc t = s;
		const u32 sch = bs_byte(t);
		S->q_scheme = sch;
		if (t.err || (prm.quality_order == 0 ? sch > 2 : sch > 7)) s.err |= t.err | DEC_ERR_FORMAT;
		if (prm.quality_order > 0 && !s.err)
		{	// TTranslationalQualityEncoder::Read (src/QualityEncoder.h:344-357): 256 presence bits; the decoder's table has
			// one row per context made of PRESENT symbols
			bs_align(t);
			u32 cnt = 0;
			for (u32 i = 0; i < 8; ++i) cnt += (u32)__popc(bs_word(t));
			S->q_cnt = cnt;
			if (t.err || cnt == 0) s.err |= t.err | DEC_ERR_FORMAT;
		}
	}
	S->err |= s.err;
}

__global__ void __launch_bounds__(64) k_dec_tags(const u8* in, const DecDesc* desc, DecState* st, RecPools rp, u8* out, u32* pool, u8* fld_pool, DecParams prm)
{
	const u32 b = blockIdx.x;
	DecState* S = &st[b];
	if (threadIdx.x != 0 || S->err) return;
	const DecDesc d = desc[b];
	BitSrc s; s.p = in + d.in_off; s.size = d.in_size; s.err = 0; s.bit = (u64)S->tag_pos * 8;
	NodePool np; np.w = pool + d.node_off; np.cap = d.node_cap; np.top = 0;
	DecField* F = (DecField*)(fld_pool + d.fld_off);
	const bool mixed = (S->flags & 4u) != 0;
	TagHead H;
	tags_header(s, np, F, mixed, H);
	u32 pos = 0, q_total = 0;
	tags_records_serial(s, np, F, H, mixed, S, d, rp, out + d.out_off, prm, &pos, &q_total);
	tags_finish(s, S, prm, pos, q_total);
}

// ---- stage 3: quality and DNA streams (wave per slot; lane 0 walks, all lanes clear the model table) ------------------------
__device__ __forceinline__ void table_fill(u32* tab, u64 words)
{
	// counters start at 1 (SymbolCoderRC: std::fill(stats, .., 1))
	u64* t8 = (u64*)tab;
	const u64 n8 = words / 2;
	for (u64 i = threadIdx.x; i < n8; i += blockDim.x) t8[i] = 0x0001000100010001ull;
	if ((words & 1) && threadIdx.x == 0) tab[words - 1] = 0x00010001u;
}

__device__ __forceinline__ bool q_special(u32 q, u32 lossy) { return lossy ? q == 0 : q >= 128; }

// TQualityOrderModeler::Decode with T*QualityEncoder::Decode + TQualityModelExt::DecodeSymbol
// (src/QualityOrderModeler.h:49-65, src/QualityEncoder.h:77-143,248-263,306-326)
template <u32 N>
__device__ __forceinline__ void qua_order_decode(BitSrc& s, u16* tab, u32 ord, u32 rescale, const u8* translate, u32 lossy,
								 const DecDesc& d, DecState* S, RecPools rp, u8* text)
{
	const u32 abits = dec_int_log2(N);
	const u64 sym_mask = ((u64)1 << abits) - 1;
	const u32 bits_lo = (ord / 2) * abits, bits_hi = (ord / 2 + 1) * abits;
	const u64 lo_mask = bits_lo ? (((u64)1 << bits_lo) - 1) : 0;
	const u64 hi_mask = ((u64)1 << bits_hi) - 1;
	const u64 swap_mask = lo_mask | ~hi_mask;
	const u64 hash_mask = ((u64)1 << (ord * abits)) - 1;
	u64 hash = 0, sym_buf = 0;
	RangeDec rd; rd_start(rd, s);
	u32 d_total = 0;
	for (u32 k = 0; k < S->n_recs && !s.err; ++k)
	{
		const u64 g = (u64)d.rec_base + k;
		const u32 ql = rp.len[g];
		u8* q = text + rp.qual_off[g];
		u32 ncount = 0;
		for (u32 j = 0; j < ql; ++j)
		{
			const u32 pctx = j * rescale / ql;
			const u64 h = (hash & hash_mask) * rescale + pctx;       // dense: the table is the decoder's own (the reference shifts by abits)
			const u32 c = rd_symbol<N>(rd, s, tab + h * N);
			const u32 qv = translate ? translate[c] : c;
			q[j] = (u8)qv;
			ncount += q_special(qv, lossy) ? 1u : 0u;
			hash <<= abits;
			const u64 next_buf = (hash >> bits_lo) & sym_mask;
			const u64 swp = (next_buf + sym_buf) / 2;
			hash &= swap_mask; hash |= swp << bits_lo; hash |= c;
			sym_buf = next_buf;
		}
		rp.kept[g] = (u16)(ql - ncount); rp.d_off[g] = d_total; d_total += ql - ncount;
	}
	rd_finish(rd, s);
	S->d_total = d_total;
}

__device__ __forceinline__ u32 dec_wave_scan(u32 v) { return wave_incl_scan_dpp(v); }
// value of lane `l` (wave-uniform) as a scalar: v_readlane with the lane number in an SGPR
__device__ __forceinline__ u32 dec_readlane(u32 v, u32 l)
{
	return (u32)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane((int)l));
}

// rd_symbol on a row held in registers (the caller stores it back): *rescaled tells whether every counter changed
template <u32 N>
__device__ __forceinline__ u32 rd_symbol_regs(RangeDec& d, BitSrc& s, u32 (&row)[N], bool* rescaled)
{
	u32 acc = 0;
#pragma unroll
	for (u32 i = 0; i < N; ++i) acc += row[i];
	*rescaled = false;
	if (acc >= (1u << 16) - N * 2)
	{
		acc = 0;
#pragma unroll
		for (u32 i = 0; i < N; ++i) { row[i] -= row[i] >> 1; acc += row[i]; }
		*rescaled = true;
	}
	d.range /= acc;
	if (d.range == 0) { s.err |= DEC_ERR_FORMAT; d.range = 1; }
	const u32 cul = div_u64_u32(d.buffer, d.range);
	u32 idx = N - 1, hi = 0, f = 0, lo = 0;
	bool found = false;
#pragma unroll
	for (u32 i = 0; i < N; ++i)
	{
		hi += row[i];
		if (!found && hi > cul) { found = true; idx = i; f = row[i]; lo = hi - row[i]; }
	}
	if (!found) { s.err |= DEC_ERR_FORMAT; f = row[N - 1]; lo = hi - f; }        // the reference walks off the row here
	const u32 rr = lo * d.range;
	d.buffer -= rr; d.low += rr;
	d.range *= f;
	while (d.range <= 0x00FFFFFFu)
	{
		if ((d.low ^ (d.low + d.range)) & 0xFF00000000000000ull)
		{
			const u32 l32 = (u32)d.low;
			d.range = (l32 | 0x00FFFFFFu) - l32;
		}
		d.buffer = (d.buffer << 8) + rd_byte(d, s);
		d.low <<= 8; d.range <<= 8;
		if (d.range == 0) { s.err |= DEC_ERR_FORMAT; d.range = 0xFFFFFFFFu; break; }
	}
#pragma unroll
	for (u32 i = 0; i < N; ++i) if (i == idx) row[i] += 2;
	return idx;
}

// TDnaRCOrderModeler::Decode (src/DnaModelerRCO.h:62-79) on one lane.  The context of symbol t+1 is (context of t << bits | symbol
// t): whatever symbol t turns out to be, its row is one of N CONSECUTIVE rows, 8 N^2 / 4 bytes of table that are known
// before symbol t is decoded.  They are requested first, so the row read of the next symbol overlaps the arithmetic of this
// one instead of following it; the current row lives in registers (it is the only row a store can have made stale).
template <u32 N>
__device__ __forceinline__ void dna_order_decode(BitSrc& s, u16* tab, u32 ord, const DecDesc& d, DecState* S, RecPools rp, u8* dst)
{
	const u32 abits = dec_int_log2(N);
	const u64 mask = ((u64)1 << (abits * ord)) - 1;
	u64 hash = 0;
	RangeDec rd; rd_start(rd, s);
	const u32 total = S->d_total;
	u32 cur[N];
#pragma unroll
	for (u32 i = 0; i < N; ++i) cur[i] = tab[i];
	for (u32 t = 0; t < total && !s.err; ++t)
	{
		const u64 nbase = (hash << abits) & mask;                 // first of the N candidate rows of symbol t+1
		const u64* cp = (const u64*)(tab + nbase * N);             // N rows x N counters x 2 bytes, 8-byte aligned
		u64 cand[N * N / 4];
#pragma unroll
		for (u32 i = 0; i < N * N / 4; ++i) cand[i] = cp[i];
		bool rescaled;
		const u32 c = rd_symbol_regs<N>(rd, s, cur, &rescaled);
		dst[t] = (u8)c;
		u16* row = tab + hash * N;
		if (rescaled) { for (u32 i = 0; i < N; ++i) row[i] = (u16)cur[i]; }
		else row[c] = (u16)cur[c];
		const u64 nh = nbase | c;
		if (nh != hash)
		{	// candidate row c = N / 4 consecutive 64-bit words
			constexpr u32 W = N / 4;
			u64 r[W];
#pragma unroll
			for (u32 w = 0; w < W; ++w) r[w] = cand[w];
#pragma unroll
			for (u32 k = 1; k < N; ++k)
				if (c == k)
				{
#pragma unroll
					for (u32 w = 0; w < W; ++w) r[w] = cand[k * W + w];
				}
#pragma unroll
			for (u32 i = 0; i < N; ++i) cur[i] = (u32)(r[i / 4] >> (16 * (i & 3))) & 0xFFFFu;
		}
		hash = nh;
	}
	rd_finish(rd, s);
}

// ---- stage 3a: quality stream of the -q0 levels (position Huffman, truncated, RLE): wave per block, lane 0 walks -----------
// The trees of a block (one per read position, ~40 internal nodes each) are parsed into the HBM pool by lane 0, copied
// into LDS by the wave when they fit (DEC_LDS_NODES words), and the symbol loop then costs one LDS read per code bit and a
// register shift for the bit stream (64-bit window, refilled 32 bits at a time) instead of two dependent global loads.
#define DEC_LDS_NODES 11776u          // 46 KB: three workgroups per CU

struct BitWin { u64 w; u32 n; u64 next; };      // w: unread bits at the top; n of them valid; next: byte position of the next refill

__device__ __forceinline__ void bw_init(BitWin& b, const BitSrc& s)
{
	b.next = s.bit >> 3; b.w = 0; b.n = 0;
	const u32 skip = (u32)s.bit & 7u;
	BitSrc t = s; t.bit = b.next * 8;
	b.w = (u64)bs_peek32(t) << 32; t.bit += 32; b.w |= (u64)bs_peek32(t);
	b.next += 8; b.n = 64;
	b.w <<= skip; b.n -= skip;
}
__device__ __forceinline__ void bw_refill(BitWin& b, const BitSrc& s)          // keeps at least 32 valid bits
{
	if (b.n < 32)
	{
		BitSrc t = s; t.bit = b.next * 8;
		b.w |= (u64)bs_peek32(t) << (32 - b.n);
		b.next += 4; b.n += 32;
	}
}
__device__ __forceinline__ u32 bw_bits(BitWin& b, const BitSrc& s, u32 n)       // n <= 32
{
	if (n == 0) return 0;
	bw_refill(b, s);
	const u32 v = (u32)(b.w >> (64 - n));
	b.w <<= n; b.n -= n;
	return v;
}
__device__ __forceinline__ void bw_finish(const BitWin& b, BitSrc& s)
{
	s.bit = b.next * 8 - b.n;
	if (s.bit > (u64)s.size * 8) s.err |= DEC_ERR_TRUNC;
}
template <typename PT>
__device__ __forceinline__ u32 bw_huff(BitWin& b, const BitSrc& s, PT T, u32* err)
{
	u32 node = 0;
	for (u32 round = 0; round < 2; ++round)
	{
		bw_refill(b, s);
		u64 w = b.w;
		for (u32 k = 1; k <= 32; ++k)
		{
			const u32 t = T[node];
			const u32 child = (w >> 63) ? (t >> 16) : (t & 0xFFFFu);
			w <<= 1;
			if (child & 0x8000u) { b.w = w; b.n -= k; return child & 0x7FFFu; }
			node = child;
		}
		b.w = w; b.n -= 32;
	}
	*err |= DEC_ERR_FORMAT;
	return 0;
}

// Plain / Truncated::DecodeRecords (src/QualityPositionModeler.cpp:189-220,291-337) over trees at `W` (HBM pool or its LDS copy)
template <typename PT>
__device__ __forceinline__ void qpos_records(BitSrc& s, PT W, u32 dir, u32 maxl, u32 n, const u8* sym, bool truncated, u32 lossy,
											 const DecDesc& d, DecState* S, RecPools rp, u8* text)
{
	BitWin b; bw_init(b, s);
	const u32 max_bits = dec_bit_length(maxl);
	const u32 variable = truncated ? bw_bits(b, s, 1) : 0u;
	const u32 hash_sym = lossy ? 1u : 2u;                    // HashSymbolQuantized / HashSymbolNormal
	u32 d_total = 0, err = 0;
	const u32 n_recs = S->n_recs;
	for (u32 k = 0; k < n_recs && !err; ++k)
	{
		const u64 g = (u64)d.rec_base + k;
		const u32 ql = rp.len[g];
		u8* q = text + rp.qual_off[g];
		u32 th = ql, ncount = 0;
		if (truncated && bw_bits(b, s, 1)) th = bw_bits(b, s, variable ? dec_bit_length(ql) : max_bits);
		if (th > ql || th > maxl) { err |= DEC_ERR_FORMAT; break; }
		for (u32 j = 0; j < th; ++j)
		{
			const u32 x = bw_huff(b, s, W + W[dir + j], &err);
			const u32 qv = x < n ? sym[x] : 255u;
			q[j] = (u8)qv; ncount += q_special(qv, lossy) ? 1u : 0u;
		}
		for (u32 j = th; j < ql; ++j) q[j] = (u8)hash_sym;
		rp.kept[g] = (u16)(ql - ncount); rp.d_off[g] = d_total; d_total += ql - ncount;
	}
	S->d_total = d_total;
	bw_finish(b, s);
	s.err |= err;
	bs_align(s);
}

__global__ void __launch_bounds__(64) k_dec_qhuff(const u8* in, const DecDesc* desc, DecState* st, RecPools rp, u8* out, u32* pool, DecParams prm)
{
	__shared__ u32 s_nodes[DEC_LDS_NODES];
	__shared__ u8 s_sym[256];
	__shared__ u32 s_par[8];
	const u32 b = blockIdx.x;
	DecState* S = &st[b];
	if (S->err || S->q_done) return;                      // wave-uniform
	const DecDesc d = desc[b];
	u8* text = out + d.out_off;
	BitSrc s; s.p = in + d.in_off; s.size = d.in_size; s.err = 0; s.bit = (u64)S->qua_pos * 8;
	NodePool np; np.w = pool + d.qnode_off; np.cap = d.qnode_cap; np.top = 0;
	const u32 lossy = prm.lossy;
	u32 q_scheme = 0, dir = 0, maxl = 0, n = 0;
	if (threadIdx.x == 0)
	{
		// scheme byte (IQualityModelerProxy::Decode, src/QualityModelerProxy.h:59-69)
		q_scheme = bs_byte(s);
		if (q_scheme > 2) s.err |= DEC_ERR_FORMAT;
		S->q_scheme = q_scheme;
		if (!s.err && q_scheme <= 1)
		{	// IQualityPositionModeler::Decode: statistics, symbols, one tree per position (src/QualityPositionModeler.cpp:39-103)
			bs_align(s);
			maxl = bs_word(s);
			for (u32 i = 0; i < 256; ++i) if (bs_bit(s)) s_sym[n++] = (u8)i;
			if (maxl > 65535u) s.err |= DEC_ERR_FORMAT;
			dir = pool_take(np, maxl ? maxl : 1u, &s.err);
			for (u32 i = 0; i < maxl && !s.err; ++i) { const u32 t = huff_load(s, np); np.w[dir + i] = t; }
		}
		s_par[0] = (!s.err && q_scheme <= 1) ? 1u : 0u; s_par[1] = np.top; s_par[2] = dir; s_par[3] = maxl; s_par[4] = n;
	}
	__syncthreads();
	const bool pos_scheme = s_par[0] != 0;
	const bool in_lds = pos_scheme && s_par[1] <= DEC_LDS_NODES;
	if (in_lds) for (u32 i = threadIdx.x; i < s_par[1]; i += blockDim.x) s_nodes[i] = np.w[i];
	__syncthreads();
	if (threadIdx.x != 0) return;
	if (pos_scheme)
	{
		if (in_lds) qpos_records(s, (const LDS_AS u32*)s_nodes, dir, maxl, n, s_sym, q_scheme == 1, lossy, d, S, rp, text);
		else qpos_records(s, (const u32*)np.w, dir, maxl, n, s_sym, q_scheme == 1, lossy, d, S, rp, text);
	}
	else if (!s.err)
	{
		// QualityRLEModeler::Decode (src/QualityRLEModeler.cpp:48-113,380-486); the runs are expanded as they are decoded
		const u32 run_len = bs_word(s);
		u8* ls = (u8*)(np.w + pool_take(np, 64, &s.err));       // 256 length symbols
		u32 qn = 0, ln = 0;
		for (u32 i = 0; i < 256; ++i) { s_sym[i] = 255; ls[i] = 255; }
		for (u32 i = 0; i < 256; ++i) if (bs_bit(s)) s_sym[qn++] = (u8)i;
		for (u32 i = 0; i < 256; ++i) if (bs_bit(s)) ls[ln++] = (u8)i;
		bs_align(s);
		if (qn == 0 || ln == 0 || run_len == 0) s.err |= DEC_ERR_FORMAT;
		u32 qdir = 0, ldir = 0, l_begin = 0, l_end = 0;
		if (qn > 1 && !s.err)
		{
			qdir = pool_take(np, qn, &s.err); ldir = pool_take(np, qn, &s.err);
			for (u32 i = 0; i < qn && !s.err; ++i) { const u32 a = huff_load(s, np); np.w[qdir + i] = a; const u32 c = huff_load(s, np); np.w[ldir + i] = c; }
			bs_align(s);
		}
		else if (!s.err)
		{
			bs_align(s);
			if (ln > 1)
			{
				l_begin = ls[bs_byte(s) & 255u];
				l_end = ls[0]; if (l_end == l_begin) l_end = ls[1];
			}
			else { l_begin = ls[0]; l_end = l_begin; }
		}
		u32 cur_len = 0, idx = 0, cur_q = 0, prev = 0, d_total = 0;
		for (u32 k = 0; k < S->n_recs && !s.err; ++k)
		{
			const u64 g = (u64)d.rec_base + k;
			const u32 ql = rp.len[g];
			u8* q = text + rp.qual_off[g];
			u32 ncount = 0;
			for (u32 j = 0; j < ql; ++j)
			{
				if (cur_len == 0)
				{
					if (idx >= run_len) { s.err |= DEC_ERR_FORMAT; break; }
					if (qn > 1)
					{
						u32 x = huff_sym(s, np.w + np.w[qdir + prev]);
						if (x >= qn) { s.err |= DEC_ERR_FORMAT; break; }
						cur_q = s_sym[x]; prev = x;
						x = huff_sym(s, np.w + np.w[ldir + prev]);
						if (x >= ln) { s.err |= DEC_ERR_FORMAT; break; }
						cur_len = (u32)ls[x] + 1;
					}
					else { cur_q = s_sym[0]; cur_len = (idx + 1 == run_len ? l_end : l_begin) + 1; }
					idx++;
				}
				q[j] = (u8)cur_q; --cur_len;
				ncount += q_special(cur_q, lossy) ? 1u : 0u;
			}
			rp.kept[g] = (u16)(ql - ncount); rp.d_off[g] = d_total; d_total += ql - ncount;
		}
		// runs the records did not consume are still read by the reference (DecodeRuns comes first)
		for (; idx < run_len && qn > 1 && !s.err; ++idx)
		{
			u32 x = huff_sym(s, np.w + np.w[qdir + prev]);
			if (x >= qn) { s.err |= DEC_ERR_FORMAT; break; }
			prev = x;
			x = huff_sym(s, np.w + np.w[ldir + prev]);
		}
		S->d_total = d_total;
		bs_align(s);
	}
	S->dna_pos = bs_pos(s);
	S->err |= s.err;
}

// The ONE-LANE form of the range-decoded levels: wave per model-table slot looping over blocks, lane 0 walks the quality stream
// and then the DNA stream with the reference's own loops (qua_order_decode, dna_order_decode), the wave clears the slot's table.
// It is what DSRC_GPU_DEC_SERIAL=1 runs (an independent second implementation for the tests, and quick on the CPU emulator);
// the product path is k_dec_qrc / k_dec_dnarc / k_dec_dna0 (k_dec_rc.h).
__global__ void __launch_bounds__(64) k_dec_streams(const u8* in, const DecDesc* desc, DecState* st, RecPools rp, u8* out, u32* pool,
													 u8* d_stream, u32* tables, DecParams prm)
{
	__shared__ u8 s_sym[256];
	__shared__ u32 s_flag;
	__shared__ u32 s_par[5];
	u32* table = tables + (u64)blockIdx.x * prm.table_words;
	for (u32 b = blockIdx.x; b < prm.n_blocks; b += gridDim.x)
	{
		DecState* S = &st[b];
		if (S->err) continue;                         // wave-uniform
		const DecDesc d = desc[b];
		u8* text = out + d.out_off;
		BitSrc s; s.p = in + d.in_off; s.size = d.in_size; s.err = 0; s.bit = (u64)S->qua_pos * 8;
		NodePool np; np.w = pool + d.qnode_off; np.cap = d.qnode_cap; np.top = 0;
		const u32 qo = prm.quality_order, lossy = prm.lossy;

		// ===== quality =====
		// scheme byte (IQualityModelerProxy::Decode, src/QualityModelerProxy.h:59-69); the lossy order proxy has none (:156-159)
		u32 q_scheme = 0, qN = 8, q_ord = qo, q_rescale = 8;
		bool q_rc = false, q_translate = false;
		const bool q_elsewhere = qo == 0;              // Huffman / RLE schemes: k_dec_qhuff has run, the DNA stream starts at dna_pos
		if (threadIdx.x == 0 && q_elsewhere) s.bit = (u64)S->dna_pos * 8;
		if (threadIdx.x == 0 && !q_elsewhere)
		{
			if (lossy) q_rc = true;
			else
			{
				q_scheme = bs_byte(s);
				if (q_scheme > 7) s.err |= DEC_ERR_FORMAT;
				else
				{
					const u32 sc = q_scheme & 3u;
					qN = 16u << sc;
					q_ord = qo == 1 ? (sc == 0 ? 3u : sc == 1 ? 2u : 1u) : (4u - sc);
					q_rescale = q_scheme < 4 ? 8u : qN;
					q_rc = true; q_translate = true;
					// TTranslationalQualityEncoder::Read (src/QualityEncoder.h:344-357)
					bs_align(s);
					u32 cnt = 0;
					for (u32 i = 0; i < 256; ++i) s_sym[i] = 255;
					for (u32 i = 0; i < 256; ++i) if (bs_bit(s)) s_sym[cnt++] = (u8)i;
					bs_align(s);
				}
			}
			s_flag = (q_rc && !s.err) ? (qN | (q_ord << 8) | (q_rescale << 16)) : 0u;
		}
		if (threadIdx.x == 0 && q_elsewhere) s_flag = 0;
		__syncthreads();
		const u32 flag = s_flag;
		if (flag)
		{
			const u32 nn = flag & 0xFFu, ord = (flag >> 8) & 0xFFu, resc = flag >> 16;
			u32 ab = 0; for (u32 t = nn; t > 1; t >>= 1) ++ab;
			const u64 words = ((u64)1 << (ab * ord)) * resc * nn / 2;
			if (words > prm.table_words) { if (threadIdx.x == 0) s.err |= DEC_ERR_POOL; }
			else table_fill(table, words);
		}
		__syncthreads();
		if (threadIdx.x == 0 && !s.err && !q_elsewhere)
		{
			S->q_scheme = q_scheme;
			if (q_rc)
			{
				const u8* tr = q_translate ? s_sym : nullptr;
				switch (qN)
				{
				case 8:   qua_order_decode<8>(s, (u16*)table, q_ord, q_rescale, tr, lossy, d, S, rp, text); break;
				case 16:  qua_order_decode<16>(s, (u16*)table, q_ord, q_rescale, tr, lossy, d, S, rp, text); break;
				case 32:  qua_order_decode<32>(s, (u16*)table, q_ord, q_rescale, tr, lossy, d, S, rp, text); break;
				case 64:  qua_order_decode<64>(s, (u16*)table, q_ord, q_rescale, tr, lossy, d, S, rp, text); break;
				default:  qua_order_decode<128>(s, (u16*)table, q_ord, q_rescale, tr, lossy, d, S, rp, text); break;
				}
			}
			else s.err |= DEC_ERR_FORMAT;          // quality_order == 0 is decoded by k_dec_qhuff
			S->dna_pos = bs_pos(s);
		}

		// ===== DNA =====
		// IDnaModelerProxy::Decode (src/DnaModelerProxy.h:61-71)
		u32 d_scheme = 0;
		if (threadIdx.x == 0)
		{
			u32 f = 0;
			if (!s.err)
			{
				d_scheme = bs_byte(s);
				S->d_scheme = d_scheme;
				if (d_scheme != 255)
				{
					if (d_scheme > 1) s.err |= DEC_ERR_FORMAT;
					else if (prm.dna_order > 0) f = d_scheme ? 8u : 4u;
				}
			}
			s_flag = f;
			// the 2-bit packing has no serial dependency: its unpacking is handed to the whole wave
			const bool b2 = !s.err && d_scheme == 0 && prm.dna_order == 0;
			s_par[0] = b2 ? 1u : 0u; s_par[1] = S->d_total; s_par[2] = (u32)s.bit; s_par[3] = (u32)(s.bit >> 32);
		}
		__syncthreads();
		const u32 dN = s_flag;
		if (s_par[0])
		{	// DnaModelerBasicB2::Decode (src/DnaModelerBasicB2.h:48-60): symbol t is bits 2t, 2t+1; a lane takes 16 symbols (32 bits)
			BitSrc t = s; t.err = 0;
			const u64 bit0 = ((u64)s_par[3] << 32) | s_par[2];
			const u32 total = s_par[1];
			u8* dst = d_stream + d.d_base;
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
		const u32 d_ord = dN == 8 ? (prm.dna_order < 7u ? prm.dna_order : 7u) : prm.dna_order;
		if (dN)
		{
			const u64 words = ((u64)1 << ((dN == 8 ? 3u : 2u) * d_ord)) * dN / 2;
			if (words > prm.table_words) { if (threadIdx.x == 0) s.err |= DEC_ERR_POOL; }
			else table_fill(table, words);
		}
		__syncthreads();
		if (threadIdx.x == 0 && !s.err && d_scheme != 255)
		{
			u8* dst = d_stream + d.d_base;
			const u32 total = S->d_total;
			if (dN == 4) dna_order_decode<4>(s, (u16*)table, d_ord, d, S, rp, dst);
			else if (dN == 8) dna_order_decode<8>(s, (u16*)table, d_ord, d, S, rp, dst);
			else if (d_scheme == 0)
			{	// unpacked above by the whole wave; the stream position moves on
				bs_skip(s, total); bs_skip(s, total);
				bs_align(s);
			}
			else
			{	// DnaModelerHuffman::Decode (src/DnaModelerHuffman.cpp:75-113)
				u32 n = 0;
				for (u32 i = 0; i < 20; ++i) s_sym[i] = 255;
				for (u32 i = 0; i < 20; ++i) if (bs_bit(s)) s_sym[n++] = (u8)i;
				np.top = 0;
				const u32 tr = huff_load(s, np);
				for (u32 t = 0; t < total && !s.err; ++t) { const u32 x = huff_sym(s, np.w + tr); dst[t] = x < 20 ? s_sym[x] : 255; }
				bs_align(s);
			}
		}
		if (threadIdx.x == 0) { S->end_pos = bs_pos(s); S->err |= s.err; }
		__syncthreads();
	}
}

// ---- stage 4: per-record backward transform and line layout (wave per record) ---------------------------------------------
// Lossless/LossyRecordsProcessor::ProcessBackward (src/RecordsProcessor.cpp:269-315,410-454), ProcessRecordToColorSpace
// (:60-101), and the separators ReadTags writes (src/BlockCompressor.cpp:524-566).
__global__ void __launch_bounds__(WG) k_dec_layout(const DecDesc* desc, DecState* st, RecPools rp, u8* out, const u8* d_stream, DecParams prm)
{
	const u32 b = blockIdx.y;
	DecState* S = &st[b];
	if (S->err) return;
	const DecDesc d = desc[b];
	u8* text = out + d.out_off;
	const u8* dsrc = d_stream + d.d_base;
	const u32 lane = lane_id();
	const u32 lossy = prm.lossy, off = prm.quality_offset;
	const u32 cs = (prm.color_space && (S->flags & 1u)) ? 1u : 0u;
	const u32 wpg = blockDim.x >> 6;
	const char* const dna_order = "AGCTNRWSKMDVHBYXU.-";          // src/RecordsProcessor.cpp:186-206; index 19 stays 255
	for (u32 r = blockIdx.x * wpg + wave_id(); r < S->n_recs; r += gridDim.x * wpg)
	{
		const u64 g = (u64)d.rec_base + r;
		const u32 ql = rp.len[g], qo_ = rp.qual_off[g], so = rp.seq_off[g], dof = rp.d_off[g], kept = rp.kept[g];
		const u32 to = rp.title_off[g], tl = rp.title_len[g];
		u32 base_rank = 0, bad = 0;
		for (u32 j0 = 0; j0 < ql; j0 += 64)
		{
			const u32 j = j0 + lane;
			const bool in = j < ql;
			const u32 q = in ? text[qo_ + j] : 0u;
			const bool special = in && q_special(q, lossy);
			const u64 m_kept = __ballot(in && !special);
			const u32 rank = base_rank + __popcll(m_kept & lanemask_lt());
			if (in)
			{
				u32 sv, qv;
				if (!lossy)
				{
					if (special) { sv = (q - 128u + 16u) / 8u + 2u; qv = q & 7u; }
					else { sv = rank < kept ? dsrc[dof + rank] : (bad = 1, 0u); qv = q; }
					qv = off + qv;
				}
				else
				{
					if (special) sv = 4; else sv = rank < kept ? dsrc[dof + rank] : (bad = 1, 0u);
					if (q >= 8) bad = 1;
					const u64 lq = 0x2825211B160F0600ull;          // 0, 6, 15, 22, 27, 33, 37, 40
					qv = off + (u32)((lq >> (8 * (q & 7u))) & 0xFFu);
				}
				u32 ch = 255;
				if (sv < 19) ch = (u8)dna_order[sv]; else if (sv > 19) bad = 1;
				text[so + j] = (u8)ch; text[qo_ + j] = (u8)qv;
			}
			base_rank += __popcll(m_kept);
		}
		if (cs)
		{
			// bases -> colours; record r's line starts one character earlier with the primer (constant over the block)
			wave_fence();
			const u32 sv0 = S->cs_seq_begin;
			const u32 c0 = sv0 < 19 ? (u32)(u8)dna_order[sv0] : 255u;
			u32 state = 0;                 // xor code of the last base that was one of ACGT: A0 C1 G2 T3; matrix A before any
			{
				const u32 x = c0 == 'A' ? 0u : c0 == 'C' ? 1u : c0 == 'G' ? 2u : c0 == 'T' ? 3u : 4u;
				if (x < 4) state = x;
			}
			for (u32 j0 = 0; j0 < ql; j0 += 64)
			{
				const u32 j = j0 + lane;
				const bool in = j < ql;
				const u32 ch = in ? text[so + j] : 0u;
				const u32 x = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
				const u64 m_acgt = __ballot(in && x < 4);
				// state seen by lane: the last ACGT among the lanes before it, else the carried one
				const u64 before = m_acgt & lanemask_lt();
				u32 st_l = state;
				const int src = before ? 63 - __clzll((long long)before) : -1;
				const u32 xs = (u32)__shfl((int)x, src < 0 ? 0 : src);
				if (src >= 0) st_l = xs;
				u32 col;
				if (x < 4) col = '.' + 2 + (st_l ^ x);
				else if (ch == 'N') col = '.';
				else col = '.' + 6;                                  // std::find(..) == end
				wave_fence();
				if (in) text[so + j] = (u8)col;
				if (m_acgt) { const int last = 63 - __clzll((long long)m_acgt); state = (u32)__shfl((int)x, last); }
			}
			if (lane == 0) { text[so - 1] = (u8)c0; text[qo_ - 1] = (u8)(S->cs_qua_begin + off); }
		}
		// separators and the plus line
		const u32 seq_end = so + ql, plus = seq_end + 1;
		if (lane == 0)
		{
			text[to + tl] = '\n'; text[seq_end] = '\n'; text[plus] = '+';
			text[plus + 1 + (prm.plus_rep ? tl - 1 : 0u)] = '\n';
			text[qo_ + ql] = '\n';
		}
		if (prm.plus_rep) for (u32 k = 1 + lane; k < tl; k += 64) text[plus + k] = text[to + k];
		if (__any(bad) && lane == 0) atomicOr(&S->err, (u32)DEC_ERR_FORMAT);
	}
}

// ---- stage 5: checksums of the decoded records (VerifyChecksum, src/BlockCompressor.cpp:576-594) ----------------------------
__global__ void __launch_bounds__(WG) k_dec_crc(const DecDesc* desc, DecState* st, RecPools rp, const u8* out, const u32* crc_tab, DecParams prm)
{
	__shared__ u32 s_tab[256 + 32];
	__shared__ u32 s_c[WAVES]; __shared__ u32 s_l[WAVES];
	const u32 b = blockIdx.x, which = blockIdx.y;                 // 0 tag, 1 sequence, 2 quality
	DecState* S = &st[b];
	if (S->err) return;
	const DecDesc d = desc[b];
	for (u32 i = threadIdx.x; i < 288; i += blockDim.x) s_tab[i] = crc_tab[i];
	__syncthreads();
	const u32* x2n = s_tab + 256;
	const u8* base = out + d.out_off;
	const u32 n = S->n_recs;
	const u32 cs = (prm.color_space && (S->flags & 1u)) ? 1u : 0u;
	u32 acc = 0;
	for (u32 base_r = 0; base_r < n; base_r += blockDim.x)
	{
		const u32 r = base_r + threadIdx.x;
		u32 c = 0, len = 0;
		if (r < n)
		{
			const u64 g = (u64)d.rec_base + r;
			const u32 o = which == 0 ? rp.title_off[g] : (which == 1 ? rp.seq_off[g] - cs : rp.qual_off[g] - cs);
			len = which == 0 ? rp.title_len[g] : rp.len[g] + cs;
			c = len ? crc_bytes(s_tab, base + o, len) : 0;
		}
		for (u32 dd = 1; dd < 64; dd <<= 1)
		{
			const u32 oc = __shfl_down(c, dd), ol = __shfl_down(len, dd);
			if ((lane_id() & (2 * dd - 1)) == 0 && lane_id() + dd < 64) { c = crc_combine(x2n, c, oc, ol); len += ol; }
		}
		if (lane_id() == 0) { s_c[wave_id()] = c; s_l[wave_id()] = len; }
		__syncthreads();
		if (threadIdx.x == 0)
			for (u32 w = 0; w < (blockDim.x >> 6); ++w) acc = crc_combine(x2n, acc, s_c[w], s_l[w]);
		__syncthreads();
	}
	if (threadIdx.x == 0) S->crc_actual[which] = acc;
}
