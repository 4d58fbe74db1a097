// Block assembly: meta stream + concatenation of the three byte-aligned sub-streams
// (BlockCompressor::StoreRecords / StoreMetaData, src/BlockCompressor.cpp:223-259,403-443),
// per-block CRC-32 (FastqChecksumHasher, src/RecordsProcessor.h:28-68; Crc32Hasher, src/Crc32.h:24-104).
#pragma once
#include "k_common.h"
#include "k_parse.h"

// AnalyzeMetaData (src/BlockCompressor.cpp:184-205) + meta size
__global__ void __launch_bounds__(64) k_meta_plan(BlkState* st, DsrcParams prm)
{
	const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= prm.n_blocks) return;
	BlkState* S = &st[b];
	u32 flags = S->flags & 4u;
	if ((u16)S->max_len != (u16)S->min_len) flags |= 2u;          // FLAG_VARIABLE_LENGTH
	if (S->cs_reduced) flags |= 1u;           // FLAG_DELTA_CONSTANT (src/BlockCompressor.cpp:190-199)
	S->flags = flags;
	u32 m = 16 + ((flags & 2u) ? 4u : 0u) + ((flags & 1u) ? 2u : 0u);
	if (prm.crc) m += (prm.tag_flags ? 0u : 4u) + 4 + (prm.lossy ? 0u : 4u);      // CALC_TAG only without -f (src/BlockCompressor.cpp:84-93)
	S->meta_bytes = m;
}

__device__ __forceinline__ void store_be32(u8* p, u32 v) { p[0] = (u8)(v >> 24); p[1] = (u8)(v >> 16); p[2] = (u8)(v >> 8); p[3] = (u8)v; }

__global__ void __launch_bounds__(WG) k_assemble(const BlkDesc* desc, const BlkState* st, const u32* word_pool, u8* out, DsrcParams prm)
{
	const u32 b = blockIdx.y;
	const BlkDesc d = desc[b];
	const BlkState* S = &st[b];
	if (S->err) return;
	u8* o = out + d.out_off;
	const u32 stride = gridDim.x * blockDim.x, tid = blockIdx.x * blockDim.x + threadIdx.x;
	if (tid == 0)
	{
		u32 at = 0;
		store_be32(o + at, S->n_recs); at += 4;
		const u32 red = S->flags & 1u;                                 // colour space, constant primer: lengths without it
		store_be32(o + at, (u16)(S->max_len - red)); at += 4;
		store_be32(o + at, S->flags); at += 4;
		store_be32(o + at, prm.record_layout ? d.chunk_size_value : (u32)((i32)(d.in_size - S->n_crlf) - S->title_cut)); at += 4;       // chunkSize = size - cut - skipped LFs (src/FastqParser.cpp:163,196)
		if (S->flags & 2u) { store_be32(o + at, (u16)(S->min_len - red)); at += 4; }
		if (red) { o[at++] = (u8)S->cs_seq_begin; o[at++] = (u8)S->cs_qua_begin; }      // src/BlockCompressor.cpp:415-422
		if (prm.crc)
		{
			if (!prm.tag_flags) { store_be32(o + at, S->crc_tag); at += 4; }
			store_be32(o + at, S->crc_seq); at += 4;
			if (!prm.lossy) { store_be32(o + at, S->crc_qua); at += 4; }
		}
	}
	// A stream is whole 32-bit words in the pool (big-endian ones where it was written through the bit sink): a thread moves a word --
	// one aligned load, one store to wherever the stream's bytes fall in the block (a byte per thread before round 5: 2.0 ms per 512
	// blocks for 1.4 GB) -- and the last thread of the stream its odd bytes.
	auto stream = [&](const u32* w, u64 at, u32 bytes, bool big_endian)
	{
		const u32 words = bytes >> 2;
		for (u32 k = tid; k < words; k += stride)
		{
			const u32 v = big_endian ? __builtin_bswap32(w[k]) : w[k];
			__builtin_memcpy(o + at + 4ull * k, &v, 4);
		}
		if (tid == 0)
		{
			const u8* src = (const u8*)w;
			for (u32 k = words * 4u; k < bytes; ++k) o[at + k] = src[k ^ (big_endian ? 3u : 0u)];
		}
	};
	u64 at = S->meta_bytes;
	stream(word_pool + d.tag_out, at, S->tag_bytes, true);
	at += S->tag_bytes;
	stream(word_pool + d.qua_out, at, S->qua_bytes, !(d.plain_mask & 1u));
	at += S->qua_bytes;
	stream(word_pool + d.dna_out, at, S->dna_bytes, !(d.plain_mask & 2u));
}

// ---- CRC-32 (poly 0xEDB88320, init/final 0xFFFFFFFF) ------------------------------------------------------
// crc(A||B) = crc(A) * x^(8|B|) mod P  xor  crc(B) for standard CRCs, so the three per-block checksums are an
// ordered reduction of per-record CRCs: one lane hashes one record, waves/workgroup combine pairwise in order.
__device__ __forceinline__ u32 crc_multmodp(u32 a, u32 b)
{
	u32 m = 1u << 31, p = 0;
	for (;;)
	{
		if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
		m >>= 1;
		b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
	}
	return p;
}

__device__ __forceinline__ u32 crc_x2nmodp(const u32* x2n, u64 n, u32 k)
{
	u32 p = 1u << 31;
	while (n)
	{
		if (n & 1) p = crc_multmodp(x2n[k & 31], p);
		n >>= 1; k++;
	}
	return p;
}

__device__ __forceinline__ u32 crc_combine(const u32* x2n, u32 c1, u32 c2, u64 len2)
{
	if (len2 == 0) return c1;
	return crc_multmodp(crc_x2nmodp(x2n, len2, 3), c1) ^ c2;
}

__device__ __forceinline__ u32 crc_bytes(const u32* tab, const u8* p, u32 n)
{
	u32 c = 0xFFFFFFFFu;
	for (u32 i = 0; i < n; ++i) c = (c >> 8) ^ tab[(p[i] ^ c) & 0xFFu];
	return c ^ 0xFFFFFFFFu;
}

// crc_tab: 256 table entries followed by 32 entries of x^(2^k) mod P
__global__ void __launch_bounds__(WG) k_crc(const u8* in, const BlkDesc* desc, BlkState* st, RecPools rp, const u32* crc_tab, DsrcParams prm)
{
	__shared__ u32 s_tab[256 + 32];
	__shared__ u32 s_c[WAVES]; __shared__ u32 s_l[WAVES];
	const u32 b = blockIdx.x, which = blockIdx.y;                 // 0 tag, 1 sequence, 2 quality
	BlkState* S = &st[b];
	const BlkDesc d = desc[b];
	for (u32 i = threadIdx.x; i < 288; i += blockDim.x) s_tab[i] = crc_tab[i];
	__syncthreads();
	const u32* x2n = s_tab + 256;
	const u8* base = in + d.in_off;
	const u32 n = S->n_recs;
	u32 acc = 0;                                                   // crc of everything so far (thread 0 only)
	for (u32 base_r = 0; base_r < n; base_r += blockDim.x)
	{
		const u32 r = base_r + threadIdx.x;
		u32 c = 0, len = 0;
		if (r < n)
		{
			const u64 g = (u64)d.rec_base + r;
			const u32 off = which == 0 ? rp.title_off[g] : (which == 1 ? rp.seq_off[g] : rp.qual_off[g]);
			len = which == 0 ? rp.title_len[g] : rp.len[g];
			c = len ? crc_bytes(s_tab, base + off, len) : 0;
		}
		// ordered pairwise combine inside the wave
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
	if (threadIdx.x == 0)
	{
		if (which == 0) S->crc_tag = acc; else if (which == 1) S->crc_seq = acc; else S->crc_qua = acc;
	}
}
