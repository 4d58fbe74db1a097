// The tag stream of a block with the WAVE on it: ReadTags (src/BlockCompressor.cpp:491-573) over TagTokenizerDecoder
// (src/TagModeler.cpp:887-1205).  k_dec_tags (k_dec.h) is the reference's loop on one lane: every byte of every title is a
// dependent global access there (~17 us per record).  Here lane j OWNS field j of the title format:
//   * its descriptor and its running state (previous value, run length) live in that lane's registers;
//   * per record only the NON-CONSTANT fields are visited one after the other -- that is the serial part, the bit stream; their
//     bits come from a register window, the Huffman trees of the header sit in LDS;
//   * then every lane writes its own field: constant text from registers, numbers formatted per lane, at offsets that are a
//     prefix sum of the field lengths.
// Blocks the fast form does not cover (raw / mixed formatting, more than 64 fields) take the one-lane loop.
#pragma once
#include "k_dec.h"
#include "k_dec_rc.h"

#define TAGW_LDS_NODES 6144u          // 24 KB of tree nodes and directories

// ---- the bit stream through the scalar cache ------------------------------------------------------------------------------------
// BitWin (k_dec.h) refills with vector loads, and a wave's wait for a vector load also waits for every store issued before it
// -- here the ~20 title bytes and record fields of the previous record, each a partial-line write.  The block is read-only
// during the pass, so the bits can come through s_load_dword (constant address space, own counter, two dwords requested ahead;
// the index is clamped to the block's last dword: what lies behind the end is never touched, a stream that runs there fails the
// position check at the end).
struct SWin { const CONST_AS u32* p4; u64 w; u32 n, q0, q1, nextw, lastw, fed; u64 origin; };

__device__ __forceinline__ u32 sw_at(const SWin& b, u32 i) { return __builtin_bswap32(b.p4[i < b.lastw ? i : b.lastw]); }
__device__ __forceinline__ void sw_init(SWin& b, const BitSrc& s)
{
	const u64 a = (u64)s.p, x = a + (s.bit >> 3);
	const u64 w0 = x & ~3ull;
	const u32 skip = (u32)(x & 3ull) * 8 + ((u32)s.bit & 7u);
	b.p4 = (const CONST_AS u32*)w0;
	b.lastw = (u32)((((a + s.size - 1) & ~3ull) - w0) >> 2);
	b.origin = (w0 - a) * 8;                                         // bit position (in the block, may be negative mod 2^64) of dword 0
	b.w = (((u64)sw_at(b, 0) << 32) | sw_at(b, 1)) << skip; b.n = 64 - skip;
	b.q0 = sw_at(b, 2); b.q1 = sw_at(b, 3); b.nextw = 4; b.fed = 2;
}
__device__ __forceinline__ void sw_refill(SWin& b)                    // keeps at least 32 valid bits
{
	if (b.n < 32)
	{
		b.w |= (u64)b.q0 << (32 - b.n);
		b.n += 32; ++b.fed;
		b.q0 = b.q1; b.q1 = sw_at(b, b.nextw); ++b.nextw;
	}
}
__device__ __forceinline__ u32 sw_bits(SWin& b, u32 n)               // n <= 32
{
	if (n == 0) return 0;
	sw_refill(b);
	const u32 v = (u32)(b.w >> (64 - n));
	b.w <<= n; b.n -= n;
	return v;
}
__device__ __forceinline__ void sw_finish(const SWin& b, BitSrc& s)
{
	s.bit = b.origin + (u64)b.fed * 32 - b.n;
	if (s.bit > (u64)s.size * 8) s.err |= DEC_ERR_TRUNC;
}
template <typename PT>
__device__ __forceinline__ u32 sw_huff(SWin& b, PT T, u32* err)
{
	u32 node = 0;
	for (u32 round = 0; round < 2; ++round)
	{
		sw_refill(b);
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

__device__ __forceinline__ u32 tagw_lane(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane((int)l)); }

template <typename PT>
__device__ __forceinline__ void tags_records_wave(BitSrc& s, PT W, const DecField* F, u32 nf, DecState* S, const DecDesc& d,
												  RecPools rp, u8* text, const DecParams& prm, u32* pos_out, u32* q_total_out)
{
	const u32 lane = lane_id();
	const bool have = lane < nf;
	const u32 cap = d.out_cap;
	const u32 len_bits = dec_bit_length((u64)(S->max_qlen - S->min_qlen));
	const u32 cs_delta = (prm.color_space && (S->flags & 1u)) ? 1u : 0u;
	const u32 min_qlen = S->min_qlen, max_qlen = S->max_qlen, n_recs = S->n_recs;
	const u64 r0 = d.rec_base;

	// ---- this lane's field ------------------------------------------------------------------------------------------------
	DecField f; memset(&f, 0, sizeof(f));
	if (have) f = F[lane];
	const u32 kind = !have ? 0u : f.is_constant ? 0u : f.is_numeric ? 1u : 2u;          // 0 constant, 1 numeric, 2 string
	const u32 meta = (u32)f.scheme | ((u32)f.has_global << 8) | (f.bits_value << 16) | (f.bits_num << 24);
	const bool delta = f.scheme == NS_DELTA_CONST || f.scheme == NS_DELTA_RLE || f.scheme == NS_DELTA_VAR;
	const u32 addc = delta ? (u32)f.min_delta : (u32)f.min_value;
	const u32 min_value = (u32)f.min_value;
	u32 prev = 0, rle_len = 0, rle_sym = 0;
	u32 mylen = kind == 0 ? f.len : 0u;                                                 // bytes of this field in the current title
	u32 myval = 0;
	// the first 16 bytes this lane writes: constant text once, digits per record
	u64 ob_lo = 0, ob_hi = 0;
	const u8* ctext = s.p + f.data_pos;
	if (have && kind == 0)
		for (u32 k = 0; k < 16 && k < f.len; ++k) { const u64 c = ctext[k]; if (k < 8) ob_lo |= c << (8 * k); else ob_hi |= c << (8 * (k - 8)); }
	// bytes of constant fields (with their separators) in front of this field
	const u32 c_in = (have && kind == 0) ? f.len + 1 : 0u;
	const u32 cpre = dec_wave_scan(c_in) - c_in;
	const u64 nc_mask = __ballot(have && kind != 0);

	SWin bw; sw_init(bw, s);
	u32 pos = 0, q_total = 0, err = 0;
	for (u32 i = 0; i < n_recs && !err; ++i)
	{
		const u32 t0 = pos;
		// ---- the bit stream: non-constant fields in order ---------------------------------------------------------------------
		u32 dyn = 0;
		for (u64 m = nc_mask; m; m &= m - 1)
		{
			const u32 j = (u32)__ffsll((long long)m) - 1u;
			const u32 kj = tagw_lane(kind, j);
			u32 flen;
			if (kj == 1)
			{	// TagTokenizerDecoder::ReadNumericField (src/TagModeler.cpp:1098-1205)
				const u32 mj = tagw_lane(meta, j);
				const u32 scheme = mj & 0xFFu, has_global = (mj >> 8) & 1u, bits_value = (mj >> 16) & 0xFFu, bits_num = mj >> 24;
				const u32 pj = tagw_lane(prev, j), aj = tagw_lane(addc, j);
				u32 rl = tagw_lane(rle_len, j), rs = tagw_lane(rle_sym, j);
				u32 v, res;
				if (i == 0)
				{
					v = sw_bits(bw, bits_value);
					if (scheme == NS_VALUE_RLE) { rl = sw_bits(bw, 8); rs = v; }
					res = v + tagw_lane(min_value, j);
				}
				else if (scheme == NS_DELTA_CONST) res = pj + aj;
				else if (scheme == NS_DELTA_RLE)
				{
					if (i == 1 || rl == 0) { v = sw_bits(bw, bits_num); rs = v; rl = sw_bits(bw, 8); }
					else { rl--; v = rs; }
					res = v + pj + aj;
				}
				else if (scheme == NS_VALUE_VAR || scheme == NS_DELTA_VAR)
				{
					v = has_global ? sw_huff(bw, W + tagw_lane(f.global_tree, j), &err) : sw_bits(bw, bits_num);
					res = scheme == NS_DELTA_VAR ? v + pj + aj : v + aj;
				}
				else if (scheme == NS_VALUE_RLE)
				{
					if (rl == 0) { v = sw_bits(bw, bits_num); rs = v; rl = sw_bits(bw, 8); }
					else { rl--; v = rs; }
					res = v + aj;
				}
				else { err |= DEC_ERR_FORMAT; res = 0; }
				// core::to_string (src/utils.h:69-97); values >= 10^9 overflow `power` in the reference
				if (res >= 1000000000u) err |= DEC_ERR_REF_UB;
				flen = 1;
				for (u32 p10 = 10; flen < 10 && res >= p10; p10 *= 10) ++flen;
				if (lane == j) { prev = res; myval = res; rle_len = rl; rle_sym = rs; mylen = flen; }
			}
			else
			{	// string field: length, then per position the template character or a Huffman symbol; written as it is decoded
				const u32 f_len = tagw_lane(f.len, j);
				flen = tagw_lane((u32)f.is_len_constant, j) ? f_len : sw_bits(bw, tagw_lane(f.bits_len, j)) + tagw_lane(f.min_len, j);
				const u32 off = t0 + tagw_lane(cpre, j) + dyn;
				const u32 data_pos = tagw_lane(f.data_pos, j), ham_bit = tagw_lane(f.ham_bit, j), dir = tagw_lane(f.local_dir, j);
				for (u32 k = 0; k < flen && !err; ++k)
				{
					bool fixed = false;
					if (k < f_len) { BitSrc t = s; t.bit = (u64)ham_bit + k; fixed = bs_bit(t) != 0; }
					u32 c;
					if (fixed) c = s.p[data_pos + k];
					else
					{
						const u32 tr = W[dir + (k < 128u ? k : 128u)];
						if (tr == 0xFFFFFFFFu) { err |= DEC_ERR_FORMAT; break; }
						c = sw_huff(bw, W + tr, &err);
					}
					if (off + k < cap) { if (lane == 0) text[off + k] = (u8)c; } else err |= DEC_ERR_TEXT;
				}
				if (lane == j) mylen = flen;
			}
			dyn += flen + 1;
		}
		// ---- the text: every lane its field ---------------------------------------------------------------------------------------
		const u32 L = have ? mylen + 1 : 0u;
		const u32 incl = dec_wave_scan(L);
		const u32 off = t0 + incl - L;
		const u32 tl = tagw_lane(incl, 63) - 1;                           // the last separator is not part of the title
		if (kind == 1)
		{
			u32 v = myval;
			ob_lo = 0; ob_hi = 0;
#pragma unroll
			for (u32 t = 0; t < 10; ++t)
			{
				if (t < mylen)
				{
					const u32 at = mylen - 1 - t;
					const u64 c = '0' + v % 10; v /= 10;
					if (at < 8) ob_lo |= c << (8 * at); else ob_hi |= c << (8 * (at - 8));
				}
			}
		}
		const u32 elen = (have && kind != 2) ? mylen : 0u;
		bool over = false;                                                // this lane ran out of the text reserved for the block
		for (u32 k = 0; __any(k < elen); ++k)
		{
			if (k < elen)
			{
				const u32 c = k < 8 ? (u32)(ob_lo >> (8 * k)) & 0xFFu : k < 16 ? (u32)(ob_hi >> (8 * (k - 8))) & 0xFFu : (u32)ctext[k];
				if (off + k < cap) text[off + k] = (u8)c; else over = true;
			}
		}
		if (have && lane + 1 < nf) { if (off + mylen < cap) text[off + mylen] = f.sep; else over = true; }
		if (__any(over)) err |= DEC_ERR_TEXT;
		pos = t0 + tl;
		pos++;                                                            // '\n'
		const u32 ql = len_bits ? sw_bits(bw, len_bits) + min_qlen : max_qlen;
		if (tl > 65535u || ql > 65535u) { err |= DEC_ERR_FORMAT; break; }
		const u64 g = r0 + i;
		const u32 so = pos + cs_delta; pos += ql + cs_delta + 1;           // sequence line + '\n'
		pos += 1 + (prm.plus_rep ? tl - 1 : 0u) + 1;                       // '+' [title] '\n'
		const u32 qo = pos + cs_delta; pos += ql + cs_delta + 1;
		if (lane == 0) { rp.title_off[g] = t0; rp.title_len[g] = (u16)tl; rp.len[g] = (u16)ql; rp.seq_off[g] = so; rp.qual_off[g] = qo; }
		q_total += ql;
		if (pos > cap) err |= DEC_ERR_TEXT;
	}
	sw_finish(bw, s);
	s.err |= err;
	*pos_out = pos; *q_total_out = q_total;
}

__global__ void __launch_bounds__(64) k_dec_tags_wave(const u8* in, const DecDesc* desc, DecState* st, RecPools rp, u8* out, u32* pool, u8* fld_pool, DecParams prm)
{
	__shared__ u32 s_nodes[TAGW_LDS_NODES];
	__shared__ DecField s_F[64];
	__shared__ u32 s_par[6];
	const u32 b = blockIdx.x;
	DecState* S = &st[b];
	if (S->err) return;                                   // wave-uniform
	__builtin_amdgcn_s_setprio(2);                        // a short stage at the head of a pass: ahead of another pass's quality waves
	const DecDesc d = desc[b];
	BitSrc s; s.p = in + d.in_off; s.size = d.in_size; s.err = 0; s.bit = (u64)S->tag_pos * 8;
	NodePool np; np.w = pool + d.node_off; np.cap = d.node_cap; np.top = 0;
	const bool mixed = (S->flags & 4u) != 0;
	u8* text = out + d.out_off;
	u32 pos = 0, q_total = 0;
	bool fast = false;
	if (!mixed) { BitSrc t = s; fast = bs_byte(t) <= 64u; }
	DecField* F = fast ? (DecField*)s_F : (DecField*)(fld_pool + d.fld_off);
	if (threadIdx.x == 0)
	{
		TagHead H;
		tags_header(s, np, F, mixed, H);
		// trees that do not fit the LDS stay with the lane that wrote them
		const bool wave = fast && np.top <= TAGW_LDS_NODES && !s.err;
		if (!wave && !s.err) tags_records_serial(s, np, F, H, mixed, S, d, rp, text, prm, &pos, &q_total);
		s_par[0] = H.nf; s_par[1] = np.top; s_par[2] = (u32)s.bit; s_par[3] = (u32)(s.bit >> 32); s_par[4] = wave ? 1u : 0u;
	}
	__syncthreads();
	if (s_par[4])
	{
		const u32 nf = s_par[0], top = s_par[1];
		s.bit = ((u64)s_par[3] << 32) | s_par[2];
		// the nodes were written by lane 0 a moment ago: read them past this CU's vector cache
		for (u32 i = threadIdx.x; i < top; i += blockDim.x) s_nodes[i] = __hip_atomic_load(np.w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__syncthreads();
		tags_records_wave(s, (const LDS_AS u32*)s_nodes, s_F, nf, S, d, rp, text, prm, &pos, &q_total);
	}
	if (threadIdx.x == 0) tags_finish(s, S, prm, pos, q_total);
}
