// k_index: FASTQ chunk -> record index            (FastqParser::ParseFrom, src/FastqParser.cpp:140-164)
// k_prep : per-base transform + block statistics   (Lossless/LossyRecordsProcessor::ProcessForward,
//                                                    src/RecordsProcessor.cpp:209-267,344-408; FinalizeStats :112-133)
//
// Layout: the raw chunk text stays where the host put it (read twice, coalesced 16 B/lane);
// the index is SoA (line starts, per-record offsets/lengths); the transformed symbols go to
// two dense per-block streams (quality, DNA) so that every later stage is a flat array pass.
#pragma once
#include "k_common.h"

// A byte is the LAST byte of a line terminator iff it is '\n', or a '\r' not followed by '\n'
// (SkipLine, src/FastqParser.h:93-115: "\r\n", "\n" and a lone "\r" all end a line).
//
// One lane classifies 64 consecutive bytes with word arithmetic: per 8-byte word an exact byte-equality mask
// (bit 7 of every byte that equals c), gathered to 8 bits by a multiply, gives 64-bit maps of the '\n' and '\r'
// positions; terminators and "\r\n" pairs are then two shifts and three logic ops.
typedef u64 __attribute__((aligned(1))) u64_text;

__device__ __forceinline__ u32 byte_eq8(u64 w, u64 c_rep)
{
	const u64 x = w ^ c_rep;
	const u64 y = (((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x) | 0x7F7F7F7F7F7F7F7Full;   // 0xFF.. except 0x7F where the byte of x is 0
	return (u32)((((~y) >> 7) * 0x0102040810204080ull) >> 56);                                            // bit k = byte k matched
}

struct LineBits { u64 term, crlf; };   // bit i: byte base+i is the last byte of a terminator / is the '\n' of a "\r\n"

__device__ __forceinline__ LineBits classify64(const u8* p, u64 base, u64 size)
{
	LineBits r; r.term = 0; r.crlf = 0;
	if (base >= size) return r;
	u64 lf = 0, cr = 0;
	if (base + DSRC_LANE_BYTES <= size)
	{
#pragma unroll
		for (u32 k = 0; k < DSRC_LANE_BYTES / 8; ++k)
		{
			const u64 w = *(const u64_text*)(p + base + 8 * k);
			lf |= (u64)byte_eq8(w, 0x0A0A0A0A0A0A0A0Aull) << (8 * k);
			cr |= (u64)byte_eq8(w, 0x0D0D0D0D0D0D0D0Dull) << (8 * k);
		}
	}
	else
	{	// last lane of the chunk: byte by byte, never past `size`
		for (u32 k = 0; base + k < size; ++k)
		{
			const u8 c = p[base + k];
			if (c == '\n') lf |= 1ull << k;
			if (c == '\r') cr |= 1ull << k;
		}
	}
	const u64 n_valid = size - base;
	const u64 valid = n_valid >= 64 ? ~0ull : ((1ull << n_valid) - 1ull);
	const bool lf_next = base + DSRC_LANE_BYTES < size && p[base + DSRC_LANE_BYTES] == '\n';
	const bool cr_prev = base > 0 && p[base - 1] == '\r';
	const u64 lf_after = (lf >> 1) | (lf_next ? 1ull << 63 : 0ull);      // bit i: byte i+1 is '\n'
	const u64 cr_before = (cr << 1) | (cr_prev ? 1ull : 0ull);          // bit i: byte i-1 is '\r'
	r.term = (lf | (cr & ~lf_after)) & valid;
	r.crlf = (lf & cr_before) & valid;
	return r;
}

// ---- pass 1: count terminators per tile --------------------------------------------------------
__global__ void __launch_bounds__(WG) k_count_lines(const u8* in, const BlkDesc* desc, BlkState* st, u32* tile_cnt, DsrcParams prm)
{
	__shared__ u32 s_tot[2];
	const u32 b = blockIdx.y, tile = blockIdx.x;
	const BlkDesc d = desc[b];
	if (tile >= d.n_tiles) return;
	if (threadIdx.x < 2) s_tot[threadIdx.x] = 0;
	__syncthreads();
	const LineBits lb = classify64(in + d.in_off, (u64)tile * DSRC_TILE_BYTES + (u64)threadIdx.x * DSRC_LANE_BYTES, d.in_size);
	const u32 cnt = wave_sum((u32)__popcll(lb.term)), crlf = wave_sum((u32)__popcll(lb.crlf));
	if (lane_id() == 0) { if (cnt) atomicAdd(&s_tot[0], cnt); if (crlf) atomicAdd(&s_tot[1], crlf); }
	__syncthreads();
	if (threadIdx.x == 0)
	{
		tile_cnt[(u64)b * prm.max_tiles + tile] = s_tot[0];
		atomicAdd(&st[b].n_term, s_tot[0]);
		if (s_tot[1]) atomicAdd(&st[b].n_crlf, s_tot[1]);
	}
}

// ---- pass 2: exclusive scan of the tile counts of one block (in place) --------------------
__global__ void __launch_bounds__(WG) k_scan_tiles(const BlkDesc* desc, u32* tile_cnt, DsrcParams prm)
{
	const u32 b = blockIdx.x;
	const u32 n = desc[b].n_tiles;
	u32* t = tile_cnt + (u64)b * prm.max_tiles;
	u32 carry = 0;
	for (u32 base = 0; base < n; base += blockDim.x)
	{
		const u32 i = base + threadIdx.x;
		const u32 v = i < n ? t[i] : 0;
		u32 tot;
		const u32 ex = block_excl_scan(v, &tot);
		if (i < n) t[i] = carry + ex;
		carry += tot;
	}
}

// ---- pass 3: line starts ------------------------------------------------------------------
__global__ void __launch_bounds__(WG) k_index_lines(const u8* in, const BlkDesc* desc, const u32* tile_base, u32* line_start, DsrcParams prm)
{
	const u32 b = blockIdx.y, tile = blockIdx.x;
	const BlkDesc d = desc[b];
	if (tile >= d.n_tiles) return;
	const u64 base = (u64)tile * DSRC_TILE_BYTES + (u64)threadIdx.x * DSRC_LANE_BYTES;
	const LineBits lb = classify64(in + d.in_off, base, d.in_size);
	u64 mask = lb.term;
	u32 total;
	u32 rank = block_excl_scan((u32)__popcll(mask), &total) + tile_base[(u64)b * prm.max_tiles + tile];
	u32* ls = line_start + d.line_base;
	if (tile == 0 && threadIdx.x == 0) ls[0] = 0;
	while (mask)
	{
		const u32 k = (u32)__ffsll((long long)mask) - 1;
		mask &= mask - 1;
		// bit 31 (chunks are < 2^31 bytes): the terminator that ends in front of this line is "\r\n" -- k_records then needs no byte of
		// the text to know where the line before ends (it fetched 18 MB per 8 MiB chunk for those two bytes per line, round 5)
		ls[++rank] = (u32)(base + k + 1) | ((u32)((lb.crlf >> k) & 1ull) << 31);
	}
}

// per-record SoA index
struct RecPools
{
	u32* title_off; u32* seq_off; u32* qual_off;
	u16* title_len; u16* len;            // sequenceLen == qualityLen for every valid record
	u16* kept; u16* trunc;               // after preprocessing: kept bases, truncatedLen
	u32* q_off; u32* d_off;              // exclusive prefix of len / kept inside the block's streams
};

#define LS_POS(x) ((x) & 0x7FFFFFFFu)
__device__ __forceinline__ void line_extent(const u32* ls, u32 n_term, u32 size, u32 i, u32* start, u32* len)
{
	if (i > n_term) { *start = size; *len = 0; return; }
	const u32 s = LS_POS(ls[i]);
	*start = s;
	if (i == n_term) { *len = size - s; return; }           // last line: no terminator (chunk.size excludes it)
	const u32 nx = ls[i + 1];
	const u32 e = LS_POS(nx) - 1 - (nx >> 31);                 // first byte of the terminator ("\r\n": two bytes, k_index_lines)
	*len = e - s;
}

// ---- pass 4: records = 4 lines; validate like ReadNextRecord (src/FastqParser.h:40-60) -----
__global__ void __launch_bounds__(WG) k_records(const u8* in, const BlkDesc* desc, BlkState* st, const u32* line_start, RecPools rp)
{
	const u32 b = blockIdx.y;
	const BlkDesc d = desc[b];
	const u32 n_term = st[b].n_term;
	const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
	const u32 n_cand = (n_term + 1 + 3) / 4;
	if (r >= n_cand || r >= d.rec_cap) return;
	const u8* p = in + d.in_off;
	const u32* ls = line_start + d.line_base;
	u32 s0, l0, s1, l1, s2, l2, s3, l3;
	line_extent(ls, n_term, d.in_size, 4 * r + 0, &s0, &l0);
	line_extent(ls, n_term, d.in_size, 4 * r + 1, &s1, &l1);
	line_extent(ls, n_term, d.in_size, 4 * r + 2, &s2, &l2);
	line_extent(ls, n_term, d.in_size, 4 * r + 3, &s3, &l3);
	const bool ok = s0 < d.in_size && l0 > 0 && p[s0] == '@' && l2 > 0 && l1 == l3;
	if (!ok) atomicMin(&st[b].first_bad, r);
	if (l0 > 65535u || l1 > 65535u || l3 > 65535u) atomicOr(&st[b].err, (u32)DSRC_ERR_LONG_LINE);
	const u64 g = (u64)d.rec_base + r;
	rp.title_off[g] = s0; rp.title_len[g] = (u16)l0;
	rp.seq_off[g] = s1;   rp.len[g] = (u16)l1;
	rp.qual_off[g] = s3;
}

// base -> index LUT (src/RecordsProcessor.cpp:186-206): A0 G1 C2 T3 N4 R5 W6 S7 K8 M9 D10 V11 H12 B13 Y14 X15 U16 .17 -18
__device__ __forceinline__ u32 dna_index(u32 c)
{
	switch (c)
	{
	case 'A': return 0; case 'G': return 1; case 'C': return 2; case 'T': return 3; case 'N': return 4;
	case 'R': return 5; case 'W': return 6; case 'S': return 7; case 'K': return 8; case 'M': return 9;
	case 'D': return 10; case 'V': return 11; case 'H': return 12; case 'B': return 13; case 'Y': return 14;
	case 'X': return 15; case 'U': return 16; case '.': return 17; case '-': return 18;
	default: return 255;
	}
}

// The switch above compiles into a tree of divergent branches: with four or five different bases in a wave every path runs.
// The same function as selects on two packed constants (5-bit entries for 'A'..'Y'), checked against the switch for all
// 256 characters at compile time.
constexpr u32 dna_index_packed(u32 c, u64 K0, u64 K1)
{
	const u32 i = c - 'A';
	const bool lo = i < 12;
	const u64 k = lo ? K0 : K1;
	const u32 j = lo ? i : i - 12;
	const u32 sh = 5 * (j < 12 ? j : 12);
	u32 v = i < 25 ? (u32)(k >> sh) & 31u : 31u;
	v = c == '.' ? 17u : v;
	v = c == '-' ? 18u : v;
	return v == 31u ? 255u : v;
}
#define DNA_INDEX_K0 0x0fa3ff607ff509a0ull
#define DNA_INDEX_K1 0xe7997019cbfffc89ull
constexpr u32 dna_index_switch_c(u32 c)
{
	switch (c)
	{
	case 'A': return 0; case 'G': return 1; case 'C': return 2; case 'T': return 3; case 'N': return 4;
	case 'R': return 5; case 'W': return 6; case 'S': return 7; case 'K': return 8; case 'M': return 9;
	case 'D': return 10; case 'V': return 11; case 'H': return 12; case 'B': return 13; case 'Y': return 14;
	case 'X': return 15; case 'U': return 16; case '.': return 17; case '-': return 18;
	default: return 255;
	}
}
constexpr bool dna_index_same()
{
	for (u32 c = 0; c < 256; ++c) if (dna_index_packed(c, DNA_INDEX_K0, DNA_INDEX_K1) != dna_index_switch_c(c)) return false;
	return true;
}
static_assert(dna_index_same(), "dna_index: packed table differs from src/RecordsProcessor.cpp:186-206");
template <bool FAST> __device__ __forceinline__ u32 dna_index_t(u32 c) { return FAST ? dna_index_packed(c, DNA_INDEX_K0, DNA_INDEX_K1) : dna_index(c); }

// lossy Illumina 8-bin quantiser (src/RecordsProcessor.cpp:318-341)
__device__ __forceinline__ u32 lossy_bin(u32 q)
{
	if (q < 2) return 0; if (q < 10) return 1; if (q < 20) return 2; if (q < 25) return 3;
	if (q < 30) return 4; if (q < 35) return 5; if (q < 40) return 6; if (q < 64) return 7;
	return 255;
}

// k_prep_stats and k_prep_write take the select form of dna_index (12.2 -> 5.5 and 10.4 -> 5.x ms per 512 blocks); false = the switch
#ifndef FAST_STATS
#define FAST_STATS true
#endif
#ifndef FAST_WRITE
#define FAST_WRITE true
#endif
// one base: returns transformed quality, sets *keep / *sidx
__device__ __forceinline__ u32 transform_base_s(u32 s, u32 qual, u32 qoff, u32 lossy, u32* sidx, bool* keep);
template <bool FAST = false>
__device__ __forceinline__ u32 transform_base(u32 base, u32 qual, u32 qoff, u32 lossy, u32* sidx, bool* keep)
{
	return transform_base_s(dna_index_t<FAST>(base), qual, qoff, lossy, sidx, keep);
}
// ... with the symbol index of the base looked up by the caller (k_prep_stats / k_prep_write: a 256-byte table in LDS -- the lanes of
// a request meet in four or five of its bytes, which the LDS hands out in one go -- instead of ~20 selects per base)
__device__ __forceinline__ void dna_index_table(u8* lut) { for (u32 i = threadIdx.x; i < 256; i += blockDim.x) lut[i] = (u8)dna_index(i); }
__device__ __forceinline__ u32 transform_base_s(u32 s, u32 qual, u32 qoff, u32 lossy, u32* sidx, bool* keep)
{
	*sidx = s;
	u32 q;
	if (!lossy)
	{
		q = (qual - qoff) & 255u;
		if (s > 3 && q < 7) { q = (q + 128u + ((s - 2u) << 3) - 16u) & 255u; *keep = false; }
		else *keep = true;
	}
	else
	{
		q = lossy_bin((qual - qoff) & 255u);
		if (s >= 4) { q = 0; *keep = false; }
		else { if (q == 0) q = 1; *keep = true; }
	}
	return q;
}

// ---- colour space: colours -> bases in place, one wave per record ---------------------------------
// IRecordsProcessor::ProcessRecordFromColorSpace (src/RecordsProcessor.cpp:25-58): character k >= 1 is looked up in the
// transition matrix of the last decoded base that was one of ACGT (matrix A before any).  With A,C,G,T = 0,1,2,3 the
// matrices are `state xor colour`, '.' and '/' give N and leave the state alone, so the state at k is the primer's
// code xor the prefix-xor of the colours: two ballots (one per colour bit) and a popcount parity per 64 characters.
// Also ColorSpaceStats::constBeginSym (src/RecordsProcessor.h:92-105) on the raw primer characters.
__device__ __forceinline__ u32 block_rec_count(const BlkState& S, const BlkDesc& d)
{
	u32 n_cand = (S.n_term + 1 + 3) / 4;
	if (n_cand > d.rec_cap) n_cand = d.rec_cap;
	return S.first_bad < n_cand ? S.first_bad : n_cand;
}

// n_recs ahead of k_prep_stats, for the kernels the colour-space path runs before it
__global__ void __launch_bounds__(64) k_rec_count(const BlkDesc* desc, BlkState* st, u32 n_blocks)
{
	const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b < n_blocks) st[b].n_recs = block_rec_count(st[b], desc[b]);
}

__device__ __forceinline__ u32 cs_state(u32 c) { return c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u; }

__global__ void __launch_bounds__(WG) k_cs_decode(u8* in, const BlkDesc* desc, BlkState* st, RecPools rp)
{
	const u32 b = blockIdx.y;
	const BlkDesc d = desc[b];
	u8* p = in + d.in_off;
	const u32 n_recs = st[b].n_recs;
	const u32 lane = lane_id();
	const u32 waves_total = gridDim.x * (blockDim.x >> 6);
	for (u32 r = blockIdx.x * (blockDim.x >> 6) + wave_id(); r < n_recs; r += waves_total)
	{
		const u64 g = (u64)d.rec_base + r;
		const u32 len = rp.len[g], so = rp.seq_off[g];
		if (len == 0) continue;
		if (lane == 0 && p[so] != p[rp.seq_off[d.rec_base]]) atomicOr(&st[b].cs_varbegin, 1u);
		u32 state = cs_state(p[so]);
		bool bad = false;
		for (u32 j0 = 0; j0 < len; j0 += 64)
		{
			const u32 j = j0 + lane;
			const bool in_r = j < len && j >= 1;
			const u32 c = in_r ? (u32)p[so + j] - '.' : 0u;          // '.' '/' '0' '1' '2' '3' -> 0..5
			if (c > 5) bad = true;                                    // the reference reads outside its 24-entry table
			const u32 col = (in_r && c >= 2 && c <= 5) ? c - 2 : 0u;
			const u64 m0 = __ballot((col & 1u) != 0), m1 = __ballot((col & 2u) != 0);
			const u64 le = lanemask_lt() | (1ull << lane);
			const u32 mine = state ^ ((u32)__popcll(m0 & le) & 1u) ^ (((u32)__popcll(m1 & le) & 1u) << 1);
			if (in_r) p[so + j] = c < 2 ? (u8)'N' : (u8)"ACGT"[mine];
			state ^= ((u32)__popcll(m0) & 1u) ^ (((u32)__popcll(m1) & 1u) << 1);
		}
		if (bad) atomicOr(&st[b].err, (u32)DSRC_ERR_BAD_BASE);
	}
}

// ---- pass 5: statistics, one wave per record ---------------------------------------------------
// Grid: x = part, y = block.  With many blocks in a batch a block is one workgroup's (gridDim.x = 1); a batch of few, large blocks (the
// reference's -m1 / -m2: 56 or 14 blocks of 64 / 256 MiB) gives every block to several workgroups, each taking every gridDim.x-th group of
// records: their sums meet in the block's state through atomics (zeroed with the state; min_len starts at ~0: k_init_state) and the
// workgroup that arrives last (a counter in scratch[7]) does what follows from them.
#define STATS_PARTS_SLOT 7
__global__ void __launch_bounds__(WG) k_prep_stats(const u8* in, const BlkDesc* desc, BlkState* st, RecPools rp, DsrcParams prm)
{
	__shared__ u32 s_qf[256];
	__shared__ u32 s_df[20];
	__shared__ u32 s_acc[8];      // rle, th, raw(qual), min, max, raw_tag, raw_dna, bad_base
	__shared__ u8 s_dna[256];
	__shared__ u32 s_last;
	dna_index_table(s_dna);       // (the barrier below)
	const u32 b = blockIdx.y, parts = gridDim.x;
	const BlkDesc d = desc[b];
	const u8* p = in + d.in_off;
	const u32 n_recs = block_rec_count(st[b], d);

	for (u32 i = threadIdx.x; i < 256; i += blockDim.x) s_qf[i] = 0;
	if (threadIdx.x < 20) s_df[threadIdx.x] = 0;
	if (threadIdx.x < 8) s_acc[threadIdx.x] = (threadIdx.x == 3) ? 0xFFFFFFFFu : 0u;
	__syncthreads();

	const u32 lane = lane_id();
	u32 a_rle = 0, a_th = 0, a_raw = 0, a_min = 0xFFFFFFFFu, a_max = 0, a_tag = 0, a_bad = 0;
	// The next record's pool entries and its first PS_AHEAD x 64 characters are requested while the current one is processed.  A
	// request made inside the record's own loop is waited for at once, and that wait (vmcnt(0)) also covers whatever has been asked
	// for the next record: with the characters beyond the first 64 and the title length fetched where they were needed, a read of 150
	// cost three memory round trips (round 5).  Every lane asks (past the end of the read for the chunk's first byte): a request
	// under a branch cannot be counted on by a later wait.
	constexpr u32 PS_AHEAD = 3;
	const u32 wstep = (blockDim.x >> 6) * parts, w0 = blockIdx.x * (blockDim.x >> 6) + wave_id();
	u32 n_len = 0, n_so = 0, n_qo = 0, n_tl = 0, n_b[PS_AHEAD], n_q[PS_AHEAD];
#pragma unroll
	for (u32 k = 0; k < PS_AHEAD; ++k) { n_b[k] = 'A'; n_q[k] = 0; }
	auto request = [&](u64 g)
	{
		n_len = rp.len[g]; n_so = rp.seq_off[g]; n_qo = rp.qual_off[g]; n_tl = rp.title_len[g];
		const u32 last = n_len ? n_len - 1u : 0u;
#pragma unroll
		for (u32 k = 0; k < PS_AHEAD; ++k)
		{
			// (a lane past the end of the read asks for the read's last character: an index clamped with v_min.  Written as a choice
			// between two addresses -- `have ? n_so + j : 0` -- the compiler turned the request into branches around two loads and a
			// copy behind `s_waitcnt vmcnt(0)`: three memory round trips per record in the one place meant to hide them)
			const u32 j = min_u32(64u * k + lane, last);
			n_b[k] = p[n_so + j]; n_q[k] = p[n_qo + j];
		}
	};
	if (w0 < n_recs) request((u64)d.rec_base + w0);
	for (u32 r = w0; r < n_recs; r += wstep)
	{
		const u64 g = (u64)d.rec_base + r;
		const u32 len = n_len, so = n_so, qo = n_qo, tl = n_tl;
		u32 c_b[PS_AHEAD], c_q[PS_AHEAD];
#pragma unroll
		for (u32 k = 0; k < PS_AHEAD; ++k) { c_b[k] = n_b[k]; c_q[k] = n_q[k]; }
		if (r + wstep < n_recs) request(g + wstep);
		u32 kept = 0, th = 0, rle = 0, carry = 255, lastq = 255;
		auto chunk = [&](const u32 j0, const u32 cb_in, const u32 cq_in)
		{
			const u32 j = j0 + lane;
			const bool in_r = j < len;
			u32 sidx = 0, q = 0; bool keep = false;
			{	// the transform on every lane, `in_r` applied afterwards: the form k_prep_write needed on gfx950 (see there); lanes past the
				// end of the read take a harmless stand-in
				const u32 cb = in_r ? cb_in : (u32)'A', cq = in_r ? cq_in : prm.quality_offset + 40u;
				const u32 qq = transform_base_s((u32)s_dna[cb & 255u], cq, prm.quality_offset, prm.lossy, &sidx, &keep);      // (the table: one LDS read for ~20 selects)
				if (in_r) q = qq; else { keep = false; sidx = 0; }
			}
			const bool k2 = in_r && keep;
			if (in_r && sidx >= 20) a_bad = 1;
			if (k2 && sidx < 20) atomicAdd(&s_df[sidx], 1u);
			kept += (u32)__popcll(__ballot(k2));
			const u32 prev = wave_shr1(q, carry);
			const u64 change = __ballot(in_r && q != prev);
			rle += (u32)__popcll(change);
			// the quality histogram: the lanes of an atomic that meet in a counter are applied one after the other, and four-level
			// qualities bring all 64 to three or four counters (k_prep_stats 8.0 ms per 512 blocks against 5.1).  Where the runs are
			// long (<= 16 changes in the 64) only the first lane of a run adds, the run's length.
			if (__popcll(change) <= 16)
			{
				const u64 starts = change | 1ull;                                          // (lane 0 starts one whatever came before)
				const u64 after = (starts >> lane) >> 1;
				const u32 n_in = len - j0 < 64u ? len - j0 : 64u;                          // the lanes of the read are the first n_in
				const u32 run = after ? (u32)__ffsll((long long)after) : n_in - lane;
				if (in_r && ((starts >> lane) & 1ull)) atomicAdd(&s_qf[q], run);
			}
			else if (in_r) atomicAdd(&s_qf[q], 1u);
			const u64 m2 = __ballot(in_r && q != 2);
			if (m2) th = j0 + 63u - (u32)__clzll((long long)m2);
			carry = (u32)__builtin_amdgcn_readlane((int)q, 63);
			const u32 last_lane = (len - 1 - j0) < 64u ? (len - 1 - j0) : 63u;
			lastq = (u32)__builtin_amdgcn_readlane((int)q, (int)last_lane);
		};
#pragma unroll
		for (u32 k = 0; k < PS_AHEAD; ++k) if (64u * k < len) chunk(64u * k, c_b[k], c_q[k]);
		for (u32 j0 = 64u * PS_AHEAD; j0 < len; j0 += 64)
		{
			const u32 j = min_u32(j0 + lane, len - 1u);
			chunk(j0, (u32)p[so + j], (u32)p[qo + j]);
		}
		if (len > 0 && lastq == 2 && rle > 0) rle -= 1;     // per-record decrement (Appendix B.13)
		if (lane == 0)
		{
			rp.kept[g] = (u16)kept;
			rp.trunc[g] = (u16)(th + (len > 0 ? 1u : 0u));
			a_rle += rle; a_th += th; a_raw += len; a_tag += tl;
			a_min = len < a_min ? len : a_min; a_max = len > a_max ? len : a_max;
		}
	}
	if (lane == 0)
	{
		atomicAdd(&s_acc[0], a_rle); atomicAdd(&s_acc[1], a_th); atomicAdd(&s_acc[2], a_raw);
		atomicMin(&s_acc[3], a_min); atomicMax(&s_acc[4], a_max); atomicAdd(&s_acc[5], a_tag);
	}
	if (a_bad) atomicOr(&s_acc[7], 1u);
	__syncthreads();

	BlkState* S = &st[b];
	if (parts > 1)
	{	// this part's sums into the block's; the last part to arrive takes the totals back into its LDS and goes on as the only one would
		for (u32 i = threadIdx.x; i < 256; i += blockDim.x) if (s_qf[i]) atomicAdd(&S->q_freq[i], s_qf[i]);
		if (threadIdx.x < 20 && s_df[threadIdx.x]) atomicAdd(&S->d_freq[threadIdx.x], s_df[threadIdx.x]);
		if (threadIdx.x == 0)
		{
			atomicAdd(&S->rle_len, s_acc[0]); atomicAdd(&S->th_len, s_acc[1]); atomicAdd(&S->raw_len, s_acc[2]);
			atomicMin(&S->min_len, s_acc[3]); atomicMax(&S->max_len, s_acc[4]); atomicAdd(&S->raw_tag, s_acc[5]);
			if (s_acc[7]) atomicOr(&S->err, (u32)DSRC_ERR_BAD_BASE);
		}
		__threadfence();
		__syncthreads();
		if (threadIdx.x == 0) s_last = atomicAdd(&S->scratch[STATS_PARTS_SLOT], 1u) == parts - 1u ? 1u : 0u;
		__syncthreads();
		if (!s_last) return;
		__threadfence();
		// (read where the atomics were applied, not through this CU's vector cache)
		for (u32 i = threadIdx.x; i < 256; i += blockDim.x) s_qf[i] = atomicAdd(&S->q_freq[i], 0u);
		if (threadIdx.x < 20) s_df[threadIdx.x] = atomicAdd(&S->d_freq[threadIdx.x], 0u);
		if (threadIdx.x == 0)
		{
			s_acc[0] = atomicAdd(&S->rle_len, 0u); s_acc[1] = atomicAdd(&S->th_len, 0u); s_acc[2] = atomicAdd(&S->raw_len, 0u);
			s_acc[3] = atomicAdd(&S->min_len, 0u); s_acc[4] = atomicAdd(&S->max_len, 0u); s_acc[5] = atomicAdd(&S->raw_tag, 0u);
			s_acc[7] = 0;                                                        // (already in S->err)
			S->scratch[STATS_PARTS_SLOT] = 0;
		}
		__syncthreads();
	}
	else
	{
		for (u32 i = threadIdx.x; i < 256; i += blockDim.x) S->q_freq[i] = s_qf[i];
		if (threadIdx.x < 20) S->d_freq[threadIdx.x] = s_df[threadIdx.x];
	}
	if (threadIdx.x == 0)
	{
		S->n_recs = n_recs;
		S->n_lines = S->n_term + 1;
		S->rle_len = s_acc[0]; S->th_len = s_acc[1]; S->raw_len = s_acc[2];
		S->min_len = s_acc[3]; S->max_len = s_acc[4];
		S->raw_tag = s_acc[5]; S->raw_dna = s_acc[2]; S->raw_qua = s_acc[2];
		u32 e = 0;
		if (n_recs == 0) e |= DSRC_ERR_NO_RECORDS;
		if (s_acc[7]) e |= DSRC_ERR_BAD_BASE;
		if (e) atomicOr(&S->err, e);
		// FinalizeStats (src/RecordsProcessor.cpp:112-133): dense ranks of the present symbols
		u32 dc = 0, qc = 0;
		for (u32 i = 0; i < 20; ++i) S->d_sym[i] = s_df[i] ? (u8)dc++ : (u8)255;
		for (u32 i = 0; i < 256; ++i) S->q_sym[i] = s_qf[i] ? (u8)qc++ : (u8)255;
		S->d_count = dc; S->q_count = qc;
	}
}

// ---- colour space with a constant primer: every record loses its first kept base and its first quality
// AFTER the statistics were taken (AnalyzeMetaData + the "2nd pass" of AnalyzeTags, src/BlockCompressor.cpp:184-199,
// 380-393).  From here on the record pools describe the shortened records; k_prep_write rebuilds the full view.
__global__ void __launch_bounds__(WG) k_cs_reduce(const u8* in, const BlkDesc* desc, BlkState* st, RecPools rp, DsrcParams prm)
{
	const u32 b = blockIdx.y;
	BlkState* S = &st[b];
	if (S->cs_varbegin) return;
	const BlkDesc d = desc[b];
	const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= S->n_recs) return;
	const u64 g = (u64)d.rec_base + r;
	const u32 len = rp.len[g];
	if (r == 0)
	{
		// csSeqBegin = records[0].sequence[0] after ProcessForward (index of the first kept base, or of base 0 if none
		// is kept); csQuaBegin = records[0].quality[0] after it
		const u8* p = in + d.in_off;
		u32 first = 255, q0 = 0;
		for (u32 j = 0; j < len && first == 255; ++j)
		{
			u32 sidx; bool keep;
			const u32 q = transform_base(p[rp.seq_off[g] + j], p[rp.qual_off[g] + j], prm.quality_offset, prm.lossy, &sidx, &keep);
			if (j == 0) q0 = q;
			if (keep) first = sidx;
		}
		S->cs_seq_begin = first != 255 ? first : (len ? dna_index(p[rp.seq_off[g]]) : 0u);
		S->cs_qua_begin = q0;
		S->cs_reduced = 1;
	}
	if (rp.kept[g] < 1 || len < 2) { atomicOr(&S->err, (u32)DSRC_ERR_BAD_BASE); return; }      // lengths wrap in the reference
	rp.len[g] = (u16)(len - 1); rp.qual_off[g] += 1; rp.kept[g] -= 1;
	if (rp.trunc[g] > 0) rp.trunc[g] -= 1;
}

// ---- pass 6: per-record stream offsets (exclusive scans of len / kept) ----------------------
__global__ void __launch_bounds__(WG) k_rec_offsets(const BlkDesc* desc, BlkState* st, RecPools rp)
{
	const u32 b = blockIdx.x;
	const BlkDesc d = desc[b];
	const u32 n = st[b].n_recs;
	u32 cq = 0, cd = 0;
	for (u32 base = 0; base < n; base += blockDim.x)
	{
		const u32 r = base + threadIdx.x;
		const u64 g = (u64)d.rec_base + r;
		const u32 vq = r < n ? rp.len[g] : 0, vd = r < n ? rp.kept[g] : 0;
		u32 tq, td;
		const u32 eq = block_excl_scan(vq, &tq);
		const u32 ed = block_excl_scan(vd, &td);
		if (r < n) { rp.q_off[g] = cq + eq; rp.d_off[g] = cd + ed; }
		cq += tq; cd += td;
	}
	if (threadIdx.x == 0) { st[b].q_total = cq; st[b].d_total = cd; }
}

// ---- pass 7: write the dense symbol streams --------------------------------------------------
// q_stream: transformed quality value per base; qp_stream: floor(j*128/len) (position context,
// any power-of-two rescale R is qp >> (7 - log2 R)); d_stream: kept base indices, compacted.
__global__ void __launch_bounds__(WG) k_prep_write(const u8* in, const BlkDesc* desc, const BlkState* st, RecPools rp,
												   u8* q_stream, u8* qp_stream, u8* d_stream, DsrcParams prm)
{
	__shared__ u8 s_dna[256];
	dna_index_table(s_dna);
	__syncthreads();
	const u32 b = blockIdx.y;
	const BlkDesc d = desc[b];
	const u8* p = in + d.in_off;
	const u32 n_recs = st[b].n_recs;
	const u32 lane = lane_id();
	const bool write_qp = st[b].min_len != st[b].max_len;      // reads of one length: the position context is a closed form of t (qua_pctx)
	const u32 red = st[b].cs_reduced;      // k_cs_reduce: first quality and first kept base are not coded
	const u32 waves_total = gridDim.x * (blockDim.x >> 6);
	// A record is a chain of dependent requests -- its pool entries, then its characters 64 at a time, each waited for at once: four
	// memory round trips per read of 150, and the CU's 32 waves were too few to hide them (round 5: 5.1 ms per 512 blocks).  The
	// next record's pool entries are requested a record ahead, and the first PW_AHEAD x 64 characters of a record all at once, by
	// every lane (past the end of the read: the chunk's first byte).
	constexpr u32 PW_AHEAD = 3;
	u32 n_rlen = 0, n_so = 0, n_qo = 0, n_qoff = 0, n_doff = 0;
	auto request = [&](u64 g) { n_rlen = rp.len[g]; n_so = rp.seq_off[g]; n_qo = rp.qual_off[g]; n_qoff = rp.q_off[g]; n_doff = rp.d_off[g]; };
	const u32 r0 = blockIdx.x * (blockDim.x >> 6) + wave_id();
	if (r0 < n_recs) request((u64)d.rec_base + r0);
	for (u32 r = r0; r < n_recs; r += waves_total)
	{
		const u64 g = (u64)d.rec_base + r;
		const u32 rlen = n_rlen, len = rlen + red, so = n_so, qo = n_qo - red;
		u8* qs = q_stream + d.q_base + n_qoff;
		u8* qps = qp_stream + d.q_base + n_qoff;
		u8* ds = d_stream + d.d_base + n_doff;
		u32 c_b[PW_AHEAD], c_q[PW_AHEAD];
#pragma unroll
		for (u32 k = 0; k < PW_AHEAD; ++k)
		{
			const u32 j = min_u32(64u * k + lane, len ? len - 1u : 0u);           // (past the end of the read: its last character -- see k_prep_stats)
			c_b[k] = p[so + j]; c_q[k] = p[qo + j];
		}
		if (r + waves_total < n_recs) request(g + waves_total);
		u32 run = 0;
		auto chunk = [&](const u32 j0, const u32 cb_in, const u32 cq_in)
		{
			const u32 j = j0 + lane;
			const bool in_r = j < len;
			u32 sidx = 0, q = 0; bool keep = false;
			{	// every lane evaluates the (branch-free) transform, lanes past the end of the read on a harmless stand-in: with the call
				// inside `if (in_r)` this kernel gave a wrong quality stream on the GPU once dna_index had become selects (DESIGN.md
				// section 10; both forms as a stand-alone kernel: tools/prep_write_repro.hip) -- the emulator build, the function on its own and
				// k_prep_stats with the same call were all right
				const u32 cb = in_r ? cb_in : (u32)'A', cq = in_r ? cq_in : prm.quality_offset + 40u;
				const u32 qq = transform_base_s((u32)s_dna[cb & 255u], cq, prm.quality_offset, prm.lossy, &sidx, &keep);
				if (in_r) q = qq; else { keep = false; sidx = 0; }
			}
			const bool k2 = in_r && keep;
			const u64 km = __ballot(k2);
			if (in_r && j >= red) { qs[j - red] = (u8)q; if (write_qp) qps[j - red] = (u8)(((j - red) * 128u) / rlen); }
			const u32 at = run + (u32)__popcll(km & lanemask_lt());
			if (k2 && at >= red) ds[at - red] = (u8)sidx;
			run += (u32)__popcll(km);
		};
#pragma unroll
		for (u32 k = 0; k < PW_AHEAD; ++k) if (64u * k < len) chunk(64u * k, c_b[k], c_q[k]);
		for (u32 j0 = 64u * PW_AHEAD; j0 < len; j0 += 64)
		{
			const u32 j = min_u32(j0 + lane, len - 1u);
			chunk(j0, (u32)p[so + j], (u32)p[qo + j]);
		}
	}
}
