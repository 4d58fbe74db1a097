// Bench input: counter-based Illumina-like FASTQ generated directly in HBM (SURVEY 8d generator,
// integer-only so that it matches dsrc_amd/synth.py illumina_fastq byte for byte).
// Not part of the compression path.
#pragma once
#include "k_common.h"

#define SYNTH_READ_LEN 150u
#define SYNTH_CHUNK 1024u

__device__ __forceinline__ u64 synth_mix64(u64 x)
{
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}

__device__ __forceinline__ u32 synth_digits(u64 v) { u32 d = 1; while (v >= 10) { v /= 10; ++d; } return d; }

__device__ __forceinline__ void synth_fields(u64 i, u32* lane, u32* tile, u32* x, u32* y)
{
	*lane = 1 + (u32)(((i - 1) / 250000) % 8);
	*tile = 1101 + (u32)(((i - 1) / 5000) % 96);
	*x = 1000 + (u32)((7 * i) % 20000);
	*y = 2000 + (u32)((13 * i) % 90000);
}

// "@SRRSYN.{i} HWI-ST1234:100:C0ABCACXX:{lane}:{tile}:{x}:{y} 1:N:0:ATCACG"
__device__ __forceinline__ u32 synth_title_len(u64 i)
{
	u32 lane, tile, x, y; synth_fields(i, &lane, &tile, &x, &y);
	return 8 + synth_digits(i) + 26 + 1 + 1 + 4 + 1 + synth_digits(x) + 1 + synth_digits(y) + 13;
}

__device__ __forceinline__ u32 synth_rec_size(u64 i) { return synth_title_len(i) + 1 + SYNTH_READ_LEN + 1 + 1 + 1 + SYNTH_READ_LEN + 1; }

__device__ __forceinline__ u8* synth_put_str(u8* p, const char* s) { while (*s) *p++ = (u8)*s++; return p; }
__device__ __forceinline__ u8* synth_put_num(u8* p, u64 v)
{
	const u32 d = synth_digits(v);
	for (u32 k = 0; k < d; ++k) { p[d - 1 - k] = (u8)('0' + v % 10); v /= 10; }
	return p + d;
}

__global__ void __launch_bounds__(256) k_synth_sizes(u64 first, u64 count, u64* chunk_tot)
{
	__shared__ u32 s_sum;
	if (threadIdx.x == 0) s_sum = 0;
	__syncthreads();
	u32 acc = 0;
	for (u32 k = threadIdx.x; k < SYNTH_CHUNK; k += blockDim.x)
	{
		const u64 r = (u64)blockIdx.x * SYNTH_CHUNK + k;
		if (r < count) acc += synth_rec_size(first + r);
	}
	atomicAdd(&s_sum, acc);
	__syncthreads();
	if (threadIdx.x == 0) chunk_tot[blockIdx.x] = s_sum;
}

__global__ void __launch_bounds__(256) k_synth_write(u64 first, u64 count, const u64* chunk_base, u8* out, u64 seed, u32 binned)
{
	__shared__ u32 s_off[SYNTH_CHUNK];
	// local exclusive scan of the record sizes of this chunk (serial per 256-thread stripe is enough here)
	for (u32 k = threadIdx.x; k < SYNTH_CHUNK; k += blockDim.x)
	{
		const u64 r = (u64)blockIdx.x * SYNTH_CHUNK + k;
		s_off[k] = r < count ? synth_rec_size(first + r) : 0;
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		u32 run = 0;
		for (u32 k = 0; k < SYNTH_CHUNK; ++k) { const u32 v = s_off[k]; s_off[k] = run; run += v; }
	}
	__syncthreads();
	for (u32 k = threadIdx.x; k < SYNTH_CHUNK; k += blockDim.x)
	{
		const u64 r = (u64)blockIdx.x * SYNTH_CHUNK + k;
		if (r >= count) continue;
		const u64 i = first + r;
		u8* p = out + chunk_base[blockIdx.x] + s_off[k];
		u32 lane, tile, x, y; synth_fields(i, &lane, &tile, &x, &y);
		p = synth_put_str(p, "@SRRSYN."); p = synth_put_num(p, i);
		p = synth_put_str(p, " HWI-ST1234:100:C0ABCACXX:"); p = synth_put_num(p, lane); *p++ = ':';
		p = synth_put_num(p, tile); *p++ = ':'; p = synth_put_num(p, x); *p++ = ':'; p = synth_put_num(p, y);
		p = synth_put_str(p, " 1:N:0:ATCACG"); *p++ = '\n';
		u8* q = p + SYNTH_READ_LEN + 3;
		for (u32 pos = 0; pos < SYNTH_READ_LEN; ++pos)
		{
			const u64 h1 = synth_mix64(((i << 10) | pos) ^ seed);
			const u64 h2 = synth_mix64(h1);
			const bool is_n = ((h1 >> 2) % 500) == 0;
			i32 sum = 0;
			for (u32 b = 0; b < 8; ++b) sum += (i32)((h2 >> (8 * b)) & 0xFF);
			const i32 num = (sum - 1020) * 4 + 104;
			const i32 z4 = num >= 0 ? num / 209 : -((-num + 208) / 209);       // floor division
			i32 qv = 38 - (i32)((6 * pos) / 100) + z4;
			qv = qv < 2 ? 2 : (qv > 40 ? 40 : qv);
			if (binned) qv = qv < 3 ? 2 : qv < 18 ? 12 : qv < 30 ? 23 : 37;      // flavour 1 of dsrcgpu_synth_fastq: four-level (NovaSeq-like) qualities
			p[pos] = is_n ? (u8)'N' : (u8)"ACGT"[h1 & 3];
			q[pos] = (u8)(33 + (is_n ? 2 : qv));
		}
		p[SYNTH_READ_LEN] = '\n'; p[SYNTH_READ_LEN + 1] = '+'; p[SYNTH_READ_LEN + 2] = '\n';
		q[SYNTH_READ_LEN] = '\n';
	}
}

// host driver; returns non-zero if the data does not fit
// binned: the same records with the qualities quantised to four levels (2, 12, 23, 37: what current instruments write)
static inline int synth_illumina_device(hipStream_t s, u64 first, u64 count, u8* d_out, u64 cap, u64* bytes, u32 binned)
{
	const u32 n_chunks = (u32)((count + SYNTH_CHUNK - 1) / SYNTH_CHUNK);
	if (n_chunks == 0) { *bytes = 0; return 0; }
	u64* d_tot = nullptr;
	if (hipMalloc((void**)&d_tot, (size_t)n_chunks * 8) != hipSuccess) return 2;
	hipLaunchKernelGGL(k_synth_sizes, dim3(n_chunks), dim3(256), 0, s, first, count, d_tot);
	u64* tot = (u64*)malloc((size_t)n_chunks * 8);
	hipMemcpyAsync(tot, d_tot, (size_t)n_chunks * 8, hipMemcpyDeviceToHost, s);
	hipStreamSynchronize(s);
	u64 run = 0;
	for (u32 i = 0; i < n_chunks; ++i) { const u64 v = tot[i]; tot[i] = run; run += v; }
	*bytes = run;
	int rc = 0;
	if (run > cap) rc = 1;
	else
	{
		hipMemcpyAsync(d_tot, tot, (size_t)n_chunks * 8, hipMemcpyHostToDevice, s);
		hipLaunchKernelGGL(k_synth_write, dim3(n_chunks), dim3(256), 0, s, first, count, d_tot, d_out, (u64)0xD5C0FFEEull, binned);
		hipStreamSynchronize(s);
	}
	free(tot); hipFree(d_tot);
	return rc;
}
