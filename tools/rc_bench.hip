// Micro-benchmark for the serial range-coder step (tools only; not part of the product).
// Variants isolate: loads, stores, the divide, the clamp branch.  Build: hipcc --offload-arch=gfx950 -O3 rc_bench.hip -o rc_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32; typedef unsigned long long u64; typedef unsigned char u8;
#define AHEAD 16
typedef u32 __attribute__((aligned(1))) u32_u;

__device__ __forceinline__ u32 rcdiv(u32 range, u64 magic) { const u32 t = __umulhi(range, (u32)magic); return (u32)(((u64)range * (u32)(magic >> 32) + t) >> 16); }

template <int V, int LPW = 64>
__global__ void __launch_bounds__(64) k(const u64* trip, const u64* magic_tab, u8* outp, u32 n, u32 stride, u32* sink)
{
	if (threadIdx.x >= LPW) return;
	const u32 lane = threadIdx.x + blockIdx.x * LPW;
	const u64* tp = trip + lane;
	u8* out = outp + (u64)lane * (2ull * n + 64);
	u64 low = 0; u32 range = 0xFFFFFFFFu, pos = 0; u32 bad = 0;
	u64 cur[AHEAD], mg[AHEAD], nxt[AHEAD];
#pragma unroll
	for (u32 i = 0; i < AHEAD; ++i) cur[i] = tp[(u64)i * stride];
#pragma unroll
	for (u32 i = 0; i < AHEAD; ++i) nxt[i] = tp[(u64)(AHEAD + i) * stride];
#pragma unroll
	for (u32 i = 0; i < AHEAD; ++i) mg[i] = magic_tab[(u32)(cur[i] >> 32) & 0xFFFF];
	for (u32 t0 = 0; t0 + 3 * AHEAD <= n; t0 += AHEAD)
	{
		u64 nm[AHEAD], far[AHEAD];
		if (V != 2)
		{
#pragma unroll
			for (u32 i = 0; i < AHEAD; ++i) nm[i] = magic_tab[(u32)(nxt[i] >> 32) & 0xFFFF];
#pragma unroll
			for (u32 i = 0; i < AHEAD; ++i) far[i] = tp[(u64)(t0 + 2 * AHEAD + i) * stride];
		}
#pragma unroll
		for (u32 i = 0; i < AHEAD; ++i)
		{
			const u64 e = cur[i];
			const u32 f = (u32)e & 0xFFFF, cum = (u32)(e >> 16) & 0xFFFF;
			u32 r;
			if (V == 5) r = range / ((u32)(e >> 32) | 1u); else r = rcdiv(range, mg[i]);
			low += (u64)r * cum;
			range = r * f;
			const u32 k8 = ((u32)__builtin_clz(range | 1u) >> 3) << 3;
			if (V == 0 || V == 5)
			{
				if (k8 && ((u32)(low >> 24) & 0xFFFFu) == 0xFFFFu) { bad++; }
			}
			else bad |= (((u32)(low >> 24) & 0xFFFFu) == 0xFFFFu) ? 1u : 0u;
			if (V != 1)
			{
				const u32 top = (u32)(low >> 32);
				*(u32_u*)(out + pos) = __builtin_bswap32(top);
			}
			pos += k8 >> 3;
			low <<= k8; range <<= k8;
			if (range < 0x1000000u) range |= 0x1000000u;     // keep the synthetic chain alive
		}
		if (V != 2)
		{
#pragma unroll
			for (u32 i = 0; i < AHEAD; ++i) { cur[i] = nxt[i]; mg[i] = nm[i]; nxt[i] = far[i]; }
		}
	}
	sink[lane] = pos + bad + (u32)low + range;
}


struct __attribute__((aligned(16))) Rec { u64 w; u32 cum; u32 pad; };
template <int V, int LPW = 64>
__global__ void __launch_bounds__(64) k6(const Rec* recs, u8* outp, u32 n, u32 stride, u32* sink)
{
	if (threadIdx.x >= LPW) return;
	const u32 lane = threadIdx.x + blockIdx.x * LPW;
	const Rec* tp = recs + lane;
	u8* out = outp + (u64)lane * (2ull * n + 64);
	u64 low = 0, acc = 0; u32 range = 0xFFFFFFFFu, pos = 0, nacc = 0; u32 bad = 0;
	Rec cur[AHEAD], nxt[AHEAD];
#pragma unroll
	for (u32 i = 0; i < AHEAD; ++i) cur[i] = tp[(u64)i * stride];
	for (u32 t0 = 0; t0 + 2 * AHEAD <= n; t0 += AHEAD)
	{
#pragma unroll
		for (u32 i = 0; i < AHEAD; ++i) nxt[i] = tp[(u64)(t0 + AHEAD + i) * stride];
#pragma unroll
		for (u32 i = 0; i < AHEAD; ++i)
		{
			const u64 e = cur[i].w;
			const u32 f = (u32)e & 0xFFFF;
			const u32 t = __umulhi(range, (u32)(e >> 16));
			const u32 r = (u32)(((u64)range * (u32)(e >> 48) + t) >> 16);
			low += (u64)r * cur[i].cum;
			range = r * f;
			const u32 k8 = ((u32)__builtin_clz(range | 1u) >> 3) << 3;
			bad |= (((u32)(low >> 24) & 0xFFFFu) == 0xFFFFu) ? 1u : 0u;
			acc = (acc << k8) | (u32)((low >> 8) >> (56 - k8));
			nacc += k8;
			low <<= k8; range <<= k8;
			if (range < 0x1000000u) range |= 0x1000000u;
			if (V == 0) { if (nacc >= 32) { *(u32_u*)(out + pos) = __builtin_bswap32((u32)(acc >> (nacc - 32))); pos += 4; nacc -= 32; } }
			else { if (nacc >= 64 - 24) { *(u32_u*)(out + pos) = __builtin_bswap32((u32)(acc >> (nacc - 32))); pos += 4; nacc -= 32; } }
		}
#pragma unroll
		for (u32 i = 0; i < AHEAD; ++i) cur[i] = nxt[i];
	}
	sink[lane] = pos + bad + (u32)low + range + nacc;
}

int main(int argc, char** argv)
{
	const u32 n = argc > 1 ? atoi(argv[1]) : 1000000, chains = argc > 2 ? atoi(argv[2]) : 512;
	const u32 stride = 64;
	std::vector<u64> h((size_t)n * stride);
	for (size_t i = 0; i < h.size(); ++i) { u32 x = (u32)(i * 2654435761u); u32 tot = 2000 + (x >> 8) % 60000; u32 f = 1 + (x >> 3) % 300; u32 c = (x >> 5) % (tot - f); h[i] = ((u64)tot << 32) | ((u64)c << 16) | f; }
	std::vector<u64> mg(65536, 0); for (u32 d = 1; d < 65536; ++d) mg[d] = ((1ull << 48) + d - 1) / d;
	u64 *d_t, *d_m; u8* d_o; u32* d_s;
	const u32 groups = chains / 64;
	hipMalloc(&d_t, h.size() * 8 * groups); hipMalloc(&d_m, mg.size() * 8); hipMalloc(&d_o, (size_t)chains * (2ull * n + 64)); hipMalloc(&d_s, chains * 4);
	for (u32 g = 0; g < groups; ++g) hipMemcpy(d_t + (size_t)g * h.size(), h.data(), h.size() * 8, hipMemcpyHostToDevice);
	hipMemcpy(d_m, mg.data(), mg.size() * 8, hipMemcpyHostToDevice);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	const char* names[] = {"v4-like (gather magic, store/sym, branchy precheck)", "no stores", "no loads in loop", "branch-free precheck", "-", "u32 divide instead of magic"};
	for (int v = 0; v < 6; ++v)
	{
		if (v == 4) continue;
		for (int rep = 0; rep < 2; ++rep)
		{
			hipEventRecord(a);
			// NB: every group reads its own copy of the triples (d_t + g*size) through blockIdx -> use one launch per variant
			switch (v)
			{
			case 0: hipLaunchKernelGGL(k<0>, dim3(groups), dim3(64), 0, 0, d_t, d_m, d_o, n, stride, d_s); break;
			case 1: hipLaunchKernelGGL(k<1>, dim3(groups), dim3(64), 0, 0, d_t, d_m, d_o, n, stride, d_s); break;
			case 2: hipLaunchKernelGGL(k<2>, dim3(groups), dim3(64), 0, 0, d_t, d_m, d_o, n, stride, d_s); break;
			case 3: hipLaunchKernelGGL(k<3>, dim3(groups), dim3(64), 0, 0, d_t, d_m, d_o, n, stride, d_s); break;
			case 5: hipLaunchKernelGGL(k<5>, dim3(groups), dim3(64), 0, 0, d_t, d_m, d_o, n, stride, d_s); break;
			}
			hipEventRecord(b); hipEventSynchronize(b);
			float ms; hipEventElapsedTime(&ms, a, b);
			if (rep) printf("variant %d %-55s: %8.2f ms  %7.1f ns/symbol\n", v, names[v], ms, ms * 1e6 / n);
		}
	}
	{
		std::vector<Rec> hr((size_t)n * stride);
		for (size_t i = 0; i < hr.size(); ++i) { u64 e = h[i]; u32 tot = (u32)(e >> 32); hr[i].w = (mg[tot] << 16) | (e & 0xFFFF); hr[i].cum = (u32)(e >> 16) & 0xFFFF; hr[i].pad = 0; }
		Rec* d_r; hipMalloc(&d_r, hr.size() * 16 * groups);
		for (u32 g = 0; g < groups; ++g) hipMemcpy(d_r + (size_t)g * hr.size(), hr.data(), hr.size() * 16, hipMemcpyHostToDevice);
		for (int v = 0; v < 2; ++v) for (int rep = 0; rep < 2; ++rep)
		{
			hipEventRecord(a);
			if (v == 0) hipLaunchKernelGGL(k6<0>, dim3(groups), dim3(64), 0, 0, d_r, d_o, n, stride, d_s);
			else hipLaunchKernelGGL(k6<1>, dim3(groups), dim3(64), 0, 0, d_r, d_o, n, stride, d_s);
			hipEventRecord(b); hipEventSynchronize(b);
			float ms; hipEventElapsedTime(&ms, a, b);
			if (rep) printf("variant 6.%d 16-byte records, accumulate, store per 4 bytes%s: %8.2f ms  %7.1f ns/symbol\n", v, v ? " (late flush)" : "", ms, ms * 1e6 / n);
		}
	}
	for (int lp = 0; lp < 3; ++lp) for (int rep = 0; rep < 2; ++rep)
	{
		const int L = lp == 0 ? 16 : (lp == 1 ? 8 : 32);
		hipEventRecord(a);
		if (lp == 0) hipLaunchKernelGGL((k<0, 16>), dim3(chains / 16), dim3(64), 0, 0, d_t, d_m, d_o, n, stride, d_s);
		else if (lp == 1) hipLaunchKernelGGL((k<0, 8>), dim3(chains / 8), dim3(64), 0, 0, d_t, d_m, d_o, n, stride, d_s);
		else hipLaunchKernelGGL((k<0, 32>), dim3(chains / 32), dim3(64), 0, 0, d_t, d_m, d_o, n, stride, d_s);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		if (rep) printf("variant 0 with %2d chains per wave: %8.2f ms  %7.1f ns/symbol\n", L, ms, ms * 1e6 / n);
	}
	for (int lp = 0; lp < 2; ++lp) for (int rep = 0; rep < 2; ++rep)
	{
		hipEventRecord(a);
		if (lp == 0) hipLaunchKernelGGL((k<1, 16>), dim3(chains / 16), dim3(64), 0, 0, d_t, d_m, d_o, n, stride, d_s);
		else hipLaunchKernelGGL((k<2, 16>), dim3(chains / 16), dim3(64), 0, 0, d_t, d_m, d_o, n, stride, d_s);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		if (rep) printf("16 chains per wave, %s: %8.2f ms  %7.1f ns/symbol\n", lp == 0 ? "no stores" : "no loads", ms, ms * 1e6 / n);
	}
	return 0;
}
