// rc_lab: what one step of the range coder's dependent chain costs on gfx950, formulation by formulation.
// One workgroup, one wave alone on its SIMD, lane = chain (as k_rc's coder wave), 64 records per chain in LDS, coded over and over.
// Prints shader clocks (s_memtime) and nanoseconds per symbol.  hipcc --offload-arch=gfx950 -O3 -o tools/rc_lab tools/rc_lab.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64;
typedef u32 __attribute__((vector_size(16))) U4;
#define ROW_U4 49
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Rec { u32 a, b, c; };

// V0: the shipped step (k_rc.h rc_step_fast): m_lo, mf = m_hi << 16 | freq, cum
__device__ __forceinline__ u32 step_v0(u64& low_, u32& range_, Rec e, u32& flag)
{
	const u64 p = (u64)range_ * (e.b >> 16) + __umulhi(range_, e.a);
	const u32 r = __builtin_amdgcn_alignbit((u32)(p >> 32), (u32)p, 16);
	const u64 low = low_ + (u64)r * e.c;
	const u32 range = r * (e.b & 0xFFFFu);
	__builtin_assume(range != 0);
	const u32 k8 = (u32)__builtin_clz(range) & 0x18u;
	const u32 z = __builtin_amdgcn_alignbit((u32)(low >> 32), (u32)low, 8);
	flag = z > flag ? z : flag;
	low_ = low << k8; range_ = range << k8;
	return ((u32)(low >> 32) & 0xFFFFFF00u) | k8;
}
// V1: the range alone: a = top dword of ceil(2^48/total) << 16, b = its low dword, c = freq; returns r | (bytes leaving) << 30
__device__ __forceinline__ u32 step_v1(u32& range_, Rec e)
{
	const u64 p = (u64)range_ * e.a + __umulhi(range_, e.b);
	const u32 r = (u32)(p >> 32);
	const u32 range = r * e.c;
	__builtin_assume(range != 0);
	const u32 k8 = (u32)__builtin_clz(range) & 0x18u;
	range_ = range << k8;
	return r | (k8 << 27);
}
// V2: the quotient through f64: a,b = the double 1/total rounded up, c = freq
__device__ __forceinline__ u32 step_v2(u32& range_, Rec e)
{
	const double inv = __hiloint2double((int)e.a, (int)e.b);
	const u32 r = (u32)((double)range_ * inv);
	const u32 range = r * e.c;
	__builtin_assume(range != 0);
	const u32 k8 = (u32)__builtin_clz(range) & 0x18u;
	range_ = range << k8;
	return r | (k8 << 27);
}
// V4: V1 with r * freq from 24-bit multiplies (full rate): r = rh * 2^16 + rl
__device__ __forceinline__ u32 step_v4(u32& range_, Rec e)
{
	const u64 p = (u64)range_ * e.a + __umulhi(range_, e.b);
	const u32 r = (u32)(p >> 32);
	const u32 range = (__umul24(r >> 16, e.c) << 16) + __umul24(r & 0xFFFFu, e.c);
	__builtin_assume(range != 0);
	const u32 k8 = (u32)__builtin_clz(range) & 0x18u;
	range_ = range << k8;
	return r | (k8 << 27);
}
// V5: everything in f64: range kept as a double (exact: < 2^48), a,b = 1/total rounded up, c = freq as float bits
__device__ __forceinline__ u32 step_v5(double& range_, Rec e)
{
	const double inv = __hiloint2double((int)e.a, (int)e.b);
	const double r = __builtin_trunc(range_ * inv);
	const double range = r * (double)__uint_as_float(e.c);
	const int ex = __builtin_amdgcn_frexp_exp(range);            // range in [2^(ex-1), 2^ex)
	const u32 k8 = (u32)(32 - ex) & 0x18u;
	range_ = __builtin_amdgcn_ldexp(range, (int)k8);
	return (u32)r | (k8 << 27);
}

// V6: wave L's step as shipped: a = r | k8 << 27, b = freq | cum << 16
__device__ __forceinline__ u32 step_v6(u64& low_, Rec e, u32& flag)
{
	const u32 r = e.a & 0x3FFFFFFFu, k8 = (e.a >> 27) & 0x18u;
	const u64 low = low_ + (u64)r * (e.b >> 16);
	const u32 z = __builtin_amdgcn_alignbit((u32)(low >> 32), (u32)low, 16);
	flag = z > flag ? z : flag;
	low_ = low << k8;
	return ((u32)(low >> 32) & 0xFFFFFF00u) | k8;
}
// V7: the same with clean inputs: a = r, b = cum, c = k8
__device__ __forceinline__ u32 step_v7(u64& low_, Rec e, u32& flag)
{
	const u64 low = low_ + (u64)e.a * e.b;
	const u32 z = __builtin_amdgcn_alignbit((u32)(low >> 32), (u32)low, 16);
	flag = z > flag ? z : flag;
	low_ = low << e.c;
	return ((u32)(low >> 32) & 0xFFFFFF00u) | e.c;
}
// V8: V7 without 64-bit instructions: 32-bit product, add with carry, the shift from alignbit + a select (c = k8)
__device__ __forceinline__ u32 step_v8(u32& hi_, u32& lo_, Rec e, u32& flag)
{
	const u32 p = e.a * e.b;
	const u32 lo = lo_ + p;
	const u32 hi = hi_ + (lo < p ? 1u : 0u);
	const u32 z = __builtin_amdgcn_alignbit(hi, lo, 16);
	flag = z > flag ? z : flag;
	const u32 sh = __builtin_amdgcn_alignbit(hi, lo, 32u - e.c);       // c = 0: shift 32 = 0 mod 32 gives lo
	hi_ = e.c ? sh : hi;
	lo_ = lo << e.c;
	return (hi & 0xFFFFFF00u) | e.c;
}

template <int V> __global__ void __launch_bounds__(64) k_lab(const Rec* recs, u32 n_chunks, u64* out)
{
	__shared__ U4 rows[64 * ROW_U4];
	__shared__ U4 codes[64 * 17];
	const u32 lane = threadIdx.x;
	for (u32 i = 0; i < 64; ++i)
	{
		const Rec e = recs[lane * 64 + i];
		u32* d = (u32*)(rows + lane * ROW_U4) + 3 * i; d[0] = e.a; d[1] = e.b; d[2] = e.c;
	}
	__syncthreads();
	u64 low = 0; u32 range = 0xFFFFFFFFu; double ranged = 4294967295.0; u32 flag = 0, acc = 0;
	const U4* row = rows + lane * ROW_U4;
	const u64 t0 = clock64(), w0 = wall_clock64();
	for (u32 ch = 0; ch < n_chunks; ++ch)
	{
#pragma unroll
		for (u32 g = 0; g < 4; ++g)
		{
			U4 q[12];
			asm volatile("" ::: "memory");             // the rows are re-read every time, as in k_rc (where the loaders rewrite them)
#pragma unroll
			for (u32 i = 0; i < 12; ++i) q[i] = row[g * 12 + i];
			const u32* d = (const u32*)q;
			u32 c[16];
#pragma unroll
			for (u32 i = 0; i < 16; ++i)
			{
				Rec e; e.a = d[3 * i]; e.b = d[3 * i + 1]; e.c = d[3 * i + 2];
				if (V == 0) c[i] = step_v0(low, range, e, flag);
				else if (V == 1) c[i] = step_v1(range, e);
				else if (V == 2) c[i] = step_v2(range, e);
				else if (V == 4) c[i] = step_v4(range, e);
				else if (V == 6) c[i] = step_v6(low, e, flag);
				else if (V == 7) c[i] = step_v7(low, e, flag);
				else if (V == 8) { u32 hi = (u32)(low >> 32), lo = (u32)low; c[i] = step_v8(hi, lo, e, flag); low = ((u64)hi << 32) | lo; }
				else c[i] = step_v5(ranged, e);
			}
			if ((V == 0 || V >= 6) && (flag >> 16) == 0xFFFFu) { acc ^= 1; flag = 0; }
#pragma unroll
			for (u32 i = 0; i < 4; ++i) { const U4 v = {c[4 * i], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3]}; codes[lane * 17 + g * 4 + i] = v; }
		}
	}
	const u64 t1 = clock64(), w1 = wall_clock64();
	const u32* cw = (const u32*)(codes + lane * 17);
	for (u32 i = 0; i < 64; ++i) acc ^= cw[i];
	out[lane] = low ^ range ^ acc ^ (u64)ranged;
	if (lane == 0) { out[64] = t1 - t0; out[65] = w1 - w0; }
}


// ---- single instructions: a chain of dependent ones, and four independent chains (issue rate) ------------------------------------
#define OPS_LIST(X) X(0, "v_mad_u64_u32") X(1, "v_lshlrev_b64") X(2, "v_mul_lo_u32") X(3, "v_mul_hi_u32") X(4, "v_add_u32") X(5, "v_alignbit_b32") X(6, "v_lshlrev_b32") X(7, "v_mad_u32_u24") X(8, "v_lshl_add_u64") X(9, "v_perm_b32")
template <int OP> __device__ __forceinline__ void one_op(u64& a, u32 b, u32 c)
{
	if (OP == 0) a = (u64)(u32)a * b + a;
	else if (OP == 1) a = (a << (c & 8u)) | 1u;
	else if (OP == 2) a = (u32)a * b;
	else if (OP == 3) a = __umulhi((u32)a, b) | 0x10000u;
	else if (OP == 4) a = (u32)a + b;
	else if (OP == 5) a = __builtin_amdgcn_alignbit((u32)a, b, c);
	else if (OP == 6) a = ((u32)a << (c & 1u)) | 1u;
	else if (OP == 7) a = __umul24((u32)a, b) + c;
	else if (OP == 8) a = (a << 1) + (((u64)c << 32) | b);
	else a = __builtin_amdgcn_perm((u32)a, b, c);
}
template <int OP, int CHAINS> __global__ void __launch_bounds__(64) k_micro(u64* out, u32 n, u32 b, u32 c)
{
	u64 a[CHAINS];
	for (int k = 0; k < CHAINS; ++k) a[k] = threadIdx.x * 977u + k + 3u;
	const u64 t0 = clock64();
	for (u32 i = 0; i < n; ++i)
	{
#pragma unroll
		for (int u = 0; u < 16; ++u)
#pragma unroll
			for (int k = 0; k < CHAINS; ++k) { one_op<OP>(a[k], b, c); asm volatile("" : "+v"(a[k])); }
	}
	const u64 t1 = clock64();
	u64 x = 0; for (int k = 0; k < CHAINS; ++k) x ^= a[k];
	out[threadIdx.x] = x;
	if (threadIdx.x == 0) out[64] = t1 - t0;
}

// V3: the same range chain on the scalar unit: one chain per wave (every lane holds the same values; the compiler keeps
// uniform values in SGPRs), records through the scalar cache
template <int WAVES> __global__ void __launch_bounds__(64 * WAVES) k_lab_salu(const Rec* recs, u32 n_chunks, u64* out)
{
	const u32 w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const Rec* my = recs + (blockIdx.x * WAVES + w) * 64;
	u32 range = 0xFFFFFFFFu, acc = 0;
	const u64 t0 = clock64(), w0 = wall_clock64();
	for (u32 ch = 0; ch < n_chunks; ++ch)
	{
#pragma unroll 16
		for (u32 i = 0; i < 64; ++i)
		{
			const Rec e = my[i];
			const u64 p = (u64)range * e.a + __umulhi(range, e.b);
			const u32 r = (u32)(p >> 32);
			const u32 rg = r * e.c;
			__builtin_assume(rg != 0);
			const u32 k8 = (u32)__builtin_clz(rg) & 0x18u;
			range = rg << k8;
			acc += r | (k8 << 27);
		}
	}
	const u64 t1 = clock64(), w1 = wall_clock64();
	if ((threadIdx.x & 63) == 0) { out[blockIdx.x * WAVES + w] = range ^ acc; }
	if (threadIdx.x == 0 && blockIdx.x == 0) { out[4096] = t1 - t0; out[4097] = w1 - w0; }
}

static u64 recip48(u32 d) { return ((1ull << 48) + d - 1) / d; }

int main()
{
	const u32 n_chunks = 4096;
	std::vector<Rec> r0(64 * 64), r1(64 * 64), r2(64 * 64), r5(64 * 64);
	u64 seed = 12345;
	auto rnd = [&]() { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return (u32)(seed >> 33); };
	for (u32 i = 0; i < 64 * 64; ++i)
	{
		const u32 total = 4 + rnd() % 4000, freq = 1 + rnd() % total, cum = rnd() % (total - freq + 1);
		const u64 m = recip48(total);
		r0[i] = {(u32)m, ((u32)(m >> 32) << 16) | freq, cum};
		const u64 m2 = m << 16;
		r1[i] = {(u32)(m2 >> 32), (u32)m2, freq};
		double inv = 1.0 / (double)total;
		if (inv * (double)total < 1.0 || (total & (total - 1))) { u64 b; memcpy(&b, &inv, 8); ++b; memcpy(&inv, &b, 8); }
		u64 b; memcpy(&b, &inv, 8);
		r2[i] = {(u32)(b >> 32), (u32)b, freq};
		float ff = (float)freq; u32 fb; memcpy(&fb, &ff, 4);
		r5[i] = {(u32)(b >> 32), (u32)b, fb};
	}
	// wave L's inputs: r and k8 as wave R would leave them
	std::vector<Rec> r6(64 * 64), r7(64 * 64);
	for (u32 i = 0; i < 64 * 64; ++i)
	{
		const u32 r = rnd() & 0x3FFFFFFFu, k8 = (rnd() % 3u) * 8u, cum = rnd() & 0xFFFFu, freq = 1 + (rnd() & 0xFFFu);
		r6[i] = {r | (k8 << 27), freq | (cum << 16), 0}; r7[i] = {r, cum, k8};
	}
	Rec* d; u64* out;
	CHECK(hipMalloc(&d, 4096 * 64 * sizeof(Rec))); CHECK(hipMalloc(&out, 8192 * 8));
	std::vector<u64> h(8192);
	hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
	printf("device %s, clockRate %d kHz, wall clock 100 MHz\n", prop.name, prop.clockRate);
	auto report = [&](const char* name, u32 idx, double nsym)
	{
		CHECK(hipMemcpy(h.data(), out, 8192 * 8, hipMemcpyDeviceToHost));
		printf("%-44s %8.1f clk/symbol  %7.2f ns/symbol   (check %016llx)\n", name, (double)h[idx] / nsym, (double)h[idx + 1] * 10.0 / nsym, (unsigned long long)h[0]);
		return 0;
	};
	const double nsym = (double)n_chunks * 64;
#define RUN(V, R, NAME) { CHECK(hipMemcpy(d, R.data(), R.size() * sizeof(Rec), hipMemcpyHostToDevice)); \
	for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_lab<V>, dim3(1), dim3(64), 0, 0, d, n_chunks, out); CHECK(hipDeviceSynchronize()); } report(NAME, 64, nsym); }
	RUN(0, r0, "V0 shipped step (low + range + flag)")
	RUN(1, r1, "V1 range alone, reciprocal << 16")
	RUN(2, r2, "V2 range alone, quotient through f64")
	RUN(4, r1, "V4 V1 + r*freq from 24-bit multiplies")
	RUN(5, r5, "V5 range kept in f64 (frexp / ldexp)")
	RUN(6, r6, "V6 wave L's step (packed r | k8, freq | cum)")
	RUN(7, r7, "V7 wave L's step, clean r, cum, k8")
	RUN(8, r7, "V8 wave L's step, 32-bit instructions only")
#define MICRO(OP, NAME) { for (int ch = 1; ch <= 4; ch += 3) { if (ch == 1) hipLaunchKernelGGL((k_micro<OP, 1>), dim3(1), dim3(64), 0, 0, out, 4096u, 0x9E3779B1u, 8u); else hipLaunchKernelGGL((k_micro<OP, 4>), dim3(1), dim3(64), 0, 0, out, 4096u, 0x9E3779B1u, 8u); \
		CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(h.data(), out, 8192 * 8, hipMemcpyDeviceToHost)); printf("%-20s %d chain(s): %6.2f clk per instruction\n", NAME, ch, (double)h[64] / (4096.0 * 16 * ch)); } }
	OPS_LIST(MICRO)
	// scalar: 1, 2, 4, 8 waves per workgroup on one CU, then 4 waves on each of 256 workgroups
	{
		std::vector<Rec> big(4096 * 64);
		for (u32 i = 0; i < 4096 * 64; ++i) big[i] = r1[i % (64 * 64)];
		CHECK(hipMemcpy(d, big.data(), big.size() * sizeof(Rec), hipMemcpyHostToDevice));
#define RUNS(W, G, NAME) { for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_lab_salu<W>, dim3(G), dim3(64 * W), 0, 0, d, n_chunks, out); CHECK(hipDeviceSynchronize()); } report(NAME, 4096, nsym); }
		RUNS(1, 1, "V3 scalar chain, 1 wave")
		RUNS(4, 1, "V3 scalar chain, 4 waves on a CU")
		RUNS(8, 1, "V3 scalar chain, 8 waves on a CU")
		RUNS(16, 1, "V3 scalar chain, 16 waves on a CU")
		RUNS(4, 256, "V3 scalar chain, 4 waves x 256 workgroups")
		RUNS(8, 256, "V3 scalar chain, 8 waves x 256 workgroups")
	}
	return 0;
}
