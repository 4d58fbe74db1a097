// Where do the 64-thread workgroups of a launch land?  Every workgroup records (XCC, SE, SH, CU, SIMD) and spins long enough for
// the whole grid to be resident at once; the host prints how many waves each SIMD got.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/hwid_probe tools/hwid_probe.hip && /tmp/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(256) k_probe(unsigned* out, unsigned spin)
{
	unsigned hw, xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	const unsigned wave = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
	if ((threadIdx.x & 63) == 0) { out[2 * wave] = hw; out[2 * wave + 1] = xcc; }
	unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < spin) { }
}

static void report(const char* what, const std::vector<unsigned>& v, unsigned n)
{
	std::map<unsigned, unsigned> per_simd, per_cu;
	for (unsigned i = 0; i < n; ++i)
	{
		const unsigned hw = v[2 * i], xcc = v[2 * i + 1] & 15;
		const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
		const unsigned cu_key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
		per_cu[cu_key]++; per_simd[(cu_key << 2) | simd]++;
	}
	std::map<unsigned, unsigned> hist;
	for (auto& kv : per_simd) hist[kv.second]++;
	printf("%-34s waves %5u  CUs used %3zu  SIMDs used %4zu  waves/SIMD histogram:", what, n, per_cu.size(), per_simd.size());
	for (auto& kv : hist) printf("  %u:%u", kv.first, kv.second);
	printf("\n");
}

int main()
{
	unsigned* d; hipMalloc(&d, 8 * 65536);
	std::vector<unsigned> h(2 * 65536);
	const unsigned spin = 100000 * 5;        // 100 MHz wall clock: 5 ms
	for (unsigned wg : {64u, 256u})
		for (unsigned waves : {300u, 1024u, 2048u, 2400u, 3600u, 4096u})
		{
			const unsigned grid = waves / (wg / 64);
			hipMemset(d, 0xFF, 8 * 65536);
			hipLaunchKernelGGL(k_probe, dim3(grid), dim3(wg), 0, 0, d, spin);
			hipDeviceSynchronize();
			hipMemcpy(h.data(), d, 8 * waves, hipMemcpyDeviceToHost);
			char what[64]; snprintf(what, sizeof(what), "one launch, %u-thread workgroups", wg);
			report(what, h, grid * (wg / 64));
		}
	// four launches of 300 waves on four streams
	hipStream_t s[4];
	for (int i = 0; i < 4; ++i) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
	hipMemset(d, 0xFF, 8 * 65536);
	hipDeviceSynchronize();
	for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(k_probe, dim3(300), dim3(64), 0, s[i], d + 2 * 300 * i, spin * 4);
	hipDeviceSynchronize();
	hipMemcpy(h.data(), d, 8 * 1200, hipMemcpyDeviceToHost);
	report("4 streams x 300 waves (64-thread)", h, 1200);
	return 0;
}
