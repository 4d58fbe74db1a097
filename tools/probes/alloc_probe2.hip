// Where is hipMalloc's cliff?  Pieces of 8 GiB one after the other up to 160 GiB, timed one by one, in a fresh process;
// then everything freed and allocated again; modes: 0 hipMalloc, 1 stream-ordered pool, 2 hipExtMallocWithFlags(uncached).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv)
{
	const int mode = argc > 1 ? atoi(argv[1]) : 0;
	const size_t piece = (size_t)(argc > 2 ? atoi(argv[2]) : 8) << 30;
	const int n = argc > 3 ? atoi(argv[3]) : 20;
	double t = now(); CK(hipFree(nullptr)); printf("mode %d: runtime up %.3f s\n", mode, now() - t);
	size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot)); printf("free %.1f of %.1f GB\n", fr / 1e9, tot / 1e9);
	std::vector<void*> ps;
	for (int round = 0; round < 2; ++round)
	{
		const double t0 = now();
		for (int i = 0; i < n; ++i)
		{
			void* p = nullptr; t = now();
			if (mode == 0) CK(hipMalloc(&p, piece));
			else if (mode == 1) { hipStream_t s = nullptr; CK(hipMallocAsync(&p, piece, s)); CK(hipStreamSynchronize(s)); }
			else if (mode == 2) CK(hipExtMallocWithFlags(&p, piece, hipDeviceMallocUncached));
			const double dt = now() - t;
			if (dt > 0.01) printf("  round %d piece %2d (%3zu GiB so far): %.3f s = %.1f GB/s\n", round, i, (size_t)(i + 1) * (piece >> 30), dt, piece / dt / 1e9);
			ps.push_back(p);
		}
		printf("round %d: %d x %zu GiB in %.3f s\n", round, n, piece >> 30, now() - t0);
		t = now();
		for (void* p : ps) { if (mode == 1) CK(hipFreeAsync(p, nullptr)); else CK(hipFree(p)); }
		if (mode == 1) CK(hipStreamSynchronize(nullptr));
		ps.clear();
		printf("  freed in %.3f s\n", now() - t);
	}
	return 0;
}
