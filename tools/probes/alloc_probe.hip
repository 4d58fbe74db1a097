// What start-up is made of: hipMalloc / hipFree by size, from one and from four threads; hipHostMalloc; host->device copies from
// pageable, page-locked and registered memory.  hipcc --offload-arch=gfx950 -O2 -o alloc_probe alloc_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main()
{
	double t = now(); CK(hipFree(nullptr)); printf("runtime up: %.3f s\n", now() - t);
	for (size_t gb : {1, 4, 10, 24, 46})
	{
		void* p = nullptr; t = now(); CK(hipMalloc(&p, gb << 30)); double a = now() - t;
		t = now(); CK(hipMemset(p, 1, gb << 30)); CK(hipDeviceSynchronize()); double m = now() - t;
		t = now(); CK(hipFree(p)); double f = now() - t;
		printf("hipMalloc %2zu GiB: %.3f s   memset %.3f s   hipFree %.3f s\n", gb, a, m, f);
	}
	{	// second time (does the runtime keep anything?)
		void* p = nullptr; t = now(); CK(hipMalloc(&p, 24ull << 30)); printf("hipMalloc 24 GiB again: %.3f s\n", now() - t); CK(hipFree(p));
	}
	{	// four threads, 10 GiB each
		std::vector<std::thread> th; void* ps[4]; double ts[4];
		t = now();
		for (int i = 0; i < 4; ++i) th.emplace_back([&, i]() { CK(hipSetDevice(0)); double t0 = now(); CK(hipMalloc(&ps[i], 10ull << 30)); ts[i] = now() - t0; });
		for (auto& x : th) x.join();
		printf("4 threads x hipMalloc 10 GiB: wall %.3f s (each %.3f %.3f %.3f %.3f)\n", now() - t, ts[0], ts[1], ts[2], ts[3]);
		for (int i = 0; i < 4; ++i) CK(hipFree(ps[i]));
	}
	{	// stream-ordered pool
		hipStream_t s; CK(hipStreamCreate(&s)); void* p = nullptr;
		t = now(); CK(hipMallocAsync(&p, 10ull << 30, s)); CK(hipStreamSynchronize(s)); printf("hipMallocAsync 10 GiB: %.3f s", now() - t);
		t = now(); CK(hipFreeAsync(p, s)); CK(hipStreamSynchronize(s)); printf("  free %.3f s", now() - t);
		t = now(); CK(hipMallocAsync(&p, 10ull << 30, s)); CK(hipStreamSynchronize(s)); printf("  again %.3f s\n", now() - t);
		CK(hipFreeAsync(p, s)); CK(hipStreamSynchronize(s));
	}
	const size_t n = 1536ull << 20;
	void* d = nullptr; CK(hipMalloc(&d, n));
	{	// pageable: fresh (untouched) and touched
		char* h = (char*)aligned_alloc(2 << 20, n);
		t = now(); memset(h, 1, n); printf("first touch of 1.5 GiB pageable: %.3f s\n", now() - t);
		for (int k = 0; k < 3; ++k) { t = now(); CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); double dt = now() - t; printf("H2D pageable 1.5 GiB: %.3f s = %.1f GB/s\n", dt, n / dt / 1e9); }
		for (int k = 0; k < 2; ++k) { t = now(); CK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); double dt = now() - t; printf("D2H pageable 1.5 GiB: %.3f s = %.1f GB/s\n", dt, n / dt / 1e9); }
		t = now(); CK(hipHostRegister(h, n, hipHostRegisterDefault)); printf("hipHostRegister 1.5 GiB: %.3f s\n", now() - t);
		for (int k = 0; k < 2; ++k) { t = now(); CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); double dt = now() - t; printf("H2D registered 1.5 GiB: %.3f s = %.1f GB/s\n", dt, n / dt / 1e9); }
		t = now(); CK(hipHostUnregister(h)); printf("hipHostUnregister: %.3f s\n", now() - t);
		free(h);
	}
	{
		void* h = nullptr; t = now(); CK(hipHostMalloc(&h, n, hipHostMallocPortable)); printf("hipHostMalloc 1.5 GiB: %.3f s\n", now() - t);
		for (int k = 0; k < 2; ++k) { t = now(); CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); double dt = now() - t; printf("H2D page-locked 1.5 GiB: %.3f s = %.1f GB/s\n", dt, n / dt / 1e9); }
		for (int k = 0; k < 2; ++k) { t = now(); CK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); double dt = now() - t; printf("D2H page-locked 1.5 GiB: %.3f s = %.1f GB/s\n", dt, n / dt / 1e9); }
		t = now(); CK(hipHostFree(h)); printf("hipHostFree: %.3f s\n", now() - t);
	}
	{	// a file in tmpfs mapped and registered: can the device copy land in the page cache directly?
		const char* path = "/dev/shm/alloc_probe.bin"; unlink(path);
		int fd = open(path, O_RDWR | O_CREAT, 0600); if (ftruncate(fd, n)) return 1;
		char* m = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		t = now(); hipError_t e = hipHostRegister(m, n, hipHostRegisterDefault); printf("hipHostRegister of a tmpfs mapping (untouched): %s, %.3f s\n", hipGetErrorString(e), now() - t);
		if (e == hipSuccess)
		{
			for (int k = 0; k < 2; ++k) { t = now(); CK(hipMemcpy(m, d, n, hipMemcpyDeviceToHost)); double dt = now() - t; printf("D2H into the registered mapping: %.3f s = %.1f GB/s\n", dt, n / dt / 1e9); }
			t = now(); CK(hipHostUnregister(m)); printf("unregister: %.3f s\n", now() - t);
		}
		else (void)hipGetLastError();
		munmap(m, n); close(fd); unlink(path);
	}
	return 0;
}
