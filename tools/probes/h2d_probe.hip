// Host -> device copies straight out of a file mapping (tmpfs): can the reader threads and their buffers go?
// variants: MAP_SHARED / MAP_PRIVATE read-only mapping; one copy of 1.5 GiB or 192 copies of 8 MiB; two threads at once.
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
	const char* path = "/dev/shm/h2d_probe.bin";
	const size_t piece = 1536ull << 20, n_pieces = 6, total = piece * n_pieces;
	{
		unlink(path); int fd = open(path, O_RDWR | O_CREAT, 0600);
		std::vector<char> buf(64 << 20, 'x');
		for (size_t o = 0; o < total; o += buf.size()) if (pwrite(fd, buf.data(), buf.size(), o) != (ssize_t)buf.size()) return 1;
		close(fd);
	}
	CK(hipFree(nullptr));
	void* d[2]; CK(hipMalloc(&d[0], piece)); CK(hipMalloc(&d[1], piece));
	hipStream_t st[2]; CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
	for (int shared = 0; shared < 2; ++shared)
	{
		int fd = open(path, O_RDONLY);
		char* m = (char*)mmap(nullptr, total, PROT_READ, shared ? MAP_SHARED : MAP_PRIVATE, fd, 0);
		if (m == MAP_FAILED) { perror("mmap"); return 1; }
		double t = now(); CK(hipMemcpy(d[0], m, piece, hipMemcpyHostToDevice)); double dt = now() - t;
		printf("%s read-only mapping: one copy of 1.5 GiB: %.3f s = %.1f GB/s\n", shared ? "MAP_SHARED" : "MAP_PRIVATE", dt, piece / dt / 1e9);
		t = now();
		for (size_t o = 0; o < piece; o += 8 << 20) CK(hipMemcpyAsync((char*)d[0] + o, m + piece + o, 8 << 20, hipMemcpyHostToDevice, st[0]));
		CK(hipStreamSynchronize(st[0])); dt = now() - t;
		printf("   192 async copies of 8 MiB: %.3f s = %.1f GB/s\n", dt, piece / dt / 1e9);
		t = now();
		std::thread a([&]() { CK(hipMemcpyAsync(d[0], m + 2 * piece, piece, hipMemcpyHostToDevice, st[0])); CK(hipStreamSynchronize(st[0])); });
		std::thread b([&]() { CK(hipMemcpyAsync(d[1], m + 3 * piece, piece, hipMemcpyHostToDevice, st[1])); CK(hipStreamSynchronize(st[1])); });
		a.join(); b.join(); dt = now() - t;
		printf("   two threads, 1.5 GiB each: %.3f s = %.1f GB/s together\n", dt, 2 * piece / dt / 1e9);
		t = now(); CK(hipMemcpy(d[0], m, piece, hipMemcpyHostToDevice)); dt = now() - t;
		printf("   the first piece again: %.3f s = %.1f GB/s\n", dt, piece / dt / 1e9);
		t = now(); munmap(m, total); close(fd); printf("   munmap %.3f s\n", now() - t);
	}
	{	// for comparison: pread into a fresh buffer by 8 threads, then the copy
		int fd = open(path, O_RDONLY);
		char* h = (char*)aligned_alloc(2 << 20, piece);
		double t = now();
		std::vector<std::thread> th;
		for (int i = 0; i < 8; ++i) th.emplace_back([&, i]() { const size_t part = piece / 8; size_t got = 0; while (got < part) { ssize_t r = pread(fd, h + i * part + got, part - got, 4 * piece + i * part + got); if (r <= 0) abort(); got += r; } });
		for (auto& x : th) x.join();
		double dt = now() - t; printf("pread of 1.5 GiB into a fresh buffer by 8 threads: %.3f s = %.1f GB/s", dt, piece / dt / 1e9);
		t = now(); CK(hipMemcpy(d[0], h, piece, hipMemcpyHostToDevice)); dt = now() - t; printf(";  then H2D %.3f s = %.1f GB/s\n", dt, piece / dt / 1e9);
		t = now(); free(h); printf("   free %.3f s\n", now() - t);
		close(fd);
	}
	unlink(path);
	return 0;
}
