// How fast can one process put N GB into a fresh file in tmpfs?  (dsrc-amd d writes 37.7 GB of text; round 4: 2.1 GB/s end to end)
// variants: fallocate alone; fallocate + mmap + MADV_POPULATE_WRITE by T threads; mmap + memcpy by T threads (no fallocate);
// pwrite by T threads (no fallocate); pwrite after fallocate.
// g++ -O2 -std=c++17 -pthread -o tmpfs_probe tmpfs_probe.cpp ; ./tmpfs_probe <GB> <threads...>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const char* PATH = "/dev/shm/tmpfs_probe.bin";
template <typename F> static void par(unsigned T, F f) { std::vector<std::thread> th; for (unsigned t = 0; t < T; ++t) th.emplace_back(f, t); for (auto& x : th) x.join(); }
int main(int argc, char** argv)
{
	const size_t total = (size_t)(atof(argc > 1 ? argv[1] : "8") * (1ull << 30));
	const size_t piece = 32ull << 20;
	std::vector<unsigned> Ts; for (int i = 2; i < argc; ++i) Ts.push_back(atoi(argv[i])); if (Ts.empty()) Ts = {1, 4, 16};
	char* src = (char*)aligned_alloc(4096, piece); memset(src, 'A', piece);
	{	// fallocate alone
		unlink(PATH); int fd = open(PATH, O_RDWR | O_CREAT, 0600);
		double t = now(); int rc = fallocate(fd, 0, 0, total); double dt = now() - t;
		printf("fallocate(%.1f GB) rc=%d: %.3f s = %.2f GB/s\n", total / 1e9, rc, dt, total / dt / 1e9);
		t = now(); close(fd); unlink(PATH); printf("  unlink: %.3f s\n", now() - t);
	}
	for (unsigned T : Ts)
	{
		{	// pwrite, no fallocate
			unlink(PATH); int fd = open(PATH, O_RDWR | O_CREAT, 0600);
			std::atomic<size_t> next(0);
			double t = now();
			par(T, [&](unsigned) { for (size_t o; (o = next.fetch_add(piece)) < total;) { size_t n = std::min(piece, total - o); if (pwrite(fd, src, n, o) != (ssize_t)n) abort(); } });
			double dt = now() - t; printf("T=%2u pwrite fresh          : %.3f s = %.2f GB/s\n", T, dt, total / dt / 1e9);
			close(fd); unlink(PATH);
		}
		{	// ftruncate + mmap + memcpy (faults inside the copy)
			unlink(PATH); int fd = open(PATH, O_RDWR | O_CREAT, 0600);
			double t = now(); if (ftruncate(fd, total)) abort();
			char* m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
			std::atomic<size_t> next(0);
			par(T, [&](unsigned) { for (size_t o; (o = next.fetch_add(piece)) < total;) memcpy(m + o, src, std::min(piece, total - o)); });
			double dt = now() - t; printf("T=%2u ftruncate+mmap+memcpy : %.3f s = %.2f GB/s\n", T, dt, total / dt / 1e9);
			t = now(); munmap(m, total); close(fd); unlink(PATH); printf("  munmap+unlink: %.3f s\n", now() - t);
		}
		{	// ftruncate + mmap + MADV_POPULATE_WRITE by T threads, then memcpy by T threads
			unlink(PATH); int fd = open(PATH, O_RDWR | O_CREAT, 0600);
			double t = now(); if (ftruncate(fd, total)) abort();
			char* m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
			std::atomic<size_t> next(0);
			par(T, [&](unsigned) { for (size_t o; (o = next.fetch_add(piece)) < total;) if (madvise(m + o, std::min(piece, total - o), 23 /*MADV_POPULATE_WRITE*/)) { perror("madvise"); abort(); } });
			double dt = now() - t; printf("T=%2u ftruncate+mmap+populate: %.3f s = %.2f GB/s", T, dt, total / dt / 1e9);
			next = 0; t = now();
			par(T, [&](unsigned) { for (size_t o; (o = next.fetch_add(piece)) < total;) memcpy(m + o, src, std::min(piece, total - o)); });
			dt = now() - t; printf("   then memcpy %.3f s = %.2f GB/s\n", dt, total / dt / 1e9);
			munmap(m, total); close(fd); unlink(PATH);
		}
	}
	{	// T files instead of one (is the limit per inode?)
		const unsigned T = Ts.back();
		std::vector<int> fds(T); char nm[64];
		for (unsigned t = 0; t < T; ++t) { snprintf(nm, sizeof nm, "%s.%u", PATH, t); unlink(nm); fds[t] = open(nm, O_RDWR | O_CREAT, 0600); }
		double t0 = now();
		par(T, [&](unsigned t) { const size_t mine = total / T; for (size_t o = 0; o < mine; o += piece) if (pwrite(fds[t], src, std::min(piece, mine - o), o) <= 0) abort(); });
		double dt = now() - t0; printf("T=%2u pwrite, one file per thread: %.3f s = %.2f GB/s\n", T, dt, total / dt / 1e9);
		for (unsigned t = 0; t < T; ++t) { close(fds[t]); snprintf(nm, sizeof nm, "%s.%u", PATH, t); unlink(nm); }
	}
	return 0;
}
