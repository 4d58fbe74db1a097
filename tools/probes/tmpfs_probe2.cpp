// After fallocate (pages exist): how fast do T threads fill one tmpfs file through pwrite and through a mapping, and what do
// munmap / exit cost?   g++ -O2 -std=c++17 -pthread -o tmpfs_probe2 tmpfs_probe2.cpp ; ./tmpfs_probe2 <GB> <threads...>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const char* PATH = "/dev/shm/tmpfs_probe2.bin";
template <typename F> static void par(unsigned T, F f) { std::vector<std::thread> th; for (unsigned t = 0; t < T; ++t) th.emplace_back(f, t); for (auto& x : th) x.join(); }
int main(int argc, char** argv)
{
	const size_t total = (size_t)(atof(argc > 1 ? argv[1] : "8") * (1ull << 30));
	const size_t piece = 32ull << 20;
	std::vector<unsigned> Ts; for (int i = 2; i < argc; ++i) Ts.push_back(atoi(argv[i])); if (Ts.empty()) Ts = {1, 4, 16};
	char* src = (char*)aligned_alloc(4096, piece); memset(src, 'A', piece);
	for (unsigned T : Ts)
	{
		{
			unlink(PATH); int fd = open(PATH, O_RDWR | O_CREAT, 0600);
			double t = now(); if (fallocate(fd, 0, 0, total)) abort(); double fa = now() - t;
			std::atomic<size_t> next(0);
			t = now();
			par(T, [&](unsigned) { for (size_t o; (o = next.fetch_add(piece)) < total;) { size_t n = std::min(piece, total - o); if (pwrite(fd, src, n, o) != (ssize_t)n) abort(); } });
			double dt = now() - t; printf("T=%2u fallocate %.3f s, pwrite into existing pages: %.3f s = %.2f GB/s\n", T, fa, dt, total / dt / 1e9);
			close(fd); unlink(PATH);
		}
		{
			unlink(PATH); int fd = open(PATH, O_RDWR | O_CREAT, 0600);
			if (fallocate(fd, 0, 0, total)) abort();
			char* m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
			std::atomic<size_t> next(0);
			double t = now();
			par(T, [&](unsigned) { for (size_t o; (o = next.fetch_add(piece)) < total;) memcpy(m + o, src, std::min(piece, total - o)); });
			double dt = now() - t; printf("T=%2u after fallocate: mmap + memcpy (PTE faults only): %.3f s = %.2f GB/s", T, dt, total / dt / 1e9);
			t = now(); munmap(m, total); printf("   munmap %.3f s", now() - t);
			t = now(); close(fd); unlink(PATH); printf("   unlink %.3f s\n", now() - t);
		}
	}
	{	// fallocate in 256 MiB pieces (what a side thread would do) and the cost of leaving with the mapping in place
		unlink(PATH); int fd = open(PATH, O_RDWR | O_CREAT, 0600);
		double t = now();
		for (size_t o = 0; o < total; o += 256ull << 20) if (fallocate(fd, 0, o, std::min<size_t>(256ull << 20, total - o))) abort();
		printf("fallocate in 256 MiB pieces: %.3f s = %.2f GB/s\n", now() - t, total / (now() - t) / 1e9);
		t = now();
		pid_t c = fork();
		if (c == 0)
		{
			char* m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
			for (size_t o = 0; o < total; o += 4096) m[o] = 1;
			_exit(0);
		}
		int st; waitpid(c, &st, 0);
		const double whole = now() - t;
		t = now();
		c = fork();
		if (c == 0)
		{
			char* m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
			for (size_t o = 0; o < total; o += 4096) m[o] = 1;
			double t1 = now(); munmap(m, total); fprintf(stderr, "  (child: munmap %.3f s)\n", now() - t1);
			_exit(0);
		}
		waitpid(c, &st, 0);
		printf("child touches every page and _exit()s: %.3f s; with munmap first: %.3f s\n", whole, now() - t);
		close(fd); unlink(PATH);
	}
	return 0;
}
