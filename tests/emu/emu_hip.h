// TEST INFRASTRUCTURE ONLY -- a tiny single-threaded emulator of the HIP subset
// used by dsrc_amd/csrc, so that the kernel *logic* (indexing, scans, ballots,
// LDS protocols, bit packing) can be exercised by the CPU test-suite in a
// container that has no GPU.  One workgroup runs at a time; its threads are
// coroutines (a register-only context switch: no signal-mask system calls) that yield at __syncthreads() and at wave-level
// exchanges (__ballot/__shfl*), which complete when every live lane of the
// 64-wide wave has arrived (kernels only use them in wave-uniform control flow).
//
// The product never includes this file: dsrc_amd/csrc includes
// <hip/hip_runtime.h>, and only tests/emu/Makefile puts tests/emu/include first
// on the include path.  It is not a fallback and is not shipped in libdsrc_gpu.so.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <algorithm>

#define DSRC_EMU_BUILD 1

struct dim3
{
	unsigned x, y, z;
	dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu
{
struct Ctx { void* sp; };       // a suspended coroutine: its stack pointer (the callee-saved registers are on its stack)
struct Thread
{
	Ctx ctx;
	std::vector<char> stack;
	int state;               // 0 run, 1 at barrier, 2 at wave op, 3 done
	uint64_t deposit;
	uint64_t gathered[64];
	uint64_t active;
};
extern dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
extern Thread* g_cur;
extern Ctx g_sched;
void yield_to_scheduler();
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
uint64_t wave_exchange(uint64_t v, uint64_t out[64]);   // returns active mask
} // namespace emu

#define threadIdx (emu::g_threadIdx)
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

static const int warpSize = 64;

struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r; r.x = x; r.y = y; return r; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

inline void __syncthreads() { emu::g_cur->state = 1; emu::yield_to_scheduler(); }
inline void __threadfence() {}
inline void __threadfence_block() {}

inline unsigned long long __ballot(int pred)
{
	uint64_t g[64];
	uint64_t act = emu::wave_exchange(pred ? 1 : 0, g);
	unsigned long long m = 0;
	for (int i = 0; i < 64; ++i)
		if (((act >> i) & 1) && g[i]) m |= 1ull << i;
	return m;
}
inline int __any(int p) { return __ballot(p) != 0; }
inline int __all(int p) { uint64_t g[64]; uint64_t act = emu::wave_exchange(p ? 1 : 0, g); for (int i = 0; i < 64; ++i) if (((act >> i) & 1) && !g[i]) return 0; return 1; }

template <typename T> inline T emu_shfl_from(T v, int src)
{
	static_assert(sizeof(T) <= 8, "shfl of <= 8 byte types only");
	uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
	uint64_t g[64];
	uint64_t act = emu::wave_exchange(raw, g);
	int lane = (int)(threadIdx.x & 63);
	if (src < 0 || src > 63 || !((act >> src) & 1)) src = lane;
	T r; memcpy(&r, &g[src], sizeof(T));
	return r;
}
template <typename T> inline T __shfl(T v, int src, int width = 64) { (void)width; return emu_shfl_from(v, src & 63); }
template <typename T> inline T __shfl_up(T v, unsigned d, int width = 64) { (void)width; int l = (int)(threadIdx.x & 63); return emu_shfl_from(v, l - (int)d >= 0 ? l - (int)d : l); }
template <typename T> inline T __shfl_down(T v, unsigned d, int width = 64) { (void)width; int l = (int)(threadIdx.x & 63); return emu_shfl_from(v, l + (int)d <= 63 ? l + (int)d : l); }
template <typename T> inline T __shfl_xor(T v, int m, int width = 64) { (void)width; int l = (int)(threadIdx.x & 63); return emu_shfl_from(v, l ^ m); }

// gfx950 intrinsics used by k_common.h / k_rc.h
#define __builtin_amdgcn_readlane(v, l) emu_shfl_from((int)(v), (int)(l))
#define __builtin_amdgcn_readfirstlane(v) emu_shfl_from((int)(v), 0)
#define __builtin_amdgcn_writelane(v, l, old) ((int)(threadIdx.x & 63) == (int)(l) ? (int)(v) : (int)(old))
#define __builtin_assume(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_alignbit(hi, lo, sh) ((unsigned)(((((uint64_t)(unsigned)(hi)) << 32) | (unsigned)(lo)) >> ((sh) & 31)))
#define __builtin_amdgcn_wave_barrier() do { uint64_t g_[64]; emu::wave_exchange(0, g_); } while (0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_global_load_lds(g, l, sz, off, aux) memcpy((char*)(void*)(l) + (threadIdx.x & 63) * ((sz) == 12 ? 16 : (sz)), (const char*)(const void*)(g) + (off), (sz))   /* measured on gfx950: the 12-byte form also advances 16 bytes per lane */

inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
inline unsigned __brev(unsigned x) { unsigned r = 0; for (int i = 0; i < 32; ++i) if (x & (1u << i)) r |= 1u << (31 - i); return r; }

template <typename T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) ((void)(*(p) = (v)))

// ---- host runtime subset ------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost, hipMemcpyDefault };
struct hipDeviceProp_t { char name[64]; size_t totalGlobalMem; int multiProcessorCount; char gcnArchName[64]; };

inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
template <typename T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return hipSuccess; }
template <typename T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc((void**)p, n, f); }
enum { hipHostMallocDefault = 0, hipHostMallocPortable = 1 };
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = 0) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (void*)1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)1; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (void*)2; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, unsigned, const uint32_t*) { *s = (void*)3; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)8 << 30; *t = (size_t)8 << 30; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof(*p)); strcpy(p->name, "emu"); strcpy(p->gcnArchName, "emu"); p->totalGlobalMem = (size_t)8 << 30; p->multiProcessorCount = 1; return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu error"; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (void*)1; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = 0) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
	emu::launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })
