// libdsrc_gpu.so -- the GPU block scheduler behind include/dsrc_gpu.h.
//
// It replaces the reference's pool of DsrcCompressor worker threads (src/DsrcWorker.cpp:30-73):
// a *batch* of FASTQ chunks is pushed through a fixed sequence of kernels, each owning whole
// blocks, with four small device->host readbacks where the reference takes data-dependent
// decisions (record counts, stream schemes, tag field kinds, final sizes).  All per-symbol and
// per-record work is in the kernels (k_*.h); the host only sizes buffers and picks scheme ids
// from block statistics exactly as the reference's *ModelerProxy::SelectSchemeId do.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <dlfcn.h>
#include <mutex>
#ifndef DSRC_REPLAY_WHATIF
#define DSRC_REPLAY_WHATIF 0     // experiments only: a k_replay PROBE mode for the whole run (wrong output)
#endif
#include <string>
#include <thread>
#include <vector>

#include "../../include/dsrc_gpu.h"
#include "dsrc_types.h"
#include "k_common.h"
#ifndef RCS_WIDE
#define RCS_WIDE 32                    // streams per k_rcs workgroup in launches of more than 640 streams (16 below)
#endif
#include "k_parse.h"
#include "k_rc.h"
#include "k_bucket.h"
#include "k_huff.h"
#include "k_tags.h"
#include "k_block.h"
#include "k_synth.h"
#include "k_dec.h"
#include "k_dec_rc.h"
#include "k_dec_tags.h"
#include "k_dec_q0.h"

namespace
{

// Environment switches.  The product reads four, all documented in INTEGRATION.md section 4: DSRC_GPU_DEBUG (1: arena use per batch,
// 2: + the host-side stage timeline), DSRC_GPU_TRACE (roctx ranges around the batch stages), DSRC_GPU_QUEUE_LANES (1: one scheduler
// lane per handle in the queue form), DSRC_GPU_DEC_TABLE_MB (HBM for the decoder's model tables) -- and GPU_MAX_HW_QUEUES in
// dsrcgpu_prepare.  Everything else is a TEST HOOK that forces a path the library otherwise chooses by itself (the ballot ranking,
// the reference loop of the range coder, the sort-and-replay front end, the one-lane decoders ...): compiled in only with
// -DDSRC_GPU_TEST_HOOKS, i.e. in tests/emu/libdsrc_emu.so and in dsrc_amd/csrc/libdsrc_gpu_hooks.so, which only tests/ load.
#ifdef DSRC_GPU_TEST_HOOKS
const char* hook_env(const char* name) { return getenv(name); }
#else
const char* hook_env(const char*) { return nullptr; }
#endif
long hook_int(const char* name, long dflt) { const char* v = hook_env(name); return v ? atol(v) : dflt; }

// DSRC_GPU_TRACE=1: roctx ranges around the stages of a batch / a decoding pass, for rocprofv3 --marker-trace timelines.
// The marker library is looked up when the switch is set and only then (rocprofiler-sdk's librocprofiler-sdk-roctx.so, else the older
// libroctx64.so); without it, or without the switch, a range costs one predictable branch.
struct Roctx
{
	int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
	Roctx()
	{
#ifndef DSRC_EMU_BUILD
		if (!getenv("DSRC_GPU_TRACE")) return;
		void* lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
		if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
		if (!lib) return;
		push = (int (*)(const char*))dlsym(lib, "roctxRangePushA"); pop = (int (*)())dlsym(lib, "roctxRangePop");
		if (!push || !pop) { push = nullptr; pop = nullptr; }
#endif
	}
};
const Roctx& roctx() { static const Roctx r; return r; }
struct TraceRange          // one stage; the previous one ends where the next begins
{
	bool open = false;
	void stage(const char* name) { const Roctx& r = roctx(); if (!r.push) return; if (open) r.pop(); r.push(name); open = true; }
	~TraceRange() { if (open) roctx().pop(); }
};

struct Arena
{
	u8* base = nullptr;
	size_t cap = 0, top = 0;
	bool failed = false;
	size_t alloc(size_t bytes, size_t align = 256)
	{
		top = (top + align - 1) / align * align;
		const size_t off = top;
		top += bytes;
		if (top > cap) failed = true;
		return off;
	}
};

// Queue form (dsrcgpu_submit / flush / collect / release): a ring of batches in page-locked memory.  submit() copies a
// chunk into the batch being filled, flush() hands that batch to the handle's scheduler thread and returns, collect()
// hands out the blocks of finished batches in submission order as pointers INTO the batch's output buffer, release()
// gives them back; a batch buffer is reused once all its blocks have been released.
struct QBatch
{
	enum State { Free, Filling, Queued, Done } state = Free;
	std::vector<int64_t> ids; std::vector<u64> in_off, sizes;
	std::vector<const u8*> ext;      // per chunk: the caller's page-locked memory (dsrcgpu_submit_pinned), or null = copied into `in`
	u8* in = nullptr; u64 in_cap = 0, in_used = 0;
	u8* out = nullptr; u64 out_cap = 0;
	std::vector<u64> o_offs, o_sizes, raw, comp;
	u32 next_collect = 0, outstanding = 0;
	int rc = 0; std::string err;
	uint64_t seq = 0;                // flush order: the batch's turn in the handle's internal chain (two lanes)
	std::vector<u32> layout;         // dsrcgpu_set_record_layout at the time of the flush: it belongs to this batch, whichever lane runs it
};
#define DSRC_QUEUE_DEPTH 6                   // batches of the ring: one being filled + up to DSRC_QUEUE_LANES_MAX running / waiting
#define DSRC_QUEUE_LANES_MAX 4
#define DSRC_QUEUE_LANES_DEFAULT 3
// HBM of the element slice: the streams of a batch go through k_part / k_model / k_place (or k_sort / k_replay) this many MiB of
// elements at a time.  A stream handed back to k_sort needs two 8-byte buffers (16 B per symbol), so 1792 MiB are ~32 streams of an
// 8 MiB chunk per launch group (rounds 1-4: 7 GiB = 128 streams, sized when every stream took that road; the bucketed path writes
// 4-byte elements into the same place).  Launches of 32 streams still fill the GPU: k_part 13 k workgroups, k_model 8-16 k.
#ifndef DSRC_SORT_SLICE_MB
#define DSRC_SORT_SLICE_MB 1792
#endif
// ... unless the batch is large anyway: with >= 2.5 GB of chunks (300 of 8 MiB) the slice is the 7 GiB of rounds 1-4 -- launch groups of
// 128 streams are 3 % faster than groups of 32 (4 x 450 blocks: 54.1 against 52.3 GB/s, profiles/r05_sweep_slice.txt), and 7 GiB are a
// sixth of such a batch's arena, not a third of it as for the 192-chunk batches of the command line
static size_t slice_budget(size_t in_total) { return (size_t)(in_total >= ((size_t)2560 << 20) ? 7168 : DSRC_SORT_SLICE_MB) << 20; }

} // namespace

// block-to-block state handed from batch to batch when several handles work on one archive (include/dsrc_gpu.h)
struct dsrcgpu_chain
{
	std::mutex m; std::condition_variable cv;
	uint64_t next_seq = 0;           // the batch whose turn it is to read `fields_cap`
	u32 fields_cap = 0;
	bool failed = false;
};

static thread_local int tl_queue_lane = 0;           // this thread is a scheduler lane of the queue form running a batch: 1 the handle's only lane, 2 one of two

struct dsrcgpu_handle
{
	dsrcgpu_settings set;
	dsrcgpu_dataset ds;
	int device = 0;
	hipStream_t stream = nullptr;
	hipStream_t rc_stream = nullptr;    // k_rc only (and the DNA chains of a verifying pass)
	Arena arena;
	u64 arena_fixed = 0;
	u32 fields_cap = 0;              // capacity of the reference's TagStats::fields vector, carried block to block
	std::vector<u32> rec_pending;    // dsrcgpu_set_record_layout: taken (under q_m) by the next batch call or the next flush, whichever comes first;
	                                 // from there it travels with that batch as a per-call argument -- no scheduler thread ever touches this field
	dsrcgpu_chain* chain = nullptr;  // if set: fields_cap comes from / goes to the chain, in batch order
	uint64_t chain_seq = 0; bool chain_taken = false; u32 chain_cap_in = 0;
	bool chain_batch_done = true;    // the batch announced by dsrcgpu_set_chain has run to completion
	u32* d_crc_tab = nullptr;
	std::string err; mutable std::mutex err_m;   // written by whichever thread fails (caller, scheduler thread, collector): guarded
	u8* last_d_out = nullptr;        // device address of the blocks the last run_batch assembled (valid until the arena is reused)
	QBatch qb[DSRC_QUEUE_DEPTH]; u32 q_fill = 0, q_collect = 0;     // ring: batches are filled, run and collected in this order
	u32 q_depth = 3;                                                 // slots of the ring in use: the lanes + 2 (decided at the first flush; 3 before)
	u32 q_pending = 0;                                               // flushed batches that still have blocks to hand out
	std::deque<u32> q_run;
	std::mutex q_m; std::condition_variable q_cv;
	std::thread q_thread; bool q_stop = false, q_started = false;
	// Queue form, second lane: consecutive batches run on two scheduler lanes (this handle and `twin`, a handle of the same settings
	// with its own arena and streams), so that the range coder of batch i -- 130 ms on a handful of CUs -- overlaps the copies and the
	// front end of batch i + 1.  What DSRC carries from block to block goes from lane to lane through `q_chain`, in flush order.
	std::vector<dsrcgpu_handle*> twins; dsrcgpu_chain* q_chain = nullptr; std::vector<std::thread> q_threads2; uint64_t q_seq = 0; bool q_lanes_decided = false;
	bool q_user_batch = false;       // (q_m) a batch call by the user has advanced h->fields_cap since the last flush: the next flush hands it to the lanes' chain
	int q_rc = 0; std::string q_err;                                 // first failure of the scheduler thread (sticky)
	float batch_ms = 0.f, rc_ms = 0.f, verify_ms = 0.f;
	u32 rc_launches = 0;
	hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
	// per-stage HIP-event timing of the last batch: pairs of events around every k_sort launch and every replay group
	std::vector<hipEvent_t> stage_ev; std::vector<u32> stage_kind; u32 stage_used = 0;       // kind: 0 sort, 1 replay
	float sort_ms = 0.f, replay_ms = 0.f, decode_stream_ms = 0.f;
	bool sort_atomic = false;        // k_sort ranks with LDS atomics (device passed k_lds_order_test), else with ballots
	bool lds64_ordered = false;      // ... and k_lds_order_test64: the bucketed path (k_part / k_model) may run
	u64 dec_table_budget = 0;        // dsrcgpu_set_table_budget: HBM a decoding pass may take for model tables (0 = automatic)
	u32* dec_tables = nullptr; u64 dec_tables_cap = 0;     // model tables of the range-decoded levels (bytes), kept between passes
	std::vector<DecHint> verify_hints;                     // run_batch -> verify_blocks: where the DNA stream of every block it wrote lies
	// Scheduler lanes inside the handle (round 6): a batch call is cut into sub-batches that run on `lanes` child handles (own arena,
	// own streams, one host thread each per call), the block-to-block state going from sub-batch to sub-batch through `sub_chain`:
	// the range coder of one sub-batch overlaps the front ends of the others without the caller holding several handles
	std::vector<dsrcgpu_handle*> subs; dsrcgpu_chain* sub_chain = nullptr;
	u32 lanes_want = 0, sub_chunks_want = 0;      // dsrcgpu_set_lanes: 0 = the defaults
	bool is_sub = false;
	u32 rc_redone = 0;               // streams coded a second time by k_rc (carry clamp under k_rcs, device hand-backs) since the handle was created
	bool rc_caps_worst = false;      // a range-coded stream has outgrown the estimate of its staging once: two bytes per symbol from now on (run_batch)
};

namespace
{

int fail(dsrcgpu_handle* h, int code, const char* fmt, ...)
{
	char buf[512];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
	if (h) { std::lock_guard<std::mutex> g(h->err_m); h->err = buf; }
	return code;
}

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(h, DSRCGPU_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)
#define KCHK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(h, DSRCGPU_E_HIP, "kernel launch failed (%s:%d): %s", __FILE__, __LINE__, hipGetErrorString(e_)); } while (0)

int ensure_arena(dsrcgpu_handle* h, size_t need)
{
	// debugging aid: DSRC_GPU_DEBUG_FILL=<byte> fills the arena before every batch -- an output that changes with the byte
	// means some kernel reads arena bytes nobody wrote
	const char* fill = hook_env("DSRC_GPU_DEBUG_FILL");
	if (h->arena.cap >= need) { h->arena.top = 0; h->arena.failed = false; if (fill) HIPCHK(hipMemsetAsync(h->arena.base, atoi(fill), h->arena.cap, h->stream)); return 0; }
	if (h->arena_fixed && need > h->arena_fixed)
		return fail(h, DSRCGPU_E_NOMEM, "batch needs %zu bytes of HBM scratch, arena is fixed at %llu", need, (unsigned long long)h->arena_fixed);
	if (h->arena.base) { HIPCHK(hipFree(h->arena.base)); h->arena.base = nullptr; h->arena.cap = 0; }
	size_t want = h->arena_fixed ? (size_t)h->arena_fixed : need + need / 64;
	hipError_t e;
	{
		// One arena at a time, process-wide.  HBM that another process (or an earlier handle) has released is wiped by the driver at
		// ~35 GB/s, and an allocation that lands on memory still waiting for that is held until it is clean
		// (profiles/r05_alloc_probe.txt: 0 ms or seconds, depending on the device's recent past).  Four instances asking at once
		// all came back together after 3.3 s (4 x 24 GB, profiles/r05_e2e_first.txt); asking in turn lets the first one start its
		// batch while the others' arenas are still being cleaned.
		static std::mutex alloc_turn[64];                 // per device: the wipe is the device's, and several GPUs of a node do not wait for each other
		std::lock_guard<std::mutex> g(alloc_turn[h->device & 63]);
		const auto t0 = std::chrono::steady_clock::now();
		e = hipMalloc((void**)&h->arena.base, want);
		if (getenv("DSRC_GPU_DEBUG")) fprintf(stderr, "[dsrc_gpu] %p arena of %.1f GB: hipMalloc took %.0f ms\n", (void*)h, want / 1e9, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
	}
	if (e != hipSuccess) return fail(h, DSRCGPU_E_NOMEM, "hipMalloc(%zu) for the batch arena failed: %s", want, hipGetErrorString(e));
	h->arena.cap = want; h->arena.top = 0; h->arena.failed = false;
	if (fill) HIPCHK(hipMemsetAsync(h->arena.base, atoi(fill), h->arena.cap, h->stream));
	return 0;
}

u32 bitlen(u64 x) { for (u32 i = 0; i < 32; ++i) if (x < (1ull << i)) return i; return 64; }
u32 log2u(u32 x) { u32 r = 0; while (x > 1) { x >>= 1; ++r; } return r; }
u32 h_huff_ws_words(u32 n) { const u32 m = n < 2 ? 2 : n; return 10 * m; }
u32 h_huff_tree_cap(u32 n) { const u32 m = n < 2 ? 2 : n; return (16 + (2 * m + m * 10) / 8 + 8 + 3) & ~3u; }
size_t al(size_t x, size_t a) { return (x + a - 1) / a * a; }

// how much arena a batch may need, from the input sizes alone (checked again while carving)
size_t estimate_arena(const dsrcgpu_handle* h, u32 n, const u64* sizes)
{
	const bool rc = h->set.dna_order > 0 || h->set.quality_order > 0;
	size_t tot = 0, mx = 0;
	for (u32 i = 0; i < n; ++i) { tot += (size_t)sizes[i] + 4096; mx = std::max(mx, (size_t)sizes[i]); }
	const size_t sort_slice = std::min(tot * 14, slice_budget(tot) + mx * 16);       // see slice_lo in run_batch
	// measured at -d3 -q2 (round 5: streams carved from the statistics, one byte and a sixteenth of staging per range-coded symbol):
	// 8.6 x the input + the slice; a batch that needs more -- other data, worst-case staging -- says so and is run again
	const size_t copy = (h->set.tag_preserve_flags || h->ds.color_space || !h->rec_pending.empty()) ? tot + (1u << 20) : 0;      // (the device form's private copy of the text)
	// (round 6: 6-byte records, 7.1 x the input measured)
	return tot * (h->rc_caps_worst ? 18 : 15) / 2 + (rc ? sort_slice : 0) + (size_t)n * (1u << 20) + (16u << 20) + copy;
}
size_t estimate_decode_arena(const dsrcgpu_handle* h, u32 n, const u64* sizes, bool own_text);

struct BatchIO
{
	const u8* d_in; const u64* offs; const u64* sizes; u32 n;
	u8* d_out; u64 out_cap;          // device output (nullptr => arena-allocated, copied to host_out)
	u8* host_out; u64 host_cap;
	u64* out_offs; u64* out_sizes; u64* raw; u64* comp;
	const std::vector<u32>* layout = nullptr;     // dsrcgpu_set_record_layout of this batch (empty / null: chunks cut from a file)
	bool in_is_callers = false;                   // d_in is the caller's device memory (dsrcgpu_compress_batch_device), not a copy in the arena
};

template <typename T> T* AP(dsrcgpu_handle* h, size_t off) { return (T*)(h->arena.base + off); }

int run_batch(dsrcgpu_handle* h, BatchIO io)
{
	const u32 B = io.n;
	if (B == 0) return DSRCGPU_OK;
	// DSRC_GPU_DEBUG=2: host-side timeline of the phases of this batch (ms since the call)
	static const bool trace = getenv("DSRC_GPU_DEBUG") && atoi(getenv("DSRC_GPU_DEBUG")) >= 2;
	const auto t_call = std::chrono::steady_clock::now();
	std::string tl;
	auto mark = [&](const char* what) { if (trace) { char b[64]; snprintf(b, sizeof b, " %s %.1f", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count()); tl += b; } };
	hipStream_t s = h->stream;
	Arena& A = h->arena;
	h->stage_used = 0; h->stage_kind.clear();
	auto stage_mark = [&](u32 kind)      // two calls bracket a stage
	{
		if (h->stage_used == h->stage_ev.size()) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return; h->stage_ev.push_back(e); }
		(void)hipEventRecord(h->stage_ev[h->stage_used++], s); h->stage_kind.push_back(kind);
	};
	const u32 dna_order = h->set.dna_order, qo = h->set.quality_order;
	const bool lossy = h->set.lossy != 0, crc = h->set.calculate_crc32 != 0;

	DsrcParams prm;
	prm.dna_order = dna_order; prm.quality_order = qo; prm.lossy = lossy; prm.crc = crc; prm.tag_flags = (u32)h->set.tag_preserve_flags;
	prm.quality_offset = h->ds.quality_offset; prm.n_blocks = B; prm.max_tiles = 1;
	static const std::vector<u32> no_layout;
	const std::vector<u32>& layout = io.layout ? *io.layout : no_layout;
	prm.record_layout = layout.empty() ? 0u : 1u;
	prm.color_space = h->ds.color_space ? 1u : 0u;
	if (prm.color_space && (prm.record_layout || prm.tag_flags))
		return fail(h, DSRCGPU_E_ARG, "colour space cannot be combined with the field filter or the record layout on the GPU path");
	if (prm.record_layout && (layout.size() != B || prm.tag_flags || crc))
		return fail(h, DSRCGPU_E_ARG, "record layout: one chunkSize per chunk of the batch, no field filter, no CRC (src/DsrcArchive.cpp:33-47)");

	std::vector<BlkDesc> desc(B);
	std::vector<BlkState> st(B);
	memset(desc.data(), 0, sizeof(BlkDesc) * B);
	size_t in_total = 0;
	for (u32 b = 0; b < B; ++b)
	{
		if (io.sizes[b] == 0 || io.sizes[b] >= (1ull << 31)) return fail(h, DSRCGPU_E_ARG, "chunk %u: size %llu out of range", b, (unsigned long long)io.sizes[b]);
		desc[b].in_off = io.offs[b]; desc[b].in_size = (u32)io.sizes[b];
		if (prm.record_layout) desc[b].chunk_size_value = layout[b];
		desc[b].n_tiles = (u32)((io.sizes[b] + DSRC_TILE_BYTES - 1) / DSRC_TILE_BYTES);
		prm.max_tiles = std::max(prm.max_tiles, desc[b].n_tiles);
		in_total += io.sizes[b];
	}
	const u8* d_in = io.d_in;
	HIPCHK(hipEventRecord(h->ev[0], s));
	if (io.in_is_callers && (prm.tag_flags || prm.color_space || prm.record_layout))
	{	// k_tag_filter, k_cs_decode and k_tag_poke rewrite the chunk text in place, as BlockCompressor::Store does with its buffer -- but
		// a batch that is run again (a larger arena, worst-case staging) must find the text as it came, and a caller's device buffer is
		// not ours to destroy: those modes work on a copy (round 5 filtered / decoded the already-rewritten text on the second pass)
		u64 lo = ~0ull, hi = 0;
		for (u32 b = 0; b < B; ++b) { lo = std::min<u64>(lo, io.offs[b]); hi = std::max<u64>(hi, io.offs[b] + io.sizes[b]); }
		const size_t o_copy = A.alloc(hi - lo + 64);
		if (A.failed) return fail(h, DSRCGPU_E_NOMEM, "arena exhausted (copy of the input)");
		HIPCHK(hipMemcpyAsync(AP<u8>(h, o_copy), io.d_in + lo, hi - lo, hipMemcpyDeviceToDevice, s));
		d_in = AP<u8>(h, o_copy) - lo;
	}
	TraceRange tr; tr.stage("dsrc batch: line index");

	// ---- phase 1: line counts ------------------------------------------------------------------
	const size_t o_desc = A.alloc(sizeof(BlkDesc) * B), o_state = A.alloc(sizeof(BlkState) * B);
	const size_t o_tiles = A.alloc((size_t)B * prm.max_tiles * 4);
	if (A.failed) return fail(h, DSRCGPU_E_NOMEM, "arena exhausted (phase 1)");
	BlkDesc* d_desc = AP<BlkDesc>(h, o_desc); BlkState* d_state = AP<BlkState>(h, o_state); u32* d_tiles = AP<u32>(h, o_tiles);
	HIPCHK(hipMemcpyAsync(d_desc, desc.data(), sizeof(BlkDesc) * B, hipMemcpyHostToDevice, s));
	HIPCHK(hipMemsetAsync(d_state, 0, sizeof(BlkState) * B, s));
	hipLaunchKernelGGL(k_init_state, dim3((B + 63) / 64), dim3(64), 0, s, d_state, B); KCHK();
	hipLaunchKernelGGL(k_count_lines, dim3(prm.max_tiles, B), dim3(WG), 0, s, d_in, d_desc, d_state, d_tiles, prm); KCHK();
	hipLaunchKernelGGL(k_scan_tiles, dim3(B), dim3(WG), 0, s, d_desc, d_tiles, prm); KCHK();
	HIPCHK(hipMemcpyAsync(st.data(), d_state, sizeof(BlkState) * B, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	mark("S1"); tr.stage("dsrc batch: records, statistics");

	// ---- phase 2: index, statistics, symbol streams --------------------------------------------------
	u64 lines = 0, recs = 0; u32 max_rec_cap = 1;
	for (u32 b = 0; b < B; ++b)
	{
		const u32 n_lines = st[b].n_term + 1;
		desc[b].line_base = (u32)lines; lines += n_lines + 2;
		desc[b].rec_cap = (n_lines + 3) / 4;
		desc[b].rec_base = (u32)recs; recs += desc[b].rec_cap + 1;
		max_rec_cap = std::max(max_rec_cap, desc[b].rec_cap);
		if (lines >= (1ull << 32) || recs >= (1ull << 32)) return fail(h, DSRCGPU_E_ARG, "batch too large for 32-bit record indices; submit fewer chunks per batch");
	}
	const size_t o_lines = A.alloc(lines * 4);
	RecPools rp;
	rp.title_off = AP<u32>(h, A.alloc(recs * 4)); rp.seq_off = AP<u32>(h, A.alloc(recs * 4)); rp.qual_off = AP<u32>(h, A.alloc(recs * 4));
	rp.title_len = AP<u16>(h, A.alloc(recs * 2)); rp.len = AP<u16>(h, A.alloc(recs * 2));
	rp.kept = AP<u16>(h, A.alloc(recs * 2)); rp.trunc = AP<u16>(h, A.alloc(recs * 2));
	rp.q_off = AP<u32>(h, A.alloc(recs * 4)); rp.d_off = AP<u32>(h, A.alloc(recs * 4));
	// With a field filter k_tag_poke overwrites the first base of a sequence line (the title that swallowed its terminator ends there), and
	// it has to run in front of the readback (record 0's template reads that byte): the symbol streams are then written first, into room
	// for the worst case, as in rounds 1-4.  Otherwise they are carved behind the readback, from the statistics (see there).
	const bool streams_early = prm.tag_flags != 0;
	u8* d_q = nullptr; u8* d_qp = nullptr; u8* d_d = nullptr;
	if (streams_early)
	{
		u64 qbytes = 0;
		for (u32 b = 0; b < B; ++b) { desc[b].q_base = qbytes; desc[b].d_base = qbytes; qbytes += al(desc[b].in_size / 2 + 64, 64); }
		const size_t o_q = A.alloc(qbytes), o_qp = A.alloc(qbytes), o_d = A.alloc(qbytes);
		d_q = AP<u8>(h, o_q); d_qp = AP<u8>(h, o_qp); d_d = AP<u8>(h, o_d);
	}
	if (A.failed) return fail(h, DSRCGPU_E_NOMEM, "arena exhausted (phase 2): need > %zu bytes", A.top);
	u32* d_lines = AP<u32>(h, o_lines);
	HIPCHK(hipMemcpyAsync(d_desc, desc.data(), sizeof(BlkDesc) * B, hipMemcpyHostToDevice, s));
	hipLaunchKernelGGL(k_index_lines, dim3(prm.max_tiles, B), dim3(WG), 0, s, d_in, d_desc, d_tiles, d_lines, prm); KCHK();
	hipLaunchKernelGGL(k_records, dim3((max_rec_cap + WG - 1) / WG, B), dim3(WG), 0, s, d_in, d_desc, d_state, d_lines, rp); KCHK();
	if (prm.tag_flags) { hipLaunchKernelGGL(k_tag_filter, dim3((max_rec_cap + WG - 1) / WG, B), dim3(WG), 0, s, const_cast<u8*>(d_in), d_desc, d_state, rp, prm); KCHK(); }
	const u32 rec_gx = std::max(1u, std::min(64u, (max_rec_cap + 4 * WAVES - 1) / (4 * WAVES)));
	if (prm.color_space)
	{	// checksums are taken over the text as it came; then colours -> bases in place
		hipLaunchKernelGGL(k_rec_count, dim3((B + 63) / 64), dim3(64), 0, s, d_desc, d_state, B); KCHK();
		if (crc) { hipLaunchKernelGGL(k_crc, dim3(B, 3), dim3(WG), 0, s, d_in, d_desc, d_state, rp, h->d_crc_tab, prm); KCHK(); }
		hipLaunchKernelGGL(k_cs_decode, dim3(rec_gx, B), dim3(WG), 0, s, const_cast<u8*>(d_in), d_desc, d_state, rp); KCHK();
	}
	// a batch of few, large blocks (-b64, -b256): several workgroups per block (k_prep_stats 21 ms per 56 blocks of 64 MiB with one)
	const u32 stats_parts = (u32)hook_int("DSRC_GPU_HOOK_STATS_PARTS", B >= 192 ? 1 : std::min<long>(16, std::max<long>(2, 512 / B)));
	hipLaunchKernelGGL(k_prep_stats, dim3(stats_parts, B), dim3(WG), 0, s, d_in, d_desc, d_state, rp, prm); KCHK();
	if (prm.color_space) { hipLaunchKernelGGL(k_cs_reduce, dim3((max_rec_cap + WG - 1) / WG, B), dim3(WG), 0, s, d_in, d_desc, d_state, rp, prm); KCHK(); }
	hipLaunchKernelGGL(k_rec_offsets, dim3(B), dim3(WG), 0, s, d_desc, d_state, rp); KCHK();
	const u32 prep_gx = std::max(1u, std::min(64u, (max_rec_cap + 4 * WAVES - 1) / (4 * WAVES)));
	if (streams_early) { hipLaunchKernelGGL(k_prep_write, dim3(prep_gx, B), dim3(WG), 0, s, d_in, d_desc, d_state, rp, d_q, d_qp, d_d, prm); KCHK(); }
	if (crc && !prm.color_space) { hipLaunchKernelGGL(k_crc, dim3(B, 3), dim3(WG), 0, s, d_in, d_desc, d_state, rp, h->d_crc_tab, prm); KCHK(); }
	if (prm.tag_flags || prm.record_layout) { hipLaunchKernelGGL(k_tag_poke, dim3((max_rec_cap + WG - 1) / WG, B), dim3(WG), 0, s, const_cast<u8*>(d_in), d_desc, d_state, rp, prm); KCHK(); }
	hipLaunchKernelGGL(k_tag_template, dim3((B + 63) / 64), dim3(64), 0, s, d_in, d_desc, d_state, rp, B); KCHK();
	HIPCHK(hipMemcpyAsync(st.data(), d_state, sizeof(BlkState) * B, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	mark("S2"); tr.stage("dsrc batch: schemes, streams, tag statistics");

	// ---- phase 3: scheme selection and buffer carving (host) -------------------------------------------
	for (u32 b = 0; b < B; ++b)
		if (st[b].err) return fail(h, DSRCGPU_E_INPUT, "chunk %u cannot be coded (error bits 0x%x, %u records)", b, st[b].err, st[b].n_recs);

	// The symbol streams are carved now that the statistics say how long they are (round 5; before: three arrays of half the chunk each,
	// 12.6 MB per 8 MiB chunk for 6.7 MB of symbols): transformed qualities, kept base indices, and -- only if some block's reads differ
	// in length -- the position contexts (reads of one length: a closed form of t, qua_pctx).  k_prep_write runs behind this readback
	// instead of in front of it: nothing the host decides depends on the streams themselves.
	if (!streams_early)
	{
		u64 qb = 0, db = 0; bool any_var = false;
		for (u32 b = 0; b < B; ++b)
		{
			desc[b].q_base = qb; qb += al((size_t)st[b].q_total + 64, 64);
			desc[b].d_base = db; db += al((size_t)st[b].d_total + 64, 64);
			any_var = any_var || st[b].min_len != st[b].max_len;
		}
		const size_t o_q = A.alloc(qb + 64), o_d = A.alloc(db + 64), o_qp = any_var ? A.alloc(qb + 64) : 0;
		if (A.failed) return fail(h, DSRCGPU_E_NOMEM, "arena exhausted (symbol streams): need > %zu bytes", A.top);
		d_q = AP<u8>(h, o_q); d_d = AP<u8>(h, o_d); d_qp = any_var ? AP<u8>(h, o_qp) : d_q;      // (never touched when every block's reads have one length)
	}

	std::vector<TagPlan> tplan(B);
	std::vector<QuaPlan> qplan(B), dplan(B);
	memset(tplan.data(), 0, sizeof(TagPlan) * B); memset(qplan.data(), 0, sizeof(QuaPlan) * B); memset(dplan.data(), 0, sizeof(QuaPlan) * B);
	std::vector<CtxJob> jobs;
	std::vector<CtxJob> qjobs, djobs;

	static const u32 QN[8] = {16, 32, 64, 128, 16, 32, 64, 128};
	static const u32 ORD1[4] = {3, 2, 1, 1}, ORD2[4] = {4, 3, 2, 1};

	// chained handles: take the state batch seq-1 left (once per batch, also when the batch is re-run with a larger
	// arena), publish ours as soon as every chunk's field count has been folded in (below)
	struct ChainTurn
	{
		dsrcgpu_handle* h; bool published = false;
		~ChainTurn()
		{
			if (!h->chain || published) return;
			std::lock_guard<std::mutex> g(h->chain->m);     // leaving without publishing = failure: release the waiters
			h->chain->failed = true; h->chain->cv.notify_all();
		}
	} turn{h};
	if (h->chain && !h->chain_taken)
	{
		std::unique_lock<std::mutex> g(h->chain->m);
		h->chain->cv.wait(g, [&] { return h->chain->failed || h->chain->next_seq == h->chain_seq; });
		if (h->chain->failed) { turn.published = true; return fail(h, DSRCGPU_E_STATE, "an earlier batch of the chain failed"); }
		h->chain_cap_in = h->chain->fields_cap; h->chain_taken = true;
	}
	// The fold runs on a working copy.  A scheduler lane of the queue form leaves h->fields_cap alone: lane 0 IS the user's handle, and
	// its view of the state is written by whichever lane completes a batch, under q_m (queue_thread) -- never from here, where the
	// other lane's completion could land between two steps of the fold.
	u32 fcap = h->chain ? h->chain_cap_in : h->fields_cap;
	if (!h->chain) turn.published = true;
	for (u32 b = 0; b < B; ++b)
	{	// TagStats::fields capacity emulation (see oracle/dsrc_oracle.c tags_init)
		u32 cap = fcap; int last = -1;
		for (u32 i = 0; i < st[b].n_fields; ++i) if (i == cap) { last = (int)i; cap = cap ? cap * 2 : 1; }
		desc[b].fields_keep_from = last < 0 ? 0u : (u32)last;
		fcap = cap;
	}
	if (tl_queue_lane != 2) h->fields_cap = fcap;
	if (h->chain && !turn.published)
	{
		std::lock_guard<std::mutex> g(h->chain->m);
		if (h->chain->next_seq == h->chain_seq) { h->chain->fields_cap = fcap; h->chain->next_seq = h->chain_seq + 1; }
		turn.published = true;
		h->chain->cv.notify_all();
	}
	// (test hook, timing only: the batch stays busy after it has published its state -- what a large batch does anyway)
	if (const long hold_ms = hook_int("DSRC_GPU_HOOK_BATCH_HOLD_MS", 0)) std::this_thread::sleep_for(std::chrono::milliseconds(hold_ms));

	for (u32 b = 0; b < B; ++b)
	{
		const BlkState& S = st[b];
		BlkDesc& D = desc[b];
		// DNA scheme: DnaNormalModelerProxy / DnaOrderModelerProxy::SelectSchemeId (src/DnaModelerProxy.h:88-123,160-172)
		if (S.d_count == 0) D.d_scheme = 255;
		else if (dna_order == 0) D.d_scheme = S.d_count <= 4 ? 0 : 1;
		else
		{
			D.d_scheme = S.d_count <= 4 ? 0 : 1;
			if (S.d_count > 8) return fail(h, DSRCGPU_E_INPUT, "chunk %u: %u DNA symbols with an order model is undefined in the reference", b, S.d_count);
		}
		// quality scheme
		if (qo == 0)
		{	// QualityNormalModelerProxy::SelectSchemeId (src/QualityModelerProxy.h:113-122)
			if ((float)S.th_len / (float)S.rle_len > 1.25f) D.q_scheme = 2;
			else if ((float)S.raw_len / (float)S.th_len > 1.10f) D.q_scheme = 1;
			else D.q_scheme = 0;
		}
		else if (lossy) D.q_scheme = 0;
		else
		{	// QualityOrderModelerProxyLossless::SelectSchemeId (src/QualityModelerProxy.h:257-283)
			u32 scheme = 255;
			for (u32 i = 0; i < 8; ++i) if ((16u << i) >= S.q_count) { scheme = i; break; }
			if (scheme != 255 && qo == 2)
			{
				const double ratio = (double)S.raw_len / (double)S.rle_len;
				if (S.max_len == S.min_len && ratio > 1.175) scheme += 4;
			}
			if (scheme > 7 || (scheme > 3 && S.q_count > 128)) return fail(h, DSRCGPU_E_INPUT, "chunk %u: %u quality symbols is undefined in the reference's order model", b, S.q_count);
			D.q_scheme = scheme;
		}
	}

	// tag value / run arrays
	for (u32 b = 0; b < B; ++b)
	{
		const BlkState& S = st[b];
		tplan[b].val = A.alloc((size_t)std::max(1u, S.n_num0) * S.n_recs * 4) / 4;
		tplan[b].rl = A.alloc((size_t)std::max(1u, S.n_num0) * 2 * S.n_recs * 2) / 2;
	}
	const size_t o_tplan = A.alloc(sizeof(TagPlan) * B), o_tres = A.alloc(sizeof(TagFieldRes) * B * DSRC_MAX_FIELDS);
	const size_t o_qplan = A.alloc(sizeof(QuaPlan) * B), o_dplan = A.alloc(sizeof(QuaPlan) * B);
	if (A.failed) return fail(h, DSRCGPU_E_NOMEM, "arena exhausted (phase 3a)");
	TagPlan* d_tplan = AP<TagPlan>(h, o_tplan); TagFieldRes* d_tres = AP<TagFieldRes>(h, o_tres);
	QuaPlan* d_qplan = AP<QuaPlan>(h, o_qplan); QuaPlan* d_dplan = AP<QuaPlan>(h, o_dplan);
	u32* wpool = AP<u32>(h, 0); u64* lpool = AP<u64>(h, 0); u16* spool = AP<u16>(h, 0);

	HIPCHK(hipMemcpyAsync(d_desc, desc.data(), sizeof(BlkDesc) * B, hipMemcpyHostToDevice, s));
	HIPCHK(hipMemcpyAsync(d_tplan, tplan.data(), sizeof(TagPlan) * B, hipMemcpyHostToDevice, s));
	if (!streams_early) { hipLaunchKernelGGL(k_prep_write, dim3(prep_gx, B), dim3(WG), 0, s, d_in, d_desc, d_state, rp, d_q, d_qp, d_d, prm); KCHK(); }
	hipLaunchKernelGGL(k_tag_scan, dim3(stats_parts, B), dim3(WG), 0, s, d_in, d_desc, d_state, rp, wpool, d_tplan); KCHK();      // (several workgroups per block in batches of few, large blocks: see k_prep_stats)
	hipLaunchKernelGGL(k_tag_numeric, dim3(B), dim3(WG), 0, s, d_desc, d_state, wpool, spool, d_tplan); KCHK();
	hipLaunchKernelGGL(k_tag_finalize, dim3(B), dim3(64), 0, s, d_state); KCHK();

	// ---- quality / DNA streams -------------------------------------------------------------------------
	// staging: zeroed region (bit-emitted streams) and plain region (byte-written range-coder output)
	size_t zero_lo = al(A.top, 256); A.top = zero_lo;
	for (u32 b = 0; b < B; ++b)
	{
		const BlkState& S = st[b]; BlkDesc& D = desc[b];
		if (qo == 0)
		{
			QuaPlan& P = qplan[b];
			P.scheme = D.q_scheme; P.blk = b;
			size_t cap_bytes;
			if (D.q_scheme <= 1)
			{
				const u32 hw = S.max_len * S.q_count;
				P.hist_words = hw; P.code_off = hw; P.len_off = 2 * hw;
				P.tree_slot = 4 + h_huff_tree_cap(S.q_count); P.tree_off = 3 * hw;
				P.n_trees = S.max_len;
				const u32 tree_words = (u32)(((size_t)S.max_len * P.tree_slot + 3) / 4);
				P.ws_slot = h_huff_ws_words(S.q_count); P.ws_off = P.tree_off + tree_words;
				P.aux_off = P.ws_off + S.max_len * P.ws_slot;
				const size_t words = (size_t)P.aux_off + S.n_recs + 4;
				P.scr = A.alloc(words * 4) / 4;
				cap_bytes = 64 + (size_t)S.max_len * P.tree_slot + ((size_t)S.q_total * (bitlen(S.q_count) + 1) + 7) / 8 + (size_t)S.n_recs * 3;
			}
			else
			{
				const u32 qn = S.q_count, hw = qn * qn + qn * 256;
				P.hist_words = hw; P.code_off = hw; P.len_off = 2 * hw; P.lf_off = 3 * hw;
				P.tree_slot = 4 + h_huff_tree_cap(256); P.tree_off = P.lf_off + 256; P.n_trees = 2 * qn;
				const u32 tree_words = (u32)(((size_t)2 * qn * P.tree_slot + 3) / 4);
				P.ws_slot = h_huff_ws_words(256); P.ws_off = P.tree_off + tree_words;
				const size_t words = (size_t)P.ws_off + (size_t)2 * qn * P.ws_slot + 4;
				P.scr = A.alloc(words * 4) / 4;
				cap_bytes = 128 + (size_t)2 * qn * P.tree_slot + ((size_t)S.q_total * 20 + 7) / 8;
			}
			D.qua_cap = (u32)(al(cap_bytes, 16) / 4 + 4);
			D.qua_out = A.alloc((size_t)D.qua_cap * 4) / 4;
		}
		if (dna_order == 0 && D.d_scheme == 1)
		{
			QuaPlan& P = dplan[b];
			P.scheme = 1; P.blk = b; P.hist_words = 0;
			P.ws_off = 32; P.ws_slot = h_huff_ws_words(20); P.tree_off = P.ws_off + P.ws_slot; P.tree_slot = 4 + h_huff_tree_cap(20);
			const size_t words = P.tree_off + P.tree_slot / 4 + 8;
			P.scr = A.alloc(words * 4) / 4;
			D.dna_cap = (u32)((8 + h_huff_tree_cap(20) + ((size_t)S.d_total * 6 + 7) / 8 + 16) / 4 + 4);
			D.dna_out = A.alloc((size_t)D.dna_cap * 4) / 4;
		}
	}
	size_t zero_hi = al(A.top, 256); A.top = zero_hi;
	// plain (not zeroed) staging + work buffers.  Room for a range-coded stream: a symbol can cost two bytes (freq 1 of a total
	// just under 2^16), but no stream costs that on average -- a row starts at 1 per symbol and gains 2 per hit, i.e. it is the KT
	// estimator, whose code length stays within (N - 1) / 2 * log2(n) bits of n * log2(N) per context: <= 7.1 bits per symbol for the
	// alphabets here.  So: a byte and a sixteenth per symbol; k_rc never writes past a stream's limit, it reports
	// DSRC_ERR_OUT_OVERFLOW, and the batch is then run again with the two bytes per symbol that cannot be exceeded (sticky per handle).
	// (test hook DSRC_GPU_HOOK_RC_BOUND_SHIFT: the estimate divided by 2^k, so that the overflow and the second pass really happen)
	const u32 bound_shift = (u32)hook_int("DSRC_GPU_HOOK_RC_BOUND_SHIFT", 0);
	auto rc_bytes_bound = [&](u32 n) -> size_t { return h->rc_caps_worst ? (size_t)n * 2 : ((size_t)n + n / 16) >> bound_shift; };
	for (u32 b = 0; b < B; ++b)
	{
		const BlkState& S = st[b]; BlkDesc& D = desc[b];
		if (qo == 0 && D.q_scheme == 2)
		{
			const size_t off = A.alloc(((size_t)S.q_total + 2) * 4) / 4;
			qplan[b].run_start = off;
		}
		if (qo > 0)
		{
			D.qua_cap = (u32)((1024 + rc_bytes_bound(S.q_total)) / 4 + 4);
			D.plain_mask |= 1u;
			D.qua_out = A.alloc((size_t)D.qua_cap * 4) / 4;
		}
		if (dna_order == 0 && D.d_scheme == 0) { D.dna_cap = (u32)((S.d_total / 4 + 16) / 4 + 4); D.dna_out = A.alloc((size_t)D.dna_cap * 4) / 4; }
		if (dna_order > 0 && D.d_scheme != 255) D.plain_mask |= 2u;
		if (dna_order > 0 || D.d_scheme == 255)
		{
			D.dna_cap = (u32)((1024 + rc_bytes_bound(S.d_total)) / 4 + 4);
			D.dna_out = A.alloc((size_t)D.dna_cap * 4) / 4;
		}
	}
	// range-coder jobs
	if (qo > 0)
		for (u32 b = 0; b < B; ++b)
		{
			const BlkState& S = st[b]; const BlkDesc& D = desc[b];
			CtxJob j; memset(&j, 0, sizeof(j));
			j.blk = b; j.n = S.q_total; j.src_off = D.q_base; j.is_dna = 0;
			if (lossy) { j.n_alpha = 8; j.order = qo; j.rescale_shift = 4; j.translate = 0; j.out_byte0 = 0; j.scheme = 0; }
			else
			{
				j.n_alpha = QN[D.q_scheme]; j.order = (qo == 1) ? ORD1[D.q_scheme & 3] : ORD2[D.q_scheme & 3];
				const u32 rescale = D.q_scheme < 4 ? 8 : j.n_alpha;
				j.rescale_shift = 7 - log2u(rescale); j.translate = 1; j.out_byte0 = 33; j.scheme = D.q_scheme;
			}
			if (j.order > 7) return fail(h, DSRCGPU_E_ARG, "quality order %u not supported", j.order);
			if (S.min_len == S.max_len && S.max_len >= 1 && S.max_len <= 65535)
			{
				const u64 m = ((1ull << 48) + (S.max_len - S.cs_reduced) - 1) / (S.max_len - S.cs_reduced);
				j.qlen = S.max_len - S.cs_reduced; j.qm_lo = (u32)m; j.qm_hi = (u32)(m >> 32);
			}
			j.alpha_bits = log2u(j.n_alpha); j.key_bits = j.alpha_bits * (j.order + 1);
			j.out_words = D.qua_out; j.out_cap = D.qua_cap * 4 - j.out_byte0 - 16;
			qjobs.push_back(j);
		}
	if (dna_order > 0)
		for (u32 b = 0; b < B; ++b)
		{
			const BlkState& S = st[b]; const BlkDesc& D = desc[b];
			if (D.d_scheme == 255) continue;
			CtxJob j; memset(&j, 0, sizeof(j));
			j.blk = b; j.n = S.d_total; j.src_off = D.d_base; j.is_dna = 1;
			j.n_alpha = D.d_scheme ? 8 : 4; j.alpha_bits = D.d_scheme ? 3 : 2;
			j.order = D.d_scheme ? std::min(dna_order, 7u) : dna_order;
			j.key_bits = j.alpha_bits * j.order; j.scheme = D.d_scheme; j.out_byte0 = 1;
			j.out_words = D.dna_out; j.out_cap = D.dna_cap * 4 - 1 - 16;
			djobs.push_back(j);
		}
	// order jobs by (kind, alphabet) so that replay launches and 64-chain groups are homogeneous
	std::stable_sort(qjobs.begin(), qjobs.end(), [](const CtxJob& a, const CtxJob& b) { return a.n_alpha < b.n_alpha; });
	std::stable_sort(djobs.begin(), djobs.end(), [](const CtxJob& a, const CtxJob& b) { return a.n_alpha < b.n_alpha; });
	jobs = qjobs; jobs.insert(jobs.end(), djobs.begin(), djobs.end());
	const u32 NJ = (u32)jobs.size();
	std::vector<RcChain> chains(NJ);
	{
		// a chain's records: its chunks of 64 six-byte records back to back (k_rc.h) -- the one per-symbol array that stays until the
		// range coder has run.  The loaders may read RC_OVERREAD records' worth past the last chain's array.
		// tests only (read per batch): 1 = the reference-loop check for every chunk of 64 symbols, 2 = every stream reports a carry clamp and goes to the redo list,
		// 3 = a recovery inside k_rcs every few chunks (rcs_recover: the walk of a chunk with the reference's loop, R and L put right)
		const u32 force_exact = hook_env("DSRC_GPU_RC_REDO") ? 2u : hook_env("DSRC_GPU_RC_RECOVER") ? 3u : hook_env("DSRC_GPU_FORCE_EXACT_RC") ? 1u : 0u;
		size_t trip_words = 0;
		std::vector<size_t> cbase(NJ + 1, 0); std::vector<u32> cpitch(NJ + 1, 0);
		for (u32 g = 0; g < NJ; g += RC_LANES)
		{
			const u32 hi = std::min(NJ, g + RC_LANES);
			u32 mx = 0; for (u32 i = g; i < hi; ++i) mx = std::max(mx, jobs[i].n);
			// a stream's array: its chunks of 64 records, 384 bytes each (k_rc.h), in 8-byte units
			const u32 pitch = (mx + 63) / 64 * 48 + 4;
			if ((u64)pitch * sizeof(RcPack) >= (1ull << 32))      // k_rc: 32-bit byte offsets inside one stream's array
				return fail(h, DSRCGPU_E_ARG, "chunk too large for the range-coder stage (a stream of %u symbols exceeds 4 GiB of records); use a smaller buffer size", mx);
			for (u32 i = g; i < hi; ++i) { cbase[i] = trip_words + (size_t)(i - g) * pitch; cpitch[i] = pitch; }
			trip_words += (size_t)pitch * (hi - g);
			if (hi == NJ) trip_words += RC_OVERREAD;                       // over-read slack behind the last array (the loaders request chunks ahead)
		}
		const size_t o_trip = A.alloc(trip_words * sizeof(RcPack) + 256);
		const size_t trip0 = (o_trip + 15) / 16 * 2;                 // in records, 16-byte aligned
		for (u32 i = 0; i < NJ; ++i)
		{
			CtxJob& j = jobs[i];
			j.passes = (j.key_bits + SORT_DIGIT_BITS - 1) / SORT_DIGIT_BITS; if (j.passes == 0) j.passes = 1;
			j.dbits = (j.key_bits + j.passes - 1) / j.passes; if (j.dbits == 0) j.dbits = 1;
			j.sorted_in_b = j.passes & 1;
			j.trip = trip0 + cbase[i];
			RcChain& c = chains[i];
			c.trip = j.trip; c.out_words = j.out_words; c.n = j.n; c.out_byte0 = j.out_byte0; c.out_cap = j.out_cap; c.blk = j.blk; c.is_dna = j.is_dna;
			c.force_exact = force_exact; c.jid = i; c.bk_on = 0;
		}
	}
	// The ping-pong sort buffers are only alive from k_sort to k_replay, so the job list is cut into slices that
	// reuse one region (kernels of consecutive slices are ordered on the stream).  What persists per block until
	// the range coder has run is the record array, which keeps many more blocks in flight per GiB of HBM.
	std::vector<u32> slice_lo;
	{
		const char* env = hook_env("DSRC_GPU_SORT_SLICE_MB");
		// ~128 streams of an 8 MiB chunk per slice.  Larger slices (one k_sort workgroup per CU) are no faster for one
		// instance, and with several instances sharing the GPU shorter launches interleave better (measured: 14 GiB
		// 17.5, 7 GiB 19.6, 3.5 GiB 19.0 GB/s with four instances)
		const size_t budget = env ? (size_t)atol(env) << 20 : slice_budget(in_total);
		size_t need_max = 128;
		for (u32 i = 0; i < NJ; ++i) need_max = std::max(need_max, ((size_t)jobs[i].n * 8 + 64) * 2);
		const u32 max_jobs = (u32)std::max<size_t>(1, budget / need_max);
		const u32 n_slices = std::max(1u, (NJ + max_jobs - 1) / max_jobs);
		const u32 per_slice = std::max(1u, (NJ + n_slices - 1) / n_slices);          // equal slices: no small last launch
		size_t cur = 0, mx = 0;
		for (u32 i = 0; i < NJ; ++i)
		{
			const size_t need = ((size_t)jobs[i].n * 8 + 64) * 2;
			if (i % per_slice == 0) { slice_lo.push_back(i); cur = 0; }
			cur += need; mx = std::max(mx, cur);
		}
		slice_lo.push_back(NJ);
		const size_t o_sort = A.alloc(mx + 64);
		for (size_t sl = 0; sl + 1 < slice_lo.size(); ++sl)
		{
			size_t off = 0;
			for (u32 i = slice_lo[sl]; i < slice_lo[sl + 1]; ++i)
			{
				jobs[i].elems = (o_sort + off) / 8; off += (size_t)jobs[i].n * 8 + 64;
				jobs[i].elems_b = (o_sort + off) / 8; off += (size_t)jobs[i].n * 8 + 64;
			}
		}
	}
	// Bucketed path (k_bucket.h): per job the bucket digit, the key mixing and its words of the `bk` pool; per launch group (the
	// streams of one alphabet size inside a slice) a fallback list that k_part fills with the streams it hands back to k_sort / k_replay
	struct BkGroup { u32 lo, hi, fb; };
	std::vector<std::vector<BkGroup> > bk_groups(slice_lo.size());
	const bool bk_enabled = hook_int("DSRC_GPU_BUCKETS", 1) != 0;
	const u32 bk_min = (u32)hook_int("DSRC_GPU_BUCKETS_MIN", 16384);   // shorter streams: a bucket per workgroup does not pay
	const bool bk_binned = hook_int("DSRC_GPU_BUCKETS_BINNED", 1) != 0;
	// elements per bucket the bucket digit aims at (a wave walks its bucket serially; longer buckets give longer runs per time bin)
	const u64 bk_elems_dna = 4096u;      // 512 buckets of 6.5 k bases: 512 rows of 8 bytes per wave, runs of 16 records per time bin
	const u64 bk_elems_qua = 2048u;
	const u32 bk_limit = (u32)hook_int("DSRC_GPU_BUCKET_LIMIT", (long)BK_LIMIT);
	const u32 bk_big = (u32)hook_int("DSRC_GPU_BUCKET_BIG", (long)BK_BIG);
	const bool use_bk = bk_enabled && NJ > 0 && h->lds64_ordered;      // k_model stands on the LDS applying atomics in lane order (k_lds_order_test)
	size_t o_bk = 0, bk_zero_words = 0, o_bcnt = 0;
	if (use_bk)
	{
		u32 cur = NJ;                                            // [0, NJ): the jobs' fallback flags
		for (size_t sl = 0; sl + 1 < slice_lo.size(); ++sl)
			for (u32 lo = slice_lo[sl]; lo < slice_lo[sl + 1];)
			{
				u32 hi = lo;
				while (hi < slice_lo[sl + 1] && jobs[hi].n_alpha == jobs[lo].n_alpha) ++hi;
				bk_groups[sl].push_back({lo, hi, cur});
				for (u32 i = lo; i < hi; ++i) jobs[i].bk_fb = cur;
				cur += 1 + (hi - lo);
				lo = hi;
			}
		for (u32 i = 0; i < NJ; ++i)
		{
			CtxJob& j = jobs[i];
			j.jid = i;
			const u32 n_bins = (j.n + BK_BIN - 1) >> BK_TB;
			j.bk_binned = bk_binned ? 1u : 0u;
			j.bk_on = (j.n >= bk_min && n_bins <= (u32)hook_int("DSRC_GPU_BUCKET_MAX_BINS", BK_MAX_BINS_WIDE) && j.key_bits <= BK_MAX_HB + BK_MAX_LB) ? 1u : 0u;
			// bucket digit: ~2048 elements per bucket on average, at most 1024 buckets, at most BK_MAX_LB key bits left for the LDS sort
			const u64 per_bucket = j.is_dna ? bk_elems_dna : bk_elems_qua;
			u32 hb = 0; while (hb < BK_MAX_HB && (per_bucket << hb) < j.n) ++hb;
			hb = std::min(hb, j.key_bits);
			if (j.key_bits > BK_MAX_LB) hb = std::max(hb, j.key_bits - BK_MAX_LB);
			j.bk_hb = hb; j.bk_lb = j.key_bits - hb;
			j.bk_mul = BK_HASH_MUL; j.bk_kmask = (u32)((1ull << j.key_bits) - 1ull);
			j.bk_limit = bk_limit; j.bk_big = bk_big;
			j.bk_narrow_bins = (u32)std::min<long>(BK_MAX_BINS, hook_int("DSRC_GPU_BUCKET_NARROW_BINS", BK_MAX_BINS));
		}
		bk_zero_words = cur;
		o_bk = A.alloc(((size_t)cur + 2 + 128) * 4 + 64);            // + 64 records' worth of nowhere (k_model: the lanes of a short window store there)
		// per (tile, bucket) element counts of k_part (u16); k_binoff turns the first row of every time bin into the buckets' offsets
		size_t cnt_words = 0;
		for (u32 i = 0; i < NJ; ++i)
			if (jobs[i].bk_on)
			{
				if (cnt_words >= (1ull << 32)) return fail(h, DSRCGPU_E_NOMEM, "batch too large for the bucket count table");
				jobs[i].bk_cnt = (u32)cnt_words;
				cnt_words += (((size_t)jobs[i].n + BK_BIN - 1) >> BK_TB) << jobs[i].bk_hb;
			}
		o_bcnt = A.alloc(cnt_words * 2 + 64);
	}
	for (u32 i = 0; i < NJ; ++i) chains[i].bk_on = use_bk ? jobs[i].bk_on : 0u;
	// redo list of the range coder (k_rc.h): count, then chain ids -- streams in which the carry clamp fired under k_rcs (the device
	// appends them), streams the bucketed front end handed back on the device (the host appends them after the state read-back)
	const size_t o_redo = A.alloc(((size_t)NJ + 2) * 4);
	const size_t o_jobs = A.alloc(sizeof(CtxJob) * std::max(1u, NJ)), o_chains = A.alloc(sizeof(RcChain) * std::max(1u, NJ));
	if (A.failed) return fail(h, DSRCGPU_E_NOMEM, "arena exhausted (phase 3b): batch needs > %zu bytes of HBM scratch", A.top);
	CtxJob* d_jobs = AP<CtxJob>(h, o_jobs); RcChain* d_chains = AP<RcChain>(h, o_chains);

	// ---- tags: dictionary resources are sized from the finalized field kinds ----------------------------------
	HIPCHK(hipMemcpyAsync(st.data(), d_state, sizeof(BlkState) * B, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	mark("S3"); tr.stage("dsrc batch: tags, order models, range coder");
	std::vector<TagFieldRes> tres((size_t)B * DSRC_MAX_FIELDS);
	memset(tres.data(), 0, sizeof(TagFieldRes) * tres.size());
	size_t tz_lo = al(A.top, 256); A.top = tz_lo;
	for (u32 b = 0; b < B; ++b)
	{
		const BlkState& S = st[b]; BlkDesc& D = desc[b];
		size_t words = 0, hdr = 64;
		tplan[b].rbits_off = (u32)words; words += S.n_recs + 4;
		tplan[b].raw_hist_off = (u32)words; words += 128 + h_huff_ws_words(128) + h_huff_tree_cap(128) / 4 + 16;
		for (u32 fi = 0; fi < S.n_fields && !S.mixed; ++fi)
		{
			const TagField& f = S.fld[fi];
			TagFieldRes& R = tres[(size_t)b * DSRC_MAX_FIELDS + fi];
			hdr += 32 + f.len0 + f.len0 / 8;
			if (f.is_string)
			{
				R.hist_off = (u32)words; words += 129 * 256;
				R.ham_off = (u32)words; words += f.len0 + 4;
				R.code_off = (u32)words; words += 129 * 256;
				R.len_off = (u32)words; words += 129 * 256;
				R.tree_slot = 4 + h_huff_tree_cap(256);
				R.tree_off = (u32)(words * 4); words += ((size_t)129 * R.tree_slot + 3) / 4;
				R.ws_slot = h_huff_ws_words(256); R.ws_off = (u32)words; words += (size_t)129 * R.ws_slot;
				hdr += (size_t)129 * R.tree_slot;
			}
			else if (f.is_numeric && !f.is_constant && f.var_stat_encode)
			{
				R.hist_off = (u32)words; words += 512;
				R.code_off = (u32)words; words += 512;
				R.len_off = (u32)words; words += 512;
				R.tree_slot = 4 + h_huff_tree_cap(512);
				R.tree_off = (u32)(words * 4); words += (R.tree_slot + 3) / 4;
				R.ws_slot = h_huff_ws_words(512); R.ws_off = (u32)words; words += R.ws_slot;
				hdr += R.tree_slot;
			}
		}
		if (S.mixed) hdr += 64 + h_huff_tree_cap(128);
		tplan[b].scr = A.alloc(words * 4) / 4;
		const size_t cap_bytes = hdr + ((size_t)S.raw_tag * 9 + 7) / 8 + (size_t)S.n_recs * ((size_t)S.n_fields * 5 + 8) + 64;
		D.tag_cap = (u32)(al(cap_bytes, 16) / 4 + 4);
		D.tag_out = A.alloc((size_t)D.tag_cap * 4) / 4;
	}
	size_t tz_hi = al(A.top, 256); A.top = tz_hi;
	if (A.failed) return fail(h, DSRCGPU_E_NOMEM, "arena exhausted (tags): batch needs > %zu bytes of HBM scratch", A.top);
	HIPCHK(hipMemsetAsync(h->arena.base + tz_lo, 0, tz_hi - tz_lo, s));
	HIPCHK(hipMemcpyAsync(d_desc, desc.data(), sizeof(BlkDesc) * B, hipMemcpyHostToDevice, s));
	HIPCHK(hipMemcpyAsync(d_tplan, tplan.data(), sizeof(TagPlan) * B, hipMemcpyHostToDevice, s));
	HIPCHK(hipMemcpyAsync(d_tres, tres.data(), sizeof(TagFieldRes) * tres.size(), hipMemcpyHostToDevice, s));
	hipLaunchKernelGGL(k_tag_hist, dim3(B), dim3(WG), 0, s, d_in, d_desc, d_state, rp, wpool, wpool, d_tplan, d_tres); KCHK();
	hipLaunchKernelGGL(k_tag_trees, dim3((DSRC_MAX_FIELDS * DSRC_MAX_STRF + 63) / 64, B), dim3(64), 0, s, d_state, wpool, d_tplan, d_tres); KCHK();
	hipLaunchKernelGGL(k_tag_emit, dim3(B), dim3(WG), 0, s, d_in, d_desc, d_state, rp, wpool, spool, wpool, wpool, d_tplan, d_tres); KCHK();
	hipLaunchKernelGGL(k_tag_raw, dim3(B), dim3(WG), 0, s, d_in, d_desc, d_state, rp, wpool, wpool, d_tplan); KCHK();

	// ---- quality / DNA stream kernels (queued behind the tag kernels on the same stream) -----------------
	if (zero_hi > zero_lo) HIPCHK(hipMemsetAsync(h->arena.base + zero_lo, 0, zero_hi - zero_lo, s));
	HIPCHK(hipMemcpyAsync(d_desc, desc.data(), sizeof(BlkDesc) * B, hipMemcpyHostToDevice, s));
	HIPCHK(hipMemcpyAsync(d_qplan, qplan.data(), sizeof(QuaPlan) * B, hipMemcpyHostToDevice, s));
	HIPCHK(hipMemcpyAsync(d_dplan, dplan.data(), sizeof(QuaPlan) * B, hipMemcpyHostToDevice, s));
	if (NJ)
	{
		HIPCHK(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(CtxJob) * NJ, hipMemcpyHostToDevice, s));
		HIPCHK(hipMemcpyAsync(d_chains, chains.data(), sizeof(RcChain) * NJ, hipMemcpyHostToDevice, s));
	}

	u32 maxq = 1, maxd = 1, max_len = 1, max_qn = 1;
	for (u32 b = 0; b < B; ++b) { maxq = std::max(maxq, st[b].q_total); maxd = std::max(maxd, st[b].d_total); max_len = std::max(max_len, st[b].max_len); max_qn = std::max(max_qn, st[b].q_count); }

	if (qo == 0)
	{
		hipLaunchKernelGGL(k_qpos_hist, dim3(B), dim3(WG), 0, s, d_desc, d_state, rp, d_q, wpool, d_qplan); KCHK();
		hipLaunchKernelGGL(k_qpos_trees, dim3((max_len + 63) / 64, B), dim3(64), 0, s, d_state, wpool, d_qplan); KCHK();
		hipLaunchKernelGGL(k_qpos_emit, dim3(B), dim3(WG), 0, s, d_desc, d_state, rp, d_q, wpool, wpool, d_qplan); KCHK();
		hipLaunchKernelGGL(k_qrle_runs, dim3(B), dim3(WG), 0, s, d_desc, d_state, d_q, wpool, d_qplan); KCHK();
		hipLaunchKernelGGL(k_qrle_hist, dim3(B), dim3(WG), 0, s, d_desc, d_state, d_q, wpool, d_qplan); KCHK();
		hipLaunchKernelGGL(k_qrle_trees, dim3((2 * max_qn + 63) / 64, B), dim3(64), 0, s, d_state, wpool, d_qplan); KCHK();
		hipLaunchKernelGGL(k_qrle_emit, dim3(B), dim3(WG), 0, s, d_desc, d_state, d_q, wpool, wpool, d_qplan); KCHK();
	}
	if (dna_order == 0)
	{
		const u32 gx = std::max(1u, std::min(64u, (maxd / 4 + WG - 1) / WG));
		hipLaunchKernelGGL(k_dna_b2, dim3(gx, B), dim3(WG), 0, s, d_desc, d_state, d_d, wpool); KCHK();
		hipLaunchKernelGGL(k_dna_huff, dim3(B), dim3(WG), 0, s, d_desc, d_state, d_d, wpool, wpool, d_dplan); KCHK();
	}
	hipLaunchKernelGGL(k_dna_none, dim3((B + 63) / 64), dim3(64), 0, s, d_desc, d_state, wpool, B); KCHK();
	h->rc_launches = 0;
	std::function<int(const BkGroup&)> launch_fallback;
	u32* d_bk = use_bk ? AP<u32>(h, o_bk) : nullptr;
	u32* d_redo = AP<u32>(h, o_redo);
	const bool rc_one_lane = hook_env("DSRC_GPU_RC_ONE_LANE") != nullptr;      // tests: k_rc (both recurrences in one lane) for every stream
	if (NJ)
	{
		const u32 nq = (u32)qjobs.size(), nd = (u32)djobs.size();
		const bool sort_atomic = h->sort_atomic;
		hipLaunchKernelGGL(k_rc_headers, dim3((NJ + 63) / 64), dim3(64), 0, s, d_jobs, NJ, d_state, wpool); KCHK();
		HIPCHK(hipMemsetAsync(d_redo, 0, ((size_t)NJ + 2) * 4, s));
		u16* d_bcnt = use_bk ? AP<u16>(h, o_bcnt) : nullptr;
		if (use_bk) HIPCHK(hipMemsetAsync(d_bk, 0, bk_zero_words * 4, s));
		const u32 max_parts = 2048u;         // waves per stream of k_replay (measured: 256 parts 21.4, 512 23.3, 1024 24.3, 2048 24.7, 3200 23.2 GB/s with five instances)
		// k_sort / k_replay_seams / k_replay over the fallback list of one launch group of the bucketed path
		launch_fallback = [&, sort_atomic, max_parts](const BkGroup& g) -> int
		{
			const u32 cnt = g.hi - g.lo;
			const u32* fb = d_bk + g.fb;
			if (sort_atomic) hipLaunchKernelGGL((k_sort<0, true>), dim3(std::min(cnt, 16u)), dim3(SORT_WG), 0, s, d_jobs, lpool, d_d, d_q, d_qp, d_state, fb);
			else hipLaunchKernelGGL((k_sort<0, false>), dim3(std::min(cnt, 16u)), dim3(SORT_WG), 0, s, d_jobs, lpool, d_d, d_q, d_qp, d_state, fb);
			u32 mxn = 1; for (u32 i = g.lo; i < g.hi; ++i) mxn = std::max(mxn, jobs[i].n);
			const u32 parts = std::max(1u, std::min(max_parts, mxn / 1024u));
			const dim3 rgrid(parts * std::min(cnt, 4u));
#define BK_REPLAY(NN) { hipLaunchKernelGGL(k_replay_seams<NN>, rgrid, dim3(REPLAY_WG), 0, s, d_jobs, const_cast<u64*>(lpool), parts, cnt, fb); \
					hipLaunchKernelGGL((k_replay<NN, 0>), rgrid, dim3(REPLAY_WG), 0, s, d_jobs, lpool, AP<RcPack>(h, 0), parts, cnt, fb); }
			switch (jobs[g.lo].n_alpha)
			{
			case 4: BK_REPLAY(4) break; case 8: BK_REPLAY(8) break; case 16: BK_REPLAY(16) break;
			case 32: BK_REPLAY(32) break; case 64: BK_REPLAY(64) break; default: BK_REPLAY(128) break;
			}
#undef BK_REPLAY
			KCHK();
			return 0;
		};
		for (size_t sl = 0; sl + 1 < slice_lo.size(); ++sl)
		{
			const u32 s_lo = slice_lo[sl], s_hi = slice_lo[sl + 1];
			stage_mark(0);
			if (use_bk)
			{	// k_bucket.h: partition, finish in LDS, place -- and behind them k_sort / k_replay for the streams k_part handed back
				u32 slice_bins = 0;
				for (u32 i = s_lo; i < s_hi; ++i) if (jobs[i].bk_on) slice_bins = std::max(slice_bins, (jobs[i].n + BK_BIN - 1) >> BK_TB);
				hipLaunchKernelGGL(k_part, dim3(std::max(1u, slice_bins), s_hi - s_lo), dim3(PART_WG), 0, s, d_jobs + s_lo, lpool, d_d, d_q, d_qp, d_state, d_bk, d_bcnt);
				KCHK();
				stage_mark(0); stage_mark(1);
				if (slice_bins) { hipLaunchKernelGGL(k_binoff, dim3(slice_bins, s_hi - s_lo), dim3(256), 0, s, d_jobs + s_lo, d_bcnt, d_bk); KCHK(); }
				for (const BkGroup& g : bk_groups[sl])
				{
					u32 hb = 0; bool any = false;
					for (u32 i = g.lo; i < g.hi; ++i) if (jobs[i].bk_on) { any = true; hb = std::max(hb, jobs[i].bk_hb); }
					if (!any) continue;
					u32 lbm = 0; for (u32 i = g.lo; i < g.hi; ++i) if (jobs[i].bk_on) lbm = std::max(lbm, jobs[i].bk_lb);
					// rows by key where a bucket's keys fit (in as little LDS as they need), else handed out on first use through a map
#define BK_LAUNCH(NN, MB, RB) hipLaunchKernelGGL((k_model<NN, MB, RB>), dim3(((1u << hb) + MD_WAVES_FOR(RB) - 1) / MD_WAVES_FOR(RB), g.hi - g.lo), dim3(64 * MD_WAVES_FOR(RB)), 0, s, d_jobs + g.lo, lpool, AP<RcPack>(h, 0), d_bk, d_bcnt, (RcPack*)(d_bk + ((bk_zero_words + 1) & ~(size_t)1)))
#define BK_FINISH(NN) { if ((1u << lbm) * 4 * MdRow<NN>::STRIDE <= 4096) BK_LAUNCH(NN, 0, 4096); \
						else if ((1u << lbm) * 4 * MdRow<NN>::STRIDE <= MD_ROW_BYTES) BK_LAUNCH(NN, 0, MD_ROW_BYTES); \
						else if (lbm <= 10) BK_LAUNCH(NN, 10, MD_ROW_BYTES); \
						else BK_LAUNCH(NN, BK_MAX_LB, MD_ROW_BYTES); }
					switch (jobs[g.lo].n_alpha)
					{
					case 4: BK_FINISH(4) break; case 8: BK_FINISH(8) break; case 16: BK_FINISH(16) break;
					case 32: BK_FINISH(32) break; case 64: BK_FINISH(64) break; default: BK_FINISH(128) break;
					}
#undef BK_LAUNCH
#undef BK_FINISH
					KCHK();
				}
				if (bk_binned && slice_bins) { hipLaunchKernelGGL(k_place, dim3(slice_bins, s_hi - s_lo), dim3(PLACE_WG), 0, s, d_jobs + s_lo, lpool, AP<RcPack>(h, 0), d_bk); KCHK(); }
				for (const BkGroup& g : bk_groups[sl])
				{	// the group's fallback list: here only where the host itself took a stream off the path (too short, too many tiles).  What
					// the device hands back (k_part / k_model: a bucket too long, too many contexts) is rare and is dealt with after the batch's
					// state read-back (below) -- rounds 4-5 launched these kernels for every group, and their empty workgroups (144 KB of LDS each)
					// queued for CUs on the instance's serial stream: 128 launches, 48 ms in the bench profile
					bool known = false;
					for (u32 i = g.lo; i < g.hi; ++i) known = known || !jobs[i].bk_on;
					if (known) { const int rc = launch_fallback(g); if (rc) return rc; }
				}
				stage_mark(1);
				continue;
			}
#ifdef DSRC_SORT_PROBE
			{	// timing experiments (not a product path): what each phase of k_sort costs, first slice of the first batches
				static int shots = 0;
				if (sl == 0 && shots < 2)
				{
					++shots;
					hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define PROBE_RUN(M) { hipEventRecord(e0, s); if (sort_atomic) hipLaunchKernelGGL((k_sort<M, true>), dim3(s_hi - s_lo), dim3(SORT_WG), 0, s, d_jobs + s_lo, lpool, d_d, d_q, d_qp, d_state, nullptr); else hipLaunchKernelGGL((k_sort<M, false>), dim3(s_hi - s_lo), dim3(SORT_WG), 0, s, d_jobs + s_lo, lpool, d_d, d_q, d_qp, d_state, nullptr); \
					hipEventRecord(e1, s); hipEventSynchronize(e1); float ms = 0; hipEventElapsedTime(&ms, e0, e1); fprintf(stderr, "[probe] k_sort<%u> %u streams: %.2f ms\n", (unsigned)M, s_hi - s_lo, ms); }
					PROBE_RUN(0) PROBE_RUN(0) PROBE_RUN(128) PROBE_RUN(129) PROBE_RUN(130) PROBE_RUN(132) PROBE_RUN(134) PROBE_RUN(136) PROBE_RUN(144) PROBE_RUN(160) PROBE_RUN(192) PROBE_RUN(255)
					hipEventDestroy(e0); hipEventDestroy(e1);
				}
			}
#endif
			if (sort_atomic) hipLaunchKernelGGL((k_sort<0, true>), dim3(s_hi - s_lo), dim3(SORT_WG), 0, s, d_jobs + s_lo, lpool, d_d, d_q, d_qp, d_state, nullptr);
			else hipLaunchKernelGGL((k_sort<0, false>), dim3(s_hi - s_lo), dim3(SORT_WG), 0, s, d_jobs + s_lo, lpool, d_d, d_q, d_qp, d_state, nullptr);
			KCHK();
			stage_mark(0); stage_mark(1);
			for (u32 lo = s_lo; lo < s_hi;)
			{
				u32 hi = lo;
				while (hi < s_hi && jobs[hi].n_alpha == jobs[lo].n_alpha) ++hi;
				const u32 cnt = hi - lo;
				u32 mxn = 1; for (u32 i = lo; i < hi; ++i) mxn = std::max(mxn, jobs[i].n);
				// REPLAY_WG/64 waves per part.  Many waves per chain = few chains in flight: the scattered 8-byte records of a
				// chain have to meet in the memory-side cache before their line is evicted, and with several scheduler instances
				// streaming through that cache the time a chain is open counts (one instance, 512 DNA chains: 32 parts 164 ms,
				// 512 parts 71 ms; five instances: 256 parts 21.4, 512 23.3, 1024 24.3, 2048 24.7, 3200 23.2 GB/s)
				const u32 parts = std::max(1u, std::min(max_parts, mxn / 1024u));
				const dim3 rgrid(parts * cnt);                           // replay_slot
#ifdef DSRC_SORT_PROBE
				{
					static int shots = 0;
					if ((jobs[lo].n_alpha == 32 || jobs[lo].n_alpha == 4) && lo == s_lo && shots < 4 && (sl == 0 || sl + 2 == slice_lo.size()))
					{
						++shots;
						hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RPROBE_RUN(NN, M) { hipEventRecord(e0, s); hipLaunchKernelGGL((k_replay<NN, M>), rgrid, dim3(REPLAY_WG), 0, s, d_jobs + lo, lpool, AP<RcPack>(h, 0), parts, cnt, nullptr); \
						hipEventRecord(e1, s); hipEventSynchronize(e1); float ms = 0; hipEventElapsedTime(&ms, e0, e1); fprintf(stderr, "[probe] k_replay<%d,%d> %u streams x %u parts: %.2f ms\n", NN, M, cnt, parts, ms); }
						hipLaunchKernelGGL(k_replay_seams<32>, rgrid, dim3(REPLAY_WG), 0, s, d_jobs + lo, const_cast<u64*>(lpool), parts, cnt, nullptr);
						if (jobs[lo].n_alpha == 32) { RPROBE_RUN(32, 0) RPROBE_RUN(32, 0) RPROBE_RUN(32, 1) RPROBE_RUN(32, 2) RPROBE_RUN(32, 8) RPROBE_RUN(32, 16) }
						else { RPROBE_RUN(4, 0) RPROBE_RUN(4, 0) RPROBE_RUN(4, 1) RPROBE_RUN(4, 2) RPROBE_RUN(4, 8) RPROBE_RUN(4, 16) }
						hipEventDestroy(e0); hipEventDestroy(e1);
					}
				}
#endif
				switch (jobs[lo].n_alpha)
				{
				case 4:   hipLaunchKernelGGL(k_replay_seams<4>, rgrid, dim3(REPLAY_WG), 0, s, d_jobs + lo, const_cast<u64*>(lpool), parts, cnt, nullptr); hipLaunchKernelGGL((k_replay<4, DSRC_REPLAY_WHATIF>), rgrid, dim3(REPLAY_WG), 0, s, d_jobs + lo, lpool, AP<RcPack>(h, 0), parts, cnt, nullptr); break;
				case 8:   hipLaunchKernelGGL(k_replay_seams<8>, rgrid, dim3(REPLAY_WG), 0, s, d_jobs + lo, const_cast<u64*>(lpool), parts, cnt, nullptr); hipLaunchKernelGGL((k_replay<8, DSRC_REPLAY_WHATIF>), rgrid, dim3(REPLAY_WG), 0, s, d_jobs + lo, lpool, AP<RcPack>(h, 0), parts, cnt, nullptr); break;
				case 16:  hipLaunchKernelGGL(k_replay_seams<16>, rgrid, dim3(REPLAY_WG), 0, s, d_jobs + lo, const_cast<u64*>(lpool), parts, cnt, nullptr); hipLaunchKernelGGL((k_replay<16, DSRC_REPLAY_WHATIF>), rgrid, dim3(REPLAY_WG), 0, s, d_jobs + lo, lpool, AP<RcPack>(h, 0), parts, cnt, nullptr); break;
				case 32:  hipLaunchKernelGGL(k_replay_seams<32>, rgrid, dim3(REPLAY_WG), 0, s, d_jobs + lo, const_cast<u64*>(lpool), parts, cnt, nullptr); hipLaunchKernelGGL((k_replay<32, DSRC_REPLAY_WHATIF>), rgrid, dim3(REPLAY_WG), 0, s, d_jobs + lo, lpool, AP<RcPack>(h, 0), parts, cnt, nullptr); break;
				case 64:  hipLaunchKernelGGL(k_replay_seams<64>, rgrid, dim3(REPLAY_WG), 0, s, d_jobs + lo, const_cast<u64*>(lpool), parts, cnt, nullptr); hipLaunchKernelGGL((k_replay<64, DSRC_REPLAY_WHATIF>), rgrid, dim3(REPLAY_WG), 0, s, d_jobs + lo, lpool, AP<RcPack>(h, 0), parts, cnt, nullptr); break;
				default:  hipLaunchKernelGGL(k_replay_seams<128>, rgrid, dim3(REPLAY_WG), 0, s, d_jobs + lo, const_cast<u64*>(lpool), parts, cnt, nullptr); hipLaunchKernelGGL((k_replay<128, DSRC_REPLAY_WHATIF>), rgrid, dim3(REPLAY_WG), 0, s, d_jobs + lo, lpool, AP<RcPack>(h, 0), parts, cnt, nullptr); break;
				}
				KCHK();
				lo = hi;
			}
			stage_mark(1);
		}
		// the serial coder runs on a stream of its own: its launch must not queue behind this instance's next kernels, nor they behind it (see dsrcgpu_create; the
		// other instances' data-parallel kernels run beside it)
		HIPCHK(hipEventRecord(h->ev[4], s));
		HIPCHK(hipStreamWaitEvent(h->rc_stream, h->ev[4], 0));
		HIPCHK(hipEventRecord(h->ev[2], h->rc_stream));
		if (rc_one_lane) hipLaunchKernelGGL(k_rc, dim3((NJ + RC_LANES - 1) / RC_LANES), dim3(64 * RC_WG_WAVES), 0, h->rc_stream, d_chains, NJ, (const u32*)nullptr, AP<RcPack>(h, 0), wpool, d_state);
		// 32 streams per workgroup where the launch is large (fewer CUs under the serial waves: 61.3 against 59.1 GB/s with four instances
		// of 900 streams), 16 where it is small and its time is what the caller waits for (76 against 101 ms per launch)
		else if (NJ > (u32)hook_int("DSRC_GPU_RC_WIDE_FROM", 640)) hipLaunchKernelGGL(k_rcs<RCS_WIDE>, dim3((NJ + RCS_WIDE - 1) / RCS_WIDE), dim3(64 * RCS_WG_WAVES), 0, h->rc_stream, d_chains, NJ, AP<RcPack>(h, 0), wpool, d_state, (const u32*)d_bk, d_redo);
		else hipLaunchKernelGGL(k_rcs<16>, dim3((NJ + 15) / 16), dim3(64 * RCS_WG_WAVES), 0, h->rc_stream, d_chains, NJ, AP<RcPack>(h, 0), wpool, d_state, (const u32*)d_bk, d_redo);
		KCHK();
		HIPCHK(hipEventRecord(h->ev[3], h->rc_stream));
		HIPCHK(hipStreamWaitEvent(s, h->ev[3], 0));
		h->rc_launches = 1;
	}

	hipLaunchKernelGGL(k_meta_plan, dim3((B + 63) / 64), dim3(64), 0, s, d_state, prm); KCHK();

	// ---- sizes -> output layout -> assembly -----------------------------------------------------------------------
	HIPCHK(hipMemcpyAsync(st.data(), d_state, sizeof(BlkState) * B, hipMemcpyDeviceToHost, s));
	std::vector<u32> bk_flags, redo;
	if (NJ)
	{
		redo.resize((size_t)NJ + 1); HIPCHK(hipMemcpyAsync(redo.data(), d_redo, ((size_t)NJ + 1) * 4, hipMemcpyDeviceToHost, s));
		if (use_bk) { bk_flags.resize(NJ); HIPCHK(hipMemcpyAsync(bk_flags.data(), d_bk, (size_t)NJ * 4, hipMemcpyDeviceToHost, s)); }
	}
	HIPCHK(hipStreamSynchronize(s));
	if (NJ && !rc_one_lane)
	{	// ---- the range coder's redo list: streams whose carry clamp fired under k_rcs (the device listed them) and streams the
		// device handed back to k_sort / k_replay (their front end runs now): k_rc, both recurrences in one lane, from the start
		std::vector<u32> list(redo.begin() + 1, redo.begin() + 1 + std::min<u32>(redo[0], NJ));
		u32 why[4] = {0, 0, 0, 0};
		for (u32& x : list) { ++why[(x >> 28) & 3u]; x &= 0x0FFFFFFFu; }
		bool handed = false;
		for (u32 i = 0; i < NJ && use_bk; ++i) if (bk_flags[i] && jobs[i].bk_on) { list.push_back(i); handed = true; }
		if (!list.empty())
		{
			mark("redo"); tr.stage("dsrc batch: range coder, redo list");
			if (handed)
				for (size_t sl = 0; sl + 1 < slice_lo.size(); ++sl)
					for (const BkGroup& g : bk_groups[sl])
					{
						bool any = false;
						for (u32 i = g.lo; i < g.hi; ++i) any = any || (bk_flags[i] && jobs[i].bk_on);
						if (any) { const int rc = launch_fallback(g); if (rc) return rc; }
					}
			std::sort(list.begin(), list.end()); list.erase(std::unique(list.begin(), list.end()), list.end());
			std::vector<u32> up(list.size() + 1); up[0] = (u32)list.size(); std::copy(list.begin(), list.end(), up.begin() + 1);
			HIPCHK(hipMemcpyAsync(d_redo, up.data(), up.size() * 4, hipMemcpyHostToDevice, s));
			hipLaunchKernelGGL(k_rc, dim3(((u32)list.size() + RC_LANES - 1) / RC_LANES), dim3(64 * RC_WG_WAVES), 0, s, d_chains, NJ, (const u32*)d_redo, AP<RcPack>(h, 0), wpool, d_state); KCHK();
			HIPCHK(hipMemcpyAsync(st.data(), d_state, sizeof(BlkState) * B, hipMemcpyDeviceToHost, s));
			HIPCHK(hipStreamSynchronize(s));
			h->rc_redone += (u32)list.size();
			if (getenv("DSRC_GPU_DEBUG")) fprintf(stderr, "[dsrc_gpu] range coder: %zu of %u streams coded again by k_rc (%u carry clamps: fix row taken %u, recovery refused %u, forced %u; the others handed back by the bucketed front end)\n", list.size(), NJ, std::min<u32>(redo[0], NJ), why[1], why[2], why[3]);
		}
	}
	mark("S4"); tr.stage("dsrc batch: assemble");
	if (!bk_flags.empty() && getenv("DSRC_GPU_DEBUG"))
	{
		u32 on = 0, back = 0;
		for (u32 i = 0; i < NJ; ++i) { on += jobs[i].bk_on; back += bk_flags[i] && jobs[i].bk_on; }
		fprintf(stderr, "[dsrc_gpu] bucketed path: %u of %u streams tried, %u handed back to k_sort / k_replay\n", on, NJ, back);
	}
	u64 total = 0;
	for (u32 b = 0; b < B; ++b)
	{
		const BlkState& S = st[b];
		if ((S.err & DSRC_ERR_OUT_OVERFLOW) && !h->rc_caps_worst && NJ)
		{	// (see rc_bytes_bound) again, with the staging sized for the worst case: with_arena_retry_ re-runs a batch that reports
			// a short arena
			h->rc_caps_worst = true; A.failed = true;
			return fail(h, DSRCGPU_E_NOMEM, "chunk %u: a range-coded stream outgrew its staging estimate; the batch is run again with worst-case staging", b);
		}
		if (S.err) return fail(h, DSRCGPU_E_INPUT, "chunk %u cannot be coded (error bits 0x%x)", b, S.err);
		if (S.tag_bytes > (u64)desc[b].tag_cap * 4 || S.qua_bytes > (u64)desc[b].qua_cap * 4 || S.dna_bytes > (u64)desc[b].dna_cap * 4)
			return fail(h, DSRCGPU_E_INPUT, "chunk %u: staging overflow (tag %u/%u qua %u/%u dna %u/%u)", b, S.tag_bytes, desc[b].tag_cap * 4, S.qua_bytes, desc[b].qua_cap * 4, S.dna_bytes, desc[b].dna_cap * 4);
		desc[b].out_off = total;
		if (h->set.verify_after_compress) { if (b == 0) h->verify_hints.resize(B); h->verify_hints[b] = DecHint{S.meta_bytes + S.tag_bytes + S.qua_bytes, S.d_total, desc[b].d_scheme, 0u}; }
		const u64 sz = (u64)S.meta_bytes + S.tag_bytes + S.qua_bytes + S.dna_bytes;
		io.out_offs[b] = total; io.out_sizes[b] = sz;
		io.raw[4 * b + 0] = 0; io.raw[4 * b + 1] = S.raw_tag; io.raw[4 * b + 2] = S.raw_dna; io.raw[4 * b + 3] = S.raw_qua;
		io.comp[4 * b + 0] = S.meta_bytes; io.comp[4 * b + 1] = S.tag_bytes; io.comp[4 * b + 2] = S.dna_bytes; io.comp[4 * b + 3] = S.qua_bytes;
		total += sz;
	}
	u8* d_out = io.d_out;
	if (!d_out)
	{
		const size_t o_out = A.alloc(total + 64);
		if (A.failed) return fail(h, DSRCGPU_E_NOMEM, "arena exhausted (output)");
		d_out = AP<u8>(h, o_out);
		if (total > io.host_cap) return fail(h, DSRCGPU_E_CAPACITY, "output needs %llu bytes, caller gave %llu", (unsigned long long)total, (unsigned long long)io.host_cap);
	}
	else if (total > io.out_cap) return fail(h, DSRCGPU_E_CAPACITY, "output needs %llu bytes, caller gave %llu", (unsigned long long)total, (unsigned long long)io.out_cap);
	h->last_d_out = d_out;
	HIPCHK(hipMemcpyAsync(d_desc, desc.data(), sizeof(BlkDesc) * B, hipMemcpyHostToDevice, s));
	hipLaunchKernelGGL(k_assemble, dim3(16, B), dim3(WG), 0, s, d_desc, d_state, wpool, d_out, prm); KCHK();
	HIPCHK(hipEventRecord(h->ev[1], s));
	if (io.host_out) HIPCHK(hipMemcpyAsync(io.host_out, d_out, total, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	mark("done");
	if (trace) fprintf(stderr, "[dsrc_gpu] %p timeline:%s\n", (void*)h, tl.c_str());
	if (getenv("DSRC_GPU_DEBUG")) fprintf(stderr, "[dsrc_gpu] batch of %u chunks, %zu input bytes: arena used %zu of %zu bytes\n", B, in_total, A.top, A.cap);
	hipEventElapsedTime(&h->batch_ms, h->ev[0], h->ev[1]);
	if (h->rc_launches) hipEventElapsedTime(&h->rc_ms, h->ev[2], h->ev[3]); else h->rc_ms = 0.f;
	h->sort_ms = h->replay_ms = 0.f;
	for (u32 i = 0; i + 1 < h->stage_used; i += 2)
	{
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, h->stage_ev[i], h->stage_ev[i + 1]) == hipSuccess) (h->stage_kind[i] == 0 ? h->sort_ms : h->replay_ms) += ms;
	}
	return DSRCGPU_OK;
}

// Arena sizes are estimated from the input sizes; if carving runs out (A.failed) the batch is simply
// re-run with a larger arena.  Compressor state is restored so that the retry is invisible.
// a batch that ends in an error before it has published its state must not leave the chain's later batches waiting
void chain_abort(dsrcgpu_handle* h)
{
	if (!h->chain) return;
	std::lock_guard<std::mutex> g(h->chain->m);
	if (h->chain->next_seq <= h->chain_seq) { h->chain->failed = true; h->chain->cv.notify_all(); }
}

template <typename F> int with_arena_retry_(dsrcgpu_handle* h, size_t initial, F&& body);
template <typename F> int with_arena_retry(dsrcgpu_handle* h, size_t initial, F&& body)
{
	const int rc = with_arena_retry_(h, initial, body);
	if (rc != DSRCGPU_OK) chain_abort(h);
	else h->chain_batch_done = true;
	return rc;
}

template <typename F> int with_arena_retry_(dsrcgpu_handle* h, size_t initial, F&& body)
{
	size_t need = initial;
	const u32 saved_cap = h->fields_cap;
	for (int attempt = 0; attempt < 8; ++attempt)
	{
		int rc = ensure_arena(h, need);
		if (rc) return rc;
		rc = body();
		// a batch that did not complete (capacity, checksum, input, memory) leaves the block-to-block state where it was: the
		// caller may run the same batch again (grow-and-retry) and must get the blocks a fresh pass would have written
		if (rc != DSRCGPU_OK && tl_queue_lane != 2) h->fields_cap = saved_cap;
		// (a fixed arena is tried again as well -- the worst-case staging of the range coder may still fit it; ensure_arena says so if not)
		if (rc != DSRCGPU_E_NOMEM || !h->arena.failed) return rc;
		need = std::max(h->arena.top + h->arena.top / 8, need + need / 4);      // A.top is a lower bound of what the failed pass needed
		if (h->arena_fixed && h->arena.top <= h->arena_fixed) need = std::min<size_t>(need, (size_t)h->arena_fixed);      // all of it, once more
	}
	return fail(h, DSRCGPU_E_NOMEM, "batch does not fit in HBM scratch after 8 attempts");
}

// ---- decompression ------------------------------------------------------------------------------------------------
struct DecodeIO
{
	const u8* d_in; const u64* offs; const u64* sizes; u32 n;
	const u64* text_caps;            // optional: text bytes to reserve per block (blocks written by the record-level API declare a running total)
	u8* d_out; u64 out_cap;          // device output (nullptr => arena-allocated, copied to host_out)
	u8* host_out; u64 host_cap;
	u64* out_offs; u64* out_sizes;
	u32* crc_ok;                     // optional, per block: 1 = the stored checksums match the decoded records
	const DecHint* hints = nullptr;  // optional (verification of blocks just written): start and length of every block's DNA stream
};

// ---- model tables of the range-decoded levels (here DENSE: every row is reachable) ------------------------
// u32 words of the table of an order-context quality scheme: alphabet^order contexts x position contexts x alphabet counters
// `cnt` = symbols present in the block (0: the whole alphabet): contexts are made of decoded symbols, so a context holding a
// symbol the block does not have is never reached and the decoder's rows are numbered in base cnt (k_dec_rc.h)
u64 q_table_words(u32 quality_order, bool lossy, u32 scheme, u32 cnt, u32* n_out)
{
	u32 n, ord, resc;
	if (lossy) { n = 8; ord = quality_order; resc = 8; }
	else
	{
		const u32 sc = scheme & 3u;
		n = 16u << sc;
		ord = quality_order == 1 ? (sc == 0 ? 3u : sc == 1 ? 2u : 1u) : (4u - sc);
		resc = scheme < 4 ? 8u : n;
	}
	if (n_out) *n_out = n;
	(void)cnt;                  // rows numbered in base cnt were measured slower (3.18 instead of 2.79 s per 2400 blocks): not used
	return ((u64)1 << (log2u(n) * ord)) * resc * n / 2;
}
// the same for the order-k DNA models: 4 symbols at the full order, 8 symbols at order <= 7 (src/DnaModelerProxy.h:196-229)
u64 d_table_words(u32 dna_order, u32 d_scheme)
{
	return d_scheme ? ((u64)1 << (3 * std::min(dna_order, 7u))) * 4 : ((u64)1 << (2 * dna_order)) * 2;
}
// the largest table the settings can ask for: one slot of the one-lane decoder (DSRC_GPU_DEC_SERIAL)
u64 dec_table_words(const dsrcgpu_settings& set)
{
	u64 q = 0, d = 0;
	const u32 qo = set.quality_order, dn = set.dna_order;
	if (qo > 0 && set.lossy) q = q_table_words(qo, true, 0, 0, nullptr);
	else if (qo > 0) for (u32 sch = 0; sch < 8; ++sch) q = std::max(q, q_table_words(qo, false, sch, 0, nullptr));
	if (dn > 0) d = std::max(d_table_words(dn, 0), d_table_words(dn, 1));
	return std::max<u64>(std::max(q, d), 16);
}

// bytes of HBM the model tables of one decode pass may take: DSRC_GPU_DEC_TABLE_MB / dsrcgpu_set_table_budget, else 70 % of
// what is free (counting the region the handle already holds), never more than 160 GiB
u64 dec_table_budget(const dsrcgpu_handle* h)
{
	if (const char* env = getenv("DSRC_GPU_DEC_TABLE_MB")) return (u64)atol(env) << 20;
	if (h->dec_table_budget) return h->dec_table_budget;
	size_t free_b = 0, total_b = 0;
	if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return (u64)8192 << 20;
	const u64 avail = (u64)free_b + h->dec_tables_cap;
	const u64 rest = avail > ((u64)2048 << 20) ? avail - ((u64)2048 << 20) : 0;
	return std::min<u64>(std::max<u64>(rest * 7 / 10, (u64)256 << 20), (u64)160 << 30);
}

// the table region lives outside the batch arena (its size is known only in the middle of a pass) and is kept between passes
int ensure_dec_tables(dsrcgpu_handle* h, u64 bytes)
{
	if (h->dec_tables_cap >= bytes) return 0;
	if (h->dec_tables) { HIPCHK(hipStreamSynchronize(h->stream)); HIPCHK(hipFree(h->dec_tables)); h->dec_tables = nullptr; h->dec_tables_cap = 0; }
	const u64 want = bytes + bytes / 16 + 4096;
	hipError_t e = hipMalloc((void**)&h->dec_tables, want);
	if (e != hipSuccess) return fail(h, DSRCGPU_E_HIP, "hipMalloc(%llu) for the decoder's model tables failed: %s", (unsigned long long)want, hipGetErrorString(e));
	h->dec_tables_cap = want;
	return 0;
}

// Rounds: the tables of as many blocks as fit the region at once; a round is one streaming clear and one decode launch.
struct DecRound { u32 first, count; };
std::vector<DecRound> plan_rounds(std::vector<DecTab>& tabs, u64 region_words)
{
	std::vector<DecRound> rounds;
	u64 top = 0; u32 first = 0;
	for (u32 i = 0; i < tabs.size(); ++i)
	{
		const u64 w = (tabs[i].words + 31) & ~31ull;                  // 128-byte aligned tables
		if (top + w > region_words && i > first) { rounds.push_back({first, i - first}); first = i; top = 0; }
		tabs[i].off = top; top += w;
	}
	if (first < tabs.size()) rounds.push_back({first, (u32)tabs.size() - first});
	return rounds;
}

int run_decode(dsrcgpu_handle* h, DecodeIO io)
{
	const u32 B = io.n;
	if (B == 0) return DSRCGPU_OK;
	// DSRC_GPU_DEBUG=2: host-side timeline of the phases of this pass (ms since the call)
	static const bool trace = getenv("DSRC_GPU_DEBUG") && atoi(getenv("DSRC_GPU_DEBUG")) >= 2;
	const auto t_call = std::chrono::steady_clock::now();
	std::string tl;
	auto mark = [&](const char* what) { if (trace) { char b[64]; snprintf(b, sizeof b, " %s %.1f", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count()); tl += b; } };
	hipStream_t s = h->stream;
	Arena& A = h->arena;
	DecParams prm; memset(&prm, 0, sizeof(prm));
	prm.dna_order = h->set.dna_order; prm.quality_order = h->set.quality_order; prm.lossy = h->set.lossy ? 1u : 0u;
	prm.crc = h->set.calculate_crc32 ? 1u : 0u; prm.quality_offset = h->ds.quality_offset; prm.n_blocks = B;
	prm.tag_flags = (u32)h->set.tag_preserve_flags; prm.plus_rep = h->ds.plus_repetition ? 1u : 0u; prm.color_space = h->ds.color_space ? 1u : 0u;
	prm.serial_quality = hook_env("DSRC_GPU_DEC_SERIAL") ? 1u : 0u;
	const bool q_rc = prm.quality_order > 0, d_rc = prm.dna_order > 0;

	std::vector<DecDesc> desc(B); std::vector<DecState> st(B);
	memset(desc.data(), 0, sizeof(DecDesc) * B);
	for (u32 b = 0; b < B; ++b)
	{
		if (io.sizes[b] < 16 || io.sizes[b] >= (1ull << 31)) return fail(h, DSRCGPU_E_ARG, "block %u: size %llu out of range", b, (unsigned long long)io.sizes[b]);
		desc[b].in_off = io.offs[b]; desc[b].in_size = (u32)io.sizes[b];
	}
	const size_t o_desc = A.alloc(sizeof(DecDesc) * B), o_state = A.alloc(sizeof(DecState) * B);
	const size_t o_qtabs = A.alloc(sizeof(DecTab) * B), o_dtabs = A.alloc(sizeof(DecTab) * B);
	const size_t o_hints = A.alloc(sizeof(DecHint) * B);
	if (A.failed) return fail(h, DSRCGPU_E_NOMEM, "arena exhausted (decode, phase 1)");
	DecDesc* d_desc = AP<DecDesc>(h, o_desc); DecState* d_state = AP<DecState>(h, o_state);
	HIPCHK(hipEventRecord(h->ev[0], s));
	TraceRange tr; tr.stage("dsrc decode: headers");
	HIPCHK(hipMemcpyAsync(d_desc, desc.data(), sizeof(DecDesc) * B, hipMemcpyHostToDevice, s));
	hipLaunchKernelGGL(k_dec_meta, dim3((B + 63) / 64), dim3(64), 0, s, io.d_in, d_desc, d_state, prm); KCHK();
	HIPCHK(hipMemcpyAsync(st.data(), d_state, sizeof(DecState) * B, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	mark("meta"); tr.stage("dsrc decode: titles");

	// what the device says about the blocks so far (mid-pass and final)
	auto check_blocks = [&](bool final) -> int
	{
		for (u32 b = 0; b < B; ++b)
		{
			const DecState& S = st[b];
			if (S.err == DEC_ERR_TEXT) return fail(h, DSRCGPU_E_CAPACITY, "block %u: the decoded text does not fit the %u bytes reserved for it", b, desc[b].out_cap);
			if (S.err) return fail(h, DSRCGPU_E_INPUT, "block %u cannot be decoded (error bits 0x%x: 1 truncated, 2 malformed, 4 text overflow, 8 undefined in the reference's decoder, 16 scratch)", b, S.err);
			if (final && S.end_pos != desc[b].in_size) return fail(h, DSRCGPU_E_INPUT, "block %u: %u of %u bytes consumed (wrong settings for this archive?)", b, S.end_pos, desc[b].in_size);
		}
		return DSRCGPU_OK;
	};

	// ---- layout: text, record tables, symbol scratch, tree pools ---------------------------------------------------------
	u64 text_total = 0, recs = 0, dbytes = 0, nodes = 0, fbytes = 0; u32 max_recs = 1;
	for (u32 b = 0; b < B; ++b)
	{
		const DecState& S = st[b]; DecDesc& D = desc[b];
		if (S.err) return fail(h, DSRCGPU_E_INPUT, "block %u cannot be decoded (error bits 0x%x: 1 truncated, 2 malformed, 8 undefined in the reference's decoder)", b, S.err);
		const u64 cap = io.text_caps ? io.text_caps[b] : (u64)S.chunk_size + 1;
		if (cap >= (1ull << 31)) return fail(h, DSRCGPU_E_ARG, "block %u declares a chunk of %llu bytes", b, (unsigned long long)cap);
		D.out_off = text_total; D.out_cap = (u32)cap; text_total += cap;
		D.rec_base = (u32)recs; D.rec_cap = S.n_recs; recs += (u64)S.n_recs + 1; max_recs = std::max(max_recs, S.n_recs);
		D.d_base = dbytes; dbytes += al(cap / 2 + 64, 64);
		D.node_cap = S.tag_nodes + 160 + 2 * DEC_STACK_SLACK; D.node_off = nodes; nodes += D.node_cap;
		// trees of the quality stream: one per position (Plain/Truncated) or 2 x alphabet (RLE); the DNA tree reuses the pool
		const u64 qn = std::min<u64>(std::max<u64>(((u64)S.max_qlen + 2) * 255, prm.quality_order == 0 ? 2ull * 256 * 256 : 64), (u64)D.in_size * 3) + 1024 + 2 * DEC_STACK_SLACK;
		D.qnode_cap = (u32)qn; D.qnode_off = nodes; nodes += qn;
		D.fld_off = fbytes; fbytes += al(sizeof(DecField) * std::max(1u, S.n_fields), 16);
		if (recs >= (1ull << 32)) return fail(h, DSRCGPU_E_ARG, "batch too large for 32-bit record indices; submit fewer blocks per batch");
	}
	RecPools rp;
	rp.title_off = AP<u32>(h, A.alloc(recs * 4)); rp.seq_off = AP<u32>(h, A.alloc(recs * 4)); rp.qual_off = AP<u32>(h, A.alloc(recs * 4));
	rp.title_len = AP<u16>(h, A.alloc(recs * 2)); rp.len = AP<u16>(h, A.alloc(recs * 2));
	rp.kept = AP<u16>(h, A.alloc(recs * 2)); rp.trunc = nullptr;
	rp.q_off = nullptr; rp.d_off = AP<u32>(h, A.alloc(recs * 4));
	const size_t o_d = A.alloc(dbytes + 64), o_nodes = A.alloc(nodes * 4 + 64), o_fld = A.alloc(fbytes + 64);
	u8* d_out = io.d_out;
	size_t o_out = 0;
	if (!d_out) o_out = A.alloc(text_total + 64);
	if (A.failed) return fail(h, DSRCGPU_E_NOMEM, "arena exhausted (decode): batch needs > %zu bytes of HBM scratch", A.top);
	if (!d_out)
	{
		d_out = AP<u8>(h, o_out);
		if (io.host_out && text_total > io.host_cap) return fail(h, DSRCGPU_E_CAPACITY, "output needs %llu bytes, caller gave %llu", (unsigned long long)text_total, (unsigned long long)io.host_cap);
	}
	else if (text_total > io.out_cap) return fail(h, DSRCGPU_E_CAPACITY, "output needs %llu bytes, caller gave %llu", (unsigned long long)text_total, (unsigned long long)io.out_cap);

	HIPCHK(hipMemcpyAsync(d_desc, desc.data(), sizeof(DecDesc) * B, hipMemcpyHostToDevice, s));
	if (prm.serial_quality) { hipLaunchKernelGGL(k_dec_tags, dim3(B), dim3(64), 0, s, io.d_in, d_desc, d_state, rp, d_out, AP<u32>(h, o_nodes), AP<u8>(h, o_fld), prm); KCHK(); }
	else { hipLaunchKernelGGL(k_dec_tags_wave, dim3(B), dim3(64), 0, s, io.d_in, d_desc, d_state, rp, d_out, AP<u32>(h, o_nodes), AP<u8>(h, o_fld), prm); KCHK(); }
	if (!q_rc)
	{	// -q0: the position schemes through per-position 6-bit tables (k_dec_q0.h), whatever that does not cover (RLE scheme, long reads,
		// large alphabets) bit by bit (k_dec_qhuff, which skips the blocks the first one has done)
		if (!prm.serial_quality) { hipLaunchKernelGGL(k_dec_qpos, dim3(B), dim3(64), 0, s, io.d_in, d_desc, d_state, rp, d_out, AP<u32>(h, o_nodes), prm); KCHK(); }
		hipLaunchKernelGGL(k_dec_qhuff, dim3(B), dim3(64), 0, s, io.d_in, d_desc, d_state, rp, d_out, AP<u32>(h, o_nodes), prm); KCHK();
	}

	if (prm.serial_quality)
	{	// the one-lane decoder: a slot of the worst-case size per wave, the waves loop over the blocks
		prm.table_words = (u32)dec_table_words(h->set);
		const u32 slots = (u32)std::max<u64>(1, std::min<u64>(B, dec_table_budget(h) / ((u64)prm.table_words * 4)));
		const int rc = ensure_dec_tables(h, (u64)slots * prm.table_words * 4 + 64);
		if (rc) return rc;
		hipLaunchKernelGGL(k_dec_streams, dim3(slots), dim3(64), 0, s, io.d_in, d_desc, d_state, rp, d_out, AP<u32>(h, o_nodes), AP<u8>(h, o_d), h->dec_tables, prm); KCHK();
	}
	else
	{
		// A stream advances one symbol per dependent row read, so the rate of the range-decoded levels is (chains in flight) /
		// latency: every block of the batch gets its own table, sized from the scheme byte of its stream
		const u64 budget_words = dec_table_budget(h) / 4;
		u64 region_words = 0;
		std::vector<DecTab> qtabs;
		if (q_rc)
		{
			HIPCHK(hipMemcpyAsync(st.data(), d_state, sizeof(DecState) * B, hipMemcpyDeviceToHost, s));
			HIPCHK(hipStreamSynchronize(s));
			mark("tags"); tr.stage("dsrc decode: quality");
			const int rc = check_blocks(false);
			if (rc) return rc;
			u64 sum = 0;
			for (u32 b = 0; b < B; ++b)
			{
				DecTab t; t.block = b; t.off = 0;
				t.words = q_table_words(prm.quality_order, prm.lossy != 0, st[b].q_scheme, st[b].q_cnt, &t.n);
				sum += (t.words + 31) & ~31ull;
				qtabs.push_back(t);
			}
			region_words = sum;
		}
		if (d_rc) region_words = std::max<u64>(region_words, (u64)B * ((d_table_words(prm.dna_order, 0) + 31) & ~31ull));
		u64 biggest = 0;
		for (const DecTab& t : qtabs) biggest = std::max<u64>(biggest, (t.words + 31) & ~31ull);
		region_words = std::max<u64>(std::min<u64>(region_words, budget_words), biggest);
		if (region_words) { const int rc = ensure_dec_tables(h, region_words * 4); if (rc) return rc; }
		region_words = h->dec_tables_cap / 4;
		auto fill_round = [&](const DecTab* d_tabs, const std::vector<DecTab>& tabs, const DecRound& r, hipStream_t on)
		{
			u64 mx = 0;
			for (u32 i = 0; i < r.count; ++i) mx = std::max(mx, tabs[r.first + i].words);
			const u32 gx = (u32)std::max<u64>(1, std::min<u64>(1024, mx / 4 / (256 * 4)));
			hipLaunchKernelGGL(k_dec_fill, dim3(gx, r.count), dim3(256), 0, on, h->dec_tables, d_tabs + r.first);
		};
		// Verification of blocks just written (io.hints): the DNA chains start at once, on the range-coder stream, next to the quality
		// chains -- if the tables of both fit the region together (one round each); otherwise one stage after the other as for archives.
		bool par = false;
		std::vector<DecTab> ptabs; u32 p4 = 0;
		if (io.hints && q_rc && d_rc && h->rc_stream)
		{
			u64 qsum = 0; for (const DecTab& t : qtabs) qsum += (t.words + 31) & ~31ull;
			for (u32 pass = 0; pass < 2; ++pass)
			{
				for (u32 b = 0; b < B; ++b)
					if (io.hints[b].d_scheme == pass) { DecTab t; t.block = b; t.off = 0; t.n = 0; t.words = d_table_words(prm.dna_order, pass); ptabs.push_back(t); }
				if (pass == 0) p4 = (u32)ptabs.size();
			}
			u64 dsum = 0; for (const DecTab& t : ptabs) dsum += (t.words + 31) & ~31ull;
			if (qsum + dsum <= budget_words && !ptabs.empty())
			{
				const int rc = ensure_dec_tables(h, (qsum + dsum) * 4);
				if (rc) return rc;
				region_words = h->dec_tables_cap / 4;
				u64 top = qsum;
				for (DecTab& t : ptabs) { t.off = top; top += (t.words + 31) & ~31ull; }
				par = true;
				DecTab* d_tabs = AP<DecTab>(h, o_dtabs); DecHint* d_hints = AP<DecHint>(h, o_hints);
				HIPCHK(hipMemcpyAsync(d_hints, io.hints, sizeof(DecHint) * B, hipMemcpyHostToDevice, s));
				HIPCHK(hipMemcpyAsync(d_tabs, ptabs.data(), sizeof(DecTab) * ptabs.size(), hipMemcpyHostToDevice, s));
				hipLaunchKernelGGL(k_dec_hint, dim3((B + 63) / 64), dim3(64), 0, s, d_state, d_hints, B); KCHK();
				HIPCHK(hipEventRecord(h->ev[2], s));
				HIPCHK(hipStreamWaitEvent(h->rc_stream, h->ev[2], 0));
				const DecRound r4{0, p4}, r8{p4, (u32)ptabs.size() - p4};
				if (r4.count)
				{
					fill_round(d_tabs, ptabs, r4, h->rc_stream); KCHK();
					hipLaunchKernelGGL(k_dec_dnarc<4>, dim3((r4.count + 63) / 64), dim3(64), 0, h->rc_stream, io.d_in, d_desc, d_state, d_tabs + r4.first, r4.count, AP<u8>(h, o_d), h->dec_tables, prm); KCHK();
				}
				if (r8.count)
				{
					fill_round(d_tabs, ptabs, r8, h->rc_stream); KCHK();
					hipLaunchKernelGGL(k_dec_dnarc<8>, dim3((r8.count + 63) / 64), dim3(64), 0, h->rc_stream, io.d_in, d_desc, d_state, d_tabs + r8.first, r8.count, AP<u8>(h, o_d), h->dec_tables, prm); KCHK();
				}
				HIPCHK(hipEventRecord(h->ev[3], h->rc_stream));
			}
		}
		if (q_rc)
		{
			const std::vector<DecRound> rounds = plan_rounds(qtabs, region_words);      // (with the DNA tables behind them: one round)
			DecTab* d_tabs = AP<DecTab>(h, o_qtabs);
			HIPCHK(hipMemcpyAsync(d_tabs, qtabs.data(), sizeof(DecTab) * qtabs.size(), hipMemcpyHostToDevice, s));
			for (const DecRound& r : rounds)
			{
				fill_round(d_tabs, qtabs, r, s); KCHK();
				// one launch per alphabet size the round contains
				u32 sizes_present = 0;
				for (u32 i = 0; i < r.count; ++i) sizes_present |= qtabs[r.first + i].n;          // 8, 16, 32, 64, 128: one bit each
				const DecTab* tp = d_tabs + r.first;
				if (sizes_present & 8u)   { hipLaunchKernelGGL(k_dec_qrc<8>, dim3(r.count), dim3(64), 0, s, io.d_in, d_desc, d_state, tp, rp, d_out, h->dec_tables, prm); KCHK(); }
				if (sizes_present & 16u)  { hipLaunchKernelGGL(k_dec_qrc<16>, dim3(r.count), dim3(64), 0, s, io.d_in, d_desc, d_state, tp, rp, d_out, h->dec_tables, prm); KCHK(); }
				if (sizes_present & 32u)  { hipLaunchKernelGGL(k_dec_qrc<32>, dim3(r.count), dim3(64), 0, s, io.d_in, d_desc, d_state, tp, rp, d_out, h->dec_tables, prm); KCHK(); }
				if (sizes_present & 64u)  { hipLaunchKernelGGL(k_dec_qrc<64>, dim3(r.count), dim3(64), 0, s, io.d_in, d_desc, d_state, tp, rp, d_out, h->dec_tables, prm); KCHK(); }
				if (sizes_present & 128u) { hipLaunchKernelGGL(k_dec_qrc<128>, dim3(r.count), dim3(64), 0, s, io.d_in, d_desc, d_state, tp, rp, d_out, h->dec_tables, prm); KCHK(); }
			}
		}
		if (par) HIPCHK(hipStreamWaitEvent(s, h->ev[3], 0));
		hipLaunchKernelGGL(k_dec_dhead, dim3((B + 63) / 64), dim3(64), 0, s, io.d_in, d_desc, d_state, prm, par ? (const DecHint*)AP<DecHint>(h, o_hints) : (const DecHint*)nullptr); KCHK();
		bool any_plain = !d_rc;
		if (par) { for (u32 b = 0; b < B; ++b) if (io.hints[b].d_scheme == 255) any_plain = true; }
		if (d_rc && !par)
		{
			HIPCHK(hipMemcpyAsync(st.data(), d_state, sizeof(DecState) * B, hipMemcpyDeviceToHost, s));
			HIPCHK(hipStreamSynchronize(s));
			mark("quality"); tr.stage("dsrc decode: DNA, layout");
			const int rc = check_blocks(false);
			if (rc) return rc;
			std::vector<DecTab> dtabs;
			u32 n4 = 0;
			for (u32 pass = 0; pass < 2; ++pass)                      // the 4-symbol blocks first, then the 8-symbol ones
			{
				for (u32 b = 0; b < B; ++b)
				{
					if (st[b].d_scheme == 255) { any_plain = true; continue; }
					if (st[b].d_scheme != pass) continue;
					DecTab t; t.block = b; t.off = 0; t.n = 0; t.words = d_table_words(prm.dna_order, pass);
					dtabs.push_back(t);
				}
				if (pass == 0) n4 = (u32)dtabs.size();
			}
			if (n4 < dtabs.size())
			{	// 8-symbol tables are larger than what was reserved from the 4-symbol size
				const u64 w8 = (d_table_words(prm.dna_order, 1) + 31) & ~31ull;
				const u64 want = std::max<u64>(std::min<u64>((u64)(dtabs.size() - n4) * w8, budget_words), w8);
				if (want > region_words) { const int rc2 = ensure_dec_tables(h, want * 4); if (rc2) return rc2; region_words = h->dec_tables_cap / 4; }
			}
			DecTab* d_tabs = AP<DecTab>(h, o_dtabs);
			std::vector<DecTab> t4(dtabs.begin(), dtabs.begin() + n4), t8(dtabs.begin() + n4, dtabs.end());
			const std::vector<DecRound> r4 = plan_rounds(t4, region_words), r8 = plan_rounds(t8, region_words);
			std::copy(t4.begin(), t4.end(), dtabs.begin()); std::copy(t8.begin(), t8.end(), dtabs.begin() + n4);
			if (!dtabs.empty()) HIPCHK(hipMemcpyAsync(d_tabs, dtabs.data(), sizeof(DecTab) * dtabs.size(), hipMemcpyHostToDevice, s));
			for (const DecRound& r : r4)
			{
				fill_round(d_tabs, dtabs, r, s); KCHK();
				hipLaunchKernelGGL(k_dec_dnarc<4>, dim3((r.count + 63) / 64), dim3(64), 0, s, io.d_in, d_desc, d_state, d_tabs + r.first, r.count, AP<u8>(h, o_d), h->dec_tables, prm); KCHK();
			}
			for (const DecRound& r8r : r8)
			{
				const DecRound r{r8r.first + n4, r8r.count};
				fill_round(d_tabs, dtabs, r, s); KCHK();
				hipLaunchKernelGGL(k_dec_dnarc<8>, dim3((r.count + 63) / 64), dim3(64), 0, s, io.d_in, d_desc, d_state, d_tabs + r.first, r.count, AP<u8>(h, o_d), h->dec_tables, prm); KCHK();
			}
		}
		if (any_plain) { hipLaunchKernelGGL(k_dec_dna0, dim3(B), dim3(64), 0, s, io.d_in, d_desc, d_state, AP<u32>(h, o_nodes), AP<u8>(h, o_d), prm); KCHK(); }
	}
	{
		const u32 gx = std::max(1u, std::min(64u, (max_recs + 4 * WAVES - 1) / (4 * WAVES)));
		hipLaunchKernelGGL(k_dec_layout, dim3(gx, B), dim3(WG), 0, s, d_desc, d_state, rp, d_out, AP<u8>(h, o_d), prm); KCHK();
	}
	if (prm.crc && io.crc_ok) { hipLaunchKernelGGL(k_dec_crc, dim3(B, 3), dim3(WG), 0, s, d_desc, d_state, rp, d_out, h->d_crc_tab, prm); KCHK(); }
	HIPCHK(hipMemcpyAsync(st.data(), d_state, sizeof(DecState) * B, hipMemcpyDeviceToHost, s));
	HIPCHK(hipEventRecord(h->ev[1], s));
	if (trace) { HIPCHK(hipStreamSynchronize(s)); mark("dna+layout"); }
	if (io.host_out) HIPCHK(hipMemcpyAsync(io.host_out, d_out, text_total, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	mark("copied");
	if (trace) fprintf(stderr, "[dsrc_gpu] %p decode timeline (%u blocks):%s\n", (void*)h, B, tl.c_str());
	{
		const int rc = check_blocks(true);
		if (rc) return rc;
	}
	for (u32 b = 0; b < B; ++b)
	{
		const DecState& S = st[b];
		io.out_offs[b] = desc[b].out_off; io.out_sizes[b] = S.text_bytes;
		if (io.crc_ok)
		{
			bool ok = true;
			if (prm.crc)
			{
				if (!prm.tag_flags) ok &= S.crc_stored[0] == S.crc_actual[0];
				ok &= S.crc_stored[1] == S.crc_actual[1];
				if (!prm.lossy) ok &= S.crc_stored[2] == S.crc_actual[2];
			}
			io.crc_ok[b] = ok ? 1u : 0u;
		}
	}
	hipEventElapsedTime(&h->batch_ms, h->ev[0], h->ev[1]);
	h->rc_ms = 0.f; h->rc_launches = 0;
	return DSRCGPU_OK;
}

size_t estimate_decode_arena(const dsrcgpu_handle* h, u32 n, const u64* sizes, bool own_text)
{
	(void)h;
	// measured at -d3 -q2: ~1.9 x the blocks for record tables, base stream and tree pools, + 3.2 x for the text when it is not the
	// caller's; a pass that needs more (highly compressible data) is re-run with what it asked for (with_arena_retry_)
	size_t tot = 0;
	for (u32 i = 0; i < n; ++i) tot += (size_t)sizes[i] + 4096;
	return tot * (own_text ? 7 : 3) + (size_t)n * (1u << 20) + (16u << 20);
}

// The reference's compressing worker decodes every block it has just written and compares the checksums
// (DsrcCompressor::Process, src/DsrcWorker.cpp:53-62).  Here: the blocks still sit in HBM; they are decoded there and only
// the verdicts come back.
int verify_blocks(dsrcgpu_handle* h, u32 n, const u64* offs, const u64* sizes)
{
	std::vector<u64> to(n), ts(n); std::vector<u32> ok(n, 0);
	DecodeIO io{h->last_d_out, offs, sizes, n, nullptr, nullptr, 0, nullptr, 0, to.data(), ts.data(), ok.data()};
	if (h->verify_hints.size() == n && !hook_env("DSRC_GPU_VERIFY_SERIAL")) io.hints = h->verify_hints.data();
	// dsrcgpu_last_timing keeps reporting the compression batch: the verifying pass has its own figure
	const float c_batch = h->batch_ms, c_rc = h->rc_ms; const u32 c_launches = h->rc_launches;
	const int rc = run_decode(h, io);
	h->verify_ms = h->batch_ms; h->batch_ms = c_batch; h->rc_ms = c_rc; h->rc_launches = c_launches;
	if (rc != DSRCGPU_OK)
	{
		if (rc == DSRCGPU_E_NOMEM) return rc;                 // the batch is re-run with a larger arena
		std::string why; { std::lock_guard<std::mutex> g(h->err_m); why = h->err; }
		return fail(h, DSRCGPU_E_CRC, "CRC32 checksums mismatch. (%s)", why.c_str());
	}
	for (u32 i = 0; i < n; ++i)
		if (!ok[i]) return fail(h, DSRCGPU_E_CRC, "CRC32 checksums mismatch.");
	return DSRCGPU_OK;
}

int check_settings(dsrcgpu_handle* h, const dsrcgpu_settings* s, const dsrcgpu_dataset* d)
{
	if (!s || !d) return fail(h, DSRCGPU_E_ARG, "null settings/dataset");
	if (s->tag_preserve_flags & ~0x7FFFFFFEull) return fail(h, DSRCGPU_E_ARG, "tag field filter (-f): field numbers 1..30 only (the reference shifts a 32-bit int)");
	if (d->quality_offset < 33 || d->quality_offset > 64) return fail(h, DSRCGPU_E_ARG, "quality offset %u outside [33,64]", d->quality_offset);
	if (s->dna_order > 9) return fail(h, DSRCGPU_E_ARG, "dna_order %u > 9", s->dna_order);
	// lossless: 0, 1, 2 (the command line's -q levels) and 3, 6 -- what wrap::DsrcArchive makes of levels 1 and 2 (qualityOrder = 3 * level,
	// src/DsrcArchive.cpp:42): the reference's lossless proxy treats every order but 1 as order 2 and takes the "F" schemes at order 2
	// only (src/QualityModelerProxy.h:225-283); the order byte goes into the archive's footer as it is
	if (!s->lossy && s->quality_order > 2 && s->quality_order != 3 && s->quality_order != 6) return fail(h, DSRCGPU_E_ARG, "lossless quality_order %u (0, 1, 2, or the record API's 3 and 6)", s->quality_order);
	if (s->lossy && s->quality_order > 6) return fail(h, DSRCGPU_E_ARG, "lossy quality_order %u > 6", s->quality_order);
	return 0;
}

// A batch call made by the USER (not by a scheduler lane of the queue form): takes the pending record layout -- it belongs to this call,
// one-shot -- and, on a handle whose queue form runs two lanes, checks that the queue has drained and notes that the handle's own copy
// of the block-to-block state is about to move (the next flush hands it to the lanes' chain).
int user_batch_begin(dsrcgpu_handle* h, std::vector<u32>& layout)
{
	std::lock_guard<std::mutex> g(h->q_m);
	if (h->q_started && h->q_pending) return fail(h, DSRCGPU_E_STATE, "batches of the queue form are still in flight on this handle");
	if (h->q_chain)
	{
		h->chain = nullptr;          // the queue is drained: no lane touches the handle until the next flush
		h->q_user_batch = true;
	}
	layout.swap(h->rec_pending); h->rec_pending.clear();
	return DSRCGPU_OK;
}

struct LanesIO
{
	const u8* d_in; const uint8_t* const* host_in;            // one of the two
	const u64* offs; const u64* sizes; u32 n;
	u8* d_out; u8* host_out; u64 cap;                        // one of the two
	u64* out_offs; u64* out_sizes; u64* raw; u64* comp;
	const std::vector<u32>* layout;
};
u32 lanes_sub_chunks(const dsrcgpu_handle* h, u32 n, const u64* sizes, u32* lanes_out);
int run_lanes(dsrcgpu_handle* h, const LanesIO& io, u32 sub, u32 lanes);
int compress_batch_host(dsrcgpu_handle* h, uint32_t n, const uint8_t* const* fastq, const uint64_t* sizes,
						uint8_t* blocks, uint64_t blocks_cap, uint64_t* block_offs, uint64_t* block_sizes,
						uint64_t* raw_sizes, uint64_t* comp_sizes, const std::vector<u32>& layout);

} // namespace

extern "C" {

int dsrcgpu_create(const dsrcgpu_settings* settings, const dsrcgpu_dataset* dataset, int device, uint64_t arena_bytes, dsrcgpu_handle** out)
{
	if (!out) return DSRCGPU_E_ARG;
	*out = nullptr;
	dsrcgpu_handle* h = new dsrcgpu_handle();
	*out = h;                       // returned even on failure so that the caller can read last_error
	int rc = check_settings(h, settings, dataset);
	if (rc) return rc;
	h->set = *settings; h->ds = *dataset; h->device = device;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(h, DSRCGPU_E_HIP, "no HIP device available: the DSRC GPU path requires an MI355X-class GPU (there is no CPU fallback)");
	if (device < 0 || device >= ndev) return fail(h, DSRCGPU_E_ARG, "device %d out of range (%d devices)", device, ndev);
	HIPCHK(hipSetDevice(device));
	{
		// non-blocking: a host framework's work on the legacy default stream (torch, RCCL bookkeeping) must not
		// serialise with the scheduler's streams.  (CUs reserved for k_rc through a CU mask were measured in rounds 2 and 4: no gain.)
		HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
		// k_rc has a stream of its own, of the SAME priority as the front end's (round 5).  Rounds 1-4 gave it the highest: its few waves
		// were not to queue behind other instances' data-parallel kernels.  Since the coder wave has its SIMD to itself and eight loader
		// waves feed it, k_rc takes its 118 ms whatever runs beside it -- and the priority only held the front ends back: 4 x 450 blocks
		// 52.3 - 54.8 GB/s with the high-priority stream, 55.1 - 56.7 without (three runs each, profiles/r05_rcprio.txt; the same
		// through a CU-mask stream over all CUs, and confining k_rc to 16 / 32 / 64 CUs of an instance's own changed nothing beyond that).
		HIPCHK(hipStreamCreateWithFlags(&h->rc_stream, hipStreamNonBlocking));
	}
	for (int i = 0; i < 5; ++i) HIPCHK(hipEventCreate(&h->ev[i]));
	h->arena_fixed = arena_bytes;
	{	// CRC tables: 256 byte-table entries + x^(2^k) mod P, k = 0..31
		u32 tab[288];
		for (u32 i = 0; i < 256; ++i) { u32 c = i; for (int j = 0; j < 8; ++j) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; tab[i] = c; }
		auto mul = [](u32 a, u32 b) { u32 m = 1u << 31, p = 0; for (;;) { if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; } m >>= 1; b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1; } return p; };
		u32 p = 1u << 30; tab[256] = p;
		for (u32 k = 1; k < 32; ++k) { p = mul(p, p); tab[256 + k] = p; }
		HIPCHK(hipMalloc((void**)&h->d_crc_tab, sizeof(tab)));
		HIPCHK(hipMemcpy(h->d_crc_tab, tab, sizeof(tab), hipMemcpyHostToDevice));
	}
	{	// which ranking variant of k_sort this device gets (once per device and process): the LDS-atomic one needs the lanes of
		// one LDS atomic instruction applied in lane order, which is measured here, not assumed.  DSRC_GPU_SORT_BALLOT=1 keeps
		// the ballot variant (tests compare the two).
		static std::mutex mu; static int known[64];               // 0 unknown, 1 ordered, 2 not
		std::lock_guard<std::mutex> lk(mu);
		int& k = known[device & 63];
		if (k == 0)
		{
			u32* d_bad = nullptr; u32 bad = 1;
			HIPCHK(hipMalloc((void**)&d_bad, 4));
			HIPCHK(hipMemsetAsync(d_bad, 0, 4, h->stream));
			#ifdef DSRC_EMU_BUILD
			hipLaunchKernelGGL(k_lds_order_test, dim3(1), dim3(256), 0, h->stream, d_bad, 20u); KCHK();       // the CPU emulator runs lanes in order anyway
			hipLaunchKernelGGL(k_lds_order_test64, dim3(1), dim3(256), 0, h->stream, d_bad, 20u); KCHK();
#else
			hipLaunchKernelGGL(k_lds_order_test, dim3(256), dim3(256), 0, h->stream, d_bad, 4096u); KCHK();   // 4 M wave patterns, < 1 ms
			hipLaunchKernelGGL(k_lds_order_test64, dim3(256), dim3(256), 0, h->stream, d_bad, 4096u); KCHK();
#endif
			HIPCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, h->stream));
			HIPCHK(hipStreamSynchronize(h->stream));
			HIPCHK(hipFree(d_bad));
			k = 1 + (int)bad;                                        // bit 0: 32-bit atomics out of order, bit 1: 64-bit ones
			if ((bad & 1u) && getenv("DSRC_GPU_DEBUG")) fprintf(stderr, "[dsrc_gpu] LDS atomics are not applied in lane order on device %d: k_sort ranks with ballots\n", device);
			if ((bad & 2u) && getenv("DSRC_GPU_DEBUG")) fprintf(stderr, "[dsrc_gpu] 64-bit LDS atomics are not applied in lane order on device %d: no bucketed path (k_model)\n", device);
		}
		h->sort_atomic = !((k - 1) & 1) && !hook_env("DSRC_GPU_SORT_BALLOT");
		h->lds64_ordered = !((k - 1) & 2) && h->sort_atomic;
	}
	if (arena_bytes) { rc = ensure_arena(h, (size_t)arena_bytes); if (rc) return rc; }
	return DSRCGPU_OK;
}

void dsrcgpu_destroy(dsrcgpu_handle* h)
{
	if (!h) return;
	(void)hipSetDevice(h->device);
	if (h->q_started)
	{
		{ std::lock_guard<std::mutex> g(h->q_m); h->q_stop = true; h->q_cv.notify_all(); }
		h->q_thread.join();
		for (std::thread& t : h->q_threads2) if (t.joinable()) t.join();
	}
	for (dsrcgpu_handle* c : h->subs) dsrcgpu_destroy(c);
	h->subs.clear();
	if (h->sub_chain) { dsrcgpu_chain_destroy(h->sub_chain); h->sub_chain = nullptr; }
	for (dsrcgpu_handle* t : h->twins) dsrcgpu_destroy(t);
	h->twins.clear();
	if (h->q_chain) { dsrcgpu_chain_destroy(h->q_chain); h->q_chain = nullptr; }
	for (QBatch& b : h->qb) { if (b.in) hipHostFree(b.in); if (b.out) hipHostFree(b.out); }
	if (h->arena.base) hipFree(h->arena.base);
	if (h->dec_tables) hipFree(h->dec_tables);
	if (h->d_crc_tab) hipFree(h->d_crc_tab);
	for (int i = 0; i < 5; ++i) if (h->ev[i]) hipEventDestroy(h->ev[i]);
	for (hipEvent_t e : h->stage_ev) hipEventDestroy(e);
	if (h->rc_stream) hipStreamDestroy(h->rc_stream);
	if (h->stream) hipStreamDestroy(h->stream);
	delete h;
}

const char* dsrcgpu_last_error(const dsrcgpu_handle* h)
{
	if (!h) return "null handle";
	// a copy per calling thread: the scheduler thread of the queue form may fail while the caller is reading
	static thread_local std::string copy;
	{ std::lock_guard<std::mutex> g(h->err_m); copy = h->err; }
	return copy.c_str();
}

int dsrcgpu_compress_batch_device(dsrcgpu_handle* h, uint32_t n, const void* d_fastq, const uint64_t* offs, const uint64_t* sizes,
								  void* d_blocks, uint64_t blocks_cap, uint64_t* block_offs, uint64_t* block_sizes,
								  uint64_t* raw_sizes, uint64_t* comp_sizes)
{
	if (!h) return DSRCGPU_E_ARG;
	if (!d_fastq || !offs || !sizes || !d_blocks || !block_offs || !block_sizes || !raw_sizes || !comp_sizes) return fail(h, DSRCGPU_E_ARG, "null argument");
	std::vector<u32> layout;
	{ const int rc = user_batch_begin(h, layout); if (rc) return rc; }
	HIPCHK(hipSetDevice(h->device));
	{
		u32 lanes = 0;
		if (const u32 sub = lanes_sub_chunks(h, n, sizes, &lanes))
			return run_lanes(h, LanesIO{(const u8*)d_fastq, nullptr, offs, sizes, n, (u8*)d_blocks, nullptr, blocks_cap, block_offs, block_sizes, raw_sizes, comp_sizes, &layout}, sub, lanes);
	}
	return with_arena_retry(h, estimate_arena(h, n, sizes), [&]() {
		BatchIO io{(const u8*)d_fastq, offs, sizes, n, (u8*)d_blocks, blocks_cap, nullptr, 0, block_offs, block_sizes, raw_sizes, comp_sizes, &layout, true};
		const int rc = run_batch(h, io);
		if (rc != DSRCGPU_OK || !h->set.verify_after_compress || !h->set.calculate_crc32) return rc;
		return verify_blocks(h, n, block_offs, block_sizes);
	});
}

int dsrcgpu_compress_batch(dsrcgpu_handle* h, uint32_t n, const uint8_t* const* fastq, const uint64_t* sizes,
						   uint8_t* blocks, uint64_t blocks_cap, uint64_t* block_offs, uint64_t* block_sizes,
						   uint64_t* raw_sizes, uint64_t* comp_sizes)
{
	if (!h) return DSRCGPU_E_ARG;
	if (!fastq || !sizes || !blocks || !block_offs || !block_sizes || !raw_sizes || !comp_sizes) return fail(h, DSRCGPU_E_ARG, "null argument");
	if (n == 0) return DSRCGPU_OK;
	std::vector<u32> layout;
	{ const int rc = user_batch_begin(h, layout); if (rc) return rc; }
	return compress_batch_host(h, n, fastq, sizes, blocks, blocks_cap, block_offs, block_sizes, raw_sizes, comp_sizes, layout);
}

} // extern "C"

namespace
{
// host-resident chunks -> blocks: the body of dsrcgpu_compress_batch, also what a scheduler lane of the queue form runs (with the
// record layout its batch was flushed with)
int compress_batch_host(dsrcgpu_handle* h, uint32_t n, const uint8_t* const* fastq, const uint64_t* sizes,
						uint8_t* blocks, uint64_t blocks_cap, uint64_t* block_offs, uint64_t* block_sizes,
						uint64_t* raw_sizes, uint64_t* comp_sizes, const std::vector<u32>& layout)
{
	HIPCHK(hipSetDevice(h->device));
	{
		u32 lanes = 0;
		if (const u32 sub = lanes_sub_chunks(h, n, sizes, &lanes))
			return run_lanes(h, LanesIO{nullptr, fastq, nullptr, sizes, n, nullptr, blocks, blocks_cap, block_offs, block_sizes, raw_sizes, comp_sizes, &layout}, sub, lanes);
	}
	std::vector<u64> offs(n);
	size_t in_bytes = 0;
	for (u32 i = 0; i < n; ++i) { offs[i] = in_bytes; in_bytes += al((size_t)sizes[i] + 16, 256); }
	return with_arena_retry(h, estimate_arena(h, n, sizes) + in_bytes, [&]() {
		const size_t o_in = h->arena.alloc(in_bytes + 256);
		if (h->arena.failed) return fail(h, DSRCGPU_E_NOMEM, "arena exhausted (input)");
		u8* d_in = h->arena.base + o_in;
		for (u32 i = 0; i < n; ++i) HIPCHK(hipMemcpyAsync(d_in + offs[i], fastq[i], sizes[i], hipMemcpyHostToDevice, h->stream));
		BatchIO io{d_in, offs.data(), sizes, n, nullptr, 0, blocks, blocks_cap, block_offs, block_sizes, raw_sizes, comp_sizes, &layout};
		const int rc = run_batch(h, io);
		if (rc != DSRCGPU_OK || !h->set.verify_after_compress || !h->set.calculate_crc32) return rc;
		return verify_blocks(h, n, block_offs, block_sizes);
	});
}

// ---- scheduler lanes inside one handle ----------------------------------------------------------------------------------------------
// The reference needs ONE DsrcCompressorMT for a file; until round 5 the GPU path needed four handles (and their HBM) to hide the
// serial range coder.  Here a batch call of n chunks is cut into K = ceil(n / sub_chunks) consecutive sub-batches; `lanes` threads take them
// in order, each on a child handle of its own (arena sized for a sub-batch); TagStats::fields' capacity goes from sub-batch k to k + 1
// through a chain, so the blocks are those of one handle fed in order.  A sub-batch assembles its blocks in its own arena; when every
// earlier sub-batch has put its blocks down it copies them behind theirs (device to device, or down to the host).
#define DSRC_LANES_DEFAULT 4
#define DSRC_SUB_BYTES_DEFAULT ((size_t)1900000000)      // (225 chunks of 8 MiB: a call of 1800 is two rounds of four lanes -- 44 GB/s; three rounds of 150: 36)

// how a batch would be cut: 0 = not at all (run_batch on the handle itself)
u32 lanes_sub_chunks(const dsrcgpu_handle* h, u32 n, const u64* sizes, u32* lanes_out)
{
	if (h->is_sub || h->chain || h->arena_fixed || tl_queue_lane || h->lanes_want == 1 || n < 2) return 0;
	size_t tot = 0; for (u32 i = 0; i < n; ++i) tot += (size_t)sizes[i];
	u32 sub = h->sub_chunks_want;
	if (!sub) sub = (u32)std::max<size_t>(1, DSRC_SUB_BYTES_DEFAULT / std::max<size_t>(1, tot / n));
	u32 K = (n + sub - 1) / sub;
	if (K < 2) return 0;
	// equal sub-batches, whole rounds of the lanes (nine sub-batches on four lanes leave three lanes idle for a third of the call:
	// 37 against 44 GB/s for calls of 1800 chunks), unless the caller named the size
	const u32 lanes = h->lanes_want ? h->lanes_want : (u32)DSRC_LANES_DEFAULT;
	if (!h->sub_chunks_want && K > lanes) K = std::min(n, (K + lanes - 1) / lanes * lanes);
	sub = (n + K - 1) / K;
	*lanes_out = std::min(lanes, (n + sub - 1) / sub);
	return sub;
}

int run_lanes(dsrcgpu_handle* h, const LanesIO& io, u32 sub, u32 lanes)
{
	const u32 n = io.n;
	// the sub-batches (equal ones: with the first of every lane of growing size, so that the lanes' front ends and range coders do not
	// all run at once, 4 x 225 blocks went from 43.5 to 39.1 GB/s)
	std::vector<u32> cut(1, 0);
	while (cut.back() < n) cut.push_back(std::min(n, cut.back() + sub));
	const u32 K = (u32)cut.size() - 1;
	while (h->subs.size() < lanes)
	{
		dsrcgpu_handle* c = nullptr;
		const int rc = dsrcgpu_create(&h->set, &h->ds, h->device, 0, &c);
		if (rc) { const std::string why = c ? dsrcgpu_last_error(c) : "out of memory"; if (c) dsrcgpu_destroy(c); return fail(h, rc, "scheduler lane: %s", why.c_str()); }
		c->is_sub = true; c->dec_table_budget = h->dec_table_budget;
		h->subs.push_back(c);
	}
	if (!h->sub_chain && dsrcgpu_chain_create(&h->sub_chain) != DSRCGPU_OK) return fail(h, DSRCGPU_E_NOMEM, "out of memory");
	{ std::lock_guard<std::mutex> g(h->sub_chain->m); h->sub_chain->next_seq = 0; h->sub_chain->fields_cap = h->fields_cap; h->sub_chain->failed = false; }
	struct Shared
	{
		std::mutex m; std::condition_variable cv;
		u32 next = 0, next_commit = 0; u64 base = 0;
		int rc = DSRCGPU_OK; std::string err;
	} sh;
	const auto t0 = std::chrono::steady_clock::now();
	std::vector<float> rc_ms(lanes, 0.f); std::vector<u32> rc_n(lanes, 0);
	const bool verify_all = h->set.verify_after_compress && h->set.calculate_crc32 && io.d_out != nullptr;
	std::vector<DecHint> hints(verify_all ? n : 0);
	auto work = [&](u32 li)
	{
		dsrcgpu_handle* c = h->subs[li];
		(void)hipSetDevice(h->device);
		for (;;)
		{
			u32 k;
			{ std::lock_guard<std::mutex> g(sh.m); if (sh.rc || sh.next >= K) return; k = sh.next++; }
			const u32 lo = cut[k], hi = cut[k + 1], nk = hi - lo;
			std::vector<u32> lay;
			if (io.layout && !io.layout->empty()) lay.assign(io.layout->begin() + lo, io.layout->begin() + hi);
			(void)dsrcgpu_set_chain(c, h->sub_chain, k);
			std::vector<u64> offs_k(nk);
			size_t in_bytes = 0;
			if (io.host_in) for (u32 i = 0; i < nk; ++i) { offs_k[i] = in_bytes; in_bytes += al((size_t)io.sizes[lo + i] + 16, 256); }
			int rc = with_arena_retry(c, estimate_arena(c, nk, io.sizes + lo) + in_bytes, [&]() -> int {
				const u8* d_in = io.d_in; const u64* offs = io.offs + lo;
				if (io.host_in)
				{
					const size_t o_in = c->arena.alloc(in_bytes + 256);
					if (c->arena.failed) return fail(c, DSRCGPU_E_NOMEM, "arena exhausted (input)");
					u8* p = c->arena.base + o_in;
					for (u32 i = 0; i < nk; ++i) if (hipMemcpyAsync(p + offs_k[i], io.host_in[lo + i], io.sizes[lo + i], hipMemcpyHostToDevice, c->stream) != hipSuccess) return fail(c, DSRCGPU_E_HIP, "copy of a chunk to the device failed");
					d_in = p; offs = offs_k.data();
				}
				BatchIO b{d_in, offs, io.sizes + lo, nk, nullptr, 0, nullptr, ~0ull, io.out_offs + lo, io.out_sizes + lo, io.raw + 4 * (size_t)lo, io.comp + 4 * (size_t)lo, &lay, io.host_in == nullptr};
				const int r = run_batch(c, b);
				// -c: with device-resident output ONE verifying pass over all the call's blocks follows (below) -- a decoding pass is a chain
				// per block and takes about as long for 2000 blocks as for 200; blocks that go down to the host are verified here, while
				// they are still on the device
				if (r != DSRCGPU_OK || !c->set.verify_after_compress || !c->set.calculate_crc32 || verify_all) return r;
				return verify_blocks(c, nk, io.out_offs + lo, io.out_sizes + lo);
			});
			if (!rc && verify_all) std::copy(c->verify_hints.begin(), c->verify_hints.end(), hints.begin() + lo);
			u64 total = 0;
			if (!rc) { total = io.out_offs[hi - 1] + io.out_sizes[hi - 1]; rc_ms[li] += c->rc_ms; ++rc_n[li]; }
			// the sub-batch's blocks go behind those of the sub-batches before it
			std::unique_lock<std::mutex> g(sh.m);
			sh.cv.wait(g, [&] { return sh.rc || sh.next_commit == k; });
			if (!rc && !sh.rc)
			{
				if (sh.base + total > io.cap) rc = fail(c, DSRCGPU_E_CAPACITY, "output needs more than %llu bytes, caller gave %llu", (unsigned long long)(sh.base + total), (unsigned long long)io.cap);
				else
				{
					const u64 base = sh.base;
					g.unlock();
					hipError_t e = io.d_out ? hipMemcpyAsync(io.d_out + base, c->last_d_out, total, hipMemcpyDeviceToDevice, c->stream)
											: hipMemcpyAsync(io.host_out + base, c->last_d_out, total, hipMemcpyDeviceToHost, c->stream);
					if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
					if (e != hipSuccess) rc = fail(c, DSRCGPU_E_HIP, "copy of a sub-batch's blocks failed: %s", hipGetErrorString(e));
					for (u32 i = lo; i < hi; ++i) io.out_offs[i] += base;
					g.lock();
					sh.base = base + total;
				}
			}
			if (rc && !sh.rc) { sh.rc = rc; std::lock_guard<std::mutex> ge(c->err_m); sh.err = c->err; }
			if (sh.rc)
			{	// nobody must wait for a sub-batch that will not come
				std::lock_guard<std::mutex> gc(h->sub_chain->m); h->sub_chain->failed = true; h->sub_chain->cv.notify_all();
			}
			sh.next_commit = k + 1;
			sh.cv.notify_all();
			if (sh.rc) return;
		}
	};
	std::vector<std::thread> th;
	for (u32 li = 1; li < lanes; ++li) th.emplace_back(work, li);
	work(0);
	for (std::thread& t : th) t.join();
	if (sh.rc) return fail(h, sh.rc, "%s", sh.err.c_str());
	const float compress_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
	if (verify_all)
	{	// the blocks are where the caller wanted them: decoded from there, all in one pass, on the handle's own lane
		h->last_d_out = io.d_out; h->verify_hints.swap(hints);
		const u32 saved = h->fields_cap;
		const int rc = with_arena_retry_(h, estimate_decode_arena(h, n, io.out_sizes, true), [&]() { return verify_blocks(h, n, io.out_offs, io.out_sizes); });
		h->fields_cap = saved;
		if (rc) return rc;
	}
	{ std::lock_guard<std::mutex> g(h->sub_chain->m); h->fields_cap = h->sub_chain->fields_cap; }
	h->batch_ms = compress_ms;
	float s_ms = 0.f; u32 s_n = 0; for (u32 li = 0; li < lanes; ++li) { s_ms += rc_ms[li]; s_n += rc_n[li]; }
	h->rc_ms = s_n ? s_ms / s_n : 0.f; h->rc_launches = K;
	h->sort_ms = h->replay_ms = 0.f;
	h->last_d_out = io.d_out;
	return DSRCGPU_OK;
}
} // namespace

extern "C" {

int dsrcgpu_set_lanes(dsrcgpu_handle* h, uint32_t lanes, uint32_t sub_batch_chunks)
{
	if (!h) return DSRCGPU_E_ARG;
	if (lanes > 16) return fail(h, DSRCGPU_E_ARG, "at most 16 scheduler lanes per handle");
	h->lanes_want = lanes; h->sub_chunks_want = sub_batch_chunks;
	return DSRCGPU_OK;
}

int dsrcgpu_decompress_batch_device(dsrcgpu_handle* h, uint32_t n, const void* d_blocks, const uint64_t* offs, const uint64_t* sizes,
									const uint64_t* text_caps, void* d_text, uint64_t text_cap, uint64_t* text_offs, uint64_t* text_sizes, uint32_t* crc_ok)
{
	if (!h) return DSRCGPU_E_ARG;
	if (!d_blocks || !offs || !sizes || !d_text || !text_offs || !text_sizes) return fail(h, DSRCGPU_E_ARG, "null argument");
	HIPCHK(hipSetDevice(h->device));
	return with_arena_retry_(h, estimate_decode_arena(h, n, sizes, false), [&]() {
		DecodeIO io{(const u8*)d_blocks, offs, sizes, n, text_caps, (u8*)d_text, text_cap, nullptr, 0, text_offs, text_sizes, crc_ok};
		return run_decode(h, io);
	});
}

int dsrcgpu_decompress_batch(dsrcgpu_handle* h, uint32_t n, const uint8_t* const* blocks, const uint64_t* sizes, const uint64_t* text_caps,
							 uint8_t* text, uint64_t text_cap, uint64_t* text_offs, uint64_t* text_sizes, uint32_t* crc_ok)
{
	if (!h) return DSRCGPU_E_ARG;
	if (!blocks || !sizes || !text || !text_offs || !text_sizes) return fail(h, DSRCGPU_E_ARG, "null argument");
	if (n == 0) return DSRCGPU_OK;
	HIPCHK(hipSetDevice(h->device));
	std::vector<u64> offs(n);
	size_t in_bytes = 0;
	for (u32 i = 0; i < n; ++i) { offs[i] = in_bytes; in_bytes += al((size_t)sizes[i] + 16, 64); }
	return with_arena_retry_(h, estimate_decode_arena(h, n, sizes, true) + in_bytes, [&]() {
		const size_t o_in = h->arena.alloc(in_bytes + 256);
		if (h->arena.failed) return fail(h, DSRCGPU_E_NOMEM, "arena exhausted (input)");
		u8* d_in = h->arena.base + o_in;
		for (u32 i = 0; i < n; ++i) HIPCHK(hipMemcpyAsync(d_in + offs[i], blocks[i], sizes[i], hipMemcpyHostToDevice, h->stream));
		DecodeIO io{d_in, offs.data(), sizes, n, text_caps, nullptr, 0, text, text_cap, text_offs, text_sizes, crc_ok};
		return run_decode(h, io);
	});
}

int dsrcgpu_decompress_block(dsrcgpu_handle* h, const uint8_t* block, uint64_t size, uint8_t* text, uint64_t text_cap, uint64_t* text_size, uint32_t* crc_ok)
{
	if (!h) return DSRCGPU_E_ARG;
	if (!text_size) return fail(h, DSRCGPU_E_ARG, "null argument");
	u64 off = 0;
	const uint8_t* ins[1] = {block};
	return dsrcgpu_decompress_batch(h, 1, ins, &size, nullptr, text, text_cap, &off, text_size, crc_ok);
}

int dsrcgpu_set_record_layout(dsrcgpu_handle* h, uint32_t n, const uint32_t* chunk_sizes)
{
	if (!h) return DSRCGPU_E_ARG;
	if (n && !chunk_sizes) return fail(h, DSRCGPU_E_ARG, "null argument");
	std::lock_guard<std::mutex> g(h->q_m);
	h->rec_pending.assign(chunk_sizes, chunk_sizes + n);
	return DSRCGPU_OK;
}

int dsrcgpu_compress_block(dsrcgpu_handle* h, const uint8_t* fastq, uint64_t size, uint8_t* block, uint64_t block_cap, uint64_t* block_size,
						   uint64_t raw_sizes[4], uint64_t comp_sizes[4])
{
	if (!h) return DSRCGPU_E_ARG;
	if (!block_size) return fail(h, DSRCGPU_E_ARG, "null argument");
	u64 off = 0;
	const uint8_t* ins[1] = {fastq};
	return dsrcgpu_compress_batch(h, 1, ins, &size, block, block_cap, &off, block_size, raw_sizes, comp_sizes);
}

namespace
{
int pinned_grow(u8*& p, u64& cap, u64 used, u64 need)
{
	if (need <= cap) return DSRCGPU_OK;
	u64 want = std::max<u64>(need + need / 4, (u64)64 << 20);
	void* q = nullptr;
	if (hipHostMalloc(&q, want, hipHostMallocPortable) != hipSuccess) return DSRCGPU_E_NOMEM;
	if (p) { if (used) memcpy(q, p, used); hipHostFree(p); }
	p = (u8*)q; cap = want;
	return DSRCGPU_OK;
}

// the handle's scheduler thread: runs queued batches in order through the synchronous batch entry point
void queue_thread(dsrcgpu_handle* h, int lane)
{
	dsrcgpu_handle* L = lane ? h->twins[lane - 1] : h;         // the lane's arena, streams, error text
	(void)hipSetDevice(h->device);
	// (test hook, timing only: lane 0 starts late, so that the twin takes the first batches)
	if (lane == 0) if (const long late_ms = hook_int("DSRC_GPU_HOOK_LANE0_DELAY_MS", 0)) std::this_thread::sleep_for(std::chrono::milliseconds(late_ms));
	for (;;)
	{
		u32 k;
		{
			std::unique_lock<std::mutex> g(h->q_m);
			h->q_cv.wait(g, [&] { return h->q_stop || !h->q_run.empty(); });
			if (h->q_run.empty()) return;
			k = h->q_run.front(); h->q_run.pop_front();
		}
		QBatch& b = h->qb[k];
		const u32 n = (u32)b.ids.size();
		std::vector<const uint8_t*> ptrs(n);
		for (u32 i = 0; i < n; ++i) ptrs[i] = b.ext[i] ? b.ext[i] : b.in + b.in_off[i];
		b.o_offs.assign(n, 0); b.o_sizes.assign(n, 0); b.raw.assign(4 * n, 0); b.comp.assign(4 * n, 0);
		u64 in_sum = 0; for (u32 i = 0; i < n; ++i) in_sum += b.sizes[i];
		u64 cap = in_sum * 2 / 5 + (u64)n * (1u << 16);          // typical ratio 0.2-0.33; the worst case once if that is short
		int rc = DSRCGPU_OK;
		for (int attempt = 0; attempt < 2; ++attempt)
		{
			rc = pinned_grow(b.out, b.out_cap, 0, cap);
			if (rc) { fail(L, rc, "cannot allocate page-locked output memory"); break; }
			if (h->q_chain) (void)dsrcgpu_set_chain(L, h->q_chain, b.seq);      // (a retry keeps the turn it has taken)
			tl_queue_lane = h->q_chain ? 2 : 1;
			rc = compress_batch_host(L, n, ptrs.data(), b.sizes.data(), b.out, b.out_cap, b.o_offs.data(), b.o_sizes.data(), b.raw.data(), b.comp.data(), b.layout);
			tl_queue_lane = 0;
			if (rc != DSRCGPU_E_CAPACITY) break;
			cap = in_sum + (u64)n * (1u << 16);
		}
		std::lock_guard<std::mutex> g(h->q_m);
		if (h->q_chain && !rc)
		{	// the handle's own view of the carried state follows the chain (dsrcgpu_get_fields_capacity after a drain)
			std::lock_guard<std::mutex> gc(h->q_chain->m);
			if (h->q_chain->next_seq == b.seq + 1) h->fields_cap = h->q_chain->fields_cap;
		}
		b.rc = rc; if (rc) { { std::lock_guard<std::mutex> g2(L->err_m); b.err = L->err; } if (!h->q_rc) { h->q_rc = rc; h->q_err = b.err; } }
		b.state = QBatch::Done; b.next_collect = 0; b.outstanding = 0;
		h->q_cv.notify_all();
	}
}

int queue_collect(dsrcgpu_handle* h, bool wait, int64_t* part_id, uint8_t** block, uint64_t* block_size, uint64_t raw_sizes[4], uint64_t comp_sizes[4])
{
	std::unique_lock<std::mutex> g(h->q_m);
	for (;;)
	{
		// the batch to collect from: the oldest flushed one that still has blocks to hand out
		while (h->q_pending)
		{
			const QBatch& c = h->qb[h->q_collect];
			const bool spent = c.state == QBatch::Free || (c.state == QBatch::Done && !c.rc && c.next_collect == c.ids.size());
			if (!spent) break;
			h->q_collect = (h->q_collect + 1) % h->q_depth; --h->q_pending;
		}
		if (!h->q_pending) return 0;                          // nothing flushed that has not been collected
		QBatch& b = h->qb[h->q_collect];
		if (b.state == QBatch::Queued)
		{
			if (!wait) return 0;
			h->q_cv.wait(g, [&] { return b.state != QBatch::Queued; });
			continue;
		}
		if (b.rc) return fail(h, b.rc, "%s", b.err.c_str());
		const u32 i = b.next_collect++;
		++b.outstanding;
		*part_id = b.ids[i]; *block = b.out + b.o_offs[i]; *block_size = b.o_sizes[i];
		for (int k = 0; k < 4; ++k) { if (raw_sizes) raw_sizes[k] = b.raw[4 * i + k]; if (comp_sizes) comp_sizes[k] = b.comp[4 * i + k]; }
		return 1;
	}
}
} // namespace

static int queue_submit(dsrcgpu_handle* h, int64_t part_id, const uint8_t* fastq, uint64_t size, bool pinned)
{
	if (!h) return DSRCGPU_E_ARG;
	if (!fastq || size == 0) return fail(h, DSRCGPU_E_ARG, "empty chunk");
	std::unique_lock<std::mutex> g(h->q_m);
	if (h->q_rc) return fail(h, h->q_rc, "%s", h->q_err.c_str());
	QBatch* b = &h->qb[h->q_fill];
	if (b->state != QBatch::Filling)
	{	// a ring slot is free again once every block of the batch it held has been collected and released.  A caller that
		// submits and collects on ONE thread would wait for itself here, so a full ring is reported, not waited for
		if (b->state != QBatch::Free) return DSRCGPU_E_BUSY;
		b->state = QBatch::Filling; b->ids.clear(); b->in_off.clear(); b->sizes.clear(); b->ext.clear(); b->in_used = 0;
	}
	if (pinned)
	{	// the chunk stays where it is: the batch's copy to the device reads the caller's page-locked memory
		b->ids.push_back(part_id); b->in_off.push_back(0); b->sizes.push_back(size); b->ext.push_back(fastq);
		return DSRCGPU_OK;
	}
	const u64 at = (b->in_used + 255) & ~(u64)255;
	if (pinned_grow(b->in, b->in_cap, b->in_used, at + size + 256) != DSRCGPU_OK) return fail(h, DSRCGPU_E_NOMEM, "cannot allocate page-locked staging memory");
	g.unlock();
	memcpy(b->in + at, fastq, size);                          // only the submitter touches a Filling batch
	b->ids.push_back(part_id); b->in_off.push_back(at); b->sizes.push_back(size); b->ext.push_back(nullptr); b->in_used = at + size;
	return DSRCGPU_OK;
}

int dsrcgpu_submit(dsrcgpu_handle* h, int64_t part_id, const uint8_t* fastq, uint64_t size) { return queue_submit(h, part_id, fastq, size, false); }
int dsrcgpu_submit_pinned(dsrcgpu_handle* h, int64_t part_id, const uint8_t* fastq, uint64_t size) { return queue_submit(h, part_id, fastq, size, true); }

int dsrcgpu_flush(dsrcgpu_handle* h)
{
	if (!h) return DSRCGPU_E_ARG;
	std::unique_lock<std::mutex> g(h->q_m);
	if (h->q_rc) return fail(h, h->q_rc, "%s", h->q_err.c_str());
	QBatch& b = h->qb[h->q_fill];
	if (b.state != QBatch::Filling || b.ids.empty()) return DSRCGPU_OK;
	if (!h->q_lanes_decided)
	{	// second lane unless the caller hands the state over himself (dsrcgpu_set_chain) or DSRC_GPU_QUEUE_LANES=1
		h->q_lanes_decided = true;
		const bool want = !h->chain && !(getenv("DSRC_GPU_QUEUE_LANES") && atoi(getenv("DSRC_GPU_QUEUE_LANES")) <= 1);
		u32 lanes = DSRC_QUEUE_LANES_DEFAULT;
		if (const char* e = getenv("DSRC_GPU_QUEUE_LANES")) lanes = (u32)std::max(1, std::min(DSRC_QUEUE_LANES_MAX, atoi(e)));
		if (want && lanes > 1 && dsrcgpu_chain_create(&h->q_chain) == DSRCGPU_OK)
		{
			for (u32 k = 1; k < lanes; ++k)
			{
				dsrcgpu_handle* t = nullptr;
				if (dsrcgpu_create(&h->set, &h->ds, h->device, 0, &t) == DSRCGPU_OK) { t->is_sub = true; h->twins.push_back(t); }
				else { if (t) dsrcgpu_destroy(t); break; }
			}
			if (!h->twins.empty()) { (void)dsrcgpu_chain_seed(h->q_chain, h->fields_cap); h->q_user_batch = false; h->q_depth = std::min<u32>(DSRC_QUEUE_DEPTH, (u32)h->twins.size() + 3); }
			else { dsrcgpu_chain_destroy(h->q_chain); h->q_chain = nullptr; }
		}
	}
	if (!h->q_started)
	{
		h->q_started = true;
		h->q_thread = std::thread(queue_thread, h, 0);
		for (u32 k = 0; k < h->twins.size(); ++k) h->q_threads2.emplace_back(queue_thread, h, (int)k + 1);
	}
	if (h->q_chain && h->q_user_batch)
	{	// the user made batch calls on the handle since the last flush (only possible with the queue drained, and noted by those calls
		// themselves under q_m -- which lane ran which batch says nothing about it): the lanes go on from the handle's state
		std::lock_guard<std::mutex> gc(h->q_chain->m);
		h->q_chain->fields_cap = h->fields_cap;
		h->q_user_batch = false;
	}
	b.seq = h->q_seq++;
	b.layout.swap(h->rec_pending); h->rec_pending.clear();
	b.state = QBatch::Queued; ++h->q_pending;
	h->q_run.push_back(h->q_fill);
	h->q_fill = (h->q_fill + 1) % h->q_depth;
	h->q_cv.notify_all();
	return DSRCGPU_OK;
}

int dsrcgpu_collect(dsrcgpu_handle* h, int64_t* part_id, uint8_t** block, uint64_t* block_size, uint64_t raw_sizes[4], uint64_t comp_sizes[4])
{
	if (!h) return DSRCGPU_E_ARG;
	if (!part_id || !block || !block_size) return fail(h, DSRCGPU_E_ARG, "null argument");
	return queue_collect(h, true, part_id, block, block_size, raw_sizes, comp_sizes);
}

int dsrcgpu_try_collect(dsrcgpu_handle* h, int64_t* part_id, uint8_t** block, uint64_t* block_size, uint64_t raw_sizes[4], uint64_t comp_sizes[4])
{
	if (!h) return DSRCGPU_E_ARG;
	if (!part_id || !block || !block_size) return fail(h, DSRCGPU_E_ARG, "null argument");
	return queue_collect(h, false, part_id, block, block_size, raw_sizes, comp_sizes);
}

int dsrcgpu_release(dsrcgpu_handle* h, uint8_t* block)
{
	if (!h) return DSRCGPU_E_ARG;
	std::lock_guard<std::mutex> g(h->q_m);
	for (QBatch& b : h->qb)
		if (b.state == QBatch::Done && b.out && block >= b.out && block < b.out + b.out_cap && b.outstanding)
		{
			if (--b.outstanding == 0 && b.next_collect == b.ids.size()) { b.state = QBatch::Free; h->q_cv.notify_all(); }
			return DSRCGPU_OK;
		}
	return fail(h, DSRCGPU_E_ARG, "dsrcgpu_release: not a block handed out by dsrcgpu_collect");
}

int dsrcgpu_selftest(dsrcgpu_handle* h, uint32_t* mismatches)
{
	if (!h || !mismatches) return DSRCGPU_E_ARG;
	HIPCHK(hipSetDevice(h->device));
	u32* d_bad = nullptr;
	HIPCHK(hipMalloc((void**)&d_bad, 4));
	HIPCHK(hipMemsetAsync(d_bad, 0, 4, h->stream));
	hipLaunchKernelGGL(k_selftest, dim3(256), dim3(256), 0, h->stream, d_bad); KCHK();
	hipLaunchKernelGGL(k_selftest_dec, dim3(256), dim3(256), 0, h->stream, d_bad); KCHK();      // the decoder's division (k_dec_rc.h)
	{	// the two-wave range coder against the reference's loop on states at the carry clamp (k_rc.h); a run in which the reference
		// never clamped has tested nothing and counts as a mismatch
		u32* d_hits = nullptr; u32 hits = 0;
		HIPCHK(hipMalloc((void**)&d_hits, 4));
		HIPCHK(hipMemsetAsync(d_hits, 0, 4, h->stream));
#ifdef DSRC_EMU_BUILD
		hipLaunchKernelGGL(k_selftest_rcs, dim3(4), dim3(256), 0, h->stream, d_bad, d_hits); KCHK();
#else
		hipLaunchKernelGGL(k_selftest_rcs, dim3(1024), dim3(256), 0, h->stream, d_bad, d_hits); KCHK();
#endif
		u8* d_scr = nullptr;
#ifdef DSRC_EMU_BUILD
		const u32 rcv_wgs = 2;
#else
		const u32 rcv_wgs = 256;
#endif
		HIPCHK(hipMalloc((void**)&d_scr, (size_t)rcv_wgs * 64 * 1024));
		hipLaunchKernelGGL(k_selftest_rcv, dim3(rcv_wgs), dim3(64), 0, h->stream, d_bad, d_hits, d_scr); KCHK();
		HIPCHK(hipMemcpyAsync(&hits, d_hits, 4, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		HIPCHK(hipFree(d_hits)); HIPCHK(hipFree(d_scr));
		if (getenv("DSRC_GPU_DEBUG")) fprintf(stderr, "[dsrc_gpu] selftest: the reference's carry clamp fired in %u of the split coder's test groups\n", hits);
		if (hits < 16) HIPCHK(hipMemsetAsync(d_bad, 0xFF, 4, h->stream));
	}
	{	// the property k_sort's atomic ranking stands on (dsrcgpu_create runs the same test to choose the variant); a violation
		// counts as a mismatch here so that the test-suite notices a device on which the ballot variant is in use
		u32* d_ord = nullptr; u32 ord = 0;
		HIPCHK(hipMalloc((void**)&d_ord, 4));
		HIPCHK(hipMemsetAsync(d_ord, 0, 4, h->stream));
#ifdef DSRC_EMU_BUILD
		hipLaunchKernelGGL(k_lds_order_test, dim3(1), dim3(256), 0, h->stream, d_ord, 20u); KCHK();
#else
		hipLaunchKernelGGL(k_lds_order_test, dim3(256), dim3(256), 0, h->stream, d_ord, 4096u); KCHK();
#endif
		HIPCHK(hipMemcpyAsync(&ord, d_ord, 4, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		HIPCHK(hipFree(d_ord));
		if (ord) HIPCHK(hipMemsetAsync(d_bad, 0xFF, 4, h->stream));
	}
	HIPCHK(hipMemcpyAsync(mismatches, d_bad, 4, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	HIPCHK(hipFree(d_bad));
	return DSRCGPU_OK;
}

int dsrcgpu_chain_create(dsrcgpu_chain** out)
{
	if (!out) return DSRCGPU_E_ARG;
	*out = new (std::nothrow) dsrcgpu_chain();
	return *out ? DSRCGPU_OK : DSRCGPU_E_NOMEM;
}

void dsrcgpu_chain_destroy(dsrcgpu_chain* c) { delete c; }

int dsrcgpu_set_chain(dsrcgpu_handle* h, dsrcgpu_chain* c, uint64_t seq)
{
	if (!h) return DSRCGPU_E_ARG;
	// Announcing the same batch again -- the caller retries it after DSRCGPU_E_CAPACITY with a larger output buffer --
	// must not take a second turn: the first attempt has already read the state of batch seq - 1 and published its
	// own (later batches may have advanced the chain since), so the retry re-uses what it read then.
	if (c && h->chain == c && h->chain_seq == seq && h->chain_taken && !h->chain_batch_done) return DSRCGPU_OK;
	h->chain = c; h->chain_seq = seq; h->chain_taken = false; h->chain_batch_done = false;
	return DSRCGPU_OK;
}

int dsrcgpu_chain_seed(dsrcgpu_chain* c, uint32_t fields_capacity)
{
	if (!c) return DSRCGPU_E_ARG;
	std::lock_guard<std::mutex> g(c->m);
	if (c->next_seq != 0) return DSRCGPU_E_STATE;
	c->fields_cap = fields_capacity;
	return DSRCGPU_OK;
}

// (under q_m: with the queue form a lane that completes a batch writes the handle's view of the state)
int dsrcgpu_set_fields_capacity(dsrcgpu_handle* h, uint32_t cap)
{
	if (!h) return DSRCGPU_E_ARG;
	std::lock_guard<std::mutex> g(h->q_m);
	// with batches in flight (one lane or two) there is no point in the archive the value could belong to, and a lane is writing the field
	if (h->q_started && h->q_pending) return fail(h, DSRCGPU_E_STATE, "batches of the queue form are still in flight on this handle");
	if (h->q_chain) h->q_user_batch = true;        // two lanes: between flushes with the queue drained the lanes go on from this value
	h->fields_cap = cap;
	return DSRCGPU_OK;
}
int dsrcgpu_get_fields_capacity(const dsrcgpu_handle* h, uint32_t* cap)
{
	if (!h || !cap) return DSRCGPU_E_ARG;
	std::lock_guard<std::mutex> g(const_cast<dsrcgpu_handle*>(h)->q_m);
	*cap = h->fields_cap;
	return DSRCGPU_OK;
}

// TagAnalyzer::InitializeFieldsStats' field split (src/TagModeler.cpp:159-222) on the title the compressor will see,
// i.e. after FastqParserExt's rewrite when a field filter is set (src/FastqParser.cpp:198-251)
uint32_t dsrcgpu_title_fields(const uint8_t* title, uint32_t len, uint64_t tag_preserve_flags)
{
	auto is_sep = [](uint8_t c) { return c == ' ' || c == '.' || c == '_' || c == ',' || c == '=' || c == ':' || c == '/' || c == '-' || c == '#' || c == 0; };
	uint32_t seps = 0;
	for (uint32_t i = 0; i < len; ++i) seps += is_sep(title[i]) ? 1u : 0u;
	if (!tag_preserve_flags) return seps + 1;
	// kept fields are copied with their end byte; every kept field but the title's last one ends in a separator, and the
	// tokenizer adds the field after the last separator
	uint32_t kept = 0;
	for (uint32_t k = 1; k <= seps && k < 31; ++k) kept += (tag_preserve_flags >> k) & 1u;
	return kept + 1;
}

uint32_t dsrcgpu_fields_capacity_after(uint32_t cap, uint32_t n_fields)
{
	for (uint32_t i = 0; i < n_fields; ++i) if (i == cap) cap = cap ? cap * 2 : 1;
	return cap;
}

int dsrcgpu_set_table_budget(dsrcgpu_handle* h, uint64_t bytes) { if (!h) return DSRCGPU_E_ARG; h->dec_table_budget = bytes; return DSRCGPU_OK; }

int dsrcgpu_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes)
{
	size_t f = 0, t = 0;
	if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) return DSRCGPU_E_HIP;
	if (free_bytes) *free_bytes = f;
	if (total_bytes) *total_bytes = t;
	return DSRCGPU_OK;
}

// Every scheduler instance drives two HIP streams, and the HIP runtime multiplexes streams onto 4 hardware queues unless
// GPU_MAX_HW_QUEUES says otherwise: unrelated instances then wait behind each other's range coder (7.7 instead of 12.9 GB/s,
// DESIGN section 3).  The runtime reads the variable when it starts, i.e. at the process's first HIP call.  The library does NOT
// touch the environment on its own when it is loaded (an embedding application's queue configuration is its own business):
// dsrcgpu_prepare is the opt-in -- a host that calls it before its first HIP call (dsrc-amd, pydsrc, bench.py do) gets 24 queues
// unless the process has already chosen a value; everybody else exports the variable themselves.

int dsrcgpu_prepare(int device)
{
	{	// once per process, never over a value the process has chosen; hosts call this from several threads (one per device)
		static std::once_flag once;
		std::call_once(once, []() { if (!getenv("GPU_MAX_HW_QUEUES")) setenv("GPU_MAX_HW_QUEUES", "24", 0); });
	}
	// first touch of a device: the HIP runtime loads the code objects and creates the context (0.3-1 s); hosts call this
	// on a side thread while they open files, so that dsrcgpu_create finds the device ready
	if (hipSetDevice(device) != hipSuccess) return DSRCGPU_E_HIP;
	return hipFree(nullptr) == hipSuccess ? DSRCGPU_OK : DSRCGPU_E_HIP;
}

int dsrcgpu_host_alloc(uint64_t bytes, void** out)
{
	if (!out) return DSRCGPU_E_ARG;
	*out = nullptr;
	// portable: the same buffer feeds scheduler instances on any device of the node
	return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocPortable) == hipSuccess ? DSRCGPU_OK : DSRCGPU_E_NOMEM;
}

int dsrcgpu_release_memory(dsrcgpu_handle* h)
{
	if (!h) return DSRCGPU_E_ARG;
	HIPCHK(hipSetDevice(h->device));
	HIPCHK(hipStreamSynchronize(h->stream));
	if (h->rc_stream) HIPCHK(hipStreamSynchronize(h->rc_stream));
	if (h->arena.base) { HIPCHK(hipFree(h->arena.base)); h->arena.base = nullptr; h->arena.cap = 0; h->arena.top = 0; }
	if (h->dec_tables) { HIPCHK(hipFree(h->dec_tables)); h->dec_tables = nullptr; h->dec_tables_cap = 0; }
	h->last_d_out = nullptr;
	for (dsrcgpu_handle* c : h->subs) { const int rc = dsrcgpu_release_memory(c); if (rc) return rc; }
	for (dsrcgpu_handle* t : h->twins) { const int rc = dsrcgpu_release_memory(t); if (rc) return rc; }
	return DSRCGPU_OK;
}

int dsrcgpu_reserve_memory(dsrcgpu_handle* h, uint64_t arena_bytes, uint64_t table_bytes)
{
	if (!h) return DSRCGPU_E_ARG;
	HIPCHK(hipSetDevice(h->device));
	if (arena_bytes > h->arena.cap) { const int rc = ensure_arena(h, (size_t)arena_bytes); if (rc) return rc; }
	if (table_bytes) { const int rc = ensure_dec_tables(h, std::min<u64>(table_bytes, dec_table_budget(h))); if (rc) return rc; }
	return DSRCGPU_OK;
}

int dsrcgpu_host_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? DSRCGPU_OK : DSRCGPU_E_HIP; }

int dsrcgpu_last_timing(const dsrcgpu_handle* h, float* batch_ms, float* rc_ms, uint32_t* rc_launches)
{
	if (!h) return DSRCGPU_E_ARG;
	if (batch_ms) *batch_ms = h->batch_ms;
	if (rc_ms) *rc_ms = h->rc_ms;
	if (rc_launches) *rc_launches = h->rc_launches;
	return DSRCGPU_OK;
}

int dsrcgpu_last_stage_timing(const dsrcgpu_handle* h, float* sort_ms, float* replay_ms)
{
	if (!h) return DSRCGPU_E_ARG;
	if (sort_ms) *sort_ms = h->sort_ms;
	if (replay_ms) *replay_ms = h->replay_ms;
	return DSRCGPU_OK;
}

int dsrcgpu_synth_illumina(dsrcgpu_handle* h, uint64_t first, uint64_t count, void* d_out, uint64_t cap, uint64_t* bytes)
{
	if (!h) return DSRCGPU_E_ARG;
	if (!d_out || !bytes) return fail(h, DSRCGPU_E_ARG, "null argument");
	HIPCHK(hipSetDevice(h->device));
	return synth_illumina_device(h->stream, first, count, (u8*)d_out, cap, bytes, 0u) ? fail(h, DSRCGPU_E_CAPACITY, "synthetic FASTQ does not fit in %llu bytes", (unsigned long long)cap) : DSRCGPU_OK;
}

int dsrcgpu_synth_fastq(dsrcgpu_handle* h, uint32_t flavour, uint64_t first, uint64_t count, void* d_out, uint64_t cap, uint64_t* bytes)
{
	if (!h) return DSRCGPU_E_ARG;
	if (!d_out || !bytes) return fail(h, DSRCGPU_E_ARG, "null argument");
	if (flavour > 1) return fail(h, DSRCGPU_E_ARG, "unknown synthetic flavour %u", flavour);
	HIPCHK(hipSetDevice(h->device));
	return synth_illumina_device(h->stream, first, count, (u8*)d_out, cap, bytes, flavour) ? fail(h, DSRCGPU_E_CAPACITY, "synthetic FASTQ does not fit in %llu bytes", (unsigned long long)cap) : DSRCGPU_OK;
}

int dsrcgpu_dev_alloc(dsrcgpu_handle* h, uint64_t bytes, void** d_ptr)
{
	if (!h || !d_ptr) return DSRCGPU_E_ARG;
	HIPCHK(hipSetDevice(h->device));
	hipError_t e = hipMalloc(d_ptr, bytes);
	if (e != hipSuccess) return fail(h, DSRCGPU_E_NOMEM, "hipMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
	return DSRCGPU_OK;
}
int dsrcgpu_dev_free(dsrcgpu_handle* h, void* d_ptr) { if (!h) return DSRCGPU_E_ARG; HIPCHK(hipSetDevice(h->device)); HIPCHK(hipFree(d_ptr)); return DSRCGPU_OK; }
int dsrcgpu_dev_upload(dsrcgpu_handle* h, void* d_dst, const void* src, uint64_t bytes) { if (!h) return DSRCGPU_E_ARG; HIPCHK(hipSetDevice(h->device)); HIPCHK(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice)); return DSRCGPU_OK; }
int dsrcgpu_dev_download(dsrcgpu_handle* h, void* dst, const void* d_src, uint64_t bytes) { if (!h) return DSRCGPU_E_ARG; HIPCHK(hipSetDevice(h->device)); HIPCHK(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost)); return DSRCGPU_OK; }

} // extern "C"
