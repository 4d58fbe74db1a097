// TEST INFRASTRUCTURE ONLY -- scheduler of the HIP emulator (see emu_hip.h).
#include "emu_hip.h"
#include <mutex>

namespace emu
{
dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
Thread* g_cur = nullptr;
Ctx g_sched;
static const std::function<void()>* g_body = nullptr;
static const size_t kStack = 256 * 1024;

// Context switch, x86-64 System V: the callee-saved registers go on the stack that is left, the stack pointer into *from; the
// other stack's are popped and `ret` continues where that coroutine called emu_switch (or, the first time, in its trampoline).
// swapcontext() did the same and two rt_sigprocmask system calls per switch: a third of the test-suite's time.
extern "C" void emu_switch(Ctx* from, Ctx* to);
asm(R"(
	.text
	.hidden emu_switch
	.globl emu_switch
	.type emu_switch,@function
emu_switch:
	pushq %rbp
	pushq %rbx
	pushq %r12
	pushq %r13
	pushq %r14
	pushq %r15
	movq %rsp, (%rdi)
	movq (%rsi), %rsp
	popq %r15
	popq %r14
	popq %r13
	popq %r12
	popq %rbx
	popq %rbp
	ret
	.size emu_switch, .-emu_switch
)");

void yield_to_scheduler()
{
	Thread* t = g_cur;
	emu_switch(&t->ctx, &g_sched);
}

uint64_t wave_exchange(uint64_t v, uint64_t out[64])
{
	Thread* t = g_cur;
	t->deposit = v;
	t->state = 2;
	yield_to_scheduler();
	memcpy(out, t->gathered, sizeof(t->gathered));
	return t->active;
}

static void trampoline()
{
	(*g_body)();
	g_cur->state = 3;
	emu_switch(&g_cur->ctx, &g_sched);
	abort();                           // a finished coroutine is never resumed
}

static std::mutex g_launch_mutex;      // one kernel at a time: host threads driving several handles take turns

void launch(dim3 grid, dim3 block, const std::function<void()>& body)
{
	std::lock_guard<std::mutex> guard(g_launch_mutex);
	const unsigned nt = block.x * block.y * block.z;
	static std::vector<Thread> pool;
	if (pool.size() < nt)
	{
		pool.resize(nt);
		for (auto& t : pool)
			if (t.stack.empty()) t.stack.resize(kStack);
	}
	g_body = &body;
	g_blockDim = block; g_gridDim = grid;
	for (unsigned bz = 0; bz < grid.z; ++bz)
	for (unsigned by = 0; by < grid.y; ++by)
	for (unsigned bx = 0; bx < grid.x; ++bx)
	{
		for (unsigned i = 0; i < nt; ++i)
		{
			Thread& t = pool[i];
			// a fresh stack: six zeroed callee-saved registers, the trampoline as the address emu_switch returns to, and a slot
			// that stands for the trampoline's own return address (so that its frame is aligned as after a call)
			uintptr_t top = ((uintptr_t)t.stack.data() + t.stack.size()) & ~(uintptr_t)15;
			void** sp = (void**)top;
			*--sp = nullptr;
			*--sp = (void*)&trampoline;
			for (int r = 0; r < 6; ++r) *--sp = nullptr;
			t.ctx.sp = sp;
			t.state = 0;
		}
		unsigned done = 0;
		while (done < nt)
		{
			bool progressed = false;
			for (unsigned i = 0; i < nt; ++i)
			{
				Thread& t = pool[i];
				if (t.state != 0) continue;
				g_cur = &t;
				g_threadIdx = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
				g_blockIdx = dim3(bx, by, bz);
				emu_switch(&g_sched, &t.ctx);
				progressed = true;
				if (t.state == 3) done++;
			}
			// release complete wave exchanges
			for (unsigned w = 0; w * 64 < nt; ++w)
			{
				unsigned lo = w * 64, hi = std::min(nt, lo + 64);
				unsigned live = 0, at = 0;
				for (unsigned i = lo; i < hi; ++i) { if (pool[i].state != 3) live++; if (pool[i].state == 2) at++; }
				if (at && at == live)
				{
					uint64_t vals[64] = {0}; uint64_t act = 0;
					for (unsigned i = lo; i < hi; ++i) if (pool[i].state == 2) { vals[i - lo] = pool[i].deposit; act |= 1ull << (i - lo); }
					for (unsigned i = lo; i < hi; ++i) if (pool[i].state == 2) { memcpy(pool[i].gathered, vals, sizeof(vals)); pool[i].active = act; pool[i].state = 0; }
					progressed = true;
				}
			}
			// release the barrier
			unsigned live = 0, atb = 0;
			for (unsigned i = 0; i < nt; ++i) { if (pool[i].state != 3) live++; if (pool[i].state == 1) atb++; }
			if (live && atb == live)
			{
				for (unsigned i = 0; i < nt; ++i) if (pool[i].state == 1) pool[i].state = 0;
				progressed = true;
			}
			if (!progressed && done < nt)
			{
				fprintf(stderr, "emu: deadlock in workgroup (%u,%u,%u): barrier/wave-op mismatch\n", bx, by, bz);
				for (unsigned i = 0; i < nt && i < 8; ++i) fprintf(stderr, "  thread %u state %d\n", i, pool[i].state);
				abort();
			}
		}
	}
	g_body = nullptr;
}
} // namespace emu
