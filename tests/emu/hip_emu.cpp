// TEST INFRASTRUCTURE ONLY -- see tests/emu/hip/hip_runtime.h.
// Runs each workgroup's threads as ucontext fibers, round-robin between barriers.
#include <hip/hip_runtime.h>
#include <ucontext.h>
#include <cstdint>
#include <vector>
#include <stdexcept>

EmuIdx threadIdx, blockIdx, blockDim, gridDim;
namespace smst { alignas(16) unsigned char smemRaw[160*1024]; }

// Context switches: a workgroup of 1024 fibers yields at every barrier, poll and s_sleep -- tens of millions of switches per test run.  glibc's
// swapcontext() makes a signal-mask system call per switch (the CPU suite spent 4 min 54 s of 8 min 23 s in the kernel); the switch below saves
// and restores the callee-saved registers and the stack pointer, nothing else (x86-64 System V).  The sanitizer build keeps swapcontext(), which
// AddressSanitizer intercepts and understands.
#if defined(__x86_64__) && !defined(__SANITIZE_ADDRESS__)
#define EMU_FAST_SWITCH 1
extern "C" void emuSwitch(void **saveSp, void *newSp);
asm(R"(
.text
.globl emuSwitch
.type emuSwitch,@function
emuSwitch:
	pushq %rbp
	pushq %rbx
	pushq %r12
	pushq %r13
	pushq %r14
	pushq %r15
	movq %rsp, (%rdi)
	movq %rsi, %rsp
	popq %r15
	popq %r14
	popq %r13
	popq %r12
	popq %rbx
	popq %rbp
	ret
.size emuSwitch, .-emuSwitch
)");
#endif

namespace {
struct Fiber {
#ifdef EMU_FAST_SWITCH
	void *sp = nullptr;
#else
	ucontext_t ctx;
#endif
	std::vector<unsigned char> stack;
	bool done = false;
	EmuIdx tid;
};
#ifdef EMU_FAST_SWITCH
void *schedulerSp = nullptr;
#else
ucontext_t schedulerCtx;
#endif
Fiber *current = nullptr;
const std::function<void()> *currentBody = nullptr;

void toScheduler() {
#ifdef EMU_FAST_SWITCH
	emuSwitch(&current->sp, schedulerSp);
#else
	swapcontext(&current->ctx, &schedulerCtx);
#endif
}
void fiberEntry() {
	(*currentBody)();
	current->done = true;
	toScheduler();
}
#ifdef EMU_FAST_SWITCH
extern "C" void emuFiberStart() { fiberEntry(); __builtin_trap(); } // (a finished fiber is never resumed)
void prepare(Fiber &f) {
	// the first switch into the fiber pops six registers and returns into emuFiberStart with the stack aligned as after a call
	uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack.data()) + f.stack.size()) & ~uintptr_t(15);
	void **sp = reinterpret_cast<void **>(top - 8);
	*--sp = reinterpret_cast<void *>(&emuFiberStart);
	for (int i = 0; i < 6; ++i) *--sp = nullptr;
	f.sp = sp;
}
void resume(Fiber &f) { emuSwitch(&schedulerSp, f.sp); }
#else
void prepare(Fiber &f) {
	getcontext(&f.ctx);
	f.ctx.uc_stack.ss_sp = f.stack.data();
	f.ctx.uc_stack.ss_size = f.stack.size();
	f.ctx.uc_link = &schedulerCtx;
	makecontext(&f.ctx, fiberEntry, 0);
}
void resume(Fiber &f) { swapcontext(&schedulerCtx, &f.ctx); }
#endif
}

void emuSyncThreads() {
	toScheduler();
}

void emuLaunch(dim3 grid, dim3 block, size_t ldsBytes, const std::function<void()> &body) {
	if (ldsBytes > sizeof(smst::smemRaw)) throw std::runtime_error("emu: LDS request too large");
	const unsigned nThreads = block.x*block.y*block.z;
	static std::vector<Fiber> fibers;
	if (fibers.size() < nThreads) fibers.resize(nThreads);
	for (unsigned i = 0; i < nThreads; ++i) if (fibers[i].stack.empty()) fibers[i].stack.resize(256*1024);
	currentBody = &body;
	gridDim = {grid.x, grid.y, grid.z};
	blockDim = {block.x, block.y, block.z};
	for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
		for (unsigned i = 0; i < nThreads; ++i) {
			Fiber &f = fibers[i];
			f.done = false;
			f.tid = {i%block.x, (i/block.x)%block.y, i/(block.x*block.y)};
			prepare(f);
		}
		bool anyAlive = true;
		while (anyAlive) {
			anyAlive = false;
			for (unsigned i = 0; i < nThreads; ++i) {
				Fiber &f = fibers[i];
				if (f.done) continue;
				current = &f;
				threadIdx = f.tid;
				blockIdx = {bx, by, bz};
				resume(f);
				if (!f.done) anyAlive = true;
			}
		}
	}
}

// ---- wave-level collectives (votes span the block and are only used by single-wave kernels): every lane deposits its value, yields once so that
// all lanes of the block have deposited, then reads.  Two alternating banks keep back-to-back collectives apart.
namespace {
float shflBankF[2][1024];
int shflBankI[2][1024];
bool anyBank[2][1024];
int collectiveSeq[1024];
}
static int laneIndex() { return (int)(threadIdx.x + blockDim.x*(threadIdx.y + blockDim.y*threadIdx.z)); }
bool emuAny(bool v) {
	const int me = laneIndex(), bank = collectiveSeq[me]++ & 1;
	anyBank[bank][me] = v;
	emuSyncThreads();
	bool r = false; // a wave-level vote: the 64 lanes of the caller's own wave (other waves may not be voting at all)
	const int n = (int)(blockDim.x*blockDim.y*blockDim.z), w0 = me & ~63;
	for (int i = w0; i < w0 + 64 && i < n; ++i) r = r || anyBank[bank][i];
	emuSyncThreads();
	return r;
}
float emuShflF(float v, int lane) {
	const int me = laneIndex(), bank = collectiveSeq[me]++ & 1;
	shflBankF[bank][me] = v;
	emuSyncThreads();
	float r = shflBankF[bank][(me & ~63) + (lane & 63)]; // source lane within the caller's own wave
	emuSyncThreads();
	return r;
}
int emuShflI(int v, int lane) {
	const int me = laneIndex(), bank = collectiveSeq[me]++ & 1;
	shflBankI[bank][me] = v;
	emuSyncThreads();
	int r = shflBankI[bank][(me & ~63) + (lane & 63)];
	emuSyncThreads();
	return r;
}

// DPP wave_shr:1 (the only control the product uses): lane i of a 64-lane wave receives lane i-1's value, lane 0 keeps
// `old`.  Lanes of a wave run consecutively in index order between yields, so lane i-1 has always executed the same
// call instance already: a per-lane ring indexed by the call counter needs no yield (at most 512 calls per lane happen
// between two yields of the recurrence wave).
namespace { int dppRing[1024][1024]; unsigned dppSeq[1024]; }
int emuDppShr1(int old, int v) {
	const int me = laneIndex();
	const unsigned seq = dppSeq[me]++ & 1023u;
	dppRing[me][seq] = v;
	return (me & 63) ? dppRing[me - 1][seq] : old;
}
