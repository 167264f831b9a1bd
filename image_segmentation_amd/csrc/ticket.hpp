// In-launch hand-off from the blocks of one reduction to the block that finishes it (instead of a second, tiny kernel):
// every block publishes its partials and draws a ticket; the block that draws the LAST one reads them all -- nobody waits.
// One 64-bit word per (slot, group): {generation : 32 | arrivals : 32}.  A launch takes the next slot of a ring (host side,
// round robin) together with a fresh generation number; an arriving block that finds another generation in the word starts
// the count over (compare-and-swap), so a word never has to be clean: counters left behind by a launch that aborted, or
// scribbled over, cannot keep a later launch from electing its finisher (rounds 2-3 reset the word by the finisher and relied
// on every launch completing).  The ring only has to be longer than the number of such launches in flight at once.
// Agent-scope release / acquire around a relaxed ticket (cdna_hip_programming.md, in-launch split-K reduction): correct
// wherever the blocks run.
#pragma once
#include "common.hpp"

constexpr int TICKET_SLOTS = 256, TICKET_GROUPS = 32;
struct TicketRef {
  unsigned long long* words;   // TICKET_GROUPS words of the slot (device memory), nullptr: no ticket array on this device
  unsigned gen;                // this launch's generation (never 0xffffffff)
};
TicketRef segk_ticket_slot();  // host: the next slot of the current device and a fresh generation (bn_pool.hip)

// true in every thread of the block that arrives LAST of `n` at `word` (the caller's global stores are published
// first; the last block may then read every other block's).  flag: one int of LDS nobody else touches across the call.
__device__ __forceinline__ bool last_arriver(unsigned long long* word, unsigned gen, unsigned n, volatile int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long old = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), nw;
    do {
      nw = ((unsigned)(old >> 32) == gen) ? old + 1ull : (((unsigned long long)gen << 32) | 1ull);
    } while (!__hip_atomic_compare_exchange_strong(word, &old, nw, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const int last = ((unsigned)nw == n) ? 1 : 0;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *flag = last;
  }
  __syncthreads();
  return *flag != 0;
}
