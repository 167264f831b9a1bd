// In-launch hand-off from the blocks of one reduction to the block that finishes it (instead of a second, tiny kernel):
// every block publishes its partials and draws a ticket; the block that draws the LAST one reads them all -- nobody waits.
// One counter per (slot, group); a launch takes the next slot (host side, round robin), the finishing block resets its
// counter, so a slot is clean again long before the ring of slots comes back to it.  Agent-scope release / acquire around a
// relaxed ticket (cdna_hip_programming.md, in-launch split-K reduction): correct wherever the blocks run.
#pragma once
#include "common.hpp"

constexpr int TICKET_SLOTS = 256, TICKET_GROUPS = 32;
unsigned* segk_ticket_slot();          // host: TICKET_GROUPS counters of the next slot of the current device (bn_pool.hip)

// true in every thread of the block that arrives LAST of `n` at `counter` (the caller's global stores are published
// first; the last block may then read every other block's).  flag: one int of LDS nobody else touches across the call.
__device__ __forceinline__ bool last_arriver(unsigned* counter, unsigned n, volatile int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == n - 1) ? 1 : 0;
    if (last) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *flag = last;
  }
  __syncthreads();
  return *flag != 0;
}
