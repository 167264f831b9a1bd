// In-launch hand-off from the blocks of one reduction to the block that finishes it (instead of a second, tiny kernel):
// every block publishes its partials and draws a ticket; the block that draws the LAST one reads them all -- nobody waits.
// One counter per (slot, group).  A launch takes the next slot of a ring (host side, round robin) and the host ZEROES the
// counters it will use on the launch's stream right before the launch (a 4..128-byte memset node, ~2 us of stream time), so
// whatever an earlier launch left behind -- an aborted launch's counts, a stray store -- cannot keep this launch from
// electing its finisher; nothing has to be reset on the device.  (Rounds 2-3 relied on the finishing block resetting its
// counter, i.e. on every launch completing.  Round 4 first tried generation-tagged 64-bit words claimed by a compare-and-
// swap: correct, and 10 x slower under contention -- 256 blocks arriving at one word took 290 us instead of 27 for the loss
// forward, 0.42 ms per training step; profiles/r04_experiments.txt.)
// Agent-scope release / acquire around a relaxed ticket (cdna_hip_programming.md, in-launch split-K reduction): correct
// wherever the blocks run.
#pragma once
#include "common.hpp"

constexpr int TICKET_SLOTS = 256, TICKET_GROUPS = 32;
// host: `groups` zeroed counters of the next slot of the current device, zeroed on stream `st` (bn_pool.hip); nullptr on failure
unsigned* segk_ticket_slot(int groups, hipStream_t st);

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
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *flag = last;
  }
  __syncthreads();
  return *flag != 0;
}
