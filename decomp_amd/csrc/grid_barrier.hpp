// Grid-wide barrier for kernels whose workgroups are all co-resident (grid <= CUs x blocks per CU the
// kernel's resources admit; the launcher sizes the grid, nothing here can check it).
//
// Hierarchical counter barrier (guide: MI355X_MICROARCH.md row "barrier-xcd", cdna_hip_programming.md
// Guideline 16): workgroups are dealt into 8 GROUPS by blockIdx % 8 -- blocks b and b + 8 are observed to
// share an XCD, which makes a group's counter line XCD-local; that is a speed assumption only.  Correctness
// never depends on placement: every workgroup publishes with its own agent-scope release before it arrives
// and acquires (agent scope) after it is released, whatever CU / XCD it runs on.
//   arrive:  all waves drain their stores (s_waitcnt vmcnt(0)), workgroup barrier, lane 0: release fence,
//            asm wait (the compiler may drop the fence's own wait, Guideline 16 pitfall 12), atomic add on
//            the group's counter (8 counters of <= 32 arrivals instead of 256 arrivals on one word)
//   last arriver of a group: atomic add on the top counter, relaxed poll of it until all groups are in,
//            then stores the epoch into the group's generation word
//   others:  relaxed poll of their group's generation word (s_sleep between polls)
//   all:     lane 0 acquire fence + asm wait, workgroup barrier.
// Counters are monotonic inside a launch (epoch e = 1, 2, ... targets e x members); the LAST workgroup to
// draw a ticket at the end of the kernel (grid_barrier_finish) re-zeroes the state for the next launch, so
// no memset node is needed between launches and a captured graph replays correctly.  Every spin is
// bounded: on expiry the timeout word is set and the kernel runs to completion with undefined results,
// which the host reports as an error -- it never hangs the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcp {

// one 128-byte line per word that is polled or added to
struct GridBarrierState {
    struct alignas(128) Line { unsigned v; unsigned pad[31]; };
    Line group_count[8];
    Line group_gen[8];
    Line top;
    Line ticket;    // grid_barrier_finish
    Line timeout;   // != 0: a spin expired (sticky until the host clears it)
};

#define DCP_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ bool gb_wait_ge(unsigned* word, unsigned target, unsigned* timeout_word) {
    // ~2^22 polls x (load round trip + s_sleep) is of the order of seconds: far beyond any legitimate wait
    for (unsigned spins = 0; spins < (1u << 22); ++spins) {
        const unsigned v = __hip_atomic_load(word, DCP_RLX_AGENT);
        if ((int)(v - target) >= 0) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_store(timeout_word, 1u, DCP_RLX_AGENT);
    return false;
}

// epoch: 1 for the first barrier of the launch, 2 for the second, ...  Uniform across the grid.
__device__ __forceinline__ void grid_barrier(GridBarrierState* bs, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every wave: its stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned nwg = gridDim.x;
        const unsigned ngroups = nwg < 8u ? nwg : 8u;
        const unsigned g = blockIdx.x & 7u;
        const unsigned members = (nwg - g + 7u) >> 3;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(&bs->group_count[g].v, 1u, DCP_RLX_AGENT);
        if (prev + 1u == epoch * members) {
            __hip_atomic_fetch_add(&bs->top.v, 1u, DCP_RLX_AGENT);
            gb_wait_ge(&bs->top.v, epoch * ngroups, &bs->timeout.v);
            __hip_atomic_store(&bs->group_gen[g].v, epoch, DCP_RLX_AGENT);
        } else {
            gb_wait_ge(&bs->group_gen[g].v, epoch, &bs->timeout.v);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// End of the kernel, after the last grid_barrier: every workgroup draws a ticket (its own stores published
// first); returns true in the ONE workgroup that drew the last ticket -- by then every other workgroup has
// left its last barrier, so that workgroup (a) may read what the others published before their ticket and
// (b) re-zeroes the barrier state for the next launch.  No workgroup waits here.
__device__ __forceinline__ bool grid_barrier_finish(GridBarrierState* bs, int* s_last /* LDS word */) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(&bs->ticket.v, 1u, DCP_RLX_AGENT);
        const bool last = (prev + 1u == gridDim.x);
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int g = 0; g < 8; ++g) {
                __hip_atomic_store(&bs->group_count[g].v, 0u, DCP_RLX_AGENT);
                __hip_atomic_store(&bs->group_gen[g].v, 0u, DCP_RLX_AGENT);
            }
            __hip_atomic_store(&bs->top.v, 0u, DCP_RLX_AGENT);
            __hip_atomic_store(&bs->ticket.v, 0u, DCP_RLX_AGENT);
        }
        *s_last = last ? 1 : 0;
    }
    __syncthreads();
    return *s_last != 0;
}

}  // namespace dcp
