// In-launch scan of the per-partition digit histograms of a radix pass — what radix_sort_spine.glsl:35-92 does in a
// dispatch of its own (one group per digit scanning all partitions), folded into the kernel that PRODUCES the
// histograms.  gfx950 only.
//
// Round 3 ran a spine kernel per pass: 256 workgroups of 1024 lanes, ~6.5 us each at 2 500 partitions — a launch, two
// dependent memory round trips and a drain for 2.5 MB of counters, four times per frame, plus digit-major histograms
// (the layout a row-scanning spine wants) that every producer wrote and every downsweep read as 256 scattered words.
// Here the histograms are PARTITION-MAJOR (hist[partition][256]: one contiguous KiB per producer workgroup, coalesced
// for the producer, the scan and the downsweep alike) and the scan is two levels of "last arriver does it":
//   level 1  partitions are grouped in chunks of C (a power of two, C^2 >= number of partitions).  A producer workgroup
//            stores its row write-through, draws a ticket on its chunk's counter, and the workgroup that draws the last
//            ticket of the chunk turns the chunk's C rows into exclusive in-chunk prefixes (in place) and publishes the
//            chunk's digit totals;
//   level 2  that workgroup then draws a ticket on the pass counter; the last of THOSE scans the <= C chunk totals
//            into chunk bases, and the digit totals into the pass's digit bases.
// Nobody waits for anybody (no spinning, no residency assumption, no dispatch-order assumption): a workgroup either
// finds it is last and does the work, or leaves.  A downsweep workgroup then needs three coalesced reads per digit:
// digit_base[d] + chunk_base[chunk][d] + hist[partition][d].
//
// Visibility between workgroups of one launch (MI355X: eight XCDs with private L2s, per-CU L1s) follows
// cdna_hip_programming.md §6 Guideline 16 in its write-through form: every word another workgroup will read in THIS
// launch is stored with an agent-scope relaxed atomic store (global_store ... sc1) and loaded with an agent-scope relaxed
// atomic load (global_load ... sc1); every storing wave drains (s_waitcnt vmcnt(0)) before the workgroup barrier that
// precedes the ticket; the ticket is an agent-scope atomic.  Words only the NEXT launch reads (the in-chunk prefixes, the
// chunk bases, the digit bases) are plain stores: the kernel boundary publishes them.  The counters are zero when the
// buffers are allocated and every last arriver puts its counter back to zero, so a launch always finds them zero.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsplat {

constexpr uint32_t HIST_BINS = 256;

// partitions per chunk: the smallest power of two >= 32 whose square covers the partitions (chunks <= C)
__host__ __device__ inline uint32_t hist_chunk_parts(uint32_t num_parts) {
    uint32_t c = 32;
    while ((uint64_t)c * c < num_parts) c <<= 1;
    return c;
}
// rows of chunk_total / chunk_base and ticket words a pass of up to max_parts partitions may need (the chunk count is
// not monotone in the partition count, but it never exceeds C(max_parts))
inline uint32_t hist_max_chunks(uint32_t max_parts) { return hist_chunk_parts(max_parts) + 1u; }

struct HistScan {
    uint32_t *chunk_total;  // [chunks][256] written write-through by the chunk's last arriver, read by the pass's
    uint32_t *chunk_base;   // [chunks][256] exclusive scan of chunk_total over the chunks (next launch reads it)
    uint32_t *digit_base;   // [257] exclusive scan of the digit totals over the digits; [256] = number of elements
    uint32_t *tickets;      // [chunks + 1] arrival counters, zero between launches; the last word is the pass's
    uint32_t pass_ticket;   // index of that last word (= rows allocated)
};

__device__ __forceinline__ void hs_store(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t hs_load(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// two neighbouring words as ONE 8-byte write-through store (a 4-byte sc1 store is a fabric write of its own)
__device__ __forceinline__ void hs_store2(uint32_t *p_even, uint32_t lo, uint32_t hi) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p_even), (unsigned long long)lo | ((unsigned long long)hi << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every lane of the workgroup calls this (it contains workgroup barriers) once the workgroup's write-through stores are
// issued.  Returns true in every lane of exactly one of the `members` workgroups that call it on `counter`: the last one
// to arrive, by which time the write-through stores of all the others are visible to agent-scope loads.  flag: one LDS
// word of the caller.
__device__ __forceinline__ bool hs_arrive_last(uint32_t *counter, uint32_t members, uint32_t *flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = t + 1u == members;
        if (last) hs_store(counter, 0u);  // (nobody else touches it any more in this launch)
        *flag = last ? 1u : 0u;
    }
    __syncthreads();
    const bool last = *flag != 0u;
    __syncthreads();  // (the flag word may be reused by the caller's next call)
    return last;
}

// The producer side.  Every lane of the workgroup calls it, once per partition the workgroup has counted:
//   value      lane d < 256: the number of this partition's elements with digit d (lanes >= 256: ignored)
//   hist       [num_parts][256] of the pass
// scratch: 8 LDS words of the caller (workgroups of 256 lanes or more).
__device__ __forceinline__ void hist_publish_and_scan(uint32_t value, uint32_t *__restrict__ hist, uint32_t part,
                                                      uint32_t num_parts, const HistScan &hs, uint32_t *scratch) {
    const uint32_t tid = threadIdx.x;
    const uint32_t C = hist_chunk_parts(num_parts);
    const uint32_t chunk = part / C, num_chunks = (num_parts + C - 1u) / C;
    const uint32_t first = chunk * C, members = min(C, num_parts - first);
    {   // the partition's row, write-through, two digits per store
        const uint32_t hi = __shfl_down(value, 1, 64);
        if (tid < HIST_BINS && (tid & 1u) == 0u) hs_store2(hist + (size_t)part * HIST_BINS + tid, value, hi);
    }
    if (!hs_arrive_last(hs.tickets + chunk, members, scratch + 4)) return;
    // ---- level 1: this workgroup completes chunk `chunk` (lane d: digit d, walking the chunk's rows in order)
    uint32_t total = 0;
    if (tid < HIST_BINS) {
        uint32_t *row = hist + (size_t)first * HIST_BINS + tid;
        uint32_t r = 0;
        for (; r + 8u <= members; r += 8u) {  // eight independent loads in flight, then the dependent chain
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = hs_load(row + (size_t)(r + k) * HIST_BINS);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                row[(size_t)(r + k) * HIST_BINS] = total;  // exclusive in-chunk prefix (the next launch reads it)
                total += v[k];
            }
        }
        for (; r < members; ++r) {
            const uint32_t v = hs_load(row + (size_t)r * HIST_BINS);
            row[(size_t)r * HIST_BINS] = total;
            total += v;
        }
    }
    {
        const uint32_t hi = __shfl_down(total, 1, 64);
        if (tid < HIST_BINS && (tid & 1u) == 0u) hs_store2(hs.chunk_total + (size_t)chunk * HIST_BINS + tid, total, hi);
    }
    if (!hs_arrive_last(hs.tickets + hs.pass_ticket, num_chunks, scratch + 4)) return;
    // ---- level 2: this workgroup completes the pass
    uint32_t sum = 0;
    if (tid < HIST_BINS) {
        const uint32_t *col = hs.chunk_total + tid;
        uint32_t *base = hs.chunk_base + tid;
        uint32_t c = 0;
        for (; c + 8u <= num_chunks; c += 8u) {
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = hs_load(col + (size_t)(c + k) * HIST_BINS);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                base[(size_t)(c + k) * HIST_BINS] = sum;
                sum += v[k];
            }
        }
        for (; c < num_chunks; ++c) {
            const uint32_t v = hs_load(col + (size_t)c * HIST_BINS);
            base[(size_t)c * HIST_BINS] = sum;
            sum += v;
        }
    }
    // exclusive scan of the digit totals over the digits (lanes 0..255; four waves)
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t incl = tid < HIST_BINS ? sum : 0u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63 && wave < 4) scratch[wave] = incl;
    __syncthreads();
    if (tid < HIST_BINS) {
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t t = scratch[w];
            if (w < wave) before += t;
            all += t;
        }
        hs.digit_base[tid] = before + incl - sum;
        if (tid == 0) hs.digit_base[HIST_BINS] = all;
    }
    __syncthreads();
}

// The consumer side (the NEXT launch): where the elements of partition `row`'s digit d start in the pass's output,
// before the partition's own local offsets.  row = index of the partition's (first) histogram row.
__device__ __forceinline__ uint32_t hist_digit_start(const uint32_t *__restrict__ hist, uint32_t row, uint32_t num_rows,
                                                     const HistScan &hs, uint32_t d) {
    const uint32_t C = hist_chunk_parts(num_rows);
    return hs.digit_base[d] + hs.chunk_base[(size_t)(row / C) * HIST_BINS + d] + hist[(size_t)row * HIST_BINS + d];
}

}  // namespace gsplat
