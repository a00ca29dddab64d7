// Pieces shared by the candidate filters of the peak stage (network-resolution, full-resolution
// and multi-scale sources): list layout, tolerances, the candidate list of a context.
#pragma once
#include <math.h>

#include "common.h"

#pragma clang fp contract(off)

#define KF_CELLS_PER_WAVE 6   // ten lanes per cell
#define KF_CELLS_PER_BLOCK 24
#define KF_LCAND 512          // candidates staged in LDS per workgroup between flushes
#define KF_OVERFLOW (1 << 30)  // or-ed into a plane's candidate count: more than any list length
// The cell list and the candidate list are filled by thousands of workgroups.  Atomics
// on one address serialise chip-wide (tens of nanoseconds each), so each list is split
// into KF_NSEG segments with a counter of their own; readers rebuild the global order
// from the 32 counts.
#define KF_NSEG 32

// Lower threshold and neighbour slack of the candidate test for an image whose
// heatmaps are bounded by absmax.
// shift: eps = 2^-shift * absmax (17 for one network-resolution or full-resolution source).
__device__ __forceinline__ void filter_slack(double thr, float absmax, float* thr_lo, float* e2, int shift = 17) {
    const double eps = ldexp((double)absmax, -shift);
    *thr_lo = nextafterf((float)(thr - eps), -INFINITY);
    *e2 = nextafterf((float)(2.0 * eps), INFINITY);
}

// One step of max |v| over a plane that does not lose what the tolerance cannot cover: a NaN or an
// infinity makes the maximum +inf.  The float32 filter multiplies sources outside an output's
// support by zero weights, which a non-finite value turns into NaN where the reference, which never
// touches that source, has a finite value; planes with such values are left to the exact kernel
// (their candidate count gets KF_OVERFLOW and no cell of theirs is listed).
__device__ __forceinline__ float absmax_step(float m, float v) {
    const float a = fabsf(v);
    return a <= m ? m : (a == a ? a : INFINITY);   // (a <= m is false for a NaN)
}
__device__ __forceinline__ bool absmax_finite(float m) { return m <= 3.4028235e38f; }

// Exclusive prefix of the segment counts (each clipped to the segment capacity) into
// pref[0 .. KF_NSEG]; call with all threads, ends with a barrier.
__device__ __forceinline__ void segment_prefix(const int32_t* counts, int seg_cap, int* pref, int tid) {
    if (tid < 64) {
        int c = tid < KF_NSEG ? min(counts[tid], seg_cap) : 0;
        int incl = c;
#pragma unroll
        for (int d = 1; d < KF_NSEG; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (tid >= d) incl += o;
        }
        if (tid < KF_NSEG) pref[tid + 1] = incl;
        if (tid == 0) pref[0] = 0;
    }
    __syncthreads();
}

// Global index -> position in the segmented array (segment * seg_cap + offset).
__device__ __forceinline__ int segment_locate(const int* pref, int seg_cap, int gi) {
    int seg = 0;
#pragma unroll
    for (int d = KF_NSEG / 2; d >= 1; d >>= 1)
        if (pref[seg + d] <= gi) seg += d;
    return seg * seg_cap + (gi - pref[seg]);
}

// Device buffer of a context grown to need_bytes (contents are not kept; recorded launch
// sequences are dropped through the generation counter).
int pd_grow(pd_ctx* ctx, void** p, size_t* cap, size_t need_bytes);
// The candidate list: room for CK = 4 * max_peaks_per_part entries per plane.
int pd_ensure_candidates(pd_ctx* ctx, int* CK_out);
