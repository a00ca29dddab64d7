// Ordering of the peaks of one image: row-major (y, then x) order of np.nonzero
// (utils/postprocess.py:63), the running ids of utils/postprocess.py:68-73 and the
// limb-candidate pair ranges.  Shared by the kernel of its own (k_sort_peaks) and by the
// kernel that finishes the peak stage of the filtered path.
#pragma once
#include "common.h"

struct SortArgs {
    const RawPeak* raw;
    const int32_t* raw_count;
    int PK, W, max_pairs;
    int32_t *pk_y, *pk_x, *pk_id;
    double* pk_score;
    int32_t *part_count, *part_base, *pair_off, *status;
    const int32_t* cand_count;  // the candidate filter's counters (for pd_filter_stats)
    int n_planes, cand_cap;
    int64_t* fstats;
};

struct SortShared {
    int32_t cnt[PD_N_PARTS];
    int32_t base[PD_N_PARTS + 1];
    int over;
    unsigned long long stat[4];
};

// All `nthreads` threads of the workgroup (a multiple of 64) call this for image b.  A part's
// peaks sit one per lane (64 at a time); the rank of a peak is the number of peaks with a
// smaller key y*W + x (keys are unique), counted by broadcasting every key across the wave.
__device__ __forceinline__ void pd_sort_image(const SortArgs& a, SortShared& sh, int b, int tid, int nthreads) {
    const int wave = tid >> 6, lane = tid & 63, n_waves = nthreads >> 6;
    const int PK = a.PK, W = a.W;
    if (tid == 0) sh.over = 0;
    if (tid < 4) sh.stat[tid] = 0ull;
    __syncthreads();
    if (tid < PD_N_PARTS) {
        int n = a.raw_count[b * PD_N_PARTS + tid];
        if (n > PK) {
            n = PK;
            sh.over = 1;
        }
        sh.cnt[tid] = n;
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int p = 0; p < PD_N_PARTS; ++p) {
            sh.base[p] = acc;
            acc += sh.cnt[p];
        }
        sh.base[PD_N_PARTS] = acc;
        // candidate-pair ranges per limb; a total of -1 tells the connection stage
        // that max_pairs_per_image is too small (not an error for peaks alone)
        long long pairs = 0;
        bool pairs_over = false;
        for (int k = 0; k < PD_N_LIMBS; ++k) {
            a.pair_off[b * (PD_N_LIMBS + 1) + k] = (int32_t)pairs;
            pairs += (long long)sh.cnt[pd_limb_from(k)] * sh.cnt[pd_limb_to(k)];
            if (pairs > a.max_pairs) {
                pairs = a.max_pairs;
                pairs_over = true;
            }
        }
        a.pair_off[b * (PD_N_LIMBS + 1) + PD_N_LIMBS] = pairs_over ? -1 : (int32_t)pairs;
        a.status[b] = sh.over ? PD_E_OVERFLOW : PD_OK;
    }
    __syncthreads();
    if (b == 0) {
        // what the candidate filter did, for pd_filter_stats: the counters themselves are
        // cleared by the pack kernel.  Behind the per-plane counts sit 3 x 32 list counters
        // (candidates, interior cells, border cells).
        unsigned long long over = 0;
        for (int i = tid; i < a.n_planes; i += nthreads) over += a.cand_count[i] > a.cand_cap;
        if (over) atomicAdd(&sh.stat[3], over);
        for (int i = tid; i < 96; i += nthreads) atomicAdd(&sh.stat[i >> 5], (unsigned long long)a.cand_count[a.n_planes + i]);
    }
    if (tid < PD_N_PARTS) a.part_count[b * PD_N_PARTS + tid] = sh.cnt[tid];
    if (tid <= PD_N_PARTS) a.part_base[b * (PD_N_PARTS + 1) + tid] = sh.base[tid];
    for (int p = wave; p < PD_N_PARTS; p += n_waves) {
        const int n = sh.cnt[p];
        const RawPeak* list = a.raw + ((size_t)b * PD_N_PARTS + p) * PK;
        const size_t o = ((size_t)b * PD_N_PARTS + p) * PK;
        const int base = sh.base[p];
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int i = c0 + lane;
            RawPeak me;
            me.y = 0;
            me.x = 0;
            me.score = 0.0;
            if (i < n) me = list[i];
            const int key = me.y * W + me.x;  // H * W < 2^31
            int rank = 0;
            for (int d0 = 0; d0 < n; d0 += 64) {
                const int j = d0 + lane;
                int other = 0x7fffffff;
                if (j < n) {
                    const RawPeak q = list[j];
                    other = q.y * W + q.x;
                }
                const int m = min(64, n - d0);
                for (int s = 0; s < m; ++s) rank += __shfl(other, s) < key;
            }
            if (i < n) {
                a.pk_y[o + rank] = me.y;
                a.pk_x[o + rank] = me.x;
                a.pk_score[o + rank] = me.score;
                a.pk_id[o + rank] = base + rank;
            }
        }
    }
    if (b == 0) {
        __syncthreads();
        if (tid < 4) a.fstats[tid] = (int64_t)sh.stat[tid];
    }
}

void pd_fill_sort_args(const pd_ctx* ctx, SortArgs* a);
