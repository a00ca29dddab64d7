// Orders the peaks each (image, part) emitted into the row-major (y, then x) order
// of np.nonzero (utils/postprocess.py:63), assigns the running ids of
// utils/postprocess.py:68-73 and lays out the limb-candidate pair ranges.
#include "common.h"

#pragma clang fp contract(off)

#define SORT_WAVES 9  // each wavefront orders parts w and w + 9

// One workgroup per image, one wavefront per part.  A part's peaks sit one per lane
// (64 at a time); the rank of a peak is the number of peaks with a smaller key
// y*W + x (keys are unique), counted by broadcasting every key across the wave.
__global__ void __launch_bounds__(SORT_WAVES * 64)
k_sort_peaks(const RawPeak* __restrict__ raw, const int32_t* __restrict__ raw_count, int PK, int W,
             int max_pairs, int32_t* pk_y, int32_t* pk_x, int32_t* pk_id, double* pk_score,
             int32_t* part_count, int32_t* part_base, int32_t* pair_off, int32_t* status,
             const int32_t* __restrict__ cand_count, int n_planes, int cand_cap, int64_t* fstats) {
    __shared__ int32_t s_cnt[PD_N_PARTS];
    __shared__ int32_t s_base[PD_N_PARTS + 1];
    __shared__ int s_over;
    __shared__ unsigned long long s_stat[4];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    if (tid == 0) s_over = 0;
    if (tid < 4) s_stat[tid] = 0ull;
    __syncthreads();
    if (tid < PD_N_PARTS) {
        int n = raw_count[b * PD_N_PARTS + tid];
        if (n > PK) {
            n = PK;
            s_over = 1;
        }
        s_cnt[tid] = n;
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int p = 0; p < PD_N_PARTS; ++p) {
            s_base[p] = acc;
            acc += s_cnt[p];
        }
        s_base[PD_N_PARTS] = acc;
        // candidate-pair ranges per limb; a total of -1 tells the connection stage
        // that max_pairs_per_image is too small (not an error for peaks alone)
        long long pairs = 0;
        bool pairs_over = false;
        for (int k = 0; k < PD_N_LIMBS; ++k) {
            pair_off[b * (PD_N_LIMBS + 1) + k] = (int32_t)pairs;
            pairs += (long long)s_cnt[pd_limb_from(k)] * s_cnt[pd_limb_to(k)];
            if (pairs > max_pairs) {
                pairs = max_pairs;
                pairs_over = true;
            }
        }
        pair_off[b * (PD_N_LIMBS + 1) + PD_N_LIMBS] = pairs_over ? -1 : (int32_t)pairs;
        status[b] = s_over ? PD_E_OVERFLOW : PD_OK;
    }
    __syncthreads();
    if (b == 0) {
        // what the candidate filter did, for pd_filter_stats: the counters themselves are
        // cleared by the pack kernel.  Behind the per-plane counts sit 3 x 32 list counters
        // (candidates, interior cells, border cells).
        unsigned long long over = 0;
        for (int i = tid; i < n_planes; i += SORT_WAVES * 64) over += cand_count[i] > cand_cap;
        if (over) atomicAdd(&s_stat[3], over);
        if (tid < 96) atomicAdd(&s_stat[tid >> 5], (unsigned long long)cand_count[n_planes + tid]);
    }
    if (tid < PD_N_PARTS) part_count[b * PD_N_PARTS + tid] = s_cnt[tid];
    if (tid <= PD_N_PARTS) part_base[b * (PD_N_PARTS + 1) + tid] = s_base[tid];
    for (int p = wave; p < PD_N_PARTS; p += SORT_WAVES) {
        const int n = s_cnt[p];
        const RawPeak* list = raw + ((size_t)b * PD_N_PARTS + p) * PK;
        const size_t o = ((size_t)b * PD_N_PARTS + p) * PK;
        const int base = s_base[p];
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
                pk_y[o + rank] = me.y;
                pk_x[o + rank] = me.x;
                pk_score[o + rank] = me.score;
                pk_id[o + rank] = base + rank;
            }
        }
    }
    if (b == 0) {
        __syncthreads();
        if (tid < 4) fstats[tid] = (int64_t)s_stat[tid];
    }
}

int pd_launch_sort_peaks(pd_ctx* ctx) {
    pd_prof_begin(ctx, PD_STAGE_SORT);
    hipLaunchKernelGGL(k_sort_peaks, dim3(ctx->B), dim3(SORT_WAVES * 64), 0, ctx->stream, ctx->raw, ctx->raw_count,
                       ctx->cfg.max_peaks_per_part, ctx->W, ctx->cfg.max_pairs_per_image, ctx->pk_y, ctx->pk_x,
                       ctx->pk_id, ctx->pk_score, ctx->part_count, ctx->part_base, ctx->pair_off, ctx->status,
                       ctx->cand_count, ctx->B * PD_N_PARTS, ctx->cand_cap > 0 ? ctx->cand_cap : 0x7fffffff, ctx->fstats);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    return PD_OK;
}
