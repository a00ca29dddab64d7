// Orders the peaks each (image, part) emitted into the row-major (y, then x) order
// of np.nonzero (utils/postprocess.py:63), assigns the running ids of
// utils/postprocess.py:68-73 and lays out the limb-candidate pair ranges.
#include "common.h"

#pragma clang fp contract(off)

// One workgroup per image.  Each part's list is ranked by the unique key y*W + x.
__global__ void __launch_bounds__(256)
k_sort_peaks(const RawPeak* __restrict__ raw, const int32_t* __restrict__ raw_count, int PK, int W,
             int max_pairs, int32_t* pk_y, int32_t* pk_x, int32_t* pk_id, double* pk_score,
             int32_t* part_count, int32_t* part_base, int32_t* pair_off, int32_t* status) {
    __shared__ int32_t s_cnt[PD_N_PARTS];
    __shared__ int32_t s_base[PD_N_PARTS + 1];
    __shared__ int s_over;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid == 0) s_over = 0;
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
    }
    __syncthreads();
    if (tid < PD_N_PARTS) part_count[b * PD_N_PARTS + tid] = s_cnt[tid];
    if (tid <= PD_N_PARTS) part_base[b * (PD_N_PARTS + 1) + tid] = s_base[tid];
    if (tid == 0) status[b] = s_over ? PD_E_OVERFLOW : PD_OK;
    for (int p = 0; p < PD_N_PARTS; ++p) {
        const int n = s_cnt[p];
        const RawPeak* list = raw + ((size_t)b * PD_N_PARTS + p) * PK;
        const size_t o = ((size_t)b * PD_N_PARTS + p) * PK;
        for (int i = tid; i < n; i += blockDim.x) {
            const RawPeak me = list[i];
            const long long key = (long long)me.y * W + me.x;
            int rank = 0;
            for (int j = 0; j < n; ++j) {
                const RawPeak o2 = list[j];
                rank += ((long long)o2.y * W + o2.x) < key;
            }
            pk_y[o + rank] = me.y;
            pk_x[o + rank] = me.x;
            pk_score[o + rank] = me.score;
            pk_id[o + rank] = s_base[p] + rank;
        }
    }
}

int pd_launch_sort_peaks(pd_ctx* ctx) {
    pd_prof_begin(ctx, PD_STAGE_SORT);
    hipLaunchKernelGGL(k_sort_peaks, dim3(ctx->B), dim3(256), 0, ctx->stream, ctx->raw, ctx->raw_count,
                       ctx->cfg.max_peaks_per_part, ctx->W, ctx->cfg.max_pairs_per_image, ctx->pk_y, ctx->pk_x,
                       ctx->pk_id, ctx->pk_score, ctx->part_count, ctx->part_base, ctx->pair_off, ctx->status);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    return PD_OK;
}
