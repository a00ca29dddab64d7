// Orders the peaks each (image, part) emitted into the row-major (y, then x) order
// of np.nonzero (utils/postprocess.py:63), assigns the running ids of
// utils/postprocess.py:68-73 and lays out the limb-candidate pair ranges.
#include "peaksort.h"

#pragma clang fp contract(off)

#define SORT_WAVES 9  // each wavefront orders parts w and w + 9

// One workgroup per image, one wavefront per two parts.
__global__ void __launch_bounds__(SORT_WAVES * 64)
k_sort_peaks(SortArgs a) {
    __shared__ SortShared sh;
    pd_sort_image(a, sh, blockIdx.x, threadIdx.x, SORT_WAVES * 64);
}

void pd_fill_sort_args(const pd_ctx* ctx, SortArgs* a) {
    a->raw = ctx->raw;
    a->raw_count = ctx->raw_count;
    a->PK = ctx->cfg.max_peaks_per_part;
    a->W = ctx->W;
    a->max_pairs = ctx->cfg.max_pairs_per_image;
    a->pk_y = ctx->pk_y;
    a->pk_x = ctx->pk_x;
    a->pk_id = ctx->pk_id;
    a->pk_score = ctx->pk_score;
    a->part_count = ctx->part_count;
    a->part_base = ctx->part_base;
    a->pair_off = ctx->pair_off;
    a->status = ctx->status;
    a->cand_count = ctx->cand_count;
    a->n_planes = ctx->B * PD_N_PARTS;
    a->cand_cap = ctx->cand_cap > 0 ? ctx->cand_cap : 0x7fffffff;
    a->fstats = ctx->fstats;
}

int pd_launch_sort_peaks(pd_ctx* ctx) {
    if (ctx->peaks_sorted) {  // the kernel that finished the peak stage has ordered them already
        ctx->peaks_sorted = 0;
        return PD_OK;
    }
    SortArgs a;
    pd_fill_sort_args(ctx, &a);
    pd_prof_begin(ctx, PD_STAGE_SORT);
    hipLaunchKernelGGL(k_sort_peaks, dim3(ctx->B), dim3(SORT_WAVES * 64), 0, ctx->stream, a);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    return PD_OK;
}
