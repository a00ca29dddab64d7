// Gaussian smoothing (sigma = 3, radius 12, reflect) + 4-neighbour NMS + peak
// emission: get_peaks of utils/postprocess.py:26-75 (twin predict.py:88-111).
//
// Bit-exactness contract (scipy correlate1d, symmetric-kernel branch): per output
//   t = x[c] * w[12];  for d = 12 .. 1:  t += (x[c-d] + x[c+d]) * w[12-d]
// in float64 with separately rounded multiply and add, axis 0 (y) first, then
// axis 1 (x); borders by half-sample reflection; a float32 source is rounded to
// float32 after each axis.  NMS compares with >= against the four neighbours,
// out-of-image neighbours count as 0.0, and requires value > threshold.
#include <stdlib.h>

#include "colloader.h"
#include "multiscale.h"
#include "peaksort.h"

#pragma clang fp contract(off)

struct SmoothArgs {
    MapSource src;
    double thr;
    int f32mode;
    int n_strips, strip_w;
    RawPeak* raw;
    int32_t* raw_count;
    int PK;
    double* dbg;  // optional (B, 18, H, W)
    // when set, only the (image, part) planes whose count exceeds over_limit are
    // processed (planes the candidate filter could not hold)
    const int32_t* only_over;
    int over_limit;
    SortArgs sort;  // k_finish_peaks: what the ordering of the peaks needs
};

// What a multi-scale source adds: the scales (a kernel argument of their own) and the x8 weights
// by phase in LDS.
#define PD_MULTI_REF const MultiSource *ms, const float4 *up8

// Column of the average over the scales of a multi-scale source, evaluated pixel by pixel from
// the network outputs (the two-step bicubic of every scale, its float32 share, the float64 sum).
// Only planes whose candidate list overflowed come this way.
template <>
struct ColLoader<PD_SRC_MULTI> {
    const MultiSource* ms;
    const float4* up;
    int b, ch, col;
    __device__ __forceinline__ void init(const MapSource&, int b_, int ch_, int c) {
        b = b_;
        ch = ch_;
        col = c;
    }
    __device__ __forceinline__ double load(int r) const {
        double acc[1];
        pd_ms_avg_at<1>(*ms, false, b, ch, r, col, up, acc);
        return acc[0];
    }
};

template <int SRC>
__device__ __forceinline__ void loader_attach(ColLoader<SRC>&, PD_MULTI_REF) {}
template <>
__device__ __forceinline__ void loader_attach<PD_SRC_MULTI>(ColLoader<PD_SRC_MULTI>& ld, PD_MULTI_REF) {
    ld.ms = ms;
    ld.up = up8;
}

__device__ __forceinline__ void emit_peak(bool is_peak, int y, int x, double score, RawPeak* list,
                                          int32_t* count, int PK) {
    const unsigned long long m = __ballot(is_peak);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(count, (int)__builtin_popcountll(m));
    base = __shfl(base, (int)__builtin_ctzll(m));
    if (is_peak) {
        const int pos = base + (int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (pos < PK) {
            RawPeak p;
            p.y = y;
            p.x = x;
            p.score = score;
            list[pos] = p;
        }
    }
}

// ---------------------------------------------------------------------------------
// Streaming kernel: one workgroup owns one (image, part, column strip) and marches
// down the rows.  Thread t owns column xs + t: the 25-row window of the unsmoothed
// map lives in its registers (slot = row mod 25, made static by unrolling 25 row
// steps), so the y pass needs no memory at all; the y-pass row goes through LDS
// once for the x pass, and the last three smoothed rows stay in registers for NMS.
// ---------------------------------------------------------------------------------
#define NTMAX 512

__device__ __forceinline__ double round_store(double v, int f32mode) {
    return f32mode ? (double)(float)v : v;
}

struct RowState {
    double s_prev1, s_prev2;  // smoothed rows y-1, y-2 of this column
};

template <int SRC, int S>
__device__ __forceinline__ void row_step(const SmoothArgs& a, ColLoader<SRC>& ld, double (&win)[25],
                                         RowState& st, double* ybuf, double* sbuf, int y, int H, int W,
                                         int c, int xs, int xe, bool col_active, bool do_x, bool do_nms,
                                         RawPeak* list, int32_t* count, double* dbg_plane) {
    const int tid = threadIdx.x;
    const int par = y & 1;
    double* yrow = ybuf + par * (NTMAX + 2 * PD_GR);
    if (y < H) {
        if (col_active) {
            int rn = y + PD_GR;
            if (rn >= H) rn = 2 * H - 1 - rn;
            win[(S + 12) % 25] = ld.load(rn);
            double t = win[S] * c_gw[12];
#pragma unroll
            for (int d = 12; d >= 1; --d) t += (win[(S + 25 - d) % 25] + win[(S + d) % 25]) * c_gw[12 - d];
            t = round_store(t, a.f32mode);
            yrow[tid + PD_GR] = t;
            // mirrored halo so the x pass reads straight offsets
            if (xs == 0 && c < PD_GR) yrow[PD_GR - 1 - c] = t;
            if (xe == W && c >= W - PD_GR) yrow[(2 * W - 1 - c) - xs + PD_GR] = t;
        }
    }
    __syncthreads();
    double s = 0.0;
    if (y < H && do_x) {
        const double* p = yrow + tid + PD_GR;
        s = p[0] * c_gw[12];
#pragma unroll
        for (int d = 12; d >= 1; --d) s += (p[-d] + p[d]) * c_gw[12 - d];
        s = round_store(s, a.f32mode);
        if (dbg_plane != nullptr && do_nms) dbg_plane[(size_t)y * W + c] = s;
    }
    sbuf[par * (NTMAX + 2) + tid + 1] = s;
    if (y >= 1) {
        bool is_peak = false;
        const double m = st.s_prev1;
        if (do_nms) {
            const double* srow = sbuf + (par ^ 1) * (NTMAX + 2);
            const double left = (c == 0) ? 0.0 : srow[tid];
            const double right = (c == W - 1) ? 0.0 : srow[tid + 2];
            const double up = (y == 1) ? 0.0 : st.s_prev2;
            const double down = (y == H) ? 0.0 : s;
            is_peak = (m >= up) && (m >= down) && (m >= left) && (m >= right) && (m > a.thr);
        }
        emit_peak(is_peak, y - 1, c, win[(S + 24) % 25], list, count, a.PK);
    }
    st.s_prev2 = st.s_prev1;
    st.s_prev1 = s;
}

template <int SRC>
__device__ __forceinline__ void stream_plane(const SmoothArgs& a, double* ybuf, double* sbuf, int b, int part, int strip,
                                             const MultiSource* ms = nullptr, const float4* up8 = nullptr);

template <int SRC>
__global__ void __launch_bounds__(NTMAX)
k_smooth_nms_stream(SmoothArgs a) {
    __shared__ double ybuf[2 * (NTMAX + 2 * PD_GR)];
    __shared__ double sbuf[2 * (NTMAX + 2)];
    const int b = blockIdx.z, part = blockIdx.y, strip = blockIdx.x;
    if (a.only_over != nullptr && a.only_over[b * PD_N_PARTS + part] <= a.over_limit) return;
    stream_plane<SRC>(a, ybuf, sbuf, b, part, strip);
}

// The end of the filtered peak stage, one workgroup per image: planes whose candidate list
// overflowed (plateaus; rare) are redone here with the exact streaming pass, strip by strip,
// then the image's peaks are ordered.  Everything an image needs is written and read by its
// own workgroup, so barriers are all the synchronisation there is.
template <int SRC>
__device__ __forceinline__ void finish_peaks_body(const SmoothArgs& a, PD_MULTI_REF) {
    __shared__ double ybuf[2 * (NTMAX + 2 * PD_GR)];
    __shared__ double sbuf[2 * (NTMAX + 2)];
    __shared__ SortShared s_sort;
    const int b = blockIdx.x;
    for (int part = 0; part < PD_N_PARTS; ++part) {
        if (a.only_over[b * PD_N_PARTS + part] <= a.over_limit) continue;  // uniform over the workgroup
        for (int strip = 0; strip < a.n_strips; ++strip) {
            __syncthreads();
            stream_plane<SRC>(a, ybuf, sbuf, b, part, strip, ms, up8);
        }
    }
    __syncthreads();
    pd_sort_image(a.sort, s_sort, b, threadIdx.x, blockDim.x);
}

template <int SRC>
__global__ void __launch_bounds__(NTMAX)
k_finish_peaks(SmoothArgs a) {
    PD_RAISE_PRIORITY(1);
    finish_peaks_body<SRC>(a, nullptr, nullptr);
}

__global__ void __launch_bounds__(NTMAX)
k_finish_peaks_ms(SmoothArgs a, const MultiSource* __restrict__ ms) {
    PD_RAISE_PRIORITY(1);
    __shared__ float4 s_up8[8];
    if (threadIdx.x < 8) {
        const float* c = a.src.up8.c[threadIdx.x];
        s_up8[threadIdx.x] = make_float4(c[0], c[1], c[2], c[3]);
    }
    __syncthreads();
    finish_peaks_body<PD_SRC_MULTI>(a, ms, s_up8);
}

template <int SRC>
__device__ __forceinline__ void stream_plane(const SmoothArgs& a, double* ybuf, double* sbuf, int b, int part, int strip,
                                             PD_MULTI_REF) {
    const int H = a.src.H, W = a.src.W;
    const int x0 = strip * a.strip_w;
    const int x1 = min(W, x0 + a.strip_w);
    const int xs = max(0, x0 - (PD_GR + 1));
    const int xe = min(W, x1 + (PD_GR + 1));
    const int tid = threadIdx.x;
    const int c = xs + tid;
    const bool col_active = c < xe;
    // columns whose smoothed value is needed: the strip plus one neighbour each side
    const bool do_x = col_active && c >= x0 - 1 && c <= x1;
    const bool do_nms = col_active && c >= x0 && c < x1;
    RawPeak* list = a.raw + ((size_t)b * PD_N_PARTS + part) * a.PK;
    int32_t* count = a.raw_count + b * PD_N_PARTS + part;
    double* dbg_plane = a.dbg ? a.dbg + ((size_t)b * PD_N_PARTS + part) * (size_t)H * W : nullptr;

    ColLoader<SRC> ld;
    loader_attach<SRC>(ld, ms, up8);
    if (col_active) ld.init(a.src, b, part, c);
    double win[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) win[i] = 0.0;
    if (col_active) {
        // rows 0..11 fill slots 0..11 and, mirrored, rows -1..-12 -> slots 24..13
#pragma unroll
        for (int r = 0; r < PD_GR; ++r) {
            const double v = ld.load(r);
            win[r] = v;
            win[24 - r] = v;
        }
    }
    RowState st;
    st.s_prev1 = 0.0;
    st.s_prev2 = 0.0;
    sbuf[tid + 1] = 0.0;
    sbuf[NTMAX + 2 + tid + 1] = 0.0;

#define PD_STEP(S)                                                                                    \
    if (y0 + S > H) break;                                                                            \
    row_step<SRC, S>(a, ld, win, st, ybuf, sbuf, y0 + S, H, W, c, xs, xe, col_active, do_x, do_nms,   \
                     list, count, dbg_plane);
    for (int y0 = 0; y0 <= H; y0 += 25) {
        PD_STEP(0) PD_STEP(1) PD_STEP(2) PD_STEP(3) PD_STEP(4) PD_STEP(5) PD_STEP(6) PD_STEP(7)
        PD_STEP(8) PD_STEP(9) PD_STEP(10) PD_STEP(11) PD_STEP(12) PD_STEP(13) PD_STEP(14)
        PD_STEP(15) PD_STEP(16) PD_STEP(17) PD_STEP(18) PD_STEP(19) PD_STEP(20) PD_STEP(21)
        PD_STEP(22) PD_STEP(23) PD_STEP(24)
    }
#undef PD_STEP
}

// ---------------------------------------------------------------------------------
// Simple kernels: one thread per pixel, any image size (also images smaller than
// the filter radius, where the reflection wraps more than once).
// ---------------------------------------------------------------------------------
__global__ void k_materialize(MapSource s, int nch, double* out) {
    // out (B, nch, H, W) float64 = source channel planes
    const size_t n = (size_t)s.H * s.W;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int ch = blockIdx.y, b = blockIdx.z;
    if (i >= n) return;
    const int Y = (int)(i / s.W), X = (int)(i % s.W);
    double v;
    if (s.kind == PD_SRC_LOWRES)
        v = (double)pd_bicubic_at(s.lo + (size_t)b * s.h * s.w * s.C, s.h, s.w, s.C, ch, Y, X, s.ty, s.tx);
    else
        v = s.full[((size_t)b * s.Cfull + ch) * n + i];
    out[((size_t)b * nch + ch) * n + i] = v;
}

__global__ void k_gauss_axis(const double* in, double* out, int H, int W, int axis, int f32mode) {
    const size_t n = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* p = in + (size_t)blockIdx.y * n;
    const int Y = (int)(i / W), X = (int)(i % W);
    double t = p[i] * c_gw[12];
    for (int d = 12; d >= 1; --d) {
        double lo, hi;
        if (axis == 0) {
            lo = p[(size_t)pd_reflect(Y - d, H) * W + X];
            hi = p[(size_t)pd_reflect(Y + d, H) * W + X];
        } else {
            lo = p[(size_t)Y * W + pd_reflect(X - d, W)];
            hi = p[(size_t)Y * W + pd_reflect(X + d, W)];
        }
        t += (lo + hi) * c_gw[12 - d];
    }
    out[(size_t)blockIdx.y * n + i] = f32mode ? (double)(float)t : t;
}

__global__ void k_nms_simple(const double* sm, const double* ori, int H, int W, double thr, RawPeak* raw,
                             int32_t* raw_count, int PK) {
    const size_t n = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int plane = blockIdx.y;  // b * 18 + part
    bool is_peak = false;
    int Y = 0, X = 0;
    double score = 0.0;
    if (i < n) {
        const double* p = sm + (size_t)plane * n;
        Y = (int)(i / W);
        X = (int)(i % W);
        const double m = p[i];
        const double up = Y > 0 ? p[i - W] : 0.0;
        const double down = Y < H - 1 ? p[i + W] : 0.0;
        const double left = X > 0 ? p[i - 1] : 0.0;
        const double right = X < W - 1 ? p[i + 1] : 0.0;
        is_peak = (m >= up) && (m >= down) && (m >= left) && (m >= right) && (m > thr);
        score = ori[(size_t)plane * n + i];
    }
    emit_peak(is_peak, Y, X, score, raw + (size_t)plane * PK, raw_count + plane, PK);
}

// Column strips of a plane: workgroups of at most four wavefronts (a row step is a chain of
// dependent float64 sums, and narrow workgroups pack onto a CU where one wide one leaves its
// registers idle: two strips of 256 threads take 1.22 ms at 368^2, batch 64, one of 384
// threads 1.76 ms, although the strips redo 26 halo columns each), and among those the split
// that wastes the fewest threads.
#define NT_PREFERRED 256
static void pick_strips(int W, int* n_strips, int* strip_w, int* nt) {
    int best_n = 1, best_w = W, best_nt = 0;
    double best_eff = -1.0;
    for (int n = 1; n <= 64; ++n) {
        const int sw = (W + n - 1) / n;
        const int cols = n == 1 ? W : sw + 2 * (PD_GR + 1);
        const int t = ((cols + 63) / 64) * 64;
        if (t > NT_PREFERRED && n < 64) continue;
        if (t > NTMAX) continue;
        const double eff = (double)W / ((double)n * t);
        if (eff > best_eff + 1e-9) {
            best_eff = eff;
            best_n = n;
            best_w = sw;
            best_nt = t;
        }
    }
    *n_strips = best_n;
    *strip_w = best_w;
    *nt = best_nt;
}

// The exact streaming kernel over all planes, or (only_over != NULL) over the planes
// whose candidate count exceeded over_limit.
int pd_launch_stream_exact(pd_ctx* ctx, double thr, int f32mode, double* dbg_smoothed,
                           const int32_t* only_over, int over_limit, int finish) {
    MapSource src = pd_heat_source(ctx);
    SmoothArgs a;
    a.src = src;
    a.thr = thr;
    a.f32mode = f32mode;
    int nt;
    pick_strips(ctx->W, &a.n_strips, &a.strip_w, &nt);
    if (nt == 0) PD_FAIL(ctx, PD_E_INVALID, "no strip geometry for W=%d", ctx->W);
    a.raw = ctx->raw;
    a.raw_count = ctx->raw_count;
    a.PK = ctx->cfg.max_peaks_per_part;
    a.dbg = dbg_smoothed;
    a.only_over = only_over;
    a.over_limit = over_limit;
    memset(&a.sort, 0, sizeof(a.sort));
    if (src.kind == PD_SRC_MULTI) {
        // only as the end of the sparse multi-scale peak stage (overflowing planes + ordering)
        if (!(finish && only_over != nullptr && dbg_smoothed == nullptr)) PD_FAIL(ctx, PD_E_STATE, "the exact kernel over every pixel needs stored maps");
        pd_fill_sort_args(ctx, &a.sort);
        ctx->peaks_sorted = 1;
        pd_prof_begin(ctx, PD_STAGE_SMOOTH_NMS);
        hipLaunchKernelGGL(k_finish_peaks_ms, dim3(ctx->B), dim3(nt), 0, ctx->stream, a, (const MultiSource*)ctx->ms_dev);
        pd_prof_end(ctx);
        PD_HIP(ctx, hipGetLastError());
        return PD_OK;
    }
    static const bool separate_sort = getenv("PD_SEPARATE_SORT") != nullptr;  // comparison runs
    if (finish && only_over != nullptr && dbg_smoothed == nullptr && !separate_sort) {
        pd_fill_sort_args(ctx, &a.sort);
        ctx->peaks_sorted = 1;
        pd_prof_begin(ctx, PD_STAGE_SMOOTH_NMS);
        if (src.kind == PD_SRC_LOWRES)
            hipLaunchKernelGGL(k_finish_peaks<PD_SRC_LOWRES>, dim3(ctx->B), dim3(nt), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL(k_finish_peaks<PD_SRC_FULLRES>, dim3(ctx->B), dim3(nt), 0, ctx->stream, a);
        pd_prof_end(ctx);
        PD_HIP(ctx, hipGetLastError());
        return PD_OK;
    }
    dim3 grid(a.n_strips, PD_N_PARTS, ctx->B);
    pd_prof_begin(ctx, PD_STAGE_SMOOTH_NMS);
    if (src.kind == PD_SRC_LOWRES)
        hipLaunchKernelGGL(k_smooth_nms_stream<PD_SRC_LOWRES>, grid, dim3(nt), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(k_smooth_nms_stream<PD_SRC_FULLRES>, grid, dim3(nt), 0, ctx->stream, a);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    return PD_OK;
}

int pd_launch_smooth_nms(pd_ctx* ctx, double thre1, double* dbg_smoothed) {
    const int B = ctx->B, H = ctx->H, W = ctx->W;
    // raw_count and the filter's counters start at zero; the pack kernel of the previous
    // decode has normally seen to that
    if (!ctx->zero_clean) {
        int rc = pd_launch_clear(ctx, ctx->zero_block, ctx->zero_bytes / sizeof(int32_t));
        if (rc != PD_OK) return rc;
    }
    ctx->zero_clean = 0;
    const int f32mode = (ctx->src_kind == PD_SRC_FULLRES) ? ctx->heat_f32 : 0;
    // float32 data is thresholded in float32 (numpy compares a float32 array with a
    // Python float in float32)
    const double thr = f32mode ? (double)(float)thre1 : thre1;
    const bool small = H < 2 * PD_GR + 2 || W < 2 * PD_GR + 2;
    if (ctx->src_kind == PD_SRC_MULTI && !ctx->ms_heat_full) {
        // the scales of a multi-scale source: candidate cells, float32 filter and exact verification
        // straight from the network outputs; where that form does not apply the average of the
        // heatmaps is written out once and the stage runs on it as on any full-resolution map
        if (ctx->impl == PD_IMPL_FAST && !small && dbg_smoothed == nullptr) {
            int rc = pd_launch_peaks_filtered_multi(ctx, thr);
            if (rc != PD_E_STATE) return rc;
        }
        int rc = pd_ms_materialise_heat(ctx);
        if (rc != PD_OK) return rc;
    }
    MapSource src = pd_heat_source(ctx);
    if (ctx->impl == PD_IMPL_FAST && !small && src.kind == PD_SRC_LOWRES && dbg_smoothed == nullptr) {
        // float32 candidate filter + exact verification; falls through to the
        // streaming kernel when the geometry has no filter tables
        int rc = pd_launch_peaks_filtered(ctx, thr);
        if (rc != PD_E_STATE) return rc;
    }
    if (ctx->impl == PD_IMPL_FAST && !small && src.kind == PD_SRC_FULLRES && !f32mode && ctx->full_absmax_ok &&
        dbg_smoothed == nullptr) {
        // the same scheme on the float64 planes the context holds
        int rc = pd_launch_peaks_filtered_full(ctx, thr);
        if (rc != PD_E_STATE) return rc;
    }
    if (ctx->impl != PD_IMPL_SIMPLE && !small) return pd_launch_stream_exact(ctx, thr, f32mode, dbg_smoothed, nullptr, 0);
    // simple path: materialise the 18 planes, filter axis 0 then axis 1, NMS
    const size_t n = (size_t)H * W;
    const size_t need = (size_t)B * PD_N_PARTS * n;
    if (ctx->tmp_smooth_cap < need) {
        PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->gen++;
        if (ctx->tmp_smooth) PD_HIP(ctx, hipFree(ctx->tmp_smooth));
        ctx->tmp_smooth = nullptr;
        ctx->tmp_smooth_cap = 0;
        PD_HIP(ctx, hipMalloc(&ctx->tmp_smooth, need * sizeof(double) * 2));
        ctx->tmp_smooth_cap = need;
    }
    if (ctx->tmp_smooth2_cap < need) {
        PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->gen++;
        if (ctx->tmp_smooth2) PD_HIP(ctx, hipFree(ctx->tmp_smooth2));
        ctx->tmp_smooth2 = nullptr;
        ctx->tmp_smooth2_cap = 0;
        PD_HIP(ctx, hipMalloc(&ctx->tmp_smooth2, need * sizeof(double)));
        ctx->tmp_smooth2_cap = need;
    }
    double* ori = ctx->tmp_smooth;         // (B,18,H,W)
    double* t0 = ctx->tmp_smooth + need;   // axis-0 result
    double* t1 = dbg_smoothed ? dbg_smoothed : ctx->tmp_smooth2;
    const int bs = 256;
    const unsigned gx = (unsigned)((n + bs - 1) / bs);
    pd_prof_begin(ctx, PD_STAGE_SMOOTH_NMS);
    hipLaunchKernelGGL(k_materialize, dim3(gx, PD_N_PARTS, B), dim3(bs), 0, ctx->stream, src, PD_N_PARTS, ori);
    hipLaunchKernelGGL(k_gauss_axis, dim3(gx, B * PD_N_PARTS), dim3(bs), 0, ctx->stream, ori, t0, H, W, 0, f32mode);
    hipLaunchKernelGGL(k_gauss_axis, dim3(gx, B * PD_N_PARTS), dim3(bs), 0, ctx->stream, t0, t1, H, W, 1, f32mode);
    hipLaunchKernelGGL(k_nms_simple, dim3(gx, B * PD_N_PARTS), dim3(bs), 0, ctx->stream, t1, ori, H, W, thr,
                       ctx->raw, ctx->raw_count, ctx->cfg.max_peaks_per_part);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    return PD_OK;
}
