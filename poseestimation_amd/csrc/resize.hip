// Bicubic resize with OpenCV INTER_CUBIC semantics (cv2.resize calls at
// predict.py:63,65,67,69) and the multi-scale average of predict.py:71-72, plus the
// layout conversions between caller arrays (B, H, W, C) and the planar float64
// planes the other stages read.
//
// Arithmetic contract: source coordinate s = (d + 0.5) * scale - 0.5 evaluated in
// double and cast to float; taps floor(s) - 1 .. floor(s) + 2 clamped to the image;
// Keys weights with A = -0.75 evaluated in float32; horizontal pass first, then
// vertical, each 4-tap sum ((s0*c0 + s1*c1) + s2*c2) + s3*c3 in float32 with
// separately rounded multiply and add; equal source and destination size = copy.
#include <math.h>

#include <vector>

#include "common.h"

#pragma clang fp contract(off)

void pd_host_cubic_table(int src_len, int dst_len, double scale, int32_t* ofs, float* coef) {
    if (scale <= 0.0) scale = 1.0 / ((double)dst_len / (double)src_len);
    const float A = -0.75f;
    for (int d = 0; d < dst_len; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        const int s = (int)floorf(f);
        const float t = f - (float)s;
        const float t1 = t + 1.0f;
        const float u = 1.0f - t;
        const float c0 = ((A * t1 - 5.0f * A) * t1 + 8.0f * A) * t1 - 4.0f * A;
        const float c1 = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
        const float c2 = ((A + 2.0f) * u - (A + 3.0f)) * u * u + 1.0f;
        const float c3 = 1.0f - c0 - c1 - c2;
        ofs[d] = s - 1;
        coef[4 * d + 0] = c0;
        coef[4 * d + 1] = c1;
        coef[4 * d + 2] = c2;
        coef[4 * d + 3] = c3;
    }
}

// Weights of the x8 table by phase: destination indices 4..11 have phases 0..7.
void pd_up8_fill(Up8* u) {
    int32_t ofs[16];
    float coef[64];
    pd_host_cubic_table(2, 16, 0.125, ofs, coef);
    for (int p = 0; p < 8; ++p)
        for (int k = 0; k < 4; ++k) u->c[p][k] = coef[4 * (4 + p) + k];
}

// Device tables: ints [y ofs | x ofs], floats [y coef | x coef].
static int upload_tables(pd_ctx* ctx, int hs, int ws, int Hd, int Wd, double sy, double sx,
                         TableSlot* slot) {
    int32_t** d_i = &slot->i;
    float** d_f = &slot->f;
    size_t* cap = &slot->cap;
    CubicTable* ty = &slot->ty;
    CubicTable* tx = &slot->tx;
    const size_t need = (size_t)Hd + Wd;
    if (slot->i && slot->key[0] == hs && slot->key[1] == ws && slot->key[2] == Hd && slot->key[3] == Wd &&
        slot->scale[0] == sy && slot->scale[1] == sx)
        return PD_OK;  // still valid
    ctx->gen++;  // the tables of this slot are rewritten (and may move)
    if (*cap < need) {
        if (*d_i) PD_HIP(ctx, hipFree(*d_i));
        if (*d_f) PD_HIP(ctx, hipFree(*d_f));
        *d_i = nullptr;
        *d_f = nullptr;
        *cap = 0;
        PD_HIP(ctx, hipMalloc(d_i, need * sizeof(int32_t)));
        PD_HIP(ctx, hipMalloc(d_f, need * 4 * sizeof(float)));
        *cap = need;
    }
    std::vector<int32_t> hi(need);
    std::vector<float> hf(need * 4);
    pd_host_cubic_table(hs, Hd, sy, hi.data(), hf.data());
    pd_host_cubic_table(ws, Wd, sx, hi.data() + Hd, hf.data() + 4 * (size_t)Hd);
    // the copies are stream-ordered behind kernels that still read the old tables
    PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PD_HIP(ctx, hipMemcpy(*d_i, hi.data(), need * sizeof(int32_t), hipMemcpyHostToDevice));
    PD_HIP(ctx, hipMemcpy(*d_f, hf.data(), need * 4 * sizeof(float), hipMemcpyHostToDevice));
    ty->ofs = *d_i;
    ty->coef = *d_f;
    tx->ofs = *d_i + Hd;
    tx->coef = *d_f + 4 * (size_t)Hd;
    slot->key[0] = hs;
    slot->key[1] = ws;
    slot->key[2] = Hd;
    slot->key[3] = Wd;
    slot->scale[0] = sy;
    slot->scale[1] = sx;
    return PD_OK;
}

// x8 tables for the fused low-resolution source, cached per geometry.
int pd_build_tables(pd_ctx* ctx, int h, int w, int H, int W) {
    if (ctx->tab_h == h && ctx->tab_w == w && ctx->tab_H == H && ctx->tab_W == W && ctx->tabs[0].i) return PD_OK;
    // cv2.resize(.., (0,0), fx=8, fy=8): scale = 1/8; rows/cols beyond the crop are never read
    int rc = upload_tables(ctx, h, w, H, W, 0.125, 0.125, &ctx->tabs[0]);
    if (rc != PD_OK) return rc;
    ctx->tab_h = h;
    ctx->tab_w = w;
    ctx->tab_H = H;
    ctx->tab_W = W;
    return PD_OK;
}

__global__ void k_hwc_to_planar(const void* __restrict__ src, int dtype, int Csrc, int Cdst, int H, int W,
                                double* __restrict__ dst) {
    const size_t n = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ch = blockIdx.y, b = blockIdx.z;
    const size_t s = ((size_t)b * n + i) * Csrc + ch;
    const double v = dtype == PD_F32 ? (double)((const float*)src)[s] : ((const double*)src)[s];
    dst[((size_t)b * Cdst + ch) * n + i] = v;
}

__global__ void k_planar_to_hwc(const double* __restrict__ src, int C, int H, int W, double* __restrict__ dst) {
    const size_t n = (size_t)H * W * C;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const int ch = (int)(i % C);
    const size_t px = i / C;
    dst[(size_t)b * n + i] = src[((size_t)b * C + ch) * (size_t)H * W + px];
}

// ---- x8 upsample, eight output rows per thread ------------------------------------------
// Output rows 8j-4 .. 8j+3 draw on the same four source rows (j-2 .. j+1, clamped) and
// differ only in the phase of their weights, so a thread forms the four horizontal sums of
// its column once and combines them eight times.  Per output: the horizontal 4-tap sum of each source row, then the vertical 4-tap sum,
// float32, multiplies and adds rounded separately.
__global__ void __launch_bounds__(256)
k_resize_up8(const float* __restrict__ src, int C, int hs, int ws, Up8 up, float* __restrict__ dst) {
    __shared__ float4 s_up[8];
    if (threadIdx.x < 8) s_up[threadIdx.x] = make_float4(up.c[threadIdx.x][0], up.c[threadIdx.x][1], up.c[threadIdx.x][2], up.c[threadIdx.x][3]);
    __syncthreads();
    const int Hd = 8 * hs, Wd = 8 * ws;
    const int X = blockIdx.x * 256 + threadIdx.x;
    if (X >= Wd) return;
    const int j = blockIdx.y;                 // 0 .. hs
    const int ch = blockIdx.z % C, b = blockIdx.z / C;
    const int x0 = pd_up8_ofs(X);
    const float4 ac = s_up[pd_up8_phase(X)];
    const int xi0 = pd_clampi(x0, 0, ws - 1) * C, xi1 = pd_clampi(x0 + 1, 0, ws - 1) * C,
              xi2 = pd_clampi(x0 + 2, 0, ws - 1) * C, xi3 = pd_clampi(x0 + 3, 0, ws - 1) * C;
    const float* img = src + (size_t)b * hs * ws * C + ch;
    float hsum[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* row = img + (size_t)pd_clampi(j - 2 + k, 0, hs - 1) * ws * C;
        hsum[k] = ((row[xi0] * ac.x + row[xi1] * ac.y) + row[xi2] * ac.z) + row[xi3] * ac.w;
    }
    float* out = dst + ((size_t)b * C + ch) * (size_t)Hd * Wd + X;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int d = 8 * j - 4 + p;
        if (d >= 0 && d < Hd)
            out[(size_t)d * Wd] = ((hsum[0] * up.c[p][0] + hsum[1] * up.c[p][1]) + hsum[2] * up.c[p][2]) + hsum[3] * up.c[p][3];
    }
}

// ---- second resize of every scale + average, one pass -------------------------------------
// out = ((0 + v_0 / n) + v_1 / n) + ..  with v_m the crop of scale m's x8 map resized to
// (H, W) (predict.py:64-65, 71-72): float32 division, float64 accumulation in scale order.
struct ScaleSrc {
    const float* up;   // planar (B, C, Hs, Ws) float32
    int Hs, Ws, crop_h, crop_w, identity;
    CubicTable ty, tx;
};
struct AvgArgs {
    ScaleSrc s[PD_MAX_SCALES];
    int n_scales, C, H, W;
    float n_div, n_rcp;   // number of scales; its reciprocal when that is exact (a power of two)
    double* dst;          // planar (B, C, H, W)
};

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
#ifndef AVG_TW
#define AVG_TW 32       // output tile of a workgroup: AVG_TW x AVG_TH = 256 pixels
#endif
#define AVG_TH (256 / AVG_TW)
#ifndef AVG_CH
#define AVG_CH 4        // channels per workgroup: taps and weights of a pixel are set up once per scale for all of
#endif                  // them; with 8 the kernel needs 182 registers and two workgroups per CU cannot hide its loads
#ifndef AVG_SLOTS
#define AVG_SLOTS 6     // patch elements per thread and channel
#endif
#define AVG_PATCH (AVG_SLOTS * 256)

// What a scale contributes to the workgroup's 32 x 8 output tile, fetched ahead of its use.
// Patch form: the part of the x8 map the tile's taps fall in, rows r_lo .. r_lo+ph-1 and
// columns c_lo .. c_lo+pw-1, read with coalesced loads (element e = row * pw + col, thread
// tid takes e = tid + 256 s) and handed to the other threads through LDS.  An identity
// scale is the thread's own pixel; a patch beyond AVG_PATCH elements (a downscale by more
// than 2.1) is not staged: its taps are gathered from memory.
struct AvgFetch {
    int kind;  // 0 identity, 1 patch, 2 gather
    int r_lo, c_lo, ph, pw;
    float v[AVG_SLOTS][AVG_CH];
};

__device__ __forceinline__ void avg_fetch(const AvgArgs& a, int m, int b, int ch0, int nch, int X0, int Y0, int X, int Y,
                                          bool inside, int tid, AvgFetch& f) {
    const ScaleSrc& sc = a.s[m];
    const size_t plane = (size_t)sc.Hs * sc.Ws;
    const float* p = sc.up + ((size_t)b * a.C + ch0) * plane;
    if (sc.identity) {
        f.kind = 0;
        const size_t o = (size_t)Y * sc.Ws + X;
#pragma unroll
        for (int c = 0; c < AVG_CH; ++c) f.v[0][c] = (inside && c < nch) ? p[c * plane + o] : 0.f;
        return;
    }
    // the offsets grow with the output coordinate: the tile's first and last give the range
    const int y_first = sc.ty.ofs[Y0], y_last = sc.ty.ofs[min(Y0 + AVG_TH - 1, a.H - 1)];
    const int x_first = sc.tx.ofs[X0], x_last = sc.tx.ofs[min(X0 + AVG_TW - 1, a.W - 1)];
    f.r_lo = pd_clampi(y_first, 0, sc.crop_h - 1);
    f.c_lo = pd_clampi(x_first, 0, sc.crop_w - 1);
    f.ph = pd_clampi(y_last + 3, 0, sc.crop_h - 1) - f.r_lo + 1;
    f.pw = pd_clampi(x_last + 3, 0, sc.crop_w - 1) - f.c_lo + 1;
    if (f.ph * f.pw > AVG_PATCH) {
        f.kind = 2;
        return;
    }
    f.kind = 1;
    const int pe = f.ph * f.pw;
#pragma unroll
    for (int s = 0; s < AVG_SLOTS; ++s) {
        const int e = tid + 256 * s;
        const int row = e / f.pw, col = e - row * f.pw;
        const float* g = p + (size_t)(f.r_lo + row) * sc.Ws + f.c_lo + col;
#pragma unroll
        for (int c = 0; c < AVG_CH; ++c) f.v[s][c] = (e < pe && c < nch) ? g[c * plane] : 0.f;
    }
}

// v / n as the reference's float32 division rounds it; for a power of two the product with
// the reciprocal is the same number.
template <bool POW2>
__device__ __forceinline__ float avg_share(float v, const AvgArgs& a) {
    return POW2 ? v * a.n_rcp : v / a.n_div;
}

template <bool POW2>
__global__ void __launch_bounds__(256)
k_resize_avg(AvgArgs a) {
    __shared__ __align__(16) float patch[AVG_CH][AVG_PATCH];
    const size_t n = (size_t)a.H * a.W;
    // Workgroup i runs on XCD i % 8, each with an L2 of its own: XCD x takes the x-th eighth
    // of the tiles in row-major order, so that tiles sharing rows and lines share an L2.
    const int tiles_x = (a.W + AVG_TW - 1) / AVG_TW, tiles = tiles_x * ((a.H + AVG_TH - 1) / AVG_TH);
    const int per_xcd = (tiles + 7) / 8;
    const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (tile >= tiles) return;
    const int tid = threadIdx.x;
    const int X0 = (tile % tiles_x) * AVG_TW, Y0 = (tile / tiles_x) * AVG_TH;
    const bool inside = X0 + (tid % AVG_TW) < a.W && Y0 + (tid / AVG_TW) < a.H;
    // threads beyond the map keep step with the others (they load and stage, and store nothing)
    const int X = min(X0 + (tid % AVG_TW), a.W - 1), Y = min(Y0 + (tid / AVG_TW), a.H - 1);
    const size_t i = (size_t)Y * a.W + X;
    const int ch0 = blockIdx.y * AVG_CH, b = blockIdx.z;
    const int nch = min(AVG_CH, a.C - ch0);
    double acc[AVG_CH];
#pragma unroll
    for (int c = 0; c < AVG_CH; ++c) acc[c] = 0.0;
    AvgFetch f;
    avg_fetch(a, 0, b, ch0, nch, X0, Y0, X, Y, inside, tid, f);
    for (int m = 0; m < a.n_scales; ++m) {
        const ScaleSrc& sc = a.s[m];
        const int kind = f.kind, r_lo = f.r_lo, c_lo = f.c_lo, pw = f.pw;
        if (kind == 0) {
#pragma unroll
            for (int c = 0; c < AVG_CH; ++c)
                if (c < nch) acc[c] = acc[c] + (double)avg_share<POW2>(f.v[0][c], a);
            if (m + 1 < a.n_scales) avg_fetch(a, m + 1, b, ch0, nch, X0, Y0, X, Y, inside, tid, f);
            continue;
        }
        if (kind == 1) {
            const int pe = f.ph * f.pw;
#pragma unroll
            for (int s = 0; s < AVG_SLOTS; ++s) {
                const int e = tid + 256 * s;
                if (e < pe) {
#pragma unroll
                    for (int c = 0; c < AVG_CH; ++c) patch[c][e] = f.v[s][c];
                }
            }
        }
        __syncthreads();
        // the next scale's loads are in flight while this one is evaluated
        if (m + 1 < a.n_scales) avg_fetch(a, m + 1, b, ch0, nch, X0, Y0, X, Y, inside, tid, f);
        const int x0 = sc.tx.ofs[X], y0 = sc.ty.ofs[Y];
        const float4 ac = *(const float4*)(sc.tx.coef + 4 * (size_t)X);
        const float4 bc = *(const float4*)(sc.ty.coef + 4 * (size_t)Y);
        const int xi0 = pd_clampi(x0, 0, sc.crop_w - 1), xi1 = pd_clampi(x0 + 1, 0, sc.crop_w - 1),
                  xi2 = pd_clampi(x0 + 2, 0, sc.crop_w - 1), xi3 = pd_clampi(x0 + 3, 0, sc.crop_w - 1);
        // away from the left and right border the four taps are adjacent
        const bool adjacent = x0 >= 0 && x0 + 3 <= sc.crop_w - 1;
        if (kind == 1) {
            int ro[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) ro[k] = (pd_clampi(y0 + k, 0, sc.crop_h - 1) - r_lo) * pw - c_lo;
#pragma unroll
            for (int c = 0; c < AVG_CH; ++c) {
                if (c < nch) {
                    const float* q = patch[c];
                    float hv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float* row = q + ro[k];
                        if (adjacent) {
                            const float4u t = *(const float4u*)(row + x0);
                            hv[k] = ((t.x * ac.x + t.y * ac.y) + t.z * ac.z) + t.w * ac.w;
                        } else {
                            hv[k] = ((row[xi0] * ac.x + row[xi1] * ac.y) + row[xi2] * ac.z) + row[xi3] * ac.w;
                        }
                    }
                    const float v = ((hv[0] * bc.x + hv[1] * bc.y) + hv[2] * bc.z) + hv[3] * bc.w;
                    acc[c] = acc[c] + (double)avg_share<POW2>(v, a);
                }
            }
        } else {
            const size_t plane = (size_t)sc.Hs * sc.Ws;
            const float* p = sc.up + ((size_t)b * a.C + ch0) * plane;
            int ro[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) ro[k] = pd_clampi(y0 + k, 0, sc.crop_h - 1) * sc.Ws;
#pragma unroll
            for (int c = 0; c < AVG_CH; ++c) {
                if (c < nch) {
                    const float* q = p + c * plane;
                    float hv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float* row = q + ro[k];
                        if (adjacent) {
                            const float4u t = *(const float4u*)(row + x0);
                            hv[k] = ((t.x * ac.x + t.y * ac.y) + t.z * ac.z) + t.w * ac.w;
                        } else {
                            hv[k] = ((row[xi0] * ac.x + row[xi1] * ac.y) + row[xi2] * ac.z) + row[xi3] * ac.w;
                        }
                    }
                    const float v = ((hv[0] * bc.x + hv[1] * bc.y) + hv[2] * bc.z) + hv[3] * bc.w;
                    acc[c] = acc[c] + (double)avg_share<POW2>(v, a);
                }
            }
        }
        __syncthreads();  // the patch is free for the next scale
    }
    if (!inside) return;
    double* out = a.dst + ((size_t)b * a.C + ch0) * n + i;
#pragma unroll
    for (int c = 0; c < AVG_CH; ++c)
        if (c < nch) out[c * n] = acc[c];
}

// All scales of one map family (C = 19 or 38): srcs[m] is (B, hs[m], ws[m], C) on the
// device; up holds the x8 maps of all scales one after the other.
int pd_launch_upsample_avg(pd_ctx* ctx, int n_scales, const float* const* srcs, int C, int B, const int32_t* hs,
                           const int32_t* ws, const int32_t* crop_h, const int32_t* crop_w, float* up, double* dst,
                           int H, int W) {
    AvgArgs a;
    memset(&a, 0, sizeof(a));
    Up8 u8;
    pd_up8_fill(&u8);
    size_t off = 0;
    for (int m = 0; m < n_scales; ++m) {
        const int Hs = 8 * hs[m], Ws = 8 * ws[m];
        ScaleSrc& sc = a.s[m];
        sc.up = up + off;
        sc.Hs = Hs;
        sc.Ws = Ws;
        sc.crop_h = crop_h[m];
        sc.crop_w = crop_w[m];
        sc.identity = crop_h[m] == H && crop_w[m] == W;
        if (!sc.identity) {
            int rc = upload_tables(ctx, crop_h[m], crop_w[m], H, W, -1.0, -1.0, &ctx->tabs[2 + m]);
            if (rc != PD_OK) return rc;
            sc.ty = ctx->tabs[2 + m].ty;
            sc.tx = ctx->tabs[2 + m].tx;
        }
        pd_prof_begin(ctx, PD_STAGE_RESIZE);
        hipLaunchKernelGGL(k_resize_up8, dim3((Ws + 255) / 256, hs[m] + 1, C * B), dim3(256), 0, ctx->stream, srcs[m], C,
                           hs[m], ws[m], u8, up + off);
        pd_prof_end(ctx);
        off += (size_t)B * C * Hs * Ws;
    }
    a.n_scales = n_scales;
    a.C = C;
    a.H = H;
    a.W = W;
    a.n_div = (float)n_scales;
    a.dst = dst;
    const unsigned tiles = (unsigned)(((W + AVG_TW - 1) / AVG_TW) * ((H + AVG_TH - 1) / AVG_TH));
    pd_prof_begin(ctx, PD_STAGE_RESIZE);
    const dim3 grid((tiles + 7) / 8 * 8, (C + AVG_CH - 1) / AVG_CH, B);
    if ((n_scales & (n_scales - 1)) == 0) {
        a.n_rcp = 1.0f / (float)n_scales;
        hipLaunchKernelGGL(k_resize_avg<true>, grid, dim3(256), 0, ctx->stream, a);
    } else {
        hipLaunchKernelGGL(k_resize_avg<false>, grid, dim3(256), 0, ctx->stream, a);
    }
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    return PD_OK;
}

int pd_launch_hwc_to_planar(pd_ctx* ctx, const void* src, int dtype, int Csrc, int Cdst, int B, int H,
                            int W, double* dst) {
    const size_t n = (size_t)H * W;
    const int nch = Csrc < Cdst ? Csrc : Cdst;
    pd_prof_begin(ctx, PD_STAGE_RESIZE);
    hipLaunchKernelGGL(k_hwc_to_planar, dim3((unsigned)((n + 255) / 256), nch, B), dim3(256), 0, ctx->stream, src,
                       dtype, Csrc, Cdst, H, W, dst);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    return PD_OK;
}

int pd_launch_planar_to_hwc(pd_ctx* ctx, const double* src, int C, int B, int H, int W, double* dst) {
    const size_t n = (size_t)H * W * C;
    pd_prof_begin(ctx, PD_STAGE_RESIZE);
    hipLaunchKernelGGL(k_planar_to_hwc, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, ctx->stream, src, C, H,
                       W, dst);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    return PD_OK;
}
