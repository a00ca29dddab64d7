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

// ---- x8 upsample: (B, hs, ws, C) float32 -> planar (B, C, Hd, Wd) float32 ----------
__global__ void k_resize_up(const float* __restrict__ src, int C, int hs, int ws, CubicTable ty,
                            CubicTable tx, float* __restrict__ dst, int Hd, int Wd) {
    const size_t n = (size_t)Hd * Wd;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ch = blockIdx.y, b = blockIdx.z;
    const int Y = (int)(i / Wd), X = (int)(i % Wd);
    const float v = pd_bicubic_at(src + (size_t)b * hs * ws * C, hs, ws, C, ch, Y, X, ty, tx);
    dst[((size_t)b * C + ch) * n + i] = v;
}

// ---- crop + resize to (H, W) + divide by the number of scales + float64 accumulate ----
// src planar (B, C, Hs, Ws) float32, the crop window is its top-left crop_h x crop_w.
__global__ void k_resize_accum(const float* __restrict__ src, int C, int Hs, int Ws, int crop_h,
                               int crop_w, CubicTable ty, CubicTable tx, int identity,
                               double* __restrict__ dst, int H, int W, float n_scales, int first) {
    const size_t n = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ch = blockIdx.y, b = blockIdx.z;
    const float* p = src + ((size_t)b * C + ch) * (size_t)Hs * Ws;
    const int Y = (int)(i / W), X = (int)(i % W);
    float v;
    if (identity) {
        v = p[(size_t)Y * Ws + X];
    } else {
        const int x0 = tx.ofs[X], y0 = ty.ofs[Y];
        const float a0 = tx.coef[4 * X + 0], a1 = tx.coef[4 * X + 1], a2 = tx.coef[4 * X + 2],
                    a3 = tx.coef[4 * X + 3];
        const int xi0 = pd_clampi(x0, 0, crop_w - 1), xi1 = pd_clampi(x0 + 1, 0, crop_w - 1),
                  xi2 = pd_clampi(x0 + 2, 0, crop_w - 1), xi3 = pd_clampi(x0 + 3, 0, crop_w - 1);
        float hv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* row = p + (size_t)pd_clampi(y0 + k, 0, crop_h - 1) * Ws;
            hv[k] = ((row[xi0] * a0 + row[xi1] * a1) + row[xi2] * a2) + row[xi3] * a3;
        }
        v = ((hv[0] * ty.coef[4 * Y + 0] + hv[1] * ty.coef[4 * Y + 1]) + hv[2] * ty.coef[4 * Y + 2]) +
            hv[3] * ty.coef[4 * Y + 3];
    }
    const double add = (double)(v / n_scales);  // float32 division, predict.py:71-72
    double* q = dst + ((size_t)b * C + ch) * n + i;
    *q = first ? 0.0 + add : *q + add;
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

int pd_launch_resize_up8(pd_ctx* ctx, const float* src, int C, int B, int hs, int ws, float* dst) {
    const int Hd = 8 * hs, Wd = 8 * ws;
    int rc = upload_tables(ctx, hs, ws, Hd, Wd, 0.125, 0.125, &ctx->tabs[1]);
    if (rc != PD_OK) return rc;
    const CubicTable ty = ctx->tabs[1].ty, tx = ctx->tabs[1].tx;
    const size_t n = (size_t)Hd * Wd;
    pd_prof_begin(ctx, PD_STAGE_RESIZE);
    hipLaunchKernelGGL(k_resize_up, dim3((unsigned)((n + 255) / 256), C, B), dim3(256), 0, ctx->stream, src, C,
                       hs, ws, ty, tx, dst, Hd, Wd);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    return PD_OK;
}

int pd_launch_resize_accum(pd_ctx* ctx, const float* up, int C, int B, int Hs, int Ws, int crop_h,
                           int crop_w, double* dst, int H, int W, float n_scales, int first) {
    CubicTable ty = {nullptr, nullptr}, tx = {nullptr, nullptr};
    const int identity = (crop_h == H && crop_w == W);
    if (!identity) {
        int rc = upload_tables(ctx, crop_h, crop_w, H, W, -1.0, -1.0, &ctx->tabs[2]);
        if (rc != PD_OK) return rc;
        ty = ctx->tabs[2].ty;
        tx = ctx->tabs[2].tx;
    }
    const size_t n = (size_t)H * W;
    pd_prof_begin(ctx, PD_STAGE_RESIZE);
    hipLaunchKernelGGL(k_resize_accum, dim3((unsigned)((n + 255) / 256), C, B), dim3(256), 0, ctx->stream, up, C,
                       Hs, Ws, crop_h, crop_w, ty, tx, identity, dst, H, W, n_scales, first);
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
