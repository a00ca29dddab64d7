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

#include <algorithm>
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
int pd_upload_tables(pd_ctx* ctx, int hs, int ws, int Hd, int Wd, double sy, double sx, TableSlot* slot) {
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
    int rc = pd_upload_tables(ctx, h, w, H, W, 0.125, 0.125, &ctx->tabs[0]);
    if (rc != PD_OK) return rc;
    ctx->tab_h = h;
    ctx->tab_w = w;
    ctx->tab_H = H;
    ctx->tab_W = W;
    return PD_OK;
}

__global__ void k_hwc_to_planar(const void* __restrict__ src, int dtype, int Csrc, int Cdst, int H, int W,
                                double* __restrict__ dst, float* absmax) {
    const size_t n = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int ch = blockIdx.y, b = blockIdx.z;
    double v = 0.0;
    if (i < n) {
        const size_t s = ((size_t)b * n + i) * Csrc + ch;
        v = dtype == PD_F32 ? (double)((const float*)src)[s] : ((const double*)src)[s];
        dst[((size_t)b * Cdst + ch) * n + i] = v;
    }
    if (absmax) pd_wave_absmax((float)fabs(v), absmax + b);
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

// ---- x8 upsample + second resize of every scale + average, one pass ----------------------
// out = ((0 + v_0 / n) + v_1 / n) + ..  with v_m the crop of scale m's x8 map resized to
// (H, W) (predict.py:63-65, 71-72): float32 division, float64 accumulation in scale order.
// The x8 maps are never written to memory: a workgroup owns a 32 x 8 output tile of four
// channels and, per scale, builds the part of the x8 map its taps fall in (the "patch") in
// LDS straight from the network output, in the reference's order of operations --
//   A  horizontal 4-tap sums of the source rows the patch draws on        (cv2.resize x8, row pass)
//   B  vertical 4-tap sums of those: the patch of the x8 map               (cv2.resize x8, column pass)
//   C  per output pixel: 4 horizontal sums over the patch, then the vertical sum   (second resize)
// every sum ((s0*c0 + s1*c1) + s2*c2) + s3*c3 in float32 with separately rounded products.
struct ScaleSrc {
    const float* lo;   // (B, hs, ws, C) float32 network output of this scale
    int hs, ws, crop_h, crop_w, identity;
    CubicTable ty, tx; // second resize: crop -> (H, W)
};
struct AvgArgs {
    ScaleSrc s[PD_MAX_SCALES];
    int n_scales, C, H, W;
    float n_div, n_rcp;   // number of scales; its reciprocal when that is exact (a power of two)
    Up8 up8;
    double* dst;          // planar (B, C, H, W)
    float* absmax;        // [B] or NULL: receives max |output| per image (bits of a non-negative float)
};

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
#define FT_TW 16        // output tile of a workgroup: FT_TW x FT_TH = 256 pixels
#define FT_TH 16
#define FT_CH 4         // channels per workgroup: they are adjacent in the (h, w, C) source, one 16-byte load
#define FT_PATCH 1408   // x8-map pixels (four channels each) staged in LDS: a downscale up to about 2.1
#define FT_HSUM 704     // horizontal sums (four channels each) staged in LDS
#define FT_SRC 192      // source pixels (four channels each) staged in LDS

// floor(e / d) for 0 <= e, e * d < 2^20, as a multiplication: m = ceil(2^20 / d).
__device__ __forceinline__ int div_magic(int d) { return ((1 << 20) + d - 1) / d; }
__device__ __forceinline__ int div_by(int e, int m) { return (int)(((unsigned)e * (unsigned)m) >> 20); }

__device__ __forceinline__ float4 tap4v(const float4& s0, const float4& s1, const float4& s2, const float4& s3, const float4& c) {
    return make_float4(((s0.x * c.x + s1.x * c.y) + s2.x * c.z) + s3.x * c.w, ((s0.y * c.x + s1.y * c.y) + s2.y * c.z) + s3.y * c.w,
                       ((s0.z * c.x + s1.z * c.y) + s2.z * c.z) + s3.z * c.w, ((s0.w * c.x + s1.w * c.y) + s2.w * c.z) + s3.w * c.w);
}

// v / n as the reference's float32 division rounds it; for a power of two the product with
// the reciprocal is the same number.
template <bool POW2>
__device__ __forceinline__ float avg_share(float v, const AvgArgs& a) {
    return POW2 ? v * a.n_rcp : v / a.n_div;
}

// nch adjacent channels of one source pixel (the others zero).
__device__ __forceinline__ float4 load_channels(const float* p, int nch) {
    if (nch == FT_CH) {
        const float4u t = *(const float4u*)p;
        return make_float4(t.x, t.y, t.z, t.w);
    }
    float4 r = make_float4(p[0], 0.f, 0.f, 0.f);
    if (nch > 1) r.y = p[1];
    if (nch > 2) r.z = p[2];
    return r;
}

__device__ __forceinline__ float tap4(float s0, float s1, float s2, float s3, const float4& c) {
    return ((s0 * c.x + s1 * c.y) + s2 * c.z) + s3 * c.w;
}

// Horizontal sums of source row srow at x8-map column col, FT_CH channels.
__device__ __forceinline__ float4 hsum_at(const float* img, int ws, int C, int srow, int col, int nch, const float4* s_up) {
    const int x0 = pd_up8_ofs(col);
    const float4 w = s_up[pd_up8_phase(col)];
    const float* row = img + (size_t)srow * ws * C;
    const float4 v0 = load_channels(row + (size_t)pd_clampi(x0, 0, ws - 1) * C, nch);
    const float4 v1 = load_channels(row + (size_t)pd_clampi(x0 + 1, 0, ws - 1) * C, nch);
    const float4 v2 = load_channels(row + (size_t)pd_clampi(x0 + 2, 0, ws - 1) * C, nch);
    const float4 v3 = load_channels(row + (size_t)pd_clampi(x0 + 3, 0, ws - 1) * C, nch);
    return make_float4(tap4(v0.x, v1.x, v2.x, v3.x, w), tap4(v0.y, v1.y, v2.y, v3.y, w),
                       tap4(v0.z, v1.z, v2.z, v3.z, w), tap4(v0.w, v1.w, v2.w, v3.w, w));
}

// One value of the x8 map at (row, col), straight from the source (patches too large to stage).
__device__ __forceinline__ float4 up8_at(const float* img, int hs, int ws, int C, int row, int col, int nch, const float4* s_up) {
    const int y0 = pd_up8_ofs(row);
    const float4 w = s_up[pd_up8_phase(row)];
    const float4 h0 = hsum_at(img, ws, C, pd_clampi(y0, 0, hs - 1), col, nch, s_up);
    const float4 h1 = hsum_at(img, ws, C, pd_clampi(y0 + 1, 0, hs - 1), col, nch, s_up);
    const float4 h2 = hsum_at(img, ws, C, pd_clampi(y0 + 2, 0, hs - 1), col, nch, s_up);
    const float4 h3 = hsum_at(img, ws, C, pd_clampi(y0 + 3, 0, hs - 1), col, nch, s_up);
    return make_float4(tap4(h0.x, h1.x, h2.x, h3.x, w), tap4(h0.y, h1.y, h2.y, h3.y, w),
                       tap4(h0.z, h1.z, h2.z, h3.z, w), tap4(h0.w, h1.w, h2.w, h3.w, w));
}

// DIRECT: some scale's patches do not fit the staging area (decided on the host from the tap
// tables); the common kernel carries no code for that case, so that it stays within the
// registers of four workgroups per CU.
template <bool POW2, bool DIRECT>
__global__ void __launch_bounds__(256, DIRECT ? 1 : 4)
k_resize_fused(AvgArgs a) {
    // pixels of four channels each (a float4), so that every tap is one 16-byte LDS access
    __shared__ float4 patch[FT_PATCH];
    __shared__ float4 hsum[FT_HSUM];
    __shared__ float4 srcp[FT_SRC];
    __shared__ float4 s_up[8];
    const int tid = threadIdx.x;
    if (tid < 8) s_up[tid] = make_float4(a.up8.c[tid][0], a.up8.c[tid][1], a.up8.c[tid][2], a.up8.c[tid][3]);
    const size_t n = (size_t)a.H * a.W;
    // Workgroup i runs on XCD i % 8, each with an L2 of its own: XCD x takes the x-th eighth
    // of the tiles in row-major order, so that tiles sharing source rows share an L2.
    const int tiles_x = (a.W + FT_TW - 1) / FT_TW, tiles = tiles_x * ((a.H + FT_TH - 1) / FT_TH);
    const int per_xcd = (tiles + 7) / 8;
    const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (tile >= tiles) return;
    const int X0 = (tile % tiles_x) * FT_TW, Y0 = (tile / tiles_x) * FT_TH;
    const bool inside = X0 + (tid % FT_TW) < a.W && Y0 + (tid / FT_TW) < a.H;
    // threads beyond the map keep step with the others (they help build the patch and store nothing)
    const int X = min(X0 + (tid % FT_TW), a.W - 1), Y = min(Y0 + (tid / FT_TW), a.H - 1);
    const int Ylast = min(Y0 + FT_TH - 1, a.H - 1), Xlast = min(X0 + FT_TW - 1, a.W - 1);
    const int ch0 = blockIdx.y * FT_CH, b = blockIdx.z;
    const int nch = min(FT_CH, a.C - ch0);
    double acc[FT_CH];
#pragma unroll
    for (int c = 0; c < FT_CH; ++c) acc[c] = 0.0;
    __syncthreads();
    for (int m = 0; m < a.n_scales; ++m) {
        const ScaleSrc& sc = a.s[m];
        const float* img = sc.lo + (size_t)b * sc.hs * sc.ws * a.C + ch0;
        // the part of the x8 map this tile's taps fall in: rows r_lo .. r_lo+ph-1, columns
        // c_lo .. c_lo+pw-1 (tap offsets grow with the output coordinate)
        int r_lo, c_lo, ph, pw;
        if (sc.identity) {
            r_lo = Y0;
            c_lo = X0;
            ph = Ylast - Y0 + 1;
            pw = Xlast - X0 + 1;
        } else {
            r_lo = pd_clampi(sc.ty.ofs[Y0], 0, sc.crop_h - 1);
            c_lo = pd_clampi(sc.tx.ofs[X0], 0, sc.crop_w - 1);
            ph = pd_clampi(sc.ty.ofs[Ylast] + 3, 0, sc.crop_h - 1) - r_lo + 1;
            pw = pd_clampi(sc.tx.ofs[Xlast] + 3, 0, sc.crop_w - 1) - c_lo + 1;
        }
        const int j_lo = pd_up8_ofs(r_lo);                         // first source row (unclamped)
        const int nj = pd_up8_ofs(r_lo + ph - 1) + 3 - j_lo + 1;   // source rows the patch draws on
        // source columns the patch draws on, after clamping to the map
        const int i_lo = pd_clampi(pd_up8_ofs(c_lo), 0, sc.ws - 1);
        const int ni = pd_clampi(pd_up8_ofs(c_lo + pw - 1) + 3, 0, sc.ws - 1) - i_lo + 1;
        const bool staged = !DIRECT || (ph * pw <= FT_PATCH && nj * pw <= FT_HSUM && nj * ni <= FT_SRC);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (staged) {
            // 0: the source pixels, element e = source row jj (clamped to the map), column ii
            {
                const int mg = div_magic(ni);
                for (int e = tid; e < nj * ni; e += 256) {
                    const int jj = div_by(e, mg), ii = e - jj * ni;
                    const unsigned off = (unsigned)((pd_clampi(j_lo + jj, 0, sc.hs - 1) * sc.ws + i_lo + ii) * a.C);
                    srcp[e] = load_channels(img + off, nch);
                }
            }
            __syncthreads();
            // A: horizontal sums, element e = source row jj, patch column c
            const int mw = div_magic(pw);
            for (int e = tid; e < nj * pw; e += 256) {
                const int jj = div_by(e, mw), c = e - jj * pw;
                const int col = c_lo + c;
                const int x0 = pd_up8_ofs(col);
                const float4* row = srcp + jj * ni - i_lo;
                const int hi = sc.ws - 1;
                hsum[e] = tap4v(row[pd_clampi(x0, 0, hi)], row[pd_clampi(x0 + 1, 0, hi)], row[pd_clampi(x0 + 2, 0, hi)],
                                row[pd_clampi(x0 + 3, 0, hi)], s_up[pd_up8_phase(col)]);
            }
            __syncthreads();
            // B: the patch.  A thread walks down a run of rows of one column and keeps the four
            // horizontal sums in use in registers (the first tap row advances every eighth row).
            const int G = max(1, 256 / pw), rpg = (ph + G - 1) / G;
            for (int it = tid; it < G * pw; it += 256) {
                const int g = div_by(it, mw), c = it - g * pw;
                const int r1 = min(ph, (g + 1) * rpg);
                int cur = -0x40000000;
                float4 h0 = make_float4(0.f, 0.f, 0.f, 0.f), h1 = h0, h2 = h0, h3 = h0;
                for (int r = g * rpg; r < r1; ++r) {
                    const int rabs = r_lo + r;
                    const int y0 = pd_up8_ofs(rabs) - j_lo;
                    if (y0 != cur) {
                        const float4* col = hsum + y0 * pw + c;
                        if (y0 == cur + 1) {
                            h0 = h1;
                            h1 = h2;
                            h2 = h3;
                        } else {
                            h0 = col[0];
                            h1 = col[pw];
                            h2 = col[2 * pw];
                        }
                        h3 = col[3 * pw];
                        cur = y0;
                    }
                    patch[r * pw + c] = tap4v(h0, h1, h2, h3, s_up[pd_up8_phase(rabs)]);
                }
            }
            __syncthreads();
            // C: the second resize from the patch
            if (sc.identity) {
                v = patch[(Y - r_lo) * pw + (X - c_lo)];
            } else {
                const int x0 = sc.tx.ofs[X], y0 = sc.ty.ofs[Y];
                const float4 ac = *(const float4*)(sc.tx.coef + 4 * (size_t)X);
                const float4 bc = *(const float4*)(sc.ty.coef + 4 * (size_t)Y);
                const int xi0 = pd_clampi(x0, 0, sc.crop_w - 1) - c_lo, xi1 = pd_clampi(x0 + 1, 0, sc.crop_w - 1) - c_lo,
                          xi2 = pd_clampi(x0 + 2, 0, sc.crop_w - 1) - c_lo, xi3 = pd_clampi(x0 + 3, 0, sc.crop_w - 1) - c_lo;
                float4 hvv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4* row = patch + (pd_clampi(y0 + k, 0, sc.crop_h - 1) - r_lo) * pw;
                    hvv[k] = tap4v(row[xi0], row[xi1], row[xi2], row[xi3], ac);
                }
                v = tap4v(hvv[0], hvv[1], hvv[2], hvv[3], bc);
            }
        } else if (DIRECT) {
            // a patch too large to stage (a strong downscale): every tap of the second resize is
            // formed from the source on the spot
            const int x0 = sc.tx.ofs[X], y0 = sc.ty.ofs[Y];
            const float4 ac = *(const float4*)(sc.tx.coef + 4 * (size_t)X);
            const float4 bc = *(const float4*)(sc.ty.coef + 4 * (size_t)Y);
            if (sc.identity) {
                v = up8_at(img, sc.hs, sc.ws, a.C, Y, X, nch, s_up);
            } else {
                // sums built up left to right, one tap at a time (loops kept rolled: this path is rare
                // and must not cost the kernel its registers)
#pragma unroll 1
                for (int k = 0; k < 4; ++k) {
                    const int r = pd_clampi(y0 + k, 0, sc.crop_h - 1);
                    float4 hs4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
                    for (int t = 0; t < 4; ++t) {
                        const float4 u = up8_at(img, sc.hs, sc.ws, a.C, r, pd_clampi(x0 + t, 0, sc.crop_w - 1), nch, s_up);
                        const float cw = t == 0 ? ac.x : (t == 1 ? ac.y : (t == 2 ? ac.z : ac.w));
                        if (t == 0)
                            hs4 = make_float4(u.x * cw, u.y * cw, u.z * cw, u.w * cw);
                        else
                            hs4 = make_float4(hs4.x + u.x * cw, hs4.y + u.y * cw, hs4.z + u.z * cw, hs4.w + u.w * cw);
                    }
                    const float rw = k == 0 ? bc.x : (k == 1 ? bc.y : (k == 2 ? bc.z : bc.w));
                    if (k == 0)
                        v = make_float4(hs4.x * rw, hs4.y * rw, hs4.z * rw, hs4.w * rw);
                    else
                        v = make_float4(v.x + hs4.x * rw, v.y + hs4.y * rw, v.z + hs4.z * rw, v.w + hs4.w * rw);
                }
            }
        }
        acc[0] = acc[0] + (double)avg_share<POW2>(v.x, a);
        acc[1] = acc[1] + (double)avg_share<POW2>(v.y, a);
        acc[2] = acc[2] + (double)avg_share<POW2>(v.z, a);
        acc[3] = acc[3] + (double)avg_share<POW2>(v.w, a);
        // no barrier here: the next scale's pass A writes hsum only (every thread is past
        // this scale's pass B), and its pass B writes the patch behind a barrier of its own
    }
    if (a.absmax) {
        float mx = 0.f;
#pragma unroll
        for (int c = 0; c < FT_CH; ++c)
            if (inside && c < nch) mx = fmaxf(mx, (float)fabs(acc[c]));
        pd_wave_absmax(mx, a.absmax + b);
    }
    if (!inside) return;
    double* out = a.dst + ((size_t)b * a.C + ch0) * n + (size_t)Y * a.W + X;
#pragma unroll
    for (int c = 0; c < FT_CH; ++c)
        if (c < nch) out[c * n] = acc[c];
}

// ---- the same pass with the index work hoisted out of a loop over channel groups ------------
// Which source pixels, horizontal sums and patch positions a thread touches depends on the
// tile and the scale only, not on the channels: a workgroup owns one output tile of one image
// and walks the channel groups, with every LDS byte offset and global element offset of its
// threads worked out once and kept in registers (16-bit fields, two per register).  The
// arithmetic of a group is that of k_resize_fused, stage by stage.
#define RT_MAXS 4        // scales per call this kernel is built for (more: k_resize_fused)
#define RT_AQ 3          // horizontal sums per thread and scale: FT_HSUM <= 3 * 256

struct TileScale {       // per scale, uniform over the workgroup
    int r_lo, c_lo, ph, pw, j_lo, nj, i_lo, ni;
};
struct ThreadScale {     // per scale, per thread
    unsigned o0;         // stage 0: element offset of this thread's source pixel (group 0), ~0u = none
    // stage A: byte offsets (< 4096) of the four source pixels in srcp, two per word; bits 12-15 of
    // a01: phase of the column, 8 = no element
    unsigned a01[RT_AQ], a23[RT_AQ];
    unsigned b;          // stage B: column | first row << 8 | end row << 20
    // stage C: byte offset of the first tap column within a patch row (bits 0-11) and of the first
    // tap row in the patch (bits 0-14); the three bits above say whether the next tap moves on by
    // one column / one row (clamped taps repeat)
    unsigned cx, cr;
};

__device__ __forceinline__ float4 lds4(const float4* base, unsigned byte_off) {
    return *(const float4*)((const char*)base + byte_off);
}

template <bool POW2, int NS>
__global__ void __launch_bounds__(256, 4)
k_resize_tiles(AvgArgs a, int groups_per_block, int images_per_block, int n_images) {
    __shared__ float4 patch[FT_PATCH];
    __shared__ float4 hsum[FT_HSUM];
    __shared__ float4 srcp[FT_SRC];
    __shared__ float4 s_up[8];
    const int tid = threadIdx.x;
    if (tid < 8) s_up[tid] = make_float4(a.up8.c[tid][0], a.up8.c[tid][1], a.up8.c[tid][2], a.up8.c[tid][3]);
    const size_t n = (size_t)a.H * a.W;
    const int tiles_x = (a.W + FT_TW - 1) / FT_TW, tiles = tiles_x * ((a.H + FT_TH - 1) / FT_TH);
    const int per_xcd = (tiles + 7) / 8;
    const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;   // tiles that share source rows share an XCD
    if (tile >= tiles) return;
    const int X0 = (tile % tiles_x) * FT_TW, Y0 = (tile / tiles_x) * FT_TH;
    // Output pixel of a thread: a wavefront owns four rows of the tile, and the sixteen lanes that
    // a 16-byte LDS read serves in one cycle (lanes 0-3, 12-15, 20-27 | 4-11, 16-19, 28-31 of each
    // half) own one row, so that the taps of stage C fall on consecutive patch pixels of one row.
    const int l5 = tid & 31;
    const bool first = l5 < 4 || (l5 >= 12 && l5 < 16) || (l5 >= 20 && l5 < 28);
    const int px = first ? (l5 < 4 ? l5 : l5 < 16 ? l5 - 8 : l5 - 12) : (l5 < 12 ? l5 - 4 : l5 < 20 ? l5 - 8 : l5 - 16);
    const int py = (tid >> 6) * 4 + ((tid >> 5) & 1) * 2 + (first ? 0 : 1);
    const bool inside = X0 + px < a.W && Y0 + py < a.H;
    const int X = min(X0 + px, a.W - 1), Y = min(Y0 + py, a.H - 1);
    const int Ylast = min(Y0 + FT_TH - 1, a.H - 1), Xlast = min(X0 + FT_TW - 1, a.W - 1);
    // a workgroup takes its tile in several images, one after the other: what the tile's threads
    // touch is worked out once for all of them
    const int b_begin = blockIdx.z * images_per_block, b_end = min(n_images, b_begin + images_per_block);
    const int n_groups = (a.C + FT_CH - 1) / FT_CH;
    const int g_begin = blockIdx.y * groups_per_block, g_end = min(n_groups, g_begin + groups_per_block);

    TileScale ts[NS];
    ThreadScale th[NS];
#pragma unroll
    for (int m = 0; m < NS; ++m) {
        const ScaleSrc& sc = a.s[m];
        TileScale& t = ts[m];
        ThreadScale& q = th[m];
        if (sc.identity) {
            t.r_lo = Y0;
            t.c_lo = X0;
            t.ph = Ylast - Y0 + 1;
            t.pw = Xlast - X0 + 1;
        } else {
            t.r_lo = pd_clampi(sc.ty.ofs[Y0], 0, sc.crop_h - 1);
            t.c_lo = pd_clampi(sc.tx.ofs[X0], 0, sc.crop_w - 1);
            t.ph = pd_clampi(sc.ty.ofs[Ylast] + 3, 0, sc.crop_h - 1) - t.r_lo + 1;
            t.pw = pd_clampi(sc.tx.ofs[Xlast] + 3, 0, sc.crop_w - 1) - t.c_lo + 1;
        }
        // uniform over the workgroup: keep them in scalar registers
        t.r_lo = __builtin_amdgcn_readfirstlane(t.r_lo);
        t.c_lo = __builtin_amdgcn_readfirstlane(t.c_lo);
        t.ph = __builtin_amdgcn_readfirstlane(t.ph);
        t.pw = __builtin_amdgcn_readfirstlane(t.pw);
        t.j_lo = pd_up8_ofs(t.r_lo);
        t.nj = pd_up8_ofs(t.r_lo + t.ph - 1) + 3 - t.j_lo + 1;
        t.i_lo = pd_clampi(pd_up8_ofs(t.c_lo), 0, sc.ws - 1);
        t.ni = pd_clampi(pd_up8_ofs(t.c_lo + t.pw - 1) + 3, 0, sc.ws - 1) - t.i_lo + 1;
        // stage 0: one source pixel per thread (nj * ni <= FT_SRC <= 256)
        q.o0 = ~0u;
        if (tid < t.nj * t.ni) {
            const int jj = tid / t.ni, ii = tid - jj * t.ni;
            q.o0 = (unsigned)((pd_clampi(t.j_lo + jj, 0, sc.hs - 1) * sc.ws + t.i_lo + ii) * a.C);
        }
        // stage A: horizontal sum e = source row jj, patch column c
#pragma unroll
        for (int k = 0; k < RT_AQ; ++k) {
            const int e = tid + 256 * k;
            unsigned ph = 8, i0 = 0, i1 = 0, i2 = 0, i3 = 0;
            if (e < t.nj * t.pw) {
                const int jj = e / t.pw, c = e - jj * t.pw;
                const int col = t.c_lo + c, x0 = pd_up8_ofs(col), hi = sc.ws - 1;
                const int rowb = jj * t.ni - t.i_lo;
                ph = (unsigned)pd_up8_phase(col);
                i0 = (unsigned)(rowb + pd_clampi(x0, 0, hi)) * 16u;
                i1 = (unsigned)(rowb + pd_clampi(x0 + 1, 0, hi)) * 16u;
                i2 = (unsigned)(rowb + pd_clampi(x0 + 2, 0, hi)) * 16u;
                i3 = (unsigned)(rowb + pd_clampi(x0 + 3, 0, hi)) * 16u;
            }
            q.a01[k] = i0 | (ph << 12) | (i1 << 16);
            q.a23[k] = i2 | (i3 << 16);
        }
        // stage B: a run of rows of one patch column
        {
            const int G = max(1, 256 / t.pw), rpg = (t.ph + G - 1) / G;
            q.b = 0;
            if (tid < G * t.pw) {
                const int g = tid / t.pw, c = tid - g * t.pw;
                const int r0 = min(t.ph, g * rpg), r1 = min(t.ph, (g + 1) * rpg);
                q.b = (unsigned)c | ((unsigned)r0 << 8) | ((unsigned)r1 << 20);
            }
        }
        // stage C: the taps of this thread's output pixel
        if (sc.identity) {
            q.cx = 0;
            q.cr = (unsigned)((Y - t.r_lo) * t.pw + (X - t.c_lo)) * 16u;
        } else {
            const int x0 = sc.tx.ofs[X], y0 = sc.ty.ofs[Y];
            int xo[4], ro[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                xo[k] = pd_clampi(x0 + k, 0, sc.crop_w - 1) - t.c_lo;
                ro[k] = pd_clampi(y0 + k, 0, sc.crop_h - 1) - t.r_lo;
            }
            q.cx = (unsigned)xo[0] * 16u | (unsigned)(xo[1] - xo[0]) << 12 | (unsigned)(xo[2] - xo[1]) << 13 |
                   (unsigned)(xo[3] - xo[2]) << 14;
            q.cr = (unsigned)(ro[0] * t.pw) * 16u | (unsigned)(ro[1] - ro[0]) << 15 | (unsigned)(ro[2] - ro[1]) << 16 |
                   (unsigned)(ro[3] - ro[2]) << 17;
        }
    }
    __syncthreads();

    float amax = 0.f;
    float4 pre = make_float4(0.f, 0.f, 0.f, 0.f);  // this thread's source pixel of the coming round
    if (g_begin < g_end && b_begin < b_end && th[0].o0 != ~0u)
        pre = load_channels(a.s[0].lo + (size_t)b_begin * a.s[0].hs * a.s[0].ws * a.C + g_begin * FT_CH + th[0].o0,
                            min(FT_CH, a.C - g_begin * FT_CH));
    const int n_rounds = g_begin < g_end ? (b_end - b_begin) * (g_end - g_begin) : 0;
    int b = b_begin, g = g_begin;
    for (int it = 0; it < n_rounds; ++it) {
        const int ch0 = g * FT_CH;
        // the (image, channel group) after this one
        const bool last_group = g + 1 == g_end;
        const int b_next = last_group ? b + 1 : b, ch_next = last_group ? g_begin * FT_CH : ch0 + FT_CH;
        const int nch = min(FT_CH, a.C - ch0);
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
        // The packed fields are re-read as opaque values in every round: otherwise the compiler
        // unpacks them (and loads the tap weights) ahead of the loop and the unpacked copies of
        // all scales no longer fit the registers of four workgroups per CU.
        int Xo = X, Yo = Y;
        asm volatile("" : "+v"(Xo), "+v"(Yo));
#pragma unroll
        for (int m = 0; m < NS; ++m) {
            ThreadScale& w = th[m];
            asm volatile("" : "+v"(w.o0), "+v"(w.b), "+v"(w.cx), "+v"(w.cr));
#pragma unroll
            for (int k = 0; k < RT_AQ; ++k) asm volatile("" : "+v"(w.a01[k]), "+v"(w.a23[k]));
        }
#pragma unroll
        for (int m = 0; m < NS; ++m) {
            const ScaleSrc& sc = a.s[m];
            const TileScale& t = ts[m];
            const ThreadScale& q = th[m];
            // 0: the source pixels the patch draws on (requested a round ago)
            if (q.o0 != ~0u) srcp[tid] = pre;
            __syncthreads();
            // the next round's source pixel is requested now and arrives while this round computes
            if (m + 1 < NS) {
                if (th[m + 1].o0 != ~0u)
                    pre = load_channels(a.s[m + 1].lo + (size_t)b * a.s[m + 1].hs * a.s[m + 1].ws * a.C + ch0 + th[m + 1].o0, nch);
            } else if (it + 1 < n_rounds) {
                if (th[0].o0 != ~0u)
                    pre = load_channels(a.s[0].lo + (size_t)b_next * a.s[0].hs * a.s[0].ws * a.C + ch_next + th[0].o0,
                                        min(FT_CH, a.C - ch_next));
            }
            // A: horizontal sums
#pragma unroll
            for (int k = 0; k < RT_AQ; ++k) {
                const unsigned ph = (q.a01[k] >> 12) & 15u;
                if (ph < 8u)
                    hsum[tid + 256 * k] = tap4v(lds4(srcp, q.a01[k] & 0xfffu), lds4(srcp, q.a01[k] >> 16),
                                                lds4(srcp, q.a23[k] & 0xffffu), lds4(srcp, q.a23[k] >> 16), s_up[ph]);
            }
            __syncthreads();
            // B: the patch of the x8 map, a run of rows per thread
            {
                const int c = (int)(q.b & 255u), r0 = (int)((q.b >> 8) & 4095u), r1 = (int)(q.b >> 20);
                int cur = -0x40000000;
                float4 h0 = make_float4(0.f, 0.f, 0.f, 0.f), h1 = h0, h2 = h0, h3 = h0;
                for (int r = r0; r < r1; ++r) {
                    const int rabs = t.r_lo + r;
                    const int y0 = pd_up8_ofs(rabs) - t.j_lo;
                    if (y0 != cur) {
                        const float4* col = hsum + y0 * t.pw + c;
                        if (y0 == cur + 1) {
                            h0 = h1;
                            h1 = h2;
                            h2 = h3;
                        } else {
                            h0 = col[0];
                            h1 = col[t.pw];
                            h2 = col[2 * t.pw];
                        }
                        h3 = col[3 * t.pw];
                        cur = y0;
                    }
                    patch[r * t.pw + c] = tap4v(h0, h1, h2, h3, s_up[pd_up8_phase(rabs)]);
                }
            }
            __syncthreads();
            // C: the second resize from the patch
            float4 v;
            if (sc.identity) {
                v = lds4(patch, q.cr);
            } else {
                const float4 ac = *(const float4*)(sc.tx.coef + 4 * (size_t)Xo);
                const float4 bc = *(const float4*)(sc.ty.coef + 4 * (size_t)Yo);
                const unsigned x0 = q.cx & 0xfffu, x1 = x0 + ((q.cx >> 8) & 16u), x2 = x1 + ((q.cx >> 9) & 16u),
                               x3 = x2 + ((q.cx >> 10) & 16u);
                const unsigned rowb = (unsigned)t.pw * 16u;
                const unsigned r0 = q.cr & 0x7fffu, r1 = r0 + ((q.cr >> 15) & 1u) * rowb, r2 = r1 + ((q.cr >> 16) & 1u) * rowb,
                               r3 = r2 + ((q.cr >> 17) & 1u) * rowb;
                // two tap rows at a time: all sixteen taps in flight at once would cost 64 registers
                const float4 v0 = tap4v(lds4(patch, r0 + x0), lds4(patch, r0 + x1), lds4(patch, r0 + x2), lds4(patch, r0 + x3), ac);
                const float4 v1 = tap4v(lds4(patch, r1 + x0), lds4(patch, r1 + x1), lds4(patch, r1 + x2), lds4(patch, r1 + x3), ac);
                __builtin_amdgcn_sched_barrier(0);
                const float4 v2 = tap4v(lds4(patch, r2 + x0), lds4(patch, r2 + x1), lds4(patch, r2 + x2), lds4(patch, r2 + x3), ac);
                const float4 v3 = tap4v(lds4(patch, r3 + x0), lds4(patch, r3 + x1), lds4(patch, r3 + x2), lds4(patch, r3 + x3), ac);
                v = tap4v(v0, v1, v2, v3, bc);
            }
            acc0 = acc0 + (double)avg_share<POW2>(v.x, a);
            acc1 = acc1 + (double)avg_share<POW2>(v.y, a);
            acc2 = acc2 + (double)avg_share<POW2>(v.z, a);
            acc3 = acc3 + (double)avg_share<POW2>(v.w, a);
            // no barrier: the next stage 0 writes srcp (last read before two barriers), the next
            // stage A writes hsum behind a barrier, the next stage B the patch behind two
        }
        if (inside) {
            double* out = a.dst + ((size_t)b * a.C + ch0) * n + (size_t)Y * a.W + X;
            out[0] = acc0;
            amax = fmaxf(amax, (float)fabs(acc0));
            if (nch > 1) {
                out[n] = acc1;
                amax = fmaxf(amax, (float)fabs(acc1));
            }
            if (nch > 2) {
                out[2 * n] = acc2;
                amax = fmaxf(amax, (float)fabs(acc2));
            }
            if (nch > 3) {
                out[3 * n] = acc3;
                amax = fmaxf(amax, (float)fabs(acc3));
            }
        }
        if (last_group) {
            if (a.absmax) pd_wave_absmax(amax, a.absmax + b);
            amax = 0.f;
            g = g_begin;
            ++b;
        } else {
            ++g;
        }
    }
}

// Whether every tile's patch of the second resize (crop -> H x W) fits the staging areas of
// k_resize_fused; the same extents the kernel derives from the device tables.
static bool patches_fit(int crop_h, int crop_w, int H, int W) {
    std::vector<int32_t> oy(H), ox(W);
    std::vector<float> cy(4 * (size_t)H), cx(4 * (size_t)W);
    pd_host_cubic_table(crop_h, H, -1.0, oy.data(), cy.data());
    pd_host_cubic_table(crop_w, W, -1.0, ox.data(), cx.data());
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    int max_ph = 0, max_pw = 0, max_nj = 0, max_ni = 0;
    for (int Y0 = 0; Y0 < H; Y0 += FT_TH) {
        const int Yl = std::min(Y0 + FT_TH - 1, H - 1);
        const int r_lo = clampi(oy[Y0], 0, crop_h - 1);
        const int ph = clampi(oy[Yl] + 3, 0, crop_h - 1) - r_lo + 1;
        const int nj = pd_up8_ofs(r_lo + ph - 1) + 3 - pd_up8_ofs(r_lo) + 1;
        max_ph = std::max(max_ph, ph);
        max_nj = std::max(max_nj, nj);
    }
    for (int X0 = 0; X0 < W; X0 += FT_TW) {
        const int Xl = std::min(X0 + FT_TW - 1, W - 1);
        const int c_lo = clampi(ox[X0], 0, crop_w - 1);
        const int pw = clampi(ox[Xl] + 3, 0, crop_w - 1) - c_lo + 1;
        max_pw = std::max(max_pw, pw);
        max_ni = std::max(max_ni, pd_up8_ofs(c_lo + pw - 1) + 3 - pd_up8_ofs(c_lo) + 1);
    }
    // (the tile kernel also packs a patch column in 8 bits and a patch row in 12)
    return max_ph * max_pw <= FT_PATCH && max_nj * max_pw <= FT_HSUM && max_nj * max_ni <= FT_SRC && max_pw <= 255 &&
           max_ph <= 4095;
}

template <bool POW2>
static void launch_tiles(int ns, dim3 grid, hipStream_t stream, const AvgArgs& a, int gpb, int ipb, int B) {
    switch (ns) {
        case 1: hipLaunchKernelGGL((k_resize_tiles<POW2, 1>), grid, dim3(256), 0, stream, a, gpb, ipb, B); break;
        case 2: hipLaunchKernelGGL((k_resize_tiles<POW2, 2>), grid, dim3(256), 0, stream, a, gpb, ipb, B); break;
        case 3: hipLaunchKernelGGL((k_resize_tiles<POW2, 3>), grid, dim3(256), 0, stream, a, gpb, ipb, B); break;
        default: hipLaunchKernelGGL((k_resize_tiles<POW2, 4>), grid, dim3(256), 0, stream, a, gpb, ipb, B); break;
    }
}

// All scales of one map family (C = 19 or 38): srcs[m] is (B, hs[m], ws[m], C) on the device.
int pd_launch_upsample_avg(pd_ctx* ctx, int n_scales, const float* const* srcs, int C, int B, const int32_t* hs,
                           const int32_t* ws, const int32_t* crop_h, const int32_t* crop_w, double* dst, int H, int W,
                           float* absmax) {
    AvgArgs a;
    memset(&a, 0, sizeof(a));
    a.absmax = absmax;
    pd_up8_fill(&a.up8);
    bool direct = false;
    for (int m = 0; m < n_scales; ++m) {
        ScaleSrc& sc = a.s[m];
        sc.lo = srcs[m];
        sc.hs = hs[m];
        sc.ws = ws[m];
        sc.crop_h = crop_h[m];
        sc.crop_w = crop_w[m];
        sc.identity = crop_h[m] == H && crop_w[m] == W;
        if (!sc.identity) {
            int rc = pd_upload_tables(ctx, crop_h[m], crop_w[m], H, W, -1.0, -1.0, &ctx->tabs[2 + m]);
            if (rc != PD_OK) return rc;
            sc.ty = ctx->tabs[2 + m].ty;
            sc.tx = ctx->tabs[2 + m].tx;
            if (!patches_fit(crop_h[m], crop_w[m], H, W)) direct = true;
        }
    }
    a.n_scales = n_scales;
    a.C = C;
    a.H = H;
    a.W = W;
    a.n_div = (float)n_scales;
    a.dst = dst;
    const unsigned tiles = (unsigned)(((W + FT_TW - 1) / FT_TW) * ((H + FT_TH - 1) / FT_TH));
    pd_prof_begin(ctx, PD_STAGE_RESIZE);
    const dim3 grid((tiles + 7) / 8 * 8, (C + FT_CH - 1) / FT_CH, B);
    const bool pow2 = (n_scales & (n_scales - 1)) == 0;
    if (pow2) a.n_rcp = 1.0f / (float)n_scales;
    // the tile kernel packs LDS offsets in 16 bits and patch columns in 8: every staged patch fits that
    static const bool old_kernel = getenv("PD_RESIZE_PER_GROUP") != nullptr;  // comparison runs
    if (!direct && n_scales <= RT_MAXS && !old_kernel) {
        const int n_groups = (C + FT_CH - 1) / FT_CH;
        // enough workgroups to fill the chip several times over, as few channel chunks as that allows
        int chunks = (int)((4096 + (size_t)tiles * B - 1) / ((size_t)tiles * B));
        chunks = chunks < 1 ? 1 : (chunks > n_groups ? n_groups : chunks);
        const int gpb = (n_groups + chunks - 1) / chunks;
        // and several images per workgroup while that leaves sixteen workgroups for every slot of
        // the chip (the set-up of a tile is a fifth of what one image's channel groups cost)
        int ipb = (int)((size_t)tiles * B / 16384);
        ipb = ipb < 1 ? 1 : (ipb > 8 ? 8 : ipb);
        static const int ipb_env = getenv("PD_RESIZE_IMAGES") ? atoi(getenv("PD_RESIZE_IMAGES")) : 0;  // comparison runs
        if (ipb_env > 0) ipb = ipb_env;
        const dim3 tgrid((tiles + 7) / 8 * 8, (n_groups + gpb - 1) / gpb, (B + ipb - 1) / ipb);
        if (pow2)
            launch_tiles<true>(n_scales, tgrid, ctx->stream, a, gpb, ipb, B);
        else
            launch_tiles<false>(n_scales, tgrid, ctx->stream, a, gpb, ipb, B);
    } else if (pow2 && !direct)
        hipLaunchKernelGGL((k_resize_fused<true, false>), grid, dim3(256), 0, ctx->stream, a);
    else if (pow2)
        hipLaunchKernelGGL((k_resize_fused<true, true>), grid, dim3(256), 0, ctx->stream, a);
    else if (!direct)
        hipLaunchKernelGGL((k_resize_fused<false, false>), grid, dim3(256), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL((k_resize_fused<false, true>), grid, dim3(256), 0, ctx->stream, a);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    return PD_OK;
}

int pd_launch_hwc_to_planar(pd_ctx* ctx, const void* src, int dtype, int Csrc, int Cdst, int B, int H,
                            int W, double* dst, float* absmax) {
    const size_t n = (size_t)H * W;
    const int nch = Csrc < Cdst ? Csrc : Cdst;
    pd_prof_begin(ctx, PD_STAGE_RESIZE);
    hipLaunchKernelGGL(k_hwc_to_planar, dim3((unsigned)((n + 255) / 256), nch, B), dim3(256), 0, ctx->stream, src,
                       dtype, Csrc, Cdst, H, W, dst, absmax);
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
