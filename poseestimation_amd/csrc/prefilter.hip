// Fast peak stage for network-resolution sources: a float32 candidate filter followed
// by exact float64 verification of the few candidates.  Results are identical to the
// exact kernel over every pixel (get_peaks, utils/postprocess.py:26-75).
//
// Filter.  The smoothed map is a linear function of the (h, w) network output:
// Gaussian (25 taps, reflect) after bicubic x8 (4 taps, clamped) composes, per axis,
// into a banded operator with 7 taps per output.  The filter evaluates
//     T[Y][t]  = sum_k Ky[Y][k] * src[s0(Y)+k][t]          (rows)
//     S~[Y][X] = sum_p Kx[X][p] * T[Y][t0(X)+p]            (columns)
// in float32 with fused multiply-adds.  Against the exact float64 value S of the
// reference (which is computed from the float32-rounded bicubic map),
//     |S~ - S| <= 53 * 2^-24 * 1.89 * M  <  eps := 2^-17 * M,   M = max |src| of the plane
// (coefficient rounding, 8-term chains on both axes, and the rounding inside the
// reference's own float32 bicubic; sum_k |K| <= 1.375 per axis).  A pixel can only be
// a peak (S > thr, S >= each in-image neighbour, S >= 0 for out-of-image ones) if
//     S~ > thr - eps  and  S~ >= S~(neighbour) - 2 eps,
// so every true peak is among the candidates.
//
// Verification.  Per candidate the exact S at the pixel and its four neighbours is
// recomputed with the reference's arithmetic (27 x 27 bicubic values, 77 y-pass sums,
// 5 x-pass sums) and the exact test decides.  Planes with more candidates than the list
// holds (plateaus) are redone by the exact streaming kernel.
#include <math.h>

#include <vector>

#include "colloader.h"

#pragma clang fp contract(off)

#define KF_RB 16       // rows per thread in the column pass
#define KF_NT 256

static const double h_gw[PD_GTAPS] = PD_GAUSS_WEIGHTS;

static int host_reflect(int i, int n) {
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - 1 - i : i;
}

// Dense composite K[out][s] = sum_d g[d] * sum_k coef[refl(out+d)][k] * [clamp(ofs+k) == s].
static void composite(int L, int l, const int32_t* ofs, const float* coef, std::vector<double>& K) {
    K.assign((size_t)L * l, 0.0);
    for (int o = 0; o < L; ++o)
        for (int d = -PD_GR; d <= PD_GR; ++d) {
            const int r = host_reflect(o + d, L);
            const double g = h_gw[PD_GR - (d < 0 ? -d : d)];
            for (int k = 0; k < 4; ++k) {
                int s = ofs[r] + k;
                s = s < 0 ? 0 : (s > l - 1 ? l - 1 : s);
                K[(size_t)o * l + s] += g * (double)coef[4 * r + k];
            }
        }
}

static int build_filter_tables(pd_ctx* ctx) {
    const int h = ctx->h, w = ctx->w, H = ctx->H, W = ctx->W;
    if (ctx->kf_h == h && ctx->kf_w == w && ctx->kf_H == H && ctx->kf_W == W) return ctx->kf_ok ? PD_OK : PD_E_STATE;
    ctx->kf_h = h;
    ctx->kf_w = w;
    ctx->kf_H = H;
    ctx->kf_W = W;
    ctx->kf_ok = 0;
    const int wg = (W + 7) / 8;
    if (h < 8 || w < 8 || H >= 65536 || W >= 65536 || wg > KF_NT) return PD_E_STATE;
    std::vector<int32_t> yofs(H), xofs(W);
    std::vector<float> yc(4 * (size_t)H), xc(4 * (size_t)W);
    pd_host_cubic_table(h, H, 0.125, yofs.data(), yc.data());
    pd_host_cubic_table(w, W, 0.125, xofs.data(), xc.data());
    std::vector<double> Ky, Kx;
    composite(H, h, yofs.data(), yc.data(), Ky);
    composite(W, w, xofs.data(), xc.data(), Kx);
    std::vector<int32_t> ystart(H);
    std::vector<float> ycoef(8 * (size_t)H);
    for (int Y = 0; Y < H; ++Y) {
        int lo = h, hi = -1;
        for (int s = 0; s < h; ++s)
            if (Ky[(size_t)Y * h + s] != 0.0) {
                if (s < lo) lo = s;
                hi = s;
            }
        int st = lo < h - 8 ? lo : h - 8;
        if (st < 0) st = 0;
        if (hi - st > 7) return PD_E_STATE;
        if (Y > 0 && st < ystart[Y - 1]) return PD_E_STATE;  // the band loader assumes monotone starts
        ystart[Y] = st;
        for (int k = 0; k < 8; ++k) ycoef[8 * (size_t)Y + k] = (float)Ky[(size_t)Y * h + st + k];
    }
    std::vector<float> xcoef((size_t)wg * 70, 0.f);
    for (int g = 0; g < wg; ++g)
        for (int jj = 0; jj < 10; ++jj) {
            const int X = 8 * g + jj - 1;
            if (X < 0 || X >= W) continue;
            const int p0 = jj == 0 ? 0 : (jj == 9 ? 2 : 1);
            const int t_lo = g - 4 + p0;
            for (int t = 0; t < w; ++t) {
                const double v = Kx[(size_t)X * w + t];
                if (v == 0.0) continue;
                if (t < t_lo || t > t_lo + 6) return PD_E_STATE;
                xcoef[((size_t)g * 10 + jj) * 7 + (t - t_lo)] = (float)v;
            }
        }
    PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->kf_ycap < (size_t)H) {
        if (ctx->kf_ystart) PD_HIP(ctx, hipFree(ctx->kf_ystart));
        if (ctx->kf_ycoef) PD_HIP(ctx, hipFree(ctx->kf_ycoef));
        ctx->kf_ystart = nullptr;
        ctx->kf_ycoef = nullptr;
        ctx->kf_ycap = 0;
        PD_HIP(ctx, hipMalloc(&ctx->kf_ystart, sizeof(int32_t) * H));
        PD_HIP(ctx, hipMalloc(&ctx->kf_ycoef, sizeof(float) * 8 * H));
        ctx->kf_ycap = H;
    }
    if (ctx->kf_xcap < (size_t)wg) {
        if (ctx->kf_xcoef) PD_HIP(ctx, hipFree(ctx->kf_xcoef));
        ctx->kf_xcoef = nullptr;
        ctx->kf_xcap = 0;
        PD_HIP(ctx, hipMalloc(&ctx->kf_xcoef, sizeof(float) * 70 * wg));
        ctx->kf_xcap = wg;
    }
    PD_HIP(ctx, hipMemcpy(ctx->kf_ystart, ystart.data(), sizeof(int32_t) * H, hipMemcpyHostToDevice));
    PD_HIP(ctx, hipMemcpy(ctx->kf_ycoef, ycoef.data(), sizeof(float) * 8 * H, hipMemcpyHostToDevice));
    PD_HIP(ctx, hipMemcpy(ctx->kf_xcoef, xcoef.data(), sizeof(float) * 70 * wg, hipMemcpyHostToDevice));
    ctx->kf_ok = 1;
    return PD_OK;
}

struct FilterArgs {
    const float* lo;  // (B, h, w, 19)
    int h, w, H, W, wg;
    int RL, BH, src_rows;  // row lanes, band height, source rows staged per band
    const int32_t* ystart;
    const float* ycoef;
    const float* xcoef;
    double thr;
    int32_t* cand;
    int32_t* cand_count;
    int CK;
};

// One workgroup per (row band, part, image).
__global__ void __launch_bounds__(KF_NT)
k_prefilter(FilterArgs a) {
    extern __shared__ __align__(16) float lds[];
    __shared__ float s_red[KF_NT / 64];
    const int tid = threadIdx.x;
    const int band = blockIdx.x, part = blockIdx.y, b = blockIdx.z;
    const int h = a.h, w = a.w, H = a.H, W = a.W;
    const int Yb = band * a.BH;               // first row this band decides
    const int nrowsT = a.BH + 2;              // rows Yb-1 .. Yb+BH
    const int tw = w + 8;                     // T row: 4 zero pads each side
    float* srcL = lds;                        // [src_rows][w]
    float* T = lds + (size_t)a.src_rows * w;  // [BH+2][w+8]
    const float* img = a.lo + (size_t)b * h * w * PD_CH_HEAT + part;

    // M = max |src| of the plane
    float m = 0.f;
    for (int i = tid; i < h * w; i += KF_NT) m = fmaxf(m, fabsf(img[(size_t)i * PD_CH_HEAT]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((tid & 63) == 0) s_red[tid >> 6] = m;

    // source rows of this band
    const int Ylo = Yb - 1 < 0 ? 0 : Yb - 1;
    const int Yhi = Yb + a.BH > H - 1 ? H - 1 : Yb + a.BH;
    const int sb0 = a.ystart[Ylo];
    const int sb1 = min(h - 1, a.ystart[Yhi] + 7);
    const int ns = min(sb1 - sb0 + 1, a.src_rows);
    for (int i = tid; i < ns * w; i += KF_NT) srcL[i] = img[((size_t)(sb0 + i / w) * w + i % w) * PD_CH_HEAT];
    __syncthreads();
    float M = s_red[0];
#pragma unroll
    for (int i = 1; i < KF_NT / 64; ++i) M = fmaxf(M, s_red[i]);

    // rows: T[Yr][t], Yr = Y - (Yb - 1)
    for (int i = tid; i < nrowsT * tw; i += KF_NT) {
        const int Yr = i / tw, tt = i - Yr * tw;
        const int Y = Yb - 1 + Yr, t = tt - 4;
        float v = 0.f;
        if (Y >= 0 && Y < H && t >= 0 && t < w) {
            const float* kc = a.ycoef + 8 * (size_t)Y;
            const float* sp = srcL + (size_t)(a.ystart[Y] - sb0) * w + t;
            v = kc[0] * sp[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                // rows past the staged range only ever meet a zero coefficient
                const int rr = a.ystart[Y] - sb0 + k;
                const float sv = rr < ns ? sp[(size_t)k * w] : 0.f;
                v = fmaf(kc[k], sv, v);
            }
        }
        T[i] = v;
    }
    __syncthreads();

    // columns + tolerant NMS: thread = (column group g, row lane rl)
    const int g = tid % a.wg, rl = tid / a.wg;
    if (rl >= a.RL) return;
    float kx[10][7];
    {
        const float* kp = a.xcoef + (size_t)g * 70;
#pragma unroll
        for (int jj = 0; jj < 10; ++jj)
#pragma unroll
            for (int p = 0; p < 7; ++p) kx[jj][p] = kp[jj * 7 + p];
    }
    const double eps = ldexp((double)M, -17);
    const float thr_lo = nextafterf((float)(a.thr - eps), -INFINITY);
    const float e2 = nextafterf((float)(2.0 * eps), INFINITY);
    float prev1[10], prev2[8];
#pragma unroll
    for (int j = 0; j < 10; ++j) prev1[j] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) prev2[j] = 0.f;
    const int Y0 = Yb + rl * KF_RB;  // first row this thread decides
    int32_t* plane_count = a.cand_count + b * PD_N_PARTS + part;
    int32_t* plane_list = a.cand + ((size_t)b * PD_N_PARTS + part) * a.CK;
    for (int it = 0; it < KF_RB + 2; ++it) {
        const int Y = Y0 - 1 + it;  // row being evaluated; row Y - 1 is decided
        const float* tp = T + (size_t)(Y - (Yb - 1)) * tw + g;
        float t[9];
#pragma unroll
        for (int p = 0; p < 9; ++p) t[p] = tp[p];
        float cur[10];
#pragma unroll
        for (int jj = 0; jj < 10; ++jj) {
            const int p0 = jj == 0 ? 0 : (jj == 9 ? 2 : 1);
            float v = kx[jj][0] * t[p0];
#pragma unroll
            for (int p = 1; p < 7; ++p) v = fmaf(kx[jj][p], t[p0 + p], v);
            cur[jj] = v;
        }
        const int Yc = Y - 1;
        if (it >= 2 && Yc < H) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float c = prev1[j + 1];
                const int X = 8 * g + j;
                const bool ok = X < W && c > thr_lo && c >= prev2[j] - e2 && c >= cur[j + 1] - e2 &&
                                c >= prev1[j] - e2 && c >= prev1[j + 2] - e2;
                if (ok) {
                    const int pos = atomicAdd(plane_count, 1);
                    if (pos < a.CK) plane_list[pos] = (Yc << 16) | X;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) prev2[j] = prev1[j + 1];
#pragma unroll
        for (int j = 0; j < 10; ++j) prev1[j] = cur[j];
    }
}

struct VerifyArgs {
    MapSource src;
    double thr;
    const int32_t* cand;
    const int32_t* cand_count;
    int CK;
    RawPeak* raw;
    int32_t* raw_count;
    int PK;
};

// One workgroup per (part, image); 32 lanes per candidate, 8 candidates per pass.
__global__ void __launch_bounds__(256)
k_verify(VerifyArgs a) {
    __shared__ double ypL[8][3][28];
    const int part = blockIdx.x, b = blockIdx.y;
    const int plane = b * PD_N_PARTS + part;
    const int n = a.cand_count[plane];
    if (n <= 0 || n > a.CK) return;  // overflowing planes are redone by the exact kernel
    const int tid = threadIdx.x;
    const int half = tid >> 5, l = tid & 31;
    const int H = a.src.H, W = a.src.W;
    const int32_t* list = a.cand + (size_t)plane * a.CK;
    for (int base = 0; base < n; base += 8) {
        const int ci = base + half;
        const bool active = ci < n;
        int Y = 0, X = 0;
        double u13 = 0.0;
        if (active && l < 27) {
            const int packed = list[ci];
            Y = packed >> 16;
            X = packed & 0xffff;
            ColLoader<PD_SRC_LOWRES> ld;
            ld.init(a.src, b, part, pd_reflect(X + l - 13, W));
            double u[27];
#pragma unroll
            for (int i = 0; i < 27; ++i) u[i] = ld.load(pd_reflect(Y - 13 + i, H));
            u13 = u[13];
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const int c = 12 + e;
                double t = u[c] * c_gw[12];
#pragma unroll
                for (int d = 12; d >= 1; --d) t += (u[c - d] + u[c + d]) * c_gw[12 - d];
                ypL[half][e][l] = t;
            }
        } else if (active) {
            const int packed = list[ci];
            Y = packed >> 16;
            X = packed & 0xffff;
        }
        __syncthreads();
        double S = 0.0;
        if (active && l < 5) {
            // l: 0 centre, 1 left, 2 right, 3 up, 4 down
            const int e = l == 3 ? 0 : (l == 4 ? 2 : 1);
            const int xo = l == 1 ? -1 : (l == 2 ? 1 : 0);
            const bool inside = (l == 1) ? X - 1 >= 0 : (l == 2) ? X + 1 < W : (l == 3) ? Y - 1 >= 0
                                : (l == 4) ? Y + 1 < H : true;
            if (inside) {
                const double* p = &ypL[half][e][13 + xo];
                S = p[0] * c_gw[12];
#pragma unroll
                for (int d = 12; d >= 1; --d) S += (p[-d] + p[d]) * c_gw[12 - d];
            }
        }
        const int hb = tid & 32;  // first lane of this half within the wave
        const double s0 = __shfl(S, hb + 0), s1 = __shfl(S, hb + 1), s2 = __shfl(S, hb + 2),
                     s3 = __shfl(S, hb + 3), s4 = __shfl(S, hb + 4);
        const double score = __shfl(u13, hb + 13);
        if (active && l == 0) {
            if (s0 >= s1 && s0 >= s2 && s0 >= s3 && s0 >= s4 && s0 > a.thr) {
                const int pos = atomicAdd(a.raw_count + plane, 1);
                if (pos < a.PK) {
                    RawPeak p;
                    p.y = Y;
                    p.x = X;
                    p.score = score;
                    a.raw[(size_t)plane * a.PK + pos] = p;
                }
            }
        }
        __syncthreads();
    }
}

int pd_launch_peaks_filtered(pd_ctx* ctx, double thr) {
    int rc = build_filter_tables(ctx);
    if (rc != PD_OK) return rc;
    const int B = ctx->B, H = ctx->H, W = ctx->W, h = ctx->h, w = ctx->w;
    const int CK = 4 * ctx->cfg.max_peaks_per_part;
    if (!ctx->cand || ctx->cand_cap != CK) {
        PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->cand) PD_HIP(ctx, hipFree(ctx->cand));
        if (ctx->cand_count) PD_HIP(ctx, hipFree(ctx->cand_count));
        ctx->cand = nullptr;
        ctx->cand_count = nullptr;
        PD_HIP(ctx, hipMalloc(&ctx->cand, sizeof(int32_t) * (size_t)ctx->cfg.max_batch * PD_N_PARTS * CK));
        PD_HIP(ctx, hipMalloc(&ctx->cand_count, sizeof(int32_t) * (size_t)ctx->cfg.max_batch * PD_N_PARTS));
        ctx->cand_cap = CK;
    }
    PD_HIP(ctx, hipMemsetAsync(ctx->cand_count, 0, sizeof(int32_t) * B * PD_N_PARTS, ctx->stream));
    FilterArgs f;
    f.lo = ctx->lo_heat;
    f.h = h;
    f.w = w;
    f.H = H;
    f.W = W;
    f.wg = (W + 7) / 8;
    f.RL = KF_NT / f.wg;
    const int rows_needed = (H + KF_RB - 1) / KF_RB;
    if (f.RL > rows_needed) f.RL = rows_needed;
    f.BH = f.RL * KF_RB;
    f.src_rows = (f.BH + 2 + 24) / 8 + 9;
    if (f.src_rows > h) f.src_rows = h;
    f.ystart = ctx->kf_ystart;
    f.ycoef = ctx->kf_ycoef;
    f.xcoef = ctx->kf_xcoef;
    f.thr = thr;
    f.cand = ctx->cand;
    f.cand_count = ctx->cand_count;
    f.CK = CK;
    const int nbands = (H + f.BH - 1) / f.BH;
    const size_t lds = sizeof(float) * ((size_t)f.src_rows * w + (size_t)(f.BH + 2) * (w + 8));
    if (lds > 60000) return PD_E_STATE;
    pd_prof_begin(ctx, PD_STAGE_FILTER);
    hipLaunchKernelGGL(k_prefilter, dim3(nbands, PD_N_PARTS, B), dim3(KF_NT), lds, ctx->stream, f);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    VerifyArgs v;
    v.src = pd_heat_source(ctx);
    v.thr = thr;
    v.cand = ctx->cand;
    v.cand_count = ctx->cand_count;
    v.CK = CK;
    v.raw = ctx->raw;
    v.raw_count = ctx->raw_count;
    v.PK = ctx->cfg.max_peaks_per_part;
    pd_prof_begin(ctx, PD_STAGE_VERIFY);
    hipLaunchKernelGGL(k_verify, dim3(PD_N_PARTS, B), dim3(256), 0, ctx->stream, v);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    // planes whose candidates did not fit
    return pd_launch_stream_exact(ctx, thr, 0, nullptr, ctx->cand_count, CK);
}
