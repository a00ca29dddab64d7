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
//     |S~ - S| <= 53 * 2^-24 * 1.89 * M  <  eps := 2^-17 * M,   M >= max |src| of the plane
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
#include <stdlib.h>

#include <vector>

#include "colloader.h"

#pragma clang fp contract(off)

#define KF_NT 256
#define KF_MAX_BANDS 32
typedef float float2v __attribute__((ext_vector_type(2)));

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
                xcoef[((size_t)jj * 7 + (t - t_lo)) * wg + g] = (float)v;  // [70][wg]: lane g reads coalesced
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
    {
        int32_t* hcopy = (int32_t*)realloc(ctx->kf_ystart_host, sizeof(int32_t) * H);
        if (!hcopy) return PD_E_STATE;
        memcpy(hcopy, ystart.data(), sizeof(int32_t) * H);
        ctx->kf_ystart_host = hcopy;
    }
    PD_HIP(ctx, hipMemcpy(ctx->kf_ystart, ystart.data(), sizeof(int32_t) * H, hipMemcpyHostToDevice));
    PD_HIP(ctx, hipMemcpy(ctx->kf_ycoef, ycoef.data(), sizeof(float) * 8 * H, hipMemcpyHostToDevice));
    PD_HIP(ctx, hipMemcpy(ctx->kf_xcoef, xcoef.data(), sizeof(float) * 70 * wg, hipMemcpyHostToDevice));
    ctx->kf_ok = 1;
    return PD_OK;
}

// (B, h, w, 19) -> planar (B, 19, h, w) through LDS so that both sides are coalesced,
// and max |value| over all 19 channels of each image (an upper bound of every plane's
// maximum, which is all the tolerance needs; non-negative floats order like their bit
// patterns, so the maximum is taken with an integer atomic).
#define KP_PIX 256
__global__ void __launch_bounds__(256)
k_heat_planar(const float* __restrict__ lo, int npix, float* __restrict__ planar, unsigned int* __restrict__ absmax_bits) {
    __shared__ float tile[KP_PIX * PD_CH_HEAT];
    __shared__ float s_red[4];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int p0 = blockIdx.x * KP_PIX;
    const int np = min(KP_PIX, npix - p0);
    const float* src = lo + ((size_t)b * npix + p0) * PD_CH_HEAT;
    float m = 0.f;
    for (int i = tid; i < np * PD_CH_HEAT; i += 256) {
        const float v = src[i];
        tile[i] = v;
        m = fmaxf(m, fabsf(v));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((tid & 63) == 0) s_red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0)
        atomicMax(absmax_bits + b, __float_as_uint(fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]))));
    float* dst = planar + (size_t)b * PD_CH_HEAT * npix + p0;
    for (int c = 0; c < PD_CH_HEAT; ++c)
        for (int i = tid; i < np; i += 256) dst[(size_t)c * npix + i] = tile[i * PD_CH_HEAT + c];
}

struct FilterArgs {
    const float* lo;  // planar copy (B, 19, h, w)
    int h, w, H, W, wg;
    int RL, RB, BH, src_rows;  // row lanes, rows per lane, band height, source rows staged per band
    const int32_t* ystart;
    const float* ycoef;
    const float* xcoef;
    const float* absmax;  // [B]
    double thr;
    int2* cand;            // (plane, y << 16 | x), all planes in one list
    int32_t* cand_count;   // [B][18] candidates seen per plane (may exceed CK)
    int32_t* cand_total;   // entries in cand
    int CK;
    int band_sb0[KF_MAX_BANDS];  // first source row staged for each band
    int band_ns[KF_MAX_BANDS];   // number of source rows staged
};

// One workgroup per (row band, part, image).
__global__ void __launch_bounds__(KF_NT)
k_prefilter(FilterArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x;
    const int band = blockIdx.x, part = blockIdx.y, b = blockIdx.z;
    const int h = a.h, w = a.w, H = a.H, W = a.W;
    const int Yb = band * a.BH;               // first row this band decides
    const int nrowsT = a.BH + 2;              // rows Yb-1 .. Yb+BH
    const int tw = w + 8;                     // T row: 4 zero pads each side
    float* kcL = lds;                                     // [BH+2][8] row-pass coefficients
    int* ysL = (int*)(lds + (size_t)nrowsT * 8);          // [BH+2] window starts (relative)
    float* srcL = lds + (size_t)nrowsT * 9;               // [src_rows + 8][w]
    float* T = srcL + (size_t)(a.src_rows + 8) * w;       // [BH+2][w+8]
    const float* img = a.lo + ((size_t)b * PD_CH_HEAT + part) * h * w;  // planar (B, 19, h, w)

    // Everything this band reads from memory is requested at once: its source rows,
    // and the row-pass coefficients and window starts of its BH + 2 rows (contiguous in
    // the tables), so the row pass below runs out of LDS with no dependent loads.
    const int sb0 = a.band_sb0[band], ns = a.band_ns[band];
    for (int i = tid; i < ns * w; i += KF_NT) srcL[i] = img[(size_t)sb0 * w + i];
    for (int i = tid; i < 8 * w; i += KF_NT) srcL[ns * w + i] = 0.f;  // rows past the range meet zero coefficients
    for (int i = tid; i < nrowsT * 8; i += KF_NT) {
        const int Y = Yb - 1 + (i >> 3);
        kcL[i] = (Y >= 0 && Y < H) ? a.ycoef[8 * (size_t)Y + (i & 7)] : 0.f;
    }
    for (int i = tid; i < nrowsT; i += KF_NT) {
        const int Y = Yb - 1 + i;
        ysL[i] = (Y >= 0 && Y < H) ? a.ystart[Y] - sb0 : 0;
    }
    __syncthreads();

    // rows: T[Yr][t], Yr = Y - (Yb - 1).  A thread owns one source column t and a run
    // of consecutive rows: the eight source values of its window stay in registers and
    // are re-read only when the window start moves (once per 8 rows)
    for (int i = tid; i < nrowsT * 8; i += KF_NT) {  // zero pads left and right of each T row
        const int r = i >> 3, p = i & 7;
        T[(size_t)r * tw + (p < 4 ? p : w + p)] = 0.f;
    }
    {
        const int n_chunks = KF_NT / w;  // w <= 256 is guaranteed by the table builder
        const int col = tid % w, chunk = tid / w;
        if (chunk < n_chunks) {
            const int per = (nrowsT + n_chunks - 1) / n_chunks;
            const int r_lo = chunk * per, r_hi = min(nrowsT, r_lo + per);
            int cur = -1;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
            for (int r = r_lo; r < r_hi; ++r) {
                const int y0 = ysL[r];
                if (y0 != cur) {
                    const float* sp = srcL + (size_t)y0 * w + col;
                    s0 = sp[0];
                    s1 = sp[w];
                    s2 = sp[2 * w];
                    s3 = sp[3 * w];
                    s4 = sp[4 * w];
                    s5 = sp[5 * w];
                    s6 = sp[6 * w];
                    s7 = sp[7 * w];
                    cur = y0;
                }
                const float4 ka = *(const float4*)(kcL + 8 * r), kb = *(const float4*)(kcL + 8 * r + 4);
                float v = ka.x * s0;
                v = fmaf(ka.y, s1, v);
                v = fmaf(ka.z, s2, v);
                v = fmaf(ka.w, s3, v);
                v = fmaf(kb.x, s4, v);
                v = fmaf(kb.y, s5, v);
                v = fmaf(kb.z, s6, v);
                v = fmaf(kb.w, s7, v);
                T[(size_t)r * tw + col + 4] = v;  // rows outside the image have zero coefficients: T = 0
            }
        }
    }
    __syncthreads();

    // columns + tolerant NMS: thread = (column group g, row lane rl)
    const int g = tid % a.wg, rl = tid / a.wg;
    if (rl >= a.RL) return;
    // coefficients: the eight real columns as four pairs for packed FMAs (all use
    // window positions 1..7), the two halo columns (positions 0..6 and 2..8) scalar
    float2v kp[4][7];
    float kl[7], kr[7];
#pragma unroll
    for (int p = 0; p < 7; ++p) {
        kl[p] = a.xcoef[(size_t)(0 * 7 + p) * a.wg + g];
        kr[p] = a.xcoef[(size_t)(9 * 7 + p) * a.wg + g];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            kp[q][p].x = a.xcoef[(size_t)((2 * q + 1) * 7 + p) * a.wg + g];
            kp[q][p].y = a.xcoef[(size_t)((2 * q + 2) * 7 + p) * a.wg + g];
        }
    }
    const double eps = ldexp((double)a.absmax[b], -17);
    const float thr_lo = nextafterf((float)(a.thr - eps), -INFINITY);
    const float e2 = nextafterf((float)(2.0 * eps), INFINITY);
    float prev1[10], prev2[8];
#pragma unroll
    for (int j = 0; j < 10; ++j) prev1[j] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) prev2[j] = 0.f;
    const int Y0 = Yb + rl * a.RB;  // first row this thread decides
    int32_t* plane_count = a.cand_count + b * PD_N_PARTS + part;
    for (int it = 0; it < a.RB + 2; ++it) {
        const int Y = Y0 - 1 + it;  // row being evaluated; row Y - 1 is decided
        const float* tp = T + (size_t)(Y - (Yb - 1)) * tw + g;
        float t[9];
#pragma unroll
        for (int p = 0; p < 9; ++p) t[p] = tp[p];
        float cur[10];
        // |S~| <= 1.375 * max |T| over the window (sum of |Kx| per column <= 1.375): where
        // that is below the threshold for the whole wavefront the ten sums are skipped and
        // zeros stand in -- such pixels cannot be candidates, and as neighbours of a
        // candidate (which is > thr_lo > 0) a zero passes exactly when the true value would
        float tmax = fabsf(t[0]);
#pragma unroll
        for (int p = 1; p < 9; ++p) tmax = fmaxf(tmax, fabsf(t[p]));
        if (__ballot(1.376f * tmax >= thr_lo) == 0ull) {
#pragma unroll
            for (int j = 0; j < 10; ++j) cur[j] = 0.f;
        } else {
            float2v acc[4];
            float hl = kl[0] * t[0], hr = kr[0] * t[2];
            {
                const float2v tt = {t[1], t[1]};
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = kp[q][0] * tt;
            }
#pragma unroll
            for (int p = 1; p < 7; ++p) {
                const float2v tt = {t[1 + p], t[1 + p]};
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_elementwise_fma(kp[q][p], tt, acc[q]);
                hl = fmaf(kl[p], t[p], hl);
                hr = fmaf(kr[p], t[2 + p], hr);
            }
            cur[0] = hl;
            cur[9] = hr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cur[2 * q + 1] = acc[q].x;
                cur[2 * q + 2] = acc[q].y;
            }
        }
        const int Yc = Y - 1;
        if (it >= 2 && Yc < H) {
            // most rows hold nothing above the threshold: one max + ballot skips them
            float mx = prev1[1];
#pragma unroll
            for (int j = 1; j < 8; ++j) mx = fmaxf(mx, prev1[j + 1]);
            if (__ballot(mx > thr_lo) != 0ull) {
                // branch-free test of the eight pixels: q >= 0 iff the pixel passes
                // (c >= n - e2 is tested as c + e2 >= n; the half-ulp rounding of that sum,
                // 6e-8 |c|, is far inside the slack between eps and the error bound)
                float best = -1.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float c = prev1[j + 1];
                    const float m4 = fmaxf(fmaxf(prev2[j], cur[j + 1]), fmaxf(prev1[j], prev1[j + 2]));
                    best = fmaxf(best, fminf(c - thr_lo, (c + e2) - m4));
                }
                if (best >= 0.f) {  // rare: some pixel of this thread's eight may be a candidate
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float c = prev1[j + 1];
                        const float ce = c + e2;
                        const int X = 8 * g + j;
                        const bool ok = X < W && c > thr_lo && ce >= prev2[j] && ce >= cur[j + 1] &&
                                        ce >= prev1[j] && ce >= prev1[j + 2];
                        if (ok) {
                            // the per-plane count decides overflow, the list itself is one
                            // global array so the verification is balanced over the chip
                            const int pos = atomicAdd(plane_count, 1);
                            if (pos < a.CK) {
                                const int gi = atomicAdd(a.cand_total, 1);
                                a.cand[gi] = make_int2(b * PD_N_PARTS + part, (Yc << 16) | X);
                            }
                        }
                    }
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
    const float* lo;  // planar copy (B, 19, h, w)
    int h, w, H, W;
    Up8 up8;
    double thr;
    const int2* cand;
    const int32_t* cand_count;
    const int32_t* cand_total;
    int CK;
    RawPeak* raw;
    int32_t* raw_count;
    int PK;
};

// Exact float64 decision for every candidate.  32 lanes per candidate, 8 candidates per
// workgroup pass, workgroups stride over the global list.  Lane l < 27 owns column
// X + l - 13: it forms the horizontal 4-tap sums of the (at most 8) source rows that the
// 27 bicubic rows Y-13 .. Y+13 draw on, then the 27 vertical sums and the three y-pass
// sums (rows Y-1, Y, Y+1); five lanes finish the x pass (centre and four neighbours).
__global__ void __launch_bounds__(256, 2)
k_verify(VerifyArgs a) {
    __shared__ double ypL[8][3][28];
    __shared__ float hvL[8][256];
    __shared__ float4 s_up[8];
    const int tid = threadIdx.x;
    const int half = tid >> 5, l = tid & 31;
    if (tid < 8) s_up[tid] = make_float4(a.up8.c[tid][0], a.up8.c[tid][1], a.up8.c[tid][2], a.up8.c[tid][3]);
    const int total = *a.cand_total;
    const int H = a.H, W = a.W, h = a.h, w = a.w;
    __syncthreads();
    for (int base = blockIdx.x * 8; base < total; base += gridDim.x * 8) {
        const int ci = base + half;
        bool active = ci < total;
        int Y = 0, X = 0, plane = 0;
        double u13 = 0.0;
        if (active) {
            const int2 e = a.cand[ci];
            plane = e.x;
            Y = e.y >> 16;
            X = e.y & 0xffff;
            active = a.cand_count[plane] <= a.CK;  // overflowing planes are redone by the exact kernel
        }
        if (active && l < 27) {
            const int b = plane / PD_N_PARTS, part = plane - b * PD_N_PARTS;
            const float* img = a.lo + ((size_t)b * PD_CH_HEAT + part) * h * w;
            // the reflected rows Y-13 .. Y+13 all lie in [max(Y-13, 0), min(Y+13, H-1)]
            const int rmin = max(Y - 13, 0);
            const int ymin = pd_up8_ofs(rmin);
            const int col = pd_reflect(X + l - 13, W);
            const int x0 = pd_up8_ofs(col);
            const float4 ac = s_up[pd_up8_phase(col)];
            const int xi0 = pd_clampi(x0, 0, w - 1), xi1 = pd_clampi(x0 + 1, 0, w - 1),
                      xi2 = pd_clampi(x0 + 2, 0, w - 1), xi3 = pd_clampi(x0 + 3, 0, w - 1);
            float4 sv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float* row = img + (size_t)pd_clampi(ymin + q, 0, h - 1) * w;
                sv[q] = make_float4(row[xi0], row[xi1], row[xi2], row[xi3]);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                hvL[q][tid] = ((sv[q].x * ac.x + sv[q].y * ac.y) + sv[q].z * ac.z) + sv[q].w * ac.w;
            double u[27];
#pragma unroll
            for (int i = 0; i < 27; ++i) {
                const int r = pd_reflect(Y - 13 + i, H);
                const int d = min(pd_up8_ofs(r) - ymin, 4);
                const float4 bc = s_up[pd_up8_phase(r)];
                const float v = ((hvL[d][tid] * bc.x + hvL[d + 1][tid] * bc.y) + hvL[d + 2][tid] * bc.z) +
                                hvL[d + 3][tid] * bc.w;
                u[i] = (double)v;
            }
            u13 = u[13];
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const int c = 12 + e;
                double t = u[c] * c_gw[12];
#pragma unroll
                for (int d = 12; d >= 1; --d) t += (u[c - d] + u[c + d]) * c_gw[12 - d];
                ypL[half][e][l] = t;
            }
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
        if (ctx->absmax) PD_HIP(ctx, hipFree(ctx->absmax));
        ctx->cand = nullptr;
        ctx->cand_count = nullptr;
        ctx->absmax = nullptr;
        PD_HIP(ctx, hipMalloc(&ctx->cand, sizeof(int2) * (size_t)ctx->cfg.max_batch * PD_N_PARTS * CK));
        PD_HIP(ctx, hipMalloc(&ctx->cand_count, sizeof(int32_t) * ((size_t)ctx->cfg.max_batch * PD_N_PARTS + 1)));
        PD_HIP(ctx, hipMalloc(&ctx->absmax, sizeof(float) * (size_t)ctx->cfg.max_batch));
        ctx->cand_cap = CK;
    }
    {
        const size_t need = sizeof(float) * (size_t)B * PD_CH_HEAT * h * w;
        if (ctx->heat_planar_cap < need) {
            PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->heat_planar) PD_HIP(ctx, hipFree(ctx->heat_planar));
            ctx->heat_planar = nullptr;
            ctx->heat_planar_cap = 0;
            PD_HIP(ctx, hipMalloc(&ctx->heat_planar, need));
            ctx->heat_planar_cap = need;
        }
    }
    FilterArgs f;
    f.lo = ctx->heat_planar;
    f.h = h;
    f.w = w;
    f.H = H;
    f.W = W;
    f.wg = (W + 7) / 8;
    f.RL = KF_NT / f.wg;
    // rows per thread: the choice that wastes the fewest row evaluations (each thread
    // evaluates RB + 2 rows, bands are RL * RB rows) within the LDS budget
    int best_rb = 0;
    long best_cost = 0;
    for (int rb = 8; rb <= 64; ++rb) {
        const int bh = f.RL * rb;
        const long lds_b = 4L * ((long)(bh + 2) * 9 + ((bh + 26) / 8 + 17) * w + (long)(bh + 2) * (w + 8));
        if (lds_b > 40000) break;
        const long cost = (long)((H + bh - 1) / bh) * f.RL * (rb + 2);
        if (best_rb == 0 || cost < best_cost) {
            best_rb = rb;
            best_cost = cost;
        }
    }
    if (best_rb == 0) return PD_E_STATE;
    f.RB = best_rb;
    f.BH = f.RL * f.RB;
    f.src_rows = (f.BH + 26) / 8 + 9;
    if (f.src_rows > h) f.src_rows = h;
    f.ystart = ctx->kf_ystart;
    f.ycoef = ctx->kf_ycoef;
    f.xcoef = ctx->kf_xcoef;
    f.absmax = ctx->absmax;
    f.thr = thr;
    f.cand = (int2*)ctx->cand;
    f.cand_count = ctx->cand_count;
    f.cand_total = ctx->cand_count + (size_t)B * PD_N_PARTS;  // one more counter behind the planes'
    f.CK = CK;
    const int nbands = (H + f.BH - 1) / f.BH;
    if (nbands > KF_MAX_BANDS) return PD_E_STATE;
    for (int band = 0; band < nbands; ++band) {
        const int Yb = band * f.BH;
        const int Ylo = Yb - 1 < 0 ? 0 : Yb - 1;
        const int Yhi = Yb + f.BH > H - 1 ? H - 1 : Yb + f.BH;
        const int sb0 = ctx->kf_ystart_host[Ylo];
        int sb1 = ctx->kf_ystart_host[Yhi] + 7;
        if (sb1 > h - 1) sb1 = h - 1;
        int ns = sb1 - sb0 + 1;
        if (ns > f.src_rows) return PD_E_STATE;
        f.band_sb0[band] = sb0;
        f.band_ns[band] = ns;
    }
    // LDS: coefficients + starts (9 per row), source rows (+8 zero rows), T rows
    const size_t lds = sizeof(float) * ((size_t)(f.BH + 2) * 9 + (size_t)(f.src_rows + 8) * w + (size_t)(f.BH + 2) * (w + 8));
    if (lds > 64000) return PD_E_STATE;
    PD_HIP(ctx, hipMemsetAsync(ctx->cand_count, 0, sizeof(int32_t) * ((size_t)B * PD_N_PARTS + 1), ctx->stream));
    PD_HIP(ctx, hipMemsetAsync(ctx->absmax, 0, sizeof(float) * B, ctx->stream));
    pd_prof_begin(ctx, PD_STAGE_PLANAR);
    hipLaunchKernelGGL(k_heat_planar, dim3((h * w + KP_PIX - 1) / KP_PIX, B), dim3(256), 0, ctx->stream, ctx->lo_heat,
                       h * w, ctx->heat_planar, (unsigned int*)ctx->absmax);
    pd_prof_end(ctx);
    pd_prof_begin(ctx, PD_STAGE_FILTER);
    hipLaunchKernelGGL(k_prefilter, dim3(nbands, PD_N_PARTS, B), dim3(KF_NT), lds, ctx->stream, f);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    VerifyArgs v;
    v.lo = ctx->heat_planar;
    v.h = h;
    v.w = w;
    v.H = H;
    v.W = W;
    pd_up8_fill(&v.up8);
    v.thr = thr;
    v.cand = (const int2*)ctx->cand;
    v.cand_count = ctx->cand_count;
    v.cand_total = f.cand_total;
    v.CK = CK;
    v.raw = ctx->raw;
    v.raw_count = ctx->raw_count;
    v.PK = ctx->cfg.max_peaks_per_part;
    // the list length is only known on the device: a fixed grid strides over it
    int vblocks = B * 12;
    if (vblocks > 1024) vblocks = 1024;
    pd_prof_begin(ctx, PD_STAGE_VERIFY);
    hipLaunchKernelGGL(k_verify, dim3(vblocks), dim3(256), 0, ctx->stream, v);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    // planes whose candidates did not fit
    return pd_launch_stream_exact(ctx, thr, 0, nullptr, ctx->cand_count, CK);
}
