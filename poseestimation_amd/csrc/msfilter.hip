// Peak stage of a multi-scale source (predict.py:49-72 feeding predict.py:88-111, twin
// utils/postprocess.py:26-75) without the averaged heatmaps ever being stored: candidate cells
// from a bound, a float32 filter on those cells, exact float64 verification of the candidates.
// Results are identical to the exact kernel run over the stored float64 average.
//
// The smoothed average is linear in the network outputs src_m of the n test scales:
//     S[Y][X] = sum_m sum_s sum_t Ky_m[Y][s] * Kx_m[X][t] * src_m[s][t],
// K_m = Gaussian (25 taps, reflect) after the second cubic resize (4 taps, clamped to the crop)
// after the x8 cubic resize (4 taps, clamped), per axis, with 1/n folded into Ky_m.  For the
// ten outputs 8c-1 .. 8c+8 of a cell (its own eight and one halo output either side) the
// sources with a non-zero weight lie in a window of at most MSF_T = 16 per scale (a downscale
// by about 2.2 at most; beyond that, or for maps too large for the staging areas, the caller
// writes the average out and runs the full-resolution filter on it).
//
// Tolerance.  With kappa = 1.375 (sum of the absolute cubic weights of one pass) and
// M >= |src_m| for all m of the plane, in units of u = 2^-24 M:
//   reference: four float32 passes, each (3 adds + 4 products) on values that grow by kappa per
//   pass, then the float32 division: at most 17 kappa^4 u = 61 u on the average, which the
//   float64 Gaussian (positive weights of sum 1) carries over unchanged;
//   filter: weights rounded to float32 (2 kappa^4), two chains of 16 fused multiply-adds per
//   scale (32 kappa^4), the sum over the scales (4 kappa^4): 136 u.
// Together under 197 u; the filter works with eps = 2^-15 M = 512 u: a pixel can only be a peak
// if S~ > thr - eps and S~ >= S~(neighbour) - 2 eps, and every such pixel is verified exactly.
// A cell can only hold such a pixel if the bound
//     U = sum_m sum_s sum_t By_m[s] * Bx_m[t] * |src_m[s][t]|,  B = largest |weight| of the cell's own 8 outputs,
// reaches thr - eps (a factor 1 + 1e-5 covers the rounding of the bound's own chains).
#include <math.h>
#include <stdlib.h>

#include <vector>

#include "colloader.h"
#include "multiscale.h"
#include "prefilter.h"

#pragma clang fp contract(off)

#define MSF_T 16        // sources per axis and scale that the ten outputs of a cell may draw on
#define MSF_EPS_SHIFT 15
#define MSF_ROW 20      // LDS row stride of the 16-wide staging rows, in floats (16-byte aligned, off the bank period)
#define MSV_P 64        // largest side of the x8-map patch the verification stages per scale
#define MSV_NJ 14       // source rows such a patch draws on, at most
#define MSV_NT 256
#define MSP_PIX 256

static const double h_gw[PD_GTAPS] = PD_GAUSS_WEIGHTS;

struct MsAxisTab {
    const float* cw;     // [cells][10][MSF_T] weight of source lo + p in output 8c - 1 + j
    const float* bw;     // [cells][MSF_T]     largest |cw| over j = 1..8
    const int32_t* lo;   // [cells]            first source of the cell's window
};
struct MsScale {
    const float* planar;  // (B, 18, hs, ws) float32 copy of the heatmaps of this scale
    int hs, ws, crop_h, crop_w, identity;
    int nq;               // float4 words of a table row that hold non-zero weights (both axes): 1..4
    MsAxisTab y, x;
    CubicTable ty, tx;    // second resize crop -> (H, W)
};
struct MsArgs {
    MsScale s[PD_MAX_SCALES];
    int n, pow2;
    float n_div, n_rcp;
    int n_planes, H, W, hc, wc;
    int lds_a, lds_r;      // floats of the two staging areas of k_ms_cells
    double thr;
    float2* slack;         // [B * 18] (thr_lo, e2) of the candidate test, per plane
    int32_t* cells;        // KF_NSEG segments of active cells (plane << 14 | cy << 7 | cx)
    int seg_cap;
    int32_t* n_cells;      // [KF_NSEG]
    int2* cand;
    int cand_seg_cap;
    int32_t* cand_count;
    int32_t* cand_n;
    int CK;
    RawPeak* raw;
    int32_t* raw_count;
    int PK;
    Up8 up8;
};

// ---------------------------------------------------------------------------------
// Host: the composite operators in cell form.
// ---------------------------------------------------------------------------------
static int host_reflect(int i, int n) {
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - 1 - i : i;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// One axis of one scale: K[o][s], o < L outputs, s < l sources = Gaussian after the second
// resize (crop -> L) after the x8 resize (l -> 8 l), times `factor`.
static void axis_composite(int l, int crop, int L, bool identity, double factor, std::vector<double>& K) {
    std::vector<int32_t> o8((size_t)8 * l), o2(L);
    std::vector<float> c8((size_t)32 * l), c2((size_t)4 * L);
    pd_host_cubic_table(l, 8 * l, 0.125, o8.data(), c8.data());
    if (!identity) pd_host_cubic_table(crop, L, -1.0, o2.data(), c2.data());
    std::vector<double> A((size_t)L * l, 0.0);
    for (int o = 0; o < L; ++o) {
        if (identity) {
            for (int q = 0; q < 4; ++q) A[(size_t)o * l + clampi(o8[o] + q, 0, l - 1)] += (double)c8[4 * o + q];
        } else {
            for (int k = 0; k < 4; ++k) {
                const int r = clampi(o2[o] + k, 0, crop - 1);
                for (int q = 0; q < 4; ++q)
                    A[(size_t)o * l + clampi(o8[r] + q, 0, l - 1)] += (double)c2[4 * o + k] * (double)c8[4 * r + q];
            }
        }
    }
    K.assign((size_t)L * l, 0.0);
    for (int o = 0; o < L; ++o)
        for (int d = -PD_GR; d <= PD_GR; ++d) {
            const double g = h_gw[PD_GR - (d < 0 ? -d : d)] * factor;
            const double* src = &A[(size_t)host_reflect(o + d, L) * l];
            double* dst = &K[(size_t)o * l];
            for (int s = 0; s < l; ++s) dst[s] += g * src[s];
        }
}

// Cell form of K; false if some cell's window is wider than MSF_T.
static bool axis_cells(int L, int l, const std::vector<double>& K, std::vector<float>& cw, std::vector<float>& bw,
                       std::vector<int32_t>& lo, int* widest) {
    const int nc = (L + 7) / 8;
    cw.assign((size_t)nc * 10 * MSF_T, 0.f);
    bw.assign((size_t)nc * MSF_T, 0.f);
    lo.assign(nc, 0);
    for (int c = 0; c < nc; ++c) {
        int first = l, last = -1;
        for (int j = 0; j < 10; ++j) {
            const int o = 8 * c - 1 + j;
            if (o < 0 || o >= L) continue;
            for (int s = 0; s < l; ++s)
                if (K[(size_t)o * l + s] != 0.0) {
                    if (s < first) first = s;
                    if (s > last) last = s;
                }
        }
        if (last < first) continue;  // (no output of the cell lies in the map: cannot happen for c < nc)
        if (last - first + 1 > MSF_T) return false;
        if (last - first + 1 > *widest) *widest = last - first + 1;
        lo[c] = first;
        for (int j = 0; j < 10; ++j) {
            const int o = 8 * c - 1 + j;
            if (o < 0 || o >= L) continue;
            for (int s = first; s <= last; ++s) {
                const float fv = (float)K[(size_t)o * l + s];
                cw[((size_t)c * 10 + j) * MSF_T + (s - first)] = fv;
                if (j >= 1 && j <= 8 && fabsf(fv) > bw[(size_t)c * MSF_T + (s - first)]) bw[(size_t)c * MSF_T + (s - first)] = fabsf(fv);
            }
        }
    }
    return true;
}

// Largest extent of the x8-map patch that a window of 27 outputs draws on (second resize crop -> L).
static int max_patch_extent(int crop, int L) {
    std::vector<int32_t> o2(L);
    std::vector<float> c2((size_t)4 * L);
    pd_host_cubic_table(crop, L, -1.0, o2.data(), c2.data());
    int worst = 0;
    for (int Y = 0; Y < L; ++Y) {
        const int a = clampi(o2[Y - 13 < 0 ? 0 : Y - 13], 0, crop - 1);
        const int b = clampi(o2[Y + 13 > L - 1 ? L - 1 : Y + 13] + 3, 0, crop - 1);
        if (b - a + 1 > worst) worst = b - a + 1;
    }
    return worst;
}

// Test hook (host only, no GPU): the composite operator of one axis of one scale and its cell form,
// as build_ms_tables makes them.  K_out [L][l] float64; cw_out [cells][10][16], bw_out [cells][16],
// lo_out [cells] with cells = (L + 7) / 8; any output may be NULL.  Returns the widest window, or
// -1 when a window exceeds 16 sources (the geometry the sparse stage does not take).
extern "C" int pd_test_ms_axis(int l, int crop, int L, int identity, double factor, double* K_out, float* cw_out, float* bw_out,
                               int32_t* lo_out) {
    if (l < 1 || crop < 1 || L < 1 || crop > 8 * l) return -2;
    std::vector<double> K;
    axis_composite(l, crop, L, identity != 0, factor, K);
    if (K_out) memcpy(K_out, K.data(), K.size() * sizeof(double));
    std::vector<float> cw, bw;
    std::vector<int32_t> lo;
    int widest = 0;
    if (!axis_cells(L, l, K, cw, bw, lo, &widest)) return -1;
    if (cw_out) memcpy(cw_out, cw.data(), cw.size() * sizeof(float));
    if (bw_out) memcpy(bw_out, bw.data(), bw.size() * sizeof(float));
    if (lo_out) memcpy(lo_out, lo.data(), lo.size() * sizeof(int32_t));
    return widest;
}

// Table offsets (floats / ints) of scale m, axis ax (0 = y, 1 = x).
struct MsLayout {
    size_t cw[PD_MAX_SCALES][2], bw[PD_MAX_SCALES][2], lo[PD_MAX_SCALES][2], n_float, n_int;
};
static MsLayout ms_layout(int n, int hc, int wc) {
    MsLayout t;
    memset(&t, 0, sizeof(t));
    size_t f = 0, i = 0;
    for (int m = 0; m < n; ++m)
        for (int ax = 0; ax < 2; ++ax) {
            const size_t nc = ax == 0 ? hc : wc;
            t.cw[m][ax] = f;
            f += nc * 10 * MSF_T;
            t.bw[m][ax] = f;
            f += nc * MSF_T;
            t.lo[m][ax] = i;
            i += nc;
        }
    t.n_float = f;
    t.n_int = i;
    return t;
}

static int build_ms_tables(pd_ctx* ctx) {
    const int n = ctx->ms_n, H = ctx->H, W = ctx->W;
    int key[3 + 4 * PD_MAX_SCALES];
    memset(key, 0, sizeof(key));
    key[0] = n;
    key[1] = H;
    key[2] = W;
    for (int m = 0; m < n; ++m) {
        key[3 + 4 * m] = ctx->ms_hs[m];
        key[4 + 4 * m] = ctx->ms_ws[m];
        key[5 + 4 * m] = ctx->ms_crop_h[m];
        key[6 + 4 * m] = ctx->ms_crop_w[m];
    }
    if (ctx->msf_key[0] == n && memcmp(key, ctx->msf_key, sizeof(key)) == 0) return ctx->msf_ok ? PD_OK : PD_E_STATE;
    memcpy(ctx->msf_key, key, sizeof(key));
    ctx->msf_ok = 0;
    ctx->gen++;
    const int hc = (H + 7) / 8, wc = (W + 7) / 8;
    if (hc > 128 || wc > 128 || H >= 65536 || W >= 65536) return PD_E_STATE;  // cells pack 7 bits per coordinate
    // staging areas of the bound kernel: |src| of one scale, its row sums, the bounds of every cell
    size_t lds_a = 0, lds_r = 0;
    for (int m = 0; m < n; ++m) {
        lds_a = std::max(lds_a, (size_t)ctx->ms_hs[m] * ctx->ms_ws[m]);
        lds_r = std::max(lds_r, (size_t)ctx->ms_hs[m] * wc);
    }
    // (beyond 64 KB a kernel's dynamic LDS has to be asked for: see the launch; a CU has 160 KB)
    if ((lds_a + lds_r + (size_t)hc * wc) * sizeof(float) > 150 * 1024) return PD_E_STATE;
    const MsLayout lay = ms_layout(n, hc, wc);
    std::vector<float> tab(lay.n_float);
    std::vector<int32_t> los(lay.n_int);
    int widest[PD_MAX_SCALES];
    memset(widest, 0, sizeof(widest));
    for (int m = 0; m < n; ++m) {
        const bool identity = ctx->ms_crop_h[m] == H && ctx->ms_crop_w[m] == W;
        if (!identity && (max_patch_extent(ctx->ms_crop_h[m], H) > MSV_P || max_patch_extent(ctx->ms_crop_w[m], W) > MSV_P)) return PD_E_STATE;
        for (int ax = 0; ax < 2; ++ax) {
            const int l = ax == 0 ? ctx->ms_hs[m] : ctx->ms_ws[m], crop = ax == 0 ? ctx->ms_crop_h[m] : ctx->ms_crop_w[m];
            const int L = ax == 0 ? H : W;
            std::vector<double> K;
            axis_composite(l, crop, L, identity, ax == 0 ? 1.0 / (double)n : 1.0, K);
            std::vector<float> cw, bw;
            std::vector<int32_t> lo;
            if (!axis_cells(L, l, K, cw, bw, lo, &widest[m])) return PD_E_STATE;
            memcpy(&tab[lay.cw[m][ax]], cw.data(), cw.size() * sizeof(float));
            memcpy(&tab[lay.bw[m][ax]], bw.data(), bw.size() * sizeof(float));
            memcpy(&los[lay.lo[m][ax]], lo.data(), lo.size() * sizeof(int32_t));
        }
    }
    int rc = pd_grow(ctx, (void**)&ctx->msf_tab, &ctx->msf_tab_cap, tab.size() * sizeof(float));
    if (rc != PD_OK) return rc;
    rc = pd_grow(ctx, (void**)&ctx->msf_lo, &ctx->msf_lo_cap, los.size() * sizeof(int32_t));
    if (rc != PD_OK) return rc;
    PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PD_HIP(ctx, hipMemcpy(ctx->msf_tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    PD_HIP(ctx, hipMemcpy(ctx->msf_lo, los.data(), los.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    ctx->msf_lds_a = (int)lds_a;
    ctx->msf_lds_r = (int)lds_r;
    for (int m = 0; m < n; ++m) ctx->msf_nq[m] = (widest[m] + 3) / 4;
    ctx->msf_ok = 1;
    return PD_OK;
}

// ---------------------------------------------------------------------------------
// (B, hs, ws, 19) -> planar (B, 18, hs, ws) for every scale, one launch; through LDS so that
// both sides are coalesced.  Channel 18 (the background map, which predict.py:91 never reads)
// is dropped here.
// ---------------------------------------------------------------------------------
struct MsPlanarArgs {
    const float* src[PD_MAX_SCALES];
    float* dst[PD_MAX_SCALES];
    int npix[PD_MAX_SCALES];
};

__global__ void __launch_bounds__(256)
k_ms_planar(MsPlanarArgs a) {
    __shared__ __align__(16) float tile[MSP_PIX * PD_CH_HEAT];
    const int m = blockIdx.z, b = blockIdx.y, tid = threadIdx.x;
    const int npix = a.npix[m];
    const int p0 = blockIdx.x * MSP_PIX;
    if (p0 >= npix) return;
    const int np = min(MSP_PIX, npix - p0);
    const float* src = a.src[m] + ((size_t)b * npix + p0) * PD_CH_HEAT;
    const int n = np * PD_CH_HEAT;
    int done = 0;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int n4 = n >> 2;
        for (int i = tid; i < n4; i += 256) ((float4*)tile)[i] = ((const float4*)src)[i];
        done = 4 * n4;
    }
    for (int i = done + tid; i < n; i += 256) tile[i] = src[i];
    __syncthreads();
    float* dst = a.dst[m] + (size_t)b * PD_N_PARTS * npix + p0;
    for (int c = 0; c < PD_N_PARTS; ++c)
        for (int i = tid; i < np; i += 256) dst[(size_t)c * npix + i] = tile[i * PD_CH_HEAT + c];
}

// ---------------------------------------------------------------------------------
// Active cells: one workgroup per (part, image) plane.  Per scale the plane's |src| is staged
// in LDS, the row sums R[s][cx] = sum_t Bx[cx][t] |src[s][t]| are formed for every source row and
// cell column, then every cell adds sum_s By[cy][s] R[s][cx] to its bound.
// ---------------------------------------------------------------------------------
// One scale of the bound for a thread's cell column: the row sums of its column for every fourth
// source row, then (behind a barrier) the bounds of its cells.  Free of branches, so that the
// loads of the row weights of all sixteen cell rows can be in flight at once.
template <int NQ>
__device__ __forceinline__ void ms_cells_scale(const MsScale& sc, const float* A, float* R, int hc, int wc, int cx, int wv, bool act,
                                               float (&U)[16]) {
    const int hs = sc.hs, ws = sc.ws;
    const int cxc = min(cx, wc - 1);
    float bw[4 * NQ];
    int col[4 * NQ];
    {
        const int lo = sc.x.lo[cxc];
        const float4* b4 = (const float4*)(sc.x.bw + (size_t)cxc * MSF_T);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4 w = b4[q];
            bw[4 * q] = w.x;
            bw[4 * q + 1] = w.y;
            bw[4 * q + 2] = w.z;
            bw[4 * q + 3] = w.w;
#pragma unroll
            for (int k = 0; k < 4; ++k) col[4 * q + k] = min(lo + 4 * q + k, ws - 1);
        }
    }
    __syncthreads();  // A is staged
    if (act)
        for (int s = wv; s < hs; s += 4) {
            const float* row = A + s * ws;
            float acc = 0.f;
#pragma unroll
            for (int p = 0; p < 4 * NQ; ++p) acc = fmaf(bw[p], row[col[p]], acc);
            R[s * wc + cx] = acc;
        }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int cy = __builtin_amdgcn_readfirstlane(min(wv + 4 * k, hc - 1));
        const int lo = sc.y.lo[cy];
        const float4* by = (const float4*)(sc.y.bw + (size_t)cy * MSF_T);
        float acc = U[k];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4 w = by[q];
            acc = fmaf(w.x, R[min(lo + 4 * q, hs - 1) * wc + cxc], acc);
            acc = fmaf(w.y, R[min(lo + 4 * q + 1, hs - 1) * wc + cxc], acc);
            acc = fmaf(w.z, R[min(lo + 4 * q + 2, hs - 1) * wc + cxc], acc);
            acc = fmaf(w.w, R[min(lo + 4 * q + 3, hs - 1) * wc + cxc], acc);
        }
        U[k] = acc;  // (cell rows beyond the map repeat the last one; they are never listed)
    }
}

// Maps of at most 64 x 64 cells: lane = cell column, wavefront = every fourth row.  A thread keeps
// its column's bound weights and clamped source columns in registers for the row sums and the
// bounds of its (at most 16) cells in registers as well.
__global__ void __launch_bounds__(256)
k_ms_cells64(MsArgs a) {
    extern __shared__ __align__(16) float lds[];
    __shared__ float s_red[4];
    __shared__ int s_cnt[4];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per_xcd = gridDim.x / 8;
    const int plane = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;  // the planes of an image on one XCD
    if (plane >= a.n_planes) return;
    const int hc = a.hc, wc = a.wc;
    const int cx = lane;
    const bool act = cx < wc;
    float* A = lds;           // [hs][ws] of the scale at hand
    float* R = A + a.lds_a;   // [hs][wc]
    float U[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) U[k] = 0.f;
    float mx = 0.f;
#pragma unroll 1
    for (int m = 0; m < a.n; ++m) {
        const MsScale& sc = a.s[m];
        const int hs = sc.hs, ws = sc.ws, nq = sc.nq;
        const float* src = sc.planar + (size_t)plane * hs * ws;
        // (eight loads per thread in flight at a time)
        for (int i0 = 0; i0 < hs * ws; i0 += 256 * 8) {
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + 256 * k + tid;
                t[k] = i < hs * ws ? src[i] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + 256 * k + tid;
                if (i < hs * ws) {
                    A[i] = fabsf(t[k]);
                    mx = absmax_step(mx, t[k]);
                }
            }
        }
        if (nq <= 2)
            ms_cells_scale<2>(sc, A, R, hc, wc, cx, wv, act, U);
        else if (nq == 3)
            ms_cells_scale<3>(sc, A, R, hc, wc, cx, wv, act, U);
        else
            ms_cells_scale<4>(sc, A, R, hc, wc, cx, wv, act, U);
        // (the next scale's first barrier stands between these reads of R and its next writes; A was
        // last read before the second barrier)
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (lane == 0) s_red[wv] = mx;
    __syncthreads();
    float thr_lo = 0.f, e2 = 0.f;
    int finite = 1;
    if (lane == 0) {
        const float pm = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        filter_slack(a.thr, pm, &thr_lo, &e2, MSF_EPS_SHIFT);
        finite = absmax_finite(pm);
        if (wv == 0) {
            a.slack[plane] = make_float2(thr_lo, e2);
            if (!finite) atomicOr(a.cand_count + plane, KF_OVERFLOW);  // the exact kernel takes this plane
        }
    }
    thr_lo = __shfl(thr_lo, 0);
    finite = __shfl(finite, 0);
    unsigned mask = 0u;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (act && finite && wv + 4 * k < hc && U[k] * 1.00001f >= thr_lo) mask |= 1u << k;
    const int mine = __builtin_popcount(mask);
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_cnt[wv] = incl;
    __syncthreads();
    const int seg = plane % KF_NSEG;
    if (tid == 0) {
        const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        s_base = total ? atomicAdd(a.n_cells + seg, total) : 0;
    }
    __syncthreads();
    int at = s_base + incl - mine;
    for (int q = 0; q < wv; ++q) at += s_cnt[q];
    int32_t* segment = a.cells + (size_t)seg * a.seg_cap;
    while (mask) {
        const int k = __builtin_ctz(mask);
        mask &= mask - 1u;
        segment[at++] = (plane << 14) | ((wv + 4 * k) << 7) | cx;
    }
}

__global__ void __launch_bounds__(256)
k_ms_cells(MsArgs a) {
    extern __shared__ __align__(16) float lds[];
    __shared__ float s_red[4];
    __shared__ int s_cnt[4];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per_xcd = gridDim.x / 8;
    const int plane = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;  // the planes of an image on one XCD
    if (plane >= a.n_planes) return;
    const int hc = a.hc, wc = a.wc, n_cells = hc * wc;
    float* U = lds;           // [hc][wc]
    float* A = U + n_cells;   // [hs][ws] of the scale at hand
    float* R = A + a.lds_a;   // [hs][wc]
    for (int e = tid; e < n_cells; e += 256) U[e] = 0.f;
    float mx = 0.f;
#pragma unroll 1
    for (int m = 0; m < a.n; ++m) {
        const MsScale& sc = a.s[m];
        const int hs = sc.hs, ws = sc.ws;
        const float* src = sc.planar + (size_t)plane * hs * ws;
        for (int i = tid; i < hs * ws; i += 256) {
            const float v = fabsf(src[i]);
            A[i] = v;
            mx = absmax_step(mx, v);
        }
        __syncthreads();
        for (int e = tid; e < hs * wc; e += 256) {
            const int s = e / wc, cx = e - s * wc;
            const int lo = sc.x.lo[cx];
            const float4* bw = (const float4*)(sc.x.bw + (size_t)cx * MSF_T);
            const float* row = A + s * ws;
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < MSF_T / 4; ++q) {
                const float4 w = bw[q];
                acc = fmaf(w.x, row[min(lo + 4 * q, ws - 1)], acc);
                acc = fmaf(w.y, row[min(lo + 4 * q + 1, ws - 1)], acc);
                acc = fmaf(w.z, row[min(lo + 4 * q + 2, ws - 1)], acc);
                acc = fmaf(w.w, row[min(lo + 4 * q + 3, ws - 1)], acc);
            }
            R[e] = acc;
        }
        __syncthreads();
        // (a cell is owned by the same thread in every scale: no barrier between the scales' sums)
        for (int e = tid; e < n_cells; e += 256) {
            const int cy = e / wc, cx = e - cy * wc;
            const int lo = sc.y.lo[cy];
            const float4* bw = (const float4*)(sc.y.bw + (size_t)cy * MSF_T);
            float acc = U[e];
#pragma unroll
            for (int q = 0; q < MSF_T / 4; ++q) {
                const float4 w = bw[q];
                acc = fmaf(w.x, R[min(lo + 4 * q, hs - 1) * wc + cx], acc);
                acc = fmaf(w.y, R[min(lo + 4 * q + 1, hs - 1) * wc + cx], acc);
                acc = fmaf(w.z, R[min(lo + 4 * q + 2, hs - 1) * wc + cx], acc);
                acc = fmaf(w.w, R[min(lo + 4 * q + 3, hs - 1) * wc + cx], acc);
            }
            U[e] = acc;
        }
        // the next scale's first barrier stands between this loop's reads of R and the next writes
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (lane == 0) s_red[wv] = mx;
    __syncthreads();
    float thr_lo = 0.f, e2 = 0.f;
    int finite = 1;
    if (lane == 0) {
        const float pm = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        filter_slack(a.thr, pm, &thr_lo, &e2, MSF_EPS_SHIFT);
        finite = absmax_finite(pm);
        if (wv == 0) {
            a.slack[plane] = make_float2(thr_lo, e2);
            if (!finite) atomicOr(a.cand_count + plane, KF_OVERFLOW);  // the exact kernel takes this plane
        }
    }
    thr_lo = __shfl(thr_lo, 0);
    finite = __shfl(finite, 0);
    // the list: a reservation per workgroup in the segment of the plane, entries in cell order per thread
    int mine = 0;
    for (int e = tid; e < n_cells; e += 256) mine += finite && U[e] * 1.00001f >= thr_lo;
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_cnt[wv] = incl;
    __syncthreads();
    const int seg = plane % KF_NSEG;
    if (tid == 0) {
        const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        s_base = total ? atomicAdd(a.n_cells + seg, total) : 0;
    }
    __syncthreads();
    int at = s_base + incl - mine;
    for (int q = 0; q < wv; ++q) at += s_cnt[q];
    int32_t* segment = a.cells + (size_t)seg * a.seg_cap;
    for (int e = tid; e < n_cells; e += 256)
        if (finite && U[e] * 1.00001f >= thr_lo) {
            const int cy = e / wc, cx = e - cy * wc;
            segment[at++] = (plane << 14) | (cy << 7) | cx;
        }
}

// ---------------------------------------------------------------------------------
// Candidate filter: one wavefront per listed cell.  Per scale the 16 x 16 source window and the
// cell's two weight tables go to wave-private LDS; x pass: 16 source rows x 10 output columns,
// 16 taps each; y pass: 10 x 10 outputs, 16 taps each (rows of 16 floats read as four 16-byte
// words).  The sums of the scales are added up in two registers per lane; tolerant NMS of the
// cell's own 8 x 8 pixels, one per lane.
// ---------------------------------------------------------------------------------
struct MsFilterShared {
    __attribute__((aligned(16))) float win[4][MSF_T * MSF_ROW];   // source window, row s
    __attribute__((aligned(16))) float wx[4][10 * MSF_ROW];       // weights of output column i
    __attribute__((aligned(16))) float wy[4][10 * MSF_ROW];       // weights of output row j
    __attribute__((aligned(16))) float tt[4][10 * MSF_ROW];       // x-pass sums, transposed: output column i, source row s
    float sm[4][100];
    int2 cand[KF_LCAND];
    int n_cand;
    int base;
};

// What one lane fetches for one (cell, scale): four window values and three weights of either table.
struct MsFetch {
    float v[4], gx[3], gy[3];
};

__device__ __forceinline__ void ms_fetch(const MsScale& sc, int plane, int cy, int cx, int lane, MsFetch& f) {
    const int hs = sc.hs, ws = sc.ws;
    const float* src = sc.planar + (size_t)plane * hs * ws;
    const int loy = sc.y.lo[cy], lox = sc.x.lo[cx];
    const float* gx = sc.x.cw + (size_t)cx * 10 * MSF_T;
    const float* gy = sc.y.cw + (size_t)cy * 10 * MSF_T;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int el = lane + 64 * k, s = el >> 4, p = el & 15;
        f.v[k] = src[(size_t)min(loy + s, hs - 1) * ws + min(lox + p, ws - 1)];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int el = lane + 64 * k;
        f.gx[k] = el < 10 * MSF_T ? gx[el] : 0.f;
        f.gy[k] = el < 10 * MSF_T ? gy[el] : 0.f;
    }
}

template <int NQ>
__device__ __forceinline__ float dotq(const float* w, const float4 (&v)[NQ]) {
    const float4* w4 = (const float4*)w;
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float4 a = w4[q];
        acc = fmaf(a.x, v[q].x, acc);
        acc = fmaf(a.y, v[q].y, acc);
        acc = fmaf(a.z, v[q].z, acc);
        acc = fmaf(a.w, v[q].w, acc);
    }
    return acc;
}

// The two passes of one scale from the staged window and weights.  x pass: lane = (source row s,
// group g), the row read once, outputs i = g, g + 4, g + 8.  y pass: lane = (output column i,
// group jg), the column of x-pass sums read once, outputs j = jg, jg + 6.
template <int NQ>
__device__ __forceinline__ void ms_passes(const float* win, const float* wx, const float* wy, float* tt, int lane, float& acc0,
                                          float& acc1) {
    {
        const int s = lane >> 2, g = lane & 3;
        float4 r[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) r[q] = ((const float4*)(win + s * MSF_ROW))[q];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int i = g + 4 * t;
            if (i < 10) tt[i * MSF_ROW + s] = dotq<NQ>(wx + i * MSF_ROW, r);
        }
    }
    pd_wave_sync();
    if (lane < 60) {
        const int jg = lane / 10, i = lane - 10 * jg;
        float4 c[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) c[q] = ((const float4*)(tt + i * MSF_ROW))[q];
        acc0 += dotq<NQ>(wy + jg * MSF_ROW, c);
        if (jg < 4) acc1 += dotq<NQ>(wy + (jg + 6) * MSF_ROW, c);
    }
}

__global__ void __launch_bounds__(256)
k_ms_cell_filter(MsArgs a) {
    __shared__ MsFilterShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int seg = blockIdx.x % KF_NSEG;
    const int wave_id = (blockIdx.x / KF_NSEG) * 4 + wv;
    const int n_waves = (gridDim.x / KF_NSEG) * 4;
    if (tid == 0) sh.n_cand = 0;
    __syncthreads();
    const int n = min(a.n_cells[seg], a.seg_cap);
    const int32_t* list = a.cells + (size_t)seg * a.seg_cap;
    float* win = sh.win[wv];
    float* wx = sh.wx[wv];
    float* wy = sh.wy[wv];
    float* tt = sh.tt[wv];
    float* sm = sh.sm[wv];
    for (int ci = wave_id; ci < n; ci += n_waves) {
        const int e = list[ci];
        const int plane = e >> 14, cy = (e >> 7) & 127, cx = e & 127;
        const float2 sl = a.slack[plane];
        float acc0 = 0.f, acc1 = 0.f;
        MsFetch f;
        ms_fetch(a.s[0], plane, cy, cx, lane, f);
#pragma unroll 1
        for (int m = 0; m < a.n; ++m) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int el = lane + 64 * k;
                win[(el >> 4) * MSF_ROW + (el & 15)] = f.v[k];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int el = lane + 64 * k;
                if (el < 10 * MSF_T) {
                    wx[(el >> 4) * MSF_ROW + (el & 15)] = f.gx[k];
                    wy[(el >> 4) * MSF_ROW + (el & 15)] = f.gy[k];
                }
            }
            const int nq = a.s[m].nq;
            pd_wave_sync();
            // the next scale's values are requested now and arrive while this one is evaluated
            if (m + 1 < a.n) ms_fetch(a.s[m + 1], plane, cy, cx, lane, f);
            if (nq <= 2)
                ms_passes<2>(win, wx, wy, tt, lane, acc0, acc1);
            else if (nq == 3)
                ms_passes<3>(win, wx, wy, tt, lane, acc0, acc1);
            else
                ms_passes<4>(win, wx, wy, tt, lane, acc0, acc1);
            pd_wave_sync();  // the staging areas are rewritten by the next scale
        }
        if (lane < 60) {
            const int jg = lane / 10, i = lane - 10 * jg;
            sm[jg * 10 + i] = acc0;
            if (jg < 4) sm[(jg + 6) * 10 + i] = acc1;
        }
        pd_wave_sync();
        {
            // outputs outside the map have zero weights, so their sums are 0: the reference's zero border
            const int j = 1 + (lane >> 3), i = 1 + (lane & 7);
            const float c = sm[j * 10 + i];
            const float m4 = fmaxf(fmaxf(sm[(j - 1) * 10 + i], sm[(j + 1) * 10 + i]), fmaxf(sm[j * 10 + i - 1], sm[j * 10 + i + 1]));
            const int Y = 8 * cy - 1 + j, X = 8 * cx - 1 + i;
            if (Y < a.H && X < a.W && c > sl.x && c + sl.y >= m4) {
                atomicAdd(a.cand_count + plane, 1);
                const int pos = atomicAdd(&sh.n_cand, 1);
                if (pos < KF_LCAND)
                    sh.cand[pos] = make_int2(plane, (Y << 16) | X);
                else
                    atomicOr(a.cand_count + plane, KF_OVERFLOW);  // no room: leave the plane to the exact kernel
            }
        }
        pd_wave_sync();
    }
    __syncthreads();
    const int nc = min(sh.n_cand, KF_LCAND);
    if (nc == 0) return;
    if (tid == 0) sh.base = atomicAdd(a.cand_n + seg, nc);
    __syncthreads();
    for (int i = tid; i < nc; i += 256) {
        const int idx = sh.base + i;
        if (idx < a.cand_seg_cap)
            a.cand[(size_t)seg * a.cand_seg_cap + idx] = sh.cand[i];
        else
            atomicOr(a.cand_count + sh.cand[i].x, KF_OVERFLOW);
    }
}

// ---------------------------------------------------------------------------------
// Exact verification: one workgroup per candidate.  The 27 x 27 window of the float64 average
// around the candidate is built scale by scale in the reference's arithmetic -- horizontal sums
// of the x8 resize, its vertical sums (the patch of the x8 map the window draws on), the second
// resize's horizontal then vertical sums, the float32 division, the float64 accumulation in
// scale order -- then smoothed with scipy's summation order at the candidate and its four
// neighbours, and the exact test decides.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float ms_share(float v, const MsArgs& a) { return a.pow2 ? v * a.n_rcp : v / a.n_div; }

// floor(e / d) for 0 <= e < 2^16, 1 <= d < 2^8, as a multiplication: m = ceil(2^24 / d).
__device__ __forceinline__ unsigned msv_magic(int d) { return ((1u << 24) + (unsigned)d - 1u) / (unsigned)d; }
__device__ __forceinline__ int msv_div(int e, unsigned m) { return (int)(((unsigned long long)(unsigned)e * m) >> 24); }

__global__ void __launch_bounds__(MSV_NT)
k_ms_verify(MsArgs a) {
    __shared__ double avg[27 * 28];
    __shared__ float patch[MSV_P * (MSV_P + 1)];
    __shared__ float hsum[MSV_NJ * MSV_P];
    __shared__ float h2[MSV_P * 28];
    __shared__ double ypL[3][28];
    __shared__ double s5[8];
    __shared__ float4 s_up[8];
    __shared__ int s_pref[KF_NSEG + 1];
    const int tid = threadIdx.x;
    if (tid >= 64 && tid < 72) {
        const int p = tid - 64;
        s_up[p] = make_float4(a.up8.c[p][0], a.up8.c[p][1], a.up8.c[p][2], a.up8.c[p][3]);
    }
    segment_prefix(a.cand_n, a.cand_seg_cap, s_pref, tid);
    const int total = s_pref[KF_NSEG];
    const int H = a.H, W = a.W;
    for (int ci = blockIdx.x; ci < total; ci += gridDim.x) {
        const int2 ce = a.cand[segment_locate(s_pref, a.cand_seg_cap, ci)];
        const int plane = ce.x, Y = ce.y >> 16, X = ce.y & 0xffff;
        // the reflected rows Y-13 .. Y+13 all lie in [rmin, rmax], likewise the columns
        const int rmin = max(Y - 13, 0), rmax = min(Y + 13, H - 1), nh = rmax - rmin + 1;
        const int cmin = max(X - 13, 0), cmax = min(X + 13, W - 1), nw = cmax - cmin + 1;
        for (int i = tid; i < 27 * 28; i += MSV_NT) avg[i] = 0.0;
        __syncthreads();
#pragma unroll 1
        for (int m = 0; m < a.n; ++m) {
            const MsScale& sc = a.s[m];
            const int hs = sc.hs, ws = sc.ws;
            const float* src = sc.planar + (size_t)plane * hs * ws;
            int pr_lo, ph, pc_lo, pw;
            if (sc.identity) {
                pr_lo = rmin;
                ph = nh;
                pc_lo = cmin;
                pw = nw;
            } else {
                pr_lo = pd_clampi(sc.ty.ofs[rmin], 0, sc.crop_h - 1);
                ph = pd_clampi(sc.ty.ofs[rmax] + 3, 0, sc.crop_h - 1) - pr_lo + 1;
                pc_lo = pd_clampi(sc.tx.ofs[cmin], 0, sc.crop_w - 1);
                pw = pd_clampi(sc.tx.ofs[cmax] + 3, 0, sc.crop_w - 1) - pc_lo + 1;
            }
            const int j_lo = pd_up8_ofs(pr_lo);
            const int nj = pd_up8_ofs(pr_lo + ph - 1) + 3 - j_lo + 1;
            const unsigned m_pw = msv_magic(pw), m_nw = msv_magic(nw);
            // x8 resize, horizontal: source row j_lo + jj (clamped), patch column c
            for (int e = tid; e < nj * pw; e += MSV_NT) {
                const int jj = msv_div(e, m_pw), c = e - jj * pw;
                const int col = pc_lo + c, x0 = pd_up8_ofs(col);
                const float4 w = s_up[pd_up8_phase(col)];
                const float* row = src + (size_t)pd_clampi(j_lo + jj, 0, hs - 1) * ws;
                hsum[jj * MSV_P + c] = ((row[pd_clampi(x0, 0, ws - 1)] * w.x + row[pd_clampi(x0 + 1, 0, ws - 1)] * w.y) +
                                        row[pd_clampi(x0 + 2, 0, ws - 1)] * w.z) +
                                       row[pd_clampi(x0 + 3, 0, ws - 1)] * w.w;
            }
            __syncthreads();
            // x8 resize, vertical: the patch
            for (int e = tid; e < ph * pw; e += MSV_NT) {
                const int r = msv_div(e, m_pw), c = e - r * pw;
                const int rabs = pr_lo + r;
                const float* hcol = hsum + (pd_up8_ofs(rabs) - j_lo) * MSV_P + c;
                const float4 w = s_up[pd_up8_phase(rabs)];
                patch[r * (MSV_P + 1) + c] = ((hcol[0] * w.x + hcol[MSV_P] * w.y) + hcol[2 * MSV_P] * w.z) + hcol[3 * MSV_P] * w.w;
            }
            __syncthreads();
            if (sc.identity) {
                for (int e = tid; e < nh * nw; e += MSV_NT) {
                    const int yo = msv_div(e, m_nw), xo = e - yo * nw;
                    avg[yo * 28 + xo] = avg[yo * 28 + xo] + (double)ms_share(patch[yo * (MSV_P + 1) + xo], a);
                }
            } else {
                // second resize, horizontal: patch row r, window column xo
                for (int e = tid; e < ph * nw; e += MSV_NT) {
                    const int r = msv_div(e, m_nw), xo = e - r * nw;
                    const int Xa = cmin + xo, x0 = sc.tx.ofs[Xa];
                    const float4 ac = *(const float4*)(sc.tx.coef + 4 * (size_t)Xa);
                    const float* prow = patch + r * (MSV_P + 1) - pc_lo;
                    h2[r * 28 + xo] = ((prow[pd_clampi(x0, 0, sc.crop_w - 1)] * ac.x + prow[pd_clampi(x0 + 1, 0, sc.crop_w - 1)] * ac.y) +
                                       prow[pd_clampi(x0 + 2, 0, sc.crop_w - 1)] * ac.z) +
                                      prow[pd_clampi(x0 + 3, 0, sc.crop_w - 1)] * ac.w;
                }
                __syncthreads();
                // second resize, vertical; float32 share of the average; float64 sum
                for (int e = tid; e < nh * nw; e += MSV_NT) {
                    const int yo = msv_div(e, m_nw), xo = e - yo * nw;
                    const int Ya = rmin + yo, y0 = sc.ty.ofs[Ya];
                    const float4 bc = *(const float4*)(sc.ty.coef + 4 * (size_t)Ya);
                    const float* hc2 = h2 + xo - pr_lo * 28;
                    const float v = ((hc2[pd_clampi(y0, 0, sc.crop_h - 1) * 28] * bc.x + hc2[pd_clampi(y0 + 1, 0, sc.crop_h - 1) * 28] * bc.y) +
                                     hc2[pd_clampi(y0 + 2, 0, sc.crop_h - 1) * 28] * bc.z) +
                                    hc2[pd_clampi(y0 + 3, 0, sc.crop_h - 1) * 28] * bc.w;
                    avg[yo * 28 + xo] = avg[yo * 28 + xo] + (double)ms_share(v, a);
                }
            }
            __syncthreads();
        }
        // Gaussian, y pass at rows Y-1, Y, Y+1 of the 27 window columns (scipy's order)
        if (tid < 81) {
            const int e = tid / 27, l = tid - 27 * e;
            const int cc = pd_reflect(X + l - 13, W) - cmin;
            const int c = 12 + e;
            double t = avg[(pd_reflect(Y - 13 + c, H) - rmin) * 28 + cc] * c_gw[12];
#pragma unroll
            for (int d = 12; d >= 1; --d)
                t += (avg[(pd_reflect(Y - 13 + c - d, H) - rmin) * 28 + cc] + avg[(pd_reflect(Y - 13 + c + d, H) - rmin) * 28 + cc]) * c_gw[12 - d];
            ypL[e][l] = t;
        }
        __syncthreads();
        if (tid < 5) {
            // 0 centre, 1 left, 2 right, 3 up, 4 down
            const int l = tid;
            const int e = l == 3 ? 0 : (l == 4 ? 2 : 1);
            const int xo = l == 1 ? -1 : (l == 2 ? 1 : 0);
            const bool inside = (l == 1) ? X - 1 >= 0 : (l == 2) ? X + 1 < W : (l == 3) ? Y - 1 >= 0 : (l == 4) ? Y + 1 < H : true;
            double S = 0.0;
            if (inside) {
                const double* p = &ypL[e][13 + xo];
                S = p[0] * c_gw[12];
#pragma unroll
                for (int d = 12; d >= 1; --d) S += (p[-d] + p[d]) * c_gw[12 - d];
            }
            s5[l] = S;
        }
        __syncthreads();
        if (tid == 0 && a.cand_count[plane] <= a.CK) {  // overflowing planes are redone by the exact kernel
            const double s0 = s5[0];
            if (s0 >= s5[1] && s0 >= s5[2] && s0 >= s5[3] && s0 >= s5[4] && s0 > a.thr) {
                const int pos = atomicAdd(a.raw_count + plane, 1);
                if (pos < a.PK) {
                    RawPeak p;
                    p.y = Y;
                    p.x = X;
                    p.score = avg[(Y - rmin) * 28 + (X - cmin)];
                    a.raw[(size_t)plane * a.PK + pos] = p;
                }
            }
        }
        __syncthreads();
    }
}

int pd_launch_peaks_filtered_multi(pd_ctx* ctx, double thr) {
    static const bool off = [] {
        const char* v = getenv("PD_MS_STORED_HEAT");
        return v && *v && *v != '0';
    }();  // comparison runs: the averaged heatmaps written out, the full-resolution filter on them
    if (off) return PD_E_STATE;
    int rc = build_ms_tables(ctx);
    if (rc != PD_OK) return rc;
    const int B = ctx->B, H = ctx->H, W = ctx->W, n = ctx->ms_n;
    const int hc = (H + 7) / 8, wc = (W + 7) / 8;
    int CK = 0;
    rc = pd_ensure_candidates(ctx, &CK);
    if (rc != PD_OK) return rc;
    const size_t planes = (size_t)B * PD_N_PARTS;
    size_t planar_floats = 0;
    for (int m = 0; m < n; ++m) planar_floats += planes * ctx->ms_hs[m] * ctx->ms_ws[m];
    rc = pd_grow(ctx, (void**)&ctx->ms_planar, &ctx->ms_planar_cap, planar_floats * sizeof(float));
    if (rc != PD_OK) return rc;
    const size_t seg_cap = ((planes + KF_NSEG - 1) / KF_NSEG) * hc * wc;
    rc = pd_grow(ctx, (void**)&ctx->cells, &ctx->cells_cap, sizeof(int32_t) * seg_cap * KF_NSEG);
    if (rc != PD_OK) return rc;
    int32_t* counters = ctx->cand_count + planes;  // candidates | active cells | (unused), KF_NSEG each

    const MsLayout lay = ms_layout(n, hc, wc);
    MsArgs a;
    memset(&a, 0, sizeof(a));
    MsPlanarArgs pa;
    memset(&pa, 0, sizeof(pa));
    size_t off_f = 0;
    int max_pix = 0;
    for (int m = 0; m < n; ++m) {
        MsScale& sc = a.s[m];
        sc.planar = ctx->ms_planar + off_f;
        sc.hs = ctx->ms_hs[m];
        sc.ws = ctx->ms_ws[m];
        sc.crop_h = ctx->ms_crop_h[m];
        sc.crop_w = ctx->ms_crop_w[m];
        sc.identity = sc.crop_h == H && sc.crop_w == W;
        sc.nq = ctx->msf_nq[m];
        sc.y.cw = ctx->msf_tab + lay.cw[m][0];
        sc.y.bw = ctx->msf_tab + lay.bw[m][0];
        sc.y.lo = ctx->msf_lo + lay.lo[m][0];
        sc.x.cw = ctx->msf_tab + lay.cw[m][1];
        sc.x.bw = ctx->msf_tab + lay.bw[m][1];
        sc.x.lo = ctx->msf_lo + lay.lo[m][1];
        if (!sc.identity) {
            sc.ty = ctx->tabs[2 + m].ty;
            sc.tx = ctx->tabs[2 + m].tx;
        }
        pa.src[m] = ctx->ms_heat[m];
        pa.dst[m] = ctx->ms_planar + off_f;
        pa.npix[m] = sc.hs * sc.ws;
        max_pix = std::max(max_pix, pa.npix[m]);
        off_f += planes * sc.hs * sc.ws;
    }
    a.n = n;
    a.pow2 = (n & (n - 1)) == 0;
    a.n_div = (float)n;
    a.n_rcp = a.pow2 ? 1.0f / (float)n : 0.f;
    a.n_planes = (int)planes;
    a.H = H;
    a.W = W;
    a.hc = hc;
    a.wc = wc;
    a.lds_a = ctx->msf_lds_a;
    a.lds_r = ctx->msf_lds_r;
    a.thr = thr;
    a.slack = ctx->plane_slack;
    a.cells = ctx->cells;
    a.seg_cap = (int)seg_cap;
    a.n_cells = counters + KF_NSEG;
    a.cand = (int2*)ctx->cand;
    a.cand_seg_cap = (int)(planes * CK / KF_NSEG);
    a.cand_count = ctx->cand_count;
    a.cand_n = counters;
    a.CK = CK;
    a.raw = ctx->raw;
    a.raw_count = ctx->raw_count;
    a.PK = ctx->cfg.max_peaks_per_part;
    pd_up8_fill(&a.up8);

    pd_prof_begin(ctx, PD_STAGE_PLANAR);
    hipLaunchKernelGGL(k_ms_planar, dim3((max_pix + MSP_PIX - 1) / MSP_PIX, B, n), dim3(256), 0, ctx->stream, pa);
    pd_prof_end(ctx);
    {
        // workgroups whose staging areas exceed 64 KB (maps of COCO size): the limit is raised once per kernel
        const size_t need = hc <= 64 && wc <= 64 ? ((size_t)a.lds_a + a.lds_r) * sizeof(float)
                                                 : ((size_t)hc * wc + a.lds_a + a.lds_r) * sizeof(float);
        static size_t allowed[2][64];  // per kernel and device; 0 = the default of 64 KB
        const int dev = ctx->cfg.device_id >= 0 && ctx->cfg.device_id < 64 ? ctx->cfg.device_id : 0;
        size_t& have = allowed[hc <= 64 && wc <= 64 ? 0 : 1][dev];
        if (need > 64 * 1024 && need > have) {
            const void* fn = hc <= 64 && wc <= 64 ? reinterpret_cast<const void*>(k_ms_cells64) : reinterpret_cast<const void*>(k_ms_cells);
            PD_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            have = 150 * 1024;
        }
    }
    pd_prof_begin(ctx, PD_STAGE_CELLS);
    if (hc <= 64 && wc <= 64)
        hipLaunchKernelGGL(k_ms_cells64, dim3((unsigned)((planes + 7) / 8 * 8)), dim3(256), ((size_t)a.lds_a + a.lds_r) * sizeof(float),
                           ctx->stream, a);
    else
        hipLaunchKernelGGL(k_ms_cells, dim3((unsigned)((planes + 7) / 8 * 8)), dim3(256),
                           ((size_t)hc * wc + a.lds_a + a.lds_r) * sizeof(float), ctx->stream, a);
    pd_prof_end(ctx);
    // the list lengths are only known on the device: a fixed grid strides over them
    size_t fblocks = (planes * hc * wc / 16 + 3) / 4;
    if (fblocks > 2048) fblocks = 2048;
    fblocks = (fblocks + KF_NSEG - 1) / KF_NSEG * KF_NSEG;
    pd_prof_begin(ctx, PD_STAGE_FILTER);
    hipLaunchKernelGGL(k_ms_cell_filter, dim3((unsigned)fblocks), dim3(256), 0, ctx->stream, a);
    pd_prof_end(ctx);
    int vblocks = B * 48;
    if (vblocks > 4096) vblocks = 4096;
    pd_prof_begin(ctx, PD_STAGE_VERIFY);
    hipLaunchKernelGGL(k_ms_verify, dim3(vblocks), dim3(MSV_NT), 0, ctx->stream, a);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    // planes whose candidates did not fit are redone by the exact kernel (which evaluates the
    // average pixel by pixel); the same kernel then orders every image's peaks
    return pd_launch_stream_exact(ctx, thr, 0, nullptr, ctx->cand_count, CK, 1);
}
