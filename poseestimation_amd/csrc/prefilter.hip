// Fast peak stage for network-resolution sources: a float32 candidate filter over the
// active part of the map followed by exact float64 verification of the few candidates.
// Results are identical to the exact kernel over every pixel (get_peaks,
// utils/postprocess.py:26-75).
//
// Filter.  The smoothed map is a linear function of the (h, w) network output:
// Gaussian (25 taps, reflect) after bicubic x8 (4 taps, clamped) composes, per axis,
// into a banded operator with 7 taps per output,
//     S~[Y][X] = sum_s sum_t Ky[Y][s] * Kx[X][t] * src[s][t].
// In units of 8 x 8 pixel cells (cell c of an axis = outputs 8c .. 8c+7, plus one halo
// output either side) output 8c-1+j draws on sources c-4+p0(j) .. c+2+p0(j), with
// p0 = 0 for j = 0, 1 for j = 1..8, 2 for j = 9; border effects (clamped taps,
// reflected rows) only change the weights, not the windows, and all cells away from
// the border share one set of weights.  The filter evaluates S~ in float32 with fused
// multiply-adds.  Against the exact float64 value S of the reference (which is computed
// from the float32-rounded bicubic map),
//     |S~ - S| <= 53 * 2^-24 * 1.89 * M  <  eps := 2^-17 * M,   M >= max |src| of the plane
// (coefficient rounding, 7-term chains on both axes, and the rounding inside the
// reference's own float32 bicubic; sum |K| <= 1.375 per axis).  A pixel can only be
// a peak (S > thr, S >= each in-image neighbour, S >= 0 for out-of-image ones) if
//     S~ > thr - eps  and  S~ >= S~(neighbour) - 2 eps,
// so every true peak is among the candidates.
//
// Active cells.  |S~| over a cell is at most
//     U = sum_s sum_t By[s] * Bx[t] * |src[s][t]|,   B = the largest |weight| any of the
// cell's eight own outputs puts on that source,
// a 7 x 7 separable sum on the network-resolution map.  Cells with U below the threshold
// cannot hold a candidate and are never touched at full resolution; typical maps leave a
// few percent of the cells.
//
// Verification.  Per candidate the exact S at the pixel and its four neighbours is
// recomputed with the reference's arithmetic (27 x 27 bicubic values, 77 y-pass sums,
// 5 x-pass sums) and the exact test decides.  Planes with more candidates than the list
// holds (plateaus) are redone by the exact streaming kernel.
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "colloader.h"
#include "prefilter.h"

#pragma clang fp contract(off)


static const double h_gw[PD_GTAPS] = PD_GAUSS_WEIGHTS;

// The weights of a cell away from the border are the same on both axes and for every map
// size.  They live in constant memory so that they reach the multiply-adds as scalar
// operands; the kernels re-derive the pointer inside the cell loop (fresh_pointer) so that
// the 70 scalar loads stay next to their uses instead of being hoisted and spilled.
__constant__ float c_kf_interior[72];
typedef __attribute__((address_space(4))) float ConstF;
__device__ __forceinline__ const ConstF* fresh_pointer(const float* p) {
    const ConstF* q = (const ConstF*)p;
    asm volatile("" : "+s"(q));
    return q;
}

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

// One axis in cell form: cw[c][j * 7 + p] = weight of source c-4+p0(j)+p in output
// 8c-1+j (zero for outputs outside the map), bw[c][p] = max over j = 1..8 of
// |cw[c][j * 7 + p]| (sources c-3 .. c+3).  False if some weight falls outside its window.
static bool cell_weights(int L, int l, const std::vector<double>& K, std::vector<float>& cw, std::vector<float>& bw,
                         std::vector<int32_t>& reg, const float* interior) {
    const int nc = (L + 7) / 8;
    cw.assign((size_t)nc * 72, 0.f);
    bw.assign((size_t)nc * 8, 0.f);
    for (int c = 0; c < nc; ++c)
        for (int j = 0; j < 10; ++j) {
            const int o = 8 * c - 1 + j;
            if (o < 0 || o >= L) continue;
            const int p0 = j == 0 ? 0 : (j == 9 ? 2 : 1);
            const int t_lo = c - 4 + p0;
            for (int s = 0; s < l; ++s) {
                const double v = K[(size_t)o * l + s];
                if (v == 0.0) continue;
                if (s < t_lo || s > t_lo + 6) return false;
                const float fv = (float)v;
                cw[(size_t)c * 72 + j * 7 + (s - t_lo)] = fv;
                if (j >= 1 && j <= 8) {
                    float* b = &bw[(size_t)c * 8 + (s - t_lo)];
                    if (fabsf(fv) > *b) *b = fabsf(fv);
                }
            }
        }
    reg.assign(nc, 0);
    if (interior)
        for (int c = 0; c < nc; ++c)
            reg[c] = c - 4 >= 0 && c + 4 <= l - 1 && memcmp(&cw[(size_t)c * 72], interior, sizeof(float) * 70) == 0;
    return true;
}

// The interior weights, from a map large enough to have cells away from every border;
// uploaded to constant memory once per device.
static int interior_weights(pd_ctx* ctx, const float** out) {
    static std::mutex mu;
    static float wts[72];
    static bool have = false;
    static bool uploaded[64];
    std::lock_guard<std::mutex> lock(mu);
    if (!have) {
        const int l = 32, L = 256;
        std::vector<int32_t> ofs(L), reg;
        std::vector<float> coef(4 * (size_t)L), cw, bw;
        std::vector<double> K;
        pd_host_cubic_table(l, L, 0.125, ofs.data(), coef.data());
        composite(L, l, ofs.data(), coef.data(), K);
        if (!cell_weights(L, l, K, cw, bw, reg, nullptr)) return PD_E_STATE;
        memset(wts, 0, sizeof(wts));
        memcpy(wts, &cw[(size_t)16 * 72], sizeof(float) * 70);
        if (memcmp(&cw[(size_t)15 * 72], wts, sizeof(float) * 70) != 0) return PD_E_STATE;
        have = true;
    }
    const int dev = ctx->cfg.device_id;
    if (dev < 0 || dev >= 64) return PD_E_STATE;
    if (!uploaded[dev]) {
        PD_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_kf_interior), wts, sizeof(wts)));
        uploaded[dev] = true;
    }
    *out = wts;
    return PD_OK;
}

static int build_filter_tables(pd_ctx* ctx) {
    const int h = ctx->h, w = ctx->w, H = ctx->H, W = ctx->W;
    if (ctx->kf_h == h && ctx->kf_w == w && ctx->kf_H == H && ctx->kf_W == W) return ctx->kf_ok ? PD_OK : PD_E_STATE;
    ctx->kf_h = h;
    ctx->kf_w = w;
    ctx->kf_H = H;
    ctx->kf_W = W;
    ctx->kf_ok = 0;
    ctx->gen++;  // the device tables are rewritten (and may move) below
    const int hc = (H + 7) / 8, wc = (W + 7) / 8;
    // cell coordinates are packed in 7 bits each; the active-cell kernel keeps one
    // plane (h x w) and its row sums (h x wc) in LDS
    if (h < 8 || w < 8 || hc > 128 || wc > 128 || (size_t)h * w + (size_t)h * wc > 15000) return PD_E_STATE;
    std::vector<int32_t> yofs(H), xofs(W);
    std::vector<float> yc(4 * (size_t)H), xc(4 * (size_t)W);
    pd_host_cubic_table(h, H, 0.125, yofs.data(), yc.data());
    pd_host_cubic_table(w, W, 0.125, xofs.data(), xc.data());
    std::vector<double> Ky, Kx;
    composite(H, h, yofs.data(), yc.data(), Ky);
    composite(W, w, xofs.data(), xc.data(), Kx);
    std::vector<float> cy, cx, by, bx;
    std::vector<int32_t> ry, rx;
    const float* interior = nullptr;
    int rc = interior_weights(ctx, &interior);
    if (rc != PD_OK) return rc;
    if (!cell_weights(H, h, Ky, cy, by, ry, interior) || !cell_weights(W, w, Kx, cx, bx, rx, interior)) return PD_E_STATE;
    // device layout: [hc][72] | [wc][72] | [hc][8] | [wc][8] floats, [hc] | [wc] flags
    std::vector<float> tab;
    tab.insert(tab.end(), cy.begin(), cy.end());
    tab.insert(tab.end(), cx.begin(), cx.end());
    tab.insert(tab.end(), by.begin(), by.end());
    tab.insert(tab.end(), bx.begin(), bx.end());
    std::vector<int32_t> reg(ry);
    reg.insert(reg.end(), rx.begin(), rx.end());
    PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->kf_tab_cap < tab.size()) {
        if (ctx->kf_tab) PD_HIP(ctx, hipFree(ctx->kf_tab));
        ctx->kf_tab = nullptr;
        ctx->kf_tab_cap = 0;
        PD_HIP(ctx, hipMalloc(&ctx->kf_tab, sizeof(float) * tab.size()));
        ctx->kf_tab_cap = tab.size();
    }
    if (ctx->kf_reg_cap < reg.size()) {
        if (ctx->kf_reg) PD_HIP(ctx, hipFree(ctx->kf_reg));
        ctx->kf_reg = nullptr;
        ctx->kf_reg_cap = 0;
        PD_HIP(ctx, hipMalloc(&ctx->kf_reg, sizeof(int32_t) * reg.size()));
        ctx->kf_reg_cap = reg.size();
    }
    PD_HIP(ctx, hipMemcpy(ctx->kf_tab, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice));
    PD_HIP(ctx, hipMemcpy(ctx->kf_reg, reg.data(), sizeof(int32_t) * reg.size(), hipMemcpyHostToDevice));
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
    __shared__ __align__(16) float tile[KP_PIX * PD_CH_HEAT];
    __shared__ float s_red[4];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int p0 = blockIdx.x * KP_PIX;
    const int np = min(KP_PIX, npix - p0);
    const float* src = lo + ((size_t)b * npix + p0) * PD_CH_HEAT;
    float m = 0.f;
    const int n = np * PD_CH_HEAT;
    int done = 0;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {  // 16 bytes per load: four times the bytes in flight per thread
        const int n4 = n >> 2;
        for (int i = tid; i < n4; i += 256) {
            const float4 v = ((const float4*)src)[i];
            ((float4*)tile)[i] = v;
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        done = 4 * n4;
    }
    for (int i = done + tid; i < n; i += 256) {
        const float v = src[i];
        tile[i] = v;
        m = fmaxf(m, fabsf(v));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((tid & 63) == 0) s_red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0 && absmax_bits)
        atomicMax(absmax_bits + b, __float_as_uint(fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]))));
    float* dst = planar + (size_t)b * PD_CH_HEAT * npix + p0;
    for (int c = 0; c < PD_CH_HEAT; ++c)
        for (int i = tid; i < np; i += 256) dst[(size_t)c * npix + i] = tile[i * PD_CH_HEAT + c];
}


// ---------------------------------------------------------------------------------
// Active cells: one workgroup per (part, image) plane.
// ---------------------------------------------------------------------------------
struct CellArgs {
    const float* lo;    // planar copy (B, 19, h, w): read, or -- with nhwc set -- written by this kernel
    const float* nhwc;  // the network output (B, h, w, 19) itself, or NULL
    float* planar_out;
    int n_planes;
    int h, w, hc, wc;
    const float* by;  // [hc][8] bound weights, sources y-3 .. y+3
    const float* bx;  // [wc][8]
    const int32_t* regy;
    const int32_t* regx;
    float2* slack;        // [B * 18] out: (thr_lo, e2) of the candidate test of a plane
    double thr;
    int32_t* cells;       // KF_NSEG segments; interior-weight cells from the front of a segment, the others from its back
    int seg_cap;
    int32_t* n_reg;       // [KF_NSEG]
    int32_t* n_edge;      // [KF_NSEG]
    int32_t* cand_count;  // [B * 18]: KF_OVERFLOW for a plane with non-finite values (the exact kernel takes it)
};

// One workgroup per (part, image) plane.  Once the plane is in LDS every thread owns a cell
// column and a band of cell rows: it walks down the band with the seven row sums of the y
// window in registers and notes its active cells as bit masks, so that the only workgroup
// barriers are the one behind the load and the two around the reservation of list space.
// 256 threads for small planes, 512 for large ones.
__global__ void __launch_bounds__(512)
k_cells(CellArgs a) {
    extern __shared__ __align__(16) float lds[];
    __shared__ float s_red[8];
    __shared__ int s_cnt[8][2];
    __shared__ int s_base[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n_waves = blockDim.x >> 6;
    // the 18 planes of an image on one XCD (workgroup i runs on XCD i % 8): they read the same
    // cache lines of the network output
    const int per_xcd = gridDim.x / 8;
    const int plane = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (plane >= a.n_planes) return;
    const int b = plane / PD_N_PARTS, part = plane - b * PD_N_PARTS;
    const int h = a.h, w = a.w;
    const int hc = a.hc, wc = a.wc;  // cells of 8 x 8 output pixels: hc <= h, wc <= w
    // thread = (cell column x, band of cell rows): as many bands as the workgroup has whole rows of threads
    const int bands = blockDim.x / wc;
    const int rows_per = (hc + bands - 1) / bands;  // <= 32: the masks below hold a bit per row
    const int x = tid % wc, band = tid / wc;
    const int ya = min(hc, band * rows_per), yb = band < bands ? min(hc, ya + rows_per) : ya;
    float* A = lds;  // [h][w] |v| of the plane
    // ---- the plane: |v| to LDS, its maximum, its planar copy (small maps: the plane is picked
    // out of the (h, w, 19) pixels, one value per 76 bytes; the lines are shared with the image's
    // other planes through the L2).  All loads of a round are issued before the first value is used.
    const float* img = a.lo + ((size_t)b * PD_CH_HEAT + part) * h * w;
    const float* pix = a.nhwc ? a.nhwc + (size_t)b * h * w * PD_CH_HEAT + part : nullptr;
    float* copy = a.nhwc ? a.planar_out + ((size_t)b * PD_CH_HEAT + part) * h * w : nullptr;
    const int NT = blockDim.x;
    float m = 0.f;
    // (nine per thread and round: a 46 x 46 plane is one round of 256 threads)
    for (int i0 = 0; i0 < h * w; i0 += NT * 9) {
        float t[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int i = i0 + k * NT + tid;
            t[k] = i < h * w ? (pix ? pix[(size_t)i * PD_CH_HEAT] : img[i]) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int i = i0 + k * NT + tid;
            if (i < h * w) {
                A[i] = fabsf(t[k]);
                m = absmax_step(m, t[k]);
                if (copy) copy[i] = t[k];
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if (lane == 0) s_red[wv] = m;
    __syncthreads();
    // the smoothed map of a part is a function of its own plane only: the plane's maximum bounds
    // the rounding of the float32 evaluation.  A few dozen float64 instructions, one lane per
    // wavefront does them.
    float thr_lo = 0.f;
    int finite = 1;
    if (lane == 0) {
        float pm = 0.f, e2;
        for (int q = 0; q < n_waves; ++q) pm = fmaxf(pm, s_red[q]);
        filter_slack(a.thr, pm, &thr_lo, &e2);
        finite = absmax_finite(pm);
        if (wv == 0) {
            a.slack[plane] = make_float2(thr_lo, e2);
            if (!finite) atomicOr(a.cand_count + plane, KF_OVERFLOW);
        }
    }
    thr_lo = __shfl(thr_lo, 0);
    finite = __shfl(finite, 0);
    // ---- the bound U of every cell of the thread's band
    unsigned m_reg = 0u, m_edge = 0u;
    if (yb > ya && finite) {
        float kx[7];
        int cx[7];
#pragma unroll
        for (int p = 0; p < 7; ++p) {
            kx[p] = a.bx[8 * x + p];
            cx[p] = pd_clampi(x - 3 + p, 0, w - 1);
        }
        const int regx = a.regx[x];
        float win[7];
        win[0] = 0.f;
        // the window of cell row y is the row sums of map rows y-3 .. y+3, clamped to the map
#pragma unroll
        for (int q = 1; q < 7; ++q) {
            const float* row = A + pd_clampi(ya - 4 + q, 0, h - 1) * w;
            float v = kx[0] * row[cx[0]];
#pragma unroll
            for (int p = 1; p < 7; ++p) v = fmaf(kx[p], row[cx[p]], v);
            win[q] = v;
        }
        for (int y = ya; y < yb; ++y) {
#pragma unroll
            for (int p = 0; p < 6; ++p) win[p] = win[p + 1];
            {
                const float* row = A + min(y + 3, h - 1) * w;
                float v = kx[0] * row[cx[0]];
#pragma unroll
                for (int p = 1; p < 7; ++p) v = fmaf(kx[p], row[cx[p]], v);
                win[6] = v;
            }
            const float4 ka = *(const float4*)(a.by + 8 * y), kb = *(const float4*)(a.by + 8 * y + 4);
            float u = ka.x * win[0];
            u = fmaf(ka.y, win[1], u);
            u = fmaf(ka.z, win[2], u);
            u = fmaf(ka.w, win[3], u);
            u = fmaf(kb.x, win[4], u);
            u = fmaf(kb.y, win[5], u);
            u = fmaf(kb.z, win[6], u);
            if (u * 1.00001f >= thr_lo) {  // the factor covers the rounding of the chains on both sides
                if (a.regy[y] & regx)
                    m_reg |= 1u << (y - ya);
                else
                    m_edge |= 1u << (y - ya);
            }
        }
    }
    // ---- list space: counts per wavefront, one reservation per workgroup and kind
    const int c_r = __builtin_popcount(m_reg), c_e = __builtin_popcount(m_edge);
    int p_r = c_r, p_e = c_e;  // inclusive prefix over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o_r = __shfl_up(p_r, d), o_e = __shfl_up(p_e, d);
        if (lane >= d) {
            p_r += o_r;
            p_e += o_e;
        }
    }
    if (lane == 63) {
        s_cnt[wv][0] = p_r;
        s_cnt[wv][1] = p_e;
    }
    __syncthreads();
    const int seg = plane % KF_NSEG;
    if (tid == 0) {
        int nr = 0, ne = 0;
        for (int q = 0; q < n_waves; ++q) {
            nr += s_cnt[q][0];
            ne += s_cnt[q][1];
        }
        s_base[0] = nr ? atomicAdd(a.n_reg + seg, nr) : 0;
        s_base[1] = ne ? atomicAdd(a.n_edge + seg, ne) : 0;
    }
    __syncthreads();
    int at_r = s_base[0] + p_r - c_r, at_e = s_base[1] + p_e - c_e;
    for (int q = 0; q < wv; ++q) {
        at_r += s_cnt[q][0];
        at_e += s_cnt[q][1];
    }
    // every cell is listed at most once and a segment has room for all cells of its planes:
    // interior-weight cells from the front of the segment, the others from its back
    int32_t* segment = a.cells + (size_t)seg * a.seg_cap;
    while (m_reg) {
        const int bit = __builtin_ctz(m_reg);
        m_reg &= m_reg - 1u;
        segment[at_r++] = (plane << 14) | ((ya + bit) << 7) | x;
    }
    while (m_edge) {
        const int bit = __builtin_ctz(m_edge);
        m_edge &= m_edge - 1u;
        segment[a.seg_cap - 1 - (at_e++)] = (plane << 14) | ((ya + bit) << 7) | x;
    }
}

// ---------------------------------------------------------------------------------
// Candidate filter over the listed cells.
// ---------------------------------------------------------------------------------
struct CellFilterArgs {
    const float* lo;  // planar copy (B, 19, h, w)
    int h, w, H, W;
    const float* cy;  // [hc][72]
    const float* cx;  // [wc][72]
    const float2* slack;   // [B * 18] (thr_lo, e2) per plane
    const int32_t* cells;
    int seg_cap;
    const int32_t* n_reg;
    const int32_t* n_edge;
    int2* cand;            // (plane, y << 16 | x), KF_NSEG segments
    int cand_seg_cap;
    int32_t* cand_count;   // [B][18] candidates seen per plane (may exceed CK)
    int32_t* cand_n;       // [KF_NSEG] entries asked for per segment (may exceed cand_seg_cap)
    int CK;
    int seg_by_xcd;        // k_cells_filter: the candidates of a plane go to one of the four segments of the XCD that staged it
};


template <int LCAND, int WSLOTS>
struct FilterSharedT {
    float tile[KF_CELLS_PER_BLOCK][9][10];  // per cell: x-pass sums, source row q, output column i
    __attribute__((aligned(16))) float wts[WSLOTS][72];  // per border cell in flight: the weights of the pass at hand
    int2 cand[LCAND];
    int n_cand;
    int base;
};
typedef FilterSharedT<KF_LCAND, KF_CELLS_PER_BLOCK> FilterShared;
// The workgroup of one small plane: a handful of candidates.
#ifndef KF_LCAND_PLANE
#define KF_LCAND_PLANE 128
#endif
#ifndef KF_PLANE_WSLOTS
#define KF_PLANE_WSLOTS KF_CELLS_PER_BLOCK
#endif
typedef FilterSharedT<KF_LCAND_PLANE, KF_PLANE_WSLOTS> PlaneFilterShared;

// One cell of the list, decoded, with the nine source values of this lane's row.
struct CellWork {
    int plane, y, x;  // plane < 0: no cell
    float thr_lo, e2;
    float v[9];
};

__device__ __forceinline__ int load_entry(const int32_t* list, int n, int ci, bool lane_ok, bool reg, int seg_cap) {
    if (!lane_ok || ci >= n) return -1;
    return list[reg ? ci : seg_cap - 1 - ci];
}

// Issues the loads of a cell: the (thr_lo, e2) pair of its image and, for lanes r < 9,
// source row y-4+r at columns x-4 .. x+4 (clamped; taps outside the map have zero weight).
template <bool REG>
__device__ __forceinline__ void load_cell(const CellFilterArgs& a, int e, int r, CellWork& c) {
    c.plane = -1;
    c.y = c.x = 0;
    c.thr_lo = c.e2 = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) c.v[k] = 0.f;
    if (e < 0) return;
    c.plane = e >> 14;
    c.y = (e >> 7) & 127;
    c.x = e & 127;
    const int b = c.plane / PD_N_PARTS, part = c.plane - b * PD_N_PARTS;
    const float2 sl = a.slack[c.plane];
    c.thr_lo = sl.x;
    c.e2 = sl.y;
    if (r < 9) {
        if (REG) {  // the 9 x 9 source window of a cell with interior weights lies inside the map
            const float* p = a.lo + (((size_t)b * PD_CH_HEAT + part) * a.h + (c.y - 4 + r)) * a.w + (c.x - 4);
#pragma unroll
            for (int k = 0; k < 9; ++k) c.v[k] = p[k];
        } else {
            const float* row = a.lo + (((size_t)b * PD_CH_HEAT + part) * a.h + pd_clampi(c.y - 4 + r, 0, a.h - 1)) * a.w;
#pragma unroll
            for (int k = 0; k < 9; ++k) c.v[k] = row[pd_clampi(c.x - 4 + k, 0, a.w - 1)];
        }
    }
}

// One pass over one kind of cell of segment `seg`: every wavefront takes six cells at a
// time (ten lanes per cell) and runs on its own: the x-pass results cross lanes through a
// wave-private LDS tile, and the loads of the next six cells are in flight while the
// current ones are evaluated.
// REG: the cells carry the interior weights, which reach the multiply-adds as scalar
// operands from constant memory; otherwise the ten lanes of a cell stage its column
// weights, then its row weights, in the wave-private LDS array `wl`.
template <bool REG>
__device__ __forceinline__ void filter_cells(const CellFilterArgs& a, FilterShared& sh, int seg, int wave_id,
                                             int n_waves, int lane) {
    const int wv = threadIdx.x >> 6;
    const int cs = lane / 10, r = lane - cs * 10;
    const bool lane_ok = cs < KF_CELLS_PER_WAVE;
    const int slot = wv * KF_CELLS_PER_WAVE + (lane_ok ? cs : 0);
    const int H = a.H, W = a.W;
    const int n = min((REG ? a.n_reg : a.n_edge)[seg], a.seg_cap);
    const int32_t* list = a.cells + (size_t)seg * a.seg_cap;
    float* wl = &sh.wts[slot][0];

    const int stride = n_waves * KF_CELLS_PER_WAVE;
    int base = wave_id * KF_CELLS_PER_WAVE;  // first cell of this wavefront's current six
    if (base >= n) return;
    CellWork cur, nxt;
    load_cell<REG>(a, load_entry(list, n, base + cs, lane_ok, REG, a.seg_cap), r, cur);
    int e_next = load_entry(list, n, base + stride + cs, lane_ok, REG, a.seg_cap);
    while (base < n) {
        if (REG) {  // the few border cells go without the look-ahead
            load_cell<REG>(a, e_next, r, nxt);
            e_next = load_entry(list, n, base + 2 * stride + cs, lane_ok, REG, a.seg_cap);
        }
        const bool act = cur.plane >= 0;
        if (!REG) {
            if (act && r < 9) {
                const float4* kq = (const float4*)(a.cx + (size_t)cur.x * 72);
                ((float4*)wl)[2 * r] = kq[2 * r];
                ((float4*)wl)[2 * r + 1] = kq[2 * r + 1];
            }
            pd_wave_sync();
        }
        // x pass: lane r < 9 owns source row y-4+r and forms the ten column sums
        if (act && r < 9) {
            float* out = &sh.tile[slot][r][0];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const ConstF* kp = fresh_pointer(c_kf_interior);
#pragma unroll
                for (int i = 5 * half; i < 5 * half + 5; ++i) {
                    const int p0 = i == 0 ? 0 : (i == 9 ? 2 : 1);
                    if (!REG) asm volatile("" ::: "memory");  // keeps the weight reads of later sums out of registers
                    float s = (REG ? kp[7 * i] : wl[7 * i]) * cur.v[p0];
#pragma unroll
                    for (int p = 1; p < 7; ++p) s = fmaf(REG ? kp[7 * i + p] : wl[7 * i + p], cur.v[p0 + p], s);
                    out[i] = s;
                }
            }
        }
        // the tile and the weights are private to this wavefront, whose LDS accesses complete in order
        pd_wave_sync();
        if (!REG) {
            if (act && r < 9) {
                const float4* kq = (const float4*)(a.cy + (size_t)cur.y * 72);
                ((float4*)wl)[2 * r] = kq[2 * r];
                ((float4*)wl)[2 * r + 1] = kq[2 * r + 1];
            }
            pd_wave_sync();
        }
        // y pass: lane r owns output column 8x-1+r and forms its ten row sums
        float S[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) S[j] = 0.f;
        if (act) {
            float t[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) t[q] = sh.tile[slot][q][r];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const ConstF* kp = fresh_pointer(c_kf_interior);
#pragma unroll
                for (int j = 5 * half; j < 5 * half + 5; ++j) {
                    const int p0 = j == 0 ? 0 : (j == 9 ? 2 : 1);
                    if (!REG) asm volatile("" ::: "memory");
                    float s = (REG ? kp[7 * j] : wl[7 * j]) * t[p0];
#pragma unroll
                    for (int p = 1; p < 7; ++p) s = fmaf(REG ? kp[7 * j + p] : wl[7 * j + p], t[p0 + p], s);
                    S[j] = s;
                }
            }
        }
        pd_wave_sync();
        // tolerant NMS of the cell's own 8 x 8 pixels (rows j = 1..8 of lanes r = 1..8);
        // sums of outputs outside the map are zero, like the reference's zero border
        const int X = 8 * cur.x - 1 + r;
        const bool own = act && r >= 1 && r <= 8 && X < W;
        unsigned hits = 0;
#pragma unroll
        for (int j = 1; j <= 8; ++j) {
            const float c = S[j];
            const float ce = c + cur.e2;
            const float m4 = pd_max_lane_neighbours(c, S[j - 1], S[j + 1]);
            hits |= (c > cur.thr_lo && ce >= m4) ? (1u << j) : 0u;
        }
        if (!own) hits = 0;
        if (hits) {  // rare
#pragma unroll
            for (int j = 1; j <= 8; ++j) {
                const int Y = 8 * cur.y - 1 + j;
                if (((hits >> j) & 1u) && Y < H) {
                    atomicAdd(a.cand_count + cur.plane, 1);
                    const int pos = atomicAdd(&sh.n_cand, 1);
                    if (pos < KF_LCAND)
                        sh.cand[pos] = make_int2(cur.plane, (Y << 16) | X);
                    else
                        atomicOr(a.cand_count + cur.plane, KF_OVERFLOW);  // no room: leave the plane to the exact kernel
                }
            }
        }
        if (REG)
            cur = nxt;
        else
            load_cell<REG>(a, load_entry(list, n, base + stride + cs, lane_ok, REG, a.seg_cap), r, cur);
        base += stride;
    }
}

// Workgroups of segment s = blockIdx.x % KF_NSEG share that segment's cells: first the
// ones with interior weights, then the border ones (taken from the last wavefront of the
// segment backwards, so that they fall to wavefronts with the fewest interior cells).
__global__ void __launch_bounds__(256)
k_cell_filter(CellFilterArgs a) {
    __shared__ FilterShared sh;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int seg = blockIdx.x % KF_NSEG;
    const int wave_id = (blockIdx.x / KF_NSEG) * 4 + wv;       // within the segment
    const int n_waves = (gridDim.x / KF_NSEG) * 4;             // the grid is a multiple of KF_NSEG
    if (tid == 0) sh.n_cand = 0;
    __syncthreads();
    filter_cells<true>(a, sh, seg, wave_id, n_waves, lane);
    filter_cells<false>(a, sh, seg, n_waves - 1 - wave_id, n_waves, lane);
    // staged candidates -> this segment of the global list; entries beyond its capacity
    // mark their plane as overflowing instead (the exact kernel then redoes it)
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
// Small maps: active cells and candidate filter of a plane in one launch.  The workgroup that
// has staged a plane in LDS and marked its active cells (as k_cells does) keeps the list in LDS
// and filters those cells itself, reading their 9 x 9 source windows from the staged plane: no
// cell list in memory, no second look at the map, one launch fewer.  Same arithmetic as
// k_cell_filter, so the same candidates.
// ---------------------------------------------------------------------------------
struct CellsFilterArgs {
    CellArgs c;
    CellFilterArgs f;
};

// The cells of one kind in the workgroup's LDS list `list` (n of them): six per wavefront pass,
// ten lanes per cell, windows from the plane `P` ([h][w], signed values).
template <bool REG>
__device__ __forceinline__ void filter_cells_lds(const CellFilterArgs& a, PlaneFilterShared& sh, const float* P, const unsigned short* list,
                                                 int n, int list_step, int plane, float thr_lo, float e2, int lane) {
    const int wv = threadIdx.x >> 6;
    constexpr bool one_wave = KF_PLANE_WSLOTS == KF_CELLS_PER_WAVE;
    if (!REG && one_wave && wv != 0) return;
    const int cs = lane / 10, r = lane - cs * 10;
    const bool lane_ok = cs < KF_CELLS_PER_WAVE;
    const int slot = wv * KF_CELLS_PER_WAVE + (lane_ok ? cs : 0);
    const int H = a.H, W = a.W, h = a.h, w = a.w;
    float* wl = &sh.wts[one_wave ? (lane_ok ? cs : 0) : slot][0];
    for (int base = wv * KF_CELLS_PER_WAVE; base < n; base += (!REG && one_wave) ? KF_CELLS_PER_WAVE : KF_CELLS_PER_BLOCK) {
        const int ci = base + cs;
        const bool act = lane_ok && ci < n;
        int cy = 0, cx = 0;
        float v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = 0.f;
        if (act) {
            const int e = list[ci * list_step];
            cy = (e >> 7) & 127;
            cx = e & 127;
            if (r < 9) {
                if (REG) {  // the 9 x 9 source window of a cell with interior weights lies inside the map
                    const float* p = P + (cy - 4 + r) * w + (cx - 4);
#pragma unroll
                    for (int k = 0; k < 9; ++k) v[k] = p[k];
                } else {
                    const float* row = P + pd_clampi(cy - 4 + r, 0, h - 1) * w;
#pragma unroll
                    for (int k = 0; k < 9; ++k) v[k] = row[pd_clampi(cx - 4 + k, 0, w - 1)];
                }
            }
        }
        if (!REG) {
            if (act && r < 9) {
                const float4* kq = (const float4*)(a.cx + (size_t)cx * 72);
                ((float4*)wl)[2 * r] = kq[2 * r];
                ((float4*)wl)[2 * r + 1] = kq[2 * r + 1];
            }
            pd_wave_sync();
        }
        // x pass: lane r < 9 owns source row y-4+r and forms the ten column sums
        if (act && r < 9) {
            float* out = &sh.tile[slot][r][0];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const ConstF* kp = fresh_pointer(c_kf_interior);
#pragma unroll
                for (int i = 5 * half; i < 5 * half + 5; ++i) {
                    const int p0 = i == 0 ? 0 : (i == 9 ? 2 : 1);
                    if (!REG) asm volatile("" ::: "memory");  // keeps the weight reads of later sums out of registers
                    float s = (REG ? kp[7 * i] : wl[7 * i]) * v[p0];
#pragma unroll
                    for (int p = 1; p < 7; ++p) s = fmaf(REG ? kp[7 * i + p] : wl[7 * i + p], v[p0 + p], s);
                    out[i] = s;
                }
            }
        }
        pd_wave_sync();  // the tile and the weights are private to this wavefront
        if (!REG) {
            if (act && r < 9) {
                const float4* kq = (const float4*)(a.cy + (size_t)cy * 72);
                ((float4*)wl)[2 * r] = kq[2 * r];
                ((float4*)wl)[2 * r + 1] = kq[2 * r + 1];
            }
            pd_wave_sync();
        }
        // y pass: lane r owns output column 8x-1+r and forms its ten row sums
        float S[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) S[j] = 0.f;
        if (act) {
            float t[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) t[q] = sh.tile[slot][q][r];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const ConstF* kp = fresh_pointer(c_kf_interior);
#pragma unroll
                for (int j = 5 * half; j < 5 * half + 5; ++j) {
                    const int p0 = j == 0 ? 0 : (j == 9 ? 2 : 1);
                    if (!REG) asm volatile("" ::: "memory");
                    float s = (REG ? kp[7 * j] : wl[7 * j]) * t[p0];
#pragma unroll
                    for (int p = 1; p < 7; ++p) s = fmaf(REG ? kp[7 * j + p] : wl[7 * j + p], t[p0 + p], s);
                    S[j] = s;
                }
            }
        }
        pd_wave_sync();
        // tolerant NMS of the cell's own 8 x 8 pixels (rows j = 1..8 of lanes r = 1..8);
        // sums of outputs outside the map are zero, like the reference's zero border
        const int X = 8 * cx - 1 + r;
        const bool own = act && r >= 1 && r <= 8 && X < W;
        unsigned hits = 0;
#pragma unroll
        for (int j = 1; j <= 8; ++j) {
            const float c = S[j];
            const float ce = c + e2;
            const float m4 = pd_max_lane_neighbours(c, S[j - 1], S[j + 1]);
            hits |= (c > thr_lo && ce >= m4) ? (1u << j) : 0u;
        }
        if (!own) hits = 0;
        if (hits) {  // rare
#pragma unroll
            for (int j = 1; j <= 8; ++j) {
                const int Y = 8 * cy - 1 + j;
                if (((hits >> j) & 1u) && Y < H) {
                    atomicAdd(a.cand_count + plane, 1);
                    const int pos = atomicAdd(&sh.n_cand, 1);
                    if (pos < KF_LCAND_PLANE)
                        sh.cand[pos] = make_int2(plane, (Y << 16) | X);
                    else
                        atomicOr(a.cand_count + plane, KF_OVERFLOW);  // no room: leave the plane to the exact kernel
                }
            }
        }
    }
}

__global__ void __launch_bounds__(256)
k_cells_filter(CellsFilterArgs args) {
    extern __shared__ __align__(16) float lds[];
    __shared__ PlaneFilterShared sh;
    __shared__ float s_red[4];
    __shared__ int s_cnt[4][2];
    const CellArgs& a = args.c;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per_xcd = gridDim.x / 8;
    const int plane = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;  // the 18 planes of an image on one XCD
    if (plane >= a.n_planes) return;
    const int b = plane / PD_N_PARTS, part = plane - b * PD_N_PARTS;
    const int h = a.h, w = a.w;
    const int hc = a.hc, wc = a.wc;
    const int bands = 256 / wc;
    const int rows_per = (hc + bands - 1) / bands;  // <= 32
    const int x = tid % wc, band = tid / wc;
    const int ya = min(hc, band * rows_per), yb = band < bands ? min(hc, ya + rows_per) : ya;
    float* P = lds;                    // [h][w] the plane
    // [hc * wc] cy << 7 | cx (both under 128): interior-weight cells from the front, the others from the back.  Two bytes
    // per entry: with four the workgroup took 33.5 KB and only four fitted a CU's LDS -- 1024 of the 1152 of a batch of 64
    unsigned short* list = (unsigned short*)(lds + h * w);
    if (tid == 0) sh.n_cand = 0;
    // ---- the plane to LDS, its maximum, its planar copy
    float* copy = a.planar_out ? a.planar_out + ((size_t)b * PD_CH_HEAT + part) * h * w : nullptr;
    float m = 0.f;
    if (a.nhwc) {  // from the network's interleaved output (leaving a planar copy behind if one is asked for)
        const float* pix = a.nhwc + (size_t)b * h * w * PD_CH_HEAT + part;
        for (int i0 = 0; i0 < h * w; i0 += 256 * 9) {
            float t[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int i = i0 + k * 256 + tid;
                t[k] = i < h * w ? pix[(size_t)i * PD_CH_HEAT] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int i = i0 + k * 256 + tid;
                if (i < h * w) {
                    P[i] = t[k];
                    m = absmax_step(m, t[k]);
                    if (copy) copy[i] = t[k];
                }
            }
        }
    } else {  // from the planar copy made beforehand
        for (int i0 = 0; i0 < h * w; i0 += 256 * 9) {
            float t[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int i = i0 + k * 256 + tid;
                t[k] = i < h * w ? copy[i] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int i = i0 + k * 256 + tid;
                if (i < h * w) {
                    P[i] = t[k];
                    m = absmax_step(m, t[k]);
                }
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if (lane == 0) s_red[wv] = m;
    __syncthreads();
    float thr_lo = 0.f, e2 = 0.f;
    int finite = 1;
    if (lane == 0) {
        const float pm = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        filter_slack(a.thr, pm, &thr_lo, &e2);
        finite = absmax_finite(pm);
        if (wv == 0 && !finite) atomicOr(args.f.cand_count + plane, KF_OVERFLOW);  // the exact kernel takes this plane
    }
    thr_lo = __shfl(thr_lo, 0);
    e2 = __shfl(e2, 0);
    finite = __shfl(finite, 0);
    // ---- the bound U of every cell of the thread's band (as in k_cells)
    unsigned m_reg = 0u, m_edge = 0u;
    if (yb > ya && finite) {
        float kx[7];
        int cx[7];
#pragma unroll
        for (int p = 0; p < 7; ++p) {
            kx[p] = a.bx[8 * x + p];
            cx[p] = pd_clampi(x - 3 + p, 0, w - 1);
        }
        const int regx = a.regx[x];
        float win[7];
        win[0] = 0.f;
#pragma unroll
        for (int q = 1; q < 7; ++q) {
            const float* row = P + pd_clampi(ya - 4 + q, 0, h - 1) * w;
            float v = kx[0] * fabsf(row[cx[0]]);
#pragma unroll
            for (int p = 1; p < 7; ++p) v = fmaf(kx[p], fabsf(row[cx[p]]), v);
            win[q] = v;
        }
        for (int y = ya; y < yb; ++y) {
#pragma unroll
            for (int p = 0; p < 6; ++p) win[p] = win[p + 1];
            {
                const float* row = P + min(y + 3, h - 1) * w;
                float v = kx[0] * fabsf(row[cx[0]]);
#pragma unroll
                for (int p = 1; p < 7; ++p) v = fmaf(kx[p], fabsf(row[cx[p]]), v);
                win[6] = v;
            }
            const float4 ka = *(const float4*)(a.by + 8 * y), kb = *(const float4*)(a.by + 8 * y + 4);
            float u = ka.x * win[0];
            u = fmaf(ka.y, win[1], u);
            u = fmaf(ka.z, win[2], u);
            u = fmaf(ka.w, win[3], u);
            u = fmaf(kb.x, win[4], u);
            u = fmaf(kb.y, win[5], u);
            u = fmaf(kb.z, win[6], u);
            if (u * 1.00001f >= thr_lo) {
                if (a.regy[y] & regx)
                    m_reg |= 1u << (y - ya);
                else
                    m_edge |= 1u << (y - ya);
            }
        }
    }
    // ---- the workgroup's cell list in LDS
    const int c_r = __builtin_popcount(m_reg), c_e = __builtin_popcount(m_edge);
    int p_r = c_r, p_e = c_e;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o_r = __shfl_up(p_r, d), o_e = __shfl_up(p_e, d);
        if (lane >= d) {
            p_r += o_r;
            p_e += o_e;
        }
    }
    if (lane == 63) {
        s_cnt[wv][0] = p_r;
        s_cnt[wv][1] = p_e;
    }
    __syncthreads();
    int at_r = p_r - c_r, at_e = p_e - c_e;
    for (int q = 0; q < wv; ++q) {
        at_r += s_cnt[q][0];
        at_e += s_cnt[q][1];
    }
    const int n_reg = s_cnt[0][0] + s_cnt[1][0] + s_cnt[2][0] + s_cnt[3][0];
    const int n_edge = s_cnt[0][1] + s_cnt[1][1] + s_cnt[2][1] + s_cnt[3][1];
    const int n_cells = hc * wc;
    while (m_reg) {
        const int bit = __builtin_ctz(m_reg);
        m_reg &= m_reg - 1u;
        list[at_r++] = (unsigned short)(((ya + bit) << 7) | x);
    }
    while (m_edge) {
        const int bit = __builtin_ctz(m_edge);
        m_edge &= m_edge - 1u;
        list[n_cells - 1 - (at_e++)] = (unsigned short)(((ya + bit) << 7) | x);
    }
    __syncthreads();
    // ---- the candidate filter on those cells
    filter_cells_lds<true>(args.f, sh, P, list, n_reg, 1, plane, thr_lo, e2, lane);
    filter_cells_lds<false>(args.f, sh, P, list + n_cells - 1, n_edge, -1, plane, thr_lo, e2, lane);
    __syncthreads();
    // which segment: spread by plane, or -- so that the verification of an image's candidates can run on the XCD whose L2
    // holds that image's part of the network output -- one of the four segments of this workgroup's XCD
    const int seg = args.f.seg_by_xcd ? (int)(blockIdx.x % 8) * (KF_NSEG / 8) + (plane & (KF_NSEG / 8 - 1)) : plane % KF_NSEG;
    if (tid == 64 && n_reg) atomicAdd(a.n_reg + seg, n_reg);  // the list lengths, for the statistics only
    if (tid == 128 && n_edge) atomicAdd(a.n_edge + seg, n_edge);
    const int nc = min(sh.n_cand, KF_LCAND_PLANE);
    if (nc == 0) return;
    if (tid == 0) sh.base = atomicAdd(args.f.cand_n + seg, nc);
    __syncthreads();
    for (int i = tid; i < nc; i += 256) {
        const int idx = sh.base + i;
        if (idx < args.f.cand_seg_cap)
            args.f.cand[(size_t)seg * args.f.cand_seg_cap + idx] = sh.cand[i];
        else
            atomicOr(args.f.cand_count + sh.cand[i].x, KF_OVERFLOW);
    }
}

// ---------------------------------------------------------------------------------
// Small maps, by tiles: one workgroup owns a tile of 8 x 8 cells of one image for all 18 parts.
// It reads the 16 x 16 network pixels its cells draw on (4 either side, clamped to the map) once,
// as whole (h, w, 19) pixels -- every byte of every cache line it touches is used -- and
// transposes them into per-part planes in LDS.  Tolerance: the float32 evaluation of an output of
// the tile, and the reference's own rounding of it, involve the sources of the tile only, so the
// bound eps = 2^-17 * M holds with M = max |src| of the part over the staged tile (instead of
// over the whole plane).  Bounds, list and filter then run as in k_cells_filter, on LDS alone;
// no planar copy is written: the verification reads the network output itself.
// ---------------------------------------------------------------------------------
#define TF_TC 8                             // cells per tile side
#define TF_SRC (TF_TC + 8)                  // staged pixels per tile side
#define TF_PSTR (TF_SRC + 1)                // row stride of a staged plane, in floats
#define TF_PLANE (TF_SRC * TF_PSTR + 1)     // plane stride: odd, so that the 19 channels of a pixel fall in distinct banks
#define TF_CELLS (TF_TC * TF_TC * PD_N_PARTS)

struct TileFilterShared {
    float P[PD_N_PARTS * TF_PLANE];
    union {
        float rs[PD_N_PARTS * (TF_TC + 6) * TF_TC];   // row sums of the bound: part, source row, cell column
        struct {
            float tile[KF_CELLS_PER_BLOCK][9][10];
            __attribute__((aligned(16))) float wts[KF_CELLS_PER_BLOCK][72];
        } f;
    } u;
    unsigned short list[TF_CELLS];  // part << 6 | cell row << 3 | cell column (within the tile); interior-weight cells from the front
    int2 cand[KF_LCAND_PLANE];
    unsigned int absmax[PD_N_PARTS];
    float thr_lo[PD_N_PARTS], e2[PD_N_PARTS];
    int n_reg, n_edge, n_cand, base;
};

template <bool REG>
__device__ __forceinline__ void filter_cells_tile(const CellFilterArgs& a, TileFilterShared& sh, const unsigned short* list, int n,
                                                  int list_step, int b, int cy0, int cx0, int lane) {
    const int wv = threadIdx.x >> 6;
    const int cs = lane / 10, r = lane - cs * 10;
    const bool lane_ok = cs < KF_CELLS_PER_WAVE;
    const int slot = wv * KF_CELLS_PER_WAVE + (lane_ok ? cs : 0);
    const int H = a.H, W = a.W;
    float* wl = &sh.u.f.wts[slot][0];
    for (int base = wv * KF_CELLS_PER_WAVE; base < n; base += KF_CELLS_PER_BLOCK) {
        const int ci = base + cs;
        const bool act = lane_ok && ci < n;
        int cy = 0, cx = 0, part = 0;
        float thr_lo = 0.f, e2 = 0.f;
        float v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = 0.f;
        if (act) {
            const int e = list[ci * list_step];
            part = e >> 6;
            const int cyl = (e >> 3) & 7, cxl = e & 7;
            cy = cy0 + cyl;
            cx = cx0 + cxl;
            thr_lo = sh.thr_lo[part];
            e2 = sh.e2[part];
            if (r < 9) {
                // rows cy-4 .. cy+4, columns cx-4 .. cx+4 of the map (clamped when staged) = rows cyl .. cyl+8 of the tile
                const float* p = sh.P + part * TF_PLANE + (cyl + r) * TF_PSTR + cxl;
#pragma unroll
                for (int k = 0; k < 9; ++k) v[k] = p[k];
            }
        }
        if (!REG) {
            if (act && r < 9) {
                const float4* kq = (const float4*)(a.cx + (size_t)cx * 72);
                ((float4*)wl)[2 * r] = kq[2 * r];
                ((float4*)wl)[2 * r + 1] = kq[2 * r + 1];
            }
            pd_wave_sync();
        }
        // x pass: lane r < 9 owns source row y-4+r and forms the ten column sums
        if (act && r < 9) {
            float* out = &sh.u.f.tile[slot][r][0];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const ConstF* kp = fresh_pointer(c_kf_interior);
#pragma unroll
                for (int i = 5 * half; i < 5 * half + 5; ++i) {
                    const int p0 = i == 0 ? 0 : (i == 9 ? 2 : 1);
                    if (!REG) asm volatile("" ::: "memory");  // keeps the weight reads of later sums out of registers
                    float s = (REG ? kp[7 * i] : wl[7 * i]) * v[p0];
#pragma unroll
                    for (int p = 1; p < 7; ++p) s = fmaf(REG ? kp[7 * i + p] : wl[7 * i + p], v[p0 + p], s);
                    out[i] = s;
                }
            }
        }
        pd_wave_sync();  // the tile and the weights are private to this wavefront
        if (!REG) {
            if (act && r < 9) {
                const float4* kq = (const float4*)(a.cy + (size_t)cy * 72);
                ((float4*)wl)[2 * r] = kq[2 * r];
                ((float4*)wl)[2 * r + 1] = kq[2 * r + 1];
            }
            pd_wave_sync();
        }
        // y pass: lane r owns output column 8x-1+r and forms its ten row sums
        float S[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) S[j] = 0.f;
        if (act) {
            float t[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) t[q] = sh.u.f.tile[slot][q][r];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const ConstF* kp = fresh_pointer(c_kf_interior);
#pragma unroll
                for (int j = 5 * half; j < 5 * half + 5; ++j) {
                    const int p0 = j == 0 ? 0 : (j == 9 ? 2 : 1);
                    if (!REG) asm volatile("" ::: "memory");
                    float s = (REG ? kp[7 * j] : wl[7 * j]) * t[p0];
#pragma unroll
                    for (int p = 1; p < 7; ++p) s = fmaf(REG ? kp[7 * j + p] : wl[7 * j + p], t[p0 + p], s);
                    S[j] = s;
                }
            }
        }
        pd_wave_sync();
        // tolerant NMS of the cell's own 8 x 8 pixels (rows j = 1..8 of lanes r = 1..8);
        // sums of outputs outside the map are zero, like the reference's zero border
        const int X = 8 * cx - 1 + r;
        const bool own = act && r >= 1 && r <= 8 && X < W;
        unsigned hits = 0;
#pragma unroll
        for (int j = 1; j <= 8; ++j) {
            const float c = S[j];
            const float ce = c + e2;
            const float m4 = pd_max_lane_neighbours(c, S[j - 1], S[j + 1]);
            hits |= (c > thr_lo && ce >= m4) ? (1u << j) : 0u;
        }
        if (!own) hits = 0;
        if (hits) {  // rare
            const int plane = b * PD_N_PARTS + part;
#pragma unroll
            for (int j = 1; j <= 8; ++j) {
                const int Y = 8 * cy - 1 + j;
                if (((hits >> j) & 1u) && Y < H) {
                    atomicAdd(a.cand_count + plane, 1);
                    const int pos = atomicAdd(&sh.n_cand, 1);
                    if (pos < KF_LCAND_PLANE)
                        sh.cand[pos] = make_int2(plane, (Y << 16) | X);
                    else
                        atomicOr(a.cand_count + plane, KF_OVERFLOW);  // no room: leave the plane to the exact kernel
                }
            }
        }
    }
}

__global__ void __launch_bounds__(256)
k_tile_filter(CellsFilterArgs args, int tiles_x, int tiles_per_image, int n_tiles) {
    __shared__ TileFilterShared sh;
    const CellArgs& a = args.c;
    const int tid = threadIdx.x, lane = tid & 63;
    // the tiles of an image on one XCD (workgroup i runs on XCD i % 8): neighbouring tiles share halo pixels
    const int per_xcd = gridDim.x / 8;
    const int t = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (t >= n_tiles) return;
    const int b = t / tiles_per_image, ti = t - b * tiles_per_image;
    const int ty = ti / tiles_x, tx = ti - ty * tiles_x;
    const int h = a.h, w = a.w, hc = a.hc, wc = a.wc;
    const int cy0 = ty * TF_TC, cx0 = tx * TF_TC;       // first cell of the tile = its first own network pixel
    if (tid < PD_N_PARTS) sh.absmax[tid] = 0u;
    if (tid == 0) {
        sh.n_reg = 0;
        sh.n_edge = 0;
        sh.n_cand = 0;
    }
    // ---- the tile's pixels, all 19 channels of each as they lie in memory, into per-part planes
    {
        const float* img = a.nhwc + (size_t)b * h * w * PD_CH_HEAT;
        constexpr int ROWF = TF_SRC * PD_CH_HEAT;          // floats of one staged row
        constexpr int NF = TF_SRC * ROWF;
        constexpr int PER = (NF + 255) / 256;
        float v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid + 256 * k;
            const int r = e / ROWF, rem = e - r * ROWF, c = rem / PD_CH_HEAT, ch = rem - c * PD_CH_HEAT;
            const int sy = pd_clampi(cy0 - 4 + r, 0, h - 1), sx = pd_clampi(cx0 - 4 + c, 0, w - 1);
            v[k] = e < NF ? img[((size_t)sy * w + sx) * PD_CH_HEAT + ch] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid + 256 * k;
            const int r = e / ROWF, rem = e - r * ROWF, c = rem / PD_CH_HEAT, ch = rem - c * PD_CH_HEAT;
            if (e < NF && ch < PD_N_PARTS) sh.P[ch * TF_PLANE + r * TF_PSTR + c] = v[k];
        }
    }
    __syncthreads();
    // ---- max |v| of every part over the tile: (thr_lo, e2) of its candidate test
    if (tid < PD_N_PARTS * 14) {
        const int part = tid / 14, sub = tid - part * 14;
        const float* pl = sh.P + part * TF_PLANE;
        float m = 0.f;
        for (int i = sub; i < TF_SRC * TF_SRC; i += 14) m = absmax_step(m, pl[(i >> 4) * TF_PSTR + (i & 15)]);
        atomicMax(&sh.absmax[part], __float_as_uint(m));
    }
    // ---- row sums of the bound: part, source row cy0-3+q (q < 14), cell column (a thread's cell
    // column is the same in every round: 256 is a multiple of 8)
    {
        const int cxl = tid & 7;
        const float* kxp = a.bx + 8 * min(cx0 + cxl, wc - 1);
        float kx[7];
#pragma unroll
        for (int p = 0; p < 7; ++p) kx[p] = kxp[p];
        for (int e = tid; e < PD_N_PARTS * 14 * TF_TC; e += 256) {
            const int part = e / (14 * TF_TC), rem = e - part * (14 * TF_TC), q = rem >> 3;
            const float* row = sh.P + part * TF_PLANE + (q + 1) * TF_PSTR + cxl + 1;   // columns cx-3 .. cx+3
            float s = kx[0] * fabsf(row[0]);
#pragma unroll
            for (int p = 1; p < 7; ++p) s = fmaf(kx[p], fabsf(row[p]), s);
            sh.u.rs[e] = s;
        }
    }
    __syncthreads();
    if (tid < PD_N_PARTS) {
        float tl, ee;
        const float pm = __uint_as_float(sh.absmax[tid]);
        filter_slack(a.thr, pm, &tl, &ee);
        if (!absmax_finite(pm)) {  // non-finite values in this part's share of the tile: the exact kernel takes the plane
            atomicOr(args.f.cand_count + b * PD_N_PARTS + tid, KF_OVERFLOW);
            tl = INFINITY;         // (and none of its cells in this tile is listed)
        }
        sh.thr_lo[tid] = tl;
        sh.e2[tid] = ee;
    }
    __syncthreads();
    // ---- the bound of every cell; active cells to the list (a thread's cell is the same in every
    // round, only the part changes)
    {
        const int cyl = (tid >> 3) & 7, cxl = tid & 7;
        const int cy = cy0 + cyl, cx = cx0 + cxl;
        if (cy < hc && cx < wc) {
            const float4 ka = *(const float4*)(a.by + 8 * cy), kb = *(const float4*)(a.by + 8 * cy + 4);
            const bool reg = (a.regy[cy] & a.regx[cx]) != 0;
            for (int e = tid; e < TF_CELLS; e += 256) {
                const int part = e >> 6;
                const float* col = sh.u.rs + (part * 14 + cyl) * TF_TC + cxl;       // rows cy-3 .. cy+3
                float u = ka.x * col[0];
                u = fmaf(ka.y, col[TF_TC], u);
                u = fmaf(ka.z, col[2 * TF_TC], u);
                u = fmaf(ka.w, col[3 * TF_TC], u);
                u = fmaf(kb.x, col[4 * TF_TC], u);
                u = fmaf(kb.y, col[5 * TF_TC], u);
                u = fmaf(kb.z, col[6 * TF_TC], u);
                const float tl = sh.thr_lo[part];
                if (u * 1.00001f >= tl && tl < INFINITY) {  // the factor covers the rounding of the chains on both sides
                    if (reg)
                        sh.list[atomicAdd(&sh.n_reg, 1)] = (unsigned short)e;
                    else
                        sh.list[TF_CELLS - 1 - atomicAdd(&sh.n_edge, 1)] = (unsigned short)e;
                }
            }
        }
    }
    __syncthreads();   // (the row sums are dead: the filter's staging areas take their place)
    const int n_reg = sh.n_reg, n_edge = sh.n_edge;
    filter_cells_tile<true>(args.f, sh, sh.list, n_reg, 1, b, cy0, cx0, lane);
    filter_cells_tile<false>(args.f, sh, sh.list + TF_CELLS - 1, n_edge, -1, b, cy0, cx0, lane);
    __syncthreads();
    const int seg = t % KF_NSEG;
    if (tid == 64 && n_reg) atomicAdd(a.n_reg + seg, n_reg);  // the list lengths, for the statistics only
    if (tid == 128 && n_edge) atomicAdd(a.n_edge + seg, n_edge);
    const int nc = min(sh.n_cand, KF_LCAND_PLANE);
    if (nc == 0) return;
    if (tid == 0) sh.base = atomicAdd(args.f.cand_n + seg, nc);
    __syncthreads();
    for (int i = tid; i < nc; i += 256) {
        const int idx = sh.base + i;
        if (idx < args.f.cand_seg_cap)
            args.f.cand[(size_t)seg * args.f.cand_seg_cap + idx] = sh.cand[i];
        else
            atomicOr(args.f.cand_count + sh.cand[i].x, KF_OVERFLOW);
    }
}

struct VerifyArgs {
    const float* lo;  // planar copy (B, 19, h, w), or the network output (B, h, w, 19) itself with px = 19
    int px;           // floats between horizontally adjacent pixels of a part: 1 (planar) or 19
    const double* full;  // or: full-resolution planes (B, Cfull, H, W)
    int Cfull;
    int h, w, H, W;
    Up8 up8;
    double thr;
    const int2* cand;
    int cand_seg_cap;
    const int32_t* cand_count;
    const int32_t* cand_n;
    int CK;
    RawPeak* raw;
    int32_t* raw_count;
    int PK;
    int by_xcd;  // the workgroups of XCD x (workgroup i runs on XCD i % 8) take the candidates of segments 4x .. 4x+3
};

// Exact float64 decision for every candidate.  32 lanes per candidate, VF_CAND candidates per
// workgroup pass, workgroups stride over the global list.  Lane l < 27 owns column
// X + l - 13: it forms the horizontal 4-tap sums of the (at most 8) source rows that the
// 27 bicubic rows Y-13 .. Y+13 draw on, then the 27 vertical sums and the three y-pass
// sums (rows Y-1, Y, Y+1); five lanes finish the x pass (centre and four neighbours).
// FULL: the unsmoothed map is read from the float64 planes instead of being formed from the
// network output.
// Threads of a verification workgroup: 32 lanes per candidate.  The LDS a candidate needs (5 KB)
// bounds how many are in flight on a CU; small workgroups fill that bound without a remainder.
#define VF_NT 128
#define VF_CAND (VF_NT / 32)

template <bool FULL>
__global__ void __launch_bounds__(VF_NT)
k_verify(VerifyArgs a) {
    PD_RAISE_PRIORITY(2);
    __shared__ double ypL[VF_CAND][3][28];
    __shared__ float hvL[8][VF_NT];
    __shared__ int rowL[VF_CAND][28];  // per candidate: first tap row | weight phase << 8 of its 27 window rows
    __shared__ float4 s_up[8];
    __shared__ int s_pref[KF_NSEG + 1];
    const int tid = threadIdx.x;
    const int half = tid >> 5, l = tid & 31;
    if (tid >= 64 && tid < 72) {
        const int p = tid - 64;
        s_up[p] = make_float4(a.up8.c[p][0], a.up8.c[p][1], a.up8.c[p][2], a.up8.c[p][3]);
    }
    segment_prefix(a.cand_n, a.cand_seg_cap, s_pref, tid);
    const int total = s_pref[KF_NSEG];
    const int H = a.H, W = a.W, h = a.h, w = a.w;
    int c_lo = 0, c_hi = total, first = blockIdx.x, step = gridDim.x;
    if (a.by_xcd) {  // (the grid is a multiple of 8)
        const int xcd = blockIdx.x % 8;
        c_lo = s_pref[xcd * (KF_NSEG / 8)];
        c_hi = s_pref[(xcd + 1) * (KF_NSEG / 8)];
        first = blockIdx.x / 8;
        step = gridDim.x / 8;
    }
    for (int base = c_lo + first * VF_CAND; base < c_hi; base += step * VF_CAND) {
        const int ci = base + half;
        bool active = ci < c_hi;
        int Y = 0, X = 0, plane = 0, plane_total = 0;
        double u13 = 0.0;
        if (active) {
            const int2 e = a.cand[segment_locate(s_pref, a.cand_seg_cap, ci)];
            plane = e.x;
            Y = e.y >> 16;
            X = e.y & 0xffff;
            plane_total = a.cand_count[plane];  // only needed when the decision is written
        }
        if (FULL && active && l < 27) {
            const int b = plane / PD_N_PARTS, part = plane - b * PD_N_PARTS;
            const double* img = a.full + ((size_t)b * a.Cfull + part) * (size_t)H * W + pd_reflect(X + l - 13, W);
            double u[27];
#pragma unroll
            for (int i = 0; i < 27; ++i) u[i] = img[(size_t)pd_reflect(Y - 13 + i, H) * W];
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
        if (!FULL && active && l < 27) {
            const int b = plane / PD_N_PARTS, part = plane - b * PD_N_PARTS;
            const int px = a.px;
            const float* img = px == 1 ? a.lo + ((size_t)b * PD_CH_HEAT + part) * h * w : a.lo + (size_t)b * h * w * PD_CH_HEAT + part;
            // the reflected rows Y-13 .. Y+13 all lie in [max(Y-13, 0), min(Y+13, H-1)]
            const int rmin = max(Y - 13, 0);
            const int ymin = pd_up8_ofs(rmin);
            const int col = pd_reflect(X + l - 13, W);
            const int x0 = pd_up8_ofs(col);
            const float4 ac = s_up[pd_up8_phase(col)];
            const int xi0 = pd_clampi(x0, 0, w - 1) * px, xi1 = pd_clampi(x0 + 1, 0, w - 1) * px,
                      xi2 = pd_clampi(x0 + 2, 0, w - 1) * px, xi3 = pd_clampi(x0 + 3, 0, w - 1) * px;
            float4 sv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float* row = img + (size_t)pd_clampi(ymin + q, 0, h - 1) * w * px;
                sv[q] = make_float4(row[xi0], row[xi1], row[xi2], row[xi3]);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                hvL[q][tid] = ((sv[q].x * ac.x + sv[q].y * ac.y) + sv[q].z * ac.z) + sv[q].w * ac.w;
            // Which source rows and which weights the 27 rows of the window take depends on Y alone:
            // lane i of the candidate works it out for row i, once, and all lanes read it back.
            // (Worked out per lane inside the unrolled loop, the compiler formed all 27 first and the
            // kernel held 238 registers.)
            {
                const int r = pd_reflect(Y - 13 + l, H);
                rowL[half][l] = min(pd_up8_ofs(r) - ymin, 4) | (pd_up8_phase(r) << 8);
            }
            pd_wave_sync();  // the 32 lanes of a candidate share a wavefront
            double u[27];
#pragma unroll
            for (int i = 0; i < 27; ++i) {
                const int m = rowL[half][i];
                const int d = m & 255;
                const float4 bc = s_up[m >> 8];
                u[i] = (double)(((hvL[d][tid] * bc.x + hvL[d + 1][tid] * bc.y) + hvL[d + 2][tid] * bc.z) + hvL[d + 3][tid] * bc.w);
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
        if (active && l == 0 && plane_total <= a.CK) {  // overflowing planes are redone by the exact kernel
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



int pd_grow(pd_ctx* ctx, void** p, size_t* cap, size_t need_bytes) {
    if (*cap >= need_bytes) return PD_OK;
    PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->gen++;
    if (*p) PD_HIP(ctx, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    PD_HIP(ctx, hipMalloc(p, need_bytes));
    *cap = need_bytes;
    return PD_OK;
}

// The candidate list: room for CK = 4 * max_peaks_per_part entries per plane.
int pd_ensure_candidates(pd_ctx* ctx, int* CK_out) {
    const int CK = 4 * ctx->cfg.max_peaks_per_part;
    const size_t planes_max = (size_t)ctx->cfg.max_batch * PD_N_PARTS;
    if (planes_max >= (1u << 17)) return PD_E_STATE;  // plane index is packed in 17 bits
    if (!ctx->cand || ctx->cand_cap != CK) {
        PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->gen++;
        if (ctx->cand) PD_HIP(ctx, hipFree(ctx->cand));
        ctx->cand = nullptr;
        PD_HIP(ctx, hipMalloc(&ctx->cand, sizeof(int2) * planes_max * CK));
        ctx->cand_cap = CK;
    }
    *CK_out = CK;
    return PD_OK;
}

int pd_launch_peaks_filtered(pd_ctx* ctx, double thr) {
    int rc = build_filter_tables(ctx);
    if (rc != PD_OK) return rc;
    const int B = ctx->B, H = ctx->H, W = ctx->W, h = ctx->h, w = ctx->w;
    const int hc = (H + 7) / 8, wc = (W + 7) / 8;
    int CK = 0;
    rc = pd_ensure_candidates(ctx, &CK);
    if (rc != PD_OK) return rc;
    rc = pd_grow(ctx, (void**)&ctx->heat_planar, &ctx->heat_planar_cap, sizeof(float) * (size_t)B * PD_CH_HEAT * h * w);
    if (rc != PD_OK) return rc;
    const size_t planes = (size_t)B * PD_N_PARTS;
    const size_t n_cells_max = planes * hc * wc;
    // a segment holds the cells of the planes p with p % KF_NSEG == its index
    const size_t seg_cap = ((planes + KF_NSEG - 1) / KF_NSEG) * hc * wc;
    rc = pd_grow(ctx, (void**)&ctx->cells, &ctx->cells_cap, sizeof(int32_t) * seg_cap * KF_NSEG);
    if (rc != PD_OK) return rc;
    // counters behind the per-plane candidate counts, KF_NSEG each: candidates, interior cells, border cells
    int32_t* counters = ctx->cand_count + planes;
    const float* tab = ctx->kf_tab;

    CellArgs c;
    c.lo = ctx->heat_planar;
    c.h = h;
    c.w = w;
    c.hc = hc;
    c.wc = wc;
    c.by = tab + (size_t)72 * (hc + wc);
    c.bx = c.by + (size_t)8 * hc;
    c.regy = ctx->kf_reg;
    c.regx = ctx->kf_reg + hc;
    c.slack = ctx->plane_slack;
    c.n_planes = (int)planes;
    // Small maps: the active-cell kernel reads the network output itself and leaves the planar
    // copy behind (one launch fewer; the decode of a small batch is bound by launch latency).
    // Large maps: the strided reads would pull every cache line through the L2 eighteen times,
    // so the copy is made first, by a kernel that reads and writes coalesced.
    static const bool planar_first = [] {
        const char* v = getenv("PD_PLANAR_FIRST");
        return v && *v && *v != '0';
    }();
    const bool small_map = (size_t)h * w <= 4096;
    const bool gather = small_map && !planar_first;
    c.nhwc = gather ? ctx->lo_heat : nullptr;
    c.planar_out = ctx->heat_planar;
    c.thr = thr;
    c.cells = ctx->cells;
    c.seg_cap = (int)seg_cap;
    c.n_reg = counters + KF_NSEG;
    c.n_edge = counters + 2 * KF_NSEG;
    c.cand_count = ctx->cand_count;

    CellFilterArgs f;
    f.lo = ctx->heat_planar;
    f.h = h;
    f.w = w;
    f.H = H;
    f.W = W;
    f.cy = tab;
    f.cx = tab + (size_t)72 * hc;
    f.slack = ctx->plane_slack;
    f.cells = ctx->cells;
    f.seg_cap = c.seg_cap;
    f.n_reg = c.n_reg;
    f.n_edge = c.n_edge;
    f.cand = (int2*)ctx->cand;
    f.cand_seg_cap = (int)(planes * CK / KF_NSEG);
    f.cand_count = ctx->cand_count;
    f.cand_n = counters;
    f.CK = CK;
    f.seg_by_xcd = 0;

    // cand_count and the list counters were cleared together with raw_count
    static const bool split_cells = [] {
        const char* v = getenv("PD_SPLIT_CELLS");
        return v && *v && *v != '0';
    }();  // comparison runs: the two launches also for small maps
    // PD_TILE_FILTER=1 (comparison runs): small maps by tiles of 8 x 8 cells for all 18 parts -- every
    // pixel of the network output read once, as a whole, no planar copy, the verification reading
    // the network output itself.  Measured at batch 64, 368^2: 52 us per launch against 40 us for the
    // per-plane form below (24 us of it before the filter phase against 21; the filter phase 28 us
    // against 18: a tile under a person holds up to several hundred active cells of its 18 parts,
    // an empty one none, where every plane holds one blob per person), 1.25 against 1.29 M images/s.
    // The strided reads of the per-plane form cost L2-to-L1 traffic, not time.
    static const bool plane_form = [] {
        const char* v = getenv("PD_TILE_FILTER");
        return !(v && *v && *v != '0');
    }();
    bool nhwc_verify = false, verify_by_xcd = false;
    if (small_map && gather && !split_cells && !plane_form && hc <= h && wc <= w && hc < 128 && wc < 128) {
        CellsFilterArgs cf;
        cf.c = c;
        cf.f = f;
        const int tiles_x = (wc + TF_TC - 1) / TF_TC, tiles_y = (hc + TF_TC - 1) / TF_TC;
        const int n_tiles = tiles_x * tiles_y * B;
        pd_prof_begin(ctx, PD_STAGE_FILTER);
        hipLaunchKernelGGL(k_tile_filter, dim3((unsigned)((n_tiles + 7) / 8 * 8)), dim3(256), 0, ctx->stream, cf, tiles_x,
                           tiles_x * tiles_y, n_tiles);
        pd_prof_end(ctx);
        PD_HIP(ctx, hipGetLastError());
        nhwc_verify = true;
    } else {
    const size_t fused_lds = sizeof(float) * (size_t)h * w + ((sizeof(unsigned short) * (size_t)hc * wc + 15) & ~(size_t)15);
    if (small_map && !split_cells && 256 / wc >= 1 && (hc + 256 / wc - 1) / (256 / wc) <= 32 && hc <= h && wc <= w && hc < 128 && wc < 128 &&
        fused_lds <= 40 * 1024) {
        CellsFilterArgs cf;
        cf.c = c;
        cf.f = f;
        static const bool keep_copy = [] {
            const char* v = getenv("PD_PLANAR_COPY");
            return v && *v && *v != '0';
        }();  // comparison runs: the planar copy written on the way and read by the verification
        if (gather && !keep_copy) {
            cf.c.planar_out = nullptr;
            cf.f.seg_by_xcd = 1;
            nhwc_verify = true;
            verify_by_xcd = true;
        }
        if (!gather) {
            pd_prof_begin(ctx, PD_STAGE_PLANAR);
            hipLaunchKernelGGL(k_heat_planar, dim3((h * w + KP_PIX - 1) / KP_PIX, B), dim3(256), 0, ctx->stream, ctx->lo_heat,
                               h * w, ctx->heat_planar, (unsigned int*)nullptr);
            pd_prof_end(ctx);
        }
        pd_prof_begin(ctx, PD_STAGE_FILTER);
        hipLaunchKernelGGL(k_cells_filter, dim3((unsigned)((planes + 7) / 8 * 8)), dim3(256), fused_lds, ctx->stream, cf);
        pd_prof_end(ctx);
        PD_HIP(ctx, hipGetLastError());
    } else {
    if (!gather) {
        pd_prof_begin(ctx, PD_STAGE_PLANAR);
        hipLaunchKernelGGL(k_heat_planar, dim3((h * w + KP_PIX - 1) / KP_PIX, B), dim3(256), 0, ctx->stream, ctx->lo_heat,
                           h * w, ctx->heat_planar, (unsigned int*)nullptr);
        pd_prof_end(ctx);
    }
    static const int cells_nt = getenv("PD_CELLS_THREADS") ? atoi(getenv("PD_CELLS_THREADS")) : 0;  // comparison runs
    const int nt_cells = cells_nt > 0 ? cells_nt : (gather ? 256 : 512);
    const int bands = nt_cells / wc;
    const int rows_per = bands ? (hc + bands - 1) / bands : 64;
    const size_t cells_lds = sizeof(float) * (size_t)h * w;
    if (nt_cells % 64 || nt_cells > 512 || rows_per > 32 || hc > h || wc > w || cells_lds > 60 * 1024) return PD_E_STATE;  // (a bit per row of a band)
    pd_prof_begin(ctx, PD_STAGE_CELLS);
    hipLaunchKernelGGL(k_cells, dim3((unsigned)((planes + 7) / 8 * 8)), dim3(nt_cells), cells_lds, ctx->stream, c);
    pd_prof_end(ctx);
    // the list lengths are only known on the device: fixed grids stride over them
    size_t fblocks = (n_cells_max / 8 + KF_CELLS_PER_BLOCK - 1) / KF_CELLS_PER_BLOCK;
    // four workgroups per CU: with more, the kernel holds every wave slot of the chip and the
    // kernels of the other decode contexts queue behind it (3 % fewer images/s at 2048)
    if (fblocks > 1024) fblocks = 1024;
    fblocks = (fblocks + KF_NSEG - 1) / KF_NSEG * KF_NSEG;  // whole workgroups per segment
    pd_prof_begin(ctx, PD_STAGE_FILTER);
    hipLaunchKernelGGL(k_cell_filter, dim3((unsigned)fblocks), dim3(256), 0, ctx->stream, f);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    }
    }

    VerifyArgs v;
    v.full = nullptr;
    v.Cfull = 0;
    v.lo = nhwc_verify ? ctx->lo_heat : ctx->heat_planar;
    v.px = nhwc_verify ? PD_CH_HEAT : 1;
    v.h = h;
    v.w = w;
    v.H = H;
    v.W = W;
    pd_up8_fill(&v.up8);
    v.thr = thr;
    v.cand = (const int2*)ctx->cand;
    v.cand_seg_cap = f.cand_seg_cap;
    v.cand_count = ctx->cand_count;
    v.cand_n = counters;
    v.CK = CK;
    v.raw = ctx->raw;
    v.raw_count = ctx->raw_count;
    v.PK = ctx->cfg.max_peaks_per_part;
    v.by_xcd = verify_by_xcd ? 1 : 0;
    int vblocks = B * 12 * (256 / VF_NT);
    if (vblocks > 1024 * (256 / VF_NT)) vblocks = 1024 * (256 / VF_NT);
    pd_prof_begin(ctx, PD_STAGE_VERIFY);
    hipLaunchKernelGGL(k_verify<false>, dim3(vblocks), dim3(VF_NT), 0, ctx->stream, v);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    // planes whose candidates did not fit; the same kernel then orders every image's peaks
    return pd_launch_stream_exact(ctx, thr, 0, nullptr, ctx->cand_count, CK, 1);
}

// ---------------------------------------------------------------------------------
// Candidate filter for full-resolution float64 maps (after the multi-scale average, or
// maps supplied by the caller): the same two-step scheme -- a float32 evaluation of the
// smoothed map with fused multiply-adds on every pixel, tolerant NMS, exact float64
// verification of the candidates (k_verify<true>).  With x the float64 map, M >= max |x|:
// rounding x and the weights to float32 and two 25-term FMA chains give
//     |S~ - S| <= 60 * 2^-24 * M  <  eps = 2^-17 * M,
// the tolerance the network-resolution filter uses.
//
// One workgroup owns one (image, part, column strip) and marches down the rows; a thread owns
// two adjacent columns, whose 25-row windows live in its registers (slot = row mod 25, made
// static by unrolling 25 row steps): the y pass needs no memory, the y-pass row goes through
// LDS once for the x pass, NMS of a row runs one step later from registers and the previous
// row's LDS copy.  Streaming: every map value is read once (8 bytes per pixel).
// ---------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
#define FF_NT 256
#define FF_COLS (2 * FF_NT)

static __constant__ float c_gwf[PD_GTAPS] = PD_GAUSS_WEIGHTS;

struct FullFilterArgs {
    const double* full;  // (B, Cfull, H, W)
    int Cfull, H, W;
    const float* absmax;  // [B] max |map value| per image (rounded up)
    double thr;
    int n_strips, strip_w;
    int2* cand;
    int cand_seg_cap;
    int32_t* cand_count;
    int32_t* cand_n;
};

__device__ __forceinline__ void ff_emit(const FullFilterArgs& a, int plane, int Y, int X) {
    atomicAdd(a.cand_count + plane, 1);
    const int seg = plane % KF_NSEG;
    const int pos = atomicAdd(a.cand_n + seg, 1);
    if (pos < a.cand_seg_cap)
        a.cand[(size_t)seg * a.cand_seg_cap + pos] = make_int2(plane, (Y << 16) | X);
    else
        atomicOr(a.cand_count + plane, KF_OVERFLOW);  // no room: the exact kernel redoes the plane
}

#define FF_AHEAD 5  // rows on their way from memory per thread; divides the 25 unrolled row steps
static_assert(25 % FF_AHEAD == 0 && PD_GR + FF_AHEAD < 2 * PD_GR + 2,
              "a row's slot must be static, and a row requested ahead must reflect into the map (H >= 2 * PD_GR + 2)");
struct FFRow {
    double x, y;
};
struct FFState {
    f2 s_prev1, s_prev2;   // smoothed rows y-1, y-2 of this thread's two columns
    FFRow pre[FF_AHEAD];   // map rows y+12 .. y+16 as loaded (converted when they enter the window,
                           // so that nothing waits for them earlier); row r sits in slot r % FF_AHEAD
};

// Two adjacent map values of row r (reflected at the lower edge).
__device__ __forceinline__ FFRow ff_load_row(const double* __restrict__ col, int r, int H, int W, bool act1) {
    if (r >= H) r = 2 * H - 1 - r;
    const double* p = col + (size_t)r * W;
    FFRow v;
    v.x = p[0];
    v.y = act1 ? p[1] : 0.0;
    return v;
}

template <int S>
__device__ __forceinline__ void ff_row_step(const FullFilterArgs& a, const double* __restrict__ col, f2 (&win)[25], FFState& st,
                                            float* ybuf, float* sbuf, int y, int H, int W, int c, int xs, int xe, bool act0,
                                            bool act1, int x0, int x1, int plane, float thr_lo, float e2) {
    const int tid = threadIdx.x;
    const int par = y & 1;
    float* yrow = ybuf + par * (FF_COLS + 2 * PD_GR + 2);
    const int li = c - xs;  // this thread's first column inside the strip
    if (y < H && act0) {
        // row y + 12 was requested five steps ago; row y + 17 is requested now into its slot
        // (12 % 5 == 17 % 5 and 25 % 5 == 0: the slot of a step is static)
        {
            const FFRow in = st.pre[(S + PD_GR) % FF_AHEAD];
            f2 v;
            v.x = (float)in.x;
            v.y = (float)in.y;
            win[(S + 12) % 25] = v;
        }
        st.pre[(S + PD_GR) % FF_AHEAD] = ff_load_row(col, y + PD_GR + FF_AHEAD, H, W, act1);
        f2 t = win[S] * c_gwf[12];
#pragma unroll
        for (int d = 12; d >= 1; --d) t = __builtin_elementwise_fma(win[(S + 25 - d) % 25] + win[(S + d) % 25], (f2)(c_gwf[12 - d]), t);
        *(f2*)(yrow + li + PD_GR) = t;
        // mirrored halo so the x pass reads straight offsets
        if (xs == 0 && c < PD_GR) {
            yrow[PD_GR - 1 - c] = t.x;
            if (c + 1 < PD_GR) yrow[PD_GR - 2 - c] = t.y;
        }
        if (xe == W && c + 1 >= W - PD_GR) {
            if (c >= W - PD_GR) yrow[(2 * W - 1 - c) - xs + PD_GR] = t.x;
            if (act1) yrow[(2 * W - 2 - c) - xs + PD_GR] = t.y;
        }
    }
    __syncthreads();
    f2 s = (f2)(0.f);
    if (y < H && act0) {
        const float* p = yrow + li;  // p[k] = column c - 12 + k
        float q[26];
#pragma unroll
        for (int k = 0; k < 13; ++k) {
            const f2 t = *(const f2*)(p + 2 * k);
            q[2 * k] = t.x;
            q[2 * k + 1] = t.y;
        }
        float sx = q[12] * c_gwf[12], sy = q[13] * c_gwf[12];
#pragma unroll
        for (int d = 12; d >= 1; --d) {
            sx = fmaf(q[12 - d] + q[12 + d], c_gwf[12 - d], sx);
            sy = fmaf(q[13 - d] + q[13 + d], c_gwf[12 - d], sy);
        }
        s.x = sx;
        s.y = act1 ? sy : 0.f;
    }
    float* srow_w = sbuf + par * (FF_COLS + 4);
    *(f2*)(srow_w + 2 * tid + 2) = s;  // srow[k + 2] = strip column k; srow[1] and the slot behind the strip stay 0
    if (y >= 1) {
        // NMS of row y - 1: centre and vertical neighbours from registers, horizontal ones from
        // the LDS copy the previous step wrote (this step's barrier has passed)
        const float* srow = sbuf + (par ^ 1) * (FF_COLS + 4);
        const f2 m = st.s_prev1;
        const f2 up = (y == 1) ? (f2)(0.f) : st.s_prev2;
        const f2 down = (y == H) ? (f2)(0.f) : s;
        const float left0 = (c == 0) ? 0.f : srow[2 * tid + 1];
        const float right1 = (c + 1 >= W - 1) ? 0.f : srow[2 * tid + 4];
        const float n0 = fmaxf(fmaxf(up.x, down.x), fmaxf(left0, (c + 1 < W) ? m.y : 0.f));
        const float n1 = fmaxf(fmaxf(up.y, down.y), fmaxf(m.x, right1));
        if (act0 && c >= x0 && c < x1 && m.x > thr_lo && m.x + e2 >= n0) ff_emit(a, plane, y - 1, c);
        if (act1 && c + 1 >= x0 && c + 1 < x1 && m.y > thr_lo && m.y + e2 >= n1) ff_emit(a, plane, y - 1, c + 1);
    }
    st.s_prev2 = st.s_prev1;
    st.s_prev1 = s;
}

__global__ void __launch_bounds__(FF_NT)
k_full_filter(FullFilterArgs a) {
    __shared__ __align__(16) float ybuf[2 * (FF_COLS + 2 * PD_GR + 2)];
    __shared__ __align__(16) float sbuf[2 * (FF_COLS + 4)];
    __shared__ float s_slack[2];
    const int H = a.H, W = a.W;
    const int b = blockIdx.z, part = blockIdx.y, strip = blockIdx.x;
    const int plane = b * PD_N_PARTS + part;
    const int x0 = strip * a.strip_w;
    const int x1 = min(W, x0 + a.strip_w);
    const int xs = max(0, x0 - (PD_GR + 1)) & ~1;
    const int xe = min(W, x1 + (PD_GR + 1));
    const int tid = threadIdx.x;
    const int c = xs + 2 * tid;
    const bool act0 = c < xe, act1 = c + 1 < xe;
    if (tid == 0) {
        float t, e;
        filter_slack(a.thr, a.absmax[b], &t, &e);
        s_slack[0] = t;
        s_slack[1] = e;
    }
    // (the workgroup has as many wavefronts as the strip has column pairs, not always FF_NT threads)
    for (int i = tid; i < 2 * (FF_COLS + 4); i += blockDim.x) sbuf[i] = 0.f;
    for (int i = tid; i < 2 * (FF_COLS + 2 * PD_GR + 2); i += blockDim.x) ybuf[i] = 0.f;
    __syncthreads();
    const float thr_lo = s_slack[0], e2 = s_slack[1];
    const double* col = a.full + ((size_t)b * a.Cfull + part) * (size_t)H * W + (act0 ? c : 0);
    f2 win[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) win[i] = (f2)(0.f);
    if (act0) {
        // rows 0..11 fill slots 0..11 and, mirrored, rows -1..-12 -> slots 24..13
#pragma unroll
        for (int r = 0; r < PD_GR; ++r) {
            const double* p = col + (size_t)r * W;
            f2 v;
            v.x = (float)p[0];
            v.y = act1 ? (float)p[1] : 0.f;
            win[r] = v;
            win[24 - r] = v;
        }
    }
    FFState st;
    st.s_prev1 = (f2)(0.f);
    st.s_prev2 = (f2)(0.f);
#pragma unroll
    for (int i = 0; i < FF_AHEAD; ++i) {
        st.pre[(PD_GR + i) % FF_AHEAD].x = 0.0;
        st.pre[(PD_GR + i) % FF_AHEAD].y = 0.0;
        if (act0) st.pre[(PD_GR + i) % FF_AHEAD] = ff_load_row(col, PD_GR + i, H, W, act1);
    }
#define FF_STEP(S)                                                                                              \
    if (y0 + S > H) break;                                                                                      \
    ff_row_step<S>(a, col, win, st, ybuf, sbuf, y0 + S, H, W, c, xs, xe, act0, act1, x0, x1, plane, thr_lo, e2);
    for (int y0 = 0; y0 <= H; y0 += 25) {
        FF_STEP(0) FF_STEP(1) FF_STEP(2) FF_STEP(3) FF_STEP(4) FF_STEP(5) FF_STEP(6) FF_STEP(7)
        FF_STEP(8) FF_STEP(9) FF_STEP(10) FF_STEP(11) FF_STEP(12) FF_STEP(13) FF_STEP(14)
        FF_STEP(15) FF_STEP(16) FF_STEP(17) FF_STEP(18) FF_STEP(19) FF_STEP(20) FF_STEP(21)
        FF_STEP(22) FF_STEP(23) FF_STEP(24)
    }
#undef FF_STEP
}

// Peak stage on the full-resolution float64 heatmaps the context holds (ctx->full_absmax must
// bound them): candidate filter, exact verification, exact kernel for overflowing planes.
int pd_launch_peaks_filtered_full(pd_ctx* ctx, double thr) {
    const int B = ctx->B, H = ctx->H, W = ctx->W;
    if (H >= 65536 || W >= 65536) return PD_E_STATE;  // candidates pack (y, x) in 16 bits each
    int CK = 0;
    int rc = pd_ensure_candidates(ctx, &CK);
    if (rc != PD_OK) return rc;
    const size_t planes = (size_t)B * PD_N_PARTS;
    int32_t* counters = ctx->cand_count + planes;
    FullFilterArgs f;
    f.full = ctx->full_heat;
    f.Cfull = PD_CH_HEAT;
    f.H = H;
    f.W = W;
    f.absmax = ctx->full_absmax;
    f.thr = thr;
    // strips: owned columns plus 13 either side (and one for the even start) within 2 * FF_NT columns
    const int halo = 2 * (PD_GR + 1) + 1;
    f.n_strips = W + 1 <= FF_COLS ? 1 : (W + (FF_COLS - halo) - 1) / (FF_COLS - halo);
    f.strip_w = (W + f.n_strips - 1) / f.n_strips;
    f.cand = (int2*)ctx->cand;
    f.cand_seg_cap = (int)(planes * CK / KF_NSEG);
    f.cand_count = ctx->cand_count;
    f.cand_n = counters;
    pd_prof_begin(ctx, PD_STAGE_FILTER);
    // threads: one per column pair of the widest strip (owned columns + 13 either side, even start)
    int nt = ((min(W, f.strip_w + halo) + 1) / 2 + 63) / 64 * 64;
    if (nt > FF_NT) nt = FF_NT;
    hipLaunchKernelGGL(k_full_filter, dim3(f.n_strips, PD_N_PARTS, B), dim3(nt), 0, ctx->stream, f);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());

    VerifyArgs v;
    v.lo = nullptr;
    v.px = 1;
    v.full = ctx->full_heat;
    v.Cfull = PD_CH_HEAT;
    v.h = v.w = 0;
    v.H = H;
    v.W = W;
    pd_up8_fill(&v.up8);
    v.thr = thr;
    v.cand = (const int2*)ctx->cand;
    v.cand_seg_cap = f.cand_seg_cap;
    v.cand_count = ctx->cand_count;
    v.cand_n = counters;
    v.CK = CK;
    v.raw = ctx->raw;
    v.raw_count = ctx->raw_count;
    v.PK = ctx->cfg.max_peaks_per_part;
    v.by_xcd = 0;
    int vblocks = B * 12 * (256 / VF_NT);
    if (vblocks > 1024 * (256 / VF_NT)) vblocks = 1024 * (256 / VF_NT);
    pd_prof_begin(ctx, PD_STAGE_VERIFY);
    hipLaunchKernelGGL(k_verify<true>, dim3(vblocks), dim3(VF_NT), 0, ctx->stream, v);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    return pd_launch_stream_exact(ctx, thr, 0, nullptr, ctx->cand_count, CK, 1);
}
