// PAF line-integral scoring of every limb candidate pair and the per-limb greedy
// matching: get_connections of utils/postprocess.py:78-160 (twin predict.py:120-174).
//
// Per pair (a, b) of limb k, in float64 with separately rounded operations:
//   step = (b - a) / 9;  p_i = i * step + a  (p_9 = b)             np.linspace, :121-122
//   index = trunc(p_i) [mode P, :123]  |  rint(p_i), half-even [mode D, predict.py:147]
//   mid_i = v0 * u0 + v1 * u1                                        :128
//   score = ((((0 + mid_0) + mid_1) + ..) + mid_9) / 10 + min(0.5 * H / norm - 1, 0)   :129
//   kept iff count(mid_i > thre2) >= 9 and score > 0                 :131-136
// Candidates are taken in descending score order, ties in a-major / b-minor order
// (stable sort, :141), a peak being used at most once per side (:144-156).
#include <math.h>
#include <stdlib.h>

#include "match.h"
#include "multiscale.h"

#pragma clang fp contract(off)

struct PairArgs {
    MapSource src;  // PAF source
    int mode;
    double thr;
    const int32_t* pk_y;
    const int32_t* pk_x;
    const int32_t* part_count;
    const int32_t* pair_off;
    int PK;
    int max_pairs;
    double* pair_score;
    int32_t* status;
    int per_image;  // workgroups per image
    const float* pairs;  // (B, 19, h, w, 2) copy of the PAFs, or NULL: read the (h, w, 38) maps
};

// Two adjacent channels (ch, ch+1) of a (h, w, C) float32 map, bicubic at (Y, X);
// same arithmetic as pd_bicubic_at.
// Taps and weights come from the closed form of the x8 table (weights by phase in LDS),
// the two channels of one pixel from one 8-byte load (ch even, C even).
__device__ __forceinline__ void bicubic2_at(const float* __restrict__ img, int h, int w, int C, int ch, int Y,
                                            int X, const float4* __restrict__ up, float* o0, float* o1) {
    // C = floats between horizontally adjacent pixels, ch = offset of the channel pair
    const int x0 = pd_up8_ofs(X);
    const int y0 = pd_up8_ofs(Y);
    const float4 ac = up[pd_up8_phase(X)];
    const float4 bc = up[pd_up8_phase(Y)];
    const int xi0 = pd_clampi(x0, 0, w - 1) * C, xi1 = pd_clampi(x0 + 1, 0, w - 1) * C,
              xi2 = pd_clampi(x0 + 2, 0, w - 1) * C, xi3 = pd_clampi(x0 + 3, 0, w - 1) * C;
    float2 v[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* row = img + ((size_t)pd_clampi(y0 + k, 0, h - 1) * w) * C + ch;
        v[k][0] = *(const float2*)(row + xi0);
        v[k][1] = *(const float2*)(row + xi1);
        v[k][2] = *(const float2*)(row + xi2);
        v[k][3] = *(const float2*)(row + xi3);
    }
    float h0[4], h1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        h0[k] = ((v[k][0].x * ac.x + v[k][1].x * ac.y) + v[k][2].x * ac.z) + v[k][3].x * ac.w;
        h1[k] = ((v[k][0].y * ac.x + v[k][1].y * ac.y) + v[k][2].y * ac.z) + v[k][3].y * ac.w;
    }
    *o0 = ((h0[0] * bc.x + h0[1] * bc.y) + h0[2] * bc.z) + h0[3] * bc.w;
    *o1 = ((h1[0] * bc.x + h1[1] * bc.y) + h1[2] * bc.z) + h1[3] * bc.w;
}

// The same for the pair-planar copy (h, w, 2): the four horizontal taps of both channels
// are 32 contiguous bytes, fetched with two 16-byte loads per row when no tap is clamped.
struct __attribute__((aligned(8))) float4a8 {
    float x, y, z, w;
};

__device__ __forceinline__ void bicubic2_planar(const float* __restrict__ img, int h, int w, int Y, int X,
                                                const float4* __restrict__ up, float* o0, float* o1) {
    const int x0 = pd_up8_ofs(X);
    if (x0 < 0 || x0 + 3 > w - 1) {  // a clamped tap: the general form
        bicubic2_at(img, h, w, 2, 0, Y, X, up, o0, o1);
        return;
    }
    const int y0 = pd_up8_ofs(Y);
    const float4 ac = up[pd_up8_phase(X)];
    const float4 bc = up[pd_up8_phase(Y)];
    float4a8 lo[4], hi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* row = img + ((size_t)pd_clampi(y0 + k, 0, h - 1) * w + x0) * 2;
        lo[k] = *(const float4a8*)row;        // (c0, c1) of taps 0 and 1
        hi[k] = *(const float4a8*)(row + 4);  // (c0, c1) of taps 2 and 3
    }
    float h0[4], h1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        h0[k] = ((lo[k].x * ac.x + lo[k].z * ac.y) + hi[k].x * ac.z) + hi[k].z * ac.w;
        h1[k] = ((lo[k].y * ac.x + lo[k].w * ac.y) + hi[k].y * ac.z) + hi[k].w * ac.w;
    }
    *o0 = ((h0[0] * bc.x + h0[1] * bc.y) + h0[2] * bc.z) + h0[3] * bc.w;
    *o1 = ((h1[0] * bc.x + h1[1] * bc.y) + h1[2] * bc.z) + h1[3] * bc.w;
}

// What the ten sample lanes of a pair need, computed once per pair.
struct PairGeo {
    double u0, u1;   // unit vector components paired with channels 2k and 2k+1
    double sy, sx;   // linspace steps
    double prior;    // min(0.5 * H / norm - 1, 0)
    int ya, xa, yb, xb;
    int k;           // limb, or -1: the pair is skipped (coincident peaks in mode P) or absent
    int pad;         // pre-test: samples found at or under the threshold so far
};

#define PAIRS_PER_CHUNK 6   // pairs a wavefront takes at a time: 6 lanes x 1 pair, then 6 pairs x 10 samples
#define PRE_PAIRS 128       // pairs a wavefront pre-tests (one lane each) before it scores the survivors

// Limb, peak indices and unit vector of candidate pair p of image b; false: the pair is
// skipped (coincident peaks in mode P).
__device__ __forceinline__ bool pair_geometry(const PairArgs& a, int b, int p, const int32_t* s_off,
                                              const int32_t* s_cnt, PairGeo& g) {
    int k = 0;
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1)
        if (k + d <= PD_N_LIMBS - 1 && s_off[k + d] <= p) k += d;
    const int pa = pd_limb_from(k), pb = pd_limb_to(k);
    const int nB = s_cnt[pb];
    const int local = p - s_off[k];
    const int ia = local / nB, ib = local - ia * nB;
    const size_t oa = ((size_t)b * PD_N_PARTS + pa) * a.PK + ia;
    const size_t ob = ((size_t)b * PD_N_PARTS + pb) * a.PK + ib;
    g.ya = a.pk_y[oa];
    g.xa = a.pk_x[oa];
    g.yb = a.pk_y[ob];
    g.xb = a.pk_x[ob];
    const long long dy = (long long)g.yb - g.ya, dx = (long long)g.xb - g.xa;
    const double d2 = (double)(dy * dy + dx * dx);
    double norm = sqrt(d2);
    bool skip = false;
    if (a.mode == PD_MODE_P)
        skip = d2 == 0.0;  // coincident peaks, utils/postprocess.py:112
    else
        norm = fmax(norm, 1e-5);  // predict.py:141
    const double uy = (double)dy / norm, ux = (double)dx / norm;
    if (a.mode == PD_MODE_P) {  // channel 2k pairs with dy, 2k+1 with dx
        g.u0 = uy;
        g.u1 = ux;
    } else {                    // channel 2k pairs with dx, 2k+1 with dy
        g.u0 = ux;
        g.u1 = uy;
    }
    g.sy = (double)dy / 9.0;
    g.sx = (double)dx / 9.0;
    g.prior = fmin(0.5 * (double)a.src.H / norm - 1.0, 0.0);
    g.k = skip ? -1 : k;
    g.pad = 0;
    return !skip;
}

// v0 * u0 + v1 * u1 at sample i of the pair (i = 9 is the second peak itself).
// MULTI: the PAF values are the average over the test scales of *ms, evaluated at the sample.
template <bool MULTI>
__device__ __forceinline__ double pair_sample(const PairArgs& a, const MultiSource* ms, int b, const PairGeo& g, int i,
                                              const float4* s_up) {
    const int H = a.src.H, W = a.src.W;
    double py, px;
    if (i == PD_MID - 1) {
        py = (double)g.yb;
        px = (double)g.xb;
    } else {
        py = (double)i * g.sy + (double)g.ya;
        px = (double)i * g.sx + (double)g.xa;
    }
    int iy, ix;
    if (a.mode == PD_MODE_P) {
        iy = (int)py;
        ix = (int)px;
    } else {
        iy = (int)rint(py);
        ix = (int)rint(px);
    }
    iy = pd_clampi(iy, 0, H - 1);
    ix = pd_clampi(ix, 0, W - 1);
    double v0, v1;
    if (MULTI) {
        double o[2];
        pd_ms_avg_at<2>(*ms, true, b, 2 * g.k, iy, ix, s_up, o);
        v0 = o[0];
        v1 = o[1];
    } else if (a.src.kind == PD_SRC_LOWRES) {
        float f0, f1;
        if (a.pairs)
            bicubic2_planar(a.pairs + ((size_t)b * PD_N_LIMBS + g.k) * a.src.h * a.src.w * 2, a.src.h, a.src.w, iy, ix,
                            s_up, &f0, &f1);
        else
            bicubic2_at(a.src.lo + (size_t)b * a.src.h * a.src.w * a.src.C, a.src.h, a.src.w, a.src.C, 2 * g.k, iy, ix,
                        s_up, &f0, &f1);
        v0 = (double)f0;
        v1 = (double)f1;
    } else {
        const size_t npx = (size_t)H * W;
        const double* pl = a.src.full + ((size_t)b * a.src.Cfull + 2 * g.k) * npx + (size_t)iy * W + ix;
        v0 = pl[0];
        v1 = pl[npx];
    }
    return v0 * g.u0 + v1 * g.u1;
}

// Scores up to six pairs whose geometry (k < 0: nothing to do) and pair indices lie in
// geo[0..5] / p6[0..5]: ten lanes per pair take one sample point each, and lane 0 of each
// group adds the ten products in sample order.
template <bool MULTI>
__device__ __forceinline__ void score_six(const PairArgs& a, const MultiSource* ms, int b, int lane, const float4* s_up,
                                          const PairGeo* geo, const int* p6, int n, double (*mid6)[10]) {
    const int cs = lane / 10, i = lane - cs * 10;
    double mid = 0.0;
    int k = -1;
    if (cs < n) k = geo[cs].k;
    if (k >= 0) {
        const PairGeo g = geo[cs];
        mid = pair_sample<MULTI>(a, ms, b, g, i, s_up);
    }
    if (cs < 6) mid6[cs][i] = mid;
    pd_wave_sync();  // lane 0 of a group reads what the other nine wrote
    if (k >= 0 && i == 0) {
        // sequential sum over the ten samples, as builtin sum() does
        const double* m = mid6[cs];
        double acc = 0.0 + m[0];
        int n_above = m[0] > a.thr;
#pragma unroll
        for (int s2 = 1; s2 < PD_MID; ++s2) {
            acc = acc + m[s2];
            n_above += m[s2] > a.thr;
        }
        const double score = acc / 10.0 + geo[cs].prior;
        a.pair_score[(size_t)b * a.max_pairs + p6[cs]] = (n_above > 8 && score > 0.0) ? score : -1.0;
    }
    pd_wave_sync();  // mid6 is overwritten by the next six pairs
}

// Every wavefront works on its own.  The kernel is bound by the latency of the sample
// gathers (every tap of every sample is a cache line of its own in the (h, w, 38)
// layout), so a wavefront takes few pairs and many wavefronts are in flight.
// Images with few candidate pairs (every wavefront gets at most two chunks of six): the
// chunks are scored directly.  Images with many: most candidates join unrelated peaks
// and fail at any sample between them, and a pair with two samples at or under the
// threshold is rejected whatever the others say -- so one lane per pair first takes
// samples 4 and 5, one lane per survivor then samples 2 and 7, and only the pairs with at
// most one of the four at or under the threshold are scored in full.
// The grid is one-dimensional and swizzled so that the workgroups of one image run on
// one XCD (workgroup i goes to XCD i % 8) and its PAF map stays in that XCD's L2.
// PRE = false: the kernel without the pre-test (every image takes the direct form, whatever its
// number of pairs: same results), a fifth of the LDS and fewer registers; the host picks it for
// decodes whose images hold few peaks.
template <bool PRE, bool MULTI>
__device__ __forceinline__ void pair_scores_body(const PairArgs& a, const MultiSource* ms) {
    __shared__ int32_t s_off[PD_N_LIMBS + 1];
    __shared__ int32_t s_cnt[PD_N_PARTS];
    __shared__ float4 s_up[8];
    __shared__ PairGeo s_geo[4][PRE ? PRE_PAIRS : 8];  // a wavefront's staged pairs: six at a time, or the pre-test's survivors
    __shared__ int s_p[4][PRE ? PRE_PAIRS : 8];
    __shared__ double s_mid[4][6][10];
    const int tid = threadIdx.x;
    const int per_xcd = gridDim.x / 8;  // the grid is a multiple of 8 workgroups
    const int logical = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    const int b = logical / a.per_image, blk = logical - b * a.per_image;
    if (tid >= 64 && tid < 72) {
        const float* c = a.src.up8.c[tid - 64];
        s_up[tid - 64] = make_float4(c[0], c[1], c[2], c[3]);
    }
    if (tid <= PD_N_LIMBS) s_off[tid] = a.pair_off[b * (PD_N_LIMBS + 1) + tid];
    if (tid < PD_N_PARTS) s_cnt[tid] = a.part_count[b * PD_N_PARTS + tid];
    __syncthreads();
    const int total = s_off[PD_N_LIMBS];
    if (total < 0) {  // more candidate pairs than the context was sized for
        if (tid == 0 && blk == 0) a.status[b] = PD_E_OVERFLOW;
        return;
    }
    const int lane = tid & 63, wv = tid >> 6;
    const int n_waves = a.per_image * 4;
    const int wave_id = blk * 4 + wv;
    PairGeo* geo = s_geo[wv];
    int* pq = s_p[wv];
    if (!PRE || total <= 2 * n_waves * PAIRS_PER_CHUNK) {
        for (int base = wave_id * PAIRS_PER_CHUNK; base < total; base += n_waves * PAIRS_PER_CHUNK) {
            if (lane < PAIRS_PER_CHUNK) {
                PairGeo g;
                g.k = -1;
                const int p = base + lane;
                if (p < total && !pair_geometry(a, b, p, s_off, s_cnt, g)) a.pair_score[(size_t)b * a.max_pairs + p] = -1.0;
                geo[lane] = g;
                pq[lane] = p;
            }
            pd_wave_sync();  // the staging area is private to this wavefront; all lanes read what six wrote
            score_six<MULTI>(a, ms, b, lane, s_up, geo, pq, min(PAIRS_PER_CHUNK, total - base), s_mid[wv]);
        }
        return;
    }
    for (int base = wave_id * PRE_PAIRS; base < total; base += n_waves * PRE_PAIRS) {
        // the geometry of this pass's pairs, one lane per pair, into the wavefront's list
        int nq = 0;
#pragma unroll 1
        for (int r = 0; r < PRE_PAIRS / 64; ++r) {
            const int p = base + r * 64 + lane;
            bool ok = false;
            PairGeo g;
            g.k = -1;
            if (p < total) {
                ok = pair_geometry(a, b, p, s_off, s_cnt, g);
                if (!ok) a.pair_score[(size_t)b * a.max_pairs + p] = -1.0;
            }
            const unsigned long long m = __ballot(ok);
            if (ok) {
                const int at = nq + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                geo[at] = g;
                pq[at] = p;
            }
            nq += __builtin_popcountll(m);
        }
        pd_wave_sync();
        // two rounds, one lane per listed pair: samples 4 and 5, then 2 and 7 of the survivors.
        // A pair with two samples at or under the threshold is out; the list is compacted in
        // place (a step writes only entries that it or an earlier step has read).
#pragma unroll 1
        for (int round = 0; round < 2; ++round) {
            const int ia = round == 0 ? 4 : 2, ib = round == 0 ? 5 : 7;
            int kept = 0;
#pragma unroll 1
            for (int q0 = 0; q0 < nq; q0 += 64) {
                const int idx = q0 + lane;
                bool survives = false;
                PairGeo g;
                int p = 0;
                if (idx < nq) {
                    g = geo[idx];
                    p = pq[idx];
                    const double ma = pair_sample<MULTI>(a, ms, b, g, ia, s_up), mb = pair_sample<MULTI>(a, ms, b, g, ib, s_up);
                    g.pad += (ma > a.thr ? 0 : 1) + (mb > a.thr ? 0 : 1);  // samples at or under the threshold so far
                    survives = g.pad < 2;
                    if (!survives) a.pair_score[(size_t)b * a.max_pairs + p] = -1.0;
                }
                pd_wave_sync();
                const unsigned long long m = __ballot(survives);
                if (survives) {
                    const int at = kept + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                    geo[at] = g;
                    pq[at] = p;
                }
                kept += __builtin_popcountll(m);
                pd_wave_sync();
            }
            nq = kept;
        }
        for (int q0 = 0; q0 < nq; q0 += PAIRS_PER_CHUNK)
            score_six<MULTI>(a, ms, b, lane, s_up, geo + q0, pq + q0, min(PAIRS_PER_CHUNK, nq - q0), s_mid[wv]);
    }
}

template <bool PRE>
__global__ void __launch_bounds__(256)
k_pair_scores(PairArgs a) {
    PD_RAISE_PRIORITY(3);
    pair_scores_body<PRE, false>(a, nullptr);
}

// The same on a multi-scale source: every PAF value is the average over the test scales of the
// two-step bicubic resize, evaluated at the sample from the network outputs themselves (a few
// thousand samples per image instead of 38 full-resolution planes).
template <bool PRE>
__global__ void __launch_bounds__(256)
k_pair_scores_ms(PairArgs a, MultiSource ms) {
    pair_scores_body<PRE, true>(a, &ms);
}

// ---------------------------------------------------------------------------------
// Matching: one wavefront per (image, limb).
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_match(MatchArgs a) {
    PD_RAISE_PRIORITY(1);
    extern __shared__ __align__(16) unsigned char dyn[];
    match_limb(a, blockIdx.x, blockIdx.y, threadIdx.x, dyn);
}

// (B, h, w, 38) -> (B, 19, h, w, 2): the two channels of a limb side by side, pixels of a
// row contiguous, so that the four horizontal taps of a sample share a cache line instead
// of sitting 152 bytes apart.  Through LDS so that both sides are coalesced.
#define PP_PIX 128
__global__ void __launch_bounds__(256)
k_paf_pairs(const float* __restrict__ lo, int npix, float* __restrict__ out) {
    __shared__ __align__(16) float tile[PP_PIX * PD_CH_PAF];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int p0 = blockIdx.x * PP_PIX;
    const int np = min(PP_PIX, npix - p0);
    const float* src = lo + ((size_t)b * npix + p0) * PD_CH_PAF;
    const int n = np * PD_CH_PAF;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {  // 16 bytes per load: four times the bytes in flight per thread
        const int n4 = n >> 2;
        for (int i = tid; i < n4; i += 256) ((float4*)tile)[i] = ((const float4*)src)[i];
        for (int i = 4 * n4 + tid; i < n; i += 256) tile[i] = src[i];
    } else {
        for (int i = tid; i < n; i += 256) tile[i] = src[i];
    }
    __syncthreads();
    float2* dst = (float2*)out + (size_t)b * PD_N_LIMBS * npix + p0;
    for (int k = 0; k < PD_N_LIMBS; ++k)
        for (int i = tid; i < np; i += 256)
            dst[(size_t)k * npix + i] = make_float2(tile[i * PD_CH_PAF + 2 * k], tile[i * PD_CH_PAF + 2 * k + 1]);
}

int pd_launch_pairs(pd_ctx* ctx, double thre2) {
    PairArgs a;
    a.src = pd_paf_source(ctx);
    a.mode = ctx->mode;
    a.thr = thre2;
    a.pk_y = ctx->pk_y;
    a.pk_x = ctx->pk_x;
    a.part_count = ctx->part_count;
    a.pair_off = ctx->pair_off;
    a.PK = ctx->cfg.max_peaks_per_part;
    a.max_pairs = ctx->cfg.max_pairs_per_image;
    a.pair_score = ctx->pair_score;
    a.status = ctx->status;
    // 4 wavefronts per workgroup, 60 pairs per wavefront pass
    int per_image = (ctx->cfg.max_pairs_per_image + 4 * PAIRS_PER_CHUNK - 1) / (4 * PAIRS_PER_CHUNK);
    if (per_image > 32) per_image = 32;  // more only adds idle workgroups (measured: 64 costs 2 % at 368^2, nothing at 656^2)
    per_image = (per_image + 7) / 8 * 8;  // the XCD swizzle needs a grid that is a multiple of 8
    a.per_image = per_image;
    // Maps that fit the caches are sampled where they lie; beyond that every tap of every
    // sample would come from memory as a line of its own, and the pair-planar copy pays.
    // "Fit": the batch within the last-level cache, or one image's maps small enough that the L2 of
    // the XCD its workgroups share keeps them while they are sampled (batch 256 at 368^2, 321 KB per
    // image, 82 MB in all: 1.64 M images/s without the copy, 1.55 M with; at 656^2, 1 MB per image:
    // 206 k without, 342 k with).
    a.pairs = nullptr;
    const size_t paf_bytes = sizeof(float) * (size_t)ctx->B * ctx->h * ctx->w * PD_CH_PAF;
    const size_t image_bytes = sizeof(float) * (size_t)ctx->h * ctx->w * PD_CH_PAF;
    static const int force_copy = [] {   // comparison runs: PD_PAF_PAIRS=0 never makes the copy, 1 always
        const char* v = getenv("PD_PAF_PAIRS");
        return v && *v ? atoi(v) : -1;
    }();
    const bool want_copy = force_copy >= 0 ? force_copy != 0 : paf_bytes > ((size_t)64 << 20) && image_bytes > ((size_t)512 << 10);
    if (ctx->src_kind == PD_SRC_LOWRES && want_copy) {
        if (ctx->paf_pairs_cap < paf_bytes) {
            PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->gen++;
            if (ctx->paf_pairs) PD_HIP(ctx, hipFree(ctx->paf_pairs));
            ctx->paf_pairs = nullptr;
            ctx->paf_pairs_cap = 0;
            PD_HIP(ctx, hipMalloc(&ctx->paf_pairs, paf_bytes));
            ctx->paf_pairs_cap = paf_bytes;
        }
        a.pairs = ctx->paf_pairs;
    }
    if (a.src.kind == PD_SRC_MULTI) {
        // (workgroups per image, measured at batch 128, 1-8 people: 8 -> 0.205 ms per launch, 16 -> 0.163,
        // 32 -> 0.149, 64 -> 0.148, 128 -> 0.154: a sample is four dependent rounds of loads, so many
        // short wavefronts beat few long ones; PD_MS_PAIR_BLOCKS sets it for comparison runs)
        static const int ms_per_image = getenv("PD_MS_PAIR_BLOCKS") ? atoi(getenv("PD_MS_PAIR_BLOCKS")) : 0;
        if (ms_per_image >= 8 && ms_per_image % 8 == 0 && ms_per_image <= 256) a.per_image = per_image = ms_per_image;
        const MultiSource ms = pd_multi_source(ctx);
        pd_prof_begin(ctx, PD_STAGE_PAIRS);
        if (ctx->pairs_pre)
            hipLaunchKernelGGL(k_pair_scores_ms<true>, dim3(per_image * ctx->B), dim3(256), 0, ctx->stream, a, ms);
        else
            hipLaunchKernelGGL(k_pair_scores_ms<false>, dim3(per_image * ctx->B), dim3(256), 0, ctx->stream, a, ms);
        pd_prof_end(ctx);
        PD_HIP(ctx, hipGetLastError());
        return PD_OK;
    }
    pd_prof_begin(ctx, PD_STAGE_PAIRS);
    if (a.pairs)
        hipLaunchKernelGGL(k_paf_pairs, dim3((ctx->h * ctx->w + PP_PIX - 1) / PP_PIX, ctx->B), dim3(256), 0, ctx->stream,
                           ctx->lo_paf, ctx->h * ctx->w, ctx->paf_pairs);
    if (ctx->pairs_pre)
        hipLaunchKernelGGL(k_pair_scores<true>, dim3(per_image * ctx->B), dim3(256), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(k_pair_scores<false>, dim3(per_image * ctx->B), dim3(256), 0, ctx->stream, a);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    return PD_OK;
}

// (tests set this to force the arg-max form)
extern "C" int pd_match_cap_override;
int pd_match_cap_override = 0;

int pd_launch_match(pd_ctx* ctx) {
    MatchArgs a;
    a.PK = ctx->cfg.max_peaks_per_part;
    a.max_pairs = ctx->cfg.max_pairs_per_image;
    // Surviving pairs of a limb that the wavefront's sort list holds (more: the arg-max form, same
    // result).  512 covers 22 x 22 peaks with every pair surviving; the list is 12 bytes per entry
    // of a one-wavefront workgroup's LDS, and with 2048 entries (25 KB) six such workgroups filled
    // a CU's LDS and kept the other contexts' kernels out (1.15 -> 1.18 M images/s at 368^2,
    // 298 -> 313 k at 656^2 with 30 people).
    a.cap = pd_match_cap_override > 0 ? pd_match_cap_override : 512;
    a.pair_score = ctx->pair_score;
    a.pair_off = ctx->pair_off;
    a.part_count = ctx->part_count;
    a.pk_id = ctx->pk_id;
    a.pk_score = ctx->pk_score;
    a.conn = ctx->conn;
    a.conn_conf = ctx->conn_conf;
    a.limb_count = ctx->limb_count;
    const size_t lds = pd_match_lds(a.cap, a.PK);
    pd_prof_begin(ctx, PD_STAGE_MATCH);
    hipLaunchKernelGGL(k_match, dim3(PD_N_LIMBS, ctx->B), dim3(64), lds, ctx->stream, a);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    return PD_OK;
}
