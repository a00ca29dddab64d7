// The multi-scale average of predict.py:49-72 evaluated at single pixels, straight from the
// network outputs of the test scales -- nothing at full resolution is stored.
//
//   avg[Y][X] = ((0 + v_0 / n) + v_1 / n) + ..        float32 division, float64 sum, scale order
//   v_m       = resize(crop(resize_x8(src_m)), (H, W))[Y][X]        cv2.resize INTER_CUBIC twice
//
// Arithmetic of both resizes as in resize.hip: horizontal 4-tap sum, then vertical, each
// ((s0*c0 + s1*c1) + s2*c2) + s3*c3 in float32 with separately rounded operations, taps clamped
// to the image, same size = copy.
#pragma once
#include "common.h"

#pragma clang fp contract(off)

#define PD_SRC_MULTI 3  // MapSource::kind: the maps are the average over the scales of a MultiSource

struct MultiSrc {
    const float* heat;  // (B, hs, ws, 19) float32 network output of this scale
    const float* paf;   // (B, hs, ws, 38)
    int hs, ws, crop_h, crop_w, identity;  // identity: the crop of the x8 map already has the size (H, W)
    CubicTable ty, tx;  // second resize crop -> (H, W); unused when identity
};

struct MultiSource {
    MultiSrc s[PD_MAX_SCALES];
    int n, pow2, H, W;
    float n_div, n_rcp;  // number of scales; its reciprocal, exact when n is a power of two
};

template <int NCH>
__device__ __forceinline__ void pd_ms_load(const float* p, float (&o)[NCH]) {
    if (NCH == 2) {  // an even channel of an even channel count: 8-byte aligned
        const float2 t = *(const float2*)p;
        o[0] = t.x;
        o[NCH - 1] = t.y;
    } else {
#pragma unroll
        for (int c = 0; c < NCH; ++c) o[c] = p[c];
    }
}

// x8 map of a (hs, ws, C) source at (row, col), NCH adjacent channels starting at img.
template <int NCH>
__device__ __forceinline__ void pd_ms_up8_at(const float* __restrict__ img, int hs, int ws, int C, int row, int col,
                                             const float4* __restrict__ up, float (&out)[NCH]) {
    const int x0 = pd_up8_ofs(col), y0 = pd_up8_ofs(row);
    const float4 ac = up[pd_up8_phase(col)], bc = up[pd_up8_phase(row)];
    const int xi0 = pd_clampi(x0, 0, ws - 1) * C, xi1 = pd_clampi(x0 + 1, 0, ws - 1) * C, xi2 = pd_clampi(x0 + 2, 0, ws - 1) * C,
              xi3 = pd_clampi(x0 + 3, 0, ws - 1) * C;
    float hv[4][NCH];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* r = img + (size_t)pd_clampi(y0 + k, 0, hs - 1) * ws * C;
        float s0[NCH], s1[NCH], s2[NCH], s3[NCH];
        pd_ms_load<NCH>(r + xi0, s0);
        pd_ms_load<NCH>(r + xi1, s1);
        pd_ms_load<NCH>(r + xi2, s2);
        pd_ms_load<NCH>(r + xi3, s3);
#pragma unroll
        for (int c = 0; c < NCH; ++c) hv[k][c] = ((s0[c] * ac.x + s1[c] * ac.y) + s2[c] * ac.z) + s3[c] * ac.w;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) out[c] = ((hv[0][c] * bc.x + hv[1][c] * bc.y) + hv[2][c] * bc.z) + hv[3][c] * bc.w;
}

// One scale's resized map at (Y, X): the 4 x 4 taps of the second resize are x8-map values at
// four (clamped) consecutive rows and columns, which together draw on at most 5 x 5 source
// pixels; those are loaded once, the horizontal sums of the x8 resize are formed once per
// (source row, x8 column), and every x8 value takes its four from them.
template <int NCH>
__device__ __forceinline__ void pd_ms_scale_at(const MultiSrc& sc, const float* __restrict__ img, int C, int Y, int X,
                                               const float4* __restrict__ up, float (&out)[NCH]) {
    if (sc.identity) {
        pd_ms_up8_at<NCH>(img, sc.hs, sc.ws, C, Y, X, up, out);
        return;
    }
    const int x0 = sc.tx.ofs[X], y0 = sc.ty.ofs[Y];
    const float4 ac = *(const float4*)(sc.tx.coef + 4 * (size_t)X);
    const float4 bc = *(const float4*)(sc.ty.coef + 4 * (size_t)Y);
    int r[4], c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r[k] = pd_clampi(y0 + k, 0, sc.crop_h - 1);
        c[k] = pd_clampi(x0 + k, 0, sc.crop_w - 1);
    }
    // first source row / column any tap draws on (unclamped); r[3] - r[0] <= 3, so the first
    // source row of a later tap row is this one or the next
    const int j_lo = pd_up8_ofs(r[0]), i_lo = pd_up8_ofs(c[0]);
    float px[5][5][NCH];
#pragma unroll
    for (int jj = 0; jj < 5; ++jj) {
        const float* row = img + (size_t)pd_clampi(j_lo + jj, 0, sc.hs - 1) * sc.ws * C;
#pragma unroll
        for (int ii = 0; ii < 5; ++ii) pd_ms_load<NCH>(row + (size_t)pd_clampi(i_lo + ii, 0, sc.ws - 1) * C, px[jj][ii]);
    }
    float hs[5][4][NCH];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool d = pd_up8_ofs(c[t]) != i_lo;
        const float4 w = up[pd_up8_phase(c[t])];
#pragma unroll
        for (int jj = 0; jj < 5; ++jj)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const float s0 = d ? px[jj][1][ch] : px[jj][0][ch], s1 = d ? px[jj][2][ch] : px[jj][1][ch],
                            s2 = d ? px[jj][3][ch] : px[jj][2][ch], s3 = d ? px[jj][4][ch] : px[jj][3][ch];
                hs[jj][t][ch] = ((s0 * w.x + s1 * w.y) + s2 * w.z) + s3 * w.w;
            }
    }
    float hv[4][NCH];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool e = pd_up8_ofs(r[k]) != j_lo;
        const float4 w = up[pd_up8_phase(r[k])];
        float u[4][NCH];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const float h0 = e ? hs[1][t][ch] : hs[0][t][ch], h1 = e ? hs[2][t][ch] : hs[1][t][ch],
                            h2 = e ? hs[3][t][ch] : hs[2][t][ch], h3 = e ? hs[4][t][ch] : hs[3][t][ch];
                u[t][ch] = ((h0 * w.x + h1 * w.y) + h2 * w.z) + h3 * w.w;
            }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) hv[k][ch] = ((u[0][ch] * ac.x + u[1][ch] * ac.y) + u[2][ch] * ac.z) + u[3][ch] * ac.w;
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) out[ch] = ((hv[0][ch] * bc.x + hv[1][ch] * bc.y) + hv[2][ch] * bc.z) + hv[3][ch] * bc.w;
}

// The average over the scales at (Y, X): channels ch .. ch+NCH-1 of the heatmaps (paf = false)
// or of the PAFs of image b.
template <int NCH>
__device__ __forceinline__ void pd_ms_avg_at(const MultiSource& ms, bool paf, int b, int ch, int Y, int X,
                                             const float4* __restrict__ up, double (&acc)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = 0.0;
    const int C = paf ? PD_CH_PAF : PD_CH_HEAT;
#pragma unroll 1
    for (int m = 0; m < ms.n; ++m) {
        const MultiSrc& sc = ms.s[m];
        const float* img = (paf ? sc.paf : sc.heat) + (size_t)b * sc.hs * sc.ws * C + ch;
        float v[NCH];
        pd_ms_scale_at<NCH>(sc, img, C, Y, X, up, v);
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = acc[c] + (double)(ms.pow2 ? v[c] * ms.n_rcp : v[c] / ms.n_div);
    }
}

struct pd_ctx;
// The MultiSource of the scales a context holds (pd_set_multiscale).
MultiSource pd_multi_source(const pd_ctx* ctx);
