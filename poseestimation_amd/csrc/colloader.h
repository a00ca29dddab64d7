// Gaussian weights and the per-column readers of the unsmoothed full-resolution
// heatmap, shared by the streaming kernel and the candidate verifier.
#pragma once
#include "common.h"

#pragma clang fp contract(off)

// scipy _gaussian_kernel1d(sigma=3, order=0, radius=12): weights of distance
// 12, 11, .. 1, 0 (index 12 = centre).
#define PD_GAUSS_WEIGHTS                                                                              \
    {0x1.763a210dfb306p-15, 0x1.4fbe39149e277p-13, 0x1.0d8a5ad43c165p-11, 0x1.8345966f69518p-10,     \
     0x1.f1e9915139406p-9,  0x1.1e6bccad344bap-7,  0x1.26defcaeb0202p-6,  0x1.0fa58939b528fp-5,      \
     0x1.bfde9c12bec92p-5,  0x1.4a614d1afd337p-4,  0x1.b42a57d56c0bep-4,  0x1.01a25f86eb137p-3,      \
     0x1.105a329f98197p-3}
static __constant__ double c_gw[PD_GTAPS] = PD_GAUSS_WEIGHTS;

template <int SRC>
struct ColLoader;

// Upsampled-on-the-fly column of a (h, w, C) float32 network output: the horizontal
// 4-tap sums of the four source rows in use are cached and shifted as the row
// advances; arithmetic identical to pd_bicubic_at.
template <>
struct ColLoader<PD_SRC_LOWRES> {
    const float* img;  // image b, channel offset applied
    int h, w, C;
    const int32_t* yofs;
    const float* ycoef;
    int xi0, xi1, xi2, xi3;
    float a0, a1, a2, a3;
    int cur;  // first tap row the cached horizontal sums belong to
    float hv[4];

    __device__ __forceinline__ float hrow(int s) const {
        s = pd_clampi(s, 0, h - 1);
        const float* row = img + ((size_t)s * w) * C;
        return ((row[(size_t)xi0 * C] * a0 + row[(size_t)xi1 * C] * a1) + row[(size_t)xi2 * C] * a2) +
               row[(size_t)xi3 * C] * a3;
    }
    __device__ __forceinline__ void init(const MapSource& s, int b, int ch, int col) {
        img = s.lo + (size_t)b * s.h * s.w * s.C + ch;
        h = s.h;
        w = s.w;
        C = s.C;
        yofs = s.ty.ofs;
        ycoef = s.ty.coef;
        const int x0 = s.tx.ofs[col];
        xi0 = pd_clampi(x0, 0, w - 1);
        xi1 = pd_clampi(x0 + 1, 0, w - 1);
        xi2 = pd_clampi(x0 + 2, 0, w - 1);
        xi3 = pd_clampi(x0 + 3, 0, w - 1);
        a0 = s.tx.coef[4 * col + 0];
        a1 = s.tx.coef[4 * col + 1];
        a2 = s.tx.coef[4 * col + 2];
        a3 = s.tx.coef[4 * col + 3];
        cur = -1000000;
    }
    __device__ __forceinline__ double load(int r) {
        const int y0 = yofs[r];
        if (y0 == cur + 1) {
            hv[0] = hv[1];
            hv[1] = hv[2];
            hv[2] = hv[3];
            hv[3] = hrow(y0 + 3);
        } else if (y0 != cur) {
            hv[0] = hrow(y0);
            hv[1] = hrow(y0 + 1);
            hv[2] = hrow(y0 + 2);
            hv[3] = hrow(y0 + 3);
        }
        cur = y0;
        const float v = ((hv[0] * ycoef[4 * r + 0] + hv[1] * ycoef[4 * r + 1]) + hv[2] * ycoef[4 * r + 2]) +
                        hv[3] * ycoef[4 * r + 3];
        return (double)v;
    }
};

// Column of a float64 plane.
template <>
struct ColLoader<PD_SRC_FULLRES> {
    const double* col;
    int W;
    __device__ __forceinline__ void init(const MapSource& s, int b, int ch, int c) {
        col = s.full + ((size_t)b * s.Cfull + ch) * (size_t)s.H * s.W + c;
        W = s.W;
    }
    __device__ __forceinline__ double load(int r) const { return col[(size_t)r * W]; }
};
