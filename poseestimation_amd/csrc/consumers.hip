// Entry points either side of the decode: the step before it (image resize + pad of the
// multi-scale loop, predict.py:56-57) and the two consumers of its results (COCO keypoint
// rows, evaluate.py:44-74; skeleton rendering, predict.py:256-307).
#include <math.h>

#include <vector>

#include "common.h"

#pragma clang fp contract(off)

#define PD_CHECK_CTX(ctx) \
    if ((ctx) == nullptr) return PD_E_INVALID;

// Scratch of these entry points: one device buffer per context, grown on demand.
static int aux_ensure(pd_ctx* ctx, size_t bytes) {
    if (ctx->aux_cap >= bytes && ctx->aux_dev) return PD_OK;
    PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->aux_dev) PD_HIP(ctx, hipFree(ctx->aux_dev));
    ctx->aux_dev = nullptr;
    ctx->aux_cap = 0;
    PD_HIP(ctx, hipMalloc((void**)&ctx->aux_dev, bytes));
    ctx->aux_cap = bytes;
    return PD_OK;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// ---------------------------------------------------------------------------------
// Person rows -> COCO keypoint rows (evaluate.py:44-74)
// ---------------------------------------------------------------------------------
// cfg.coco_to_ours, cfgs/config_vgg19.py:48
static __constant__ int c_coco_to_ours[PD_COCO_KEYPOINTS] = {0, 15, 14, 17, 16, 5, 2, 6, 3, 7, 4, 11, 8, 12, 9, 13, 10};

struct CocoArgs {
    int B, PK, max_persons, mode;
    const int32_t *part_base, *person_count;
    const int32_t *pk_y, *pk_x;
    const double* persons;
    double* rows;           // [n_rows][17][3]
    double* scores;         // [n_rows]
    int32_t* image_of_row;  // [n_rows]
    long long* total;
};

// One workgroup per image; a thread per (person, keypoint).  `candidate[peak_idx]` of
// evaluate.py:55 indexes the flattened peak list of the image (predict.py:179), whose first two
// columns are (x, y) in mode D and (y, x) in mode P.
__global__ void __launch_bounds__(256)
k_coco_rows(CocoArgs a) {
    __shared__ long long s_red[4];
    __shared__ int s_pbase[PD_N_PARTS + 1];
    const int b = blockIdx.x, tid = threadIdx.x;
    long long before = 0;
    for (int j = tid; j < b; j += 256) before += a.person_count[j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d);
    if ((tid & 63) == 0) s_red[tid >> 6] = before;
    if (tid <= PD_N_PARTS) s_pbase[tid] = a.part_base[b * (PD_N_PARTS + 1) + tid];
    __syncthreads();
    const long long base = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    const int n = a.person_count[b];
    if (b == a.B - 1 && tid == 0) *a.total = base + n;
    const int n_peaks = s_pbase[PD_N_PARTS];
    for (int t = tid; t < n * (PD_COCO_KEYPOINTS + 1); t += 256) {
        const int p = t / (PD_COCO_KEYPOINTS + 1), k = t - p * (PD_COCO_KEYPOINTS + 1);
        const double* person = a.persons + ((size_t)b * a.max_persons + p) * 20;
        const long long row = base + p;
        if (k == PD_COCO_KEYPOINTS) {
            a.scores[row] = person[18];       // evaluate.py:45 person[-2]
            a.image_of_row[row] = b;
            continue;
        }
        const int id = (int)person[c_coco_to_ours[k]];
        double c0 = 0.0, c1 = 0.0, v = 0.0;   // a missing part stays (0, 0, 0), evaluate.py:53-54
        if (id >= 0 && id < n_peaks) {
            int q = 0;
            while (id >= s_pbase[q + 1]) ++q;
            const size_t o = ((size_t)b * PD_N_PARTS + q) * a.PK + (id - s_pbase[q]);
            const double y = (double)a.pk_y[o], x = (double)a.pk_x[o];
            c0 = a.mode == PD_MODE_D ? x : y;
            c1 = a.mode == PD_MODE_D ? y : x;
            v = 1.0;
        }
        double* out = a.rows + (row * PD_COCO_KEYPOINTS + k) * 3;
        out[0] = c0;
        out[1] = c1;
        out[2] = v;
    }
}

extern "C" int pd_coco_rows(pd_ctx* ctx, double* keypoints, double* scores, int32_t* image_index, int64_t capacity,
                            int64_t* n_rows) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    if (!ctx->have_peaks || !ctx->have_persons)
        PD_FAIL(ctx, PD_E_STATE, "pd_coco_rows: needs the peaks and the person rows of a decode");
    if (!n_rows || capacity < 0) PD_FAIL(ctx, PD_E_INVALID, "pd_coco_rows: n_rows is required");
    const size_t rows_max = (size_t)ctx->B * ctx->cfg.max_persons;
    const size_t off_scores = align256(rows_max * PD_COCO_KEYPOINTS * 3 * sizeof(double));
    const size_t off_image = off_scores + align256(rows_max * sizeof(double));
    const size_t off_total = off_image + align256(rows_max * sizeof(int32_t));
    int rc = aux_ensure(ctx, off_total + 256);
    if (rc != PD_OK) return rc;
    CocoArgs a;
    a.B = ctx->B;
    a.PK = ctx->cfg.max_peaks_per_part;
    a.max_persons = ctx->cfg.max_persons;
    a.mode = ctx->mode;
    a.part_base = ctx->part_base;
    a.person_count = ctx->person_count;
    a.pk_y = ctx->pk_y;
    a.pk_x = ctx->pk_x;
    a.persons = ctx->persons;
    a.rows = (double*)ctx->aux_dev;
    a.scores = (double*)(ctx->aux_dev + off_scores);
    a.image_of_row = (int32_t*)(ctx->aux_dev + off_image);
    a.total = (long long*)(ctx->aux_dev + off_total);
    hipLaunchKernelGGL(k_coco_rows, dim3(ctx->B), dim3(256), 0, ctx->stream, a);
    PD_HIP(ctx, hipGetLastError());
    long long total = 0;
    PD_HIP(ctx, hipMemcpyAsync(&total, a.total, sizeof(total), hipMemcpyDeviceToHost, ctx->stream));
    PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_rows = total;
    if (total > capacity)
        PD_FAIL(ctx, PD_E_OVERFLOW, "pd_coco_rows: %lld person rows, room for %lld", total, (long long)capacity);
    if (total == 0) return PD_OK;
    if (keypoints)
        PD_HIP(ctx, hipMemcpy(keypoints, a.rows, (size_t)total * PD_COCO_KEYPOINTS * 3 * sizeof(double), hipMemcpyDeviceToHost));
    if (scores) PD_HIP(ctx, hipMemcpy(scores, a.scores, (size_t)total * sizeof(double), hipMemcpyDeviceToHost));
    if (image_index)
        PD_HIP(ctx, hipMemcpy(image_index, a.image_of_row, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost));
    return PD_OK;
}

// ---------------------------------------------------------------------------------
// Image side of the scale loop (predict.py:56-57): cubic resize by (fx, fy) = (scale, scale),
// then padding below / right to a multiple of the stride with the pad value.
// ---------------------------------------------------------------------------------
struct ImageArgs {
    const void* src;  // (h, w, C) uint8 or float32
    int src_u8;
    int h, w, C;
    int rh, rw;       // resized size
    int oh, ow;       // padded size
    int identity;     // rh == h && rw == w: cv2.resize copies
    CubicTable ty, tx;
    float pad;
    void* dst;        // (oh, ow, C) float32 or uint8
    int dst_u8;
};

__device__ __forceinline__ float image_at(const ImageArgs& a, int y, int x, int c) {
    const size_t o = ((size_t)y * a.w + x) * a.C + c;
    return a.src_u8 ? (float)((const unsigned char*)a.src)[o] : ((const float*)a.src)[o];
}

// uint8 output: round half to even and saturate, as OpenCV's saturate_cast<uchar>(float) does
__device__ __forceinline__ void image_store(const ImageArgs& a, size_t o, float v) {
    if (a.dst_u8) {
        const float r = rintf(v);
        ((unsigned char*)a.dst)[o] = (unsigned char)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
    } else {
        ((float*)a.dst)[o] = v;
    }
}

__global__ void __launch_bounds__(256)
k_image_resize_pad(ImageArgs a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)a.oh * a.ow) return;
    const int Y = (int)(i / a.ow), X = (int)(i - (size_t)Y * a.ow);
    const size_t o = i * a.C;
    if (Y >= a.rh || X >= a.rw) {
        for (int c = 0; c < a.C; ++c) image_store(a, o + c, a.pad);
        return;
    }
    if (a.identity) {
        for (int c = 0; c < a.C; ++c) image_store(a, o + c, image_at(a, Y, X, c));
        return;
    }
    const int x0 = a.tx.ofs[X], y0 = a.ty.ofs[Y];
    const float4 ac = *(const float4*)(a.tx.coef + 4 * (size_t)X);
    const float4 bc = *(const float4*)(a.ty.coef + 4 * (size_t)Y);
    const int xi0 = pd_clampi(x0, 0, a.w - 1), xi1 = pd_clampi(x0 + 1, 0, a.w - 1), xi2 = pd_clampi(x0 + 2, 0, a.w - 1),
              xi3 = pd_clampi(x0 + 3, 0, a.w - 1);
    for (int c = 0; c < a.C; ++c) {
        float hv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int s = pd_clampi(y0 + k, 0, a.h - 1);
            hv[k] = ((image_at(a, s, xi0, c) * ac.x + image_at(a, s, xi1, c) * ac.y) + image_at(a, s, xi2, c) * ac.z) +
                    image_at(a, s, xi3, c) * ac.w;
        }
        image_store(a, o + c, ((hv[0] * bc.x + hv[1] * bc.y) + hv[2] * bc.z) + hv[3] * bc.w);
    }
}

// cv2.resize(img, (0, 0), fx=s, fy=s): the destination size is the product rounded half to even.
static int resized_len(int n, double scale) { return (int)nearbyint((double)n * scale); }

extern "C" int pd_image_resize_shape(int h, int w, double scale, int stride, int32_t* resized_h, int32_t* resized_w,
                                     int32_t* pad_down, int32_t* pad_right) {
    if (h < 1 || w < 1 || !(scale > 0.0) || stride < 1) return PD_E_INVALID;
    const int rh = resized_len(h, scale), rw = resized_len(w, scale);
    if (rh < 1 || rw < 1) return PD_E_INVALID;
    if (resized_h) *resized_h = rh;
    if (resized_w) *resized_w = rw;
    if (pad_down) *pad_down = (stride - rh % stride) % stride;     // predict.py:23
    if (pad_right) *pad_right = (stride - rw % stride) % stride;   // predict.py:24
    return PD_OK;
}

extern "C" int pd_image_resize_pad(pd_ctx* ctx, const void* img, int img_dtype, int h, int w, int channels, double scale,
                                   int stride, double pad_value, void* out, int out_dtype, int out_h, int out_w, int mem) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    if (!img || !out) PD_FAIL(ctx, PD_E_INVALID, "pd_image_resize_pad: img and out are required");
    if ((img_dtype != PD_U8 && img_dtype != PD_F32) || (out_dtype != PD_U8 && out_dtype != PD_F32))
        PD_FAIL(ctx, PD_E_INVALID, "pd_image_resize_pad: dtypes must be PD_U8 or PD_F32");
    if (channels < 1 || channels > 4) PD_FAIL(ctx, PD_E_INVALID, "pd_image_resize_pad: %d channels", channels);
    int32_t rh, rw, pd, pr;
    if (pd_image_resize_shape(h, w, scale, stride, &rh, &rw, &pd, &pr) != PD_OK)
        PD_FAIL(ctx, PD_E_INVALID, "pd_image_resize_pad: bad size %dx%d, scale %g or stride %d", h, w, scale, stride);
    if (out_h != rh + pd || out_w != rw + pr)
        PD_FAIL(ctx, PD_E_INVALID, "pd_image_resize_pad: the output is %dx%d, given %dx%d (pd_image_resize_shape)",
                rh + pd, rw + pr, out_h, out_w);
    const size_t in_bytes = (size_t)h * w * channels * (img_dtype == PD_U8 ? 1 : 4);
    const size_t out_bytes = (size_t)out_h * out_w * channels * (out_dtype == PD_U8 ? 1 : 4);
    ImageArgs a;
    memset(&a, 0, sizeof(a));
    a.identity = rh == h && rw == w;
    if (!a.identity) {
        // source step per destination pixel = 1 / scale on both axes (fx, fy given, no dsize)
        int rc = pd_upload_tables(ctx, h, w, rh, rw, 1.0 / scale, 1.0 / scale, &ctx->tabs[1]);
        if (rc != PD_OK) return rc;
        a.ty = ctx->tabs[1].ty;
        a.tx = ctx->tabs[1].tx;
    }
    a.src = img;
    a.dst = out;
    if (mem == PD_MEM_HOST) {
        int rc = aux_ensure(ctx, align256(in_bytes) + out_bytes);
        if (rc != PD_OK) return rc;
        PD_HIP(ctx, hipMemcpyAsync(ctx->aux_dev, img, in_bytes, hipMemcpyHostToDevice, ctx->stream));
        a.src = ctx->aux_dev;
        a.dst = ctx->aux_dev + align256(in_bytes);
    }
    a.src_u8 = img_dtype == PD_U8;
    a.dst_u8 = out_dtype == PD_U8;
    a.h = h;
    a.w = w;
    a.C = channels;
    a.rh = rh;
    a.rw = rw;
    a.oh = out_h;
    a.ow = out_w;
    a.pad = (float)pad_value;
    const size_t px = (size_t)out_h * out_w;
    hipLaunchKernelGGL(k_image_resize_pad, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, ctx->stream, a);
    PD_HIP(ctx, hipGetLastError());
    if (mem == PD_MEM_HOST) {
        PD_HIP(ctx, hipMemcpyAsync(out, a.dst, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return PD_OK;
}

// ---------------------------------------------------------------------------------
// Skeleton rendering (predict.py:256-307): filled circles of radius 4 on the peaks, the blend
// written as part.jpg, then one filled polygon per (limb, person) blended into the canvas in
// draw order.  The shapes follow OpenCV's drawing algorithms (imgproc/drawing.cpp) as published:
// cv2.circle's integer midpoint walk; cv2.ellipse2Poly's vertices at whole degrees from a
// float32 sine table of 7 decimals, rounded half to even; cv2.fillConvexPoly's outline lines plus
// its scan in 16.16 fixed point.  OpenCV itself is not available to check against, and edges that
// cross the image border are walked unclipped here (cv2 first moves their end points onto the
// border).  The host rasterises every polygon into a bit mask over its bounding box; the kernel
// looks pixels up in the masks.
// ---------------------------------------------------------------------------------
static __constant__ unsigned char c_colors[18][3] = {  // predict.py:258-260
    {255, 0, 0},   {255, 85, 0},  {255, 170, 0}, {255, 255, 0}, {170, 255, 0}, {85, 255, 0},
    {0, 255, 0},   {0, 255, 85},  {0, 255, 170}, {0, 255, 255}, {0, 170, 255}, {0, 85, 255},
    {0, 0, 255},   {85, 0, 255},  {170, 0, 255}, {255, 0, 255}, {255, 0, 170}, {255, 0, 85}};

// half-width of the filled rows of cv2.circle(radius 4, thickness -1) at |dy| = 0 .. 4
static __constant__ int c_disc4[5] = {4, 3, 3, 2, 0};

struct ShapeMask {  // one polygon: bits of its bounding box (clipped to the image), row-major, 32 pixels a word
    int x0, y0, x1, y1;
    int words_per_row, colour;
    unsigned int offset;  // first word in the mask array
    int pad;
};

struct RenderArgs {
    const unsigned char* img;  // (H, W, 3)
    int H, W;
    int n_circles, n_shapes;
    const int32_t* circles;    // [n][3]: x, y, colour index
    const ShapeMask* shapes;
    const unsigned int* bits;
    unsigned char* part_out;   // (H, W, 3) or NULL
    unsigned char* final_out;  // (H, W, 3) or NULL
};

// saturate_cast<uchar>(a * wa + b * wb), float32, round half to even (cv2.addWeighted on 8-bit images)
__device__ __forceinline__ float blend(float a, float wa, float b, float wb) {
    const float r = rintf(a * wa + b * wb);
    return r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
}

__global__ void __launch_bounds__(256)
k_render(RenderArgs a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)a.H * a.W) return;
    const int y = (int)(i / a.W), x = (int)(i - (size_t)y * a.W);
    const float i0 = a.img[3 * i], i1 = a.img[3 * i + 1], i2 = a.img[3 * i + 2];
    float c0 = i0, c1 = i1, c2 = i2;
    for (int k = 0; k < a.n_circles; ++k) {
        const int dx = abs(x - a.circles[3 * k]), dy = abs(y - a.circles[3 * k + 1]);
        if (dy <= 4 && dx <= c_disc4[dy]) {  // cv2.circle(canvas, peak, 4, colour, thickness=-1), predict.py:269
            const unsigned char* col = c_colors[a.circles[3 * k + 2] % 18];
            c0 = col[0];
            c1 = col[1];
            c2 = col[2];
        }
    }
    if (a.part_out) {  // cv2.addWeighted(img, 0.3, canvas, 0.7, 0), predict.py:277
        a.part_out[3 * i] = (unsigned char)blend(i0, 0.3f, c0, 0.7f);
        a.part_out[3 * i + 1] = (unsigned char)blend(i1, 0.3f, c1, 0.7f);
        a.part_out[3 * i + 2] = (unsigned char)blend(i2, 0.3f, c2, 0.7f);
    }
    if (!a.final_out) return;
    for (int k = 0; k < a.n_shapes; ++k) {
        const ShapeMask m = a.shapes[k];
        if (x < m.x0 || x > m.x1 || y < m.y0 || y > m.y1) continue;
        const int bx = x - m.x0;
        const unsigned int word = a.bits[m.offset + (size_t)(y - m.y0) * m.words_per_row + (bx >> 5)];
        if ((word >> (bx & 31)) & 1u) {  // canvas = addWeighted(canvas, 0.4, cur_canvas, 0.6, 0), predict.py:300
            const unsigned char* col = c_colors[m.colour % 18];
            c0 = blend(c0, 0.4f, (float)col[0], 0.6f);
            c1 = blend(c1, 0.4f, (float)col[1], 0.6f);
            c2 = blend(c2, 0.4f, (float)col[2], 0.6f);
        }
    }
    a.final_out[3 * i] = (unsigned char)c0;
    a.final_out[3 * i + 1] = (unsigned char)c1;
    a.final_out[3 * i + 2] = (unsigned char)c2;
}

// ---- the polygon of cv2.ellipse2Poly(centre, axes, angle, 0, 360, 1) and its filled mask (host) ----
namespace {
struct Pt {
    long long x, y;
};

const float* sine_table() {  // sines of 0 .. 450 degrees to 7 decimals, as float32
    static float tab[451];
    static bool made = false;
    if (!made) {
        for (int d = 0; d <= 450; ++d) tab[d] = (float)(nearbyint(sin(d * (3.14159265358979323846 / 180.0)) * 1e7) / 1e7);
        made = true;
    }
    return tab;
}

void ellipse_polygon(int cx, int cy, int ax, int ay, int angle, std::vector<Pt>& pts) {
    const float* st = sine_table();
    while (angle < 0) angle += 360;
    while (angle > 360) angle -= 360;
    const double alpha = st[450 - angle], beta = st[angle];
    pts.clear();
    for (int i = 0; i <= 360; ++i) {
        const double x = (double)ax * st[450 - i], y = (double)ay * st[i];
        Pt p;
        p.x = (long long)nearbyint(cx + x * alpha - y * beta);  // (round half to even)
        p.y = (long long)nearbyint(cy + x * beta + y * alpha);
        if (pts.empty() || pts.back().x != p.x || pts.back().y != p.y) pts.push_back(p);
    }
    if (pts.size() == 1) pts.assign(2, Pt{cx, cy});
}

struct MaskWriter {
    ShapeMask m;
    std::vector<unsigned int>* bits;
    void set(long long x, long long y) {
        if (x < m.x0 || x > m.x1 || y < m.y0 || y > m.y1) return;
        const long long bx = x - m.x0;
        (*bits)[m.offset + (size_t)(y - m.y0) * m.words_per_row + (size_t)(bx >> 5)] |= 1u << (bx & 31);
    }
};

// the 8-connected line between two pixels: from the left end, a step along the longer axis per
// pixel, one along the shorter axis whenever the error term is negative
void draw_line(MaskWriter& w, Pt p1, Pt p2) {
    long long dx = p2.x - p1.x, dy = p2.y - p1.y;
    if (dx < 0) {
        dx = -dx;
        dy = -dy;
        p1 = p2;
    }
    const long long sy = dy < 0 ? -1 : 1;
    if (dy < 0) dy = -dy;
    const bool steep = dy > dx;
    const long long major = steep ? dy : dx, minor = steep ? dx : dy;
    long long err = major - 2 * minor, x = p1.x, y = p1.y;
    for (long long i = 0; i <= major; ++i) {
        w.set(x, y);
        const bool neg = err < 0;
        err += -2 * minor + (neg ? 2 * major : 0);
        if (steep) {
            y += sy;
            if (neg) x += 1;
        } else {
            x += 1;
            if (neg) y += sy;
        }
    }
}

void fill_polygon(MaskWriter& w, const std::vector<Pt>& v, int H, int W) {
    const long long ONE = 1ll << 16;
    const int n = (int)v.size();
    Pt p0 = v[n - 1];
    long long xmin = v[0].x, xmax = v[0].x, ymin = v[0].y, ymax = v[0].y;
    int imin = 0;
    for (int i = 0; i < n; ++i) {
        if (v[i].y < ymin) {
            ymin = v[i].y;
            imin = i;
        }
        ymax = std::max(ymax, v[i].y);
        xmax = std::max(xmax, v[i].x);
        xmin = std::min(xmin, v[i].x);
        draw_line(w, p0, v[i]);
        p0 = v[i];
    }
    if (n < 3 || xmax < 0 || ymax < 0 || xmin >= W || ymin >= H) return;
    ymax = std::min<long long>(ymax, H - 1);
    struct Edge {
        int idx, di;
        long long x, dx, ye;
    } edge[2] = {{imin, 1, -ONE, 0, ymin}, {imin, n - 1, -ONE, 0, ymin}};
    int edges = n;
    long long y = ymin;
    do {
        for (Edge& e : edge) {
            if (y < e.ye) continue;
            int idx0 = e.idx, idx = idx0 + e.di;
            if (idx >= n) idx -= n;
            for (; edges-- > 0;) {
                const long long ty = v[idx].y;
                if (ty > y) {
                    const long long xs = v[idx0].x * ONE, xe = v[idx].x * ONE;
                    e.ye = ty;
                    e.dx = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));  // (truncating division)
                    e.x = xs;
                    e.idx = idx;
                    break;
                }
                idx0 = idx;
                idx += e.di;
                if (idx >= n) idx -= n;
            }
        }
        if (edges < 0) break;
        if (y >= 0) {
            const int left = edge[0].x > edge[1].x ? 1 : 0, right = 1 - left;
            long long xx1 = (edge[left].x + (ONE >> 1)) >> 16, xx2 = (edge[right].x + (ONE >> 1)) >> 16;
            if (xx2 >= 0 && xx1 < W) {
                xx1 = std::max<long long>(xx1, 0);
                xx2 = std::min<long long>(xx2, W - 1);
                for (long long x = xx1; x <= xx2; ++x) w.set(x, y);
            }
        }
        edge[0].x += edge[0].dx;
        edge[1].x += edge[1].dx;
    } while (++y <= ymax);
}
}  // namespace

extern "C" int pd_render(pd_ctx* ctx, const uint8_t* img, int H, int W, int n_circles, const int32_t* circles,
                         int n_ellipses, const int32_t* ellipses, uint8_t* part_out, uint8_t* final_out) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    if (!img || H < 1 || W < 1 || n_circles < 0 || n_ellipses < 0 || (n_circles && !circles) || (n_ellipses && !ellipses))
        PD_FAIL(ctx, PD_E_INVALID, "pd_render: bad arguments");
    // (cx, cy, a, b, angle in whole degrees, colour) -> polygon -> bit mask over its bounding box
    std::vector<ShapeMask> shapes((size_t)n_ellipses);
    std::vector<unsigned int> bits;
    std::vector<Pt> poly;
    for (int i = 0; i < n_ellipses; ++i) {
        const int32_t* e = ellipses + 6 * (size_t)i;
        if (e[2] < 0 || e[3] < 0 || e[2] > (1 << 20) || e[3] > (1 << 20) || abs(e[0]) > (1 << 24) || abs(e[1]) > (1 << 24))
            PD_FAIL(ctx, PD_E_INVALID, "pd_render: ellipse %d out of range", i);
        ellipse_polygon(e[0], e[1], e[2], e[3], e[4], poly);
        long long x0 = poly[0].x, x1 = x0, y0 = poly[0].y, y1 = y0;
        for (const Pt& p : poly) {
            x0 = std::min(x0, p.x);
            x1 = std::max(x1, p.x);
            y0 = std::min(y0, p.y);
            y1 = std::max(y1, p.y);
        }
        MaskWriter w;
        w.bits = &bits;
        ShapeMask& m = w.m;
        m.x0 = (int)std::max<long long>(x0, 0);
        m.y0 = (int)std::max<long long>(y0, 0);
        m.x1 = (int)std::min<long long>(x1, W - 1);
        m.y1 = (int)std::min<long long>(y1, H - 1);
        m.colour = e[5];
        m.pad = 0;
        m.offset = (unsigned int)bits.size();
        m.words_per_row = m.x1 >= m.x0 ? (m.x1 - m.x0) / 32 + 1 : 0;
        if (m.x1 >= m.x0 && m.y1 >= m.y0) {
            bits.resize(bits.size() + (size_t)m.words_per_row * (m.y1 - m.y0 + 1), 0u);
            fill_polygon(w, poly, H, W);
        } else {
            m.x1 = m.x0 - 1;  // nothing of it inside the image
        }
        shapes[(size_t)i] = m;
    }
    const size_t img_bytes = (size_t)H * W * 3;
    const size_t off_part = align256(img_bytes), off_final = off_part + align256(img_bytes);
    const size_t off_circ = off_final + align256(img_bytes);
    const size_t off_shape = off_circ + align256((size_t)n_circles * 3 * sizeof(int32_t));
    const size_t off_bits = off_shape + align256(shapes.size() * sizeof(ShapeMask));
    int rc = aux_ensure(ctx, off_bits + align256(bits.size() * sizeof(unsigned int)) + 256);
    if (rc != PD_OK) return rc;
    unsigned char* d = ctx->aux_dev;
    PD_HIP(ctx, hipMemcpyAsync(d, img, img_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (n_circles)
        PD_HIP(ctx, hipMemcpyAsync(d + off_circ, circles, (size_t)n_circles * 3 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    if (n_ellipses) {
        PD_HIP(ctx, hipMemcpyAsync(d + off_shape, shapes.data(), shapes.size() * sizeof(ShapeMask), hipMemcpyHostToDevice, ctx->stream));
        if (!bits.empty())
            PD_HIP(ctx, hipMemcpyAsync(d + off_bits, bits.data(), bits.size() * sizeof(unsigned int), hipMemcpyHostToDevice, ctx->stream));
    }
    RenderArgs a;
    a.img = d;
    a.H = H;
    a.W = W;
    a.n_circles = n_circles;
    a.n_shapes = n_ellipses;
    a.circles = (const int32_t*)(d + off_circ);
    a.shapes = (const ShapeMask*)(d + off_shape);
    a.bits = (const unsigned int*)(d + off_bits);
    a.part_out = part_out ? d + off_part : nullptr;
    a.final_out = final_out ? d + off_final : nullptr;
    const size_t px = (size_t)H * W;
    hipLaunchKernelGGL(k_render, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, ctx->stream, a);
    PD_HIP(ctx, hipGetLastError());
    if (part_out) PD_HIP(ctx, hipMemcpyAsync(part_out, d + off_part, img_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (final_out) PD_HIP(ctx, hipMemcpyAsync(final_out, d + off_final, img_bytes, hipMemcpyDeviceToHost, ctx->stream));
    // the host arrays above are pageable: the copies out of them have completed by the time this returns
    PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PD_OK;
}
