// pd_decode_batch_static: the whole decode of a small batch without a context and without
// any allocation, for callers that cannot call hipMalloc (code under HIP graph capture,
// tight serving loops).  Everything pd_decode_batch keeps in library-owned device memory --
// raw and ordered peak lists, counters, candidate-pair scores, connections, person rows,
// bicubic tap tables -- lives in one file-scope device array, carved up below; the kernels
// are the ones the context path launches (peak stage: the exact float64 streaming kernel).
//
// The array exists once per GPU and keeps the last call's contents: the counters that must
// start at zero are cleared by the first kernel of every call (the one that fills the tap
// tables; no memset is queued, so a caller that captures the call records kernel nodes only),
// everything else is written before it is read.  Calls that overlap on one GPU race on it.
#include <atomic>
#include <mutex>

#include "common.h"

#pragma clang fp contract(off)

#define ST_ALIGN 256
#define ST_BYTES PD_STATIC_SCRATCH_BYTES

static __device__ __align__(256) unsigned char scratch[ST_BYTES];

// The device address of `scratch` on each GPU (every GPU holds its own copy of the array at
// its own address).  In host code the array's name is only a handle for hipGetSymbolAddress.
static std::atomic<unsigned char*> g_scratch_ptr[64];
static pd_ctx g_view[64];  // host-side description of the carve-up, one per device (static storage: zero-initialised)
static std::mutex g_mu[64];
static std::mutex g_setup_mu;

int pd_static_setup(int device, char* err, size_t err_len) {
    if (device < 0 || device >= 64) {
        snprintf(err, err_len, "static scratch: device %d out of range", device);
        return PD_E_INVALID;
    }
    std::lock_guard<std::mutex> lock(g_setup_mu);
    if (g_scratch_ptr[device].load(std::memory_order_acquire)) return PD_OK;
    int cur = -1;
    hipError_t e = hipGetDevice(&cur);
    if (e == hipSuccess && cur != device) e = hipSetDevice(device);
    unsigned char* p = nullptr;
    if (e == hipSuccess) e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    if (e != hipSuccess || p == nullptr) {
        snprintf(err, err_len, "hipGetSymbolAddress(scratch) on device %d failed: %s", device, hipGetErrorString(e));
        return PD_E_HIP;
    }
    g_scratch_ptr[device].store(p, std::memory_order_release);
    return PD_OK;
}

// Fills the x8 tap tables (first tap and Keys weights per destination index) from the
// closed form: identical to the host-built tables of the context path.  Also clears the
// counters that every decode starts from (zero[0 .. zero_words)).
__global__ void k_fill_up8_tables(int32_t* ofs_y, float* coef_y, int H, int32_t* ofs_x, float* coef_x, int W, Up8 up,
                                  int32_t* zero, int zero_words) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int j = i; j < zero_words; j += gridDim.x * blockDim.x) zero[j] = 0;
    if (i < H) {
        ofs_y[i] = pd_up8_ofs(i);
        const int p = pd_up8_phase(i);
#pragma unroll
        for (int k = 0; k < 4; ++k) coef_y[4 * i + k] = up.c[p][k];
    }
    if (i < W) {
        ofs_x[i] = pd_up8_ofs(i);
        const int p = pd_up8_phase(i);
#pragma unroll
        for (int k = 0; k < 4; ++k) coef_x[4 * i + k] = up.c[p][k];
    }
}

struct Carver {
    unsigned char* base;
    size_t at;
    template <typename T>
    T* take(size_t count) {
        T* p = (T*)(base + at);
        at = (at + count * sizeof(T) + ST_ALIGN - 1) & ~(size_t)(ST_ALIGN - 1);
        return p;
    }
};

extern "C" int pd_decode_batch_static(const float* heat, const float* paf, int B, int h, int w, int H, int W,
                                      double thre1, double thre2, int mode, void* out, uint64_t out_capacity,
                                      void* stream) {
    if (!heat || !paf || !out) {
        pd_set_global_error("pd_decode_batch_static: heat, paf and out are required");
        return PD_E_INVALID;
    }
    if (((uintptr_t)heat | (uintptr_t)paf | (uintptr_t)out) & 15) {
        pd_set_global_error("pd_decode_batch_static: device arrays must be 16-byte aligned");
        return PD_E_INVALID;
    }
    if (mode != PD_MODE_P && mode != PD_MODE_D) {
        pd_set_global_error("pd_decode_batch_static: mode %d", mode);
        return PD_E_INVALID;
    }
    if (B < 1 || B > PD_STATIC_MAX_BATCH) {
        pd_set_global_error("pd_decode_batch_static: batch %d is outside 1..%d (the static scratch array is sized for that)", B,
                            PD_STATIC_MAX_BATCH);
        return PD_E_INVALID;
    }
    if (H > PD_STATIC_MAX_SIDE || W > PD_STATIC_MAX_SIDE || H < 2 * PD_GR + 2 || W < 2 * PD_GR + 2) {
        pd_set_global_error("pd_decode_batch_static: image %dx%d is outside %d..%d per side", H, W, 2 * PD_GR + 2, PD_STATIC_MAX_SIDE);
        return PD_E_INVALID;
    }
    if (h < 1 || w < 1 || H > 8 * h || W > 8 * w || H <= 8 * (h - 1) || W <= 8 * (w - 1)) {
        pd_set_global_error("pd_decode_batch_static: %dx%d is not a crop of the x8 map of %dx%d", H, W, h, w);
        return PD_E_INVALID;
    }
    const ResLayout lay = pd_res_layout(B);
    if (out_capacity < lay.off_payload) {
        pd_set_global_error("pd_decode_batch_static: %llu bytes cannot hold the count tables of %d images (%zu bytes)",
                            (unsigned long long)out_capacity, B, lay.off_payload);
        return PD_E_INVALID;
    }
    int device = -1;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) {
        pd_set_global_error("pd_decode_batch_static: no current HIP device");
        return PD_E_HIP;
    }
    if (!g_scratch_ptr[device].load(std::memory_order_acquire)) {  // first call on this device (no context was created on it before)
        char err[256];
        const int rc = pd_static_setup(device, err, sizeof(err));
        if (rc != PD_OK) {
            pd_set_global_error("%s", err);
            return rc;
        }
    }
    std::lock_guard<std::mutex> lock(g_mu[device]);  // host-side state only: overlapping calls still race on the GPU
    pd_ctx* v = &g_view[device];
    const size_t PK = PD_STATIC_PEAKS, NB = PD_STATIC_MAX_BATCH;
    Carver c = {g_scratch_ptr[device].load(std::memory_order_acquire), 0};
    v->cfg.device_id = device;
    v->cfg.max_batch = PD_STATIC_MAX_BATCH;
    v->cfg.max_h = v->cfg.max_w = PD_STATIC_MAX_SIDE;
    v->cfg.max_peaks_per_part = PD_STATIC_PEAKS;
    v->cfg.max_persons = PD_STATIC_PERSONS;
    v->cfg.max_pairs_per_image = PD_STATIC_PAIRS;
    v->mode = mode;
    v->impl = PD_IMPL_STREAM;
    v->stream = (hipStream_t)stream;
    v->host_results = 1;
    v->zero_bytes = (2 * NB * PD_N_PARTS + 96) * sizeof(int32_t);
    v->zero_block = c.take<int32_t>(v->zero_bytes / sizeof(int32_t));
    v->raw_count = v->zero_block;
    v->cand_count = v->raw_count + NB * PD_N_PARTS;
    v->fstats = c.take<int64_t>(4);
    v->raw = c.take<RawPeak>(NB * PD_N_PARTS * PK);
    v->pk_y = c.take<int32_t>(NB * PD_N_PARTS * PK);
    v->pk_x = c.take<int32_t>(NB * PD_N_PARTS * PK);
    v->pk_id = c.take<int32_t>(NB * PD_N_PARTS * PK);
    v->pk_score = c.take<double>(NB * PD_N_PARTS * PK);
    v->part_count = c.take<int32_t>(NB * PD_N_PARTS);
    v->part_base = c.take<int32_t>(NB * (PD_N_PARTS + 1));
    v->pair_off = c.take<int32_t>(NB * (PD_N_LIMBS + 1));
    v->pair_score = c.take<double>(NB * (size_t)PD_STATIC_PAIRS);
    v->conn = c.take<pd_conn>(NB * PD_N_LIMBS * PK);
    v->conn_conf = c.take<double>(NB * PD_N_LIMBS * PK * 2);
    v->limb_count = c.take<int32_t>(NB * PD_N_LIMBS);
    v->persons = c.take<double>(NB * (size_t)PD_STATIC_PERSONS * 20);
    v->person_count = c.take<int32_t>(NB);
    v->status = c.take<int32_t>(NB);
    int32_t* tab_i = c.take<int32_t>(2 * PD_STATIC_MAX_SIDE);
    float* tab_f = c.take<float>(8 * PD_STATIC_MAX_SIDE);
    if (c.at > ST_BYTES) {  // a build-time property; checked here because the carve-up is run-time code
        pd_set_global_error("pd_decode_batch_static: the carve-up needs %zu bytes, the scratch array has %u", c.at, ST_BYTES);
        return PD_E_STATE;
    }
    TableSlot& t = v->tabs[0];
    t.i = tab_i;
    t.f = tab_f;
    t.ty.ofs = tab_i;
    t.ty.coef = tab_f;
    t.tx.ofs = tab_i + PD_STATIC_MAX_SIDE;
    t.tx.coef = tab_f + 4 * PD_STATIC_MAX_SIDE;
    v->src_kind = PD_SRC_LOWRES;
    v->lo_heat = heat;
    v->lo_paf = paf;
    v->B = B;
    v->h = h;
    v->w = w;
    v->H = H;
    v->W = W;
    v->heat_f32 = 0;
    v->res_dev = (unsigned char*)out;
    v->res_user_cap = (size_t)out_capacity;
    v->zero_clean = 1;  // the array keeps the last call's counters: the table kernel below clears them
    v->have_peaks = v->have_conns = v->have_persons = 0;
    v->err[0] = 0;

    Up8 up;
    pd_up8_fill(&up);
    const int n = H > W ? H : W;
    hipLaunchKernelGGL(k_fill_up8_tables, dim3((n + 255) / 256), dim3(256), 0, v->stream, tab_i, tab_f, H,
                       tab_i + PD_STATIC_MAX_SIDE, tab_f + 4 * PD_STATIC_MAX_SIDE, W, up, v->zero_block,
                       (int)(v->zero_bytes / sizeof(int32_t)));
    int rc = pd_launch_smooth_nms(v, thre1, nullptr);
    if (rc == PD_OK) rc = pd_launch_sort_peaks(v);
    v->have_peaks = 1;
    if (rc == PD_OK) rc = pd_launch_pairs(v, thre2);
    if (rc == PD_OK) rc = pd_launch_match(v);
    v->have_conns = 1;
    if (rc == PD_OK) rc = pd_launch_parse(v);
    v->have_persons = 1;
    if (rc == PD_OK) rc = pd_launch_pack(v);
    if (rc != PD_OK) pd_set_global_error("pd_decode_batch_static: %s", v->err);
    return rc;
}
