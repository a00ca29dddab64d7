// Shared declarations of libposedecode (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "../../include/posedecode.h"

// Every floating-point result on this path must be the separately rounded
// multiply / add the CPU reference performs: no fused multiply-add anywhere.
#pragma clang fp contract(off)

#define PD_GR 12      // Gaussian radius: int(4.0 * 3 + 0.5)
#define PD_GTAPS 13   // centre + 12 symmetric pairs
#define PD_MID 10     // sample points per limb candidate (utils/postprocess.py:79)
#define PD_MAX_SCALES 8  // scales of one pd_upsample_avg call

enum { PD_SRC_NONE = 0, PD_SRC_LOWRES = 1, PD_SRC_FULLRES = 2 };  // and PD_SRC_MULTI = 3 (multiscale.h)

struct RawPeak {
    int32_t y, x;
    double score;
};

// Limb table, cfgs/config_vgg19.py:56-57.
__host__ __device__ inline int pd_limb_from(int k) {
    const int t[19] = {1, 8, 9, 1, 11, 12, 1, 2, 3, 2, 1, 5, 6, 5, 1, 0, 0, 14, 15};
    return t[k];
}
__host__ __device__ inline int pd_limb_to(int k) {
    const int t[19] = {8, 9, 10, 11, 12, 13, 2, 3, 4, 16, 5, 6, 7, 17, 0, 14, 15, 16, 17};
    return t[k];
}

// Bicubic tap table of one axis: for destination index d the first source tap
// (floor(s) - 1, unclamped) and the four float32 Keys weights (A = -0.75).
struct CubicTable {
    const int32_t* ofs;  // [n]
    const float* coef;   // [n][4]
};

// Closed form of the x8 tap table (cv2.resize(.., fx=8, fy=8)): the source coordinate
// of destination index d is (d - 3.5) / 8, so the first tap is ((d - 4) >> 3) - 1 and
// the fractional part -- hence the four weights -- depends only on (d - 4) & 7.  The
// weights are the table's own entries (computed once on the host), so values are
// bit-identical to the table path.
struct Up8 {
    float c[8][4];
};
__host__ __device__ inline int pd_up8_ofs(int d) { return ((d - 4) >> 3) - 1; }
__host__ __device__ inline int pd_up8_phase(int d) { return (d - 4) & 7; }

// Where the full-resolution heat / PAF values come from.
struct MapSource {
    int kind;             // PD_SRC_LOWRES or PD_SRC_FULLRES
    int h, w;             // network-resolution size (lowres)
    int H, W;             // full-resolution size
    const float* lo;      // (B, h, w, C) float32
    int C;                // channels of lo
    const double* full;   // (B, Cfull, H, W) float64 planes
    int Cfull;
    CubicTable ty, tx;    // x8 tables of length H and W
    Up8 up8;              // the same weights by phase
};

// Byte layout of the packed result buffer of a batch of B images:
//   int64 header[8]: B, n_peaks, n_conns, n_persons, first_bad_image (or -1), 0, 0, 0
//   int32 part_count[B][18], limb_count[B][19], person_count[B], status[B]
//   (16-byte aligned) pd_peak[n_peaks], pd_conn[n_conns], double[n_persons][20]
struct ResLayout {
    size_t off_part, off_limb, off_pers, off_stat, off_payload;
};
__host__ __device__ inline ResLayout pd_res_layout(int B) {
    ResLayout l;
    l.off_part = 64;
    l.off_limb = l.off_part + (size_t)B * PD_N_PARTS * 4;
    l.off_pers = l.off_limb + (size_t)B * PD_N_LIMBS * 4;
    l.off_stat = l.off_pers + (size_t)B * 4;
    l.off_payload = (l.off_stat + (size_t)B * 4 + 15) & ~(size_t)15;
    return l;
}

// Everything a captured pd_decode_batch launch sequence depends on.
struct GraphKey {
    const float* heat;
    const float* paf;
    int B, h, w, H, W;
    double t1, t2;
    int mode, impl;
    uint32_t prof;
    int host_results;
    int pack_direct;
    int pairs_pre;
    const void* skel;
    uint64_t skel_cap;
    uint64_t gen;  // pd_ctx::gen at the time: any device buffer or table a launch refers to that was
                   // reallocated or rewritten since makes the recorded sequence stale
};

#define PD_GRAPH_SLOTS 8
struct GraphSlot {
    GraphKey key;
    hipGraphExec_t exec;  // NULL: the key was seen once and ran eagerly
    int seen;
    size_t res_have;
    int prof_n;
    uint64_t used;        // pd_ctx::graph_clock at the last use
};

struct TableSlot {
    int32_t* i;
    float* f;
    size_t cap;
    CubicTable ty, tx;
    int key[4];      // hs, ws, Hd, Wd the tables were built for
    double scale[2];
};

// One queued pd_submit call.
struct SubmitJob {
    const float* heat;
    const float* paf;
    int B, h, w, H, W;
    double t1, t2;
    void* skel;
    uint64_t skel_cap;
};

struct pd_ctx {
    pd_config cfg;
    int mode, impl;
    hipStream_t stream;
    char err[512];

    // ---- current source ----
    int src_kind;
    int B, h, w, H, W;
    int heat_f32;  // full-res heat came from float32 data: round after each filter axis
    const float* lo_heat;
    const float* lo_paf;
    float* stage_heat;  // staging for host-resident network outputs
    float* stage_paf;
    size_t stage_heat_cap, stage_paf_cap;
    double* full_heat;  // (B, 19, H, W)
    double* full_paf;   // (B, 38, H, W)
    size_t full_heat_cap, full_paf_cap;
    int have_full_heat, have_full_paf;
    // ---- multi-scale source (pd_set_multiscale): the network outputs of every test scale ----
    int ms_n;
    const float* ms_heat[PD_MAX_SCALES];  // (B, hs, ws, 19)
    const float* ms_paf[PD_MAX_SCALES];   // (B, hs, ws, 38)
    int32_t ms_hs[PD_MAX_SCALES], ms_ws[PD_MAX_SCALES], ms_crop_h[PD_MAX_SCALES], ms_crop_w[PD_MAX_SCALES];
    int ms_heat_full;  // the heat average of the current source has been written to full_heat
    void* ms_dev;      // device copy of the MultiSource (for the kernel that takes it by address)
    unsigned char ms_dev_image[640];  // what ms_dev holds
    // sparse peak stage of a multi-scale source: composite weights in cell form, planar copies of the heatmaps
    int msf_ok;
    int msf_key[3 + 4 * PD_MAX_SCALES];  // the geometry the tables were built for
    float* msf_tab;
    int32_t* msf_lo;
    size_t msf_tab_cap, msf_lo_cap;      // in bytes
    int msf_lds_a, msf_lds_r;            // floats of the bound kernel's staging areas
    int msf_nq[PD_MAX_SCALES];           // float4 words of a table row in use, per scale
    float* ms_planar;                    // per scale (B, 18, hs, ws) float32
    size_t ms_planar_cap;
    void* tmp_in;  // staging for host-resident full-resolution input
    size_t tmp_in_cap;
    double* tmp_smooth;  // (B, 18, H, W) scratch of the simple path / debug hook
    size_t tmp_smooth_cap;
    double* tmp_smooth2;
    size_t tmp_smooth2_cap;

    // bicubic tap tables on the device: slot 0 = x8 tables of the fused low-res
    // source, slot 1 / 2 = first / second resize of pd_upsample_avg
    TableSlot tabs[2 + PD_MAX_SCALES];  // 0: x8 of the fused source, 1: spare, 2 + m: second resize of scale m
    int tab_h, tab_w, tab_H, tab_W;  // what the slot-0 tables were built for

    // ---- candidate filter (float32 composite of bicubic x8 and Gaussian) ----
    int kf_ok;                       // tables valid for (kf_h, kf_w, kf_H, kf_W)
    int kf_h, kf_w, kf_H, kf_W;
    float* kf_tab;                   // per 8-pixel cell row / column: 10 x 7 composite weights and bound weights
    int32_t* kf_reg;                 // [hc] | [wc]: 1 = the cell row / column has the interior weights
    size_t kf_tab_cap, kf_reg_cap;   // in elements
    int32_t* cells;                  // active cells: regular ones from the front, border ones from the back
    size_t cells_cap;
    int32_t* cand;                   // candidate list, int2 (plane, y << 16 | x)
    int32_t* cand_count;             // [B][18] candidates per plane | list length | cell counts
    float* full_absmax;              // [max_batch] max |value| of the full-resolution heatmaps per image (rounded up)
    int full_absmax_ok;              // full_absmax bounds the maps in full_heat
    float2* plane_slack;             // [max_batch * 18] (thr_lo, e2) of the candidate test, per plane
    float* heat_planar;              // (B, 19, h, w) copy of the network heatmaps
    size_t heat_planar_cap;
    float* paf_pairs;                // (B, 19, h, w, 2) copy of the PAFs, made for large batches only
    size_t paf_pairs_cap;
    int cand_cap;                    // CK
    // ---- peaks ----
    int peaks_sorted;    // the peak stage has ordered the peaks itself: pd_launch_sort_peaks has nothing to do
    int32_t* zero_block; // raw_count | cand_count + list counters: zero at the start of every decode
    size_t zero_bytes;
    int zero_clean;      // the block is zero (or a kernel that zeroes it is queued): pd_pack leaves it so
    // Two contexts may launch on one stream (pd_share_stream): the second decode queues behind the
    // first, and a context then waits for its own last decode through an event, not for the stream.
    int stream_shared;    // the stream belongs to another context: not destroyed with this one
    pd_ctx* owner;        // that context
    int n_sharers;        // contexts that launch on this one's stream
    int destroy_pending;  // pd_destroy was called while sharers were left: freed with the last of them
    hipEvent_t done_ev;   // recorded behind every whole decode of a context whose stream carries two contexts
    int done_valid;       // done_ev marks the end of everything this context has queued
    int32_t* pack_sync;   // k_parse_pack's arrival counter, tag and one published word per image (NULL: not available)
    int64_t* fstats;     // what pd_filter_stats reports, summed up by k_sort_peaks before the counters go
    RawPeak* raw;        // [B][18][PK] unordered
    int32_t* raw_count;  // [B][18]
    int32_t *pk_y, *pk_x, *pk_id;  // [B][18][PK] sorted (y, x) ascending
    double* pk_score;
    int32_t* part_count;  // [B][18]
    int32_t* part_base;   // [B][19] exclusive prefix of part_count
    int32_t* pair_off;    // [B][20] exclusive prefix over limbs of nA*nB
    double* pair_score;   // [B][max_pairs] ; < 0 = rejected
    // ---- connections ----
    pd_conn* conn;        // [B][19][PK]
    double* conn_conf;    // [B][19][PK][2]
    int32_t* limb_count;  // [B][19], -1 = special
    // ---- persons ----
    double* persons;        // [B][max_persons][20]
    int32_t* person_count;  // [B]
    int32_t* status;        // [B]
    // ---- packed results ----
    // one contiguous result buffer (see ResLayout) on the device and its pinned
    // host mirror; res_spec bytes are copied speculatively right behind the pack
    // kernel, the remainder (if any) once the totals are known
    unsigned char* res_dev;
    unsigned char* res_host;
    size_t res_cap, res_spec, res_have;
    size_t res_user_cap;  // != 0: res_dev is a caller's buffer of this many bytes (the pack kernel checks it)
    // The pack kernel stores the results to the pinned host mirror itself (no copy node behind a
    // decode) while results are small; megabytes of results per decode go faster through a copy
    // that overlaps with the kernels of other contexts.  pack_policy: -1 by the size of the last
    // results, 0 always copy, 1 always direct; pack_direct: what the next pack does.
    int pack_policy, pack_direct;
    size_t last_result_bytes;
    // Pair scoring with or without its pre-test (the pre-test pays when images hold many peaks; the
    // kernel without it is leaner): by the peaks per image of the last fetched results, before any
    // by the size of the map.  Same results either way.
    int pairs_pre;
    long long last_peaks_per_image;  // -1: no results fetched yet
    int have_peaks, have_conns, have_persons, packed;
    unsigned char* skel_dst;  // caller's device buffer for the skeleton copy of the results (or NULL)
    size_t skel_cap;
    int host_results;         // 0: with a skeleton buffer set, results go there only (no host mirror)
    int32_t* dbg_found;       // [max_batch][PK] len(found) per limb-18 connection, when asked for (pd_debug_found)
    // ---- scratch of the entry points either side of the decode (COCO rows, image resize, rendering) ----
    unsigned char* aux_dev;
    size_t aux_cap;
    // ---- stage timing (pd_profile) ----
    uint32_t prof_on;               // bit s set: stage s is timed
    int prof_open;                  // a start event is waiting for its stop
    int prof_n, prof_cap;           // prof_n: pairs recorded since the last harvest
    hipEvent_t* prof_ev;  // [2*i] start, [2*i+1] stop
    int* prof_stage;
    double prof_ms[PD_N_STAGES];
    int64_t prof_cnt[PD_N_STAGES];
    // ---- launcher thread of pd_submit ----
    std::thread* worker;
    std::mutex* wk_mu;
    std::condition_variable* wk_cv;
    std::deque<SubmitJob>* wk_jobs;
    int wk_busy;   // a job is being launched
    int wk_stop;
    int wk_rc;     // first failure since the last pd_flush
    char wk_err[512];
    // ---- HIP graph of the pd_decode_batch launch sequence ----
    int use_graph;
    uint64_t gen;         // bumped whenever a library-owned device buffer moves or a device table is rewritten
    // a call is run eagerly the first time its key is seen, recorded the second time and
    // replayed from then on; several keys are kept because a multi-GPU rank rotates its
    // skeleton buffers (two send sets x decodes per round)
    GraphSlot graphs[PD_GRAPH_SLOTS];
    uint64_t graph_clock;
};

// Error text for entry points that have no context (read back with pd_last_error(NULL)).
void pd_set_global_error(const char* fmt, ...);
// Per-device set-up of the allocation-free entry point (address of its scratch array).
int pd_static_setup(int device, char* err, size_t err_len);

// Zeroes `words` 32-bit words at `p` with a kernel on the context's stream.  Decode paths never
// queue a memset: a sequence a caller captures into a HIP graph then holds kernel nodes only
// (a memset node replayed next to a live RCCL communicator ended in a GPU memory fault on ROCm 7.2).
int pd_launch_clear(pd_ctx* ctx, void* p, size_t words);
__global__ void pd_k_clear_words(int32_t* p, int n);

// Record a start / stop event pair around a stage's kernel launch when profiling is on.
void pd_prof_begin(pd_ctx* ctx, int stage);
void pd_prof_end(pd_ctx* ctx);

#define PD_FAIL(ctx, code, ...)                                  \
    do {                                                         \
        snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__);   \
        return (code);                                           \
    } while (0)

#define PD_HIP(ctx, call)                                                              \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #call,   \
                     hipGetErrorString(e_), __FILE__, __LINE__);                       \
            return PD_E_HIP;                                                           \
        }                                                                              \
    } while (0)

// ---- device helpers shared by the kernels -------------------------------------

__device__ __forceinline__ int pd_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Orders the LDS (and memory) accesses of one wavefront's lanes among themselves: the
// hardware executes a wavefront's LDS instructions in order, so all that is needed is that
// the compiler keeps them in program order across this point.  For code whose LDS area
// belongs to a single wavefront, in place of a workgroup barrier.
__device__ __forceinline__ void pd_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// The value of the lane before / behind this one (0 at the ends of the wavefront): a whole-wave
// shift by one in the data path of a move, where __shfl_up / __shfl_down go through the LDS
// crossbar with an address computation and a wait.
__device__ __forceinline__ float pd_lane_prev(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float pd_lane_next(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

// max(prev(v), next(v), a, b), prev / next as above: the two shifts ride in the data path of two
// maximum instructions (three instructions where moves, their zeroed destinations and the
// maxima take eight).  A shifted operand must not have been written by the two instructions
// before, which the leading no-op sees to.
__device__ __forceinline__ float pd_max_lane_neighbours(float v, float a, float b) {
    float p, q;
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_max_f32_dpp %1, %2, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_max_f32 %0, %0, %1"
        : "=&v"(p), "=&v"(q)
        : "v"(v), "v"(a), "v"(b));
    return p;
}

// scipy mode='reflect' index (d c b a | a b c d | d c b a) for any offset.
__device__ __forceinline__ int pd_reflect(int i, int n) {
    if (i >= 0 && i < n) return i;
    int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - 1 - i : i;
}

// One bicubic sample of channel ch of a (h, w, C) float32 map at full-resolution
// (Y, X): horizontal 4-tap sums of four source rows, then the vertical 4-tap sum,
// each evaluated left to right in float32 with clamped taps (OpenCV resize order).
__device__ __forceinline__ float pd_bicubic_at(const float* __restrict__ img, int h, int w, int C,
                                               int ch, int Y, int X, const CubicTable& ty,
                                               const CubicTable& tx) {
    const int x0 = tx.ofs[X];
    const int y0 = ty.ofs[Y];
    const float a0 = tx.coef[4 * X + 0], a1 = tx.coef[4 * X + 1], a2 = tx.coef[4 * X + 2],
                a3 = tx.coef[4 * X + 3];
    const int xi0 = pd_clampi(x0, 0, w - 1), xi1 = pd_clampi(x0 + 1, 0, w - 1),
              xi2 = pd_clampi(x0 + 2, 0, w - 1), xi3 = pd_clampi(x0 + 3, 0, w - 1);
    float hv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int s = pd_clampi(y0 + k, 0, h - 1);
        const float* row = img + ((size_t)s * w) * C + ch;
        hv[k] = ((row[(size_t)xi0 * C] * a0 + row[(size_t)xi1 * C] * a1) + row[(size_t)xi2 * C] * a2) +
                row[(size_t)xi3 * C] * a3;
    }
    return ((hv[0] * ty.coef[4 * Y + 0] + hv[1] * ty.coef[4 * Y + 1]) + hv[2] * ty.coef[4 * Y + 2]) +
           hv[3] * ty.coef[4 * Y + 3];
}

// ---- stage launchers (defined in the .hip files) --------------------------------
int pd_launch_smooth_nms(pd_ctx* ctx, double thre1, double* dbg_smoothed);
// finish != 0 (only with only_over): one workgroup per image redoes the listed planes and orders the peaks
int pd_launch_stream_exact(pd_ctx* ctx, double thr, int f32mode, double* dbg_smoothed,
                           const int32_t* only_over, int over_limit, int finish = 0);
// PD_E_STATE when the geometry has no filter tables (caller falls back)
int pd_launch_peaks_filtered(pd_ctx* ctx, double thr);
int pd_launch_peaks_filtered_full(pd_ctx* ctx, double thr);
int pd_launch_sort_peaks(pd_ctx* ctx);
int pd_launch_pairs(pd_ctx* ctx, double thre2);
int pd_launch_match(pd_ctx* ctx);
int pd_launch_parse(pd_ctx* ctx);
int pd_launch_pack(pd_ctx* ctx);
bool pd_parse_pack_fused(const pd_ctx* ctx);
int pd_launch_parse_pack(pd_ctx* ctx);  // person assembly + packing in one launch
int pd_launch_hwc_to_planar(pd_ctx* ctx, const void* src, int dtype, int Csrc, int Cdst, int B, int H,
                            int W, double* dst, float* absmax);
int pd_launch_planar_to_hwc(pd_ctx* ctx, const double* src, int C, int B, int H, int W, double* dst);
int pd_build_tables(pd_ctx* ctx, int h, int w, int H, int W);
void pd_host_cubic_table(int src_len, int dst_len, double scale, int32_t* ofs, float* coef);
// Device tap tables of a resize (hs, ws) -> (Hd, Wd) in a slot of the context; sy / sx <= 0: the
// ratio of the sizes, else the given source step per destination pixel (cv2.resize with fx, fy).
int pd_upload_tables(pd_ctx* ctx, int hs, int ws, int Hd, int Wd, double sy, double sx, TableSlot* slot);
void pd_up8_fill(Up8* u);
int pd_launch_upsample_avg(pd_ctx* ctx, int n_scales, const float* const* srcs, int C, int B, const int32_t* hs,
                           const int32_t* ws, const int32_t* crop_h, const int32_t* crop_w, double* dst, int H, int W,
                           float* absmax);
// Open persons per image a context can be made for: the table form of the person assembly (k_parse) holds the rows of an
// image, 20 doubles each, in one workgroup's LDS (150 KB of the CU's 160).
#define PD_MAX_PERSONS_LIMIT 960

// Issue priority of the calling wavefront (0 lowest .. 3): the short serial kernels of the chain raise theirs, so that
// next to the chip-filling kernels of other contexts their few wavefronts are not served in turn with everybody else.
// Levels (-DPD_SETPRIO=n, comparison builds): 1 = k_finish_peaks, k_match, k_parse_pack; 2 = k_verify as well; 3 = the
// pair kernel too.  Measured at batch 64, 368^2, three alternating runs each: 1.369 M images/s without, 1.372 M with
// level 1, 1.384 M with level 2, 1.347 M with level 3; at 656^2, batch 256: 343 k without, 346 k with level 2.
#ifndef PD_SETPRIO
#define PD_SETPRIO 2
#endif
#define PD_RAISE_PRIORITY(level)                                      \
    do {                                                              \
        if (PD_SETPRIO >= (level)) __builtin_amdgcn_s_setprio(3);     \
    } while (0)

// max over the wavefront of a non-negative float, rounded up one ulp (an infinity stays one), folded
// into *dst (as bits)
__device__ __forceinline__ void pd_wave_absmax(float m, float* dst) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    const unsigned int bits = m <= 3.4028235e38f ? __float_as_uint(m) + 1u : 0x7f800000u;
    // the running maximum only grows: a stale read costs one needless atomic at most
    if ((threadIdx.x & 63) == 0 && m > 0.f && bits > __atomic_load_n((unsigned int*)dst, __ATOMIC_RELAXED))
        atomicMax((unsigned int*)dst, bits);
}
MapSource pd_heat_source(const pd_ctx* ctx);
MapSource pd_paf_source(const pd_ctx* ctx);
// Multi-scale source: writes the float64 average of the heatmaps to full_heat (the peak stage
// then reads it as a full-resolution source).
int pd_ms_materialise_heat(pd_ctx* ctx);
// PD_E_STATE when the geometry has no tables for the sparse multi-scale peak stage (caller falls back)
int pd_launch_peaks_filtered_multi(pd_ctx* ctx, double thr);
