// C ABI of libposedecode: context management, sources, stage entry points, results.
#include <stdlib.h>

#include <new>

#include <vector>

#include "multiscale.h"

#pragma clang fp contract(off)

#include <stdarg.h>

static thread_local char g_err[512] = "";

void pd_set_global_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define PD_CHECK_CTX(ctx) \
    if ((ctx) == nullptr) return PD_E_INVALID;

__global__ void pd_k_clear_words(int32_t* p, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0;
}

int pd_launch_clear(pd_ctx* ctx, void* p, size_t words) {
    if (words == 0) return PD_OK;
    const unsigned blocks = (unsigned)((words + 255) / 256 > 64 ? 64 : (words + 255) / 256);
    hipLaunchKernelGGL(pd_k_clear_words, dim3(blocks), dim3(256), 0, ctx->stream, (int32_t*)p, (int)words);
    PD_HIP(ctx, hipGetLastError());
    return PD_OK;
}

static int dev_alloc(pd_ctx* ctx, void** p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    PD_HIP(ctx, hipMalloc(p, bytes));
    return PD_OK;
}

static int ensure(pd_ctx* ctx, void** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes && *p) return PD_OK;
    PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->gen++;  // a recorded launch sequence may hold the old address
    if (*p) PD_HIP(ctx, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    PD_HIP(ctx, hipMalloc(p, bytes));
    *cap = bytes;
    return PD_OK;
}

extern "C" const char* pd_version(void) { return "posedecode 0.1 (gfx950)"; }

extern "C" const char* pd_last_error(pd_ctx* ctx) { return ctx ? ctx->err : g_err; }

extern "C" int pd_create(const pd_config* cfg, pd_ctx** out) {
    if (!cfg || !out) return PD_E_INVALID;
    *out = nullptr;
    if (cfg->max_batch < 1 || cfg->max_h < 1 || cfg->max_w < 1 || cfg->max_peaks_per_part < 1 ||
        cfg->max_persons < 1 || cfg->max_pairs_per_image < 1) {
        snprintf(g_err, sizeof(g_err), "pd_create: capacities must be positive");
        return PD_E_INVALID;
    }
    if (cfg->max_persons > PD_MAX_PERSONS_LIMIT) {  // the table form of the person assembly keeps an image's rows in LDS
        snprintf(g_err, sizeof(g_err), "pd_create: max_persons %d exceeds %d (160 bytes of LDS per open person)", cfg->max_persons,
                 PD_MAX_PERSONS_LIMIT);
        return PD_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        snprintf(g_err, sizeof(g_err), "pd_create: no HIP device visible");
        return PD_E_HIP;
    }
    if (cfg->device_id < 0 || cfg->device_id >= ndev) {
        snprintf(g_err, sizeof(g_err), "pd_create: device %d of %d", cfg->device_id, ndev);
        return PD_E_INVALID;
    }
    pd_ctx* ctx = new (std::nothrow) pd_ctx;
    if (!ctx) return PD_E_HIP;
    memset(ctx, 0, sizeof(*ctx));
    ctx->cfg = *cfg;
    ctx->mode = PD_MODE_P;
    ctx->impl = PD_IMPL_FAST;
    ctx->use_graph = getenv("PD_NO_GRAPH") ? 0 : 1;
    ctx->host_results = 1;
    // PD_PACK_COPY=1: always a device-to-host copy behind the pack; PD_PACK_DIRECT=1: never
    ctx->pack_policy = getenv("PD_PACK_COPY") ? 0 : (getenv("PD_PACK_DIRECT") ? 1 : -1);
    ctx->pack_direct = ctx->pack_policy != 0;
    ctx->last_peaks_per_image = -1;
    ctx->pairs_pre = 1;
#define CR(call)                                                                     \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != PD_OK) {                                                          \
            snprintf(g_err, sizeof(g_err), "%s", ctx->err);                          \
            pd_destroy(ctx);                                                         \
            return rc_;                                                              \
        }                                                                            \
    } while (0)
#define CRH(call)                                                                    \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            snprintf(g_err, sizeof(g_err), "%s: %s", #call, hipGetErrorString(e_));  \
            pd_destroy(ctx);                                                         \
            return PD_E_HIP;                                                         \
        }                                                                            \
    } while (0)
    CRH(hipSetDevice(cfg->device_id));
    if (pd_static_setup(cfg->device_id, ctx->err, sizeof(ctx->err)) != PD_OK) {
        snprintf(g_err, sizeof(g_err), "%s", ctx->err);
        pd_destroy(ctx);
        return PD_E_HIP;
    }
    CRH(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    const size_t B = cfg->max_batch, PK = cfg->max_peaks_per_part;
    CR(dev_alloc(ctx, (void**)&ctx->raw, B * PD_N_PARTS * PK * sizeof(RawPeak)));
    // counters that start every decode at zero share one allocation and one memset
    ctx->zero_bytes = (2 * B * PD_N_PARTS + 96) * sizeof(int32_t);
    CR(dev_alloc(ctx, (void**)&ctx->zero_block, ctx->zero_bytes));
    CRH(hipMemset(ctx->zero_block, 0, ctx->zero_bytes));
    CR(dev_alloc(ctx, (void**)&ctx->pack_sync, (16 + B) * sizeof(int32_t)));  // arrival counter, tag, a word per image
    CRH(hipMemset(ctx->pack_sync, 0, (16 + B) * sizeof(int32_t)));
    ctx->zero_clean = 1;
    CR(dev_alloc(ctx, (void**)&ctx->full_absmax, B * sizeof(float)));
    CR(dev_alloc(ctx, (void**)&ctx->plane_slack, B * PD_N_PARTS * sizeof(float2)));
    CR(dev_alloc(ctx, (void**)&ctx->fstats, 4 * sizeof(int64_t)));
    CRH(hipMemset(ctx->fstats, 0, 4 * sizeof(int64_t)));
    ctx->raw_count = ctx->zero_block;
    ctx->cand_count = ctx->raw_count + B * PD_N_PARTS;                 // [B][18] + 3 x 32 list counters
    CR(dev_alloc(ctx, (void**)&ctx->pk_y, B * PD_N_PARTS * PK * sizeof(int32_t)));
    CR(dev_alloc(ctx, (void**)&ctx->pk_x, B * PD_N_PARTS * PK * sizeof(int32_t)));
    CR(dev_alloc(ctx, (void**)&ctx->pk_id, B * PD_N_PARTS * PK * sizeof(int32_t)));
    CR(dev_alloc(ctx, (void**)&ctx->pk_score, B * PD_N_PARTS * PK * sizeof(double)));
    CR(dev_alloc(ctx, (void**)&ctx->part_count, B * PD_N_PARTS * sizeof(int32_t)));
    CR(dev_alloc(ctx, (void**)&ctx->part_base, B * (PD_N_PARTS + 1) * sizeof(int32_t)));
    CR(dev_alloc(ctx, (void**)&ctx->pair_off, B * (PD_N_LIMBS + 1) * sizeof(int32_t)));
    CR(dev_alloc(ctx, (void**)&ctx->pair_score, B * (size_t)cfg->max_pairs_per_image * sizeof(double)));
    CR(dev_alloc(ctx, (void**)&ctx->conn, B * PD_N_LIMBS * PK * sizeof(pd_conn)));
    CR(dev_alloc(ctx, (void**)&ctx->conn_conf, B * PD_N_LIMBS * PK * 2 * sizeof(double)));
    CR(dev_alloc(ctx, (void**)&ctx->limb_count, B * PD_N_LIMBS * sizeof(int32_t)));
    CR(dev_alloc(ctx, (void**)&ctx->persons, B * (size_t)cfg->max_persons * 20 * sizeof(double)));
    CR(dev_alloc(ctx, (void**)&ctx->person_count, B * sizeof(int32_t)));
    CR(dev_alloc(ctx, (void**)&ctx->status, B * sizeof(int32_t)));
    {
        const ResLayout lay = pd_res_layout((int)B);
        ctx->res_cap = lay.off_payload + B * PD_N_PARTS * PK * sizeof(pd_peak) + B * PD_N_LIMBS * PK * sizeof(pd_conn) +
                       B * (size_t)cfg->max_persons * 20 * sizeof(double);
        // typical images need a few KB; anything beyond the speculative prefix costs a second copy
        ctx->res_spec = lay.off_payload + 65536 + B * 6144;
        if (ctx->res_spec > ctx->res_cap) ctx->res_spec = ctx->res_cap;
    }
    CR(dev_alloc(ctx, (void**)&ctx->res_dev, ctx->res_cap));
    CRH(hipHostMalloc((void**)&ctx->res_host, ctx->res_cap, hipHostMallocDefault));
    CRH(hipMemset(ctx->status, 0, B * sizeof(int32_t)));
    if (cfg->fullres) {
        const size_t n = (size_t)cfg->max_h * cfg->max_w;
        CR(ensure(ctx, (void**)&ctx->full_heat, &ctx->full_heat_cap, B * PD_CH_HEAT * n * sizeof(double)));
        CR(ensure(ctx, (void**)&ctx->full_paf, &ctx->full_paf_cap, B * PD_CH_PAF * n * sizeof(double)));
    }
#undef CR
#undef CRH
    *out = ctx;
    return PD_OK;
}

static void stop_worker(pd_ctx* ctx) {
    if (!ctx->worker) return;
    {
        std::lock_guard<std::mutex> lock(*ctx->wk_mu);
        ctx->wk_stop = 1;
    }
    ctx->wk_cv->notify_all();
    ctx->worker->join();
    delete ctx->worker;
    delete ctx->wk_jobs;
    delete ctx->wk_cv;
    delete ctx->wk_mu;
    ctx->worker = nullptr;
}

extern "C" int pd_destroy(pd_ctx* ctx) {
    PD_CHECK_CTX(ctx);
    if (ctx->n_sharers > 0) {
        // other contexts still launch on this one's stream: it goes with the last of them
        ctx->destroy_pending = 1;
        return PD_OK;
    }
    pd_ctx* const owner = ctx->owner;
    stop_worker(ctx);
    (void)hipSetDevice(ctx->cfg.device_id);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    void* ptrs[] = {ctx->stage_heat, ctx->stage_paf, ctx->full_heat, ctx->full_paf, ctx->tmp_in,
                    ctx->tmp_smooth, ctx->tmp_smooth2, ctx->raw, ctx->zero_block, ctx->pk_y,
                    ctx->pk_x, ctx->pk_id, ctx->pk_score, ctx->part_count, ctx->part_base, ctx->pair_off,
                    ctx->pair_score, ctx->conn, ctx->conn_conf, ctx->limb_count, ctx->persons,
                    ctx->person_count, ctx->status, ctx->res_dev, ctx->kf_tab, ctx->kf_reg, ctx->cells, ctx->cand,
                    ctx->heat_planar, ctx->paf_pairs, ctx->fstats, ctx->full_absmax, ctx->aux_dev, ctx->dbg_found, ctx->plane_slack,
                    ctx->msf_tab, ctx->msf_lo, ctx->ms_planar, ctx->ms_dev, ctx->pack_sync};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (TableSlot& t : ctx->tabs) {
        if (t.i) (void)hipFree(t.i);
        if (t.f) (void)hipFree(t.f);
    }
    for (GraphSlot& g : ctx->graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (ctx->res_host) (void)hipHostFree(ctx->res_host);
    for (int i = 0; i < 2 * ctx->prof_cap; ++i) (void)hipEventDestroy(ctx->prof_ev[i]);
    free(ctx->prof_ev);
    free(ctx->prof_stage);
    if (ctx->done_ev) (void)hipEventDestroy(ctx->done_ev);
    if (ctx->stream && !ctx->stream_shared) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    if (owner && --owner->n_sharers == 0 && owner->destroy_pending) return pd_destroy(owner);
    return PD_OK;
}

extern "C" int pd_set_mode(pd_ctx* ctx, int mode) {
    PD_CHECK_CTX(ctx);
    if (mode != PD_MODE_P && mode != PD_MODE_D) PD_FAIL(ctx, PD_E_INVALID, "pd_set_mode: mode %d", mode);
    ctx->mode = mode;
    return PD_OK;
}

extern "C" int pd_set_impl(pd_ctx* ctx, int impl) {
    PD_CHECK_CTX(ctx);
    if (impl != PD_IMPL_FAST && impl != PD_IMPL_SIMPLE && impl != PD_IMPL_STREAM) PD_FAIL(ctx, PD_E_INVALID, "pd_set_impl: impl %d", impl);
    ctx->impl = impl;
    return PD_OK;
}

extern "C" void* pd_stream(pd_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

MapSource pd_heat_source(const pd_ctx* ctx) {
    MapSource s;
    memset(&s, 0, sizeof(s));
    s.kind = ctx->src_kind;
    if (ctx->src_kind == PD_SRC_MULTI && ctx->ms_heat_full) s.kind = PD_SRC_FULLRES;
    s.h = ctx->h;
    s.w = ctx->w;
    s.H = ctx->H;
    s.W = ctx->W;
    s.lo = ctx->lo_heat;
    s.C = PD_CH_HEAT;
    s.full = ctx->full_heat;
    s.Cfull = PD_CH_HEAT;
    s.ty = ctx->tabs[0].ty;
    s.tx = ctx->tabs[0].tx;
    pd_up8_fill(&s.up8);
    return s;
}

MapSource pd_paf_source(const pd_ctx* ctx) {
    MapSource s = pd_heat_source(ctx);
    s.kind = ctx->src_kind;
    s.lo = ctx->lo_paf;
    s.C = PD_CH_PAF;
    s.full = ctx->full_paf;
    s.Cfull = PD_CH_PAF;
    return s;
}

static void reset_results(pd_ctx* ctx) {
    ctx->have_peaks = ctx->have_conns = ctx->have_persons = ctx->packed = 0;
    ctx->done_valid = 0;
}

// On a stream that carries two contexts a context waits for its own work through an event, so
// that it does not also wait for what the other context has queued behind it.  The event is
// recorded behind a whole decode; work queued any other way (the staged entry points) makes it
// stale, and the wait then falls back to the stream.
static inline bool waits_by_event(const pd_ctx* ctx) { return ctx->stream_shared || ctx->n_sharers > 0; }
static int record_done(pd_ctx* ctx) {
    if (!waits_by_event(ctx)) return PD_OK;
    if (!ctx->done_ev) PD_HIP(ctx, hipEventCreateWithFlags(&ctx->done_ev, hipEventDisableTiming));
    PD_HIP(ctx, hipEventRecord(ctx->done_ev, ctx->stream));
    ctx->done_valid = 1;
    return PD_OK;
}

#define PD_PACK_DIRECT_LIMIT ((size_t)1 << 20)  // bytes of packed results up to which the pack kernel writes the host mirror

static int check_batch(pd_ctx* ctx, int B, int H, int W, const char* who) {
    if (B < 1 || B > ctx->cfg.max_batch) PD_FAIL(ctx, PD_E_INVALID, "%s: batch %d exceeds max_batch %d", who, B, ctx->cfg.max_batch);
    if (H < 1 || W < 1 || H > ctx->cfg.max_h || W > ctx->cfg.max_w)
        PD_FAIL(ctx, PD_E_INVALID, "%s: image %dx%d exceeds the context's %dx%d", who, H, W, ctx->cfg.max_h, ctx->cfg.max_w);
    return PD_OK;
}

extern "C" int pd_set_lowres(pd_ctx* ctx, const float* heat, const float* paf, int B, int h, int w, int H,
                             int W, int mem) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    int rc = check_batch(ctx, B, H, W, "pd_set_lowres");
    if (rc != PD_OK) return rc;
    if (h < 1 || w < 1 || H > 8 * h || W > 8 * w || H <= 8 * (h - 1) || W <= 8 * (w - 1))
        PD_FAIL(ctx, PD_E_INVALID, "pd_set_lowres: %dx%d is not a crop of the x8 map of %dx%d", H, W, h, w);
    if (!heat && !paf) PD_FAIL(ctx, PD_E_INVALID, "pd_set_lowres: no input");
    const size_t nh = (size_t)B * h * w * PD_CH_HEAT, np = (size_t)B * h * w * PD_CH_PAF;
    if (mem == PD_MEM_HOST) {
        if (heat) {
            rc = ensure(ctx, (void**)&ctx->stage_heat, &ctx->stage_heat_cap, nh * sizeof(float));
            if (rc != PD_OK) return rc;
            PD_HIP(ctx, hipMemcpyAsync(ctx->stage_heat, heat, nh * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
            ctx->lo_heat = ctx->stage_heat;
        }
        if (paf) {
            rc = ensure(ctx, (void**)&ctx->stage_paf, &ctx->stage_paf_cap, np * sizeof(float));
            if (rc != PD_OK) return rc;
            PD_HIP(ctx, hipMemcpyAsync(ctx->stage_paf, paf, np * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
            ctx->lo_paf = ctx->stage_paf;
        }
    } else {
        // the kernels read the maps with 8- and 16-byte loads
        if (((uintptr_t)heat | (uintptr_t)paf) & 15)
            PD_FAIL(ctx, PD_E_INVALID, "pd_set_lowres: device arrays must be 16-byte aligned");
        if (heat) ctx->lo_heat = heat;
        if (paf) ctx->lo_paf = paf;
    }
    rc = pd_build_tables(ctx, h, w, H, W);
    if (rc != PD_OK) return rc;
    ctx->src_kind = PD_SRC_LOWRES;
    ctx->B = B;
    ctx->h = h;
    ctx->w = w;
    ctx->H = H;
    ctx->W = W;
    ctx->heat_f32 = 0;
    reset_results(ctx);
    return PD_OK;
}

static int ensure_full(pd_ctx* ctx, int B, int H, int W, bool heat, bool paf) {
    const size_t n = (size_t)H * W;
    int rc;
    if (heat) {
        rc = ensure(ctx, (void**)&ctx->full_heat, &ctx->full_heat_cap, (size_t)B * PD_CH_HEAT * n * sizeof(double));
        if (rc != PD_OK) return rc;
    }
    if (paf) {
        rc = ensure(ctx, (void**)&ctx->full_paf, &ctx->full_paf_cap, (size_t)B * PD_CH_PAF * n * sizeof(double));
        if (rc != PD_OK) return rc;
    }
    return PD_OK;
}

extern "C" int pd_load_fullres(pd_ctx* ctx, const void* heat, int heat_dtype, int heat_channels,
                               const void* paf, int paf_dtype, int paf_channels, int B, int H, int W, int mem) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    int rc = check_batch(ctx, B, H, W, "pd_load_fullres");
    if (rc != PD_OK) return rc;
    if (heat && heat_channels < PD_N_PARTS) PD_FAIL(ctx, PD_E_INVALID, "pd_load_fullres: heatmap needs >= 18 channels, got %d", heat_channels);
    if (paf && paf_channels < PD_CH_PAF) PD_FAIL(ctx, PD_E_INVALID, "pd_load_fullres: pafs need >= 38 channels, got %d", paf_channels);
    if (ctx->src_kind == PD_SRC_FULLRES && (ctx->B != B || ctx->H != H || ctx->W != W)) {
        ctx->have_full_heat = ctx->have_full_paf = 0;
    }
    rc = ensure_full(ctx, B, H, W, heat != nullptr, paf != nullptr);
    if (rc != PD_OK) return rc;
    const size_t n = (size_t)H * W;
    for (int which = 0; which < 2; ++which) {
        const void* src = which == 0 ? heat : paf;
        if (!src) continue;
        const int dtype = which == 0 ? heat_dtype : paf_dtype;
        const int C = which == 0 ? heat_channels : paf_channels;
        const int Cdst = which == 0 ? PD_CH_HEAT : PD_CH_PAF;
        double* dst = which == 0 ? ctx->full_heat : ctx->full_paf;
        const size_t bytes = (size_t)B * n * C * (dtype == PD_F32 ? 4 : 8);
        const void* dsrc = src;
        if (mem == PD_MEM_HOST) {
            rc = ensure(ctx, &ctx->tmp_in, &ctx->tmp_in_cap, bytes);
            if (rc != PD_OK) return rc;
            PD_HIP(ctx, hipMemcpyAsync(ctx->tmp_in, src, bytes, hipMemcpyHostToDevice, ctx->stream));
            dsrc = ctx->tmp_in;
        }
        float* amax = nullptr;
        if (which == 0) {  // the candidate filter of the peak stage needs a bound on the heatmaps
            rc = pd_launch_clear(ctx, ctx->full_absmax, (size_t)B);
            if (rc != PD_OK) return rc;
            amax = ctx->full_absmax;
        }
        rc = pd_launch_hwc_to_planar(ctx, dsrc, dtype, C, Cdst, B, H, W, dst, amax);
        if (rc != PD_OK) return rc;
        if (which == 0) {
            ctx->full_absmax_ok = 1;
            ctx->heat_f32 = dtype == PD_F32;
            ctx->have_full_heat = 1;
        } else {
            ctx->have_full_paf = 1;
        }
    }
    ctx->src_kind = PD_SRC_FULLRES;
    ctx->B = B;
    ctx->H = H;
    ctx->W = W;
    ctx->h = ctx->w = 0;
    reset_results(ctx);
    return PD_OK;
}

extern "C" int pd_upsample_avg(pd_ctx* ctx, int n_scales, const float* const* heat, const float* const* paf,
                               const int32_t* hs, const int32_t* ws, const int32_t* crop_h,
                               const int32_t* crop_w, int H, int W, int B, int mem) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    int rc = check_batch(ctx, B, H, W, "pd_upsample_avg");
    if (rc != PD_OK) return rc;
    if (n_scales < 1 || !heat || !paf || !hs || !ws || !crop_h || !crop_w) PD_FAIL(ctx, PD_E_INVALID, "pd_upsample_avg: bad arguments");
    for (int m = 0; m < n_scales; ++m) {
        if (hs[m] < 1 || ws[m] < 1 || crop_h[m] < 1 || crop_w[m] < 1 || crop_h[m] > 8 * hs[m] || crop_w[m] > 8 * ws[m])
            PD_FAIL(ctx, PD_E_INVALID, "pd_upsample_avg: scale %d: crop %dx%d of x8 map of %dx%d", m, crop_h[m], crop_w[m], hs[m], ws[m]);
    }
    if (n_scales > PD_MAX_SCALES) PD_FAIL(ctx, PD_E_INVALID, "pd_upsample_avg: at most %d scales", PD_MAX_SCALES);
    rc = ensure_full(ctx, B, H, W, true, true);
    if (rc != PD_OK) return rc;
    size_t n_lo_all = 0;
    for (int m = 0; m < n_scales; ++m) n_lo_all += (size_t)B * hs[m] * ws[m];
    for (int which = 0; which < 2; ++which) {
        const int C = which == 0 ? PD_CH_HEAT : PD_CH_PAF;
        const float* const* given = which == 0 ? heat : paf;
        double* dst = which == 0 ? ctx->full_heat : ctx->full_paf;
        const float* srcs[PD_MAX_SCALES];
        if (mem == PD_MEM_HOST) {
            float** st = which == 0 ? &ctx->stage_heat : &ctx->stage_paf;
            size_t* cap = which == 0 ? &ctx->stage_heat_cap : &ctx->stage_paf_cap;
            rc = ensure(ctx, (void**)st, cap, n_lo_all * C * sizeof(float));
            if (rc != PD_OK) return rc;
            size_t off = 0;
            for (int m = 0; m < n_scales; ++m) {
                const size_t cnt = (size_t)B * hs[m] * ws[m] * C;
                PD_HIP(ctx, hipMemcpyAsync(*st + off, given[m], cnt * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
                srcs[m] = *st + off;
                off += cnt;
            }
        } else {
            for (int m = 0; m < n_scales; ++m) srcs[m] = given[m];
        }
        float* amax = nullptr;
        if (which == 0) {
            rc = pd_launch_clear(ctx, ctx->full_absmax, (size_t)B);
            if (rc != PD_OK) return rc;
            amax = ctx->full_absmax;
        }
        rc = pd_launch_upsample_avg(ctx, n_scales, srcs, C, B, hs, ws, crop_h, crop_w, dst, H, W, amax);
        if (rc != PD_OK) return rc;
    }
    ctx->src_kind = PD_SRC_FULLRES;
    ctx->B = B;
    ctx->H = H;
    ctx->W = W;
    ctx->h = ctx->w = 0;
    ctx->heat_f32 = 0;
    ctx->full_absmax_ok = 1;
    ctx->have_full_heat = ctx->have_full_paf = 1;
    reset_results(ctx);
    return PD_OK;
}

MultiSource pd_multi_source(const pd_ctx* ctx) {
    MultiSource ms;
    memset(&ms, 0, sizeof(ms));
    ms.n = ctx->ms_n;
    ms.pow2 = (ctx->ms_n & (ctx->ms_n - 1)) == 0;
    ms.H = ctx->H;
    ms.W = ctx->W;
    ms.n_div = (float)ctx->ms_n;
    ms.n_rcp = ms.pow2 ? 1.0f / (float)ctx->ms_n : 0.f;
    for (int m = 0; m < ctx->ms_n; ++m) {
        MultiSrc& sc = ms.s[m];
        sc.heat = ctx->ms_heat[m];
        sc.paf = ctx->ms_paf[m];
        sc.hs = ctx->ms_hs[m];
        sc.ws = ctx->ms_ws[m];
        sc.crop_h = ctx->ms_crop_h[m];
        sc.crop_w = ctx->ms_crop_w[m];
        sc.identity = sc.crop_h == ctx->H && sc.crop_w == ctx->W;
        if (!sc.identity) {
            sc.ty = ctx->tabs[2 + m].ty;
            sc.tx = ctx->tabs[2 + m].tx;
        }
    }
    return ms;
}

// The network outputs of every test scale as the source of the decode: the averaged maps of
// predict.py:51-72 are evaluated where the stages need them and are not stored.
extern "C" int pd_set_multiscale(pd_ctx* ctx, int n_scales, const float* const* heat, const float* const* paf,
                                 const int32_t* hs, const int32_t* ws, const int32_t* crop_h, const int32_t* crop_w,
                                 int H, int W, int B, int mem) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    int rc = check_batch(ctx, B, H, W, "pd_set_multiscale");
    if (rc != PD_OK) return rc;
    if (n_scales < 1 || !heat || !paf || !hs || !ws || !crop_h || !crop_w) PD_FAIL(ctx, PD_E_INVALID, "pd_set_multiscale: bad arguments");
    if (n_scales > PD_MAX_SCALES) PD_FAIL(ctx, PD_E_INVALID, "pd_set_multiscale: at most %d scales", PD_MAX_SCALES);
    for (int m = 0; m < n_scales; ++m) {
        if (!heat[m] || !paf[m]) PD_FAIL(ctx, PD_E_INVALID, "pd_set_multiscale: scale %d: heat and paf are required", m);
        if (hs[m] < 1 || ws[m] < 1 || crop_h[m] < 1 || crop_w[m] < 1 || crop_h[m] > 8 * hs[m] || crop_w[m] > 8 * ws[m])
            PD_FAIL(ctx, PD_E_INVALID, "pd_set_multiscale: scale %d: crop %dx%d of x8 map of %dx%d", m, crop_h[m], crop_w[m], hs[m], ws[m]);
        if (mem == PD_MEM_DEVICE && (((uintptr_t)heat[m] | (uintptr_t)paf[m]) & 15))
            PD_FAIL(ctx, PD_E_INVALID, "pd_set_multiscale: device arrays must be 16-byte aligned");
    }
    ctx->done_valid = 0;
    if (mem == PD_MEM_HOST) {
        // (every scale's array starts 16-byte aligned in the staging area)
        size_t n_lo_all = 0;
        for (int m = 0; m < n_scales; ++m) n_lo_all += ((size_t)B * hs[m] * ws[m] + 3) & ~(size_t)3;
        rc = ensure(ctx, (void**)&ctx->stage_heat, &ctx->stage_heat_cap, n_lo_all * PD_CH_HEAT * sizeof(float));
        if (rc != PD_OK) return rc;
        rc = ensure(ctx, (void**)&ctx->stage_paf, &ctx->stage_paf_cap, n_lo_all * PD_CH_PAF * sizeof(float));
        if (rc != PD_OK) return rc;
        size_t off = 0;
        for (int m = 0; m < n_scales; ++m) {
            const size_t px = (size_t)B * hs[m] * ws[m];
            PD_HIP(ctx, hipMemcpyAsync(ctx->stage_heat + off * PD_CH_HEAT, heat[m], px * PD_CH_HEAT * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
            PD_HIP(ctx, hipMemcpyAsync(ctx->stage_paf + off * PD_CH_PAF, paf[m], px * PD_CH_PAF * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
            ctx->ms_heat[m] = ctx->stage_heat + off * PD_CH_HEAT;
            ctx->ms_paf[m] = ctx->stage_paf + off * PD_CH_PAF;
            off += (px + 3) & ~(size_t)3;
        }
    } else {
        for (int m = 0; m < n_scales; ++m) {
            ctx->ms_heat[m] = heat[m];
            ctx->ms_paf[m] = paf[m];
        }
    }
    for (int m = 0; m < n_scales; ++m) {
        ctx->ms_hs[m] = hs[m];
        ctx->ms_ws[m] = ws[m];
        ctx->ms_crop_h[m] = crop_h[m];
        ctx->ms_crop_w[m] = crop_w[m];
        if (crop_h[m] != H || crop_w[m] != W) {
            rc = pd_upload_tables(ctx, crop_h[m], crop_w[m], H, W, -1.0, -1.0, &ctx->tabs[2 + m]);
            if (rc != PD_OK) return rc;
        }
    }
    ctx->ms_n = n_scales;
    ctx->ms_heat_full = 0;
    ctx->src_kind = PD_SRC_MULTI;
    ctx->B = B;
    ctx->H = H;
    ctx->W = W;
    ctx->h = ctx->w = 0;
    ctx->heat_f32 = 0;
    ctx->have_full_heat = ctx->have_full_paf = 0;
    reset_results(ctx);
    {
        // the scales also live in device memory, for the kernel that reads them through a pointer;
        // rewritten (stream-ordered) only when they change
        static_assert(sizeof(MultiSource) <= sizeof(ctx->ms_dev_image), "ms_dev_image holds a MultiSource");
        const MultiSource ms = pd_multi_source(ctx);
        if (!ctx->ms_dev) {
            PD_HIP(ctx, hipMalloc(&ctx->ms_dev, sizeof(ctx->ms_dev_image)));
            memset(ctx->ms_dev_image, 0xff, sizeof(ctx->ms_dev_image));
        }
        if (memcmp(&ms, ctx->ms_dev_image, sizeof(ms)) != 0) {
            memcpy(ctx->ms_dev_image, &ms, sizeof(ms));
            PD_HIP(ctx, hipMemcpyAsync(ctx->ms_dev, ctx->ms_dev_image, sizeof(ms), hipMemcpyHostToDevice, ctx->stream));
        }
    }
    return PD_OK;
}

int pd_ms_materialise_heat(pd_ctx* ctx) {
    if (ctx->src_kind != PD_SRC_MULTI) PD_FAIL(ctx, PD_E_STATE, "no multi-scale source set");
    if (ctx->ms_heat_full) return PD_OK;
    int rc = ensure_full(ctx, ctx->B, ctx->H, ctx->W, true, false);
    if (rc != PD_OK) return rc;
    rc = pd_launch_clear(ctx, ctx->full_absmax, (size_t)ctx->B);
    if (rc != PD_OK) return rc;
    rc = pd_launch_upsample_avg(ctx, ctx->ms_n, ctx->ms_heat, PD_CH_HEAT, ctx->B, ctx->ms_hs, ctx->ms_ws, ctx->ms_crop_h,
                                ctx->ms_crop_w, ctx->full_heat, ctx->H, ctx->W, ctx->full_absmax);
    if (rc != PD_OK) return rc;
    ctx->full_absmax_ok = 1;
    ctx->have_full_heat = 1;
    ctx->ms_heat_full = 1;
    return PD_OK;
}

extern "C" int pd_read_fullres(pd_ctx* ctx, double* heat_out, double* paf_out) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    if (ctx->src_kind != PD_SRC_FULLRES) PD_FAIL(ctx, PD_E_STATE, "pd_read_fullres: no full-resolution maps in the context");
    const size_t n = (size_t)ctx->H * ctx->W;
    for (int which = 0; which < 2; ++which) {
        double* out = which == 0 ? heat_out : paf_out;
        if (!out) continue;
        const int C = which == 0 ? PD_CH_HEAT : PD_CH_PAF;
        const double* src = which == 0 ? ctx->full_heat : ctx->full_paf;
        const size_t bytes = (size_t)ctx->B * n * C * sizeof(double);
        int rc = ensure(ctx, &ctx->tmp_in, &ctx->tmp_in_cap, bytes);
        if (rc != PD_OK) return rc;
        rc = pd_launch_planar_to_hwc(ctx, src, C, ctx->B, ctx->H, ctx->W, (double*)ctx->tmp_in);
        if (rc != PD_OK) return rc;
        PD_HIP(ctx, hipMemcpyAsync(out, ctx->tmp_in, bytes, hipMemcpyDeviceToHost, ctx->stream));
        PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return PD_OK;
}

extern "C" int pd_peaks(pd_ctx* ctx, double thre1) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    if (ctx->src_kind == PD_SRC_NONE) PD_FAIL(ctx, PD_E_STATE, "pd_peaks: no source set");
    if (ctx->src_kind == PD_SRC_LOWRES && !ctx->lo_heat) PD_FAIL(ctx, PD_E_STATE, "pd_peaks: no heatmaps set");
    if (ctx->src_kind == PD_SRC_FULLRES && !ctx->have_full_heat) PD_FAIL(ctx, PD_E_STATE, "pd_peaks: no heatmaps loaded");
    ctx->done_valid = 0;
    int rc = pd_launch_smooth_nms(ctx, thre1, nullptr);
    if (rc != PD_OK) return rc;
    rc = pd_launch_sort_peaks(ctx);
    if (rc != PD_OK) return rc;
    ctx->have_peaks = 1;
    ctx->have_conns = ctx->have_persons = ctx->packed = 0;
    return PD_OK;
}

extern "C" int pd_debug_smoothed(pd_ctx* ctx, double* out) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    if (ctx->src_kind == PD_SRC_NONE || !out) PD_FAIL(ctx, PD_E_STATE, "pd_debug_smoothed: no source set");
    const size_t bytes = (size_t)ctx->B * PD_N_PARTS * ctx->H * ctx->W * sizeof(double);
    double* dev = nullptr;
    PD_HIP(ctx, hipMalloc(&dev, bytes));
    int rc = pd_launch_smooth_nms(ctx, 1e300, dev);
    if (rc == PD_OK) {
        hipError_t e = hipMemcpyAsync(out, dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            snprintf(ctx->err, sizeof(ctx->err), "pd_debug_smoothed: %s", hipGetErrorString(e));
            rc = PD_E_HIP;
        }
    }
    (void)hipFree(dev);
    return rc;
}

// Destroys every recorded launch sequence (the stream is drained first).
static int drop_graphs(pd_ctx* ctx) {
    bool any = false;
    for (GraphSlot& g : ctx->graphs) any = any || g.exec != nullptr;
    if (any) {
        PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
        PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    for (GraphSlot& g : ctx->graphs) {
        if (g.exec) PD_HIP(ctx, hipGraphExecDestroy(g.exec));
        memset(&g, 0, sizeof(g));
    }
    return PD_OK;
}

extern "C" int pd_set_graph_replay(pd_ctx* ctx, int enable) {
    PD_CHECK_CTX(ctx);
    ctx->use_graph = enable ? 1 : 0;
    if (!enable) return drop_graphs(ctx);
    return PD_OK;
}

extern "C" int pd_set_skeleton_buffer(pd_ctx* ctx, void* device_ptr, uint64_t capacity) {
    PD_CHECK_CTX(ctx);
    if (device_ptr) {
        if ((uintptr_t)device_ptr & 15) PD_FAIL(ctx, PD_E_INVALID, "pd_set_skeleton_buffer: the buffer must be 16-byte aligned");
        const size_t least = pd_res_layout(ctx->cfg.max_batch).off_payload;
        if (capacity < least)
            PD_FAIL(ctx, PD_E_INVALID, "pd_set_skeleton_buffer: %llu bytes cannot hold the count tables of %d images (%zu bytes)",
                    (unsigned long long)capacity, ctx->cfg.max_batch, least);
    }
    // pointer and capacity are part of the key a recorded launch sequence is replayed under
    ctx->skel_dst = (unsigned char*)device_ptr;
    ctx->skel_cap = device_ptr ? (size_t)capacity : 0;
    return PD_OK;
}

extern "C" int pd_set_host_results(pd_ctx* ctx, int enable) {
    PD_CHECK_CTX(ctx);
    ctx->host_results = enable ? 1 : 0;
    return PD_OK;
}

// With a skeleton buffer set and host results switched off, a pack writes the skeleton
// buffer only.
static inline bool skeleton_only(const pd_ctx* ctx) { return ctx->skel_dst != nullptr && !ctx->host_results; }

extern "C" int pd_filter_stats(pd_ctx* ctx, int64_t* out) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    if (!out) PD_FAIL(ctx, PD_E_INVALID, "pd_filter_stats: out is NULL");
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!ctx->fstats) return PD_OK;
    PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PD_HIP(ctx, hipMemcpy(out, ctx->fstats, 4 * sizeof(int64_t), hipMemcpyDeviceToHost));
    return PD_OK;
}

extern "C" int pd_set_peaks(pd_ctx* ctx, const int32_t* counts, const int32_t* y, const int32_t* x,
                            const double* score, const int32_t* id) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    if (!counts) PD_FAIL(ctx, PD_E_INVALID, "pd_set_peaks: counts is NULL");
    if (ctx->src_kind == PD_SRC_NONE) PD_FAIL(ctx, PD_E_STATE, "pd_set_peaks: set a source first (it fixes the image size)");
    const int PK = ctx->cfg.max_peaks_per_part;
    int32_t base[PD_N_PARTS + 1], pair_off[PD_N_LIMBS + 1];
    int acc = 0;
    for (int p = 0; p < PD_N_PARTS; ++p) {
        if (counts[p] < 0) PD_FAIL(ctx, PD_E_INVALID, "pd_set_peaks: negative count");
        if (counts[p] > PK) PD_FAIL(ctx, PD_E_OVERFLOW, "pd_set_peaks: %d peaks of part %d exceed max_peaks_per_part %d", counts[p], p, PK);
        base[p] = acc;
        acc += counts[p];
    }
    base[PD_N_PARTS] = acc;
    long long pairs = 0;
    for (int k = 0; k < PD_N_LIMBS; ++k) {
        pair_off[k] = (int32_t)pairs;
        pairs += (long long)counts[pd_limb_from(k)] * counts[pd_limb_to(k)];
        if (pairs > ctx->cfg.max_pairs_per_image)
            PD_FAIL(ctx, PD_E_OVERFLOW, "pd_set_peaks: %lld candidate pairs exceed max_pairs_per_image %d", pairs, ctx->cfg.max_pairs_per_image);
    }
    pair_off[PD_N_LIMBS] = (int32_t)pairs;
    PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int p = 0; p < PD_N_PARTS; ++p) {
        const size_t o = (size_t)p * PK, n = counts[p];
        if (n == 0) continue;
        PD_HIP(ctx, hipMemcpy(ctx->pk_y + o, y + base[p], n * sizeof(int32_t), hipMemcpyHostToDevice));
        PD_HIP(ctx, hipMemcpy(ctx->pk_x + o, x + base[p], n * sizeof(int32_t), hipMemcpyHostToDevice));
        PD_HIP(ctx, hipMemcpy(ctx->pk_score + o, score + base[p], n * sizeof(double), hipMemcpyHostToDevice));
        PD_HIP(ctx, hipMemcpy(ctx->pk_id + o, id + base[p], n * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    PD_HIP(ctx, hipMemcpy(ctx->part_count, counts, sizeof(int32_t) * PD_N_PARTS, hipMemcpyHostToDevice));
    PD_HIP(ctx, hipMemcpy(ctx->part_base, base, sizeof(base), hipMemcpyHostToDevice));
    PD_HIP(ctx, hipMemcpy(ctx->pair_off, pair_off, sizeof(pair_off), hipMemcpyHostToDevice));
    PD_HIP(ctx, hipMemset(ctx->status, 0, sizeof(int32_t)));
    ctx->B = 1;
    ctx->have_peaks = 1;
    ctx->have_conns = ctx->have_persons = ctx->packed = 0;
    return PD_OK;
}

extern "C" int pd_connections(pd_ctx* ctx, double thre2) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    if (!ctx->have_peaks) PD_FAIL(ctx, PD_E_STATE, "pd_connections: no peaks (run pd_peaks or pd_set_peaks)");
    if (ctx->src_kind == PD_SRC_LOWRES && !ctx->lo_paf) PD_FAIL(ctx, PD_E_STATE, "pd_connections: no PAFs set");
    if (ctx->src_kind == PD_SRC_FULLRES && !ctx->have_full_paf) PD_FAIL(ctx, PD_E_STATE, "pd_connections: no PAFs loaded");
    ctx->done_valid = 0;
    int rc = pd_launch_pairs(ctx, thre2);
    if (rc != PD_OK) return rc;
    rc = pd_launch_match(ctx);
    if (rc != PD_OK) return rc;
    ctx->have_conns = 1;
    ctx->have_persons = ctx->packed = 0;
    return PD_OK;
}

extern "C" int pd_set_connections(pd_ctx* ctx, const int32_t* counts, const double* rows) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    if (!counts) PD_FAIL(ctx, PD_E_INVALID, "pd_set_connections: counts is NULL");
    if (!rows)
        for (int k = 0; k < PD_N_LIMBS; ++k)
            if (counts[k] > 0) PD_FAIL(ctx, PD_E_INVALID, "pd_set_connections: rows is NULL but limb %d has %d connections", k, counts[k]);
    const int PK = ctx->cfg.max_peaks_per_part;
    PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    size_t at = 0;
    for (int k = 0; k < PD_N_LIMBS; ++k) {
        const int n = counts[k];
        if (n > PK) PD_FAIL(ctx, PD_E_OVERFLOW, "pd_set_connections: %d connections of limb %d exceed max_peaks_per_part %d", n, k, PK);
        if (n <= 0) continue;
        pd_conn* tmp = (pd_conn*)malloc(sizeof(pd_conn) * n);
        double* cf = (double*)malloc(sizeof(double) * 2 * n);
        if (!tmp || !cf) {
            free(tmp);
            free(cf);
            PD_FAIL(ctx, PD_E_INVALID, "pd_set_connections: out of host memory");
        }
        for (int i = 0; i < n; ++i) {
            const double* r = rows + (at + i) * 5;
            tmp[i].id_a = (int32_t)r[0];  // int(connection[0]), utils/postprocess.py:188
            tmp[i].id_b = (int32_t)r[1];
            tmp[i].ia = tmp[i].ib = -1;
            tmp[i].score = r[2];
            cf[2 * i] = r[3];
            cf[2 * i + 1] = r[4];
        }
        hipError_t e1 = hipMemcpy(ctx->conn + (size_t)k * PK, tmp, sizeof(pd_conn) * n, hipMemcpyHostToDevice);
        hipError_t e2 = hipMemcpy(ctx->conn_conf + (size_t)k * PK * 2, cf, sizeof(double) * 2 * n, hipMemcpyHostToDevice);
        free(tmp);
        free(cf);
        PD_HIP(ctx, e1);
        PD_HIP(ctx, e2);
        at += n;
    }
    PD_HIP(ctx, hipMemcpy(ctx->limb_count, counts, sizeof(int32_t) * PD_N_LIMBS, hipMemcpyHostToDevice));
    PD_HIP(ctx, hipMemset(ctx->status, 0, sizeof(int32_t)));
    ctx->B = 1;
    ctx->have_conns = 1;
    ctx->have_persons = ctx->packed = 0;
    return PD_OK;
}

extern "C" int pd_parse(pd_ctx* ctx) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    if (!ctx->have_conns) PD_FAIL(ctx, PD_E_STATE, "pd_parse: no connections (run pd_connections or pd_set_connections)");
    ctx->done_valid = 0;
    int rc = pd_launch_parse(ctx);
    if (rc != PD_OK) return rc;
    ctx->have_persons = 1;
    ctx->packed = 0;
    return PD_OK;
}

extern "C" int pd_debug_found(pd_ctx* ctx, int enable, int32_t* out) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    const size_t bytes = (size_t)ctx->cfg.max_batch * ctx->cfg.max_peaks_per_part * sizeof(int32_t);
    if (out) {
        if (!ctx->dbg_found) PD_FAIL(ctx, PD_E_STATE, "pd_debug_found: recording is off");
        PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PD_HIP(ctx, hipMemcpy(out, ctx->dbg_found, bytes, hipMemcpyDeviceToHost));
    }
    if (enable && !ctx->dbg_found) {
        PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->gen++;
        PD_HIP(ctx, hipMalloc((void**)&ctx->dbg_found, bytes));
        PD_HIP(ctx, hipMemset(ctx->dbg_found, 0, bytes));
    } else if (!enable && ctx->dbg_found) {
        PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->gen++;
        PD_HIP(ctx, hipFree(ctx->dbg_found));
        ctx->dbg_found = nullptr;
    }
    return PD_OK;
}

static int after_pack(pd_ctx* ctx);

extern "C" int pd_pack(pd_ctx* ctx) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    if (!ctx->have_peaks && !ctx->have_conns && !ctx->have_persons) PD_FAIL(ctx, PD_E_STATE, "pd_pack: nothing to pack");
    ctx->done_valid = 0;
    int rc = pd_launch_pack(ctx);
    if (rc != PD_OK) return rc;
    return after_pack(ctx);
}

// Host side of a queued pack: what of the results is (or will be) in the host mirror.
static int after_pack(pd_ctx* ctx) {
    if (skeleton_only(ctx)) {  // nothing for pd_fetch and its like: they report PD_E_STATE
        ctx->res_have = 0;
        ctx->packed = 0;
        return PD_OK;
    }
    if (ctx->pack_direct) {  // the pack kernel has stored everything to the host mirror as well
        ctx->res_have = ctx->res_cap;
        ctx->packed = 1;
        return PD_OK;
    }
    // header, count tables and (speculatively) the head of the payload in one copy
    const ResLayout lay = pd_res_layout(ctx->B);
    size_t spec = lay.off_payload + 65536 + (size_t)ctx->B * 6144;
    if (spec > ctx->res_cap) spec = ctx->res_cap;
    PD_HIP(ctx, hipMemcpyAsync(ctx->res_host, ctx->res_dev, spec, hipMemcpyDeviceToHost, ctx->stream));
    ctx->res_have = spec;
    ctx->packed = 1;
    return PD_OK;
}

// Folds the event pairs recorded since the last harvest into the per-stage sums.
// The stream must be idle.
static int prof_harvest(pd_ctx* ctx) {
    for (int i = 0; i < ctx->prof_n; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]) != hipSuccess) {
            // events replayed from a graph could not be read on this runtime: drop the
            // sample and go back to eager launches, which always can
            (void)hipGetLastError();
            ctx->use_graph = 0;
            (void)drop_graphs(ctx);
            break;
        }
        const int s = ctx->prof_stage[i];
        ctx->prof_ms[s] += (double)t;
        ctx->prof_cnt[s] += 1;
    }
    ctx->prof_n = 0;
    return PD_OK;
}

// Waits for the packed results and completes the host mirror.
static int finish_results(pd_ctx* ctx) {
    // (on a stream that carries two contexts: for this context's last decode only, not for what
    // the other context has queued behind it)
    if (waits_by_event(ctx) && ctx->done_valid)
        PD_HIP(ctx, hipEventSynchronize(ctx->done_ev));
    else
        PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->prof_n > 0) {
        int rc = prof_harvest(ctx);
        if (rc != PD_OK) return rc;
    }
    const int64_t* head = (const int64_t*)ctx->res_host;
    const ResLayout lay = pd_res_layout(ctx->B);
    const size_t end = lay.off_payload + (size_t)head[1] * sizeof(pd_peak) + (size_t)head[2] * sizeof(pd_conn) +
                       (size_t)head[3] * 20 * sizeof(double);
    if (end > ctx->res_have) {
        PD_HIP(ctx, hipMemcpyAsync(ctx->res_host + ctx->res_have, ctx->res_dev + ctx->res_have, end - ctx->res_have,
                                   hipMemcpyDeviceToHost, ctx->stream));
        PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->res_have = end;
    }
    if (ctx->pack_direct) ctx->res_have = end;  // what pd_results_host reports: the bytes in use
    ctx->last_result_bytes = end;
    ctx->last_peaks_per_image = ctx->B > 0 ? (long long)head[1] / ctx->B : -1;
    return PD_OK;
}

// More than about 160 peaks per image means more candidate pairs than the direct form of the
// pair kernel takes in two passes (19 limbs x (peaks / 18)^2 > 1536).  PD_PAIRS_PRE=0/1 in the
// environment fixes the choice (comparison runs).
static void choose_pairs_kernel(pd_ctx* ctx) {
    static const int forced = [] {
        const char* v = getenv("PD_PAIRS_PRE");
        return v && *v ? atoi(v) : -1;
    }();
    if (forced >= 0)
        ctx->pairs_pre = forced != 0;
    else if (ctx->last_peaks_per_image >= 0)
        ctx->pairs_pre = ctx->last_peaks_per_image > 160;
    else
        ctx->pairs_pre = (long long)ctx->H * ctx->W > 512LL * 512LL;
}

static int decode_fullres_enqueue(pd_ctx* ctx, double thre1, double thre2);

extern "C" int pd_decode_fullres(pd_ctx* ctx, double thre1, double thre2) {
    PD_CHECK_CTX(ctx);
    int rc = decode_fullres_enqueue(ctx, thre1, thre2);
    if (rc != PD_OK) return rc;
    return record_done(ctx);
}

extern "C" int pd_decode_multiscale(pd_ctx* ctx, int n_scales, const float* const* heat, const float* const* paf,
                                    const int32_t* hs, const int32_t* ws, const int32_t* crop_h, const int32_t* crop_w,
                                    int H, int W, int B, double thre1, double thre2, int mem) {
    PD_CHECK_CTX(ctx);
    int rc = pd_set_multiscale(ctx, n_scales, heat, paf, hs, ws, crop_h, crop_w, H, W, B, mem);
    if (rc != PD_OK) return rc;
    if (ctx->pack_policy < 0) ctx->pack_direct = ctx->last_result_bytes <= PD_PACK_DIRECT_LIMIT;
    rc = decode_fullres_enqueue(ctx, thre1, thre2);
    if (rc != PD_OK) return rc;
    return record_done(ctx);
}

static int decode_fullres_enqueue(pd_ctx* ctx, double thre1, double thre2) {
    choose_pairs_kernel(ctx);  // (inside pd_decode_batch: the same answer as the one its launch-sequence key holds)
    int rc = pd_peaks(ctx, thre1);
    if (rc != PD_OK) return rc;
    rc = pd_connections(ctx, thre2);
    if (rc != PD_OK) return rc;
    if (pd_parse_pack_fused(ctx)) {
        rc = pd_launch_parse_pack(ctx);
        if (rc != PD_OK) return rc;
        return after_pack(ctx);
    }
    rc = pd_parse(ctx);
    if (rc != PD_OK) return rc;
    return pd_pack(ctx);
}

static int decode_batch_enqueue(pd_ctx* ctx, const float* heat, const float* paf, int B, int h, int w, int H, int W,
                                double thre1, double thre2, int mem);

extern "C" int pd_decode_batch(pd_ctx* ctx, const float* heat, const float* paf, int B, int h, int w, int H,
                               int W, double thre1, double thre2, int mem) {
    PD_CHECK_CTX(ctx);
    if (!heat || !paf) PD_FAIL(ctx, PD_E_INVALID, "pd_decode_batch: heat and paf are required");
    ctx->done_valid = 0;
    int rc = decode_batch_enqueue(ctx, heat, paf, B, h, w, H, W, thre1, thre2, mem);
    if (rc != PD_OK) return rc;
    return record_done(ctx);
}

extern "C" int pd_share_stream(pd_ctx* ctx, pd_ctx* owner) {
    PD_CHECK_CTX(ctx);
    if (!owner || owner == ctx || owner->stream_shared || ctx->stream_shared)
        PD_FAIL(ctx, PD_E_INVALID, "pd_share_stream: the owner must be another context with a stream of its own, and this one must still have its own");
    if (owner->cfg.device_id != ctx->cfg.device_id) PD_FAIL(ctx, PD_E_INVALID, "pd_share_stream: the contexts are on different GPUs");
    if (ctx->worker) PD_FAIL(ctx, PD_E_STATE, "pd_share_stream: not for a context with a launcher thread (pd_submit)");
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int rc = drop_graphs(ctx);
    if (rc != PD_OK) return rc;
    PD_HIP(ctx, hipStreamDestroy(ctx->stream));
    ctx->stream = owner->stream;
    ctx->stream_shared = 1;
    ctx->owner = owner;
    owner->n_sharers++;
    owner->done_valid = 0;  // (the owner waits through its event from now on as well)
    ctx->done_valid = 0;
    ctx->gen++;
    return PD_OK;
}

static int decode_batch_enqueue(pd_ctx* ctx, const float* heat, const float* paf, int B, int h, int w, int H, int W,
                                double thre1, double thre2, int mem) {
    int rc = pd_set_lowres(ctx, heat, paf, B, h, w, H, W, mem);
    if (rc != PD_OK) return rc;
    if (mem != PD_MEM_DEVICE || !ctx->use_graph) {
        if (ctx->pack_policy < 0) ctx->pack_direct = ctx->last_result_bytes <= PD_PACK_DIRECT_LIMIT;
    choose_pairs_kernel(ctx);
        return decode_fullres_enqueue(ctx, thre1, thre2);
    }

    // Device-resident input: the whole launch sequence (six to nine kernels, the result copy
    // of large results and, when profiling, the event pairs) is captured into a
    // HIP graph the second time the same call is seen and replayed from then on, which
    // takes the per-launch host cost out of the steady state.
    if (ctx->pack_policy < 0) ctx->pack_direct = ctx->last_result_bytes <= PD_PACK_DIRECT_LIMIT;
    choose_pairs_kernel(ctx);
    GraphKey key;
    memset(&key, 0, sizeof(key));
    key.pack_direct = ctx->pack_direct;
    key.pairs_pre = ctx->pairs_pre;
    key.heat = heat;
    key.paf = paf;
    key.B = B;
    key.h = h;
    key.w = w;
    key.H = H;
    key.W = W;
    key.t1 = thre1;
    key.t2 = thre2;
    key.mode = ctx->mode;
    key.impl = ctx->impl;
    key.prof = ctx->prof_on;
    key.skel = ctx->skel_dst;
    key.skel_cap = ctx->skel_cap;
    key.host_results = ctx->host_results;
    key.gen = ctx->gen;
    GraphSlot* slot = nullptr;
    for (GraphSlot& g : ctx->graphs)
        if (g.seen && memcmp(&key, &g.key, sizeof(key)) == 0) slot = &g;
    const uint64_t now = ++ctx->graph_clock;
    if (slot && slot->exec) {
        // the recorded sequence holds no clear: it relies on the counters being zero
        if (!ctx->zero_clean) {
            rc = pd_launch_clear(ctx, ctx->zero_block, ctx->zero_bytes / sizeof(int32_t));
            if (rc != PD_OK) return rc;
        }
        PD_HIP(ctx, hipGraphLaunch(slot->exec, ctx->stream));
        slot->used = now;
        ctx->zero_clean = 1;
        ctx->have_peaks = ctx->have_conns = ctx->have_persons = 1;
        ctx->packed = skeleton_only(ctx) ? 0 : 1;
        ctx->res_have = slot->res_have;
        ctx->prof_n = slot->prof_n;
        return PD_OK;
    }
    if (!slot) {
        // first sight of this call: run it eagerly (this also makes every lazy allocation) and
        // remember it in a free slot, else in the one that has gone unused the longest
        GraphSlot* victim = &ctx->graphs[0];
        for (GraphSlot& g : ctx->graphs) {
            if (!g.seen) {
                victim = &g;
                break;
            }
            if (g.used < victim->used) victim = &g;
        }
        if (victim->exec) {
            PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
            PD_HIP(ctx, hipGraphExecDestroy(victim->exec));
        }
        memset(victim, 0, sizeof(*victim));
        victim->key = key;
        victim->seen = 1;
        victim->used = now;
        return decode_fullres_enqueue(ctx, thre1, thre2);
    }
    slot->used = now;
    if (ctx->prof_n > 0) {  // pairs of earlier eager calls must not be mixed into the captured set
        PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        rc = prof_harvest(ctx);
        if (rc != PD_OK) return rc;
    }
    hipGraph_t graph = nullptr;
    // Normally the pack kernel of the previous decode has left the counters zero; if not, they
    // are cleared in front of the recording (by a kernel: the library queues no memset on a
    // decode path, so no sequence it or a caller records can hold a memset node).
    if (!ctx->zero_clean) {
        rc = pd_launch_clear(ctx, ctx->zero_block, ctx->zero_bytes / sizeof(int32_t));
        if (rc != PD_OK) return rc;
        ctx->zero_clean = 1;
    }
    PD_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    rc = decode_fullres_enqueue(ctx, thre1, thre2);
    const hipError_t e_end = hipStreamEndCapture(ctx->stream, &graph);
    if (rc != PD_OK || e_end != hipSuccess || graph == nullptr) {
        if (graph) (void)hipGraphDestroy(graph);
        ctx->use_graph = 0;  // capture is not possible here: stay eager from now on
        ctx->prof_n = 0;
        if (rc != PD_OK) return rc;
        return decode_fullres_enqueue(ctx, thre1, thre2);
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t e_inst = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e_inst != hipSuccess) {
        ctx->use_graph = 0;
        ctx->prof_n = 0;
        return decode_fullres_enqueue(ctx, thre1, thre2);
    }
    if (ctx->gen != key.gen) {  // something moved while recording: the sequence is already stale
        (void)hipGraphExecDestroy(exec);
        memset(slot, 0, sizeof(*slot));
        ctx->prof_n = 0;
        return decode_fullres_enqueue(ctx, thre1, thre2);
    }
    slot->exec = exec;
    slot->res_have = ctx->res_have;
    slot->prof_n = ctx->prof_n;
    PD_HIP(ctx, hipGraphLaunch(exec, ctx->stream));
    return PD_OK;
}

// ---- pd_submit: the same decode, launched by a thread of the context's own -----------
static void worker_main(pd_ctx* ctx) {
    (void)hipSetDevice(ctx->cfg.device_id);  // a new thread starts on device 0
    for (;;) {
        SubmitJob job;
        {
            std::unique_lock<std::mutex> lock(*ctx->wk_mu);
            ctx->wk_cv->wait(lock, [ctx] { return ctx->wk_stop || !ctx->wk_jobs->empty(); });
            if (ctx->wk_jobs->empty()) return;  // stop requested and nothing left
            job = ctx->wk_jobs->front();
            ctx->wk_jobs->pop_front();
            ctx->wk_busy = 1;
        }
        int rc = PD_OK;
        if (ctx->wk_rc == PD_OK) {  // after a failure the rest of the queue is dropped
            if (job.skel) rc = pd_set_skeleton_buffer(ctx, job.skel, job.skel_cap);
            if (rc == PD_OK)
                rc = pd_decode_batch(ctx, job.heat, job.paf, job.B, job.h, job.w, job.H, job.W, job.t1, job.t2, PD_MEM_DEVICE);
        }
        {
            std::lock_guard<std::mutex> lock(*ctx->wk_mu);
            if (rc != PD_OK && ctx->wk_rc == PD_OK) {
                ctx->wk_rc = rc;
                snprintf(ctx->wk_err, sizeof(ctx->wk_err), "%s", ctx->err);
            }
            ctx->wk_busy = 0;
        }
        ctx->wk_cv->notify_all();
    }
}

extern "C" int pd_submit(pd_ctx* ctx, const float* heat, const float* paf, int B, int h, int w, int H, int W,
                         double thre1, double thre2, void* skeleton_ptr, uint64_t skeleton_capacity) {
    PD_CHECK_CTX(ctx);
    if (!heat || !paf) PD_FAIL(ctx, PD_E_INVALID, "pd_submit: heat and paf are required");
    if (ctx->stream_shared) PD_FAIL(ctx, PD_E_STATE, "pd_submit: not for a context on another context's stream (two threads would launch on it)");
    if (!ctx->worker) {
        ctx->wk_mu = new std::mutex();
        ctx->wk_cv = new std::condition_variable();
        ctx->wk_jobs = new std::deque<SubmitJob>();
        ctx->wk_busy = ctx->wk_stop = 0;
        ctx->wk_rc = PD_OK;
        ctx->worker = new std::thread(worker_main, ctx);
    }
    SubmitJob job = {heat, paf, B, h, w, H, W, thre1, thre2, skeleton_ptr, skeleton_capacity};
    {
        std::lock_guard<std::mutex> lock(*ctx->wk_mu);
        ctx->wk_jobs->push_back(job);
    }
    ctx->wk_cv->notify_all();
    return PD_OK;
}

extern "C" int pd_flush(pd_ctx* ctx) {
    PD_CHECK_CTX(ctx);
    if (!ctx->worker) return PD_OK;
    std::unique_lock<std::mutex> lock(*ctx->wk_mu);
    ctx->wk_cv->wait(lock, [ctx] { return ctx->wk_jobs->empty() && !ctx->wk_busy; });
    const int rc = ctx->wk_rc;
    if (rc != PD_OK) {
        snprintf(ctx->err, sizeof(ctx->err), "%s", ctx->wk_err);
        ctx->wk_rc = PD_OK;
    }
    return rc;
}

void pd_prof_begin(pd_ctx* ctx, int stage) {
    ctx->prof_open = 0;
    if (!((ctx->prof_on >> stage) & 1)) return;
    if (ctx->prof_n == ctx->prof_cap) {
        const int ncap = ctx->prof_cap ? ctx->prof_cap * 2 : 256;
        hipEvent_t* ev = (hipEvent_t*)realloc(ctx->prof_ev, sizeof(hipEvent_t) * 2 * ncap);
        int* st = (int*)realloc(ctx->prof_stage, sizeof(int) * ncap);
        if (ev) ctx->prof_ev = ev;
        if (st) ctx->prof_stage = st;
        if (!ev || !st) return;
        for (int i = 2 * ctx->prof_cap; i < 2 * ncap; ++i)
            if (hipEventCreate(&ctx->prof_ev[i]) != hipSuccess) return;
        ctx->prof_cap = ncap;
    }
    ctx->prof_stage[ctx->prof_n] = stage;
    (void)hipEventRecord(ctx->prof_ev[2 * ctx->prof_n], ctx->stream);
    ctx->prof_open = 1;
}

void pd_prof_end(pd_ctx* ctx) {
    if (!ctx->prof_open || ctx->prof_n >= ctx->prof_cap) return;
    (void)hipEventRecord(ctx->prof_ev[2 * ctx->prof_n + 1], ctx->stream);
    ++ctx->prof_n;
    ctx->prof_open = 0;
}

extern "C" int pd_profile(pd_ctx* ctx, int enable) {
    PD_CHECK_CTX(ctx);
    ctx->prof_on = enable ? ~0u : 0u;
    return PD_OK;
}

extern "C" int pd_profile_stages(pd_ctx* ctx, uint32_t stage_mask) {
    PD_CHECK_CTX(ctx);
    ctx->prof_on = stage_mask;
    return PD_OK;
}

extern "C" int pd_profile_read(pd_ctx* ctx, double* ms, int64_t* launches) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int rc = prof_harvest(ctx);
    if (rc != PD_OK) return rc;
    for (int s = 0; s < PD_N_STAGES; ++s) {
        if (ms) ms[s] = ctx->prof_ms[s];
        if (launches) launches[s] = ctx->prof_cnt[s];
        ctx->prof_ms[s] = 0.0;
        ctx->prof_cnt[s] = 0;
    }
    return PD_OK;
}

extern "C" int pd_packed_device(pd_ctx* ctx, void** peaks, void** conns, void** persons, void** part_count,
                                void** limb_count, void** person_count) {
    PD_CHECK_CTX(ctx);
    if (!ctx->packed) PD_FAIL(ctx, PD_E_STATE, "pd_packed_device: call pd_pack (or a decode entry point) first");
    int rc = finish_results(ctx);
    if (rc != PD_OK) return rc;
    const int64_t* head = (const int64_t*)ctx->res_host;
    unsigned char* pay = ctx->res_dev + pd_res_layout(ctx->B).off_payload;
    if (peaks) *peaks = pay;
    if (conns) *conns = pay + (size_t)head[1] * sizeof(pd_peak);
    if (persons) *persons = pay + (size_t)head[1] * sizeof(pd_peak) + (size_t)head[2] * sizeof(pd_conn);
    if (part_count) *part_count = ctx->part_count;
    if (limb_count) *limb_count = ctx->limb_count;
    if (person_count) *person_count = ctx->person_count;
    return PD_OK;
}

extern "C" int pd_sync_totals(pd_ctx* ctx, pd_totals* totals) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    if (!ctx->packed && !totals && skeleton_only(ctx)) {
        // nothing was packed for the host: wait for the stream; the per-image status is in
        // the skeleton buffer
        PD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return PD_OK;
    }
    if (!ctx->packed) PD_FAIL(ctx, PD_E_STATE, "pd_sync_totals: call pd_pack (or a decode entry point) first");
    int rc = finish_results(ctx);
    if (rc != PD_OK) return rc;
    const int64_t* head = (const int64_t*)ctx->res_host;
    if (totals) {
        totals->n_images = ctx->B;
        totals->n_peaks = head[1];
        totals->n_conns = head[2];
        totals->n_persons = head[3];
        totals->first_bad_image = (int32_t)head[4];
    }
    if (head[4] >= 0) {
        const int32_t st = ((const int32_t*)(ctx->res_host + pd_res_layout(ctx->B).off_stat))[head[4]];
        if (st == PD_E_INDEX)
            PD_FAIL(ctx, PD_E_INDEX, "image %lld: a connection touches more than two persons (predict.py:199 IndexError)", (long long)head[4]);
        PD_FAIL(ctx, PD_E_OVERFLOW, "image %lld: a capacity of the context was exceeded (peaks per part %d, pairs %d, persons %d)",
                (long long)head[4], ctx->cfg.max_peaks_per_part, ctx->cfg.max_pairs_per_image, ctx->cfg.max_persons);
    }
    return PD_OK;
}

extern "C" int pd_results_host(pd_ctx* ctx, const void** base, uint64_t* nbytes) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    if (!ctx->packed) PD_FAIL(ctx, PD_E_STATE, "pd_results_host: call pd_pack (or a decode entry point) first");
    int rc = finish_results(ctx);
    if (rc != PD_OK) return rc;
    if (base) *base = ctx->res_host;
    if (nbytes) *nbytes = ctx->res_have;
    return PD_OK;
}

extern "C" int pd_fetch(pd_ctx* ctx, int32_t* part_count, int32_t* limb_count, int32_t* person_count,
                        int32_t* status, pd_peak* peaks, pd_conn* conns, double* persons) {
    PD_CHECK_CTX(ctx);
    PD_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    if (!ctx->packed) PD_FAIL(ctx, PD_E_STATE, "pd_fetch: call pd_pack (or a decode entry point) first");
    int rc = finish_results(ctx);
    if (rc != PD_OK) return rc;
    const size_t B = ctx->B;
    const ResLayout lay = pd_res_layout(ctx->B);
    const unsigned char* r = ctx->res_host;
    const int64_t* head = (const int64_t*)r;
    const unsigned char* pay = r + lay.off_payload;
    if (part_count) memcpy(part_count, r + lay.off_part, B * PD_N_PARTS * sizeof(int32_t));
    if (limb_count) memcpy(limb_count, r + lay.off_limb, B * PD_N_LIMBS * sizeof(int32_t));
    if (person_count) memcpy(person_count, r + lay.off_pers, B * sizeof(int32_t));
    if (status) memcpy(status, r + lay.off_stat, B * sizeof(int32_t));
    if (peaks) memcpy(peaks, pay, (size_t)head[1] * sizeof(pd_peak));
    if (conns) memcpy(conns, pay + (size_t)head[1] * sizeof(pd_peak), (size_t)head[2] * sizeof(pd_conn));
    if (persons)
        memcpy(persons, pay + (size_t)head[1] * sizeof(pd_peak) + (size_t)head[2] * sizeof(pd_conn),
               (size_t)head[3] * 20 * sizeof(double));
    return PD_OK;
}
