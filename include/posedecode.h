/*
 * libposedecode -- C ABI of the MI355X (gfx950) OpenPose decode path.
 *
 * The reference has no FFI layer: its boundary is plain Python functions
 * (utils/postprocess.py:26,78,163 and the decode half of predict.py:45-254).
 * Each entry point below names the reference interface it stands in for.  All
 * entry points return an int status (0 = PD_OK, < 0 = error, text via
 * pd_last_error).  Pointers are plain host or device addresses as stated by the
 * `mem` argument; no framework types cross this boundary.
 *
 * Threading: one context per GPU stream; a context is not re-entrant, distinct
 * contexts are independent.  The library owns all device memory inside a
 * context; the caller owns every buffer it passes in.  Variable-length outputs
 * are returned packed, with their sizes reported first (pd_sync_totals), and an
 * image that exceeds a capacity given at pd_create makes the call fail with
 * PD_E_OVERFLOW -- results are never truncated silently.
 */
#ifndef POSEDECODE_H
#define POSEDECODE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PD_N_PARTS 18 /* cfg.ch_heats - 1, cfgs/config_vgg19.py:76 */
#define PD_N_LIMBS 19 /* len(cfg.map_idx), cfgs/config_vgg19.py:59-63 */
#define PD_CH_HEAT 19
#define PD_CH_PAF 38

enum {
    PD_OK = 0,
    PD_E_INVALID = -1,  /* bad argument / shape (Python: ValueError) */
    PD_E_HIP = -2,      /* HIP runtime error */
    PD_E_OVERFLOW = -3, /* a capacity from pd_config was exceeded */
    PD_E_INDEX = -4,    /* mode D: a connection touches > 2 persons; predict.py:195-200 raises IndexError */
    PD_E_STATE = -5     /* stage called before the stage that feeds it */
};

enum { PD_MODE_P = 0, /* utils/postprocess.py semantics */
       PD_MODE_D = 1  /* predict.py detect() semantics */ };
enum { PD_MEM_HOST = 0, PD_MEM_DEVICE = 1 };
enum { PD_F32 = 0, PD_F64 = 1 };
/* How the peak stage is computed; every choice returns identical results. */
enum { PD_IMPL_FAST = 0,   /* float32 candidate filter + exact float64 verification of the
                              candidates where the source is network-resolution, else STREAM */
       PD_IMPL_SIMPLE = 1, /* one-thread-per-pixel kernels, any image size; on-GPU cross-check */
       PD_IMPL_STREAM = 2  /* exact float64 streaming kernel over every pixel */ };

typedef struct pd_ctx pd_ctx;

typedef struct {
    int32_t device_id;
    int32_t max_batch;           /* images per call */
    int32_t max_h, max_w;        /* full-resolution image size */
    int32_t max_peaks_per_part;  /* per image and body part */
    int32_t max_persons;         /* per image, before pruning; at most 960 (160 bytes of LDS per open person) */
    int32_t max_pairs_per_image; /* sum over limbs of nA*nB */
    int32_t fullres;             /* != 0: allocate the full-resolution planes needed by
                                    pd_upsample_avg / pd_load_fullres */
} pd_config;

typedef struct {
    int32_t y, x;  /* pixel coordinates */
    double score;  /* unsmoothed heatmap value at (y, x) */
} pd_peak;

typedef struct {
    int32_t id_a, id_b; /* global peak ids */
    int32_t ia, ib;     /* index within the part's peak list */
    double score;       /* PAF line integral + distance prior */
} pd_conn;

typedef struct {
    int32_t n_images;
    int32_t first_bad_image; /* -1, or the first image whose status is not PD_OK */
    int64_t n_peaks, n_conns, n_persons;
} pd_totals;

int pd_create(const pd_config* cfg, pd_ctx** out);
int pd_destroy(pd_ctx* ctx);
const char* pd_last_error(pd_ctx* ctx);
const char* pd_version(void);

/* Which of the two decode variants of the reference to follow (see the enum). */
int pd_set_mode(pd_ctx* ctx, int mode);
int pd_set_impl(pd_ctx* ctx, int impl);
/* The HIP stream the context launches on (a hipStream_t), for callers that time
 * launches with events or order their own work against it. */
void* pd_stream(pd_ctx* ctx);
/* Puts ctx on the stream of another context of the same GPU (ctx gives up its own).  Two
 * contexts used in turn on one stream keep that stream two decodes deep: the second context's
 * pd_decode_batch queues behind the first's while the host still fetches the first's results
 * (pd_sync_totals / pd_fetch of a sharing context wait for that context's own last decode, not
 * for the stream).  The chip serves four hardware queues at a time, so more streams do not help;
 * a second decode on each does.  Both contexts must be driven by one host thread (no pd_submit),
 * and the owner must outlive ctx.  No counterpart in the reference (it decodes one image at a
 * time on the CPU). */
int pd_share_stream(pd_ctx* ctx, pd_ctx* owner);

/* ---- stage 1: predict.py:51-72 (multi-scale upsample, un-pad, resize, average) ----
 * heat[m] is (B, hs[m], ws[m], 19) float32, paf[m] is (B, hs[m], ws[m], 38):
 * the network outputs of scale m.  Each is resized x8 with the bicubic kernel
 * (predict.py:63,67), cropped to crop_h[m] x crop_w[m] (:64,:68), resized to
 * H x W (:65,:69), divided by n_scales in float32 and accumulated in float64
 * (:71-72).  The averages stay in the context as the source of pd_peaks /
 * pd_connections; pd_read_fullres copies them out as (B, H, W, C) float64. */
int pd_upsample_avg(pd_ctx* ctx, int n_scales, const float* const* heat, const float* const* paf,
                    const int32_t* hs, const int32_t* ws, const int32_t* crop_h,
                    const int32_t* crop_w, int H, int W, int B, int mem);
int pd_read_fullres(pd_ctx* ctx, double* heat_out, double* paf_out);

/* The same scale loop (predict.py:51-72) as the source of the decode that follows it
 * (predict.py:82-254), without the averaged maps ever being stored: the arguments are those of
 * pd_upsample_avg, the arrays of every scale must stay valid (device-resident ones: unchanged)
 * until the decode has finished.  pd_set_multiscale only installs the source -- pd_peaks /
 * pd_connections / pd_parse / pd_decode_fullres then work on it -- and pd_decode_multiscale is
 * the whole of detect() behind the network calls in one enqueue: the peak stage finds the
 * candidate cells from bounds on the smoothed average, filters them in float32 and decides
 * every candidate in exact float64 arithmetic from the four sources (heat channel 18, which
 * predict.py:91 never reads, is not touched); the PAF average is evaluated at the sample
 * points of the candidate pairs only (predict.py:136-152).  Results are those of
 * pd_upsample_avg + pd_decode_fullres, bit for bit.  pd_read_fullres is not available on
 * this source (there are no maps to read; use pd_upsample_avg for them). */
int pd_set_multiscale(pd_ctx* ctx, int n_scales, const float* const* heat, const float* const* paf,
                      const int32_t* hs, const int32_t* ws, const int32_t* crop_h,
                      const int32_t* crop_w, int H, int W, int B, int mem);
int pd_decode_multiscale(pd_ctx* ctx, int n_scales, const float* const* heat, const float* const* paf,
                         const int32_t* hs, const int32_t* ws, const int32_t* crop_h,
                         const int32_t* crop_w, int H, int W, int B, double thre1, double thre2, int mem);

/* Full-resolution maps supplied by the caller, as the arguments of
 * get_peaks(heatmap, ..) / get_connections(.., pafs, ..) in utils/postprocess.py:
 * (B, H, W, C) arrays of float32 or float64.  Either pointer may be NULL to keep
 * what the context already holds.  A float32 heatmap is smoothed with float32
 * storage after each axis and thresholded in float32, as scipy/numpy do. */
int pd_load_fullres(pd_ctx* ctx, const void* heat, int heat_dtype, int heat_channels,
                    const void* paf, int paf_dtype, int paf_channels, int B, int H, int W, int mem);

/* Network-resolution source for the fused path: (B, h, w, 19) and (B, h, w, 38)
 * float32; the full-resolution map is the x8 bicubic upsample cropped to H x W
 * (8(h-1) < H <= 8h), evaluated on the fly and never written to memory. */
int pd_set_lowres(pd_ctx* ctx, const float* heat, const float* paf, int B, int h, int w, int H,
                  int W, int mem);

/* ---- stage 2+3: get_peaks, utils/postprocess.py:26-75 (predict.py:88-111) ---- */
int pd_peaks(pd_ctx* ctx, double thre1);
/* Peaks supplied by the caller (one image): counts[18], then y/x/score/id arrays
 * concatenated part by part -- the `peaks` argument of get_connections. */
int pd_set_peaks(pd_ctx* ctx, const int32_t* counts, const int32_t* y, const int32_t* x,
                 const double* score, const int32_t* id);

/* ---- stage 4+5: get_connections, utils/postprocess.py:78-160 (predict.py:120-174) ---- */
int pd_connections(pd_ctx* ctx, double thre2);
/* Connections supplied by the caller (one image): counts[19] (-1 = limb listed in
 * special_limb_idx), rows of 5 doubles [id_a, id_b, score, conf_a, conf_b]. */
int pd_set_connections(pd_ctx* ctx, const int32_t* counts, const double* rows);

/* ---- stage 6: multi_person_parse, utils/postprocess.py:163-237 (predict.py:179-252) ---- */
int pd_parse(pd_ctx* ctx);

/* ---- the whole decode for a batch: predict.py:82-254 on B images ----
 * Enqueues set_lowres + peaks + connections + parse + packing on the context's
 * stream and returns without waiting. */
int pd_decode_batch(pd_ctx* ctx, const float* heat, const float* paf, int B, int h, int w, int H,
                    int W, double thre1, double thre2, int mem);
/* The same from the full-resolution maps the context holds (after
 * pd_upsample_avg or pd_load_fullres). */
int pd_decode_fullres(pd_ctx* ctx, double thre1, double thre2);

/* ---- results ----
 * pd_pack gathers whatever stages have run into the packed result buffer;
 * pd_sync_totals waits for the stream and reports sizes (and PD_E_OVERFLOW /
 * PD_E_INDEX if an image failed); pd_fetch copies the packed results to host
 * arrays sized from those totals:
 *   part_count   [B][18]  peaks per part
 *   limb_count   [B][19]  connections per limb, -1 where nA*nB == 0 ("special")
 *   person_count [B]
 *   status       [B]
 *   peaks        n_peaks entries, image-major, then part-major, then (y, x) ascending
 *   conns        n_conns entries, image-major, then limb-major, in greedy pick order
 *   persons      n_persons x 20 doubles, image-major, in row order
 * Any output pointer may be NULL. */
int pd_pack(pd_ctx* ctx);
int pd_sync_totals(pd_ctx* ctx, pd_totals* totals);
int pd_fetch(pd_ctx* ctx, int32_t* part_count, int32_t* limb_count, int32_t* person_count,
             int32_t* status, pd_peak* peaks, pd_conn* conns, double* persons);

/* The packed results in the context's own pinned host buffer, without a copy: valid
 * until the next stage call on this context.  Layout: int64 header[8] = {B, n_peaks,
 * n_conns, n_persons, first_bad_image, 0, 0, 0}; int32 part_count[B][18],
 * limb_count[B][19], person_count[B], status[B]; then, 16-byte aligned, pd_peak[n_peaks],
 * pd_conn[n_conns], double[n_persons][20]. */
int pd_results_host(pd_ctx* ctx, const void** base, uint64_t* nbytes);

/* Device addresses of the packed results and count tables (valid after
 * pd_sync_totals until the next stage call on this context), for callers that
 * forward them without a host round trip (e.g. a collective gather).  Any output
 * pointer may be NULL. */
int pd_packed_device(pd_ctx* ctx, void** peaks, void** conns, void** persons, void** part_count,
                     void** limb_count, void** person_count);

/* Per-stage timing with HIP events recorded on the context's stream around every
 * kernel launch of a stage.  pd_profile(ctx, 1) starts collecting;
 * pd_profile_read waits for the stream and returns, per stage, the summed
 * milliseconds and the number of launches since the last read. */
enum {
    PD_STAGE_RESIZE = 0,     /* bicubic resize / average / layout kernels */
    PD_STAGE_SMOOTH_NMS = 1, /* Gaussian + NMS + peak emission (exact kernel); in the filtered path: the
                                redo of overflowing planes and the ordering of the peaks */
    PD_STAGE_SORT = 2,       /* peak ordering and id assignment, where it is a kernel of its own */
    PD_STAGE_PAIRS = 3,      /* PAF line integrals */
    PD_STAGE_MATCH = 4,      /* per-limb sort + greedy matching */
    PD_STAGE_PARSE = 5,      /* person assembly (pd_decode_batch into the library's own buffer: assembly and packing share a launch, timed here) */
    PD_STAGE_PACK = 6,       /* result packing */
    PD_STAGE_FILTER = 7,     /* float32 candidate filter (active cells, or every pixel of full-resolution maps) */
    PD_STAGE_VERIFY = 8,     /* exact float64 verification of the candidates */
    PD_STAGE_PLANAR = 9,     /* (h, w, 19) -> planar copy of the heatmaps (maps beyond 4096 pixels only) */
    PD_STAGE_CELLS = 10,     /* active-cell scan at network resolution (maps beyond 4096 pixels; smaller ones: inside PD_STAGE_FILTER) */
    PD_N_STAGES = 11
};
int pd_profile(pd_ctx* ctx, int enable);
/* The same, restricted to the stages whose bit (1u << PD_STAGE_x) is set: timing one
 * stage costs one event pair per launch instead of one per kernel. */
int pd_profile_stages(pd_ctx* ctx, uint32_t stage_mask);
int pd_profile_read(pd_ctx* ctx, double* ms, int64_t* launches);

/* The smoothed heatmap the peak stage compares (scipy.ndimage.gaussian_filter,
 * sigma = 3), (B, 18, H, W) float64; test hook for the filter alone. */
int pd_debug_smoothed(pd_ctx* ctx, double* out);

/* Graph replay of pd_decode_batch on device-resident inputs (on by default, off with
 * PD_NO_GRAPH=1 in the environment at pd_create): a call is run eagerly the first time
 * its arguments are seen, recorded into a HIP graph the second time and replayed from
 * then on.  Up to eight argument sets are kept per context (least recently used goes
 * first); whatever a recorded sequence depends on -- pointers, sizes, thresholds, mode,
 * skeleton buffer and capacity, profiling mask, and the addresses and contents of the
 * context's own buffers and tables -- is part of the key, so a replay never runs on
 * stale state.  Disabling destroys the recorded sequences. */
int pd_set_graph_replay(pd_ctx* ctx, int enable);

/* pd_decode_batch on device-resident inputs, issued by a launcher thread that belongs to
 * the context: the call only queues the request and returns, so one host thread can keep
 * several contexts busy while their kernels are being launched in parallel (a decode is
 * about fifteen launches).  skeleton_ptr, when not NULL, is installed with
 * pd_set_skeleton_buffer before the decode.  Requests of one context run in order.
 * pd_flush waits until every queued request has been launched (not: executed) and
 * returns the first error since the last flush; no other entry point of the context
 * may be called between pd_submit and the pd_flush that follows it.  No counterpart in
 * the reference (single-threaded Python). */
int pd_submit(pd_ctx* ctx, const float* heat, const float* paf, int B, int h, int w, int H, int W,
              double thre1, double thre2, void* skeleton_ptr, uint64_t skeleton_capacity);
int pd_flush(pd_ctx* ctx);

/* A caller-owned device buffer that receives, with every pd_pack (so with every decode),
 * a second copy of the packed results without the connection lists: the layout of the
 * result buffer with n_conns = 0 and a zero limb table -- what a multi-GPU job gathers
 * (peaks and person rows are all evaluate.py:44-74 reads).  header[5] != 0 means the
 * buffer was too small for the payload (header[5] = bytes needed); the header and the
 * count tables are always written.  NULL removes the buffer.  The buffer belongs to the
 * context's stream until the results have been waited for.  No counterpart in the
 * reference: the reference is single-process. */
int pd_set_skeleton_buffer(pd_ctx* ctx, void* device_ptr, uint64_t capacity);

/* enable = 0: while a skeleton buffer is set, a pack writes that buffer only -- no packed
 * result buffer, no copy to the host mirror -- and pd_sync_totals (with totals; with
 * NULL it only waits for the stream) / pd_fetch / pd_results_host / pd_packed_device report
 * PD_E_STATE; per-image status and the first bad image are in the skeleton buffer's
 * header and tables.  For ranks of a multi-GPU job
 * whose results leave through the gather.  On by default; no effect without a skeleton
 * buffer.  No counterpart in the reference. */
int pd_set_host_results(pd_ctx* ctx, int enable);

/* Counters of the last peak stage that went through the candidate filter
 * (PD_IMPL_FAST on network-resolution sources): out[0] candidates handed to the exact
 * verification, out[1] / out[2] active 8 x 8 pixel cells with interior / border
 * weights, out[3] planes redone by the exact kernel because their candidate list
 * overflowed.  Waits for the stream.  No counterpart in the reference; diagnostics. */
int pd_filter_stats(pd_ctx* ctx, int64_t* out);

/* pd_decode_batch for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the
 * reference (Python, garbage-collected).  The call takes no context, allocates nothing
 * and frees nothing: all intermediate state (peak lists, counters, candidate-pair scores,
 * connections, person rows, tap tables) lives in one device array of
 * PD_STATIC_SCRATCH_BYTES that is part of the library's code object and stays resident on
 * every GPU the library runs on until the process ends.
 *
 * heat (B, h, w, 19) and paf (B, h, w, 38): float32 device arrays, 16-byte aligned.
 * out: device buffer of out_capacity bytes that receives the packed results in the
 * layout of pd_results_host (header, count tables, peaks, connections, person rows);
 * header[4] is the first image whose status is not PD_OK (or -1), header[5] != 0 means
 * the payload did not fit (header[5] = bytes needed; header and count tables are always
 * written).  stream: the hipStream_t to launch on (NULL = the default stream); the call
 * only enqueues, the caller waits for its stream.  Runs on the current device.
 * Limits (PD_E_INVALID with a message from pd_last_error(NULL) beyond them): B <=
 * PD_STATIC_MAX_BATCH, 26 <= H, W <= PD_STATIC_MAX_SIDE; per image at most
 * PD_STATIC_PEAKS peaks per part, PD_STATIC_PERSONS open persons and PD_STATIC_PAIRS
 * candidate pairs (an image beyond those gets status PD_E_OVERFLOW, as with a context).
 * Every call on a GPU uses that GPU's one array: calls that overlap on a GPU race, also
 * across streams and threads.  The first call on a device (or any pd_create on it)
 * looks up the array's address and must not run under stream capture; every later call
 * enqueues kernels only (no memset, no copy, no host allocation), so a captured call records
 * kernel nodes and nothing else. */
#define PD_STATIC_MAX_BATCH 16
#define PD_STATIC_MAX_SIDE 512
#define PD_STATIC_PEAKS 64
#define PD_STATIC_PERSONS 64
#define PD_STATIC_PAIRS 16384
#define PD_STATIC_SCRATCH_BYTES (4u << 20)
int pd_decode_batch_static(const float* heat, const float* paf, int B, int h, int w, int H, int W,
                           double thre1, double thre2, int mode, void* out, uint64_t out_capacity,
                           void* stream);

/* The stray debug output of utils/postprocess.py:224-225 -- print(connection, len(found))
 * for every connection of limb 18 -- is not produced by the library; with recording on
 * (enable != 0) pd_parse runs the table form of the parser and notes len(found) of every
 * limb-18 connection, and a call with out != NULL copies those numbers out
 * ([max_batch][max_peaks_per_part] int32, connection order) so that the caller can print the
 * reference's lines.  enable = 0 switches recording off again. */
int pd_debug_found(pd_ctx* ctx, int enable, int32_t* out);

/* ---- next to the decode: its consumers and the step before it ---- */

/* Person rows -> COCO keypoint rows, the loop of evaluate.py:44-74 over a decoded batch,
 * computed on the device from the peaks and person rows the context holds.  Row r (image-
 * major, then in person-row order) is person r: keypoints[r][k] = (c0, c1, 1) of the peak
 * that person[cfg.coco_to_ours[k]] names (cfgs/config_vgg19.py:48), i.e. the first two
 * columns of predict.py:179's `candidate` -- (x, y) in mode D, (y, x) in mode P -- or
 * (0, 0, 0) where the part is missing (evaluate.py:53-54); scores[r] = person[-2]
 * (evaluate.py:45); image_index[r] = the image of the batch.  keypoints is
 * [capacity][17][3] float64; *n_rows receives the number of rows; more rows than capacity
 * -> PD_E_OVERFLOW with *n_rows set and nothing copied.  Output pointers are host
 * addresses; any may be NULL.  Waits for the context's stream. */
#define PD_COCO_KEYPOINTS 17
int pd_coco_rows(pd_ctx* ctx, double* keypoints, double* scores, int32_t* image_index,
                 int64_t capacity, int64_t* n_rows);

/* The image side of the scale loop, predict.py:56-57:
 *   scale_img = cv2.resize(img, (0, 0), fx=scale, fy=scale, interpolation=cv2.INTER_CUBIC)
 *   scale_img_padded, pad = pad_right_down_corner(scale_img, stride, pad_value)
 * img is (h, w, channels) PD_U8 or PD_F32, channels <= 4; out is (out_h, out_w, channels)
 * PD_F32 or PD_U8 with out_h / out_w = the resized size (h * scale, w * scale rounded half
 * to even) padded below / right to a multiple of stride -- pd_image_resize_shape computes
 * them and the pad list entries pad[2], pad[3].  The cubic is evaluated in float32 with the
 * arithmetic of pd_upsample_avg; PD_U8 output rounds half to even and saturates.  (OpenCV's
 * own 8-bit path is a fixed-point variant whose last bit depends on the build; there is no
 * fixture for it.)  mem says where img and out live. */
enum { PD_U8 = 2 };
int pd_image_resize_shape(int h, int w, double scale, int stride, int32_t* resized_h,
                          int32_t* resized_w, int32_t* pad_down, int32_t* pad_right);
int pd_image_resize_pad(pd_ctx* ctx, const void* img, int img_dtype, int h, int w, int channels,
                        double scale, int stride, double pad_value, void* out, int out_dtype,
                        int out_h, int out_w, int mem);

/* visualize(), predict.py:256-307, as a per-pixel kernel.  img is (H, W, 3) uint8 on the
 * host.  circles[n][3] = (x, y, colour index): the filled radius-4 circles of predict.py:269
 * in draw order; part_out receives cv2.addWeighted(img, 0.3, canvas, 0.7, 0)
 * (predict.py:277, part.jpg).  ellipses[m][6] = (cx, cy, a, b, angle in whole degrees,
 * colour index): the arguments predict.py:298 hands to cv2.ellipse2Poly for every (limb,
 * person), in draw order; each polygon is filled and blended 0.4 / 0.6 into the canvas
 * (predict.py:299-300); final_out receives the canvas (final.jpg).  Colours are
 * predict.py:258-260's table.  The shapes follow OpenCV's drawing algorithms as published
 * (midpoint circle; vertices at whole degrees from a 7-decimal float32 sine table, rounded
 * half to even; outline lines + 16.16 fixed-point scan fill), except that edges crossing
 * the image border are not clipped first.  OpenCV is not available to check against and
 * the reference holds no image of its output: parity of the pixels is not pinned.
 * Either output may be NULL. */
int pd_render(pd_ctx* ctx, const uint8_t* img, int H, int W, int n_circles, const int32_t* circles,
              int n_ellipses, const int32_t* ellipses, uint8_t* part_out, uint8_t* final_out);

/* ---- multi-GPU collect without a framework in the process ----
 * No counterpart in the reference (single process).  A communicator wraps RCCL, which is
 * loaded at run time (PD_RCCL_LIBRARY names the file; default librccl.so.1).  One rank
 * makes the 128-byte id with pd_comm_unique_id and hands it to the others by any means;
 * every rank then calls pd_comm_create(id, world, rank, device). */
typedef struct pd_comm pd_comm;
int pd_comm_unique_id(void* id128);
int pd_comm_create(const void* id128, int world, int rank, int device_id, pd_comm** out);
int pd_comm_destroy(pd_comm* comm);
const char* pd_comm_last_error(pd_comm* comm);

/* Collects the skeleton buffers (pd_set_skeleton_buffer) of every rank's n_ctx contexts on
 * rank `root`: an all-gather of the byte counts, then grouped sends / receives of exactly
 * those bytes, then one copy to recv_host.  On the root, recv_host receives the buffers
 * back to back (each starting 16-byte aligned) in rank-major, then context order -- the
 * image order of a batch sharded in contiguous blocks -- and sizes[rank * n_ctx + i]
 * their lengths; *total_bytes is the room they take.  A recv_capacity that is too small
 * is reported by PD_E_OVERFLOW on the root after the exchange has completed on every rank
 * (nothing is copied to recv_host; the other ranks return PD_OK).  Waits for the contexts'
 * streams first and for the collect before it returns; n_ctx <= 16 and equal on all ranks. */
int pd_gather(pd_comm* comm, pd_ctx* const* ctxs, int n_ctx, int root, void* recv_host,
              uint64_t recv_capacity, uint64_t* sizes, uint64_t* total_bytes);

/* Part of the same exchange, for callers that gather fixed-size slices themselves (one slice of
 * `capacity` bytes per decode, n_slices of them back to back at src_dev -- what a framework's gather
 * collective delivers): stores, of every slice, the bytes its header says are in use (header, count
 * tables, peaks, person rows; the tables alone if the header reports a buffer that was too small) to the
 * same offsets of dst, which may be pinned host memory.  Queued on `stream`; the bytes of dst behind a
 * slice's used part are left as they were.  capacity: a multiple of 16; both addresses 16-byte aligned. */
int pd_copy_used_skeletons(const void* src_dev, void* dst, int n_slices, uint64_t capacity, void* stream);

/* ---- test hook (host only, touches no GPU) ----
 * The linear operator the sparse multi-scale peak stage evaluates for one axis of one scale --
 * scipy's Gaussian (sigma 3, reflect) after the second cubic resize (crop -> L outputs, predict.py:65)
 * after the x8 resize (l sources, predict.py:63), times `factor` -- as a dense matrix K_out[L][l]
 * (float64) and in the cell form the kernels read: cw_out[cells][10][16] = weight of source
 * lo_out[c] + p in output 8c - 1 + j, bw_out[cells][16] = largest magnitude over the cell's own
 * outputs j = 1..8, cells = (L + 7) / 8.  Any output may be NULL.  Returns the widest window of a
 * cell, -1 when one exceeds 16 sources, -2 for bad arguments.  No counterpart in the reference. */
int pd_test_ms_axis(int l, int crop, int L, int identity, double factor, double* K_out, float* cw_out,
                    float* bw_out, int32_t* lo_out);

#ifdef __cplusplus
}
#endif
#endif /* POSEDECODE_H */
