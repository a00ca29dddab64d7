// Person assembly: multi_person_parse of utils/postprocess.py:163-237 (mode P) and
// its inlined twin predict.py:179-252 (mode D), plus the packing of all per-image
// results into contiguous arrays.
//
// The algorithm is sequential over limbs and, within a limb, over connections in
// pick order; one wavefront runs it per image with the person table in LDS.  Lanes
// parallelise the "which persons already hold id_a or id_b" scan and the row moves.
#include "common.h"

#pragma clang fp contract(off)

struct ParseArgs {
    int mode, PK, max_persons;
    const pd_conn* conn;
    const double* conn_conf;
    const int32_t* limb_count;
    double* persons;
    int32_t* person_count;
    int32_t* status;
};

__global__ void __launch_bounds__(64)
k_parse(ParseArgs a) {
    extern __shared__ __align__(16) double tab[];  // [max_persons][20]
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    int n = 0;           // live rows
    int err = PD_OK;
    for (int k = 0; k < PD_N_LIMBS; ++k) {
        const int nc = a.limb_count[b * PD_N_LIMBS + k];
        if (nc <= 0) continue;  // special limb or no connection
        const int pa = pd_limb_from(k), pb = pd_limb_to(k);
        const pd_conn* rows = a.conn + ((size_t)b * PD_N_LIMBS + k) * a.PK;
        const double* conf = a.conn_conf + ((size_t)b * PD_N_LIMBS + k) * a.PK * 2;
        for (int ci = 0; ci < nc; ++ci) {
            const pd_conn r = rows[ci];
            const double ida = (double)r.id_a, idb = (double)r.id_b;
            const double conf_a = conf[2 * ci], conf_b = conf[2 * ci + 1];
            // find(): rows with person[a] == id_a or person[b] == id_b, in row order
            int found = 0, j0 = -1, j1 = -1;
            for (int base = 0; base < n; base += 64) {
                const int j = base + lane;
                const bool hit = j < n && (tab[j * 20 + pa] == ida || tab[j * 20 + pb] == idb);
                unsigned long long m = __ballot(hit);
                while (m != 0ull && found < 3) {
                    const int f = base + __builtin_ctzll(m);
                    if (found == 0) j0 = f;
                    else if (found == 1) j1 = f;
                    ++found;
                    m &= m - 1ull;
                }
            }
            if (found == 0) {
                if (a.mode == PD_MODE_D || k < 17) {  // utils/postprocess.py:196 / predict.py:228
                    if (n >= a.max_persons) {
                        err = PD_E_OVERFLOW;
                    } else {
                        if (lane < 18) tab[n * 20 + lane] = lane == pa ? ida : (lane == pb ? idb : -1.0);
                        if (lane == 18) tab[n * 20 + 18] = (conf_a + conf_b) + r.score;
                        if (lane == 19) tab[n * 20 + 19] = 2.0;
                        ++n;
                    }
                }
            } else if (found == 1) {
                if (lane == 0 && tab[j0 * 20 + pb] != idb) {  // :210 / predict.py:205
                    tab[j0 * 20 + pb] = idb;
                    tab[j0 * 20 + 19] += 1.0;
                    tab[j0 * 20 + 18] += conf_b + r.score;
                }
            } else if (found == 2) {
                const bool mine = lane < 18 && tab[j0 * 20 + lane] >= 0.0 && tab[j1 * 20 + lane] >= 0.0;
                const bool overlap = __ballot(mine) != 0ull;  // is_isolated, :14-16
                if (!overlap) {
                    // merge row j1 into j0 (:218-222 / predict.py:216-219), delete j1
                    if (lane < 18) {
                        if (a.mode == PD_MODE_P)
                            tab[j0 * 20 + lane] = (tab[j0 * 20 + lane] + 1.0) + tab[j1 * 20 + lane];
                        else
                            tab[j0 * 20 + lane] = tab[j0 * 20 + lane] + (tab[j1 * 20 + lane] + 1.0);
                    }
                    if (lane == 18) tab[j0 * 20 + 18] = (tab[j0 * 20 + 18] + tab[j1 * 20 + 18]) + r.score;
                    if (lane == 19) tab[j0 * 20 + 19] = tab[j0 * 20 + 19] + tab[j1 * 20 + 19];
                    __syncthreads();
                    if (lane < 20)
                        for (int j = j1; j < n - 1; ++j) tab[j * 20 + lane] = tab[(j + 1) * 20 + lane];
                    --n;
                } else if (lane == 0 && (a.mode == PD_MODE_P || tab[j0 * 20 + pb] != idb)) {
                    tab[j0 * 20 + pb] = idb;  // :210-214 (always) / predict.py:222-225 (if different)
                    tab[j0 * 20 + 19] += 1.0;
                    tab[j0 * 20 + 18] += conf_b + r.score;
                }
            } else if (a.mode == PD_MODE_D) {
                err = PD_E_INDEX;  // predict.py:199 subset_idx[found] with found == 2
            }
            __syncthreads();
            if (err == PD_E_INDEX) break;
        }
        if (err == PD_E_INDEX) break;
    }
    // prune (:228-232 / predict.py:248-252) and write out in row order
    double* out = a.persons + (size_t)b * a.max_persons * 20;
    int kept = 0;
    for (int base = 0; base < n; base += 64) {
        const int j = base + lane;
        bool keep = false;
        if (j < n) {
            const double cnt = tab[j * 20 + 19], sc = tab[j * 20 + 18];
            keep = !(cnt < 4.0 || sc / cnt < 0.4);
        }
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int pos = kept + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            for (int c = 0; c < 20; ++c) out[(size_t)pos * 20 + c] = tab[j * 20 + c];
        }
        kept += __builtin_popcountll(m);
    }
    if (lane == 0) {
        a.person_count[b] = err == PD_E_INDEX ? 0 : kept;
        if (err != PD_OK && a.status[b] == PD_OK) a.status[b] = err;
    }
}

// ---- packing --------------------------------------------------------------------
struct PackArgs {
    int B, PK, max_persons;
    int have_peaks, have_conns, have_persons;
    const int32_t *part_count, *part_base, *limb_count, *person_count, *status;
    const int32_t *pk_y, *pk_x;
    const double* pk_score;
    const pd_conn* conn;
    const double* persons;
    int64_t* pack_off;  // [3][B+1]
    unsigned char* res; // result buffer: header | count tables | peaks | conns | persons
    ResLayout lay;
};

// Exclusive prefix sums over the images (one workgroup; B is at most a few thousand).
__global__ void __launch_bounds__(256)
k_pack_offsets(PackArgs a) {
    __shared__ long long s_part[3][256];
    __shared__ int s_bad[256];
    const int tid = threadIdx.x;
    const int per = (a.B + 255) / 256;
    const int lo = tid * per, hi = min(a.B, lo + per);
    long long acc[3] = {0, 0, 0};
    int bad = 0x7fffffff;
    for (int b = lo; b < hi; ++b) {
        if (a.have_peaks) acc[0] += a.part_base[b * (PD_N_PARTS + 1) + PD_N_PARTS];
        if (a.have_conns)
            for (int k = 0; k < PD_N_LIMBS; ++k) acc[1] += max(0, a.limb_count[b * PD_N_LIMBS + k]);
        if (a.have_persons) acc[2] += a.person_count[b];
        if (a.status[b] != PD_OK && b < bad) bad = b;
    }
    for (int q = 0; q < 3; ++q) s_part[q][tid] = acc[q];
    s_bad[tid] = bad;
    __syncthreads();
    if (tid == 0) {
        int fb = 0x7fffffff;
        for (int q = 0; q < 3; ++q) {
            long long run = 0;
            for (int t = 0; t < 256; ++t) {
                const long long v = s_part[q][t];
                s_part[q][t] = run;
                run += v;
            }
            ((int64_t*)a.res)[1 + q] = run;
            a.pack_off[q * (a.B + 1) + a.B] = run;
        }
        for (int t = 0; t < 256; ++t) fb = min(fb, s_bad[t]);
        ((int64_t*)a.res)[0] = a.B;
        ((int64_t*)a.res)[4] = fb == 0x7fffffff ? -1 : fb;
    }
    __syncthreads();
    long long run[3] = {s_part[0][tid], s_part[1][tid], s_part[2][tid]};
    for (int b = lo; b < hi; ++b) {
        for (int q = 0; q < 3; ++q) a.pack_off[q * (a.B + 1) + b] = run[q];
        if (a.have_peaks) run[0] += a.part_base[b * (PD_N_PARTS + 1) + PD_N_PARTS];
        if (a.have_conns)
            for (int k = 0; k < PD_N_LIMBS; ++k) run[1] += max(0, a.limb_count[b * PD_N_LIMBS + k]);
        if (a.have_persons) run[2] += a.person_count[b];
    }
}

__global__ void __launch_bounds__(128)
k_pack(PackArgs a) {
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    // count tables of this image
    if (tid < PD_N_PARTS) ((int32_t*)(a.res + a.lay.off_part))[b * PD_N_PARTS + tid] = a.have_peaks ? a.part_count[b * PD_N_PARTS + tid] : 0;
    if (tid < PD_N_LIMBS) ((int32_t*)(a.res + a.lay.off_limb))[b * PD_N_LIMBS + tid] = a.have_conns ? a.limb_count[b * PD_N_LIMBS + tid] : 0;
    if (tid == 0) {
        ((int32_t*)(a.res + a.lay.off_pers))[b] = a.have_persons ? a.person_count[b] : 0;
        ((int32_t*)(a.res + a.lay.off_stat))[b] = a.status[b];
    }
    const int64_t n_peaks = a.pack_off[a.B], n_conns = a.pack_off[(a.B + 1) + a.B];
    pd_peak* out_peaks = (pd_peak*)(a.res + a.lay.off_payload);
    pd_conn* out_conns = (pd_conn*)(a.res + a.lay.off_payload + n_peaks * sizeof(pd_peak));
    double* out_persons = (double*)(a.res + a.lay.off_payload + n_peaks * sizeof(pd_peak) + n_conns * sizeof(pd_conn));
    if (a.have_peaks) {
        pd_peak* out = out_peaks + a.pack_off[b];
        for (int p = 0; p < PD_N_PARTS; ++p) {
            const int n = a.part_count[b * PD_N_PARTS + p];
            const int base = a.part_base[b * (PD_N_PARTS + 1) + p];
            const size_t o = ((size_t)b * PD_N_PARTS + p) * a.PK;
            for (int i = tid; i < n; i += blockDim.x) {
                pd_peak q;
                q.y = a.pk_y[o + i];
                q.x = a.pk_x[o + i];
                q.score = a.pk_score[o + i];
                out[base + i] = q;
            }
        }
    }
    if (a.have_conns) {
        pd_conn* out = out_conns + a.pack_off[(a.B + 1) + b];
        int run = 0;
        for (int k = 0; k < PD_N_LIMBS; ++k) {
            const int n = max(0, a.limb_count[b * PD_N_LIMBS + k]);
            const pd_conn* rows = a.conn + ((size_t)b * PD_N_LIMBS + k) * a.PK;
            for (int i = tid; i < n; i += blockDim.x) out[run + i] = rows[i];
            run += n;
        }
    }
    if (a.have_persons) {
        double* out = out_persons + a.pack_off[2 * (a.B + 1) + b] * 20;
        const int n = a.person_count[b] * 20;
        const double* src = a.persons + (size_t)b * a.max_persons * 20;
        for (int i = tid; i < n; i += blockDim.x) out[i] = src[i];
    }
}

int pd_launch_parse(pd_ctx* ctx) {
    ParseArgs a;
    a.mode = ctx->mode;
    a.PK = ctx->cfg.max_peaks_per_part;
    a.max_persons = ctx->cfg.max_persons;
    a.conn = ctx->conn;
    a.conn_conf = ctx->conn_conf;
    a.limb_count = ctx->limb_count;
    a.persons = ctx->persons;
    a.person_count = ctx->person_count;
    a.status = ctx->status;
    const size_t lds = (size_t)a.max_persons * 20 * sizeof(double);
    pd_prof_begin(ctx, PD_STAGE_PARSE);
    hipLaunchKernelGGL(k_parse, dim3(ctx->B), dim3(64), lds, ctx->stream, a);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    return PD_OK;
}

int pd_launch_pack(pd_ctx* ctx) {
    PackArgs a;
    a.B = ctx->B;
    a.PK = ctx->cfg.max_peaks_per_part;
    a.max_persons = ctx->cfg.max_persons;
    a.have_peaks = ctx->have_peaks;
    a.have_conns = ctx->have_conns;
    a.have_persons = ctx->have_persons;
    a.part_count = ctx->part_count;
    a.part_base = ctx->part_base;
    a.limb_count = ctx->limb_count;
    a.person_count = ctx->person_count;
    a.status = ctx->status;
    a.pk_y = ctx->pk_y;
    a.pk_x = ctx->pk_x;
    a.pk_score = ctx->pk_score;
    a.conn = ctx->conn;
    a.persons = ctx->persons;
    a.pack_off = ctx->pack_off;
    a.res = ctx->res_dev;
    a.lay = pd_res_layout(ctx->B);
    pd_prof_begin(ctx, PD_STAGE_PACK);
    hipLaunchKernelGGL(k_pack_offsets, dim3(1), dim3(256), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_pack, dim3(ctx->B), dim3(128), 0, ctx->stream, a);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    return PD_OK;
}
