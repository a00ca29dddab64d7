// Per-limb stable sort + greedy matching of the scored candidate pairs
// (utils/postprocess.py:141-158, twin predict.py:161-170), one wavefront per (image, limb).
#pragma once
#include "common.h"

struct MatchArgs {
    int PK, max_pairs, cap;  // cap: candidates sorted in LDS (power of two)
    const double* pair_score;
    const int32_t* pair_off;
    const int32_t* part_count;
    const int32_t* pk_id;
    const double* pk_score;
    pd_conn* conn;
    double* conn_conf;
    int32_t* limb_count;
};

__device__ __forceinline__ bool cand_before(double sa, int ia, double sb, int ib) {
    return sa > sb || (sa == sb && ia < ib);
}

// LDS bytes one wavefront needs (a multiple of 16).
__host__ __device__ inline size_t pd_match_lds(int cap, int PK) { return ((size_t)cap * 12 + 2 * (size_t)PK + 15) & ~(size_t)15; }

// Limb k of image b by one wavefront; dyn: pd_match_lds(cap, PK) bytes of LDS that belong to
// this wavefront alone.
__device__ __forceinline__ void match_limb(const MatchArgs& a, int k, int b, int lane, unsigned char* dyn) {
    double* s_score = (double*)dyn;
    int32_t* s_idx = (int32_t*)(dyn + (size_t)a.cap * 8);
    unsigned char* usedA = dyn + (size_t)a.cap * 12;
    unsigned char* usedB = usedA + a.PK;
    const int pa = pd_limb_from(k), pb = pd_limb_to(k);
    const int nA = a.part_count[b * PD_N_PARTS + pa], nB = a.part_count[b * PD_N_PARTS + pb];
    int32_t* out_count = a.limb_count + b * PD_N_LIMBS + k;
    if (nA == 0 || nB == 0) {  // utils/postprocess.py:92-95
        if (lane == 0) *out_count = -1;
        return;
    }
    if (a.pair_off[b * (PD_N_LIMBS + 1) + PD_N_LIMBS] < 0) {  // pair capacity exceeded, image flagged
        if (lane == 0) *out_count = 0;
        return;
    }
    const int off = a.pair_off[b * (PD_N_LIMBS + 1) + k];
    const int npairs = a.pair_off[b * (PD_N_LIMBS + 1) + k + 1] - off;
    const double* ps = a.pair_score + (size_t)b * a.max_pairs + off;
    for (int i = lane; i < a.PK; i += 64) {
        usedA[i] = 0;
        usedB[i] = 0;
    }
    // compact the surviving candidates, keeping a-major / b-minor order
    int total = 0;
    for (int base = 0; base < npairs; base += 64) {
        const int i = base + lane;
        const double s = i < npairs ? ps[i] : -1.0;
        const bool ok = s > 0.0;
        const unsigned long long m = __ballot(ok);
        const int pos = total + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (ok && pos < a.cap) {
            s_score[pos] = s;
            s_idx[pos] = i;
        }
        total += __builtin_popcountll(m);
    }
    const int limit = nA < nB ? nA : nB;
    pd_conn* rows = a.conn + ((size_t)b * PD_N_LIMBS + k) * a.PK;
    double* conf = a.conn_conf + ((size_t)b * PD_N_LIMBS + k) * a.PK * 2;
    const int32_t* idA = a.pk_id + ((size_t)b * PD_N_PARTS + pa) * a.PK;
    const int32_t* idB = a.pk_id + ((size_t)b * PD_N_PARTS + pb) * a.PK;
    const double* scA = a.pk_score + ((size_t)b * PD_N_PARTS + pa) * a.PK;
    const double* scB = a.pk_score + ((size_t)b * PD_N_PARTS + pb) * a.PK;
    int count = 0;
    pd_wave_sync();
    if (total <= a.cap) {
        // wave-level bitonic sort on (score descending, pair index ascending)
        int n2 = 64;
        while (n2 < total) n2 <<= 1;
        for (int i = total + lane; i < n2; i += 64) {
            s_score[i] = -INFINITY;
            s_idx[i] = 0x7fffffff;
        }
        pd_wave_sync();
        for (int kk = 2; kk <= n2; kk <<= 1) {
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int t = lane; t < n2; t += 64) {
                    const int q = t ^ j;
                    if (q > t) {
                        const double st = s_score[t], sq = s_score[q];
                        const int it = s_idx[t], iq = s_idx[q];
                        const bool fwd = (t & kk) == 0;
                        const bool t_first = cand_before(st, it, sq, iq);
                        if (fwd ? !t_first : t_first) {
                            s_score[t] = sq;
                            s_idx[t] = iq;
                            s_score[q] = st;
                            s_idx[q] = it;
                        }
                    }
                }
                pd_wave_sync();
            }
        }
        // greedy pick, 64 sorted candidates at a time: the first lane whose two
        // peaks are still free wins, then the others re-check
        for (int base = 0; base < total && count < limit; base += 64) {
            const int i = base + lane;
            bool have = i < total;
            const int idx = have ? s_idx[i] : 0;
            const double sc = have ? s_score[i] : 0.0;
            const int ia = idx / nB, ib = idx - ia * nB;
            // every lane fetches what a pick of its candidate would store, so that the
            // pick loop itself waits for no memory
            pd_conn r;
            r.id_a = have ? idA[ia] : 0;
            r.id_b = have ? idB[ib] : 0;
            r.ia = ia;
            r.ib = ib;
            r.score = sc;
            const double conf_a = have ? scA[ia] : 0.0, conf_b = have ? scB[ib] : 0.0;
            while (count < limit) {
                const bool ok = have && !usedA[ia] && !usedB[ib];
                const unsigned long long m = __ballot(ok);
                if (m == 0ull) break;
                const int first = __builtin_ctzll(m);
                if (lane == first) {
                    rows[count] = r;
                    conf[2 * count] = conf_a;
                    conf[2 * count + 1] = conf_b;
                    usedA[ia] = 1;
                    usedB[ib] = 1;
                    have = false;
                }
                ++count;
                pd_wave_sync();
            }
        }
    } else {
        // more survivors than the LDS list holds: repeated arg-max over the pair
        // scores in memory (the same picks as sort + greedy, in the same order)
        while (count < limit) {
            double best = -1.0;
            int best_i = 0x7fffffff;
            for (int i = lane; i < npairs; i += 64) {
                const double s = ps[i];
                if (s > 0.0) {
                    const int ia = i / nB, ib = i - ia * nB;
                    if (!usedA[ia] && !usedB[ib] && cand_before(s, i, best, best_i)) {
                        best = s;
                        best_i = i;
                    }
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const double os = __shfl_xor(best, d);
                const int oi = __shfl_xor(best_i, d);
                if (cand_before(os, oi, best, best_i)) {
                    best = os;
                    best_i = oi;
                }
            }
            if (!(best > 0.0)) break;
            const int ia = best_i / nB, ib = best_i - ia * nB;
            if (lane == 0) {
                pd_conn r;
                r.id_a = idA[ia];
                r.id_b = idB[ib];
                r.ia = ia;
                r.ib = ib;
                r.score = best;
                rows[count] = r;
                conf[2 * count] = scA[ia];
                conf[2 * count + 1] = scB[ib];
                usedA[ia] = 1;
                usedB[ib] = 1;
            }
            ++count;
            pd_wave_sync();
        }
    }
    if (lane == 0) *out_count = count;
}

