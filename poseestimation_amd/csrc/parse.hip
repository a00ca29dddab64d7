// Person assembly: multi_person_parse of utils/postprocess.py:163-237 (mode P) and
// its inlined twin predict.py:179-252 (mode D), plus the packing of all per-image
// results into one contiguous buffer.
//
// The algorithm is sequential over limbs and, within a limb, over connections in
// pick order; one wavefront runs it per image.  k_parse_wave is the fast form (lane j
// owns person j, at most 64 open persons); k_parse keeps whole rows in an LDS table of
// max_persons rows and takes the images the fast form hands over.
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
    int only_flagged;  // != 0: only images the fast kernel handed over (count -1)
    int32_t* dbg_found;  // [B][PK] or NULL: len(found) of every limb-18 connection (utils/postprocess.py:224-225)
};

// What the assembly leaves in memory (person rows, counts, status) is stored write-through at
// agent scope: k_parse_pack hands it to another workgroup inside the launch, and such stores
// need no cache write-back before the hand-over.
__device__ __forceinline__ void put_out(double* p, double v) {
    __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void put_out(int32_t* p, int32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The table form for image b; tab holds max_persons rows of 20 doubles in LDS.  Run by one
// wavefront (its lanes are ordered among themselves with pd_wave_sync).  Returns the number of
// person rows it wrote.
__device__ int parse_table(const ParseArgs& a, int b, int lane, double* tab) {
    int n = 0;           // live rows
    int err = PD_OK;
    // the limb's connection counts, one per lane, fetched once
    const int my_nc = lane < PD_N_LIMBS ? a.limb_count[b * PD_N_LIMBS + lane] : 0;
    for (int k = 0; k < PD_N_LIMBS; ++k) {
        const int nc = __builtin_amdgcn_readlane(my_nc, k);
        if (nc <= 0) continue;  // special limb or no connection
        const int pa = pd_limb_from(k), pb = pd_limb_to(k);
        const pd_conn* rows = a.conn + ((size_t)b * PD_N_LIMBS + k) * a.PK;
        const double* conf = a.conn_conf + ((size_t)b * PD_N_LIMBS + k) * a.PK * 2;
        for (int ci = 0; ci < nc; ++ci) {
            const pd_conn r = rows[ci];
            const double ida = (double)r.id_a, idb = (double)r.id_b;
            const double conf_a = conf[2 * ci], conf_b = conf[2 * ci + 1];
            // find(): rows with person[a] == id_a or person[b] == id_b, in row order
            int found = 0, j0 = -1, j1 = -1, hits = 0;
            for (int base = 0; base < n; base += 64) {
                const int j = base + lane;
                const bool hit = j < n && (tab[j * 20 + pa] == ida || tab[j * 20 + pb] == idb);
                unsigned long long m = __ballot(hit);
                hits += __builtin_popcountll(m);
                while (m != 0ull && found < 3) {
                    const int f = base + __builtin_ctzll(m);
                    if (found == 0) j0 = f;
                    else if (found == 1) j1 = f;
                    ++found;
                    m &= m - 1ull;
                }
            }
            if (a.dbg_found && k == 18 && lane == 0) a.dbg_found[(size_t)b * a.PK + ci] = hits;
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
                    pd_wave_sync();
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
            pd_wave_sync();
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
            for (int c = 0; c < 20; ++c) put_out(out + (size_t)pos * 20 + c, tab[j * 20 + c]);
        }
        kept += __builtin_popcountll(m);
    }
    if (lane == 0) {
        put_out(a.person_count + b, err == PD_E_INDEX ? 0 : kept);
        if (err != PD_OK && a.status[b] == PD_OK) put_out(a.status + b, err);
    }
    return err == PD_E_INDEX ? 0 : kept;
}

__global__ void __launch_bounds__(64)
k_parse(ParseArgs a) {
    extern __shared__ __align__(16) double tab[];  // [max_persons][20]
    const int b = blockIdx.x;
    if (a.only_flagged && a.person_count[b] != -1) return;
    parse_table(a, b, threadIdx.x, tab);
}

// ---- one lane per person -----------------------------------------------------------
// Lane j owns person j: its score and part count live in that lane's registers and its
// 18 part ids in LDS as ids[part][j] (int32, exact: ids are small integers and a merge
// only ever forms -1 + 1 + id).  "Which persons hold id_a or id_b" is two LDS reads per
// lane and one ballot.  All connections of the image are staged into LDS up front with
// one round trip to memory.  At most 64 open persons and WAVE_CONNS connections; an
// image beyond that is handed to k_parse (person_count = -1).
#define WAVE_CONNS 1024
#define WAVE_IDS 2560  // peak ids the per-limb lookup tables cover

__device__ __forceinline__ double bcast_d(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// What a connection hands the person it extends or opens; tagged with its limb.
struct WaveRec {
    int ab;     // id_a << 16 | id_b
    int meta;   // (limb + 1) << 8 | REC_NEW | connection
    double sc;  // extension: conf_b + score; new person: (conf_a + conf_b) + score
};
#define REC_NEW 0x80
#define REC_CONN 0x3f
#define REC_TAG 0x1f00

// The wavefront's LDS.
struct WaveShared {
    int ids[PD_N_PARTS + 1][64];  // (+ a spare row: where the lanes with nothing to write write)
    __attribute__((aligned(16))) double c_all[WAVE_CONNS * 4];  // staged connections; the table form's rows on hand-over
    __attribute__((aligned(16))) WaveRec rec[64 + 1];  // person (lane) <- the connection that matches its near end, or opens it
    short rec_b[64 + 2];               // person <- the connection that matches its far end: (limb + 1) << 8 | connection
    short who[WAVE_IDS + 4];           // peak id -> the person that holds it (-1: nobody); exact while every limb went the fast way
    short chk[WAVE_IDS + 4];           // peak id -> connection of the current limb (written and read back by the same lanes)
    unsigned char part_of[WAVE_IDS]; // peak id -> the part it was seen as (0xff: not seen)
};

// What the wavefront carries from limb to limb.
struct WaveState {
    double my_score;
    int my_cnt;
    int n;       // open persons
    int rc;      // 0 go on, 1 hand the image to the table form, 2 mode-D index error
    int staged;  // connections staged so far
    int by_id;   // the lookup by peak id is valid: ids in range, every id one part's, no limb went through the loop yet
};

__device__ __forceinline__ void wave_parse_begin(WaveShared& sh, WaveState& st, int lane) {
    int* who2 = (int*)sh.who;
    for (int i = lane; i < WAVE_IDS / 2; i += 64) who2[i] = -1;
    int* part4 = (int*)sh.part_of;
    for (int i = lane; i < WAVE_IDS / 4; i += 64) part4[i] = -1;
    sh.rec[lane].meta = 0;
    sh.rec_b[lane] = 0;
#pragma unroll
    for (int p = 0; p < PD_N_PARTS; ++p) sh.ids[p][lane] = -1;
    st.my_score = 0.0;
    st.my_cnt = 0;
    st.n = 0;
    st.rc = 0;
    st.staged = 0;
    st.by_id = 1;
    pd_wave_sync();
}

// Measurement build only (-DPP_STAMPS, tools/parse_pack_split.py): where a workgroup's time goes.
// Per workgroup, in units of the 100 MHz wall clock: start, image number drawn, row count known
// (assembly done), each copy wavefront done, look-back done, rows written, (last image) header written;
// 8: connections staged, 9: limbs applied.
#ifdef PP_STAMPS
__device__ long long g_pp_stamps[4096 * 16];
#define PP_STAMP(slot, cond)                                               \
    do {                                                                   \
        if ((cond) && blockIdx.x < 4096) g_pp_stamps[blockIdx.x * 16 + (slot)] = (long long)wall_clock64(); \
    } while (0)
extern "C" int pd_debug_pp_stamps(long long* out, int n_blocks) {
    if (n_blocks < 0 || n_blocks > 4096) return PD_E_INVALID;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp_stamps), (size_t)n_blocks * 16 * sizeof(long long)) == hipSuccess ? PD_OK : PD_E_HIP;
}
#define PP_NOTE(slot, cond, value)                                         \
    do {                                                                   \
        if ((cond) && blockIdx.x < 4096) g_pp_stamps[blockIdx.x * 16 + (slot)] = (long long)(value); \
    } while (0)
#else
#define PP_STAMP(slot, cond) do { } while (0)
#define PP_NOTE(slot, cond, value) do { } while (0)
#endif

// Limbs k0 .. k1-1 of image b: their connections are staged into LDS behind the ones staged
// before (one round trip to memory), then applied in limb order.
__device__ __forceinline__ void wave_parse_limbs(const ParseArgs& a, int b, int lane, int k0, int k1, WaveShared& sh,
                                                 WaveState& st) {
    if (st.rc != 0) return;
    double* c_score = sh.c_all;
    double* c_ca = sh.c_all + WAVE_CONNS;
    double* c_cb = sh.c_all + 2 * WAVE_CONNS;
    int* c_ida = (int*)(sh.c_all + 3 * WAVE_CONNS);
    int* c_idb = c_ida + WAVE_CONNS;
    const bool mine = lane >= k0 && lane < k1;
    const int my_nc = mine ? max(0, a.limb_count[b * PD_N_LIMBS + lane]) : 0;
    // exclusive prefix of the connection counts over the limbs of this call
    int off = my_nc;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        const int v = __shfl_up(off, d);
        if (lane >= d) off += v;
    }
    const int total = __builtin_amdgcn_readlane(off, PD_N_LIMBS - 1);
    off -= my_nc;
    if (st.staged + total > WAVE_CONNS) {
        st.rc = 1;
        return;
    }
    const int at0 = st.staged;  // where this call's connections start in the staging area
    // four connections per lane in flight, the limb of a flat index found from the prefix
    // sums held across the lanes
    bool id_trouble = false;  // an id out of range or seen as two different parts: no lookup by id for this image
    for (int i0 = 0; i0 < total; i0 += 256) {
        pd_conn r[4];
        double cf0[4], cf1[4];
        int pa_u[4], pb_u[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 64 + lane;
            r[u].id_a = r[u].id_b = 0;
            r[u].score = cf0[u] = cf1[u] = 0.0;
            pa_u[u] = pb_u[u] = 0;
            if (i < total) {
                // limbs without connections share their offset with the next one: the scan
                // lands on the last limb whose offset is <= i, which is the one that holds i
                int k = k0, off_k = 0;
#pragma unroll
                for (int q = 1; q < PD_N_LIMBS; ++q) {
                    const int oq = __builtin_amdgcn_readlane(off, q);
                    if (q > k0 && q < k1 && oq <= i) {
                        k = q;
                        off_k = oq;
                    }
                }
                const int local = i - off_k;
                const size_t at = ((size_t)b * PD_N_LIMBS + k) * a.PK + local;
                r[u] = a.conn[at];
                cf0[u] = a.conn_conf[2 * at];
                cf1[u] = a.conn_conf[2 * at + 1];
                pa_u[u] = pd_limb_from(k);
                pb_u[u] = pd_limb_to(k);
            }
        }
        bool ranged[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 64 + lane;
            ranged[u] = false;
            if (i < total) {
                c_ida[at0 + i] = r[u].id_a;
                c_idb[at0 + i] = r[u].id_b;
                c_score[at0 + i] = r[u].score;
                c_ca[at0 + i] = cf0[u];
                c_cb[at0 + i] = cf1[u];
                ranged[u] = r[u].id_a >= 0 && r[u].id_a < WAVE_IDS && r[u].id_b >= 0 && r[u].id_b < WAVE_IDS;
                if (!ranged[u]) id_trouble = true;
                // the part an id stands for: as seen by the rounds before ...
                if (ranged[u]) {
                    const int sa = sh.part_of[r[u].id_a], sb = sh.part_of[r[u].id_b];
                    if ((sa != 0xff && sa != pa_u[u]) || (sb != 0xff && sb != pb_u[u])) id_trouble = true;
                }
            }
        }
        pd_wave_sync();
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ranged[u]) {
                sh.part_of[r[u].id_a] = (unsigned char)pa_u[u];
                sh.part_of[r[u].id_b] = (unsigned char)pb_u[u];
            }
        pd_wave_sync();
        // ... and by this round (of two lanes that wrote different parts, one reads the other's)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ranged[u] && (sh.part_of[r[u].id_a] != pa_u[u] || sh.part_of[r[u].id_b] != pb_u[u])) id_trouble = true;
    }
    if (__ballot(id_trouble) != 0ull) st.by_id = 0;
    st.staged += total;
    pd_wave_sync();
    PP_STAMP(8, lane == 0);
    double my_score = st.my_score;
    int my_cnt = st.my_cnt;
    int n = st.n;
    int rc = 0;
    int by_id = st.by_id;
    const int my_pa = pd_limb_from(min(lane, PD_N_LIMBS - 1)), my_pb = pd_limb_to(min(lane, PD_N_LIMBS - 1));
    // the limbs of this call that have connections, and the connections of the first of them (one per lane)
    unsigned todo = (unsigned)__ballot(mine && my_nc > 0);
    int nx_ida = 0, nx_idb = 0;
    double nx_score = 0.0, nx_ca = 0.0, nx_cb = 0.0;
    if (todo != 0u) {
        const int ci = min(at0 + __builtin_amdgcn_readlane(off, __builtin_ctz(todo)) + lane, WAVE_CONNS - 1);
        nx_ida = c_ida[ci];
        nx_idb = c_idb[ci];
        nx_score = c_score[ci];
        nx_ca = c_ca[ci];
        nx_cb = c_cb[ci];
    }
    while (todo != 0u && rc == 0) {
        const int k = __builtin_ctz(todo);
        todo &= todo - 1u;
        const int nc = __builtin_amdgcn_readlane(my_nc, k), o = at0 + __builtin_amdgcn_readlane(off, k);
        const int pa = __builtin_amdgcn_readlane(my_pa, k), pb = __builtin_amdgcn_readlane(my_pb, k);
        // Whole limb at once, one lane per connection and one per person.  When no two connections
        // of the limb share a peak (always so for connections that come from the matching
        // stage), no person matches two different connections and no connection two different
        // persons, the sequential loop below would touch disjoint persons: each person whose
        // near end matches takes its connection's far end, each unmatched connection opens a
        // person, in connection order.  A connection finds its persons through the table
        // peak id -> person, which the fast way keeps exact; anything else (shared peaks, two
        // owners, merges, ids out of range) takes the loop, for this limb and all after it.
        // Straight-line code: every lane makes every LDS access, the lanes with nothing to read
        // or write at spare entries behind the tables; two round trips to LDS per limb, the
        // connections of the next limb fetched during the first.
        if (by_id && nc <= 64) {
            const bool is_c = lane < nc;
            const int ida = nx_ida, idb = nx_idb;
            const double l_score = nx_score, l_ca = nx_ca, l_cb = nx_cb;
            const int ia = is_c ? ida : WAVE_IDS, ib = is_c ? idb : WAVE_IDS + 1;
            sh.chk[ia] = (short)lane;
            sh.chk[ib] = (short)lane;
            pd_wave_sync();
            const int wa_r = sh.who[ia], wb_r = sh.who[ib], ka = sh.chk[ia], kb = sh.chk[ib];
            {
                const int kn = todo != 0u ? __builtin_ctz(todo) : k;
                const int ci = min(at0 + __builtin_amdgcn_readlane(off, kn) + lane, WAVE_CONNS - 1);
                nx_ida = c_ida[ci];
                nx_idb = c_idb[ci];
                nx_score = c_score[ci];
                nx_ca = c_ca[ci];
                nx_cb = c_cb[ci];
            }
            const int wa = is_c ? wa_r : -1, wb = is_c ? wb_r : -1;
            const bool dup = is_c && (ka != lane || kb != lane);
            const bool shared_conn = wa >= 0 && wb >= 0 && wa != wb;
            const bool unmatched = is_c && wa < 0 && wb < 0;
            const unsigned long long um = __ballot(unmatched);
            const int n_new = (a.mode == PD_MODE_D || k < 17) ? __builtin_popcountll(um) : 0;
            const int tag = (k + 1) << 8;
            const int slot = n + __builtin_popcountll(um & ((1ull << lane) - 1ull));
            const bool opens = unmatched && n_new > 0 && slot < 64;
            WaveRec r;
            r.ab = (ida << 16) | idb;
            r.meta = tag | lane | (wa >= 0 ? 0 : REC_NEW);
            // extension: :210 / predict.py:205; new person: utils/postprocess.py:196-205 / predict.py:228-244
            r.sc = wa >= 0 ? l_cb + l_score : (l_ca + l_cb) + l_score;
            sh.rec[wa >= 0 ? wa : (opens ? slot : 64)] = r;
            sh.rec_b[wb >= 0 ? wb : 64] = (short)(tag | lane);
            pd_wave_sync();
            const WaveRec mine_rec = sh.rec[lane];
            const int mine_b = sh.rec_b[lane];
            const int col_b = sh.ids[pb][lane];
            const bool got_a = lane < n && (mine_rec.meta & REC_TAG) == tag && !(mine_rec.meta & REC_NEW);
            const bool got_b = lane < n && (mine_b & REC_TAG) == tag;
            const bool two = got_a && got_b && (mine_rec.meta & REC_CONN) != (mine_b & REC_CONN);
            const bool fast = __ballot(dup || shared_conn || two) == 0ull;
            if (fast) {
                if (n + n_new > 64 || n + n_new > a.max_persons) {  // more open persons than lanes: the table form takes the image
                    rc = 1;
                    break;
                }
                const int tb = mine_rec.ab & 0xffff, ta = mine_rec.ab >> 16;
                const bool is_new = lane >= n && lane < n + n_new;
                const bool ext = got_a && col_b != tb;
                sh.ids[is_new ? pa : PD_N_PARTS][lane] = ta;
                sh.ids[ext || is_new ? pb : PD_N_PARTS][lane] = tb;
                sh.who[is_new ? ta : WAVE_IDS + 2] = (short)lane;
                sh.who[ext || is_new ? tb : WAVE_IDS + 2] = (short)lane;
                my_cnt = is_new ? 2 : my_cnt + (ext ? 1 : 0);
                const double grown = my_score + mine_rec.sc;
                my_score = is_new ? mine_rec.sc : (ext ? grown : my_score);
                n += n_new;
                pd_wave_sync();
                continue;
            }
        }
        by_id = 0;  // the loop does not keep the table peak id -> person
        int col_a = sh.ids[pa][lane], col_b = sh.ids[pb][lane];
        for (int c0 = 0; c0 < nc && rc == 0; c0 += 64) {
            // this chunk's connections, one per lane, broadcast below with readlane
            const int mine_at = min(o + c0 + lane, WAVE_CONNS - 1);
            const int l_ida = c_ida[mine_at], l_idb = c_idb[mine_at];
            const double l_score = c_score[mine_at], l_ca = c_ca[mine_at], l_cb = c_cb[mine_at];
            const int m_end = min(64, nc - c0);
            for (int s = 0; s < m_end; ++s) {
                const int ida = __builtin_amdgcn_readlane(l_ida, s), idb = __builtin_amdgcn_readlane(l_idb, s);
                const double score = bcast_d(l_score, s), conf_a = bcast_d(l_ca, s), conf_b = bcast_d(l_cb, s);
                const bool hit = lane < n && (col_a == ida || col_b == idb);
                const unsigned long long m = __ballot(hit);
                const int found = __builtin_popcountll(m);
                if (found == 0) {
                    if (a.mode == PD_MODE_D || k < 17) {  // utils/postprocess.py:196 / predict.py:228
                        if (n >= 64 || n >= a.max_persons) {
                            rc = 1;
                            break;
                        }
                        if (lane == n) {
                            col_a = ida;
                            col_b = idb;
                            sh.ids[pa][lane] = ida;
                            sh.ids[pb][lane] = idb;
                            my_score = (conf_a + conf_b) + score;
                            my_cnt = 2;
                        }
                        ++n;
                    }
                } else if (found == 1) {
                    const int j0 = __builtin_ctzll(m);
                    if (lane == j0 && col_b != idb) {  // :210 / predict.py:205
                        col_b = idb;
                        sh.ids[pb][lane] = idb;
                        my_cnt += 1;
                        my_score += conf_b + score;
                    }
                } else if (found == 2) {
                    const int j0 = __builtin_ctzll(m), j1 = __builtin_ctzll(m & (m - 1ull));
                    pd_wave_sync();  // the table in LDS is complete up to here
                    const bool both = lane < PD_N_PARTS && sh.ids[lane][j0] >= 0 && sh.ids[lane][j1] >= 0;
                    const bool overlap = __ballot(both) != 0ull;  // is_isolated, :14-16
                    if (!overlap) {
                        // merge person j1 into j0 (:218-222 / predict.py:216-219), delete j1
                        if (lane < PD_N_PARTS) sh.ids[lane][j0] = (sh.ids[lane][j0] + 1) + sh.ids[lane][j1];
                        const double s1 = __shfl(my_score, j1);
                        const int c1 = __shfl(my_cnt, j1);
                        if (lane == j0) {
                            my_score = (my_score + s1) + score;
                            my_cnt += c1;
                        }
                        pd_wave_sync();
                        // close the gap: persons above j1 move down one lane
                        int nxt[PD_N_PARTS];
#pragma unroll
                        for (int p = 0; p < PD_N_PARTS; ++p) nxt[p] = sh.ids[p][lane < 63 ? lane + 1 : lane];
                        const double nxs = __shfl_down(my_score, 1);
                        const int nxc = __shfl_down(my_cnt, 1);
                        pd_wave_sync();
                        if (lane >= j1) {
#pragma unroll
                            for (int p = 0; p < PD_N_PARTS; ++p) sh.ids[p][lane] = lane == 63 ? -1 : nxt[p];
                            my_score = nxs;
                            my_cnt = nxc;
                        }
                        --n;
                        pd_wave_sync();
                        col_a = sh.ids[pa][lane];
                        col_b = sh.ids[pb][lane];
                    } else if (lane == j0 && (a.mode == PD_MODE_P || col_b != idb)) {
                        col_b = idb;  // :210-214 (always) / predict.py:222-225 (if different)
                        sh.ids[pb][lane] = idb;
                        my_cnt += 1;
                        my_score += conf_b + score;
                    }
                } else if (a.mode == PD_MODE_D) {
                    rc = 2;  // predict.py:199 subset_idx[found] with found == 2
                    break;
                }
            }
        }
        pd_wave_sync();  // this limb's writes are visible before the next limb loads its columns
    }
    st.my_score = my_score;
    st.my_cnt = my_cnt;
    st.n = n;
    st.rc = rc;
    st.by_id = by_id;
    PP_STAMP(9, lane == 0);
}

// Prune (:228-232 / predict.py:248-252), or hand the image over.  Returns the number of person
// rows of the image (-1: left to k_parse) and stores it.  direct: the rows are still with the
// lanes -- lane j with keep holds row number rank -- and wave_write_row puts them where the
// caller wants them; otherwise the table form has written them to the staging area already.
__device__ __forceinline__ int wave_parse_prune(const ParseArgs& a, int b, int lane, WaveShared& sh, WaveState& st, bool& keep,
                                                int& rank, bool& direct) {
    keep = false;
    rank = 0;
    direct = false;
    if (st.rc == 1) {  // more connections or open persons than the wavefront form holds: the table form redoes this image
        if (a.only_flagged) {
            if (lane == 0) put_out(a.person_count + b, -1);
            return -1;
        }
        pd_wave_sync();
        return parse_table(a, b, lane, sh.c_all);
    }
    if (st.rc == 2) {
        if (lane == 0) {
            put_out(a.person_count + b, 0);
            if (a.status[b] == PD_OK) put_out(a.status + b, (int32_t)PD_E_INDEX);
        }
        return 0;
    }
    if (lane < st.n) keep = !((double)st.my_cnt < 4.0 || st.my_score / (double)st.my_cnt < 0.4);
    const unsigned long long m = __ballot(keep);
    rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
    direct = true;
    const int kept = __builtin_popcountll(m);
    if (lane == 0) put_out(a.person_count + b, (int32_t)kept);
    return kept;
}

// The row of this lane's person (18 part ids, score, part count) to out and, if mirror != 0, to
// the same place mirror bytes further on.
__device__ __forceinline__ void wave_write_row(double* out, long long mirror, int lane, const WaveShared& sh, const WaveState& st) {
    double v[20];
#pragma unroll
    for (int p = 0; p < PD_N_PARTS; ++p) v[p] = (double)sh.ids[p][lane];
    v[18] = st.my_score;
    v[19] = (double)st.my_cnt;
#pragma unroll
    for (int q = 0; q < 20; ++q) out[q] = v[q];
    if (mirror != 0) {
        double* out2 = (double*)((unsigned char*)out + mirror);
#pragma unroll
        for (int q = 0; q < 20; ++q) out2[q] = v[q];
    }
}

__global__ void __launch_bounds__(64)
k_parse_wave(ParseArgs a) {
    __shared__ WaveShared sh;
    const int b = blockIdx.x, lane = threadIdx.x;
    WaveState st;
    wave_parse_begin(sh, st, lane);
    wave_parse_limbs(a, b, lane, 0, PD_N_LIMBS, sh, st);
    bool keep, direct;
    int rank;
    wave_parse_prune(a, b, lane, sh, st, keep, rank, direct);
    if (direct && keep) wave_write_row(a.persons + ((size_t)b * a.max_persons + rank) * 20, 0, lane, sh, st);
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
    unsigned char* res;  // result buffer: header | count tables | peaks | conns | persons
    unsigned char* res2; // or NULL: a second copy of every byte, written as it is produced (the
                         // pinned host mirror, stored to over PCIe: no copy kernel behind the pack)
    unsigned long long cap;  // bytes behind res, 0 = large enough by construction
    ResLayout lay;
    int32_t* zero;           // counters to clear for the next decode (or NULL)
    int zero_words;
};

// One workgroup per image.  Every workgroup sums the item counts of all images itself
// (the prefix over the images before it gives its offsets, the sum over all of them the
// section bases), so no separate scan launch is needed; B is at most a few thousand.
__global__ void __launch_bounds__(256)
k_pack(PackArgs a) {
    __shared__ long long s_red[4][6];
    __shared__ int s_redbad[4];
    __shared__ long long s_tot[6];  // prefix (peaks, conns, persons), totals (peaks, conns, persons)
    __shared__ int s_pbase[PD_N_PARTS + 1], s_lbase[PD_N_LIMBS + 1];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    // every stage that reads the per-decode counters has run: leave them zero for the next decode
    if (a.zero)
        for (int i = b * 256 + tid; i < a.zero_words; i += gridDim.x * 256) a.zero[i] = 0;
    long long acc[6] = {0, 0, 0, 0, 0, 0};
    int bad = 0x7fffffff;
    for (int j = tid; j < a.B; j += 256) {
        const long long np = a.have_peaks ? a.part_base[j * (PD_N_PARTS + 1) + PD_N_PARTS] : 0;
        const long long nq = a.have_persons ? a.person_count[j] : 0;
        acc[3] += np;
        acc[5] += nq;
        if (j < b) {
            acc[0] += np;
            acc[2] += nq;
        }
        if (a.status[j] != PD_OK) bad = min(bad, j);
    }
    if (a.have_conns)
        for (int e = tid; e < a.B * PD_N_LIMBS; e += 256) {
            const long long nc = max(0, a.limb_count[e]);
            acc[4] += nc;
            if (e / PD_N_LIMBS < b) acc[1] += nc;
        }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
        for (int q = 0; q < 6; ++q) acc[q] += __shfl_xor(acc[q], d);
        bad = min(bad, __shfl_xor(bad, d));
    }
    if ((tid & 63) == 0) {
        for (int q = 0; q < 6; ++q) s_red[tid >> 6][q] = acc[q];
        s_redbad[tid >> 6] = bad;
    }
    if (tid <= PD_N_PARTS) s_pbase[tid] = a.have_peaks ? a.part_base[b * (PD_N_PARTS + 1) + tid] : 0;
    __syncthreads();
    if (tid < 6) s_tot[tid] = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
    if (tid == 6) {
        int run = 0;
        for (int k = 0; k < PD_N_LIMBS; ++k) {
            s_lbase[k] = run;
            run += a.have_conns ? max(0, a.limb_count[b * PD_N_LIMBS + k]) : 0;
        }
        s_lbase[PD_N_LIMBS] = run;
    }
    __syncthreads();
    const unsigned long long need = a.lay.off_payload + (unsigned long long)s_tot[3] * sizeof(pd_peak) +
                                    (unsigned long long)s_tot[4] * sizeof(pd_conn) + (unsigned long long)s_tot[5] * 20 * sizeof(double);
    const bool fits = a.cap == 0 || need <= a.cap;
    const long long mirror = a.res2 ? a.res2 - a.res : 0;  // byte distance to the second copy
#define PD_PUT(ptr, value)                                                              \
    do {                                                                                \
        auto* p_ = (ptr);                                                               \
        const auto v_ = (value);                                                        \
        *p_ = v_;                                                                       \
        if (a.res2) *(decltype(p_))((unsigned char*)p_ + mirror) = v_;                  \
    } while (0)
    if (b == 0 && tid == 0) {
        int64_t* head = (int64_t*)a.res;
        const int fb = min(min(s_redbad[0], s_redbad[1]), min(s_redbad[2], s_redbad[3]));
        PD_PUT(head + 0, (int64_t)a.B);
        PD_PUT(head + 1, (int64_t)s_tot[3]);
        PD_PUT(head + 2, (int64_t)s_tot[4]);
        PD_PUT(head + 3, (int64_t)s_tot[5]);
        PD_PUT(head + 4, (int64_t)(fb == 0x7fffffff ? -1 : fb));
        PD_PUT(head + 5, (int64_t)(fits ? 0 : (int64_t)need));  // != 0: the buffer is too small, no payload was written
    }
    // count tables of this image
    if (tid < PD_N_PARTS)
        PD_PUT((int32_t*)(a.res + a.lay.off_part) + b * PD_N_PARTS + tid, a.have_peaks ? a.part_count[b * PD_N_PARTS + tid] : 0);
    if (tid < PD_N_LIMBS)
        PD_PUT((int32_t*)(a.res + a.lay.off_limb) + b * PD_N_LIMBS + tid, a.have_conns ? a.limb_count[b * PD_N_LIMBS + tid] : 0);
    if (tid == 0) {
        PD_PUT((int32_t*)(a.res + a.lay.off_pers) + b, a.have_persons ? a.person_count[b] : 0);
        PD_PUT((int32_t*)(a.res + a.lay.off_stat) + b, a.status[b]);
    }
    const long long n_peaks = s_tot[3], n_conns = s_tot[4];
    pd_peak* out_peaks = (pd_peak*)(a.res + a.lay.off_payload) + s_tot[0];
    pd_conn* out_conns = (pd_conn*)(a.res + a.lay.off_payload + n_peaks * sizeof(pd_peak)) + s_tot[1];
    double* out_persons = (double*)(a.res + a.lay.off_payload + n_peaks * sizeof(pd_peak) + n_conns * sizeof(pd_conn)) +
                          s_tot[2] * 20;
    if (a.have_peaks && fits) {
        const int n = s_pbase[PD_N_PARTS];
        for (int i = tid; i < n; i += 256) {
            int p = 0;
            while (i >= s_pbase[p + 1]) ++p;
            const size_t o = ((size_t)b * PD_N_PARTS + p) * a.PK + (i - s_pbase[p]);
            pd_peak q;
            q.y = a.pk_y[o];
            q.x = a.pk_x[o];
            q.score = a.pk_score[o];
            PD_PUT(out_peaks + i, q);
        }
    }
    if (a.have_conns && fits) {
        const int n = s_lbase[PD_N_LIMBS];
        for (int i = tid; i < n; i += 256) {
            int k = 0;
            while (i >= s_lbase[k + 1]) ++k;
            PD_PUT(out_conns + i, a.conn[((size_t)b * PD_N_LIMBS + k) * a.PK + (i - s_lbase[k])]);
        }
    }
    if (a.have_persons && fits) {
        const int n = a.person_count[b] * 20;
        const double* src = a.persons + (size_t)b * a.max_persons * 20;
        for (int i = tid; i < n; i += 256) PD_PUT(out_persons + i, src[i]);
    }
#undef PD_PUT
}

// ---- person assembly and packing in one launch ----------------------------------------
// One workgroup of four wavefronts per image.  Wavefront 0 assembles the persons of the image
// (the body of k_parse_wave); meanwhile the other three, each on its own, sum the peak and
// connection counts of all images (which the earlier launches left complete) and copy this
// image's count tables, peaks and connections to their places in the result buffer.
//
// The person rows of an image start behind those of all images before it.  As soon as wavefront
// 0 knows its image's row count it publishes (tag | not-ok bit | count) in a word of its own and
// then reads the words of all images before it, waiting for those not published yet (a
// decoupled look-back: nobody waits for a later image).  It then writes its rows from its
// registers straight to their place.  The image a workgroup works on is the number it draws from
// an arrival counter, not its position in the grid: every image a workgroup waits for therefore
// belongs to a workgroup that is already running, whatever the order the hardware starts them in.
// The workgroup of the last image has seen every count: it writes the header, gives the words of
// the images beyond this batch the tag as well and stores the tag for the next launch, whose tag
// differs -- every word always carries the previous launch's tag, so no word needs clearing.
struct ParsePackArgs {
    ParseArgs p;
    PackArgs k;
    int32_t* sync;  // [0] arrival counter (zero between launches), [1] tag of the last launch, [PP_WORDS + j] word of image j
    int max_batch;  // words behind PP_WORDS
};

static_assert(sizeof(pd_conn) == 24, "k_parse_pack copies a connection as three 8-byte words");
#define PP_WORDS 16            // ints in front of the published words
#define PP_NOT_OK 512          // bit of a published word: the image's status is not PD_OK
#define PP_TAG_SHIFT 10        // the tag sits above count (< 512) and PP_NOT_OK
#define PP_TAG_MASK 0x1fffff


__global__ void __launch_bounds__(256)
k_parse_pack(ParsePackArgs a) {
    PD_RAISE_PRIORITY(1);
    __shared__ WaveShared sh;
    __shared__ int s_pb[3][PD_N_PARTS + 1], s_lb[3][PD_N_LIMBS + 1];
    __shared__ long long s_tot[2];
    __shared__ int s_b, s_tag;
    const PackArgs& k = a.k;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    PP_STAMP(0, tid == 0);
    WaveState st;
    int t = 0, last_tag = 0;
    if (tid == 0) {
        t = __hip_atomic_fetch_add(a.sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (the workgroup that stores the new tag does so after every other one has published, so after this load)
        last_tag = __hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wv == 0) wave_parse_begin(sh, st, lane);  // (while the two answers are on their way)
    if (tid == 0) {
        if (t == (int)gridDim.x - 1) __hip_atomic_store(a.sync, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // all have drawn
        s_b = t;
        s_tag = (last_tag + 1) & PP_TAG_MASK;
    }
    __syncthreads();
    const int b = s_b, tag = s_tag;
    PP_STAMP(1, tid == 0);
    // every stage that reads the per-decode counters has run: leave them zero for the next decode
    if (k.zero)
        for (int i = b * 256 + tid; i < k.zero_words; i += gridDim.x * 256) k.zero[i] = 0;
    const long long mirror = k.res2 ? k.res2 - k.res : 0;  // byte distance to the second copy
#define PD_PUT(ptr, value)                                                \
    do {                                                                  \
        auto* p_ = (ptr);                                                 \
        const auto v_ = (value);                                          \
        *p_ = v_;                                                         \
        if (k.res2) *(decltype(p_))((unsigned char*)p_ + mirror) = v_;    \
    } while (0)
    bool keep = false, direct = false;
    int rank = 0, count = 0, stv = PD_OK;
    long long pre = 0;  // person rows of the images before this one
    int bad = 0x7fffffff;
    if (wv == 0) {
        wave_parse_limbs(a.p, b, lane, 0, PD_N_LIMBS, sh, st);
        count = wave_parse_prune(a.p, b, lane, sh, st, keep, rank, direct);
        // (lane 0 may have stored this status itself just now: same lane, same address, in order)
        stv = __builtin_amdgcn_readfirstlane(lane == 0 ? __hip_atomic_load(a.p.status + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0);
        if (lane == 0)
            __hip_atomic_store(a.sync + PP_WORDS + b, (tag << PP_TAG_SHIFT) | (stv != PD_OK ? PP_NOT_OK : 0) | count, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        PP_STAMP(2, lane == 0);
        if (stv != PD_OK) bad = b;
        int rows = 0;
        for (int j = lane; j < b; j += 64) {
            int v = __hip_atomic_load(a.sync + PP_WORDS + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while ((v >> PP_TAG_SHIFT) != tag) {
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(a.sync + PP_WORDS + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            rows += v & (PP_NOT_OK - 1);
            if (v & PP_NOT_OK) bad = min(bad, j);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            rows += __shfl_xor(rows, d);
            bad = min(bad, __shfl_xor(bad, d));
        }
        pre = rows;
        PP_STAMP(6, lane == 0);
    } else {
        const int w = wv - 1;
        long long pre_p = 0, tot_p = 0, pre_c = 0, tot_c = 0;
        for (int j = lane; j < k.B; j += 64) {
            const long long np = k.part_base[j * (PD_N_PARTS + 1) + PD_N_PARTS];
            tot_p += np;
            if (j < b) pre_p += np;
        }
        if (k.have_conns)
            for (int e = lane; e < k.B * PD_N_LIMBS; e += 64) {
                const long long nc = max(0, k.limb_count[e]);
                tot_c += nc;
                if (e / PD_N_LIMBS < b) pre_c += nc;
            }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            pre_p += __shfl_xor(pre_p, d);
            tot_p += __shfl_xor(tot_p, d);
            pre_c += __shfl_xor(pre_c, d);
            tot_c += __shfl_xor(tot_c, d);
        }
        // where the parts / limbs of this image start in its runs of peaks / connections
        if (lane <= PD_N_PARTS) s_pb[w][lane] = k.part_base[b * (PD_N_PARTS + 1) + lane];
        const int my_nc = lane < PD_N_LIMBS && k.have_conns ? max(0, k.limb_count[b * PD_N_LIMBS + lane]) : 0;
        int incl = my_nc;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        if (lane < PD_N_LIMBS) s_lb[w][lane + 1] = incl;
        if (lane == 0) s_lb[w][0] = 0;
        pd_wave_sync();
        if (w == 0) {
            if (lane < PD_N_PARTS) PD_PUT((int32_t*)(k.res + k.lay.off_part) + b * PD_N_PARTS + lane, k.part_count[b * PD_N_PARTS + lane]);
            if (lane < PD_N_LIMBS) PD_PUT((int32_t*)(k.res + k.lay.off_limb) + b * PD_N_LIMBS + lane, k.have_conns ? k.limb_count[b * PD_N_LIMBS + lane] : 0);
            if (lane == 0) {
                s_tot[0] = tot_p;
                s_tot[1] = tot_c;
            }
        }
        pd_peak* out_peaks = (pd_peak*)(k.res + k.lay.off_payload) + pre_p;
        pd_conn* out_conns = (pd_conn*)(k.res + k.lay.off_payload + tot_p * sizeof(pd_peak)) + pre_c;
        // a caller's buffer (cap != 0) gets its peaks and connections only if all of them fit
        const bool lists_fit = k.cap == 0 || k.lay.off_payload + (unsigned long long)tot_p * sizeof(pd_peak) +
                                                     (unsigned long long)tot_c * sizeof(pd_conn) <= k.cap;
        const int n_p = lists_fit ? s_pb[w][PD_N_PARTS] : 0, n_c = lists_fit ? s_lb[w][PD_N_LIMBS] : 0;
        for (int i = w * 64 + lane; i < n_p; i += 192) {
            int p = 0;
            while (i >= s_pb[w][p + 1]) ++p;
            const size_t o = ((size_t)b * PD_N_PARTS + p) * k.PK + (i - s_pb[w][p]);
            pd_peak q;
            q.y = k.pk_y[o];
            q.x = k.pk_x[o];
            q.score = k.pk_score[o];
            PD_PUT(out_peaks + i, q);
        }
        for (int i = w * 64 + lane; i < n_c; i += 192) {
            int l = 0;
            while (i >= s_lb[w][l + 1]) ++l;
            // (three 8-byte words: a 24-byte struct copy goes through the stack)
            const unsigned long long* src = (const unsigned long long*)(k.conn + ((size_t)b * PD_N_LIMBS + l) * k.PK + (i - s_lb[w][l]));
            unsigned long long* dst = (unsigned long long*)(out_conns + i);
            const unsigned long long c0 = src[0], c1 = src[1], c2 = src[2];
            PD_PUT(dst, c0);
            PD_PUT(dst + 1, c1);
            PD_PUT(dst + 2, c2);
        }
        PP_STAMP(2 + wv, lane == 0);
    }
    __syncthreads();  // the totals of peaks and connections (wavefront 1) reach wavefront 0
    if (wv != 0) return;
    const long long n_peaks = s_tot[0], n_conns = s_tot[1];
    double* out_persons = (double*)(k.res + k.lay.off_payload + n_peaks * sizeof(pd_peak) + n_conns * sizeof(pd_conn)) + pre * 20;
    double* staged = a.p.persons + (size_t)b * a.p.max_persons * 20;
    // bytes of the buffer up to the end of this image's rows: a caller's buffer (cap != 0) gets the rows of the images
    // that fit; the last image reports what the whole result needs, and a reader looks at that before the payload
    const unsigned long long rows_end = k.lay.off_payload + (unsigned long long)n_peaks * sizeof(pd_peak) +
                                        (unsigned long long)n_conns * sizeof(pd_conn) + (unsigned long long)(pre + count) * 20 * sizeof(double);
    const bool rows_fit = k.cap == 0 || rows_end <= k.cap;
    if (direct) {
        if (keep) {
            wave_write_row(staged + (size_t)rank * 20, 0, lane, sh, st);
            if (rows_fit) wave_write_row(out_persons + (size_t)rank * 20, mirror, lane, sh, st);
        }
    } else if (count > 0 && rows_fit) {
        // the table form left the rows in the staging area (stored write-through; drained here)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int i = lane; i < count * 20; i += 64) {
            const unsigned long long v = __hip_atomic_load((const unsigned long long*)staged + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            PD_PUT((unsigned long long*)out_persons + i, v);
        }
    }
    if (lane == 0) {
        PD_PUT((int32_t*)(k.res + k.lay.off_pers) + b, (int32_t)count);
        PD_PUT((int32_t*)(k.res + k.lay.off_stat) + b, (int32_t)stv);
    }
    if (b == (int)gridDim.x - 1) {
        if (lane == 0) {
            int64_t* head = (int64_t*)k.res;
            PD_PUT(head + 0, (int64_t)k.B);
            PD_PUT(head + 1, (int64_t)n_peaks);
            PD_PUT(head + 2, (int64_t)n_conns);
            PD_PUT(head + 3, (int64_t)(pre + count));
            PD_PUT(head + 4, (int64_t)(bad == 0x7fffffff ? -1 : bad));
            PD_PUT(head + 5, (int64_t)(rows_fit ? 0 : rows_end));  // != 0: the caller's buffer is too small, ignore the payload
        }
        for (int j = k.B + lane; j < a.max_batch; j += 64)
            __hip_atomic_store(a.sync + PP_WORDS + j, tag << PP_TAG_SHIFT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) __hip_atomic_store(a.sync + 1, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PP_STAMP(7, lane == 0);
#undef PD_PUT
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
    if (lds > 64 * 1024) {  // beyond the default limit of dynamic LDS (pd_create keeps max_persons within 150 KB)
        static size_t allowed[64];  // per device
        size_t& have = allowed[ctx->cfg.device_id >= 0 && ctx->cfg.device_id < 64 ? ctx->cfg.device_id : 0];
        if (have < lds) {
            PD_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_parse), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            have = 150 * 1024;
        }
    }
    pd_prof_begin(ctx, PD_STAGE_PARSE);
    a.dbg_found = ctx->dbg_found;
    if (ctx->impl == PD_IMPL_SIMPLE || ctx->dbg_found) {
        a.only_flagged = 0;
        hipLaunchKernelGGL(k_parse, dim3(ctx->B), dim3(64), lds, ctx->stream, a);
    } else {
        // one lane per person; images with more than 64 open persons take the table form, inside
        // the same kernel when its rows fit the staging area, else in a kernel of its own
        a.only_flagged = (size_t)a.max_persons * 20 > (size_t)WAVE_CONNS * 4;
        hipLaunchKernelGGL(k_parse_wave, dim3(ctx->B), dim3(64), 0, ctx->stream, a);
        if (a.only_flagged) hipLaunchKernelGGL(k_parse, dim3(ctx->B), dim3(64), lds, ctx->stream, a);
    }
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    return PD_OK;
}

static void fill_pack_args(pd_ctx* ctx, PackArgs& a) {
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
    a.res = ctx->res_dev;
    a.res2 = ctx->pack_direct && ctx->res_user_cap == 0 ? ctx->res_host : nullptr;
    a.cap = ctx->res_user_cap;
    a.lay = pd_res_layout(ctx->B);
    a.zero = ctx->zero_block;
    a.zero_words = (int)(ctx->zero_bytes / sizeof(int32_t));
}

// Whether person assembly and packing of the current decode can share a launch: one destination
// (the library's own result buffer, or a skeleton buffer alone, whose capacity the kernel checks as it
// goes), the wavefront form of the assembly with its hand-over inside the kernel.  PD_SEPARATE_PACK=1 in
// the environment keeps the two launches (comparison runs).
bool pd_parse_pack_fused(const pd_ctx* ctx) {
    static const bool separate = [] {
        const char* v = getenv("PD_SEPARATE_PACK");
        return v && *v && *v != '0';
    }();
    // (either the library's own result buffer, or a skeleton buffer alone: with both, two packing launches follow the assembly)
    const bool own = !ctx->skel_dst && ctx->host_results && ctx->res_user_cap == 0;
    const bool skeleton_alone = ctx->skel_dst && !ctx->host_results;
    return !separate && ctx->pack_sync && ctx->impl != PD_IMPL_SIMPLE && !ctx->dbg_found && (own || skeleton_alone) &&
           ctx->have_peaks && ctx->have_conns &&
           (size_t)ctx->cfg.max_persons * 20 <= (size_t)WAVE_CONNS * 4 && ctx->cfg.max_persons < PP_NOT_OK;
}

int pd_launch_parse_pack(pd_ctx* ctx) {
    ParsePackArgs a;
    a.p.mode = ctx->mode;
    a.p.PK = ctx->cfg.max_peaks_per_part;
    a.p.max_persons = ctx->cfg.max_persons;
    a.p.conn = ctx->conn;
    a.p.conn_conf = ctx->conn_conf;
    a.p.limb_count = ctx->limb_count;
    a.p.persons = ctx->persons;
    a.p.person_count = ctx->person_count;
    a.p.status = ctx->status;
    a.p.only_flagged = 0;
    a.p.dbg_found = nullptr;
    ctx->have_persons = 1;
    fill_pack_args(ctx, a.k);
    if (ctx->skel_dst) {  // the caller's device buffer: the same layout without the connection lists, its capacity checked
        a.k.res = ctx->skel_dst;
        a.k.res2 = nullptr;
        a.k.cap = ctx->skel_cap;
        a.k.have_conns = 0;
    }
    a.sync = ctx->pack_sync;
    a.max_batch = ctx->cfg.max_batch;
    pd_prof_begin(ctx, PD_STAGE_PARSE);
    hipLaunchKernelGGL(k_parse_pack, dim3(ctx->B), dim3(256), 0, ctx->stream, a);
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    ctx->zero_clean = 1;
    return PD_OK;
}

int pd_launch_pack(pd_ctx* ctx) {
    PackArgs a;
    fill_pack_args(ctx, a);
    pd_prof_begin(ctx, PD_STAGE_PACK);
    if (!ctx->skel_dst || ctx->host_results) {
        hipLaunchKernelGGL(k_pack, dim3(ctx->B), dim3(256), 0, ctx->stream, a);
        a.zero = nullptr;
    }
    if (ctx->skel_dst) {
        // the caller's device buffer gets the same layout without the connection lists
        a.res = ctx->skel_dst;
        a.res2 = nullptr;
        a.cap = ctx->skel_cap;
        a.have_conns = 0;
        hipLaunchKernelGGL(k_pack, dim3(ctx->B), dim3(256), 0, ctx->stream, a);
    }
    pd_prof_end(ctx);
    PD_HIP(ctx, hipGetLastError());
    ctx->zero_clean = 1;
    return PD_OK;
}
