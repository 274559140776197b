// mrz_seq_wide.h -- the WIDE BATCH ENGINE of the sequencer workgroup.
//
// hash_search's loop (src/rzip.c:548-599) must be replayed in order, but most of what a candidate costs is
// read-only: the probe walk of find_best_match / insert_hash (:426-462, :262-297) and the byte compares of
// single_match_len (:372-397).  A batch takes the next MRZ_W (= 64 x waves) candidates of the position stream,
// ONE LANE PER CANDIDATE, and runs in phases, all waves of the workgroup together:
//
//   A  formation   the next entries of the front end's candidate list (a few per thread: the list was made under the
//                  mask of a launch or two ago, the ones that still pass minimum_tag_mask are compacted into the
//                  lanes); the cull window: a bitmask of the entries ahead of tag_clean_ptr that fail the next
//                  mask (clean_one_from_hash, :313-321);
//   B  walk        every lane walks its probe chain in the table AS IT STANDS (8 slots = one 128-B line per step,
//                  long chains are finished by the lane's whole wave, 64 slots per step): first empty slot, the
//                  tag-equal entries in probe order, where insert_hash's walk stops (empty / due-for-culling
//                  overwrite / lower-ranked occupant to displace / max_chain_len-th tag-equal entry => eviction)
//                  and the walk of a displaced occupant;
//   B2 conflicts   the slots a lane would write go into a hash of 64-slot blocks in LDS; a lane whose read
//                  ranges hold a write of an EARLIER lane is marked stale.  A stale lane whose writers are all
//                  sound walks once more with those writes laid over the table (overlay: the common case is the
//                  next position having the same tag -- the XOR tag only sees the multiset of its 31 bytes);
//   C  pairs       all (candidate, tag-equal entry) pairs are laid end to end and dealt one per thread: 64 bytes
//                  each way, stored RAW (independent of last_match) so that they can be re-bounded later;
//   E  commit      segments of lanes are committed in order.  A segment ends at the first lane that is stale,
//                  needs the cooperative path, or has a match running past the 64-byte reach.  Within a segment
//                  workgroup-wide scans give the sequential quantities: victim_round per evicting lane,
//                  hash_count before each lane (saturating prefix sum), which lanes cull and which sweep entry
//                  each culls (rank into the cull window), and the lazy-match fold (:586-599) as a prefix
//                  maximum (first longest wins) whose first lane meeting the emit rule (:592) ends the segment.
//                  After an emission the batch goes on behind the match (the lanes inside it are dropped); a
//                  match that ends before the emitting position (p jumps BACK, :596) re-executes that one lane.
//                  A stale lane at the head of a segment walks again against the table as committed so far.
//
// What is committed is exactly what the reference's loop would have done: a lane only commits if no earlier
// lane's write (insert, displaced re-insert, cull) lies in the slots it read -- or it has read them again since.
#pragma once
#include "mrz_seq_common.h"

#define MRZ_W MRZ_SEQ_THREADS
#ifndef MRZ_POOL
#define MRZ_POOL (2 * MRZ_W)  // chunks of 4 tag-equal entries: 8 entries per lane on average (a lane holds up to 16)
#endif
#define MRZ_PAIR_MAX (MRZ_POOL * 4)
#define MRZ_BH_SIZE 2048              // block-hash entries (64-slot blocks written by this batch)
#define MRZ_BH_WRITERS 5
#define MRZ_BHO_MAX 192
#ifndef MRZ_OV_ROUNDS
#define MRZ_OV_ROUNDS 3  // overlay rounds of a preparation (a repaired writer makes its readers repairable)
#endif
#ifndef MRZ_CW_WORDS
#define MRZ_CW_WORDS 32  // cull window: 32 x 64 slots ahead of tag_clean_ptr
#endif
#define MRZ_NW_MAX 192
#define MRZ_XW_MAX 48
#ifndef MRZ_BULK_MIN
#ifndef MRZ_BULK_MIN
#define MRZ_BULK_MIN 96  // leading lanes worth a workgroup-wide bulk commit
#endif
#endif
#ifndef MRZ_WR_MAX
#define MRZ_WR_MAX 3   // earlier writers of this batch an overlay walk can take into account
#endif
#define MRZ_OV_MAX (2 * MRZ_WR_MAX)  // overlay entries of one lane: insert + occupant slot of each writer
#define MRZ_OFF_BITS 40
#define MRZ_OFF_MASK ((1ull << MRZ_OFF_BITS) - 1)

struct mrz_chunk4 {
    unsigned long long e[4];  // offset | slot << 40
    unsigned short raw[4];    // fwd (7 bits) | fwd may continue << 7 | bwd (7 bits) << 8
};

struct mrz_wide_lds {
    int ctl[16];          // control words between wave 0 and the others
    int nb, total;        // lanes of the prepared batch; candidates in its bitmap window
    int bulk_y, bulk_a1, bulk_a2, bulk_end;  // bulk commit done by the pre-commit step: lanes [first_live, bulk_y), totals
    int first_live;       // first lane behind L.p (where the commit starts or resumes)
    int rank0;            // culls of this batch's cull window used up before that lane
    long long snap64[10];  // the snapshot a preparation works from (token, epoch, window base, masks)
    mrz_lead lead;        // the matcher's state after a commit, for the waves that did not run it
    unsigned long long hand[32];  // the hand-over block of mrz_wide_shared: loaded at a turn, stored at its end
    int64_t prep_min_mask, prep_tag_mask;  // the masks the batch was prepared under
    int64_t cw_base, w_end, floor_prep;
    int64_t adv_to;  // see mrz_wide_prep
    int cw_len;
    // per-lane facts other lanes / waves need
    int64_t q[MRZ_W];
    int h[MRZ_W], len1[MRZ_W], h2[MRZ_W], len2[MRZ_W], wslot[MRZ_W], w2[MRZ_W];
    unsigned char kind[MRZ_W], kind2[MRZ_W], ns[MRZ_W], lf[MRZ_W], nchunk[MRZ_W];
    int64_t t[MRZ_W], occ_off[MRZ_W], occ_t[MRZ_W];
    unsigned short supp_w[MRZ_W], supp_w2[MRZ_W];  // first later lane that overwrites this lane's insert / occupant slot
    int nw_cnt;                      // writes of overlay-walked lanes: slots other lanes have to re-check
    int nw_slot[MRZ_NW_MAX];
    unsigned short nw_lane[MRZ_NW_MAX];
    // what the committing wave needs of every lane (written by the lane's own thread at the end of the preparation)
    int64_t blen[MRZ_W], boff[MRZ_W];  // best match of the lane's entries under the last_match of the preparation
    int brev[MRZ_W];
    unsigned short bhm[MRZ_W];         // tag_hits << 8 | tag_misses of that look-up
    unsigned short dep[MRZ_WR_MAX][MRZ_W];  // lanes whose speculated writes an overlay walk has assumed (MRZ_W: none)
    unsigned char exec[MRZ_W];         // 0 not yet, 1 committed as prepared, 2 dropped / went through the cooperative path
    int xw_n;                          // slots written by cooperative hand-overs inside this batch
    int xw_slot[MRZ_XW_MAX];
    int pbase[MRZ_W];
    unsigned short chunk_id[MRZ_W][4];
    unsigned short pair_owner[MRZ_PAIR_MAX];
    mrz_chunk4 pool[MRZ_POOL];
    unsigned bh_key[MRZ_BH_SIZE];
    unsigned bh_cnt[MRZ_BH_SIZE];
    unsigned short bh_lane[MRZ_BH_SIZE][MRZ_BH_WRITERS];
    int bho_n;  // writers beyond MRZ_BH_WRITERS of a block: (block key, lane)
    unsigned bho_key[MRZ_BHO_MAX];
    unsigned short bho_lane[MRZ_BHO_MAX];
    mrz_u64 cw[MRZ_CW_WORDS];
    int cwcum[MRZ_CW_WORDS + 1];
    int cw_list[MRZ_W];   // slot of the r-th failing entry of the window, r < min(cwcum[last], MRZ_W)
    int wt1[MRZ_SEQ_WAVES], wt3[MRZ_SEQ_WAVES], wt5[MRZ_SEQ_WAVES];
    unsigned long long wt2[MRZ_SEQ_WAVES], wt4[MRZ_SEQ_WAVES];
    int wmin[4][MRZ_SEQ_WAVES];
    int pool_top;
    mrz_coop_lds coop;
};

enum { MRZ_CTL_MODE, MRZ_CTL_WIDTH };

// ---- workgroup-wide scans (all threads; one barrier each; `wt` must not be reused before another barrier) ----
template <int NW>
__device__ __forceinline__ void mrz_prep_sync() {
    if (NW == 1)
        MRZ_WAVE_SYNC();
    else
        __syncthreads();
}

template <int NW>
__device__ __forceinline__ int mrz_wide_incl(int v, int *wt, int lane, int wave, int *total) {
    const int incl = mrz_wave_incl_sum(v, lane);
    if (NW == 1) {
        *total = mrz_lane_read(incl, 63);
        return incl;
    }
    if (lane == 63) wt[wave] = incl;
    __syncthreads();
    int add = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const int x = wt[w];
        tot += x;
        if (w < wave) add += x;
    }
    *total = tot;
    return incl + add;
}

__device__ __forceinline__ mrz_u64 mrz_wide_incl64(mrz_u64 v, unsigned long long *wt, int lane, int wave) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const mrz_u64 o = (mrz_u64)mrz_shfl64((int64_t)v, lane - d >= 0 ? lane - d : lane);
        if (lane >= d) v += o;
    }
    if (lane == 63) wt[wave] = v;
    __syncthreads();
    mrz_u64 add = 0;
#pragma unroll
    for (int w = 0; w < MRZ_SEQ_WAVES; w++)
        if (w < wave) add += wt[w];
    return v + add;
}

__device__ __forceinline__ mrz_u64 mrz_wide_inclmax64(mrz_u64 v, unsigned long long *wt, int lane, int wave) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const mrz_u64 o = (mrz_u64)mrz_shfl64((int64_t)v, lane - d >= 0 ? lane - d : lane);
        if (lane >= d && o > v) v = o;
    }
    if (lane == 63) wt[wave] = v;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < MRZ_SEQ_WAVES; w++)
        if (w < wave && wt[w] > v) v = wt[w];
    return v;
}

// first thread (lowest tid) for which `flag` holds, or `none`; up to 4 independent reductions share one barrier
__device__ __forceinline__ int mrz_wave_first(bool flag, int wave, int none) {
    const mrz_u64 m = __ballot(flag);
    return m ? wave * 64 + (__ffsll((long long)m) - 1) : none;
}
__device__ __forceinline__ int mrz_wide_min_read(const int *wm) {
    int r = wm[0];
#pragma unroll
    for (int w = 1; w < MRZ_SEQ_WAVES; w++) r = wm[w] < r ? wm[w] : r;
    return r;
}

#ifdef MRZ_SEQ_COUNTS
// diagnostics: number of threads of the workgroup for which `flag` holds (all threads call; two barriers each, so
// these only exist in -DMRZ_SEQ_COUNTS builds and not in the timing build)
#define ST_COUNT(k, flag)                                                                           \
    do {                                                                                            \
        const mrz_u64 m__ = __ballot(flag);                                                         \
        int t__;                                                                                    \
        (void)mrz_wide_incl<NW>(lane == 0 ? __popcll(m__) : 0, S->wt5, lane, wave, &t__);           \
        mrz_prep_sync<NW>();                                                                            \
        ST_ADD(k, t__);                                                                             \
    } while (0)
#else
#define ST_COUNT(k, flag) ((void)0)
#endif

// ---- tag-equal entries of a lane: chunks of 4 out of a pool in LDS ----
__device__ __forceinline__ bool mrz_pool_put(mrz_wide_lds *S, int gl, int idx, int64_t off, int slot) {
    if ((idx >> 2) >= S->nchunk[gl]) {  // (a lane that walks again keeps its chunks)
        // the chunk of this entry has not been handed out: now, if this is its first entry -- else the pool had run
        // dry when it was asked for (the walk goes on counting, the lane is complex): nothing to store into
        if ((idx & 3) != 0) return false;
        const int c = atomicAdd(&S->pool_top, 1);
        if (c >= MRZ_POOL) return false;
        S->chunk_id[gl][idx >> 2] = (unsigned short)c;
        S->nchunk[gl] = (unsigned char)((idx >> 2) + 1);
    }
    const int c = S->chunk_id[gl][idx >> 2];
#ifdef MRZ_EMU_LDS_PER_BLOCK
    if (c >= MRZ_POOL) __builtin_trap();  // (test emulator: ids are poisoned at the start of a batch)
#endif
    S->pool[c].e[idx & 3] = (unsigned long long)off | ((unsigned long long)(unsigned)slot << MRZ_OFF_BITS);
    return true;
}
__device__ __forceinline__ unsigned long long mrz_pool_get(const mrz_wide_lds *S, int gl, int idx) {
    return S->pool[S->chunk_id[gl][idx >> 2]].e[idx & 3];
}

// raw 64-byte probe of one (candidate, entry) pair, independent of last_match (single_match_len,
// src/rzip.c:372-397): bits 0-6 equal bytes forward (capped by end - q), bit 7 "forward runs past the reach",
// bits 8-14 equal bytes backward among the 64 before (pieces that would start before byte 0 of the chunk count
// as equal: mrz_pair_eval knows from `op` which ones those are)
__device__ static unsigned mrz_lane_probe_raw(const uint8_t *__restrict__ buf, int64_t q, int64_t op, int64_t end) {
    if (op >= q) return 0;
    int64_t maxf = end - q;
    if (maxf < 0) maxf = 0;
    const int64_t last_ok = end + (MRZ_MIN_MATCH - 16);
    uint4 fa[4], fb[4], ba[4], bb[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int64_t pa = q + j * 16, pb = op + j * 16;
        pa = pa < last_ok ? pa : last_ok;
        pb = pb < last_ok ? pb : last_ok;
        fa[j] = mrz_ld16(buf + pa);
        fb[j] = mrz_ld16(buf + pb);
        int64_t ra = q - (j + 1) * 16, rb = op - (j + 1) * 16;
        ra = ra > 0 ? ra : 0;
        rb = rb > 0 ? rb : 0;
        ba[j] = mrz_ld16(buf + ra);
        bb[j] = mrz_ld16(buf + rb);
    }
    int rawf;
    {
        const int d0 = mrz_first_diff16_bf(fa[0], fb[0]), d1 = mrz_first_diff16_bf(fa[1], fb[1]);
        const int d2 = mrz_first_diff16_bf(fa[2], fb[2]), d3 = mrz_first_diff16_bf(fa[3], fb[3]);
        rawf = d0 < 16 ? d0 : 16 + (d1 < 16 ? d1 : 16 + (d2 < 16 ? d2 : 16 + d3));
    }
    const int fwd = rawf < maxf ? rawf : (int)maxf;
    const unsigned lngf = (rawf == 64 && maxf > 64) ? 1u : 0u;
    const int edge = op < 64 ? (int)(op >> 4) : 4;
    int rawb;
    {
        const int e0 = edge > 0 ? mrz_top_equal16_bf(ba[0], bb[0]) : 16, e1 = edge > 1 ? mrz_top_equal16_bf(ba[1], bb[1]) : 16;
        const int e2 = edge > 2 ? mrz_top_equal16_bf(ba[2], bb[2]) : 16, e3 = edge > 3 ? mrz_top_equal16_bf(ba[3], bb[3]) : 16;
        rawb = e0 < 16 ? e0 : 16 + (e1 < 16 ? e1 : 16 + (e2 < 16 ? e2 : 16 + e3));
    }
    return (unsigned)fwd | (lngf << 7) | ((unsigned)rawb << 8);
}

// single_match_len's result for a raw probe under the current last_match; *lng: the match runs past the reach
__device__ __forceinline__ void mrz_pair_eval(unsigned raw, int64_t q, int64_t op, int64_t floor_p, int64_t *len,
                                              int *rev, bool *lng) {
    *len = 0;
    *rev = 0;
    *lng = false;
    if (op >= q) return;
    const int fwd = (int)(raw & 127u), rawb = (int)((raw >> 8) & 127u);
    int64_t maxb = q - floor_p;
    if (op < maxb) maxb = op;
    if (maxb < 0) maxb = 0;
    const int edge = op < 64 ? (int)(op >> 4) : 4;
    const bool l = ((raw >> 7) & 1u) || (rawb == 64 && maxb > 64) || (edge < 4 && rawb >= 16 * edge && maxb > 16 * edge);
    if (l) {
        *lng = true;
        return;
    }
    const int rv = rawb < maxb ? rawb : (int)maxb;
    *rev = rv;
    const int tot = fwd + rv;
    *len = tot >= MRZ_MIN_MATCH ? tot : 0;
}

// ---- phase B: the probe walk of one lane (and of the occupant it displaces) -----------------------------
struct mrz_ov {  // writes of earlier lanes laid over the table for an overlay walk
    int n;
    int slot[MRZ_OV_MAX];
    int64_t off[MRZ_OV_MAX], t[MRZ_OV_MAX];
};

struct mrz_wl {
    int fe, wslot, kind, nsame, h2, w2, kind2, len2;
    bool cplx;
    int why;  // first reason for cplx (diagnostics): 1 walk budget, 2 cascade, 3 pool, 4 too many tag-equal entries
    int64_t occ_t, occ_off;
};

__device__ __forceinline__ mrz_slot mrz_tab_load(const mrz_slot *tab, int slot, const mrz_ov *ov) {
    mrz_slot e = tab[slot];
    if (ov) {
#pragma unroll
        for (int k = 0; k < MRZ_OV_MAX; k++)
            if (k < ov->n && ov->slot[k] == slot) {  // a later overlay entry wins (written later)
                e.off = ov->off[k];
                e.t = ov->t[k];
            }
    }
    return e;
}

// `go`: this lane walks (the others only take part in the wave-cooperative finish of long chains).  `ov`: per-lane
// overlay (nullptr-equivalent when ov.n == 0 and USE_OV is false).
template <bool USE_OV>
__device__ static void mrz_wide_walk(const mrz_cfg &C, mrz_wide_lds *S, bool go, int gl, int lane, int64_t t, bool do_ins,
                                     int64_t better, const mrz_ov &ovl, mrz_wl &o) {
    const mrz_slot *tab = C.tab;
    const int smask = (int)C.slot_mask;
    const int max_chain = (int)C.max_chain;
    const int h = (int)(t & C.slot_mask);
    const int my_rank = mrz_ones_rank(t);
    const mrz_ov *ov = USE_OV ? &ovl : nullptr;
    int fe = -1, wslot = -1, kind = -1;  // kind: 0 empty, 1 overwrite, 2 displace, 3 evict
    int nsame = 0, round = 0;
    bool cplx = false, evict = false;
    int why = 0;
    int64_t occ_t = 0, occ_off = 0;
    const int gl0 = gl - lane;  // first lane of this wave
#define MRZ_CPLX(code)         \
    do {                       \
        cplx = true;           \
        if (!why) why = (code); \
    } while (0)
    {
        bool walking = go;
        int s = h, steps = 0;
        int ustep = 0;
        while (true) {
            const mrz_u64 m_walk = __ballot(walking);
            if (!m_walk) break;
            if (ustep >= MRZ_WALK_LANE_STEPS && __popcll(m_walk) <= MRZ_WALK_COOP_MAX) break;
            ustep++;
            if (walking) {
                mrz_slot e[MRZ_WALK_SLOTS];
#pragma unroll
                for (int k = 0; k < MRZ_WALK_SLOTS; k++) e[k] = mrz_tab_load(tab, (s + k) & smask, ov);
#pragma unroll
                for (int k = 0; k < MRZ_WALK_SLOTS; k++) {
                    if (!walking) continue;
                    const int slot = (s + k) & smask;
                    if ((e[k].off | e[k].t) == 0) {
                        fe = slot;
                        if (do_ins && wslot < 0 && !evict) {
                            wslot = slot;
                            kind = 0;
                        }
                        walking = false;
                        continue;
                    }
                    if (do_ins && wslot < 0 && !evict) {
                        if ((e[k].t & better) != better) {
                            wslot = slot;
                            kind = 1;
                        } else if (mrz_ones_rank(e[k].t) < my_rank) {
                            wslot = slot;
                            kind = 2;
                            occ_t = e[k].t;
                            occ_off = e[k].off;
                        } else if (e[k].t == t) {
                            if (++round == max_chain) {
                                evict = true;
                                kind = 3;
                            }
                        }
                    }
                    if (e[k].t == t) {
                        if (nsame < MRZ_SMAX) {
                            if (!mrz_pool_put(S, gl, nsame, e[k].off, slot)) MRZ_CPLX(3);
                        } else
                            MRZ_CPLX(4);
                        nsame++;
                    }
                }
                s += MRZ_WALK_SLOTS;
                if (walking && ++steps >= MRZ_WALK_STEPS) {
                    MRZ_CPLX(1);
                    walking = false;
                }
            }
        }
        // stragglers: the rest of a long chain, the whole wave on one lane's chain (64 slots per step)
        for (mrz_u64 todo = __ballot(walking); todo; todo &= todo - 1) {
            const int ol = __ffsll((long long)todo) - 1;
            const int gl_o = gl0 + ol;
            const int64_t t_o = mrz_bcast64(t, ol);
            const int rank_o = mrz_lane_read(my_rank, ol);
            const bool ins_o = mrz_lane_read((int)do_ins, ol) != 0;
            int s_o = mrz_lane_read(s, ol), fe_o = -1;
            int wslot_o = mrz_lane_read(wslot, ol), kind_o = mrz_lane_read(kind, ol), round_o = mrz_lane_read(round, ol);
            int nsame_o = mrz_lane_read(nsame, ol);
            bool evict_o = mrz_lane_read((int)evict, ol) != 0, cplx_o = mrz_lane_read((int)cplx, ol) != 0;
            int why_o = mrz_lane_read(why, ol);
            int64_t occ_t_o = mrz_bcast64(occ_t, ol), occ_off_o = mrz_bcast64(occ_off, ol);
            mrz_ov ov_o;
            if (USE_OV) {  // the straggler's overlay, broadcast to its wave
                ov_o.n = mrz_lane_read(ovl.n, ol);
#pragma unroll
                for (int k = 0; k < MRZ_OV_MAX; k++) {
                    ov_o.slot[k] = mrz_lane_read(ovl.slot[k], ol);
                    ov_o.off[k] = mrz_bcast64(ovl.off[k], ol);
                    ov_o.t[k] = mrz_bcast64(ovl.t[k], ol);
                }
            }
            for (int cstep = 0; fe_o < 0; cstep++) {
                if (cstep >= MRZ_WALK_COOP_STEPS) {
                    cplx_o = true;
                    if (!why_o) why_o = 1;
                    break;
                }
                const int slot = (s_o + lane) & smask;
                const mrz_slot e = mrz_tab_load(tab, slot, USE_OV ? &ov_o : nullptr);
                const bool empty = (e.off | e.t) == 0;
                const mrz_u64 m_empty = __ballot(empty);
                const int fe_idx = m_empty ? __ffsll((long long)m_empty) - 1 : 64;
                const mrz_u64 valid = mrz_low_mask(fe_idx);
                const mrz_u64 m_same = __ballot(!empty && e.t == t_o) & valid;
                if (ins_o && wslot_o < 0 && !evict_o) {
                    const mrz_u64 m_worse = __ballot(!empty && (e.t & better) != better) & valid;
                    const mrz_u64 m_lower = __ballot(!empty && mrz_ones_rank(e.t) < rank_o) & valid & ~m_worse;
                    const mrz_u64 m_stop = m_worse | m_lower;
                    const int ks = m_stop ? __ffsll((long long)m_stop) - 1 : 64;
                    const int nq = __popcll(m_same & ~m_worse & mrz_low_mask(ks));
                    if (round_o + nq >= max_chain) {
                        evict_o = true;
                        kind_o = 3;
                        round_o = max_chain;
                    } else {
                        round_o += nq;
                        if (ks < 64) {
                            wslot_o = (s_o + ks) & smask;
                            kind_o = ((m_worse >> ks) & 1) ? 1 : 2;
                            if (kind_o == 2) {
                                occ_t_o = mrz_bcast64(e.t, ks);
                                occ_off_o = mrz_bcast64(e.off, ks);
                            }
                        } else if (fe_idx < 64) {
                            wslot_o = (s_o + fe_idx) & smask;
                            kind_o = 0;
                        }
                    }
                }
                // room for the new tag-equal entries: chunks are handed out by one lane, then everyone stores
                const int nnew = __popcll(m_same);
                if (nnew) {
                    int need = nsame_o + nnew;
                    if (need > MRZ_SMAX) need = MRZ_SMAX;
                    for (int c = (nsame_o + 3) >> 2; c < (need + 3) >> 2; c++)
                        if (lane == 0 && c >= S->nchunk[gl_o]) {
                            const int id = atomicAdd(&S->pool_top, 1);
                            S->chunk_id[gl_o][c] = id < MRZ_POOL ? (unsigned short)id : (unsigned short)0xffff;
                            if (id < MRZ_POOL) S->nchunk[gl_o] = (unsigned char)(c + 1);
                        }
                    MRZ_WAVE_SYNC();
                    bool bad = false;
                    if ((m_same >> lane) & 1) {
                        const int idx = nsame_o + __popcll(m_same & mrz_low_mask(lane));
                        if (idx < MRZ_SMAX) {
                            const int id = S->chunk_id[gl_o][idx >> 2];
                            if (id == 0xffff || (idx >> 2) >= S->nchunk[gl_o])  // (never handed out: the pool had run dry)
                                bad = true;
                            else {
#ifdef MRZ_EMU_LDS_PER_BLOCK
                                if (id >= MRZ_POOL) __builtin_trap();
#endif
                                S->pool[id].e[idx & 3] =
                                    (unsigned long long)e.off | ((unsigned long long)(unsigned)slot << MRZ_OFF_BITS);
                            }
                        }
                    }
                    if (__ballot(bad)) {
                        cplx_o = true;
                        if (!why_o) why_o = 3;
                    }
                    nsame_o += nnew;
                    if (nsame_o > MRZ_SMAX) {
                        cplx_o = true;
                        if (!why_o) why_o = 4;
                    }
                }
                if (fe_idx < 64) fe_o = (s_o + fe_idx) & smask;
                s_o += 64;
            }
            if (lane == ol) {
                walking = false;
                fe = fe_o;
                wslot = wslot_o;
                kind = kind_o;
                round = round_o;
                nsame = nsame_o;
                evict = evict_o;
                cplx = cplx_o;
                why = why_o;
                occ_t = occ_t_o;
                occ_off = occ_off_o;
            }
        }
        MRZ_WAVE_SYNC();
    }
    if (evict && max_chain > MRZ_SMAX) MRZ_CPLX(4);

    // ---- walk of a displaced occupant (src/rzip.c:275-278) ----
    int h2 = 0, w2 = -1, kind2 = -1, len2 = 0;
    {
        bool walking = go && !cplx && kind == 2;
        const int rank2 = mrz_ones_rank(occ_t);
        h2 = (int)(occ_t & C.slot_mask);
        int s = h2, steps = 0, round2 = 0;
        int ustep = 0;
        while (true) {
            const mrz_u64 m_walk = __ballot(walking);
            if (!m_walk) break;
            if (ustep >= MRZ_WALK_LANE_STEPS && __popcll(m_walk) <= MRZ_WALK_COOP_MAX) break;
            ustep++;
            if (walking) {
                mrz_slot e[MRZ_WALK_SLOTS];
#pragma unroll
                for (int k = 0; k < MRZ_WALK_SLOTS; k++) e[k] = mrz_tab_load(tab, (s + k) & smask, ov);
#pragma unroll
                for (int k = 0; k < MRZ_WALK_SLOTS; k++) {
                    if (!walking) continue;
                    const int slot = (s + k) & smask;
                    if ((e[k].off | e[k].t) == 0) {
                        w2 = slot;
                        kind2 = 0;
                        walking = false;
                    } else if ((e[k].t & better) != better) {
                        w2 = slot;
                        kind2 = 1;
                        walking = false;
                    } else if (mrz_ones_rank(e[k].t) < rank2) {
                        MRZ_CPLX(2);  // second-level displacement: cooperative path
                        walking = false;
                    } else if (e[k].t == occ_t) {
                        if (++round2 == max_chain) {
                            MRZ_CPLX(2);
                            walking = false;
                        }
                    }
                }
                s += MRZ_WALK_SLOTS;
                if (walking && ++steps >= MRZ_WALK_STEPS) {
                    MRZ_CPLX(1);
                    walking = false;
                }
            }
        }
        for (mrz_u64 todo = __ballot(walking); todo; todo &= todo - 1) {
            const int ol = __ffsll((long long)todo) - 1;
            const int64_t ot = mrz_bcast64(occ_t, ol);
            const int rk = mrz_lane_read(rank2, ol);
            int s_o = mrz_lane_read(s, ol), r2 = mrz_lane_read(round2, ol);
            int w2_o = -1, kind2_o = -1;
            bool cplx_o = false, done = false;
            mrz_ov ov_o;
            if (USE_OV) {
                ov_o.n = mrz_lane_read(ovl.n, ol);
#pragma unroll
                for (int k = 0; k < MRZ_OV_MAX; k++) {
                    ov_o.slot[k] = mrz_lane_read(ovl.slot[k], ol);
                    ov_o.off[k] = mrz_bcast64(ovl.off[k], ol);
                    ov_o.t[k] = mrz_bcast64(ovl.t[k], ol);
                }
            }
            for (int cstep = 0; !done; cstep++) {
                if (cstep >= MRZ_WALK_COOP_STEPS) {
                    cplx_o = true;
                    break;
                }
                const int slot = (s_o + lane) & smask;
                const mrz_slot e = mrz_tab_load(tab, slot, USE_OV ? &ov_o : nullptr);
                const bool empty = (e.off | e.t) == 0;
                const bool worse = !empty && (e.t & better) != better;
                const bool lower = !empty && !worse && mrz_ones_rank(e.t) < rk;
                const mrz_u64 m_empty = __ballot(empty), m_worse = __ballot(worse), m_lower = __ballot(lower);
                const mrz_u64 m_stop = m_empty | m_worse | m_lower;
                const int ks = m_stop ? __ffsll((long long)m_stop) - 1 : 64;
                const int nq = __popcll(__ballot(!empty && !worse && !lower && e.t == ot) & mrz_low_mask(ks));
                if (r2 + nq >= max_chain) {
                    cplx_o = true;
                    done = true;
                } else if (ks < 64) {
                    if ((m_lower >> ks) & 1)
                        cplx_o = true;  // second-level displacement: cooperative path
                    else {
                        w2_o = (s_o + ks) & smask;
                        kind2_o = ((m_empty >> ks) & 1) ? 0 : 1;
                    }
                    done = true;
                } else {
                    r2 += nq;
                    s_o += 64;
                }
            }
            if (lane == ol) {
                walking = false;
                w2 = w2_o;
                kind2 = kind2_o;
                if (cplx_o) MRZ_CPLX(2);
            }
        }
        if (w2 >= 0) len2 = ((w2 - h2) & smask) + 1;
    }
    o.fe = fe;
    o.wslot = wslot;
    o.kind = kind;
    o.nsame = nsame;
    o.h2 = h2;
    o.w2 = w2;
    o.kind2 = kind2;
    o.len2 = len2;
    o.cplx = cplx;
    o.why = why;
    o.occ_t = occ_t;
    o.occ_off = occ_off;
}

// cyclic interval test: does slot x lie in [a, a + la) (mod table size)?
#undef MRZ_CPLX
__device__ __forceinline__ bool mrz_in_range(int x, int a, int la, int smask) { return ((x - a) & smask) < la; }
// do the cyclic intervals [a, a + la) and [b, b + lb) share a slot?
__device__ __forceinline__ bool mrz_ranges_meet(int a, int la, int b, int lb, int smask) {
    return la > 0 && lb > 0 && ((((b - a) & smask) < la) || (((a - b) & smask) < lb));
}

// -DMRZ_DBG_HITS (tools/dbg_runs.py): every counted look-up leaves its hits / misses, and every preparation and
// pre-commit a signature of what it found, in a buffer indexed by position -- two runs that should be identical can
// be compared position by position (this is how the dry-pool bug of mrz_pool_put was found)
#ifdef MRZ_DBG_HITS
__device__ unsigned *mrz_dbg_hits;  // per position: (hits << 16 | misses) added by whoever counts them, + 1<<31 per visit
#define DBG_HITS(q, h, m) atomicAdd(&mrz_dbg_hits[(q)], ((unsigned)(h) << 16) | (unsigned)(m) | (1u << 28))
__device__ long long mrz_dbg_n;  // positions: plane k of the buffer starts at k * mrz_dbg_n
#define DBG_PLANE(k, q, v) atomicAdd(&mrz_dbg_hits[(long long)(k) * mrz_dbg_n + (q)], (unsigned)(v))
#else
#define DBG_HITS(q, h, m) ((void)0)
#define DBG_PLANE(k, q, v) ((void)0)
#endif
// all of this wave's global stores have completed (acknowledged by the L2)
#ifdef __HIP_DEVICE_COMPILE__
#define MRZ_STORES_DONE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define MRZ_STORES_DONE() ((void)0)
#endif
// ---- block hash of this batch's writes (phase B2) ----
__device__ __forceinline__ unsigned mrz_bh_hash(unsigned block) { return (block * 2654435761u) >> (32 - 11); }

__device__ static void mrz_bh_add(mrz_wide_lds *S, int slot, int gl) {
    const unsigned key = ((unsigned)slot >> 6) + 1u;
    unsigned i = mrz_bh_hash(key);
    for (int n = 0; n < MRZ_BH_SIZE; n++) {
        const unsigned k = atomicCAS(&S->bh_key[i], 0u, key);
        if (k == 0u || k == key) {
            const unsigned c = atomicAdd(&S->bh_cnt[i], 1u);
            if (c < MRZ_BH_WRITERS)
                S->bh_lane[i][c] = (unsigned short)gl;
            else {
                // (many lanes of one tag appending to one chain) the rest of a crowded block's writers: one list
                const int at = atomicAdd(&S->bho_n, 1);
                if (at < MRZ_BHO_MAX) {
                    S->bho_key[at] = key;
                    S->bho_lane[at] = (unsigned short)gl;
                }
            }
            return;
        }
        i = (i + 1) & (MRZ_BH_SIZE - 1);
    }
}

// the writes lane `wl` would make, as far as they are known before the scans: its insert slot (an evicting lane
// overwrites one of the tag-equal entries of its own chain: any slot of [h, h + len1)), and the slot its displaced
// occupant moves to.  Does one of them lie in [a, a + la) or [b, b + lb)?
__device__ __forceinline__ bool mrz_writes_hit(const mrz_wide_lds *S, int wl, int a, int la, int b, int lb, int smask) {
    const int k = S->kind[wl];
    bool hit;
    if (k == 3)
        hit = mrz_ranges_meet(S->h[wl], S->len1[wl], a, la, smask) || mrz_ranges_meet(S->h[wl], S->len1[wl], b, lb, smask);
    else {
        const int ws = S->wslot[wl];
        hit = (la > 0 && mrz_in_range(ws, a, la, smask)) || (lb > 0 && mrz_in_range(ws, b, lb, smask));
        if (k == 2) {
            const int w2 = S->w2[wl];
            hit = hit || (la > 0 && mrz_in_range(w2, a, la, smask)) || (lb > 0 && mrz_in_range(w2, b, lb, smask));
        }
    }
    return hit;
}

// Reader side: does an earlier lane of this batch write into [a, a + la) or [b, b + lb)?  Returns the number of
// such lanes found (capped), their ids in wr[0..1]; -1 when a block has more writers than the hash records.
__device__ static int mrz_bh_readers(const mrz_wide_lds *S, int gl, int a, int la, int b, int lb, int smask, int *wr) {
    int found = 0;
    for (int part = 0; part < 2; part++) {
        const int r0 = part ? b : a, rl = part ? lb : la;
        if (rl <= 0) continue;
        const int last = ((r0 + rl - 1) & smask) >> 6;
        for (int blk = r0 >> 6;; blk = (blk + 1) & (smask >> 6)) {
            const unsigned key = (unsigned)blk + 1u;
            unsigned i = mrz_bh_hash(key);
            for (int n = 0; n < MRZ_BH_SIZE; n++) {
                const unsigned k = S->bh_key[i];
                if (k == 0u) break;
                if (k == key) {
                    unsigned c = S->bh_cnt[i];
                    if (c > MRZ_BH_WRITERS) {
                        // more writers than the entry records: the rest are in the overflow list
                        const int no = S->bho_n;
                        if (no > MRZ_BHO_MAX) return -1;
                        for (int z = 0; z < no; z++) {
                            if (S->bho_key[z] != key) continue;
                            const int wl = S->bho_lane[z];
                            if (wl >= gl || !mrz_writes_hit(S, wl, a, la, b, lb, smask)) continue;
                            bool dup = false;
                            for (int y = 0; y < found && y < MRZ_WR_MAX; y++) dup = dup || wr[y] == wl;
                            if (!dup) {
                                if (found < MRZ_WR_MAX) wr[found] = wl;
                                found++;
                            }
                        }
                        c = MRZ_BH_WRITERS;
                    }
                    for (unsigned j = 0; j < c; j++) {
                        const int wl = S->bh_lane[i][j];
                        if (wl >= gl) continue;
                        if (!mrz_writes_hit(S, wl, a, la, b, lb, smask)) continue;
                        bool dup = false;
                        for (int z = 0; z < found && z < MRZ_WR_MAX; z++) dup = dup || wr[z] == wl;
                        if (!dup) {
                            if (found < MRZ_WR_MAX) wr[found] = wl;
                            found++;
                        }
                    }
                    break;
                }
                i = (i + 1) & (MRZ_BH_SIZE - 1);
            }
            if (blk == last) break;
        }
    }
    return found;
}

// best of a lane's tag-equal entries under the current last_match, in probe order: first longest wins
// (find_best_match, src/rzip.c:443-454)
__device__ static void mrz_lane_best(const mrz_wide_lds *S, int gl, int nsame, int64_t q, int64_t floor_p, int64_t *blen,
                                     int64_t *boff, int *brev, int *hits, int *misses, bool *need_long) {
    int64_t best = 0, best_off = 0;
    int best_rev = 0, h = 0, m = 0;
    bool lng = false;
    for (int k = 0; k < nsame; k++) {
        const int c = S->chunk_id[gl][k >> 2];
        const unsigned long long v = S->pool[c].e[k & 3];
        const unsigned raw = S->pool[c].raw[k & 3];
        const int64_t op = (int64_t)(v & MRZ_OFF_MASK);
        int64_t ml;
        int rv;
        bool l;
        mrz_pair_eval(raw, q, op, floor_p, &ml, &rv, &l);
        lng = lng || l;
        if (ml) {
            if (ml > best) {
                best = ml;
                best_off = op - rv;
                best_rev = rv;
            }
            h++;
        } else
            m++;
    }
    *blen = lng ? 0 : best;
    *boff = best_off;
    *brev = best_rev;
    *hits = h;
    *misses = m;
    *need_long = lng;
}

// slot of the failing entry of rank `r` in the cull window
__device__ __forceinline__ int mrz_cw_slot_search(const mrz_wide_lds *S, int64_t cw_base, int r) {
    int lo = 0, hi = MRZ_CW_WORDS - 1;
#pragma unroll
    for (int it = 0; it < 5; it++) {
        const int mid = (lo + hi + 1) >> 1;
        if (S->cwcum[mid] <= r)
            lo = mid;
        else
            hi = mid - 1;
    }
    return (int)(cw_base + lo * 64 + mrz_select64(S->cw[lo], r - S->cwcum[lo]));
}

// the same from the list the preparation has laid out (ranks beyond it: the search)
__device__ __forceinline__ int mrz_cw_slot(const mrz_wide_lds *S, int64_t cw_base, int r) {
    return r < MRZ_W ? S->cw_list[r] : mrz_cw_slot_search(S, cw_base, r);
}

struct mrz_wide_ret {
    int used;          // candidates committed by this step
    bool ok;           // false: event list overflow
    bool whole;        // every lane of the batch has been dealt with (committed, dropped or handed over)
    bool stop_batch;   // the masks have moved / the cull window no longer holds: batches prepared ahead are void
    bool rebulk;       // a long clean stretch lies ahead (from S->first_live, cull rank S->rank0): bulk-commit it, come back
};

// lane flags published for the committing wave
#define MRZ_LF_CONF 1
#define MRZ_LF_CPLX 2
#define MRZ_LF_ACT 4
#define MRZ_LF_INS 8
#define MRZ_LF_LONG 16
#define MRZ_LF_REVS 32   // some entry has equal bytes before it: its result depends on how close last_match is

#ifndef MRZ_RAW_PER_THREAD
#define MRZ_RAW_PER_THREAD 4  // list entries a thread examines in the formation: a batch window holds at most 4 x threads
#endif

// PREPARATION of one wide batch (phases A-C): all threads of the workgroup.  The batch is a WINDOW OF THE CANDIDATE
// LIST: entries [i0, i0 + pw) of the segment's list (pw <= MRZ_RAW_PER_THREAD x threads), of which those that still
// pass min_mask -- at most MRZ_W of them -- get one lane each.  Nothing here depends on the matcher's moving state
// except the two masks (min_mask decides who is a candidate, tag_mask who inserts) and the table itself -- which may
// be BEHIND: when several sequencer workgroups take turns, a batch is prepared while earlier batches are still being
// committed, and the pre-commit step checks every lane against the log of blocks written since (mrz_wide_precommit).
// Leaves everything the commit needs in LDS; S->nb == 0 means the window held no candidate.
template <int NW>
__device__ static void mrz_wide_prep(const mrz_cfg &C, mrz_wide_lds *S, const mrz_cands &K, int64_t lim, int64_t i0, int pw,
                                     int64_t min_pos, int64_t min_mask, int64_t tag_mask, int tid, int lane, int wave,
                                     int64_t *stat) {
    constexpr int WT = 64 * NW;  // threads (= lanes of the batch at most) taking part
    const uint8_t *__restrict__ buf = C.buf;
    const int smask = (int)C.slot_mask;
    const int64_t better = (min_mask << 1) | 1;
    PROF_T0();

    // ---- A: formation ----------------------------------------------------------------------------------
    int nraw = (K.n - i0) < (int64_t)pw ? (int)(K.n - i0) : pw;
    if (nraw > MRZ_RAW_PER_THREAD * WT) nraw = MRZ_RAW_PER_THREAD * WT;
    int64_t qr[MRZ_RAW_PER_THREAD], tr[MRZ_RAW_PER_THREAD];
    bool ar[MRZ_RAW_PER_THREAD];
    int cnt_a = 0;
#pragma unroll
    for (int e = 0; e < MRZ_RAW_PER_THREAD; e++) {
        const int r = MRZ_RAW_PER_THREAD * tid + e;
        qr[e] = 0;
        tr[e] = 0;
        ar[e] = false;
        if (r < nraw) {
            const mrz_cand c = K.cand[i0 + r];
            qr[e] = c.off;
            tr[e] = c.t;
            ar[e] = c.off >= min_pos && c.off <= lim && (c.t & min_mask) == min_mask;
        }
        cnt_a += ar[e] ? 1 : 0;
    }
    // where the window ends: just before the first entry of the next one
    int64_t wend = lim;
    if (tid == 0 && i0 + nraw < K.n) {
        const int64_t nx = K.cand[i0 + nraw].off - 1;
        wend = nx < lim ? nx : lim;
    }
    for (int i = tid; i < MRZ_BH_SIZE; i += WT) {
        S->bh_key[i] = 0u;
        S->bh_cnt[i] = 0u;
    }
    if (tid == 0) {
        S->pool_top = 0;
        S->bho_n = 0;
        S->nw_cnt = 0;
        S->xw_n = 0;
    }
    S->exec[tid] = 0;
    S->nchunk[tid] = 0;
#ifdef MRZ_EMU_LDS_PER_BLOCK
    // (test emulator: a chunk id left over from an earlier batch must never be used -- make such a use fault)
    for (int k = 0; k < 4; k++) S->chunk_id[tid][k] = (unsigned short)0xffff;
#endif
    S->supp_w[tid] = (unsigned short)MRZ_W;
    S->supp_w2[tid] = (unsigned short)MRZ_W;
    const int64_t cw_base = 0;  // the cull window is the pre-commit step's business
    const int cw_len = 0;
    int total;
    const int incl_a = mrz_wide_incl<NW>(cnt_a, S->wt3, lane, wave, &total);
    {
        int at = incl_a - cnt_a;
#pragma unroll
        for (int e = 0; e < MRZ_RAW_PER_THREAD; e++)
            if (ar[e]) {
                if (at < MRZ_W) {
                    S->q[at] = qr[e];
                    S->t[at] = tr[e];
                }
                at++;
            }
    }
    const int nb = total < MRZ_W ? total : MRZ_W;
    mrz_prep_sync<NW>();
    if (tid == 0) {
        S->nb = nb;
        S->total = total;  // > nb: more candidates than a batch has lanes
        S->w_end = wend;
        // how far the matcher may move once every lane has been committed: nowhere beyond the last lane when candidates
        // were left out, else to the window's end
        S->adv_to = total > MRZ_W ? (int64_t)-1 : wend;
        S->prep_min_mask = min_mask;
        S->prep_tag_mask = tag_mask;
    }
    const bool have = tid < nb;
    const int64_t q = have ? S->q[tid] : 0, t = have ? S->t[tid] : 0;
    mrz_prep_sync<NW>();
    if (nb == 0) return;
    const bool act = have && (t & min_mask) == min_mask;
    bool ins = act && (t & tag_mask) == tag_mask;
    const bool loose = tag_mask != better;
    S->q[tid] = have ? q : (int64_t)0x7fffffffffffffffll;
    PROF_ADD(MRZ_ST_T_FORM);

    // ---- B: walk -----------------------------------------------------------------------------------------
    mrz_wl wl;
    mrz_ov ov0;
    ov0.n = 0;
    mrz_wide_walk<false>(C, S, act, tid, lane, t, ins, better, ov0, wl);
    bool cplx = act && wl.cplx;
    bool conf = false;
    ST_COUNT(MRZ_ST_X_WALK, cplx && wl.why == 1);
    ST_COUNT(MRZ_ST_X_CASC, cplx && wl.why == 2);
    ST_COUNT(MRZ_ST_X_POOL, cplx && wl.why == 3);
    ST_COUNT(MRZ_ST_X_SAME, cplx && wl.why == 4);
    int h = (int)(t & C.slot_mask);
    int len1 = (act && !cplx) ? ((wl.fe - h) & smask) + 1 : 0;
    int dep[MRZ_WR_MAX];  // lanes whose writes this lane's (overlay) walk has assumed
#pragma unroll
    for (int k = 0; k < MRZ_WR_MAX; k++) dep[k] = -1;

    // post-walk rules that involve the cull window; publishes the lane's facts
    auto post_walk = [&]() {
        if (act && !cplx) {
            // a write that changes the set of failing entries inside the window would change the sweep
            if (ins && cw_len > 0) {
                if ((wl.kind == 1 || loose) && wl.kind != 3 && wl.wslot >= cw_base && wl.wslot < cw_base + cw_len) cplx = true;
                if (wl.kind == 2 && (wl.kind2 == 1 || loose) && wl.w2 >= cw_base && wl.w2 < cw_base + cw_len) cplx = true;
            }
            // reads inside the window: an earlier lane's cull may empty a slot this lane has read
            if (cw_len > 0 && (mrz_ranges_meet(h, len1, (int)cw_base, cw_len, smask) ||
                               mrz_ranges_meet(wl.h2, wl.len2, (int)cw_base, cw_len, smask)))
                conf = true;
        }
        S->h[tid] = h;
        S->len1[tid] = (act && !cplx) ? len1 : 0;
        S->h2[tid] = wl.h2;
        S->len2[tid] = (act && !cplx) ? wl.len2 : 0;
        S->wslot[tid] = wl.wslot;
        S->w2[tid] = wl.w2;
        S->kind[tid] = (unsigned char)((act && !cplx && ins) ? wl.kind : 255);
        S->kind2[tid] = (unsigned char)wl.kind2;
        S->ns[tid] = (unsigned char)(wl.nsame < MRZ_SMAX ? wl.nsame : MRZ_SMAX);
        S->occ_off[tid] = wl.occ_off;
        S->occ_t[tid] = wl.occ_t;
    };
    S->t[tid] = t;
    {
        const bool c0 = cplx;
        post_walk();
        (void)c0;
        ST_COUNT(MRZ_ST_X_WIN, cplx && !c0);
    }
    // B2: the block hash of the writes, then every lane looks for earlier writers in its read ranges
    if (act && !cplx && ins) {
        if (wl.kind == 3) {
            const int last = ((h + len1 - 1) & smask) >> 6;
            for (int blk = h >> 6;; blk = (blk + 1) & (smask >> 6)) {
                mrz_bh_add(S, blk << 6, tid);
                if (blk == last) break;
            }
        } else {
            mrz_bh_add(S, wl.wslot, tid);
            if (wl.kind == 2) mrz_bh_add(S, wl.w2, tid);
        }
    }
    PROF_ADD(MRZ_ST_T_WALK);
    mrz_prep_sync<NW>();
    int wr[MRZ_WR_MAX];
#pragma unroll
    for (int k = 0; k < MRZ_WR_MAX; k++) wr[k] = -1;
    int nwr = 0;
    if (act && !cplx) {
        nwr = mrz_bh_readers(S, tid, h, len1, wl.h2, wl.len2, smask, wr);
        if (nwr != 0) conf = true;
    }
    PROF_ADD(MRZ_ST_T_CONF);

    // ---- B3: overlay walk.  A stale lane whose (one or two) earlier writers are themselves sound walks once more
    // with those lanes' writes laid over the table -- what it would read had they been committed.  The common case
    // is the next position carrying the same tag (the XOR tag only sees the multiset of the 31 bytes).
    int cwhy = 0;  // why a stale lane could not be repaired here (0 = not tried yet; diagnostics, and who tries again)
    // Rounds: a lane whose writer was itself stale can be repaired once that writer has been (its repaired writes are
    // published and listed in nw_*), so the phase is run up to MRZ_OV_ROUNDS times, until no lane is eligible.
    for (int round = 0; round < MRZ_OV_ROUNDS; round++) {
        if (round > 0 && act && !cplx && conf && (cwhy == 3 || cwhy == 4 || cwhy == 7)) {
            // who writes into what this lane has read, as things stand now: the block hash (whose lanes' published
            // writes are up to date) and the repaired lanes' new writes
            for (int k = 0; k < MRZ_WR_MAX; k++) wr[k] = -1;
            nwr = mrz_bh_readers(S, tid, h, len1, wl.h2, wl.len2, smask, wr);
            int nw0 = S->nw_cnt;
            if (nw0 > MRZ_NW_MAX) nw0 = MRZ_NW_MAX;
            for (int k = 0; k < nw0 && nwr >= 0; k++) {
                const int wlane = S->nw_lane[k];
                if (wlane >= tid) continue;
                const int sl = S->nw_slot[k];
                if (!((len1 > 0 && mrz_in_range(sl, h, len1, smask)) || (wl.len2 > 0 && mrz_in_range(sl, wl.h2, wl.len2, smask))))
                    continue;
                bool dup = false;
                for (int z = 0; z < nwr && z < MRZ_WR_MAX; z++) dup = dup || wr[z] == wlane;
                if (!dup) {
                    if (nwr < MRZ_WR_MAX) wr[nwr] = wlane;
                    nwr++;
                }
            }
            cwhy = 0;
        }
        const bool win_stale = act && !cplx && conf && nwr == 0;  // stale only because of the cull window
        S->lf[tid] = (unsigned char)((conf ? 1 : 0) | (cplx ? 2 : 0) | (act ? 4 : 0));
        mrz_prep_sync<NW>();
        bool elig = act && !cplx && conf && !win_stale && nwr >= 1 && nwr <= MRZ_WR_MAX && cwhy == 0;
        if (act && !cplx && conf && !elig && cwhy == 0) cwhy = win_stale ? 1 : 4;
        mrz_ov ov;
        ov.n = 0;
        int ov_src[MRZ_OV_MAX];  // writer lane * 2 + (0: its insert slot, 1: its occupant's new slot)
#pragma unroll
        for (int k = 0; k < MRZ_OV_MAX; k++) {
            ov_src[k] = -1;
            ov.slot[k] = -1;
            ov.off[k] = 0;
            ov.t[k] = 0;
        }
        if (elig) {
            // in lane order: a later writer's store is the one that stays
#pragma unroll
            for (int x = 1; x < MRZ_WR_MAX; x++)
#pragma unroll
                for (int y2 = MRZ_WR_MAX - 1; y2 >= 1; y2--)
                    if (y2 < nwr && wr[y2 - 1] > wr[y2]) {
                        const int z__ = wr[y2 - 1];
                        wr[y2 - 1] = wr[y2];
                        wr[y2] = z__;
                    }
#pragma unroll
            for (int k = 0; k < MRZ_WR_MAX; k++) {
                if (k >= nwr) continue;
                const int i = wr[k];
                const int ki = S->kind[i];
                if ((S->lf[i] & 3) != 0 || ki == 3 || ki == 255) {
                    elig = false;
                    if (!cwhy) cwhy = ki == 3 ? 2 : 3;
                }
            }
            // the window read rule would make it stale again anyway
            if (cw_len > 0 && (mrz_ranges_meet(h, len1, (int)cw_base, cw_len, smask) ||
                               mrz_ranges_meet(wl.h2, wl.len2, (int)cw_base, cw_len, smask))) {
                elig = false;
                if (!cwhy) cwhy = 1;
            }
        }
        // entries 2k (the occupant writer k displaces) and 2k + 1 (its insert), statically indexed so that the
        // overlay stays in registers; unused entries keep slot -1 and never match
#pragma unroll
        for (int k = 0; k < MRZ_WR_MAX; k++)
            if (elig && k < nwr) {
                const int i = wr[k];
                if (S->kind[i] == 2) {
                    ov.slot[2 * k] = S->w2[i];
                    ov.off[2 * k] = S->occ_off[i];
                    ov.t[2 * k] = S->occ_t[i];
                    ov_src[2 * k] = i * 2 + 1;
                }
                ov.slot[2 * k + 1] = S->wslot[i];
                ov.off[2 * k + 1] = S->q[i];
                ov.t[2 * k + 1] = S->t[i];
                ov_src[2 * k + 1] = i * 2;
            }
        ov.n = MRZ_OV_MAX;
        const mrz_u64 m_el = __ballot(elig);
        int n_el;
        (void)mrz_wide_incl<NW>(lane == 0 ? __popcll(m_el) : 0, S->wt5, lane, wave, &n_el);
        if (!n_el) break;  // uniform
        {
            ST_ADD(MRZ_ST_OVL, n_el);
            mrz_wl wn;
            mrz_wide_walk<true>(C, S, elig, tid, lane, t, ins, better, ov, wn);
            bool good = false;
            if (elig) {
                good = !wn.cplx && wn.kind != 3;
                // A lane that overwrites (or displaces) an entry one of its writers has just put there: inside one
                // segment both would store to the same slot.  The later store is the one that stays, so the writer is
                // told to leave its store out when both commit together (supp_w / supp_w2, looked at in the commit).
                // Only the insert slot may tie; an occupant moving onto a writer's slot waits for its turn.
                int tie_src = -1;
                if (good && ins)
                    for (int k = 0; k < MRZ_OV_MAX; k++)
                        if (k < ov.n) {
                            if (wn.kind == 2 && ov.slot[k] == wn.w2) {
                                good = false;
                                cwhy = 6;
                            } else if (ov.slot[k] == wn.wslot)
                                tie_src = ov_src[k];  // the last writer of that slot (overlay order = lane order)
                        }
                const int nlen1 = good ? ((wn.fe - h) & smask) + 1 : 0;
                if (good) {
                    // no other earlier writer may reach into what it has read now
                    int wr2[MRZ_WR_MAX];
                    for (int k = 0; k < MRZ_WR_MAX; k++) wr2[k] = -1;
                    const int n2 = mrz_bh_readers(S, tid, h, nlen1, wn.h2, wn.len2, smask, wr2);
                    if (n2 < 0 || n2 > nwr) good = false;
                    for (int k = 0; k < n2 && k < MRZ_WR_MAX && good; k++) {
                        bool known = false;
                        for (int z = 0; z < nwr; z++) known = known || wr2[k] == wr[z];
                        if (!known) good = false;
                    }
                    if (cw_len > 0 && (mrz_ranges_meet(h, nlen1, (int)cw_base, cw_len, smask) ||
                                       mrz_ranges_meet(wn.h2, wn.len2, (int)cw_base, cw_len, smask)))
                        good = false;
                    if (ins && cw_len > 0) {
                        if ((wn.kind == 1 || loose) && wn.wslot >= cw_base && wn.wslot < cw_base + cw_len) good = false;
                        if (wn.kind == 2 && (wn.kind2 == 1 || loose) && wn.w2 >= cw_base && wn.w2 < cw_base + cw_len)
                            good = false;
                    }
                }
                // its new writes: the lanes behind it have to look at them again
                if (good && ins) {
                    const int nn = wn.kind == 2 ? 2 : 1;
                    const int at = atomicAdd(&S->nw_cnt, nn);
                    if (at + nn > MRZ_NW_MAX)
                        good = false;
                    else {
                        S->nw_slot[at] = wn.wslot;
                        S->nw_lane[at] = (unsigned short)tid;
                        if (nn == 2) {
                            S->nw_slot[at + 1] = wn.w2;
                            S->nw_lane[at + 1] = (unsigned short)tid;
                        }
                    }
                }
                if (good && tie_src >= 0) {
                    if (tie_src & 1)
                        S->supp_w2[tie_src >> 1] = (unsigned short)tid;
                    else
                        S->supp_w[tie_src >> 1] = (unsigned short)tid;
                }
                if (good) {
                    wl = wn;
                    len1 = nlen1;
                    conf = false;
#pragma unroll
                    for (int k = 0; k < MRZ_WR_MAX; k++) dep[k] = k < nwr ? wr[k] : -1;
                    S->len1[tid] = len1;
                    S->h2[tid] = wl.h2;
                    S->len2[tid] = wl.len2;
                    S->wslot[tid] = wl.wslot;
                    S->w2[tid] = wl.w2;
                    S->kind[tid] = (unsigned char)(ins ? wl.kind : 255);
                    S->kind2[tid] = (unsigned char)wl.kind2;
                    S->ns[tid] = (unsigned char)(wl.nsame < MRZ_SMAX ? wl.nsame : MRZ_SMAX);
                    S->occ_off[tid] = wl.occ_off;
                    S->occ_t[tid] = wl.occ_t;
                }
                if (!good && !cwhy) cwhy = 5;
                // a lane whose overlay walk failed keeps the facts of its first walk published: they are what the
                // lanes behind it were checked against, and it walks again when its turn comes
            }
            mrz_prep_sync<NW>();
            // every lane whose walk stands looks at the new writes of the overlay lanes before it
            int nw = S->nw_cnt;
            if (nw > MRZ_NW_MAX) nw = MRZ_NW_MAX;
            if (act && !cplx && !conf) {
                for (int k = 0; k < nw; k++) {
                    const int wlane = S->nw_lane[k];
                    bool mine = false;
#pragma unroll
                    for (int z = 0; z < MRZ_WR_MAX; z++) mine = mine || wlane == dep[z];
                    if (wlane >= tid || mine) continue;
                    const int sl = S->nw_slot[k];
                    if ((len1 > 0 && mrz_in_range(sl, h, len1, smask)) || (wl.len2 > 0 && mrz_in_range(sl, wl.h2, wl.len2, smask))) {
                        conf = true;
                        cwhy = 7;
                    }
                }
            }
#ifdef MRZ_SEQ_STATS
            {
                const mrz_u64 mg = __ballot(elig && good && !conf);
                int tg;
                (void)mrz_wide_incl<NW>(lane == 0 ? __popcll(mg) : 0, S->wt3, lane, wave, &tg);
                ST_ADD(MRZ_ST_OVL_OK, tg);
            }
#endif
        }
    }
    ST_COUNT(MRZ_ST_C_WIN, conf && cwhy == 1);
    ST_COUNT(MRZ_ST_C_EVICT, conf && cwhy == 2);
    ST_COUNT(MRZ_ST_C_DEEP, conf && cwhy == 3);
    ST_COUNT(MRZ_ST_C_MANY, conf && cwhy == 4);
    ST_COUNT(MRZ_ST_C_FAIL, conf && cwhy == 5);
    ST_COUNT(MRZ_ST_C_TIE, conf && cwhy == 6);
    ST_COUNT(MRZ_ST_C_NW, conf && cwhy == 7);
    PROF_ADD(MRZ_ST_T_OVL);

    // ---- C: pairs (lanes whose walk stands) ----------------------------------------------------------------
    auto do_pairs = [&](bool need) {
        const int ns = need ? (wl.nsame < MRZ_SMAX ? wl.nsame : MRZ_SMAX) : 0;
        int P;
        const int pincl = mrz_wide_incl<NW>(ns, S->wt3, lane, wave, &P);
        const int pb = pincl - ns;
        S->pbase[tid] = pb;
        for (int k = 0; k < ns; k++) S->pair_owner[pb + k] = (unsigned short)tid;
        mrz_prep_sync<NW>();
        for (int i = tid; i < P; i += WT) {
            const int o = S->pair_owner[i];
            const int k = i - S->pbase[o];
            const int c = S->chunk_id[o][k >> 2];
            const int64_t op = (int64_t)(S->pool[c].e[k & 3] & MRZ_OFF_MASK);
            S->pool[c].raw[k & 3] = (unsigned short)mrz_lane_probe_raw(buf, S->q[o], op, C.end);
        }
        mrz_prep_sync<NW>();
    };
    do_pairs(act && !cplx && !conf);
    PROF_ADD(MRZ_ST_T_PAIRS);
    ST_ADD(MRZ_ST_BATCHES, 1);
    ST_ADD(MRZ_ST_FORMED, nb);
#ifdef MRZ_SEQ_STATS
    {
        const mrz_u64 mc = __ballot(conf);
        int tc;
        const int ic = mrz_wide_incl<NW>(lane == 0 ? __popcll(mc) : 0, S->wt5, lane, wave, &tc);
        (void)ic;
        ST_ADD(MRZ_ST_CONF0, tc);
    }
#endif

    // ---- what the pre-commit step needs of this lane (the rest is in the per-lane arrays already) --------------------
    {
        bool revs = false;
        if (act && !cplx && !conf) {
            const int nsx = wl.nsame < MRZ_SMAX ? wl.nsame : MRZ_SMAX;
            for (int k = 0; k < nsx; k++) revs = revs || ((S->pool[S->chunk_id[tid][k >> 2]].raw[k & 3] >> 8) & 127) != 0;
        }
#pragma unroll
        for (int k = 0; k < MRZ_WR_MAX; k++) S->dep[k][tid] = (unsigned short)(dep[k] >= 0 ? dep[k] : MRZ_W);
        S->len1[tid] = (act && !cplx) ? len1 : 0;
        S->len2[tid] = (act && !cplx) ? wl.len2 : 0;
        S->lf[tid] = (unsigned char)((conf ? MRZ_LF_CONF : 0) | (cplx ? MRZ_LF_CPLX : 0) | (act ? MRZ_LF_ACT : 0) |
                                     (ins ? MRZ_LF_INS : 0) | (revs ? MRZ_LF_REVS : 0));
#ifdef MRZ_DBG_HITS
        if (have) {
            unsigned hsh = (unsigned)(conf ? 1 : 0) | (cplx ? 2 : 0) | (ins ? 4 : 0) | (revs ? 8 : 0);
            hsh = hsh * 31u + (unsigned)wl.nsame;
            hsh = hsh * 31u + (unsigned)wl.wslot;
            hsh = hsh * 31u + (unsigned)wl.kind;
            hsh = hsh * 31u + (unsigned)len1;
            if (act && !cplx && !conf)
                for (int k = 0; k < (wl.nsame < MRZ_SMAX ? wl.nsame : MRZ_SMAX); k++) {
                    const int c = S->chunk_id[tid][k >> 2];
                    hsh = hsh * 31u + (unsigned)(S->pool[c].e[k & 3] & MRZ_OFF_MASK) * 7u + S->pool[c].raw[k & 3];
                }
            DBG_PLANE(1, q, hsh | 1u);
        }
#endif
    }
    mrz_prep_sync<NW>();
}

// BULK COMMIT (E0): the lanes from `s_from` up to the first one that is stale, complex or has a match to fold need none
// of the lazy-match logic (no match can be adopted or emitted among them when none is pending): all waves commit them
// at once -- scans for victim_round, hash_count and the cull ranks (from rank `rank0` of the cull window on) as in the
// commit proper -- and leave wave 0 the totals in S->bulk_*.  Match-free stretches (noise) are committed entirely here:
// by the pre-commit step for the head of the batch, and again whenever the commit finds a long clean stretch behind a
// hand-over (`late`: the staleness rules of the commit -- writers that did not commit as prepared, hand-over writes --
// are applied first).  All threads; ends in a barrier.
template <int NW>
__device__ static __attribute__((noinline)) void mrz_wide_bulk(const mrz_cfg &C, const mrz_lead &L, mrz_wide_lds *S, unsigned *__restrict__ wlog,
                                     unsigned batch_no, int s_from, int rank0, bool late, int tid, int lane, int wave) {
    const int min_lanes = late ? 64 : MRZ_BULK_MIN;  // (the commit asks when a whole window of 64 is clean)
    mrz_slot *tab = C.tab;
    const int smask = (int)C.slot_mask;
    const int max_chain = (int)C.max_chain;
    const int64_t better = (L.min_mask << 1) | 1;
    const bool loose = L.tag_mask != better;
    const int nb = S->nb;
    const int64_t cw_base = S->cw_base;
    const bool have = tid < nb && tid >= s_from;
    const int f = have ? S->lf[tid] : 0;
    const bool act = (f & MRZ_LF_ACT) != 0, ins = act && (f & MRZ_LF_INS);
    const bool cplx = (f & MRZ_LF_CPLX) != 0, lng = (f & MRZ_LF_LONG) != 0;
    bool conf = (f & MRZ_LF_CONF) != 0;
    const int64_t q = S->q[tid];
    const int kind = S->kind[tid], kind2 = S->kind2[tid];
    const int wslot0 = S->wslot[tid], w2 = S->w2[tid];
    if (late && act && !cplx && !conf) {
#pragma unroll
        for (int k = 0; k < MRZ_WR_MAX; k++) {
            const int dk = S->dep[k][tid];
            if (dk < s_from && S->exec[dk] != 1) conf = true;
        }
        const int nx = S->xw_n;
        const int h = S->h[tid], len1 = S->len1[tid], h2 = S->h2[tid], len2 = S->len2[tid];
        for (int k = 0; k < nx && !conf; k++) {
            const int sl = S->xw_slot[k];
            if ((len1 > 0 && mrz_in_range(sl, h, len1, smask)) || (len2 > 0 && mrz_in_range(sl, h2, len2, smask))) conf = true;
        }
    }
    const int bm = act ? (S->bhm[tid] & 255) : 0;
    const bool stopish = have && act && (cplx || conf || lng || S->blen[tid] > 0);
    {
        const int fs = mrz_wave_first(stopish, wave, nb);
        if (lane == 0) S->wmin[0][wave] = fs;
    }
    mrz_prep_sync<NW>();
    int x0 = S->wmin[0][0];
#pragma unroll
    for (int w = 1; w < NW; w++) x0 = S->wmin[0][w] < x0 ? S->wmin[0][w] : x0;
    if (L.cur_len >= MRZ_MIN_MATCH) x0 = 0;  // a pending match may be emitted at any lane
    const int s0 = s_from;
    if (x0 - s0 >= min_lanes) {
        const bool inb = tid >= s0 && tid < x0;
        const bool a_ins = inb && ins;
        const bool a_ev = a_ins && kind == 3;
        const int d = a_ins ? (kind == 0 ? 1 : (kind == 2 ? (kind2 == 0 ? 1 : 0) : 0)) : 0;
        int dummy;
        const int i1 = mrz_wide_incl<NW>(d | (a_ev ? 1 << 10 : 0) | (a_ins ? 1 << 20 : 0), S->wt1, lane, wave, &dummy);
        int wslot = wslot0;
        if (a_ev) {
            const int er = ((i1 >> 10) & 1023) - 1;
            const int vr = (int)(((unsigned)L.victim_round + (unsigned)er) % (unsigned)max_chain);
            wslot = (int)(mrz_pool_get(S, tid, vr) >> MRZ_OFF_BITS);
        }
        int64_t c_before = L.count + ((i1 & 1023) - d);
        if (c_before > C.limit) c_before = C.limit;
        const bool cull = a_ins && (c_before + d > C.limit);
        const int i2 = mrz_wide_incl<NW>((cull ? 1 : 0) | ((inb && act ? bm : 0) << 10), S->wt3, lane, wave, &dummy);
        int cslot = -1;
        bool overflow = false;
        if (cull) {
            const int cr = rank0 + (i2 & 1023) - 1;
            if (cr >= S->cwcum[MRZ_CW_WORDS])
                overflow = true;
            else
                cslot = mrz_cw_slot(S, cw_base, cr);
        }
        {
            const int fo_ = mrz_wave_first(overflow, wave, MRZ_W);
            const int fc_ = mrz_wave_first(cull, wave, MRZ_W);
            if (lane == 0) {
                S->wmin[1][wave] = fo_;
                S->wmin[2][wave] = fc_;
            }
        }
        mrz_prep_sync<NW>();
        int o_lane = S->wmin[1][0], c_lane = S->wmin[2][0];
#pragma unroll
        for (int w = 1; w < NW; w++) {
            o_lane = S->wmin[1][w] < o_lane ? S->wmin[1][w] : o_lane;
            c_lane = S->wmin[2][w] < c_lane ? S->wmin[2][w] : c_lane;
        }
        int y = x0;
        bool endb = false;
        if (o_lane < y) {
            y = o_lane;
            endb = true;
        }
        if (loose && c_lane + 1 <= y) {  // the first cull ever switches the insert mask (:583)
            y = c_lane + 1;
            endb = true;
        }
        if (tid >= s0 && tid < y && act) DBG_HITS(q, 0, bm);
        if (tid >= s0 && tid < y) {
            if (a_ins) {
                const unsigned stamp = batch_no + 1u;
                if (kind == 2 && S->supp_w2[tid] >= y) {
                    mrz_slot oc;
                    oc.off = S->occ_off[tid];
                    oc.t = S->occ_t[tid];
                    tab[w2] = oc;
                    wlog[w2 >> MRZ_WLOG_SHIFT] = stamp;
                }
                if (S->supp_w[tid] >= y) {
                    mrz_slot nw;
                    nw.off = q;
                    nw.t = S->t[tid];
                    tab[wslot] = nw;
                    wlog[wslot >> MRZ_WLOG_SHIFT] = stamp;
                }
                if (cslot >= 0) {
                    mrz_slot z;
                    z.off = 0;
                    z.t = 0;
                    tab[cslot] = z;
                    wlog[cslot >> MRZ_WLOG_SHIFT] = stamp;
                }
            }
            S->exec[tid] = 1;
        }
        if (y > s0 && tid == y - 1) {
            S->bulk_a1 = i1;
            S->bulk_a2 = i2;
        }
        if (tid == 0) {
            S->bulk_y = y > s0 ? y : 0;
            S->bulk_end = endb ? 1 : 0;
        }
    } else if (tid == 0)
        S->bulk_y = 0;
    // the table stores above must have landed before anybody reads the table again (wave 0's hand-overs, the next
    // preparation): a workgroup barrier alone does not wait for them
    MRZ_STORES_DONE();
    mrz_prep_sync<NW>();
}

// PRE-COMMIT of a prepared batch: all threads of the workgroup, once the matcher's state L is this workgroup's to move
// (its turn has come).  Brings the prepared lanes up to date with what has happened since they were prepared:
//   * lanes at or before L.p are dropped (an emitted match covers them);
//   * a lane whose read ranges touch a block of table slots written after `snap` (the number of batches that had been
//     committed when the preparation began; wlog[block] = 1 + the batch that wrote it last) is stale;
//   * the cull window -- the failing entries ahead of tag_clean_ptr (clean_one_from_hash, src/rzip.c:313-321) -- is
//     loaded now; lanes that read inside it are stale, lanes whose insert would change it are complex;
//   * every lane's best match under the current last_match;
// then the bulk commit (E0): the leading lanes up to the first one that is stale, complex or has a match to fold need
// none of the lazy-match logic (no match can be adopted or emitted among them when none is pending): all waves commit
// them at once -- scans for victim_round, hash_count and the cull ranks as in the commit proper -- and hand wave 0 the
// totals.  Match-free stretches (noise) are committed entirely here.
template <int NW>
__device__ static __attribute__((noinline)) void mrz_wide_precommit(const mrz_cfg &C, const mrz_lead &L, mrz_wide_lds *S, unsigned *__restrict__ wlog,
                                          unsigned snap, unsigned batch_no, int tid, int lane, int wave, int64_t *stat) {
    mrz_slot *tab = C.tab;
    const int smask = (int)C.slot_mask;
    const int64_t better = (L.min_mask << 1) | 1;
    const bool loose = L.tag_mask != better;
    const int nb = S->nb;
    if (nb == 0) {
        if (tid == 0) S->bulk_y = 0;
        mrz_prep_sync<NW>();
        return;
    }
    PROF_T0();
    // ---- this lane, from what its preparation has published
    const bool have = tid < nb;
    const int f0 = have ? S->lf[tid] : 0;
    const int64_t q = S->q[tid < MRZ_W ? tid : 0];
    const bool live = have && q > L.p;  // not covered by a match emitted since
    const bool act = live && (f0 & MRZ_LF_ACT);
    const bool ins = act && (f0 & MRZ_LF_INS);
    bool cplx = act && (f0 & MRZ_LF_CPLX), conf = act && (f0 & MRZ_LF_CONF);
    const int h = S->h[tid], len1 = S->len1[tid], h2 = S->h2[tid], len2 = S->len2[tid];
    const int kind = S->kind[tid], kind2 = S->kind2[tid];
    const int wslot0 = S->wslot[tid], w2 = S->w2[tid];
    if (act && !cplx && !conf) {
        // an overlay walk that has assumed the writes of a lane that is not going to run (the matcher is past it)
#pragma unroll
        for (int k = 0; k < MRZ_WR_MAX; k++) {
            const int dk = S->dep[k][tid];
            if (dk < MRZ_W && S->q[dk] <= L.p) conf = true;
        }
    }
    {
        // first lane behind the matcher's position (the lanes are in position order); read after the barriers below
        const int fl = mrz_wave_first(live, wave, nb);
        if (lane == 0) S->wmin[3][wave] = fl;
    }
    ST_COUNT(MRZ_ST_W_DROP, have && !live);
    const bool conf_before = conf;
    (void)conf_before;
    // written since the preparation began?  The log entries of the ends of the two read ranges are asked for now and
    // looked at after the cull window has been loaded (one trip to memory for both)
    const bool ask_log = act && !cplx && !conf && snap < batch_no;
    const int lmask = smask >> MRZ_WLOG_SHIFT;
    const int b1a = h >> MRZ_WLOG_SHIFT, b1b = ((h + (len1 > 0 ? len1 - 1 : 0)) & smask) >> MRZ_WLOG_SHIFT;
    const int b2a = h2 >> MRZ_WLOG_SHIFT, b2b = ((h2 + (len2 > 0 ? len2 - 1 : 0)) & smask) >> MRZ_WLOG_SHIFT;
    unsigned g1a = 0, g1b = 0, g2a = 0, g2b = 0;
    if (ask_log && len1 > 0) {
        g1a = wlog[b1a & lmask];
        g1b = wlog[b1b];
    }
    if (ask_log && len2 > 0) {
        g2a = wlog[b2a & lmask];
        g2b = wlog[b2b];
    }

    // the cull window (only when this batch can reach the limit)
    const int64_t cw_base = L.clean_ptr;
    const bool want_cw = L.count + nb > C.limit;
    const int cw_len = want_cw ? MRZ_CW_WORDS * 64 : 0;
    if (want_cw) {
        for (int b = 0; b < MRZ_CW_WORDS / NW; b++) {
            const int wi = b * NW + wave;
            const int64_t slot = cw_base + (int64_t)wi * 64 + lane;
            mrz_slot e;
            e.off = 0;
            e.t = 0;
            if (slot < C.nslots) e = tab[slot];
            const mrz_u64 m = __ballot(((e.off | e.t) != 0) && ((e.t & better) != better));
            if (lane == 0) S->cw[wi] = m;
        }
    } else if (tid < MRZ_CW_WORDS)
        S->cw[tid] = 0ull;
    if (tid == 0) {
        S->cw_base = cw_base;
        S->cw_len = cw_len;
        S->floor_prep = L.last_match > 0 ? L.last_match : 0;
    }
    mrz_prep_sync<NW>();
    if (wave == 0) {
        const int c = lane < MRZ_CW_WORDS ? __popcll(S->cw[lane]) : 0;
        const int ci = mrz_wave_incl_sum(c, lane);
        if (lane < MRZ_CW_WORDS) S->cwcum[lane + 1] = ci;
        if (lane == 0) S->cwcum[0] = 0;
    }
    mrz_prep_sync<NW>();
    for (int r = tid; r < MRZ_W; r += 64 * NW)
        if (r < S->cwcum[MRZ_CW_WORDS]) S->cw_list[r] = mrz_cw_slot_search(S, cw_base, r);
    PROF_ADD(MRZ_ST_T_PC_CW);

    if (ask_log) {
        if (g1a > snap || g1b > snap || g2a > snap || g2b > snap) conf = true;
        // (ranges of more than two log blocks: the ones in between)
        for (int part = 0; part < 2 && !conf; part++) {
            const int ba = part ? b2a : b1a, bb = part ? b2b : b1b, rl = part ? len2 : len1;
            if (rl <= 0 || ba == bb) continue;
            for (int blk = (ba + 1) & lmask; blk != bb; blk = (blk + 1) & lmask)
                if (wlog[blk] > snap) conf = true;
        }
    }
    if (act && !cplx && !conf) {
        // a write that changes the set of failing entries inside the window would change the sweep; reads inside the
        // window: an earlier lane's cull may empty a slot this lane has read
        if (cw_len > 0) {
            if (ins) {
                if ((kind == 1 || loose) && kind != 3 && wslot0 >= cw_base && wslot0 < cw_base + cw_len) cplx = true;
                if (kind == 2 && (kind2 == 1 || loose) && w2 >= cw_base && w2 < cw_base + cw_len) cplx = true;
            }
            if (mrz_ranges_meet(h, len1, (int)cw_base, cw_len, smask) || mrz_ranges_meet(h2, len2, (int)cw_base, cw_len, smask))
                conf = true;
        }
    }
    ST_COUNT(MRZ_ST_W_STALE, conf && !conf_before);
    PROF_ADD(MRZ_ST_T_PC_LOG);
    {
        const int64_t floor_p = L.last_match > 0 ? L.last_match : 0;
        int64_t bl = 0, bo = 0;
        int br = 0, bh = 0, bm = 0;
        bool lng = false;
        if (act && !cplx && !conf) mrz_lane_best(S, tid, S->ns[tid], q, floor_p, &bl, &bo, &br, &bh, &bm, &lng);
        S->blen[tid] = bl;
        S->boff[tid] = bo;
        S->brev[tid] = br;
        S->bhm[tid] = (unsigned short)((bh << 8) | bm);
        if (have) {
            S->lf[tid] = (unsigned char)((conf ? MRZ_LF_CONF : 0) | (cplx ? MRZ_LF_CPLX : 0) | (act ? MRZ_LF_ACT : 0) |
                                         (ins ? MRZ_LF_INS : 0) | (lng ? MRZ_LF_LONG : 0) | (f0 & MRZ_LF_REVS));
            if (!live) S->exec[tid] = 2;
#ifdef MRZ_DBG_HITS
            if (live) {
                unsigned hsh = (unsigned)(conf ? 1 : 0) | (cplx ? 2 : 0) | (lng ? 4 : 0);
                hsh = hsh * 31u + (unsigned)bl;
                hsh = hsh * 31u + (unsigned)bh * 17u + (unsigned)bm;
                hsh = hsh * 31u + (unsigned)floor_p;
                DBG_PLANE(2, q, hsh | 1u);
            }
#endif
        }

        PROF_ADD(MRZ_ST_T_PC_BEST);
    }
    int s0 = S->wmin[3][0];  // (written before the cull window's barriers)
#pragma unroll
    for (int w = 1; w < NW; w++) s0 = S->wmin[3][w] < s0 ? S->wmin[3][w] : s0;
    if (tid == 0) {
        S->first_live = s0;
        S->rank0 = 0;
    }
    mrz_wide_bulk<NW>(C, L, S, wlog, batch_no, s0, 0, false, tid, lane, wave);
    PROF_ADD(MRZ_ST_T_PC_BULK);
}

// COMMIT of a prepared batch (phase E): wave 0 alone, no workgroup barrier.  Windows of up to 64 lanes starting at the
// first lane not yet dealt with; a window is cut at the first lane that cannot be committed as prepared.
__device__ static void mrz_wide_commit(const mrz_cfg &C, mrz_lead &L, mrz_wide_lds *S, unsigned *__restrict__ wlog,
                                       unsigned batch_no, int lane, int64_t *stat, mrz_wide_ret *ret) {
    const unsigned stamp = batch_no + 1u;  // what this batch leaves in the write log
    mrz_slot *tab = C.tab;
    const int smask = (int)C.slot_mask;
    const int max_chain = (int)C.max_chain;
    const int64_t better = (L.min_mask << 1) | 1;
    const bool loose = L.tag_mask != better;
    const int nb = mrz_uni(S->nb), total = mrz_uni(S->total);
    const int64_t cw_base = mrz_uni64(S->cw_base);
    const int cw_len = mrz_uni(S->cw_len);
    const int64_t floor_prep = mrz_uni64(S->floor_prep);
    ret->used = 0;
    ret->ok = true;
    ret->whole = true;
    ret->stop_batch = false;
    (void)total;
    PROF_T0();
    if (nb == 0) return;
    const int rank0 = mrz_uni(S->rank0);
    int s = mrz_uni(S->first_live), cw_used = rank0, committed = 0, iters = 0;
    bool bulk_end = false;
    bool moved = false;  // something has been committed or handed over since the last bulk step
    ret->rebulk = false;
    {
        // what the pre-commit step has committed in bulk already: lanes [first_live, y)
        const int y = mrz_uni(S->bulk_y);
        if (y > 0) {
            const int a1 = mrz_uni(S->bulk_a1), a2 = mrz_uni(S->bulk_a2);
            const int dsum = a1 & 1023, esum = (a1 >> 10) & 1023, isum = (a1 >> 20) & 1023;
            const int csum = a2 & 1023, msum = a2 >> 10;
            L.inserts += isum;
            int64_t cnew = L.count + dsum;
            if (cnew > C.limit) cnew = C.limit;
            L.count = cnew;
            if (csum) {
                L.clean_ptr = mrz_cw_slot(S, cw_base, rank0 + csum - 1);
                L.tag_mask = better;
                cw_used = rank0 + csum;
            }
            if (esum) L.victim_round = (int64_t)(((unsigned)L.victim_round + (unsigned)esum) % (unsigned)max_chain);
            L.tag_misses += msum;
            L.p = mrz_uni64(S->q[y - 1]);
            committed = y - s;
            s = y;
            bulk_end = mrz_uni(S->bulk_end) != 0;
            ST_ADD(MRZ_ST_SEGMENTS, 1);
        }
    }
    // a lane resolved by the long-match path: its exact result, valid for the last_match it was measured under
    int res_lane = -1;
    int64_t res_len = 0, res_off = 0, res_floor = -1;
    int res_rev = 0, res_h = 0, res_m = 0;
    while (s < nb && !bulk_end) {
        if (++iters > 8 * MRZ_W) {  // cannot happen: every iteration commits, drops or repairs a lane
            if (lane == 0) C.st->error = 3;
            ret->ok = false;
            break;
        }
        const int64_t floor_p = L.last_match > 0 ? L.last_match : 0;
        const int f_s = mrz_uni(S->lf[s]);
        const bool s_quick = (f_s & MRZ_LF_ACT) && (f_s & (MRZ_LF_CONF | MRZ_LF_CPLX));  // no need to look at the window
        const int i = s + lane;
        const bool have = !s_quick && i < nb;
        const int ii = have ? i : nb - 1;
        const int f = have ? S->lf[ii] : 0;
        const bool act = (f & MRZ_LF_ACT) != 0, ins = (f & MRZ_LF_INS) != 0, cplx = (f & MRZ_LF_CPLX) != 0;
        bool conf = (f & MRZ_LF_CONF) != 0;
        const int64_t q = S->q[ii];
        // late staleness: a lane whose speculated writes this lane had laid over its walk did not commit them, or a
        // cooperative hand-over inside this batch wrote into what it has read
        if (have && act && !cplx && !conf) {
            // (a lane of this very window either commits together with this one or cuts the segment before it)
#pragma unroll
            for (int k = 0; k < MRZ_WR_MAX; k++) {
                const int dk = S->dep[k][ii];
                if (dk < s && S->exec[dk] != 1) conf = true;
            }
            const int nx = S->xw_n;
            if (nx > 0) {
                const int h = S->h[ii], len1 = S->len1[ii], h2 = S->h2[ii], len2 = S->len2[ii];
                for (int k = 0; k < nx && !conf; k++) {
                    const int sl = S->xw_slot[k];
                    if ((len1 > 0 && mrz_in_range(sl, h, len1, smask)) || (len2 > 0 && mrz_in_range(sl, h2, len2, smask)))
                        conf = true;
                }
            }
        }
        // the lane's best match under the current last_match
        int64_t blen = 0, boff = 0;
        int brev = 0, bh = 0, bm = 0;
        bool lng = false;
        if (have && act && !cplx && !conf) {
            if (i == res_lane && floor_p == res_floor) {
                blen = res_len;
                boff = res_off;
                brev = res_rev;
                bh = res_h;
                bm = res_m;
            } else if ((f & MRZ_LF_REVS) && (q - floor_p <= 64 || (floor_p != floor_prep && q - floor_prep <= 64))) {
                mrz_lane_best(S, ii, S->ns[ii], q, floor_p, &blen, &boff, &brev, &bh, &bm, &lng);
            } else {
                blen = S->blen[ii];
                boff = S->boff[ii];
                brev = S->brev[ii];
                const int hm = S->bhm[ii];
                bh = hm >> 8;
                bm = hm & 255;
                lng = (f & MRZ_LF_LONG) != 0;
            }
        }
#ifdef MRZ_DBG_HITS
        int dbg_sub = 0;
        if (have && act && !cplx && !conf)
            dbg_sub = (i == res_lane && floor_p == res_floor) ? 3 : (((f & MRZ_LF_REVS) && (q - floor_p <= 64 || (floor_p != floor_prep && q - floor_prep <= 64))) ? 2 : 1);
#endif
        const bool stop = have && act && (cplx || conf || lng);
        const mrz_u64 m_stop = __ballot(stop);
        int nseg = nb - s < 64 ? nb - s : 64;
        if (m_stop) nseg = __ffsll((long long)m_stop) - 1;
        if (s_quick) nseg = 0;
        if (nseg == 0) {
            // ---- lane s cannot be committed as prepared ---------------------------------------------------------
            const bool s_cplx = s_quick ? (f_s & MRZ_LF_CPLX) != 0 : mrz_lane_read((int)cplx, 0) != 0;
            const bool s_conf = s_quick ? (f_s & MRZ_LF_CONF) != 0 : mrz_lane_read((int)conf, 0) != 0;
            if (s_cplx || s_conf) {
                // stale (an earlier lane wrote into what it read) or beyond the per-lane walk (long chain, deep
                // cascade): the cooperative path replays this one candidate in full against the table as committed
                // so far; the batch goes on behind it
                ST_ADD(s_cplx ? MRZ_ST_CUT_CPLX : MRZ_ST_REWALK, 1);
                PROF_ADD(MRZ_ST_H_PRE);
                const int64_t pre_min = L.min_mask, pre_tag = L.tag_mask, pre_events = L.n_events;
                const int64_t q_e = mrz_uni64(S->q[s]);
                L.p = q_e;
                const bool okc = mrz_seq_candidate(C, L, &S->coop, mrz_uni64(S->t[s]), lane, stat);
                PROF_ADD(MRZ_ST_H_CAND);
                committed += 1;
                moved = true;
                if (!okc) {
                    ret->ok = false;
                    break;
                }
                if (lane == 0) S->exec[s] = 2;
                const int nwx = mrz_uni(S->coop.n_written);
                const int64_t cullx = mrz_uni64(S->coop.cull_slot);
                bool stop_batch = L.min_mask != pre_min || L.tag_mask != pre_tag;
                ST_ADD(MRZ_ST_E_MASK, stop_batch ? 1 : 0);
                // its cull must be the one the window expects next; its insert must not have touched the window
                if (cullx >= 0) {
                    if (cw_len == 0 || cw_used >= S->cwcum[MRZ_CW_WORDS] || mrz_cw_slot(S, cw_base, cw_used) != (int)cullx) {
                        stop_batch = true;
                        ST_ADD(MRZ_ST_E_CULL, 1);
                    } else
                        cw_used++;
                }
                int nx = mrz_uni(S->xw_n);
                if (nx + nwx + 1 > MRZ_XW_MAX) {
                    ST_ADD(MRZ_ST_E_XW, 1);
                    stop_batch = true;
                    if (lane < nwx) wlog[S->coop.pend_h[lane] >> MRZ_WLOG_SHIFT] = stamp;
                    if (cullx >= 0 && lane == 0) wlog[cullx >> MRZ_WLOG_SHIFT] = stamp;
                } else {
                    // its writes: the lanes behind it look at them when their window comes up
                    if (lane < nwx) {
                        const int64_t hs = S->coop.pend_h[lane];
                        S->xw_slot[nx + lane] = (int)hs;
                        wlog[hs >> MRZ_WLOG_SHIFT] = stamp;
                    }
                    if (cullx >= 0 && lane == 0) {
                        S->xw_slot[nx + nwx] = (int)cullx;
                        wlog[cullx >> MRZ_WLOG_SHIFT] = stamp;
                    }
                    const bool inwin = lane < nwx && cw_len > 0 && S->coop.pend_h[lane] >= cw_base &&
                                       S->coop.pend_h[lane] < cw_base + cw_len;
                    if (__ballot(inwin)) {
                        // A write into the cull window changes which entries the sweep finds: the window's mask is
                        // brought up to date (ahead of what the sweep has passed -- behind it nothing matters any more,
                        // and the ranks already used must keep their meaning), counts and list are laid out again.
                        ST_ADD(MRZ_ST_E_INWIN, 1);
                        const int64_t passed = cw_used > 0 ? (int64_t)mrz_cw_slot(S, cw_base, cw_used - 1) : cw_base - 1;
                        MRZ_WAVE_SYNC();
                        for (int k = 0; k < nwx; k++) {
                            const int64_t X = mrz_uni64(S->coop.pend_h[k]);
                            if (X < cw_base || X >= cw_base + cw_len || X <= passed) continue;
                            const int64_t tX = mrz_uni64(S->coop.pend_t[k]);
                            const bool failing = (tX & better) != better;  // (the slot now holds an entry)
                            if (lane == 0) {
                                const int bit = (int)(X - cw_base);
                                const mrz_u64 m = 1ull << (bit & 63);
                                S->cw[bit >> 6] = failing ? (S->cw[bit >> 6] | m) : (S->cw[bit >> 6] & ~m);
                            }
                            MRZ_WAVE_SYNC();
                        }
                        {
                            const int c = lane < MRZ_CW_WORDS ? __popcll(S->cw[lane]) : 0;
                            const int ci = mrz_wave_incl_sum(c, lane);
                            if (lane < MRZ_CW_WORDS) S->cwcum[lane + 1] = ci;
                            MRZ_WAVE_SYNC();
                            const int tot = S->cwcum[MRZ_CW_WORDS];
                            for (int r = lane; r < MRZ_W && r < tot; r += 64) S->cw_list[r] = mrz_cw_slot_search(S, cw_base, r);
                            MRZ_WAVE_SYNC();
                        }
                        // (the ranks used so far lie behind `passed`: their bits were not touched)
                    }
                    nx += nwx + (cullx >= 0 ? 1 : 0);
                    if (lane == 0) S->xw_n = nx;
                }
                MRZ_WAVE_SYNC();
                int s_next = s + 1;
                if (L.n_events != pre_events) {
                    ST_ADD(MRZ_ST_EMITS, 1);
                    if (L.last_match >= q_e) {
                        int lo = s + 1, hi = nb;
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (mrz_uni64(S->q[mid]) > L.last_match)
                                hi = mid;
                            else
                                lo = mid + 1;
                        }
                        s_next = lo;
                        for (int k = s + 1 + lane; k < s_next; k += 64) S->exec[k] = 2;  // dropped
                        if (s_next >= nb) ST_ADD(MRZ_ST_SKIPOUT, 1);
                    } else {
                        ST_ADD(MRZ_ST_BACKJUMP, 1);
                        s_next = s;  // comes by q_e again, after its own insert: the cooperative path once more
                        if (lane == 0) S->lf[s] = (unsigned char)(S->lf[s] | MRZ_LF_CONF);
                    }
                }
                MRZ_WAVE_SYNC();
                s = s_next;
                PROF_ADD(MRZ_ST_H_POST);
                if (stop_batch) {
                    ret->stop_batch = true;
                    break;
                }
                continue;
            }
            // tag-equal entries beyond the 64-byte reach: measured exactly (striped rounds, compare farm) and folded
            // in probe order
            ST_ADD(MRZ_ST_LONGRES, 1);
            const int nsx = mrz_uni(S->ns[s]);
            const int64_t qx = mrz_uni64(S->q[s]);
            if (lane < nsx) {
                const int c = S->chunk_id[s][lane >> 2];
                const unsigned long long v = S->pool[c].e[lane & 3];
                const int64_t op = (int64_t)(v & MRZ_OFF_MASK);
                int64_t ml;
                int rv;
                bool l;
                mrz_pair_eval(S->pool[c].raw[lane & 3], qx, op, floor_p, &ml, &rv, &l);
                S->coop.same_off[lane] = op;
                S->coop.pair_res[lane] = l ? -1 : (int)((ml << 8) | rv);
            }
            MRZ_WAVE_SYNC();
            int64_t xb = 0, xoff = 0, xrev = 0;
            int xh = 0, xm = 0;
            mrz_resolve_entries(C, L, &S->coop, qx, nsx, lane, stat, &xb, &xoff, &xrev, &xh, &xm);
            res_lane = s;
            res_len = xb;
            res_off = xoff;
            res_rev = (int)xrev;
            res_h = xh;
            res_m = xm;
            res_floor = floor_p;
            PROF_ADD(MRZ_ST_T_LONG);
            continue;
        }

        // a whole window of clean lanes without a match, and more behind it: all waves commit such a stretch faster
        bool ask_bulk = moved && nseg == 64 && nb - s >= 64 + MRZ_W / 8 && L.cur_len < MRZ_MIN_MATCH &&
                        !__ballot(have && act && blen > 0);
        if (ask_bulk) {
            // ... and the lanes of the window behind this one look clean as well (as prepared: a peek, not the rules)
            const int i2 = s + 64 + lane;
            const bool in2 = i2 < nb && lane < MRZ_W / 8;
            const int f2 = in2 ? S->lf[i2] : 0;
            const bool dirty2 = in2 && (f2 & MRZ_LF_ACT) && ((f2 & (MRZ_LF_CONF | MRZ_LF_CPLX | MRZ_LF_LONG)) || S->blen[i2] > 0);
            if (__ballot(dirty2)) ask_bulk = false;
        }
        if (ask_bulk) {
            ret->rebulk = true;
            ST_ADD(MRZ_ST_REBULK, 1);
            if (lane == 0) {
                S->first_live = s;
                S->rank0 = cw_used;
            }
            break;
        }
        moved = true;
        // ---- the segment [s, s + nseg): sequential quantities by wave scans ----------------------------------------
        ST_ADD(MRZ_ST_SEGMENTS, 1);
        const bool inseg = lane < nseg;
        const int kind = S->kind[ii], kind2 = S->kind2[ii];
        const bool a_ins = inseg && ins && act;
        const bool a_ev = a_ins && kind == 3;
        const int d = a_ins ? (kind == 0 ? 1 : (kind == 2 ? (kind2 == 0 ? 1 : 0) : 0)) : 0;
        const int i1 = mrz_wave_incl_sum(d | (a_ev ? 1 << 8 : 0) | (a_ins ? 1 << 16 : 0), lane);
        int wslot = S->wslot[ii];
        if (a_ev) {
            // victim_round for evicting lanes (static victim_round, src/rzip.c:259,283-289)
            const int er = ((i1 >> 8) & 255) - 1;
            const int vr = (int)(((unsigned)L.victim_round + (unsigned)er) % (unsigned)max_chain);
            wslot = (int)(mrz_pool_get(S, ii, vr) >> MRZ_OFF_BITS);
        }
        // hash_count before each lane: saturating prefix sum of the per-lane deltas
        int64_t c_before = L.count + ((i1 & 255) - d);
        if (c_before > C.limit) c_before = C.limit;
        const bool cull = a_ins && (c_before + d > C.limit);
        const int i2 = mrz_wave_incl_sum((cull ? 1 : 0) | ((inseg && act ? bh : 0) << 8) | ((inseg && act ? bm : 0) << 20), lane);
        int cslot = -1;
        bool overflow = false;
        if (cull) {
            const int cr = cw_used + (i2 & 255) - 1;
            if (cr >= S->cwcum[MRZ_CW_WORDS])
                overflow = true;  // the sweep leaves the window (or wraps / promotes)
            else
                cslot = mrz_cw_slot(S, cw_base, cr);
        }
        // ---- the lazy-match fold (src/rzip.c:586-599) as a prefix maximum: first longest wins ---------------
        mrz_u64 K = mrz_wave_incl_max64((inseg && act) ? (((mrz_u64)blen << 7) | (mrz_u64)(127 - (lane + 1))) : 0ull, lane);
        {
            const mrz_u64 k0 = ((mrz_u64)L.cur_len << 7) | 127ull;
            if (k0 > K) K = k0;
        }
        const int arel = 127 - (int)(K & 127ull);  // 0: the match carried in; else lane arel - 1 of this window
        const int64_t a_q = mrz_shfl64(q, arel > 0 ? arel - 1 : 0);
        const int a_rev = __shfl(brev, arel > 0 ? arel - 1 : 0, MRZ_WAVE);
        const int64_t a_off = mrz_shfl64(boff, arel > 0 ? arel - 1 : 0);
        const int64_t curlen = (int64_t)(K >> 7);
        const int64_t curp = arel == 0 ? L.cur_p : a_q - a_rev;
        const int64_t curofs = arel == 0 ? L.cur_ofs : a_off;
        const bool emit = inseg && act && curlen >= MRZ_MIN_MATCH && (curlen >= MRZ_GREAT_MATCH || q >= curp + MRZ_MIN_MATCH);
        const mrz_u64 m_emit = __ballot(emit), m_over = __ballot(overflow), m_cull = __ballot(cull);
        const int e_lane = m_emit ? __ffsll((long long)m_emit) - 1 : 64;
        const int o_lane = m_over ? __ffsll((long long)m_over) - 1 : 64;
        const int c_lane = m_cull ? __ffsll((long long)m_cull) - 1 : 64;
        int y = nseg;  // lanes [0, y) of this window commit
        bool end_batch = false;
        if (o_lane < y) {
            y = o_lane;
            end_batch = true;
        }
        // the first cull ever switches the insert mask (:583): nothing after it in this batch
        if (loose && c_lane + 1 <= y) {
            y = c_lane + 1;
            end_batch = true;
        }
        bool emission = false;
        if (e_lane + 1 <= y) {
            y = e_lane + 1;
            emission = true;
            end_batch = loose && c_lane + 1 <= y;
        }
        if (y == 0) {
            // the very next lane cannot be served from the cull window (it is used up, or the sweep wraps / the mask
            // is promoted): the lane is marked complex and goes through the cooperative path in the next iteration;
            // the window is void after that
            ST_ADD(MRZ_ST_CUT_OVERFLOW, 1);
            if (lane == 0) S->lf[s] = (unsigned char)(S->lf[s] | MRZ_LF_CPLX);
            MRZ_WAVE_SYNC();
            continue;
        }
        PROF_ADD(MRZ_ST_T_SCAN);

        // ---- commit lanes [0, y) of the window ----------------------------------------------------------------
        if (lane < y && a_ins) {
            if (kind == 2 && S->supp_w2[ii] >= s + y) {
                mrz_slot oc;
                oc.off = S->occ_off[ii];
                oc.t = S->occ_t[ii];
                const int w2s = S->w2[ii];
                tab[w2s] = oc;
                wlog[w2s >> MRZ_WLOG_SHIFT] = stamp;
            }
            if (S->supp_w[ii] >= s + y) {
                mrz_slot nw;
                nw.off = q;
                nw.t = S->t[ii];
                tab[wslot] = nw;
                wlog[wslot >> MRZ_WLOG_SHIFT] = stamp;
            }
            if (cslot >= 0) {
                mrz_slot z;
                z.off = 0;
                z.t = 0;
                tab[cslot] = z;
                wlog[cslot >> MRZ_WLOG_SHIFT] = stamp;
            }
        }
        if (lane < y) S->exec[ii] = 1;
        {
            const int a1 = mrz_lane_read(i1, y - 1), a2 = mrz_lane_read(i2, y - 1);
            const int dsum = a1 & 255, esum = (a1 >> 8) & 255, isum = (a1 >> 16) & 255;
            const int csum = a2 & 255, hsum = (a2 >> 8) & 4095, msum = (a2 >> 20) & 4095;
            if (inseg && lane < y && act) DBG_HITS(q, bh, bm);
#ifdef MRZ_DBG_HITS
            if (inseg && lane < y && act) DBG_PLANE(3, q, ((unsigned)dbg_sub << 28) | ((unsigned)floor_p & 0xfffffffu));
#endif
            L.inserts += isum;
            int64_t cnew = L.count + dsum;
            if (cnew > C.limit) cnew = C.limit;
            L.count = cnew;
            if (csum) {
                L.clean_ptr = mrz_cw_slot(S, cw_base, cw_used + csum - 1);
                L.tag_mask = better;
                cw_used += csum;
            }
            if (esum) L.victim_round = (int64_t)(((unsigned)L.victim_round + (unsigned)esum) % (unsigned)max_chain);
            L.tag_hits += hsum;
            L.tag_misses += msum;
            L.cur_len = mrz_bcast64(curlen, y - 1);
            L.cur_p = mrz_bcast64(curp, y - 1);
            L.cur_ofs = mrz_bcast64(curofs, y - 1);
            L.p = mrz_bcast64(q, y - 1);
            committed += y;
        }
        int s_next = s + y;
        if (emission) {
            ST_ADD(MRZ_ST_EMITS, 1);
            const int64_t q_e = L.p;
            if (L.n_events >= C.event_cap) {  // cannot happen: matches are >= 31 bytes and disjoint
                if (lane == 0) C.st->error = 1;
                ret->ok = false;
                break;
            }
            if (lane == 0) {
                mrz_event ev;
                ev.p = L.cur_p;
                ev.ofs = L.cur_ofs;
                ev.len = L.cur_len;
                C.events[L.n_events] = ev;
            }
            L.n_events++;
            L.last_len = L.cur_len;
            L.mbytes += L.cur_len;
            L.last_match = L.cur_p + L.cur_len;
            L.cur_p = L.p = L.last_match;
            L.cur_len = 0;
            if (L.last_match >= q_e) {
                // the lanes inside the match are dropped: first lane behind it (mostly still in this window)
                int lo = s + y, hi = nb;
                {
                    const mrz_u64 m_beh = __ballot(have && lane >= y && q > L.last_match);
                    if (m_beh)
                        lo = hi = s + (__ffsll((long long)m_beh) - 1);
                    else if (s + 64 < nb)
                        lo = s + 64;
                    else
                        lo = hi = nb;
                }
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (mrz_uni64(S->q[mid]) > L.last_match)
                        hi = mid;
                    else
                        lo = mid + 1;
                }
                s_next = lo;
                for (int k = s + y + lane; k < s_next; k += 64) S->exec[k] = 2;
                if (s_next >= nb) ST_ADD(MRZ_ST_SKIPOUT, 1);
            } else {
                // the match ends before the emitting position: the loop goes BACK (p = last_match, :596) and comes
                // by q_e again -- the only candidate of (last_match, q_e] -- after its own insert
                ST_ADD(MRZ_ST_BACKJUMP, 1);
                s_next = s + y - 1;
                if (lane == 0) {
                    S->lf[s_next] = (unsigned char)(S->lf[s_next] | MRZ_LF_CONF);
                    S->exec[s_next] = 2;  // what it wrote stands, but lanes that assumed its writes read again
                }
            }
        }
        MRZ_WAVE_SYNC();
        s = s_next;
        PROF_ADD(MRZ_ST_T_COMMIT);
        if (end_batch) {
            ST_ADD(MRZ_ST_E_WINDOW, 1);
            ret->stop_batch = true;
            break;
        }
    }
    if (bulk_end) {
        ST_ADD(MRZ_ST_E_BULK, 1);
        ret->stop_batch = true;
    }
    ret->whole = s >= nb && !ret->stop_batch;
    ST_ADD(MRZ_ST_COMMITTED, committed);
    ret->used = committed;
    MRZ_STORES_DONE();  // (before the barrier behind which the other waves may prepare the rest of the window)
}
