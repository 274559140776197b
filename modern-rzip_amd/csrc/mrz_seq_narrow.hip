// mrz_seq_narrow.hip -- the NARROW engine of the sequencer: one wave walks the state machine, 64 lanes wide.
//
// Same state machine, same matcher state (mrz_seq_state), same candidate list and same compare farm protocol as the
// wide engine in mrz_sequencer.hip -- the shared pieces (cooperative path, farm, long-match compares, lower bound into
// the candidate list) are mrz_seq_common.h's, included by both; the host picks one of the two kernels per segment
// launch (mrz_capi.hip) from the regime the previous segments were in.  This one has the shortest dependency chain per
// emitted match: it is what streams that are one long match after another run on (BASELINE configs[1]); the wide
// engine (512 candidates per batch, commit in segments) is what dense candidates without long matches run on (text,
// noise).
//
// One workgroup of three waves runs it (block `xcd` of the grid; the other blocks are the compare farm's helpers):
//
//   wave 0 ("leader") walks the state machine; all its control values are wave-uniform (kept in SGPRs via
//   readlane/readfirstlane) and every step is 64 lanes wide --
//     * candidates come from the front end's list (mrz_tagscan.hip): the `continue` at src/rzip.c:573 means ONLY
//       positions passing minimum_tag_mask run the loop body, emit test included; a cursor into the list follows the
//       matcher's position (lower bound = tile offset + popcount of the tile's pass bits after a match has moved it);
//     * find_best_match (:426-462) and the probe walk of insert_hash (:262-297) share one pass over the chain: 64
//       consecutive slots (1 KiB, coalesced) per step; ballots give first-empty, tag-equal lanes, and the insert
//       walk's stop (empty / due-for-culling / lower-ranked occupant / the max_chain_len-th same-tag entry);
//     * cascades of displaced occupants are collected and written back innermost-first like the reference's recursion;
//     * clean_one_from_hash (:305-328): 64 slots per sweep step;
//     * single_match_len (:372-397): every tag-equal entry of a step is extended by its own lane, 64 B each way, in
//       one load round trip (most differ there);
//   wave 1 ("stripe helper") waits on an LDS mailbox: a lone entry that runs past the 64-byte reach is extended by
//   both waves (4 KiB each per round, 64 lanes x 16 B x 4 pieces, ballot + ffs for the first mismatch); the last wave
//   ("scout") runs ahead of the leader and touches what it will need next.
//   Most candidates do not go one at a time: the BATCH ENGINE (mrz_batch_step) processes up to 64 consecutive
//   candidates, one lane each, speculatively against the table as it stands, and commits the prefix that provably
//   equals the sequential result.  Tag-equal entries that are all long go to the COMPARE FARM.
// Emitted matches go to an event list; record encoding, literal gathering and the CRC are separate parallel kernels.
//
// Bound: latency -- one dependency chain of table probes, data probes and cross-CU hand-offs; DESIGN.md 4.2.
#define MRZ_NARROW_ENGINE 1
#ifndef MRZ_SEQ_WAVES
#define MRZ_SEQ_WAVES 3  // leader, one stripe helper, scout.  Measured 8 -> 3: -6 % on the benchmark chunk (no VGPR
                         // spills, a third fewer SGPR spills, lighter helper workgroups); 2 (no scout) is slower
#endif
#include "mrz_seq_common.h"

// wave 0 = leader, waves 1..MRZ_STRIPE_WAVES-1 = striping helpers, last wave = scout (prefetcher)
#define MRZ_HAVE_SCOUT (MRZ_SEQ_WAVES > 2)
#ifndef MRZ_WIDTH_MULT
#define MRZ_WIDTH_MULT 2
#endif
#ifndef MRZ_WIDTH_DENSE
#define MRZ_WIDTH_DENSE 12
#endif
#ifndef MRZ_WIDTH_MULT_DENSE
#define MRZ_WIDTH_MULT_DENSE 8
#endif


// The scout (last wave of the leader's workgroup) never decides anything: it reads the
// leader's published position and pulls into this CU's L1 / this XCD's L2 what the leader
// will need next -- the tags of the following 128 candidates, the first table line of each
// of their probe chains, the bytes at the offsets of tag-equal entries, and the cull sweep
// window -- so that the leader's dependent loads hit cache instead of HBM.
struct mrz_scout_args {
    const uint8_t *buf;
    const mrz_slot *tab;
    mrz_cands K;
    int64_t lim, slot_mask, nslots;
};

__device__ static void mrz_scout_loop(const mrz_scout_args &S, mrz_mailbox *mb, int lane) {
    int seen = 0;
    unsigned sink = 0;
    while (true) {
        int s;
        while ((s = mrz_uni(mrz_mb_load(&mb->scout_seq))) == seen) {
            if (mrz_uni(mrz_mb_load(&mb->quit))) {
                if (sink == 0x9e3779b9u) mb->res[MRZ_SEQ_WAVES - 1] = (int64_t)sink;  // keeps the prefetch loads alive
                return;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        seen = s;
        const int64_t pos = mrz_uni64(mb->scout_pos) + 1;
        if (pos > S.lim) continue;
        // cull sweep window
        const int64_t cp = mrz_uni64(mb->scout_clean);
        if (cp + lane * 4 < S.nslots) sink += (unsigned)S.tab[cp + lane * 4].off;  // 4 KiB ahead of the sweep
        // the next 128 entries of the candidate list
        const int64_t i0 = mrz_cand_lower_bound(S.K, pos, lane);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int64_t i = i0 + half * 64 + lane;
            if (i < S.K.n) {
                const mrz_cand c = S.K.cand[i];
                if (c.off <= S.lim) {
                    const int64_t h = c.t & S.slot_mask;
                    const mrz_slot e0 = S.tab[h];
                    const mrz_slot e1 = S.tab[(h + 7) & S.slot_mask];  // the line may straddle
                    sink += (unsigned)e0.off + (unsigned)e1.off;
                    if (e0.t == c.t && e0.off > 0) sink += S.buf[e0.off];
                    sink += S.buf[c.off];
                }
            }
        }
    }
}

// helper waves: serve striped forward-extension rounds until the leader says quit
__device__ static void mrz_helper_loop(const uint8_t *__restrict__ buf, mrz_mailbox *mb, int wave, int lane) {
    int seen = 0;
    while (true) {
        int s;
        while ((s = mrz_uni(mrz_mb_load(&mb->seq))) == seen) __builtin_amdgcn_s_sleep(2);
        seen = s;
        if (mrz_uni(mrz_mb_load(&mb->quit))) return;
        const int64_t p0 = mrz_uni64(mb->p0), op = mrz_uni64(mb->op), maxf = mrz_uni64(mb->maxf);
        const int64_t base = mrz_uni64(mb->base) + (int64_t)wave * MRZ_STRIPE;
        const int64_t r = mrz_wave_fwd_stripe(buf, p0, op, maxf, base, lane);
        if (lane == 0) {
            mb->res[wave] = r;
            mrz_mb_add(&mb->done, 1);
        }
    }
}

#define MRZ_FILTER_SIZE 4096
#define MRZ_CULL_WINDOW 4  // x 64 slots scanned ahead of tag_clean_ptr per batch

struct mrz_batch_lds {
    int pref[64];
    int64_t qpos[64];
    int64_t same_off[64][MRZ_SMAX];
    int same_slot[64][MRZ_SMAX];
    int pair_res[64][MRZ_SMAX];  // (len << 8) | rev, or -1 = needs the cooperative path
    unsigned filter[MRZ_FILTER_SIZE];
    mrz_coop_lds coop;  // staging of the cooperative path (one candidate; long entries of one batch lane)
};




__device__ __forceinline__ unsigned mrz_filter_slot(int slot) {
    return ((unsigned)(slot >> 6) * 2654435761u) >> 20;  // 12 bits
}

// ---- the batch engine -------------------------------------------------------------
// Up to 64 consecutive candidates are processed at once, ONE LANE PER CANDIDATE,
// against the table as it stands at the start of the batch:
//   1. every lane walks its own probe chain (4 slots = 64 B per step) and records
//      first-empty, the tag-equal entries, and where insert_hash's walk stops
//      (empty / due-for-culling overwrite / lower-ranked occupant to displace / the
//      max_chain_len-th tag-equal entry => victim eviction);
//   1b. lanes that displace an occupant walk that occupant's chain too;
//   2. every lane extends its tag-equal candidates itself, 64 B each way;
//   3. wave scans turn the per-lane facts into the sequential quantities:
//      victim_round per evicting lane (prefix count), hash_count before each lane
//      (saturating prefix sum), which lanes cull and which sweep entry each culls
//      (rank into the ballot of failing entries ahead of tag_clean_ptr);
//   4. a lane may only be committed if no EARLIER lane's write (insert, displaced
//      re-insert, cull) falls inside the slots it read: an LDS filter keyed by
//      64-slot block flags suspects, suspects are checked exactly.  The batch is
//      cut at the first lane that conflicts or needs the cooperative path
//      (long match, long chain, deep cascade, sweep wrap, mask transition), the
//      lazy-match fold (:586-599) may cut it earlier at an emission;
//   5. all surviving lanes write their slots in one go.
// The committed prefix is exactly what the reference's loop would have done.
// Returns the number of list entries consumed (>= 1), or 0 when the first candidate has to go through
// mrz_seq_candidate.  The batch is entries [ci, ci + width) of the candidate list (those at positions <= lim).
__device__ static int mrz_batch_step(const mrz_cfg &C, mrz_lead &L, mrz_batch_lds *B, const mrz_cands &K, int64_t ci,
                                     int64_t lim, unsigned epoch, int width, int lane, bool *ok, int64_t *stat) {
    const uint8_t *__restrict__ buf = C.buf;
    mrz_slot *tab = C.tab;
    const int smask = (int)C.slot_mask;
    const int max_chain = (int)C.max_chain;
    const int64_t better = (L.min_mask << 1) | 1;
    *ok = true;
    PROF_T0();

    // ---- formation: lane r takes entry ci + r of the list -------------------
    int64_t q = 0, t = 0;
    bool have = lane < width && ci + lane < K.n;
    if (have) {
        const mrz_cand c = K.cand[ci + lane];
        q = c.off;
        t = c.t;
        have = q <= lim;
    }
    const bool act = have && (t & L.min_mask) == L.min_mask;
    const bool do_ins = act && (t & L.tag_mask) == L.tag_mask;

    PROF_ADD(MRZ_ST_T_FORM);
    // ---- phase 1: per-lane probe walk ------------------------------------------
    const int h = (int)(t & C.slot_mask);
    const int my_rank = mrz_ones_rank(t);
    int fe = -1, wslot = -1, kind = -1;  // kind: 0 empty, 1 overwrite, 2 displace, 3 evict
    int nsame = 0, round = 0;
    bool cplx = false, evict = false;
    int why = 0;  // first reason this lane needs the cooperative path (diagnostics)
    int64_t occ_t = 0, occ_off = 0;
    {
        bool walking = act;
        int s = h, steps = 0;
        int ustep = 0;  // per-lane steps taken so far (uniform)
        while (true) {
            const mrz_u64 m_walk = __ballot(walking);
            if (!m_walk) break;
            // a few long chains left after the first steps: finish those 64 slots at a time (below)
            if (ustep >= MRZ_WALK_LANE_STEPS && __popcll(m_walk) <= MRZ_WALK_COOP_MAX) break;
            ustep++;
            if (walking) {
                mrz_slot e[MRZ_WALK_SLOTS];
#pragma unroll
                for (int k = 0; k < MRZ_WALK_SLOTS; k++) e[k] = tab[(s + k) & smask];
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
                            B->same_off[lane][nsame] = e[k].off;
                            B->same_slot[lane][nsame] = slot;
                        } else {
                            cplx = true;
                            if (!why) why = MRZ_ST_CUT_WALK;
                        }
                        nsame++;
                    }
                }
                s += MRZ_WALK_SLOTS;
                if (walking && ++steps >= MRZ_WALK_STEPS) {
                    cplx = true;
                    if (!why) why = MRZ_ST_CUT_WALK;
                    walking = false;
                }
            }
        }
            // ---- stragglers: the rest of a long chain, the whole wave on one lane's chain (64 slots per step).
        // Same decisions as the per-slot code above, taken on ballots: first empty slot, tag-equal entries
        // in probe order, and -- while the insert position is still open -- the first slot that is due for
        // culling or holds a lower-ranked tag, unless max_chain_len tag-equal entries come first.
        for (mrz_u64 todo = __ballot(walking); todo; todo &= todo - 1) {
            const int o = __ffsll((long long)todo) - 1;
            const int64_t t_o = mrz_bcast64(t, o);
            const int rank_o = mrz_lane_read(my_rank, o);
            const bool ins_o = mrz_lane_read((int)do_ins, o) != 0;
            int s_o = mrz_lane_read(s, o), fe_o = -1;
            int wslot_o = mrz_lane_read(wslot, o), kind_o = mrz_lane_read(kind, o), round_o = mrz_lane_read(round, o);
            int nsame_o = mrz_lane_read(nsame, o), why_o = mrz_lane_read(why, o);
            bool evict_o = mrz_lane_read((int)evict, o) != 0, cplx_o = mrz_lane_read((int)cplx, o) != 0;
            int64_t occ_t_o = mrz_bcast64(occ_t, o), occ_off_o = mrz_bcast64(occ_off, o);
            for (int cstep = 0; fe_o < 0; cstep++) {
                if (cstep >= MRZ_WALK_COOP_STEPS) {
                    cplx_o = true;
                    if (!why_o) why_o = MRZ_ST_CUT_WALK;
                    break;
                }
                const int slot = (s_o + lane) & smask;
                const mrz_slot e = tab[slot];
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
                if ((m_same >> lane) & 1) {
                    const int idx = nsame_o + __popcll(m_same & mrz_low_mask(lane));
                    if (idx < MRZ_SMAX) {
                        B->same_off[o][idx] = e.off;
                        B->same_slot[o][idx] = slot;
                    }
                }
                nsame_o += __popcll(m_same);
                if (nsame_o > MRZ_SMAX) {
                    cplx_o = true;
                    if (!why_o) why_o = MRZ_ST_CUT_WALK;
                }
                if (fe_idx < 64) fe_o = (s_o + fe_idx) & smask;
                s_o += 64;
            }
            if (lane == o) {
                walking = false;
                fe = fe_o;
                wslot = wslot_o;
                kind = kind_o;
                round = round_o;
                nsame = nsame_o;
                why = why_o;
                evict = evict_o;
                cplx = cplx_o;
                occ_t = occ_t_o;
                occ_off = occ_off_o;
            }
        }
        MRZ_WAVE_SYNC();
    }
    if (evict && max_chain > MRZ_SMAX) {
        cplx = true;
        if (!why) why = MRZ_ST_CUT_WALK;
    }
    const int len1 = ((fe - h) & smask) + 1;  // slots [h, fe] were read

    PROF_ADD(MRZ_ST_T_WALK);
    // ---- phase 1b: walk of a displaced occupant (src/rzip.c:275-278) ------------
    int h2 = 0, w2 = -1, kind2 = -1, len2 = 0;
    {
        bool walking = act && !cplx && kind == 2;
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
                for (int k = 0; k < MRZ_WALK_SLOTS; k++) e[k] = tab[(s + k) & smask];
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
                        cplx = true;  // second-level displacement: cooperative path
                        if (!why) why = MRZ_ST_CUT_CASCADE;
                        walking = false;
                    } else if (e[k].t == occ_t) {
                        if (++round2 == max_chain) {
                            cplx = true;
                            if (!why) why = MRZ_ST_CUT_CASCADE;
                            walking = false;
                        }
                    }
                }
                s += MRZ_WALK_SLOTS;
                if (walking && ++steps >= MRZ_WALK_STEPS) {
                    cplx = true;
                    if (!why) why = MRZ_ST_CUT_WALK;
                    walking = false;
                }
            }
        }
        // stragglers, 64 slots per step: first of {empty, due for culling, lower-ranked} decides, unless
        // max_chain_len entries with the occupant's tag come first (then, as for a lower-ranked slot, the
        // cascade goes to the cooperative path)
        for (mrz_u64 todo = __ballot(walking); todo; todo &= todo - 1) {
            const int o = __ffsll((long long)todo) - 1;
            const int64_t ot = mrz_bcast64(occ_t, o);
            const int rk = mrz_lane_read(rank2, o);
            int s_o = mrz_lane_read(s, o), r2 = mrz_lane_read(round2, o);
            int w2_o = -1, kind2_o = -1, why_o = mrz_lane_read(why, o);
            bool cplx_o = false, done = false;
            for (int cstep = 0; !done; cstep++) {
                if (cstep >= MRZ_WALK_COOP_STEPS) {
                    cplx_o = true;
                    if (!why_o) why_o = MRZ_ST_CUT_WALK;
                    break;
                }
                const int slot = (s_o + lane) & smask;
                const mrz_slot e = tab[slot];
                const bool empty = (e.off | e.t) == 0;
                const bool worse = !empty && (e.t & better) != better;
                const bool lower = !empty && !worse && mrz_ones_rank(e.t) < rk;
                const mrz_u64 m_empty = __ballot(empty), m_worse = __ballot(worse), m_lower = __ballot(lower);
                const mrz_u64 m_stop = m_empty | m_worse | m_lower;
                const int ks = m_stop ? __ffsll((long long)m_stop) - 1 : 64;
                const int nq = __popcll(__ballot(!empty && !worse && !lower && e.t == ot) & mrz_low_mask(ks));
                if (r2 + nq >= max_chain) {
                    cplx_o = true;
                    if (!why_o) why_o = MRZ_ST_CUT_CASCADE;
                    done = true;
                } else if (ks < 64) {
                    if ((m_lower >> ks) & 1) {
                        cplx_o = true;  // second-level displacement: cooperative path
                        if (!why_o) why_o = MRZ_ST_CUT_CASCADE;
                    } else {
                        w2_o = (s_o + ks) & smask;
                        kind2_o = ((m_empty >> ks) & 1) ? 0 : 1;
                    }
                    done = true;
                } else {
                    r2 += nq;
                    s_o += 64;
                }
            }
            if (lane == o) {
                walking = false;
                w2 = w2_o;
                kind2 = kind2_o;
                if (cplx_o) cplx = true;
                why = why_o;
            }
        }
        if (w2 >= 0) len2 = ((w2 - h2) & smask) + 1;
    }

    PROF_ADD(MRZ_ST_T_WALK2);
    // ---- phase 2: match extension (src/rzip.c:372-397), one (candidate, entry) pair per
    // lane per round: the pairs of all lanes are laid end to end and dealt out 64 at a time
    int64_t best = 0, best_off = 0, best_rev = 0;
    int hits = 0, misses = 0;
    bool needs_long = false;  // some entry runs past the 64-byte reach: resolved cooperatively at the cut
    {
        const int ns = (act && !cplx) ? (nsame < MRZ_SMAX ? nsame : MRZ_SMAX) : 0;
        const int pincl = mrz_wave_incl_sum(ns, lane);
        const int npairs = mrz_lane_read(pincl, 63);
        if (npairs) {
            B->pref[lane] = pincl - ns;
            B->qpos[lane] = q;
            MRZ_WAVE_SYNC();
            for (int base = 0; base < npairs; base += 64) {
                const int i = base + lane;
                if (i < npairs) {
                    int lo = 0, hi = 63;
#pragma unroll
                    for (int it = 0; it < 6; it++) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (B->pref[mid] <= i)
                            lo = mid;
                        else
                            hi = mid - 1;
                    }
                    const int k = i - B->pref[lo];
                    int64_t ml, rv;
                    bool lng;
                    mrz_lane_match_len(buf, B->qpos[lo], B->same_off[lo][k], C.end, L.last_match, &ml, &rv, &lng);
                    B->pair_res[lo][k] = lng ? -1 : (int)((ml << 8) | rv);
                }
            }
            MRZ_WAVE_SYNC();
            // every owner folds its entries in probe order: first longest wins (:446-450)
            for (int k = 0; k < ns; k++)
                if (B->pair_res[lane][k] < 0) needs_long = true;
            for (int k = 0; k < ns && !needs_long; k++) {
                const int r = B->pair_res[lane][k];
                const int64_t ml = r >> 8, rv = r & 0xff;
                if (ml) {
                    if (ml > best) {
                        best = ml;
                        best_off = B->same_off[lane][k] - rv;
                        best_rev = rv;
                    }
                    hits++;
                } else
                    misses++;
            }
            ST_ADD(MRZ_ST_PAIRS, npairs);
        }
    }

    PROF_ADD(MRZ_ST_T_PAIRS);
    // ---- phase 3: sequential quantities by wave scans ------------------------------
    const bool ins = do_ins && !cplx;
    // victim_round for evicting lanes (static victim_round, :259,283-289)
    const mrz_u64 m_evict = __ballot(ins && kind == 3);
    if (ins && kind == 3) {
        const int er = __popcll(m_evict & mrz_low_mask(lane));
        const int vr = (int)(((unsigned)L.victim_round + (unsigned)er) % (unsigned)max_chain);
        wslot = B->same_slot[lane][vr];
    }
    // hash_count before each lane: saturating prefix sum of the per-lane deltas
    int d = 0;
    if (ins) d = (kind == 0) ? 1 : (kind == 2 ? (kind2 == 0 ? 1 : 0) : 0);
    const int dincl = mrz_wave_incl_sum(d, lane);
    int64_t c_before = L.count + (dincl - d);
    if (c_before > C.limit) c_before = C.limit;
    const bool cull = ins && (c_before + d > C.limit);
    const mrz_u64 m_cull = __ballot(cull);
    int cslot = -1;
    if (m_cull) {
        // failing entries ahead of tag_clean_ptr (clean_one_from_hash, :313-321)
        mrz_u64 fmask[MRZ_CULL_WINDOW];
        int fcum[MRZ_CULL_WINDOW + 1];
        fcum[0] = 0;
#pragma unroll
        for (int b = 0; b < MRZ_CULL_WINDOW; b++) {
            const int64_t s = L.clean_ptr + b * 64 + lane;
            mrz_slot e;
            e.off = 0;
            e.t = 0;
            if (s < C.nslots) e = tab[s];
            fmask[b] = __ballot(((e.off | e.t) != 0) && ((e.t & better) != better));
            fcum[b + 1] = fcum[b] + __popcll(fmask[b]);
        }
        if (cull) {
            const int cr = __popcll(m_cull & mrz_low_mask(lane));
            if (cr >= fcum[MRZ_CULL_WINDOW]) {
                cplx = true;  // sweep leaves the window (or wraps / promotes): cooperative path
                if (!why) why = MRZ_ST_CUT_CULL;
            }
            else {
#pragma unroll
                for (int b = 0; b < MRZ_CULL_WINDOW; b++)
                    if (cslot < 0 && cr < fcum[b + 1])
                        cslot = (int)(L.clean_ptr + b * 64 + mrz_select64(fmask[b], cr - fcum[b]));
            }
        }
        // a write that changes the set of failing entries inside the scanned window would
        // change the sweep: overwrites of failing entries, and -- before the insert mask has
        // switched to `better` -- any insert (its tag may itself fail `better`)
        const int64_t win_end = L.clean_ptr + MRZ_CULL_WINDOW * 64;
        const bool loose = L.tag_mask != better;
        if (ins && (kind == 1 || loose) && wslot >= L.clean_ptr && wslot < win_end) {
            cplx = true;
            if (!why) why = MRZ_ST_CUT_CULL;
        }
        if (ins && kind == 2 && (kind2 == 1 || loose) && w2 >= L.clean_ptr && w2 < win_end) {
            cplx = true;
            if (!why) why = MRZ_ST_CUT_CULL;
        }
    }

    PROF_ADD(MRZ_ST_T_SCANS);
    // ---- phase 4: conflicts with earlier lanes' writes -------------------------------
    const bool writes = act && !cplx && do_ins;
    {
        const unsigned tagv = (epoch << 6) | (unsigned)(63 - lane);
        if (writes) {
            atomicMax(&B->filter[mrz_filter_slot(wslot)], tagv);
            if (kind == 2) atomicMax(&B->filter[mrz_filter_slot(w2)], tagv);
            if (cslot >= 0) atomicMax(&B->filter[mrz_filter_slot(cslot)], tagv);
        }
        MRZ_WAVE_SYNC();
        bool suspect = false;
        if (act && !cplx) {
            // blocks overlapped by [h, h+len1) and [h2, h2+len2)
            for (int x = h >> 6; !suspect; x = (x + 1) & (smask >> 6)) {
                const unsigned f = B->filter[((unsigned)x * 2654435761u) >> 20];
                if ((f >> 6) == epoch && (int)(63 - (f & 63)) < lane) suspect = true;
                if (x == (((h + len1 - 1) & smask) >> 6)) break;
            }
            if (len2 > 0)
                for (int x = h2 >> 6; !suspect; x = (x + 1) & (smask >> 6)) {
                    const unsigned f = B->filter[((unsigned)x * 2654435761u) >> 20];
                    if ((f >> 6) == epoch && (int)(63 - (f & 63)) < lane) suspect = true;
                    if (x == (((h2 + len2 - 1) & smask) >> 6)) break;
                }
        }
        mrz_u64 m_sus = __ballot(suspect);
        bool conflict = false;
        while (m_sus) {
            const int j = __ffsll((long long)m_sus) - 1;
            m_sus &= m_sus - 1;
            const int jh = mrz_lane_read(h, j), jl = mrz_lane_read(len1, j);
            const int jh2 = mrz_lane_read(h2, j), jl2 = mrz_lane_read(len2, j);
            bool hitj = false;
            if (writes && lane < j) {
                hitj = (((wslot - jh) & smask) < jl) || (jl2 > 0 && ((wslot - jh2) & smask) < jl2);
                if (kind == 2) hitj = hitj || (((w2 - jh) & smask) < jl) || (jl2 > 0 && ((w2 - jh2) & smask) < jl2);
                if (cslot >= 0)
                    hitj = hitj || (((cslot - jh) & smask) < jl) || (jl2 > 0 && ((cslot - jh2) & smask) < jl2);
            }
            if (__ballot(hitj) && lane == j) conflict = true;
        }
        if (conflict) {
            cplx = true;
            if (!why) why = MRZ_ST_CUT_CONFLICT;
        }
    }

    PROF_ADD(MRZ_ST_T_CONFLICT);
    // ---- cut the batch -------------------------------------------------------------
    const mrz_u64 m_have = __ballot(have);
    const mrz_u64 m_cplx = __ballot(have && act && cplx);
    const mrz_u64 m_long = __ballot(have && act && !cplx && needs_long);
    const mrz_u64 m_stop_any = m_cplx | m_long;
    int n_ok = m_stop_any ? __ffsll((long long)m_stop_any) - 1 : __popcll(m_have);
    ST_ADD(MRZ_ST_BATCH_FORMED, __popcll(m_have));
    if (m_stop_any && ((m_cplx >> n_ok) & 1)) {
        const int reason = mrz_lane_read(why, n_ok);
        if (reason > 0 && reason < MRZ_ST_N) ST_ADD(reason, 1);
    } else if (m_stop_any) {
        // The cut lane is sound except that some of its tag-equal entries run past the per-lane
        // reach: extend those with the whole workgroup (striped long path), fold the lane's
        // entries in probe order, and keep the lane as the last one of this batch.
        const int x = n_ok;
        ST_ADD(MRZ_ST_CUT_LONG, 1);
        const int nsx = mrz_lane_read(nsame < MRZ_SMAX ? nsame : MRZ_SMAX, x);
        const int64_t qx = mrz_bcast64(q, x);
        int64_t xb = 0, xoff = 0, xrev = 0;
        int xh = 0, xm = 0;
        if (lane < nsx) {
            B->coop.same_off[lane] = B->same_off[x][lane];
            B->coop.pair_res[lane] = B->pair_res[x][lane];
        }
        MRZ_WAVE_SYNC();
        if (!mrz_resolve_entries(C, L, &B->coop, qx, nsx, lane, stat, &xb, &xoff, &xrev, &xh, &xm)) {
            *ok = false;
            return 0;
        }
        PROF_ADD(MRZ_ST_T_LONG);
        if (lane == x) {
            best = xb;
            best_off = xoff;
            best_rev = xrev;
            hits = xh;
            misses = xm;
        }
        n_ok = x + 1;
    }
    // the first cull ever switches the insert mask (:583): nothing after it in this batch
    if (L.tag_mask != better) {
        const mrz_u64 mc = __ballot(cull && !cplx) & mrz_low_mask(n_ok);
        if (mc) n_ok = __ffsll((long long)mc);
    }
    if (n_ok == 0) {
        PROF_ADD(MRZ_ST_T_COMMIT);
        return 0;
    }

    // ---- lazy-match fold over the surviving lanes (src/rzip.c:586-599) ---------------
    int64_t cur_p = L.cur_p, cur_len = L.cur_len, cur_ofs = L.cur_ofs;
    int emit_lane = -1;
    {
        int start = 0;
        while (true) {
            const mrz_u64 in_range = mrz_low_mask(n_ok) & ~mrz_low_mask(start);
            const mrz_u64 m_thr = (cur_len >= MRZ_MIN_MATCH)
                                      ? (__ballot(act && q >= cur_p + MRZ_MIN_MATCH) & in_range)
                                      : (__ballot(false) & 0ull);
            const mrz_u64 m_adopt = __ballot(act && best > cur_len) & in_range;
            const int e_lane = m_thr ? __ffsll((long long)m_thr) - 1 : 64;
            const int a_lane = m_adopt ? __ffsll((long long)m_adopt) - 1 : 64;
            if (e_lane == 64 && a_lane == 64) break;
            if (e_lane < a_lane) {
                emit_lane = e_lane;
                break;
            }
            const int64_t aq = mrz_bcast64(q, a_lane);
            const int64_t ab = mrz_bcast64(best, a_lane);
            cur_p = aq - mrz_bcast64(best_rev, a_lane);
            cur_len = ab;
            cur_ofs = mrz_bcast64(best_off, a_lane);
            if (cur_len >= MRZ_GREAT_MATCH || aq >= cur_p + MRZ_MIN_MATCH) {
                emit_lane = a_lane;
                break;
            }
            start = a_lane + 1;
        }
    }
    if (emit_lane >= 0) {
        n_ok = emit_lane + 1;
        ST_ADD(MRZ_ST_BATCH_EMITS, 1);
    }
    PROF_ADD(MRZ_ST_T_FOLD);
    ST_ADD(MRZ_ST_BATCHES, 1);
    ST_ADD(MRZ_ST_BATCH_LANES, n_ok);
    const mrz_u64 keep = mrz_low_mask(n_ok);
    const bool mine = have && ((keep >> lane) & 1);

    // ---- phase 5: commit -------------------------------------------------------------
    if (mine && act && do_ins) {
        if (kind == 2) {
            mrz_slot o;
            o.off = occ_off;
            o.t = occ_t;
            tab[w2] = o;
        }
        mrz_slot n;
        n.off = q;
        n.t = t;
        tab[wslot] = n;
        if (cslot >= 0) {
            mrz_slot z;
            z.off = 0;
            z.t = 0;
            tab[cslot] = z;
        }
    }
    const mrz_u64 m_ins = __ballot(mine && act && do_ins);
    L.inserts += __popcll(m_ins);
    const int dtot = mrz_lane_read(dincl, n_ok - 1);
    int64_t cnew = L.count + dtot;
    if (cnew > C.limit) cnew = C.limit;
    L.count = cnew;
    const mrz_u64 m_cull_kept = m_cull & keep;
    if (m_cull_kept) {
        const int lastc = 63 - __clzll((long long)m_cull_kept);
        L.clean_ptr = mrz_lane_read(cslot, lastc);
        L.tag_mask = better;
    }
    if (m_evict & keep)  // 32-bit arithmetic: both operands are small, a 64-bit modulo is ~200 instructions
        L.victim_round = (int64_t)(((unsigned)L.victim_round + (unsigned)__popcll(m_evict & keep)) % (unsigned)max_chain);
    const int hsum = mrz_wave_incl_sum(mine ? hits : 0, lane);
    const int msum = mrz_wave_incl_sum(mine ? misses : 0, lane);
    L.tag_hits += mrz_lane_read(hsum, 63);
    L.tag_misses += mrz_lane_read(msum, 63);
    L.cur_p = cur_p;
    L.cur_len = cur_len;
    L.cur_ofs = cur_ofs;
    L.p = mrz_bcast64(q, n_ok - 1);
    if (emit_lane >= 0) {
        // the emission itself: cur already adopted, so only the emit half runs
        *ok = mrz_select_emit(C, L, 0, 0, 0, lane);
    }
    PROF_ADD(MRZ_ST_T_COMMIT);
    return n_ok;
}

__global__ __launch_bounds__(MRZ_SEQ_THREADS) void mrz_seq_narrow_kernel(mrz_seq_args a) {
    __shared__ mrz_mailbox mbox;
    __shared__ mrz_batch_lds batch;

    const int lane = threadIdx.x & 63;
    const int wave = mrz_uni((int)(threadIdx.x >> 6));
    mrz_seq_state *st = a.st;
    mrz_mailbox *mb = &mbox;

    if (st->finished || st->error) return;
    // block `xcd` is the sequencer workgroup (so that concurrent contexts sit on different XCDs), the others helpers
    const int lead_block = (a.xcd & 7) < (int)gridDim.x ? (a.xcd & 7) : 0;
#if MRZ_HELPER_WGS > 0
    if ((int)blockIdx.x != lead_block) {
        if (a.gmailbox) mrz_helper_wg(a.buf, (mrz_gmailbox *)a.gmailbox);
        return;
    }
#else
    if ((int)blockIdx.x != lead_block) return;
#endif
    mrz_cands K;  // the segment the front end has laid out
    K.cand = a.cand;
    K.tile_off = a.tile_off;
    K.bitmap = a.bitmap;
    K.seg_start = st->seg_start;
    K.seg_end = st->seg_end;
    K.n = st->n_cand;
    if (K.seg_end <= K.seg_start) {  // (nothing was scanned: the chunk is done)
#if MRZ_HELPER_WGS > 0
        if (threadIdx.x == 0 && a.gmailbox) mrz_g_storeu(&((mrz_gmailbox *)a.gmailbox)->quit, 1ull);
#endif
        return;
    }
    const int64_t seg_start = K.seg_start;
    const int64_t lim = (st->end < K.seg_end - 1) ? st->end : K.seg_end - 1;  // last candidate position of this launch
    if (threadIdx.x == 0) {
        mb->seq = 0;
        mb->done = 0;
        mb->quit = 0;
        mb->scout_seq = 0;
        mb->scout_pos = st->p;
        mb->scout_clean = st->clean_ptr;
    }
    for (int i = threadIdx.x; i < MRZ_FILTER_SIZE; i += MRZ_SEQ_THREADS) batch.filter[i] = 0;
    __syncthreads();
    if (MRZ_HAVE_SCOUT && wave == MRZ_SEQ_WAVES - 1) {
        mrz_scout_args S;
        S.buf = a.buf;
        S.tab = a.tab;
        S.K = K;
        S.lim = lim;
        S.slot_mask = st->slot_mask;
        S.nslots = st->slot_mask + 1;
        mrz_scout_loop(S, mb, lane);
        return;
    }
    if (wave != 0) {
        mrz_helper_loop(a.buf, mb, wave, lane);
        return;
    }

    mrz_cfg C;
    C.buf = a.buf;
    C.tab = a.tab;
    C.events = a.events;
    C.st = st;
    C.end = st->end;
    C.limit = st->limit;
    C.max_chain = st->max_chain;
    C.slot_mask = st->slot_mask;
    C.nslots = st->slot_mask + 1;
    C.event_cap = st->event_cap;
    C.gmb = (mrz_gmailbox *)a.gmailbox;
    unsigned long long gseq = 0;
    C.gseq = &gseq;
    int gnw = 0;
    C.gnw = &gnw;
    C.n_helpers = a.n_helpers;
    int64_t farm_hint = 0;
    C.farm_hint = &farm_hint;
    int long_seen = 0;
    C.long_seen = &long_seen;
    int mb_seq = 0;
    C.mb = mb;
    C.mb_seq = &mb_seq;
    mrz_lead L;
    L.p = st->p;
    L.cur_p = st->cur_p;
    L.cur_ofs = st->cur_ofs;
    L.cur_len = st->cur_len;
    L.last_match = st->last_match;
    L.min_mask = st->min_mask;
    L.tag_mask = st->tag_mask;
    L.count = st->count;
    L.clean_ptr = st->clean_ptr;
    L.victim_round = st->victim_round;
    L.n_events = st->n_events;
    L.inserts = st->inserts;
    L.tag_hits = st->tag_hits;
    L.tag_misses = st->tag_misses;
    L.last_len = 0;
    L.mbytes = 0;

    const int64_t hint_p0 = L.p, hint_ev0 = L.n_events;

    unsigned epoch = 1;
    bool ok = true;
    int width = 64;            // batch width, adapted to how many lanes recent batches could commit
    int low_yield = 0;         // consecutive batches that committed <= 2 candidates
    int seq_credit = 0;        // candidates to run through the cooperative path before batching again
    // Right after an emission the next candidate often has long matches again (repetitive input): a batch
    // would be formed, walked and probed only to be cut at its first lane.  Four saturating counters, indexed by
    // the classes (short / >= GREAT_MATCH) of the last two emitted matches, learn whether that is so; when it
    // is, the first candidate after an emission goes straight through the cooperative path.
    bool after_emit = false;
    int emit_cls = 0;  // bit 0: last emitted match was great, bit 1: the one before
    int pred_long[4] = { 0, 0, 0, 0 };
#ifdef MRZ_SEQ_STATS
    int64_t stat[MRZ_ST_N];
    for (int k = 0; k < MRZ_ST_N; k++) stat[k] = 0;
#else
    int64_t *stat = nullptr;
#endif

    int sc_seq = 0;
    int64_t sc_last = -1;
    // The cursor into the candidate list: entry ci is the first one behind position cur_from - 1, as long as the
    // matcher's position moves from candidate to candidate; after an emission (the position jumps, forward or back)
    // it is looked up again.
    int64_t ci = 0, ci_pos = -1;  // ci is valid for L.p == ci_pos
    while (ok) {
        PROF_T0();
        if (MRZ_HAVE_SCOUT && L.p != sc_last) {
            sc_last = L.p;
            sc_seq++;
            if (lane == 0) {
                mb->scout_pos = L.p;
                mb->scout_clean = L.clean_ptr;
                mrz_mb_store(&mb->scout_seq, sc_seq);
            }
        }
        // ---- the next candidates behind position p ------------------------------------
        int64_t pos = L.p + 1;
        if (pos < seg_start) pos = seg_start;
        if (pos > lim) break;
        if (L.p != ci_pos) {
            ci = mrz_cand_lower_bound(K, pos, lane);
            ci_pos = L.p;
        }
        // the first one: still a candidate under the mask reached by now (src/rzip.c:573)?  Entries that are not are
        // skipped 64 at a time.
        int64_t q0 = 0, t0 = 0;
        bool more = false;
        while (true) {
            int64_t q = 0, t = 0;
            bool in = ci + lane < K.n;
            if (in) {
                const mrz_cand c = K.cand[ci + lane];
                q = c.off;
                t = c.t;
                in = q <= lim;
            }
            const mrz_u64 m_in = __ballot(in);
            const mrz_u64 m_pass = __ballot(in && (t & L.min_mask) == L.min_mask);
            if (m_pass) {
                const int fl = __ffsll((long long)m_pass) - 1;
                ci += fl;
                q0 = mrz_bcast64(q, fl);
                t0 = mrz_bcast64(t, fl);
                more = true;
                break;
            }
            if (m_in != ~0ull) break;  // the list (or the launch's range) ends here
            ci += 64;
        }
        if (!more) {
            L.p = lim;
            break;
        }
        PROF_ADD(MRZ_ST_T_WINDOW);
        int used = 0;
        const int64_t ev_before = L.n_events;
        const bool first_after_emit = after_emit;
        const bool go_seq = first_after_emit && pred_long[emit_cls] >= 2;
        long_seen = 0;
#ifndef MRZ_NO_BATCH
        if (seq_credit > 0)
            seq_credit--;  // a stretch where every candidate has long matches: one at a time is cheaper
        else if (!go_seq) {
            used = mrz_batch_step(C, L, &batch, K, ci, lim, epoch, width, lane, &ok, stat);
            epoch++;
            // adapt the width: shrink towards what could be committed, grow back when all of it was
            if (used >= width)
                width = width * 2 > 64 ? 64 : width * 2;
            else {
                // dense stretches (many lanes committed) regrow fast: a conflict cut there says little about the next batch
                const int want = (used >= MRZ_WIDTH_DENSE ? MRZ_WIDTH_MULT_DENSE : MRZ_WIDTH_MULT) * used + 4;
                width = want < 8 ? 8 : (want > 64 ? 64 : want);
            }
            if (used <= 2) {
                if (++low_yield >= MRZ_LOW_YIELD_RUNS) {
                    seq_credit = MRZ_SEQ_CREDIT;
                    low_yield = 0;
                }
            } else
                low_yield = 0;
        }
#endif
#ifdef MRZ_SEQ_PROFILE
        prof_t0 = (int64_t)__builtin_amdgcn_s_memtime();  // the batch booked its own time
#endif
        if (used == 0 && ok) {
            ST_ADD(MRZ_ST_SEQ, 1);
            // the first candidate through the cooperative path
            L.p = q0;
            ok = mrz_seq_candidate(C, L, &batch.coop, t0, lane, stat);
            used = 1;
            PROF_ADD(MRZ_ST_T_SEQ);
        }
        // the cursor follows as long as the position is the last consumed candidate's
        if (L.n_events == ev_before) {
            ci += used;
            ci_pos = L.p;
        } else
            ci_pos = -2;  // (the position has jumped: the cursor is looked up again)
        if (first_after_emit) {
            const bool first_was_long = long_seen && used <= 1;
            int &c = pred_long[emit_cls];
            c = first_was_long ? (c < 3 ? c + 1 : 3) : (c > 0 ? c - 1 : 0);
        }
        after_emit = L.n_events != ev_before;
        if (after_emit) emit_cls = ((emit_cls << 1) & 2) | (L.last_len >= MRZ_GREAT_MATCH ? 1 : 0);
    }

    // release the helpers, then publish the state for the next segment's launch
#if MRZ_HELPER_WGS > 0
    if (lane == 0 && C.gmb) mrz_g_storeu(&C.gmb->quit, 1ull);
#endif
    if (lane == 0) {
        mrz_mb_store(&mb->quit, 1);
        mrz_mb_store(&mb->seq, mb_seq + 1);
        st->p = L.p;
        st->cur_p = L.cur_p;
        st->cur_ofs = L.cur_ofs;
        st->cur_len = L.cur_len;
        st->last_match = L.last_match;
        st->min_mask = L.min_mask;
        st->tag_mask = L.tag_mask;
        st->count = L.count;
        st->clean_ptr = L.clean_ptr;
        st->victim_round = L.victim_round;
        st->n_events = L.n_events;
        st->inserts = L.inserts;
        st->tag_hits = L.tag_hits;
        st->tag_misses = L.tag_misses;
        st->finished = L.p >= C.end ? 1 : 0;
        st->hint_positions = L.p - hint_p0;
        st->hint_events = L.n_events - hint_ev0;
        st->hint_matched = L.mbytes;
#ifdef MRZ_SEQ_STATS
        for (int k = 0; k < MRZ_ST_N; k++) st->prof[k] += stat[k];
#endif
    }
}

extern "C" hipError_t mrz_launch_sequencer_narrow(hipStream_t stream, const uint8_t *buf, mrz_slot *tab, const mrz_cand *cand,
                                                  const int *tile_off, const mrz_u64 *bitmap, mrz_event *events,
                                                  mrz_seq_state *st, void *gmailbox, int n_helpers, int xcd) {
    mrz_seq_args a;
    a.buf = buf;
    a.tab = tab;
    a.cand = cand;
    a.tile_off = tile_off;
    a.bitmap = bitmap;
    a.events = events;
    a.st = st;
    a.gmailbox = gmailbox;
    a.xcd = xcd & 7;
    a.deep_bits = 0;
#if MRZ_HELPER_WGS == 0
    n_helpers = 0;
#endif
    if (n_helpers > MRZ_HELPER_WGS) n_helpers = MRZ_HELPER_WGS;
    if (n_helpers < 0 || !gmailbox) n_helpers = 0;
    a.n_helpers = n_helpers;
    if (gmailbox) {
        hipError_t e = hipMemsetAsync(gmailbox, 0, sizeof(mrz_gmailbox), stream);
        if (e != hipSuccess) return e;
    }
    unsigned grid = (unsigned)(1 + a.n_helpers);
    if (grid < (unsigned)(a.xcd + 1)) grid = (unsigned)(a.xcd + 1);
    hipLaunchKernelGGL(mrz_seq_narrow_kernel, dim3(grid), dim3(MRZ_SEQ_THREADS), 0, stream, a);
    return hipGetLastError();
}

extern "C" size_t mrz_seq_narrow_mailbox_size(void) { return MRZ_HELPER_WGS > 0 ? sizeof(mrz_gmailbox) : 0; }
