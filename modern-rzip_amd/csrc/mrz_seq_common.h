// mrz_seq_common.h -- pieces shared by the two sequencer kernels (the wide engine of mrz_sequencer.hip /
// mrz_seq_wide.h and the narrow engine of mrz_seq_narrow.hip): the candidate list's lower bound, the wave-wide
// compares of single_match_len, the compare farm (helper workgroups on the other CUs), the matcher state, and the
// COOPERATIVE PATH: one candidate at a time, the whole of wave 0 on its probe chain (any chain length, any match
// length, cascades, chain-limit evictions, sweep wrap, mask promotion).  Both batch engines fall back to it for
// whatever they cannot prove.  The narrow engine defines MRZ_NARROW_ENGINE (and its own MRZ_SEQ_WAVES) before
// including this file: its workgroup stripes long forward extensions over a second wave through an LDS mailbox.
#pragma once
#include "mrz_device.h"
#include <stdlib.h>

// cross-lane LDS exchange inside one wave: the hardware runs the lanes in lockstep, the CPU
// emulator needs a rendezvous
#ifdef __HIP_DEVICE_COMPILE__
#define MRZ_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define MRZ_WAVE_SYNC() (void)__ballot(1)
#endif

#ifndef MRZ_WIDE_WAVES
#define MRZ_WIDE_WAVES 8  // waves of the wide engine's workgroup = 64-lane slices of a wide batch
#endif                    // (the CPU emulator build takes 2: 512 fibers per barrier are slow, the logic is the same)
#ifndef MRZ_SEQ_WAVES
#define MRZ_SEQ_WAVES MRZ_WIDE_WAVES
#endif
#define MRZ_SEQ_THREADS (64 * MRZ_SEQ_WAVES)
// waves of the sequencer workgroup a long forward extension is striped over (narrow engine: leader + stripe helper)
#ifdef MRZ_NARROW_ENGINE
#define MRZ_STRIPE_WAVES (MRZ_SEQ_WAVES > 2 ? MRZ_SEQ_WAVES - 1 : MRZ_SEQ_WAVES)
#else
#define MRZ_STRIPE_WAVES 1
#endif
#ifndef MRZ_HELPER_WAVES
#define MRZ_HELPER_WAVES 3  // waves of a helper workgroup that compare (the others leave at once)
#endif
#define MRZ_CASCADE_MAX 64
#ifndef MRZ_SEQ_CREDIT
#define MRZ_SEQ_CREDIT 8       // candidates sent through the cooperative path after repeated tiny batches
#endif
#ifndef MRZ_LOW_YIELD_RUNS
#define MRZ_LOW_YIELD_RUNS 2
#endif

// optional in-kernel cycle accounting (diagnostic builds only: -DMRZ_SEQ_PROFILE)
#ifdef MRZ_SEQ_PROFILE
#define PROF_T0() int64_t prof_t0 = (int64_t)__builtin_amdgcn_s_memtime()
#define PROF_T0R() prof_t0 = (int64_t)__builtin_amdgcn_s_memtime()
#define PROF_ADD(k)                                                      \
    do {                                                                 \
        const int64_t now__ = (int64_t)__builtin_amdgcn_s_memtime();     \
        stat[k] += now__ - prof_t0;                                      \
        prof_t0 = now__;                                                 \
    } while (0)
#else
#define PROF_T0()
#define PROF_T0R()
#define PROF_ADD(k)
#endif

// diagnostics kept in mrz_seq_state.prof: counted only in -DMRZ_SEQ_STATS / -DMRZ_SEQ_PROFILE builds (the array is
// indexed dynamically, so it lives in scratch memory: not something to pay for in the product build)
#if defined(MRZ_SEQ_PROFILE) && !defined(MRZ_SEQ_STATS)
#define MRZ_SEQ_STATS 1
#endif
#ifdef MRZ_SEQ_STATS
#define ST_ADD(k, v) (stat[k] += (v))
#else
#define ST_ADD(k, v) ((void)0)
#endif
// the wide engine's write log: one batch stamp per 2^MRZ_WLOG_SHIFT table slots (4 slots = one 64-byte line)
#ifndef MRZ_WLOG_SHIFT
#define MRZ_WLOG_SHIFT 2
#endif
enum { MRZ_ST_BATCHES, MRZ_ST_FORMED, MRZ_ST_COMMITTED, MRZ_ST_SEGMENTS, MRZ_ST_EMITS, MRZ_ST_BACKJUMP, MRZ_ST_REWALK,
       MRZ_ST_LONGRES, MRZ_ST_SEQ, MRZ_ST_CUT_CPLX, MRZ_ST_CUT_OVERFLOW, MRZ_ST_SKIPOUT, MRZ_ST_CONF0, MRZ_ST_PAIRS,
       MRZ_ST_T_FORM, MRZ_ST_T_WALK, MRZ_ST_T_CONF, MRZ_ST_T_PAIRS, MRZ_ST_T_LOOP, MRZ_ST_T_REWALK, MRZ_ST_T_LONG,
       MRZ_ST_T_SEQ, MRZ_ST_FARMED, MRZ_ST_L_POST, MRZ_ST_L_STRIPE, MRZ_ST_L_BWD, MRZ_ST_L_WAIT, MRZ_ST_L_ROUNDS,
       MRZ_ST_F_POST, MRZ_ST_F_WAIT, MRZ_ST_F_FOLD, MRZ_ST_S_TAB, MRZ_ST_S_PAIR, MRZ_ST_S_INS, MRZ_ST_OVL, MRZ_ST_OVL_OK,
       MRZ_ST_X_WALK, MRZ_ST_X_CASC, MRZ_ST_X_POOL, MRZ_ST_X_WIN, MRZ_ST_X_SAME, MRZ_ST_C_WIN, MRZ_ST_C_EVICT, MRZ_ST_C_DEEP,
       MRZ_ST_C_MANY, MRZ_ST_C_FAIL, MRZ_ST_C_TIE, MRZ_ST_C_NW, MRZ_ST_T_OVL, MRZ_ST_H_PRE, MRZ_ST_H_CAND, MRZ_ST_H_POST,
       MRZ_ST_T_SCAN, MRZ_ST_T_FOLD, MRZ_ST_T_COMMIT, MRZ_ST_REPREP, MRZ_ST_W_STALE, MRZ_ST_W_DROP, MRZ_ST_RESET,
       MRZ_ST_T_TURN, MRZ_ST_T_PREP, MRZ_ST_T_PRECOMMIT, MRZ_ST_E_MASK, MRZ_ST_E_CULL, MRZ_ST_E_XW, MRZ_ST_E_INWIN,
       MRZ_ST_E_WINDOW, MRZ_ST_E_BULK, MRZ_ST_E_MORE, MRZ_ST_T_PC_CW, MRZ_ST_T_PC_LOG, MRZ_ST_T_PC_BEST, MRZ_ST_T_PC_BULK, MRZ_ST_T_TURNWORK, MRZ_ST_T_SNAP, MRZ_ST_REBULK,
       // (the narrow engine's own)
       MRZ_ST_BATCH_LANES, MRZ_ST_CUT_LONG, MRZ_ST_CUT_WALK, MRZ_ST_CUT_CONFLICT, MRZ_ST_CUT_CULL, MRZ_ST_BATCH_EMITS,
       MRZ_ST_CUT_CASCADE, MRZ_ST_BATCH_FORMED, MRZ_ST_T_WALK2, MRZ_ST_T_SCANS, MRZ_ST_T_CONFLICT, MRZ_ST_T_WINDOW,
       // (the deep engine's own)
       MRZ_ST_D_BATCHES, MRZ_ST_D_LANES, MRZ_ST_D_ROUNDS, MRZ_ST_D_RESCANNED, MRZ_ST_D_COOP, MRZ_ST_D_T_FORM, MRZ_ST_D_T_SCAN,
       MRZ_ST_D_T_COMMIT, MRZ_ST_D_T_RESCAN, MRZ_ST_D_T_TOTAL, MRZ_ST_D_LAUNCHES, MRZ_ST_D_T_COOP, MRZ_ST_D_COOP_REC,
       MRZ_ST_D_RESOLVED, MRZ_ST_D_S_COOP, MRZ_ST_D_S_CONFLICT, MRZ_ST_D_S_CULLED, MRZ_ST_D_S_NOCULL, MRZ_ST_D_S_STALE, MRZ_ST_D_C_OVER_ALT, MRZ_ST_D_C_OVER_NOALT, MRZ_ST_D_C_EMPTY, MRZ_ST_D_C_DISPLACE, MRZ_ST_D_C_OTHER,
       MRZ_ST_N };
static_assert(MRZ_ST_N <= (int)(sizeof(((mrz_seq_state *)0)->prof) / sizeof(int64_t)), "mrz_seq_state.prof holds the counters");

struct mrz_seq_args {
    const uint8_t *buf;
    mrz_slot *tab;
    const mrz_cand *cand;     // the front end's candidate list of this segment: {position, tag}, position order
    const int *tile_off;      // list offset of every 4096-position tile of the segment
    const mrz_u64 *bitmap;    // pass bits of the segment, 64 positions per word (bit 0 of word 0 = st->seg_start)
    mrz_event *events;
    mrz_seq_state *st;        // carries the segment's geometry: seg_start, seg_end, n_cand (mrz_tagscan.hip)
    void *gmailbox;           // mrz_gmailbox in device memory, zeroed by the host before every launch
    int n_helpers;            // helper workgroups in this launch
    int xcd;                  // block index mod 8 of the sequencer workgroups
    int deep_bits;            // wide engine: a launch ends where minimum_tag_mask reaches this many bits (the deep engine's
                              // regime: the host relaunches the rest of the segment there); 0 = never
};

// the candidate list of one segment, as the sequencers see it
struct mrz_cands {
    const mrz_cand *cand;
    const int *tile_off;
    const mrz_u64 *bitmap;
    int64_t seg_start, seg_end, n;
};

// Index of the first list entry at or behind position `pos` (wave-wide, all lanes; one round trip): the tile's offset
// plus the pass bits of the tile below `pos`.
__device__ static int64_t mrz_cand_lower_bound(const mrz_cands &K, int64_t pos, int lane) {
    if (pos <= K.seg_start) return 0;
    if (pos >= K.seg_end) return K.n;
    const int64_t rel = pos - K.seg_start;
    const int64_t tile = rel >> MRZ_TILE_SHIFT;
    const int r = (int)(rel & (MRZ_TILE - 1));
    const mrz_u64 w = K.bitmap[tile * (MRZ_TILE / 64) + lane];
    const int base = K.tile_off[tile];
    int c = 0;
    if (lane < (r >> 6))
        c = __popcll(w);
    else if (lane == (r >> 6))
        c = __popcll(w & mrz_low_mask(r & 63));
    const int sum = mrz_lane_read(mrz_wave_incl_sum(c, lane), 63);
    return (int64_t)mrz_uni(base) + sum;
}

// workgroup-scope accesses to LDS control words
__device__ __forceinline__ int mrz_mb_load(int *p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void mrz_mb_store(int *p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ void mrz_mb_add(int *p, int v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// LDS mailbox between the narrow engine's leader and its helper waves: one long forward extension at a time, striped
// over the waves of the workgroup; and the scout's hand-off words
struct mrz_mailbox {
    int64_t p0, op, maxf, base;     // compare buf[p0+x] with buf[op+x] for x in [base + wave*STRIPE, +STRIPE), x < maxf
    int64_t res[MRZ_SEQ_WAVES];     // per wave: first stop offset of its stripe, or -1
    int seq;                        // bumped by the leader for every round; helpers wait on it
    int done;                       // helpers add 1 when their stripe is finished
    int quit;
    int scout_seq;                  // bumped whenever scout_pos changes
    int64_t scout_pos;              // the leader's position: the scout warms the caches for what follows
    int64_t scout_clean;            // tag_clean_ptr, for the cull sweep window
};

#ifndef MRZ_STRIPE_PIECES
#define MRZ_STRIPE_PIECES 4
#endif
#define MRZ_STRIPE (MRZ_STRIPE_PIECES * 1024)

// Forward compare of one 4 KiB stripe starting at `base` (64 lanes x 16 B x 4
// pieces, all loads issued before the first compare).  Returns the offset (from
// p0) at which `while (p < end && buf[p] == buf[op])` (src/rzip.c:378) stops if
// that lies inside or before this stripe's reach, else -1.
template <int PIECES>
__device__ static int64_t mrz_wave_fwd_stripe_n(const uint8_t *__restrict__ buf, int64_t p0, int64_t op, int64_t maxf,
                                                int64_t base, int lane) {
    uint4 a[PIECES], b[PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; j++) {
        const int64_t off = base + j * 1024 + lane * 16;
        if (off < maxf) {
            a[j] = mrz_ld16(buf + p0 + off);
            b[j] = mrz_ld16(buf + op + off);
        }
    }
    int64_t found = -1;
#pragma unroll
    for (int j = 0; j < PIECES; j++) {
        if (found >= 0) continue;
        const int64_t off = base + j * 1024 + lane * 16;
        int lane_len = 0;
        bool full = false;
        if (off < maxf) {
            const int64_t rem = maxf - off;
            const int lim = rem < 16 ? (int)rem : 16;
            const int d = mrz_first_diff16(a[j], b[j]);
            lane_len = d < lim ? d : lim;
            full = lane_len == 16;
        }
        const mrz_u64 stop = __ballot(!full);
        if (stop) {
            const int fl = __ffsll((long long)stop) - 1;
            found = base + j * 1024 + (int64_t)fl * 16 + mrz_lane_read(lane_len, fl);
        }
    }
    return found;
}

__device__ __forceinline__ int64_t mrz_wave_fwd_stripe(const uint8_t *__restrict__ buf, int64_t p0, int64_t op,
                                                       int64_t maxf, int64_t base, int lane) {
    return mrz_wave_fwd_stripe_n<MRZ_STRIPE_PIECES>(buf, p0, op, maxf, base, lane);
}

// Backward half of single_match_len (src/rzip.c:386-391), wave-wide.
__device__ static int64_t mrz_wave_bwd(const uint8_t *__restrict__ buf, int64_t p0, int64_t op, int64_t maxb,
                                       int lane) {
    if (maxb <= 0) return 0;
    for (int64_t base = 0;; base += 1024) {
        const int64_t off = base + lane * 16;
        int lane_len = 0;
        bool full = false;
        if (off < maxb) {
            const int64_t rem = maxb - off;
            const int lim = rem < 16 ? (int)rem : 16;
            int cnt;
            if (op - off - 16 >= 0) {
                cnt = mrz_top_equal16(mrz_ld16(buf + p0 - off - 16), mrz_ld16(buf + op - off - 16));
            } else {
                cnt = 0;
                while (cnt < lim && buf[p0 - off - 1 - cnt] == buf[op - off - 1 - cnt]) cnt++;
            }
            lane_len = cnt < lim ? cnt : lim;
            full = lane_len == 16;
        }
        const mrz_u64 stop = __ballot(!full);
        if (stop) {
            const int fl = __ffsll((long long)stop) - 1;
            return base + (int64_t)fl * 16 + mrz_lane_read(lane_len, fl);
        }
    }
}

// Long candidate: forward in 4 KiB rounds per wave (64 lanes x 16 B x 4 pieces, ballot + ffs for the first mismatch),
// backward once.  In the narrow engine's workgroup (mb != nullptr) the forward extension is striped over
// MRZ_STRIPE_WAVES waves through the LDS mailbox: the leader folds the per-wave results and does the backward
// extension while the helper is busy with the first round.  With `cont_base` the caller takes over after the first
// round that finds no difference (the compare farm continues from there): returns -1.
__device__ static int64_t mrz_long_match_len(const uint8_t *__restrict__ buf, mrz_mailbox *mb, int *mb_seq, int64_t p0,
                                             int64_t op, int64_t end, int64_t last_match, int64_t *rev_out, int lane,
                                             int64_t *stat = nullptr, int64_t *cont_base = nullptr) {
    *rev_out = 0;
#ifdef MRZ_SEQ_PROFILE
    int64_t lt0 = (int64_t)__builtin_amdgcn_s_memtime();
#define LPROF(k)                                                         \
    do {                                                                 \
        if (stat) {                                                      \
            const int64_t now__ = (int64_t)__builtin_amdgcn_s_memtime(); \
            stat[k] += now__ - lt0;                                      \
            lt0 = now__;                                                 \
        }                                                                \
    } while (0)
#else
#define LPROF(k)
#endif
    if (op >= p0) return 0;
    const int64_t maxf = end - p0;
    const int64_t floor_p = last_match > 0 ? last_match : 0;
    int64_t maxb = p0 - floor_p;
    if (op < maxb) maxb = op;
    int64_t fwd = 0, rev = 0;
    bool have_rev = false;
    const bool striped = MRZ_STRIPE_WAVES > 1 && mb != nullptr;
    const int64_t round_bytes = (int64_t)(striped ? MRZ_STRIPE_WAVES : 1) * MRZ_STRIPE;
    if (maxf > 0) {
        for (int64_t base = 0;; base += round_bytes) {
            if (striped) {
                if (lane == 0) {
                    mb->p0 = p0;
                    mb->op = op;
                    mb->maxf = maxf;
                    mb->base = base;
                    mb->done = 0;
                }
                *mb_seq += 1;
                if (lane == 0) mrz_mb_store(&mb->seq, *mb_seq);
            }
            LPROF(MRZ_ST_L_POST);
            int64_t best = mrz_wave_fwd_stripe(buf, p0, op, maxf, base, lane);  // the leader's own stripe (wave 0)
            LPROF(MRZ_ST_L_STRIPE);
            if (!have_rev) {
                rev = mrz_wave_bwd(buf, p0, op, maxb, lane);
                have_rev = true;
            }
            LPROF(MRZ_ST_L_BWD);
            if (striped) {
                while (mrz_uni(mrz_mb_load(&mb->done)) < MRZ_STRIPE_WAVES - 1) __builtin_amdgcn_s_sleep(1);
                for (int w = 1; w < MRZ_STRIPE_WAVES && best < 0; w++) best = mrz_uni64(mb->res[w]);
            }
            LPROF(MRZ_ST_L_WAIT);
#ifdef MRZ_SEQ_STATS
            if (stat) stat[MRZ_ST_L_ROUNDS] += 1;
#endif
            if (best >= 0) {
                fwd = best;
                break;
            }
            if (cont_base) {  // the caller continues from here (compare farm); returns -1
                *cont_base = base + round_bytes;
                *rev_out = rev;
                return -1;
            }
        }
    }
    if (!have_rev) rev = mrz_wave_bwd(buf, p0, op, maxb, lane);
    *rev_out = rev;
    const int64_t len = fwd + rev;
    return len < MRZ_MIN_MATCH ? 0 : len;
}

// One 64-slot step of insert_hash's probe walk (src/rzip.c:264-297) over the
// slots already loaded into `e`.  round / victim_h carry across steps.
// Returns true when the walk stops in this step; then *stop_slot is the slot to
// write and *kind says why: 0 empty, 1 overwrite (due for culling / chain limit:
// hash_count was decremented), 2 displace (occupant must be re-inserted first).
__device__ __forceinline__ bool mrz_insert_step(const mrz_slot e, bool empty, int64_t t, int my_rank, int64_t slot0,
                                                int64_t slot_mask, int64_t better, int64_t max_chain,
                                                int64_t *round, int64_t *victim_h, int64_t *count,
                                                int64_t *victim_round, int64_t *stop_slot, int *kind, int64_t *occ_t,
                                                int64_t *occ_off) {
    const bool minbit = !empty && ((e.t & better) != better);
    const bool lesser = !empty && (mrz_ones_rank(e.t) < my_rank);
    const bool same = !empty && (e.t == t);
    const mrz_u64 m_stop = __ballot(empty || minbit || lesser);
    const int first_stop = m_stop ? __ffsll((long long)m_stop) - 1 : MRZ_WAVE;
    const mrz_u64 m_same = __ballot(same) & mrz_low_mask(first_stop);
    const int cnt = __popcll(m_same);
    // victim_h is latched at the same-tag entry whose round == victim_round (:283)
    const int64_t kv = *victim_round - *round;
    if (kv >= 0 && kv < cnt) *victim_h = (slot0 + mrz_nth_set(m_same, (int)kv)) & slot_mask;
    const int64_t k = max_chain - *round;  // this many more same-tag entries trip the limit
    if (k <= cnt) {
        // chain limit reached before any other stop: evict the victim (:284-291)
        *count -= 1;
        int64_t vr = *victim_round + 1;
        if (vr == max_chain) vr = 0;
        *victim_round = vr;
        *stop_slot = *victim_h;
        *kind = 1;
        return true;
    }
    if (first_stop < MRZ_WAVE) {
        *stop_slot = (slot0 + first_stop) & slot_mask;
        const int64_t et = mrz_bcast64(e.t, first_stop);
        const int64_t eo = mrz_bcast64(e.off, first_stop);
        if ((eo | et) == 0)
            *kind = 0;  // empty slot
        else if ((et & better) != better) {
            *count -= 1;  // due for culling: overwrite (:267-270)
            *kind = 1;
        } else {
            *kind = 2;  // outranked occupant (:275-278)
            *occ_t = et;
            *occ_off = eo;
        }
        return true;
    }
    *round += cnt;
    return false;
}

// ---- the compare farm: helper workgroups on the other CUs ---------------------------------
// A look-up on repetitive input can find max_chain_len tag-equal entries that are ALL tens of
// KiB long (every earlier copy of the same text): megabytes to compare for one candidate.  One
// CU keeps only ~8 KiB of loads in flight (~20 GB/s on cold data), so the compares are spread
// over the whole chip: the grid carries helper workgroups (MRZ_HELPER_WAVES comparing waves each, about one per CU)
// that wait on a mailbox in device memory.  A round compares, for every pending
// entry, G consecutive stripes: the pending entries are compacted into 2^c columns, helper
// ticket w takes stripe w >> c of column w & (2^c - 1) (2 KiB per wave; from the second round on
// 8-32 KiB per wave, 2-8 KiB per step) and reports where the compare stops inside its stripe,
// or "equal throughout"; the row of helpers after the forward rows measures the backward
// halves.  A single entry that is still equal after the first round gets all helpers.
//
// Hand-off protocol.  Every mailbox word carries the round number in its top 24 bits and the
// payload (an offset < 2^40) in the low 40, is written with ONE agent-scope (sc1) atomic store
// and read with an agent-scope atomic load, so no ordering between words is needed: the reader
// polls until every word it needs shows the round it waits for.  The leader writes the whole
// job descriptor with one store instruction (24 lanes), a helper fetches it with one load
// instruction; results come back the same way, one word per helper.  The mailbox is zeroed by
// the host before every launch and a launch runs fewer than 2^24 rounds, so a tag never
// repeats.  Helpers take a ticket when they start; the leader only addresses tickets it has
// seen, so the scheme does not depend on every workgroup of the grid being resident; if an
// answer does not arrive in ~0.5 s the leader gives the farm up for the launch and compares
// locally.  The compared bytes themselves are read-only input.  All spins are bounded.
#ifndef MRZ_HELPER_WGS
#define MRZ_HELPER_WGS 240  // most of the 256 CUs; the launcher may ask for fewer
#endif
#ifndef MRZ_HELPERS_PER_CU
#define MRZ_HELPERS_PER_CU 1
#endif
#define MRZ_FARM_ENTRIES 16
#define MRZ_FARM_WATCH ((MRZ_HELPER_WGS + 63) / 64)  // result words a leader lane watches
#ifndef MRZ_FARM_WAVE_BYTES
#define MRZ_FARM_WAVE_BYTES 2048
#endif
#ifndef MRZ_FARM_HELPER_SLEEP
#define MRZ_FARM_HELPER_SLEEP 2
#endif
#ifndef MRZ_FARM_LEADER_SLEEP
#define MRZ_FARM_LEADER_SLEEP 1
#endif
#ifndef MRZ_FARM_HINT_MIN
#define MRZ_FARM_HINT_MIN 8192  // single long entry: farm first when the last long match reached this far
#endif
#define MRZ_FARM_SPW (MRZ_HELPER_WAVES * MRZ_FARM_WAVE_BYTES)  // bytes of each stream per helper and round
#define MRZ_FARM_GMAX 255  // rows fit the 8-bit field of the job word
#ifndef MRZ_FARM_ROWS0
#define MRZ_FARM_ROWS0 14  // forward rows of a first round
#endif
#ifndef MRZ_FARM_BULK_MULT
#define MRZ_FARM_BULK_MULT 16  // 2 KiB sub-stripes per wave from the third round on (32 KiB per wave)
#endif
#define MRZ_FARM_SHIFT 40
#define MRZ_FARM_PAYLOAD ((1ull << MRZ_FARM_SHIFT) - 1)
#define MRZ_FARM_NONE MRZ_FARM_PAYLOAD
#define MRZ_SPIN_LIMIT (1 << 20)          // leader: ~0.5 s of polling for an answer that takes microseconds; then
                                           // the farm is given up for this launch and the compare is done locally
#define MRZ_HELPER_SPIN_LIMIT (1ll << 34)  // helpers: idle for as long as a launch may run

struct mrz_gmailbox {
    unsigned long long quit;   // set by the leader when the launch is over
    unsigned long long ready;  // ticket counter: helpers that have started
    unsigned long long pad0[14];
    // job descriptor: 0 p0, 1 maxf, 2 backward floor, 3 base, 4 nsx | G << 8 | want_rev << 16, 5-7 spare,
    // 8.. entry offsets (an offset >= p0 means "not pending")
    unsigned long long words[8 + MRZ_FARM_ENTRIES];
    unsigned long long pad1[8];
    unsigned long long rev[MRZ_FARM_ENTRIES];                          // backward length per entry
    unsigned long long res[MRZ_HELPER_WGS > 0 ? MRZ_HELPER_WGS : 1];   // forward stop per helper, or NONE
    long long dbg[5][MRZ_FARM_ENTRIES];                                // helper phase times (profile builds)
};

__device__ __forceinline__ unsigned long long mrz_g_loadu(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mrz_g_storeu(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#if MRZ_HELPER_WGS > 0
__device__ static void mrz_helper_wg(const uint8_t *__restrict__ buf, mrz_gmailbox *g) {
    __shared__ unsigned long long s_job[2][26];  // double-buffered by round parity: no barrier after reading it
    __shared__ unsigned long long s_min;         // lowest stop offset over the waves of this round
    __shared__ unsigned s_cnt;                   // waves that have contributed
    const int lane = threadIdx.x & 63;
    const int wave = mrz_uni((int)(threadIdx.x >> 6));
    if (wave >= MRZ_HELPER_WAVES) return;  // the grid's workgroups have the sequencer's size; a helper uses 3 waves
    if (threadIdx.x == 0) {
        s_job[0][25] = __hip_atomic_fetch_add(&g->ready, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_min = MRZ_FARM_NONE;
        s_cnt = 0;
    }
    __syncthreads();
    const int me = (int)s_job[0][25];  // my ticket
    if (me >= MRZ_HELPER_WGS) return;
    unsigned long long seen = 0;
    int par = 0;
    while (true) {
        par ^= 1;
        if (wave == 0) {
            long long spins = 0;
            while (true) {
                unsigned long long w = 0;
                if (lane < 8 + MRZ_FARM_ENTRIES)
                    w = mrz_g_loadu(&g->words[lane]);
                else if (lane == 8 + MRZ_FARM_ENTRIES)
                    w = mrz_g_loadu(&g->quit);
                const unsigned long long tag = w >> MRZ_FARM_SHIFT;
                const unsigned long long tag0 = (unsigned long long)mrz_bcast64((int64_t)tag, 0);
                const bool quit = mrz_bcast64((int64_t)w, 8 + MRZ_FARM_ENTRIES) != 0;
                const bool fresh = tag0 != seen && __ballot(lane < 8 + MRZ_FARM_ENTRIES && tag != tag0) == 0;
                const bool giveup = quit || spins++ >= MRZ_HELPER_SPIN_LIMIT;
                if (fresh || giveup) {
                    if (lane < 8 + MRZ_FARM_ENTRIES) s_job[par][lane] = w & MRZ_FARM_PAYLOAD;
                    if (lane == 8 + MRZ_FARM_ENTRIES) s_job[par][lane] = giveup ? ~0ull : tag0;
                    break;
                }
                __builtin_amdgcn_s_sleep(MRZ_FARM_HELPER_SLEEP);
            }
        }
        __syncthreads();
        const unsigned long long tag = s_job[par][8 + MRZ_FARM_ENTRIES];
        if (tag == ~0ull) return;
        seen = tag;
        // cfg: columns (log2) | forward rows << 8 | want_rev << 16 | sub-stripes per helper << 24
        const int cfg = (int)s_job[par][4];
        const int lgc = cfg & 0xff, G = (cfg >> 8) & 0xff, mult = (cfg >> 24) & 0xff;
        const bool want_rev = (cfg >> 16) & 1;
        const int e = me & ((1 << lgc) - 1), s = me >> lgc;  // column = slot of a pending entry, stripe row
        const bool bwd_job = want_rev && s == G;              // the row after the forward rows goes backward
        if (s >= G && !bwd_job) continue;
        const int64_t p0 = (int64_t)s_job[par][0], op = (int64_t)s_job[par][8 + e];
        if (op >= p0) continue;
        if (bwd_job) {
            // backward half of single_match_len for entry e (one wave: the room is p0 - last_match, mostly small)
            if (wave == 0) {
                const int64_t floor_p = (int64_t)s_job[par][2];
                int64_t maxb = p0 - floor_p;
                if (op < maxb) maxb = op;
                const int64_t rev = mrz_wave_bwd(buf, p0, op, maxb, lane);
                if (lane == 0) mrz_g_storeu(&g->rev[e], (tag << MRZ_FARM_SHIFT) | (unsigned long long)rev);
            }
            continue;
        }
        const int64_t maxf = (int64_t)s_job[par][1], base = (int64_t)s_job[par][3];
        // this helper's stripe: each wave mult x 2 KiB of it, 2 KiB (bulk rounds: 8 KiB) at a time until a difference
        int64_t r = -1;
        {
            const int64_t off0 = base + (int64_t)s * mult * MRZ_FARM_SPW + (int64_t)wave * mult * MRZ_FARM_WAVE_BYTES;
            if (mult >= 4) {  // bulk rounds: 8 KiB per step (16 loads of 16 B in flight per lane)
                for (int k = 0; k < mult * MRZ_FARM_WAVE_BYTES && r < 0; k += 8192)
                    r = mrz_wave_fwd_stripe_n<8>(buf, p0, op, maxf, off0 + k, lane);
            } else
                for (int k = 0; k < mult && r < 0; k++)
                    r = mrz_wave_fwd_stripe_n<MRZ_FARM_WAVE_BYTES / 1024>(buf, p0, op, maxf,
                                                                          off0 + (int64_t)k * MRZ_FARM_WAVE_BYTES, lane);
        }
        // the last wave to arrive publishes the workgroup's answer (no barrier)
        if (lane == 0) {
            if (r >= 0) atomicMin(&s_min, (unsigned long long)r);
            if (atomicAdd(&s_cnt, 1u) == MRZ_HELPER_WAVES - 1) {
                const unsigned long long best = atomicExch(&s_min, MRZ_FARM_NONE);
                s_cnt = 0;
                mrz_g_storeu(&g->res[me], (tag << MRZ_FARM_SHIFT) | best);
            }
        }
    }
}
#endif

// ---- definitions shared by the batch engine and the cooperative path ---------------------
#define MRZ_SMAX 16
#ifndef MRZ_WALK_SLOTS
#define MRZ_WALK_SLOTS 8   // slots (16 B each) a lane loads per walk step: one 128-B line when aligned
#endif
#ifndef MRZ_WALK_STEPS
#define MRZ_WALK_STEPS 12
#endif
#ifndef MRZ_WALK_LANE_STEPS
#define MRZ_WALK_LANE_STEPS 2   // per-lane steps before long chains may be finished cooperatively
#endif
#ifndef MRZ_WALK_COOP_MAX
#define MRZ_WALK_COOP_MAX 6     // ... when at most this many lanes are still walking
#endif
#define MRZ_WALK_COOP_STEPS 2   // 64-slot steps per straggler (16 + 128 slots in all)

// LDS staging of the cooperative path: the tag-equal entries of ONE candidate in probe order and their per-lane
// results, the farm's per-entry minimum, and the write-back list of an insert cascade
struct mrz_coop_lds {
    int64_t same_off[MRZ_SMAX];
    int pair_res[MRZ_SMAX];           // (len << 8) | rev, or -1 = runs past the 64-byte reach
    unsigned long long farm_min[16];  // per entry: lowest stop offset any helper reported
    int64_t pend_h[MRZ_CASCADE_MAX], pend_t[MRZ_CASCADE_MAX], pend_o[MRZ_CASCADE_MAX];
    int n_written;      // slots the last candidate's insert wrote: pend_h[0 .. n_written)
    int64_t cull_slot;  // slot its cull emptied, or -1
};

// branch-free versions of mrz_first_diff16 / mrz_top_equal16: this code runs per lane with diverging data, and
// every early return there is an EXEC-mask branch (the branchy form of mrz_lane_match_len was ~740 instructions,
// a third of them control flow)
__device__ __forceinline__ int mrz_first_diff16_bf(uint4 a, uint4 b) {
    const uint32_t d0 = a.x ^ b.x, d1 = a.y ^ b.y, d2 = a.z ^ b.z, d3 = a.w ^ b.w;
    int r = 16;
    r = d3 ? 12 + ((__ffs((int)d3) - 1) >> 3) : r;
    r = d2 ? 8 + ((__ffs((int)d2) - 1) >> 3) : r;
    r = d1 ? 4 + ((__ffs((int)d1) - 1) >> 3) : r;
    r = d0 ? ((__ffs((int)d0) - 1) >> 3) : r;
    return r;
}
__device__ __forceinline__ int mrz_top_equal16_bf(uint4 a, uint4 b) {
    const uint32_t d0 = a.x ^ b.x, d1 = a.y ^ b.y, d2 = a.z ^ b.z, d3 = a.w ^ b.w;
    int r = 16;
    r = d0 ? 12 + (__clz((int)d0) >> 3) : r;
    r = d1 ? 8 + (__clz((int)d1) >> 3) : r;
    r = d2 ? 4 + (__clz((int)d2) >> 3) : r;
    r = d3 ? (__clz((int)d3) >> 3) : r;
    return r;
}

// per-lane forward/backward extension of one candidate (single_match_len, src/rzip.c:372-397), 64 B reach each
// way: all sixteen 16-byte pieces are loaded at once (addresses clamped into the chunk; what a clamped piece
// holds never matters because the counts are capped by maxf / maxb), then evaluated without branches.  A match
// that runs past the reach -- or whose backward part touches the first bytes of the chunk, where a 16-byte piece
// cannot be loaded -- is reported as `is_long` and measured exactly by the cooperative / farm path.
__device__ static void mrz_lane_match_len(const uint8_t *__restrict__ buf, int64_t q, int64_t op, int64_t end,
                                          int64_t last_match, int64_t *len, int64_t *rev, bool *is_long) {
    *len = 0;
    *rev = 0;
    *is_long = false;
    if (op >= q) return;
    int64_t maxf = end - q;
    if (maxf < 0) maxf = 0;
    const int64_t floor_p = last_match > 0 ? last_match : 0;
    int64_t maxb = q - floor_p;
    if (op < maxb) maxb = op;
    if (maxb < 0) maxb = 0;
    const int64_t last_ok = end + (MRZ_MIN_MATCH - 16);  // chunk size - 16: the last 16-byte piece inside the chunk
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
    // forward: equal bytes among the first 64
    int rawf;
    {
        const int d0 = mrz_first_diff16_bf(fa[0], fb[0]), d1 = mrz_first_diff16_bf(fa[1], fb[1]);
        const int d2 = mrz_first_diff16_bf(fa[2], fb[2]), d3 = mrz_first_diff16_bf(fa[3], fb[3]);
        rawf = d0 < 16 ? d0 : 16 + (d1 < 16 ? d1 : 16 + (d2 < 16 ? d2 : 16 + d3));
    }
    const int64_t fwd = rawf < maxf ? rawf : maxf;
    bool lng = rawf == 64 && maxf > 64;
    // backward: pieces at or beyond `edge` would start before byte 0 of the chunk
    const int edge = op < 64 ? (int)(op >> 4) : 4;
    int rawb;
    {
        const int e0 = edge > 0 ? mrz_top_equal16_bf(ba[0], bb[0]) : 16, e1 = edge > 1 ? mrz_top_equal16_bf(ba[1], bb[1]) : 16;
        const int e2 = edge > 2 ? mrz_top_equal16_bf(ba[2], bb[2]) : 16, e3 = edge > 3 ? mrz_top_equal16_bf(ba[3], bb[3]) : 16;
        rawb = e0 < 16 ? e0 : 16 + (e1 < 16 ? e1 : 16 + (e2 < 16 ? e2 : 16 + e3));
    }
    const int64_t rv = rawb < maxb ? rawb : maxb;
    lng = lng || (rawb == 64 && maxb > 64) || (edge < 4 && rawb >= 16 * edge && maxb > 16 * edge);
    *is_long = lng;
    if (lng) return;
    *rev = rv;
    const int64_t l = fwd + rv;
    *len = l >= MRZ_MIN_MATCH ? l : 0;
}

// ---- leader state ----------------------------------------------------------------
struct mrz_lead {  // wave-uniform; what hash_search keeps in locals / rzip_state
    int64_t p, cur_p, cur_ofs, cur_len, last_match;
    int64_t min_mask, tag_mask, count, clean_ptr, victim_round;
    int64_t n_events, inserts, tag_hits, tag_misses;
    int64_t last_len;  // length of the last emitted match (scheduling hint only)
    int64_t mbytes;    // bytes of the matches emitted in this launch (regime hint only)
};

struct mrz_cfg {
    const uint8_t *buf;
    mrz_slot *tab;
    mrz_event *events;
    mrz_seq_state *st;
    int64_t end, limit, max_chain, slot_mask, nslots, event_cap;
    mrz_gmailbox *gmb;
    unsigned long long *gseq;  // leader's copy of the global round counter
    int *gnw;                  // helper tickets the leader has seen so far
    int n_helpers;             // helper workgroups in this launch
    int64_t *farm_hint;        // forward length of the last long match: go to the farm at once when it was big
    int *long_seen;            // set when a look-up had entries beyond the 64-byte reach (scheduling hint only)
    mrz_mailbox *mb;           // narrow engine: LDS mailbox of its stripe helper wave (nullptr in the wide engine)
    int *mb_seq;               // ... and the leader's copy of its round counter
};

// lazy selection + emission (src/rzip.c:586-599) for the candidate at L.p whose
// look-up returned (mlen, m_off, m_rev).  Returns false on event-list overflow.
__device__ __forceinline__ bool mrz_select_emit(const mrz_cfg &C, mrz_lead &L, int64_t mlen, int64_t m_off,
                                                int64_t m_rev, int lane) {
    if (mlen > L.cur_len) {
        L.cur_p = L.p - m_rev;
        L.cur_len = mlen;
        L.cur_ofs = m_off;
    }
    if ((L.cur_len >= MRZ_GREAT_MATCH || L.p >= L.cur_p + MRZ_MIN_MATCH) && L.cur_len >= MRZ_MIN_MATCH) {
        if (L.n_events >= C.event_cap) {  // cannot happen: matches are >= 31 bytes and disjoint
            if (lane == 0) C.st->error = 1;
            return false;
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
    }
    return true;
}

// clean_one_from_hash (src/rzip.c:305-328), 64 slots per sweep step
__device__ static void mrz_cull_one(const mrz_cfg &C, mrz_lead &L, int lane) {
    mrz_slot *tab = C.tab;
    while (true) {
        const int64_t better2 = (L.min_mask << 1) | 1;
        bool culled = false;
        while (L.clean_ptr < C.nslots) {
            const int64_t s = L.clean_ptr + lane;
            mrz_slot e;
            e.off = 0;
            e.t = 0;
            if (s < C.nslots) e = tab[s];
            const bool hit = ((e.off | e.t) != 0) && ((e.t & better2) != better2);
            const mrz_u64 m = __ballot(hit);
            if (m) {
                const int fl = __ffsll((long long)m) - 1;
                L.clean_ptr += fl;
                if (lane == fl) {
                    mrz_slot z;
                    z.off = 0;
                    z.t = 0;
                    tab[s] = z;
                }
                L.count--;
                culled = true;
                break;
            }
            L.clean_ptr += MRZ_WAVE;
        }
        if (culled) {
            L.tag_mask = better2;
            return;
        }
        L.min_mask = better2;
        L.clean_ptr = 0;
    }
}

#if MRZ_HELPER_WGS > 0
// Farm rounds for the pending entries of one look-up at p0.  Lane e (< nsx <= 16) passes its entry in my_op /
// my_pending.  Rounds continue from offset `base` until every pending entry has hit its first difference (or maxf);
// on return lane e holds the forward stop offset (from p0) in *my_fwd and, with want_rev, the backward length in
// *my_rev.  Needs at least one full row of 16 helper tickets (two with want_rev).
__device__ static bool mrz_farm(const mrz_cfg &C, mrz_coop_lds *B, int64_t p0, int64_t maxf, int64_t floor_p,
                                int64_t base, int nsx, int64_t my_op, bool my_pending, bool want_rev, int lane,
                                int64_t *stat, int64_t *my_fwd, int64_t *my_rev) {
    mrz_gmailbox *g = C.gmb;
#ifdef MRZ_SEQ_PROFILE
    int64_t prof_t0 = (int64_t)__builtin_amdgcn_s_memtime();
#endif
    mrz_u64 pending = __ballot(my_pending && lane < nsx);
    int64_t fwd = 0, rev = 0;
    int rounds = 0;
    while (pending) {
        // The pending entries are compacted into the first `np` columns; helper ticket w works on column
        // w & (ncols - 1), stripe row w >> lgc.  Rows 0..G-1 go forward, row G backward (first round only).
        // From the second round on every wave takes several 2 KiB sub-stripes: entries that are still equal
        // are long, and a match of gigabytes (a stream that repeats itself exactly) should move at HBM speed.
        const int np = __popcll(pending);
        int lgc = 0;
        while ((1 << lgc) < np) lgc++;
        const int ncols = 1 << lgc;
        int G = (*C.gnw >> lgc) - (want_rev ? 1 : 0);
        if (G > MRZ_FARM_GMAX) G = MRZ_FARM_GMAX;
        // first round: most matches end within its reach, and every extra helper is one more answer to wait for
        if (rounds == 0 && G > MRZ_FARM_ROWS0) G = MRZ_FARM_ROWS0;
        const int mult = rounds == 0 ? 1 : (rounds == 1 ? 4 : MRZ_FARM_BULK_MULT);
        const int nass = G << lgc;
        const int my_col = __popcll(pending & mrz_low_mask(lane));  // column of entry `lane` (if pending)
        const bool mine = lane < nsx && ((pending >> lane) & 1);
        *C.gseq += 1;
        const unsigned long long seq = *C.gseq;
        // post: one store instruction carries the whole job
        {
            const int col = lane - 8;  // lanes 8..23 carry the entry offsets by column
            if (mine) B->farm_min[my_col] = (unsigned long long)my_op;  // compaction through LDS
            MRZ_WAVE_SYNC();
            const int64_t op_of = (col >= 0 && col < np) ? (int64_t)B->farm_min[col] : 0;
            MRZ_WAVE_SYNC();
            unsigned long long v = 0;
            if (lane == 0) v = (unsigned long long)p0;
            if (lane == 1) v = (unsigned long long)(maxf > 0 ? maxf : 0);
            if (lane == 2) v = (unsigned long long)floor_p;
            if (lane == 3) v = (unsigned long long)base;
            if (lane == 4) v = (unsigned long long)(lgc | (G << 8) | ((want_rev ? 1 : 0) << 16) | (mult << 24));
            if (lane >= 8) v = (unsigned long long)((col < np) ? op_of : p0);
            if (lane < 8 + MRZ_FARM_ENTRIES) mrz_g_storeu(&g->words[lane], (seq << MRZ_FARM_SHIFT) | v);
        }
        PROF_ADD(MRZ_ST_F_POST);
        // which result words this lane watches
        int colw[MRZ_FARM_WATCH];
        bool watch[MRZ_FARM_WATCH];
#pragma unroll
        for (int j = 0; j < MRZ_FARM_WATCH; j++) {
            const int w = lane + 64 * j;
            colw[j] = w & (ncols - 1);
            watch[j] = w < nass && colw[j] < np;
        }
        const bool watch_rev = want_rev && mine;
        unsigned long long val[MRZ_FARM_WATCH], rv = 0, ready = 0;
#pragma unroll
        for (int j = 0; j < MRZ_FARM_WATCH; j++) val[j] = 0;
        int spins = 0;
        while (true) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < MRZ_FARM_WATCH; j++)
                if (watch[j]) val[j] = mrz_g_loadu(&g->res[lane + 64 * j]);
            if (watch_rev) rv = mrz_g_loadu(&g->rev[my_col]);
            if (lane == 63) ready = mrz_g_loadu(&g->ready);
#pragma unroll
            for (int j = 0; j < MRZ_FARM_WATCH; j++)
                if (watch[j]) ok = ok && (val[j] >> MRZ_FARM_SHIFT) == seq;
            if (watch_rev) ok = ok && (rv >> MRZ_FARM_SHIFT) == seq;
            if (__ballot(!ok) == 0) break;
            if (spins++ >= MRZ_SPIN_LIMIT) {
                // helpers never answered (preempted, not resident): no more farm rounds in this launch.  A late
                // answer carries this round's number and is never looked at again.
                *C.gnw = -1;
                return false;
            }
            __builtin_amdgcn_s_sleep(MRZ_FARM_LEADER_SLEEP);
        }
        {
            int seen = (int)mrz_bcast64((int64_t)ready, 63);
            if (seen > C.n_helpers) seen = C.n_helpers;
            if (seen > *C.gnw) *C.gnw = seen;
        }
        PROF_ADD(MRZ_ST_F_WAIT);
        ST_ADD(MRZ_ST_FARMED, 1);
        // fold: the stop of a column is the lowest offset any of its stripes reported
        if (lane < MRZ_FARM_ENTRIES) B->farm_min[lane] = MRZ_FARM_NONE;
        MRZ_WAVE_SYNC();
#pragma unroll
        for (int j = 0; j < MRZ_FARM_WATCH; j++) {
            const unsigned long long off = val[j] & MRZ_FARM_PAYLOAD;
            if (watch[j] && off != MRZ_FARM_NONE) atomicMin(&B->farm_min[colw[j]], off);
        }
        MRZ_WAVE_SYNC();
        const unsigned long long m = mine ? B->farm_min[my_col] : MRZ_FARM_NONE;
        const bool resolved = mine && m != MRZ_FARM_NONE;
        if (resolved) fwd = (int64_t)m;
        if (watch_rev) rev = (int64_t)(rv & MRZ_FARM_PAYLOAD);
        pending &= ~__ballot(resolved);
        base += (int64_t)G * mult * MRZ_FARM_SPW;
        want_rev = false;
        rounds++;
        PROF_ADD(MRZ_ST_F_FOLD);
    }
    *my_fwd = fwd;
    *my_rev = rev;
    return true;
}

// refresh the count of helpers that have started (only until all of them have)
__device__ __forceinline__ void mrz_farm_census(const mrz_cfg &C) {
    if (C.gmb && *C.gnw >= 0 && *C.gnw < C.n_helpers) {
        int seen = (int)mrz_uni64((int64_t)mrz_g_loadu(&C.gmb->ready));
        if (seen > C.n_helpers) seen = C.n_helpers;
        *C.gnw = seen;
    }
}
#endif

// Exact evaluation of up to MRZ_SMAX tag-equal entries of ONE candidate at position qx:
// B->same_off[k] are the entries in probe order, B->pair_res[k] their per-lane
// results ((len << 8) | rev, or -1 when an extension ran past the 64-byte reach).  Long
// entries are extended by the compare farm (two or more of them, or one when the last long
// match was big) or by this workgroup's striped path, which hands over to the farm after its
// first round; then everything is folded in probe order (first longest wins,
// src/rzip.c:446-450).  Accumulates into *xb/*xoff/*xrev/*xh/*xm.
__device__ static bool mrz_resolve_entries(const mrz_cfg &C, mrz_lead &L, mrz_coop_lds *B,
                                           int64_t qx, int nsx, int lane, int64_t *stat,
                                           int64_t *xb, int64_t *xoff, int64_t *xrev, int *xh, int *xm) {
    const uint8_t *__restrict__ buf = C.buf;
    int my_r = 0;            // lane k: result of entry k
    int64_t my_op = 0, my_ml = 0, my_rv = 0;
    if (lane < nsx) {
        my_r = B->pair_res[lane];
        my_op = B->same_off[lane];
    }
    const mrz_u64 longmask = __ballot(lane < nsx && my_r < 0);
    const int nlong = __popcll(longmask);
    if (nlong) *C.long_seen = 1;
    const int64_t floor_p = L.last_match > 0 ? L.last_match : 0;
    bool farmed = false;
#if MRZ_HELPER_WGS > 0
    if (nlong && C.gmb && nsx <= MRZ_FARM_ENTRIES) {
        mrz_farm_census(C);
        if (*C.gnw >= 2 * MRZ_FARM_ENTRIES && (nlong >= 2 || *C.farm_hint >= MRZ_FARM_HINT_MIN)) {
            int64_t fw, rv;
            if (mrz_farm(C, B, qx, C.end - qx, floor_p, 0, nsx, my_op, my_r < 0 && my_op < qx, true, lane, stat, &fw,
                         &rv)) {
                if (my_r < 0) {
                    my_ml = my_op < qx ? fw + rv : 0;
                    my_rv = rv;
                    if (my_ml < MRZ_MIN_MATCH) my_ml = 0;
                }
                farmed = true;
            }  // else: the farm gave up, everything is measured locally below
        }
    }
#endif
    if (!farmed) {
        for (int k = 0; k < nsx; k++) {
            if (!((longmask >> k) & 1)) continue;
            const int64_t op = mrz_bcast64(my_op, k);
            int64_t rv = 0, cont = 0;
            int64_t ml = mrz_long_match_len(buf, C.mb, C.mb_seq, qx, op, C.end, L.last_match, &rv, lane, stat,
#if MRZ_HELPER_WGS > 0
                                            (C.gmb && *C.gnw >= MRZ_FARM_ENTRIES) ? &cont : nullptr
#else
                                            nullptr
#endif
            );
#if MRZ_HELPER_WGS > 0
            if (ml < 0) {
                // still equal after the local round: the rest of the forward compare goes to the farm
                int64_t fw, dummy;
                if (mrz_farm(C, B, qx, C.end - qx, floor_p, cont, 1, op, true, false, lane, stat, &fw, &dummy)) {
                    fw = mrz_bcast64(fw, 0);
                    ml = fw + rv;
                    if (ml < MRZ_MIN_MATCH) ml = 0;
                } else  // the farm gave up: all of it locally
                    ml = mrz_long_match_len(buf, C.mb, C.mb_seq, qx, op, C.end, L.last_match, &rv, lane, stat, nullptr);
            }
#endif
            if (lane == k) {
                my_ml = ml;
                my_rv = rv;
            }
        }
    }
    if (lane < nsx && my_r >= 0) {
        my_ml = my_r >> 8;
        my_rv = my_r & 0xff;
    }
#if MRZ_HELPER_WGS > 0
    if (nlong) {
        // remember how far long matches reach here (decides farm-first for single long entries)
        int64_t far = 0;
        for (int k = 0; k < nsx; k++)
            if ((longmask >> k) & 1) {
                const int64_t v = mrz_bcast64(my_ml, k);
                if (v > far) far = v;
            }
        *C.farm_hint = far;
    }
#endif
    for (int k = 0; k < nsx; k++) {
        const int64_t ml = mrz_bcast64(my_ml, k);
        if (ml) {
            if (ml > *xb) {
                const int64_t rv = mrz_bcast64(my_rv, k);
                *xb = ml;
                *xoff = mrz_bcast64(my_op, k) - rv;
                *xrev = rv;
            }
            *xh += 1;
        } else
            *xm += 1;
    }
    return true;
}

// One candidate, fully in order: the wave-cooperative path (any chain length,
// any match length, cascades, chain-limit evictions, mask promotion).
__device__ static bool mrz_seq_candidate(const mrz_cfg &C, mrz_lead &L, mrz_coop_lds *B, int64_t t, int lane,
                                         int64_t *stat) {
    const uint8_t *__restrict__ buf = C.buf;
    mrz_slot *tab = C.tab;
    const int64_t p = L.p, end = C.end, slot_mask = C.slot_mask, max_chain = C.max_chain;
    // ---- one pass over the chain: find_best_match (:426-462) and, when this
    // position is inserted (:579), the probe walk of insert_hash ----------
    PROF_T0();
    const bool do_insert = (t & L.tag_mask) == L.tag_mask;
    const int64_t better = (L.min_mask << 1) | 1;
    const int my_rank = mrz_ones_rank(t);
    if (lane == 0) {
        B->n_written = 0;
        B->cull_slot = -1;
    }
    int64_t mlen = 0, m_off = 0, m_rev = 0;
    bool ins_found = !do_insert;
    int64_t ins_slot = 0, occ_t = 0, occ_off = 0;
    int ins_kind = 0;
    if (do_insert) {
        L.inserts++;
        L.count++;
    }
    {
        const int64_t h0 = t & slot_mask;
        int64_t round = 0, victim_h = 0;
        for (int64_t b = 0;; b += MRZ_WAVE) {
            const int64_t s = (h0 + b + lane) & slot_mask;
            const mrz_slot e = tab[s];
            const bool empty = (e.off | e.t) == 0;
            const mrz_u64 m_empty = __ballot(empty);
            const int first_empty = m_empty ? __ffsll((long long)m_empty) - 1 : MRZ_WAVE;
            PROF_ADD(MRZ_ST_S_TAB);
            if (!ins_found)
                ins_found = mrz_insert_step(e, empty, t, my_rank, h0 + b, slot_mask, better, max_chain, &round,
                                            &victim_h, &L.count, &L.victim_round, &ins_slot, &ins_kind, &occ_t,
                                            &occ_off);
            const mrz_u64 m_same = __ballot(!empty && e.t == t) & mrz_low_mask(first_empty);
            mrz_u64 todo = m_same;
            while (todo) {
                // tag-equal entries of this step in probe order, MRZ_SMAX per pass: lane k takes the k-th
                const int my_idx = __popcll(todo & mrz_low_mask(lane));
                const bool is_mine = ((todo >> lane) & 1) && my_idx < MRZ_SMAX;
                if (is_mine) B->same_off[my_idx] = e.off;
                const int total = __popcll(todo);
                const int npass = total < MRZ_SMAX ? total : MRZ_SMAX;
                // drop the entries taken in this pass from `todo`
                mrz_u64 rest = todo;
                for (int k = 0; k < npass; k++) rest &= rest - 1;
                MRZ_WAVE_SYNC();
                if (lane < npass) {
                    int64_t ml, rv;
                    bool lng;
                    mrz_lane_match_len(buf, p, B->same_off[lane], end, L.last_match, &ml, &rv, &lng);
                    B->pair_res[lane] = lng ? -1 : (int)((ml << 8) | rv);
                }
                MRZ_WAVE_SYNC();
                PROF_ADD(MRZ_ST_S_PAIR);
                int xh = 0, xm = 0;
                if (!mrz_resolve_entries(C, L, B, p, npass, lane, stat, &mlen, &m_off, &m_rev, &xh, &xm))
                    return false;
                L.tag_hits += xh;
                L.tag_misses += xm;
#ifdef MRZ_DBG_HITS_COOP
                if (lane == 0) MRZ_DBG_HITS_COOP(p, xh, xm);
#endif
                todo = rest;
                PROF_T0R();
            }
            if (first_empty < MRZ_WAVE) break;
        }
    }

    // ---- insert + cull (:579-584) -------------------------------------
    PROF_T0R();
    if (do_insert) {
        int np = 0;
        int64_t it = t, io = p;
        while (true) {
            if (np >= MRZ_CASCADE_MAX) {  // cannot happen: every level has a strictly lower rank
                if (lane == 0) C.st->error = 2;
                return false;
            }
            if (lane == 0) {
                B->pend_h[np] = ins_slot;
                B->pend_t[np] = it;
                B->pend_o[np] = io;
            }
            np++;
            if (ins_kind != 2) break;
            // re-insert the displaced occupant: its own probe walk
            it = occ_t;
            io = occ_off;
            const int64_t h0 = it & slot_mask;
            const int rank2 = mrz_ones_rank(it);
            int64_t round = 0, victim_h = 0;
            for (int64_t b = 0;; b += MRZ_WAVE) {
                const int64_t s = (h0 + b + lane) & slot_mask;
                const mrz_slot e = tab[s];
                const bool empty = (e.off | e.t) == 0;
                if (mrz_insert_step(e, empty, it, rank2, h0 + b, slot_mask, better, max_chain, &round, &victim_h,
                                    &L.count, &L.victim_round, &ins_slot, &ins_kind, &occ_t, &occ_off))
                    break;
            }
        }
        if (lane == 0) B->n_written = np;
        // write back innermost-first (the recursion's return order)
        while (np-- > 0) {
            const int64_t hs = mrz_uni64(B->pend_h[np]);
            if (lane == 0) {
                mrz_slot w;
                w.off = B->pend_o[np];
                w.t = B->pend_t[np];
                tab[hs] = w;
            }
        }
        if (L.count > C.limit) {
            mrz_cull_one(C, L, lane);
            if (lane == 0) B->cull_slot = L.clean_ptr;
        }
    }
    const bool okk = mrz_select_emit(C, L, mlen, m_off, m_rev, lane);
    PROF_ADD(MRZ_ST_S_INS);
    return okk;
}

