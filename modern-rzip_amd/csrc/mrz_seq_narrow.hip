// mrz_seq_narrow.hip -- the NARROW engine of the sequencer: one wave walks the state machine, 64 lanes wide.
//
// Same state machine, same matcher state (mrz_seq_state) and same compare farm protocol as the wide engine in
// mrz_sequencer.hip; the host picks one of the two kernels per segment launch (mrz_capi.hip) from the regime
// the previous segments were in.  This one has the shortest dependency chain per emitted match: it is what
// streams that are one long match after another run on (BASELINE configs[1]); the wide engine (512 candidates
// per batch, commit in segments) is what dense candidates without long matches run on (text, noise).
//
// mrz_sequencer.hip -- the exact, order-preserving core of the rzip stage.
//
// hash_search's main loop (src/rzip.c:548-599) is a state machine whose table
// contents at position p depend on every earlier decision (skipped inserts
// inside emitted matches :596-598, probe-order evictions :264-297, in-place
// culling without tombstones :305-328, the process-lifetime victim_round :259),
// so it has to be replayed in position order to stay bit-exact.  What the GPU
// changes is the width of every step.  One workgroup runs it:
//
//   wave 0 ("leader") walks the state machine; all its control values are
//   wave-uniform (kept in SGPRs via readlane/readfirstlane) and every step is 64
//   lanes wide --
//     * candidate discovery: the tag-scan bitmap is read 4096 positions per load
//       and the next candidate found with one ballot (the `continue` at :573
//       means ONLY positions passing minimum_tag_mask run the loop body, emit
//       test included);
//     * find_best_match (:426-462) and the probe walk of insert_hash (:262-297)
//       share one pass over the chain: 64 consecutive slots (1 KiB, coalesced)
//       per step; ballots give first-empty, tag-equal lanes, and the insert
//       walk's stop (empty / due-for-culling / lower-ranked occupant / the
//       max_chain_len-th same-tag entry, counted with popcounts of the ballot);
//     * cascades of displaced occupants are collected and written back
//       innermost-first exactly like the reference's recursion;
//     * clean_one_from_hash (:305-328): 64 slots per sweep step.
//     * single_match_len (:372-397): every tag-equal entry of a step is extended by
//       its own lane, 64 B each way, in one load round trip (most differ there);
//   wave 1 ("stripe helper") waits on an LDS mailbox: a lone entry that runs past
//   the 64-byte reach is extended by both waves (4 KiB each per round, 64 lanes x
//   16 B x 4 pieces, ballot + ffs for the first mismatch); the last wave ("scout")
//   runs ahead of the leader and touches what it will need next.
//   Most candidates do not go one at a time: the BATCH ENGINE (mrz_batch_step)
//   processes up to 64 consecutive candidates, one lane each, speculatively against
//   the table as it stands, and commits the prefix that provably equals the
//   sequential result.  Tag-equal entries that are all long go to the COMPARE FARM:
//   helper workgroups on the other CUs (this kernel's blocks 1..) behind a mailbox
//   in device memory.  Both are described where they are defined.
// Emitted matches go to an event list; record encoding, literal gathering and
// the CRC are separate parallel kernels.
//
// Bound: latency -- one dependency chain of table probes, data probes and
// cross-CU hand-offs; see DESIGN.md 4.2 for the measured split.
#include "mrz_device.h"
#include <stdlib.h>

// cross-lane LDS exchange inside one wave: the hardware runs the lanes in lockstep, the CPU
// emulator needs a rendezvous
#ifdef __HIP_DEVICE_COMPILE__
#define MRZ_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define MRZ_WAVE_SYNC() (void)__ballot(1)
#endif

#ifndef MRZ_SEQ_WAVES
#define MRZ_SEQ_WAVES 3  // leader, one stripe helper, scout.  Measured 8 -> 3: -6 % on the benchmark chunk (no VGPR
                         // spills, a third fewer SGPR spills, lighter helper workgroups); 2 (no scout) is slower
#endif
#define MRZ_SEQ_THREADS (64 * MRZ_SEQ_WAVES)
// wave 0 = leader, waves 1..MRZ_STRIPE_WAVES-1 = striping helpers, last wave = scout (prefetcher)
#define MRZ_STRIPE_WAVES (MRZ_SEQ_WAVES > 2 ? MRZ_SEQ_WAVES - 1 : MRZ_SEQ_WAVES)
#define MRZ_HAVE_SCOUT (MRZ_SEQ_WAVES > 2)
#define MRZ_CASCADE_MAX 64
#ifndef MRZ_SEQ_CREDIT
#define MRZ_SEQ_CREDIT 8       // candidates sent through the cooperative path after repeated tiny batches
#endif
#ifndef MRZ_WIDTH_MULT
#define MRZ_WIDTH_MULT 2
#endif
#ifndef MRZ_WIDTH_DENSE
#define MRZ_WIDTH_DENSE 12
#endif
#ifndef MRZ_WIDTH_MULT_DENSE
#define MRZ_WIDTH_MULT_DENSE 8
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
enum { MRZ_ST_BATCHES, MRZ_ST_BATCH_LANES, MRZ_ST_SEQ, MRZ_ST_CUT_LONG, MRZ_ST_CUT_WALK, MRZ_ST_CUT_CONFLICT,
       MRZ_ST_CUT_CULL, MRZ_ST_BATCH_EMITS, MRZ_ST_CUT_CASCADE, MRZ_ST_PAIRS, MRZ_ST_BATCH_FORMED,
       MRZ_ST_T_FORM, MRZ_ST_T_WALK, MRZ_ST_T_WALK2, MRZ_ST_T_PAIRS, MRZ_ST_T_SCANS, MRZ_ST_T_CONFLICT, MRZ_ST_T_COMMIT,
       MRZ_ST_T_SEQ, MRZ_ST_T_WINDOW, MRZ_ST_T_LONG, MRZ_ST_T_FOLD, MRZ_ST_FARMED, MRZ_ST_L_POST, MRZ_ST_L_STRIPE, MRZ_ST_L_BWD, MRZ_ST_L_WAIT,
       MRZ_ST_L_ROUNDS, MRZ_ST_F_POST, MRZ_ST_F_WAIT, MRZ_ST_F_FOLD, MRZ_ST_F_HELPER,
       MRZ_ST_H_FIELDS, MRZ_ST_H_FWD, MRZ_ST_H_BWD, MRZ_ST_H_DRAIN, MRZ_ST_H_ROUNDS, MRZ_ST_S_TAB, MRZ_ST_S_PAIR,
       MRZ_ST_S_INS, MRZ_ST_N };

struct mrz_seq_args {
    const uint8_t *buf;
    mrz_slot *tab;
    const int64_t *tags;      // dense tags of this segment
    const mrz_u64 *bitmap;    // candidate bitmap of this segment (64 positions per word)
    mrz_event *events;
    mrz_seq_state *st;
    int64_t seg_start;
    int64_t seg_len;
    void *gmailbox;           // mrz_gmailbox in device memory, zeroed by the host before every launch
    int n_helpers;            // helper workgroups in this launch (grid size - 1)
};

// LDS mailbox between the leader and the helper waves: one long forward
// extension at a time, striped over all waves of the workgroup
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

// workgroup-scope accesses to the mailbox words
__device__ __forceinline__ int mrz_mb_load(int *p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void mrz_mb_store(int *p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void mrz_mb_add(int *p, int v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

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

// Long candidate: the forward extension is striped over every wave of the
// workgroup (W x 4 KiB per round); the leader folds the per-wave results and does
// the backward extension while the helpers are busy with the first round.
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
    if (maxf > 0) {
        for (int64_t base = 0;; base += (int64_t)MRZ_STRIPE_WAVES * MRZ_STRIPE) {
            if (MRZ_STRIPE_WAVES > 1) {
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
            if (MRZ_STRIPE_WAVES > 1) {
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
                *cont_base = base + (int64_t)MRZ_STRIPE_WAVES * MRZ_STRIPE;
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

// The scout (last wave of the leader's workgroup) never decides anything: it reads the
// leader's published position and pulls into this CU's L1 / this XCD's L2 what the leader
// will need next -- the tags of the following 128 candidates, the first table line of each
// of their probe chains, the bytes at the offsets of tag-equal entries, and the cull sweep
// window -- so that the leader's dependent loads hit cache instead of HBM.
struct mrz_scout_args {
    const uint8_t *buf;
    const mrz_slot *tab;
    const int64_t *tags;
    const mrz_u64 *bitmap;
    int64_t seg_start, lim, nwords, slot_mask, nslots;
};

__device__ static void mrz_scout_loop(const mrz_scout_args &S, mrz_mailbox *mb, int lane) {
    __shared__ int sc_pref[64];
    __shared__ mrz_u64 sc_word[64];
    int seen = 0;
    unsigned sink = 0;
    while (true) {
        int s;
        while ((s = mrz_uni(mrz_mb_load(&mb->scout_seq))) == seen) {
            if (mrz_uni(mrz_mb_load(&mb->quit))) {
                if (sink == 0x9e3779b9u) sc_pref[0] = (int)sink;  // keeps the prefetch loads alive
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
        const int64_t wb = S.seg_start + ((pos - S.seg_start) >> 12 << 12);
        const int64_t idx = ((wb - S.seg_start) >> 6) + lane;
        mrz_u64 w = idx < S.nwords ? S.bitmap[idx] : 0ull;
        const int64_t lane_lo = wb + (int64_t)lane * 64;
        if (pos > lane_lo) {
            const int64_t sh = pos - lane_lo;
            w = sh >= 64 ? 0ull : (w >> sh) << sh;
        }
        const int cnt = __popcll(w);
        const int incl = mrz_wave_incl_sum(cnt, lane);
        const int total = mrz_lane_read(incl, 63);
        sc_pref[lane] = incl - cnt;
        sc_word[lane] = w;
        MRZ_WAVE_SYNC();
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int r = half * 64 + lane;
            if (r < total) {
                int lo = 0, hi = 63;
#pragma unroll
                for (int it = 0; it < 6; it++) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (sc_pref[mid] <= r)
                        lo = mid;
                    else
                        hi = mid - 1;
                }
                const int64_t qq = wb + (int64_t)lo * 64 + mrz_select64(sc_word[lo], r - sc_pref[lo]);
                if (qq <= S.lim) {
                    const int64_t t = S.tags[qq - S.seg_start];
                    const int64_t h = t & S.slot_mask;
                    const mrz_slot e0 = S.tab[h];
                    const mrz_slot e1 = S.tab[(h + 7) & S.slot_mask];  // the line may straddle
                    sink += (unsigned)e0.off + (unsigned)e1.off;
                    if (e0.t == t && e0.off > 0) sink += S.buf[e0.off];
                    sink += S.buf[qq];
                }
            }
        }
        MRZ_WAVE_SYNC();
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

// ---- the compare farm: helper workgroups on the other CUs ---------------------------------
// A look-up on repetitive input can find max_chain_len tag-equal entries that are ALL tens of
// KiB long (every earlier copy of the same text): megabytes to compare for one candidate.  One
// CU keeps only ~8 KiB of loads in flight (~20 GB/s on cold data), so the compares are spread
// over the whole chip: the grid carries helper workgroups (MRZ_SEQ_WAVES waves each, about one per CU)
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
#define MRZ_FARM_SPW (MRZ_SEQ_WAVES * MRZ_FARM_WAVE_BYTES)  // bytes of each stream per helper and round
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
            if (atomicAdd(&s_cnt, 1u) == MRZ_SEQ_WAVES - 1) {
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
#define MRZ_FILTER_SIZE 4096
#define MRZ_CULL_WINDOW 4  // x 64 slots scanned ahead of tag_clean_ptr per batch

struct mrz_batch_lds {
    int pref[64];
    mrz_u64 word[64];
    int64_t qpos[64];
    int64_t same_off[64][MRZ_SMAX];
    int same_slot[64][MRZ_SMAX];
    int pair_res[64][MRZ_SMAX];  // (len << 8) | rev, or -1 = needs the cooperative path
    unsigned filter[MRZ_FILTER_SIZE];
    unsigned long long farm_min[16];  // per entry: lowest stop offset any helper reported
};




__device__ __forceinline__ unsigned mrz_filter_slot(int slot) {
    return ((unsigned)(slot >> 6) * 2654435761u) >> 20;  // 12 bits
}

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
    int *sc_seq;               // scout hand-off counter (shared with the main loop)
    int64_t *pred_end;         // where the match being measured is expected to end (last end + last stride), or -1
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
__device__ static bool mrz_farm(const mrz_cfg &C, mrz_batch_lds *B, int64_t p0, int64_t maxf, int64_t floor_p,
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
// B->same_off[row][k] are the entries in probe order, B->pair_res[row][k] their per-lane
// results ((len << 8) | rev, or -1 when an extension ran past the 64-byte reach).  Long
// entries are extended by the compare farm (two or more of them, or one when the last long
// match was big) or by this workgroup's striped path, which hands over to the farm after its
// first round; then everything is folded in probe order (first longest wins,
// src/rzip.c:446-450).  Accumulates into *xb/*xoff/*xrev/*xh/*xm.
__device__ static bool mrz_resolve_entries(const mrz_cfg &C, mrz_lead &L, mrz_batch_lds *B, mrz_mailbox *mb,
                                           int *mb_seq, int64_t qx, int nsx, int row, int lane, int64_t *stat,
                                           int64_t *xb, int64_t *xoff, int64_t *xrev, int *xh, int *xm) {
    const uint8_t *__restrict__ buf = C.buf;
    int my_r = 0;            // lane k: result of entry k
    int64_t my_op = 0, my_ml = 0, my_rv = 0;
    if (lane < nsx) {
        my_r = B->pair_res[row][lane];
        my_op = B->same_off[row][lane];
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
#ifdef MRZ_PRED_SCOUT  // measured: -2.8 % on the benchmark stream (the extra loads compete with the farm), off
            // While the helpers compare, the scout warms the caches for what comes AFTER this match: on a stream that
            // is one long match after another the ends are evenly spaced, so the next position is predictable (pure
            // prefetch: a wrong guess costs nothing but the loads).
            if (MRZ_HAVE_SCOUT && *C.pred_end > qx) {
                *C.sc_seq += 1;
                if (lane == 0) {
                    mb->scout_pos = *C.pred_end;
                    mrz_mb_store(&mb->scout_seq, *C.sc_seq);
                }
            }
#endif
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
            int64_t ml = mrz_long_match_len(buf, mb, mb_seq, qx, op, C.end, L.last_match, &rv, lane, stat,
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
                    ml = mrz_long_match_len(buf, mb, mb_seq, qx, op, C.end, L.last_match, &rv, lane, stat, nullptr);
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
__device__ static bool mrz_seq_candidate(const mrz_cfg &C, mrz_lead &L, mrz_batch_lds *B, mrz_mailbox *mb, int *mb_seq,
                                         int64_t *pend_h, int64_t *pend_t, int64_t *pend_o, int64_t t, int lane,
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
                if (is_mine) B->same_off[0][my_idx] = e.off;
                const int total = __popcll(todo);
                const int npass = total < MRZ_SMAX ? total : MRZ_SMAX;
                // drop the entries taken in this pass from `todo`
                mrz_u64 rest = todo;
                for (int k = 0; k < npass; k++) rest &= rest - 1;
                MRZ_WAVE_SYNC();
                if (lane < npass) {
                    int64_t ml, rv;
                    bool lng;
                    mrz_lane_match_len(buf, p, B->same_off[0][lane], end, L.last_match, &ml, &rv, &lng);
                    B->pair_res[0][lane] = lng ? -1 : (int)((ml << 8) | rv);
                }
                MRZ_WAVE_SYNC();
                PROF_ADD(MRZ_ST_S_PAIR);
                int xh = 0, xm = 0;
                if (!mrz_resolve_entries(C, L, B, mb, mb_seq, p, npass, 0, lane, stat, &mlen, &m_off, &m_rev, &xh, &xm))
                    return false;
                L.tag_hits += xh;
                L.tag_misses += xm;
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
                pend_h[np] = ins_slot;
                pend_t[np] = it;
                pend_o[np] = io;
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
        // write back innermost-first (the recursion's return order)
        while (np-- > 0) {
            const int64_t hs = mrz_uni64(pend_h[np]);
            if (lane == 0) {
                mrz_slot w;
                w.off = pend_o[np];
                w.t = pend_t[np];
                tab[hs] = w;
            }
        }
        if (L.count > C.limit) mrz_cull_one(C, L, lane);
    }
    const bool okk = mrz_select_emit(C, L, mlen, m_off, m_rev, lane);
    PROF_ADD(MRZ_ST_S_INS);
    return okk;
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
// Returns the number of candidates consumed from this window (>= 1), or 0 when the
// first candidate has to go through mrz_seq_candidate.  `w` is this lane's bitmap
// word of the 4096-position window starting at `wb`, already masked to (L.p, lim].
__device__ static int mrz_batch_step(const mrz_cfg &C, mrz_lead &L, mrz_batch_lds *B, const int64_t *__restrict__ tags,
                                     int64_t seg_start, int64_t wb, mrz_u64 w, unsigned epoch, int width, int lane,
                                     bool *ok, int64_t *stat, mrz_mailbox *mb, int *mb_seq) {
    const uint8_t *__restrict__ buf = C.buf;
    mrz_slot *tab = C.tab;
    const int smask = (int)C.slot_mask;
    const int max_chain = (int)C.max_chain;
    const int64_t better = (L.min_mask << 1) | 1;
    *ok = true;
    PROF_T0();

    // ---- formation: lane r takes the r-th candidate of the window ------------
    const int cnt = __popcll(w);
    const int incl = mrz_wave_incl_sum(cnt, lane);
    const int total = mrz_lane_read(incl, 63);
    const int nb = total < width ? total : width;
    B->pref[lane] = incl - cnt;
    B->word[lane] = w;
    MRZ_WAVE_SYNC();
    const bool have = lane < nb;
    int wlo = 0, whi = 63;
#pragma unroll
    for (int it = 0; it < 6; it++) {
        const int mid = (wlo + whi + 1) >> 1;
        if (B->pref[mid] <= lane)
            wlo = mid;
        else
            whi = mid - 1;
    }
    int64_t q = 0, t = 0;
    if (have) {
        q = wb + (int64_t)wlo * 64 + mrz_select64(B->word[wlo], lane - B->pref[wlo]);
        t = tags[q - seg_start];
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
        if (!mrz_resolve_entries(C, L, B, mb, mb_seq, qx, nsx, x, lane, stat, &xb, &xoff, &xrev, &xh, &xm)) {
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
    __shared__ int64_t pend_h[MRZ_CASCADE_MAX], pend_t[MRZ_CASCADE_MAX], pend_o[MRZ_CASCADE_MAX];
    __shared__ mrz_mailbox mbox;
    __shared__ mrz_batch_lds batch;

    const int lane = threadIdx.x & 63;
    const int wave = mrz_uni((int)(threadIdx.x >> 6));
    mrz_seq_state *st = a.st;
    mrz_mailbox *mb = &mbox;

    if (st->finished || st->error) return;
#if MRZ_HELPER_WGS > 0
    if (blockIdx.x != 0) {
        mrz_helper_wg(a.buf, (mrz_gmailbox *)a.gmailbox);
        return;
    }
#endif
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
        S.tags = a.tags;
        S.bitmap = a.bitmap;
        S.seg_start = a.seg_start;
        const int64_t s_end = a.seg_start + a.seg_len;
        S.lim = (st->end < s_end - 1) ? st->end : s_end - 1;
        S.nwords = (a.seg_len + 63) / 64;
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
    const int64_t seg_start = a.seg_start;
    const int64_t seg_end = a.seg_start + a.seg_len;
    const int64_t lim = (C.end < seg_end - 1) ? C.end : seg_end - 1;  // last candidate position of this launch
    const int64_t nwords = (a.seg_len + 63) / 64;

    int64_t win_base = -1;
    mrz_u64 myword = 0;
    int mb_seq = 0;
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
    C.sc_seq = &sc_seq;
    int64_t pred_end = -1, prev_end = -1;
    C.pred_end = &pred_end;
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
        // ---- the 4096-position bitmap window that holds position p + 1 ------------
        int64_t pos = L.p + 1;
        if (pos < seg_start) pos = seg_start;
        if (pos > lim) break;
        const int64_t wb = seg_start + ((pos - seg_start) >> 12 << 12);
        if (wb != win_base) {
            const int64_t idx = ((wb - seg_start) >> 6) + lane;
            myword = idx < nwords ? a.bitmap[idx] : 0ull;
            win_base = wb;
        }
        // this lane's word restricted to [pos, lim]
        const int64_t lane_lo = wb + (int64_t)lane * 64;
        mrz_u64 w = myword;
        if (pos > lane_lo) {
            const int64_t sh = pos - lane_lo;
            w = sh >= 64 ? 0ull : (w >> sh) << sh;
        }
        if (lim < lane_lo + 63) {
            const int64_t keepbits = lim - lane_lo + 1;
            w = keepbits <= 0 ? 0ull : (w & mrz_low_mask((int)keepbits));
        }
        const mrz_u64 any = __ballot(w != 0ull);
        if (!any) {
            // nothing left in this window
            const int64_t nxt = wb + 4096;
            L.p = (nxt - 1 < lim) ? nxt - 1 : lim;
            continue;
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
            used = mrz_batch_step(C, L, &batch, a.tags, seg_start, wb, w, epoch, width, lane, &ok, stat, mb, &mb_seq);
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
            // first candidate of the window through the cooperative path
            const int fl = __ffsll((long long)any) - 1;
            const mrz_u64 wl = (mrz_u64)mrz_bcast64((int64_t)w, fl);
            L.p = wb + (int64_t)fl * 64 + (__ffsll((long long)wl) - 1);
            const int64_t t = mrz_uni64(a.tags[L.p - seg_start]);
            if ((t & L.min_mask) == L.min_mask)  // src/rzip.c:573 with the mask reached by now
                ok = mrz_seq_candidate(C, L, &batch, mb, &mb_seq, pend_h, pend_t, pend_o, t, lane, stat);
            PROF_ADD(MRZ_ST_T_SEQ);
        }
        if (first_after_emit) {
            const bool first_was_long = long_seen && used <= 1;
            int &c = pred_long[emit_cls];
            c = first_was_long ? (c < 3 ? c + 1 : 3) : (c > 0 ? c - 1 : 0);
        }
        after_emit = L.n_events != ev_before;
        if (after_emit) {  // stride predictor of match ends
            const int64_t stride = prev_end >= 0 ? L.last_match - prev_end : 0;
            prev_end = L.last_match;
            pred_end = (stride > 0 && stride < (1ll << 28)) ? L.last_match + stride : -1;
        }
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

extern "C" hipError_t mrz_launch_sequencer_narrow(hipStream_t stream, const uint8_t *buf, mrz_slot *tab, const int64_t *tags,
                                                  const mrz_u64 *bitmap, mrz_event *events, mrz_seq_state *st,
                                                  int64_t seg_start, int64_t seg_len, void *gmailbox, int n_helpers) {
    mrz_seq_args a;
    a.buf = buf;
    a.tab = tab;
    a.tags = tags;
    a.bitmap = bitmap;
    a.events = events;
    a.st = st;
    a.seg_start = seg_start;
    a.seg_len = seg_len;
    a.gmailbox = gmailbox;
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
    hipLaunchKernelGGL(mrz_seq_narrow_kernel, dim3(1 + a.n_helpers), dim3(MRZ_SEQ_THREADS), 0, stream, a);
    return hipGetLastError();
}

extern "C" size_t mrz_seq_narrow_mailbox_size(void) { return MRZ_HELPER_WGS > 0 ? sizeof(mrz_gmailbox) : 0; }
