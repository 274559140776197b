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
//     * single_match_len (:372-397), short path: up to 8 tag-equal entries at
//       once, 8 lanes x 16 B each, forwards and backwards in ONE load round trip
//       (most candidates differ within 128 bytes);
//   waves 1..W-1 ("helpers") wait on an LDS mailbox.  A candidate that runs
//   past 128 bytes takes the long path: its forward extension is striped over
//   all W waves (W x 4 KiB per round, every wave 64 lanes x 16 B x 4 pieces,
//   ballot + ffs for the first mismatch) and the leader folds the per-wave
//   results.  Candidates are folded in probe order, so ties resolve exactly as
//   in the reference.
// Emitted matches go to an event list; record encoding, literal gathering and
// the CRC are separate parallel kernels.
//
// Bound: latency (dependent 1 KiB probes into the table, which lives in this
// XCD's L2 / the Infinity Cache) and this CU's L1/L2 bandwidth for the match
// extension -- not HBM bandwidth.
#include "mrz_device.h"

#ifndef MRZ_SEQ_WAVES
#define MRZ_SEQ_WAVES 16
#endif
#define MRZ_SEQ_THREADS (64 * MRZ_SEQ_WAVES)
#define MRZ_CASCADE_MAX 64

// optional in-kernel cycle accounting (diagnostic builds only: -DMRZ_SEQ_PROFILE)
#ifdef MRZ_SEQ_PROFILE
#define PROF_DECL int64_t prof_t0 = 0, prof_acc[16] = { 0 }
#define PROF_START() prof_t0 = (int64_t)__builtin_amdgcn_s_memtime()
#define PROF_STOP(k)                                                     \
    do {                                                                 \
        const int64_t now__ = (int64_t)__builtin_amdgcn_s_memtime();     \
        prof_acc[k] += now__ - prof_t0;                                  \
        prof_t0 = now__;                                                 \
    } while (0)
#define PROF_COUNT(k) prof_acc[k] += 1
#else
#define PROF_DECL
#define PROF_START()
#define PROF_STOP(k)
#define PROF_COUNT(k)
#endif

struct mrz_seq_args {
    const uint8_t *buf;
    mrz_slot *tab;
    const int64_t *tags;      // dense tags of this segment
    const mrz_u64 *bitmap;    // candidate bitmap of this segment (64 positions per word)
    mrz_event *events;
    mrz_seq_state *st;
    int64_t seg_start;
    int64_t seg_len;
};

// LDS mailbox between the leader and the helper waves: one long forward
// extension at a time, striped over all waves of the workgroup
struct mrz_mailbox {
    int64_t p0, op, maxf, base;     // compare buf[p0+x] with buf[op+x] for x in [base + wave*STRIPE, +STRIPE), x < maxf
    int64_t res[MRZ_SEQ_WAVES];     // per wave: first stop offset of its stripe, or -1
    int seq;                        // bumped by the leader for every round; helpers wait on it
    int done;                       // helpers add 1 when their stripe is finished
    int quit;
    int pad;
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

#define MRZ_STRIPE_PIECES 4
#define MRZ_STRIPE (MRZ_STRIPE_PIECES * 1024)

// Forward compare of one 4 KiB stripe starting at `base` (64 lanes x 16 B x 4
// pieces, all loads issued before the first compare).  Returns the offset (from
// p0) at which `while (p < end && buf[p] == buf[op])` (src/rzip.c:378) stops if
// that lies inside or before this stripe's reach, else -1.
__device__ static int64_t mrz_wave_fwd_stripe(const uint8_t *__restrict__ buf, int64_t p0, int64_t op, int64_t maxf,
                                              int64_t base, int lane) {
    uint4 a[MRZ_STRIPE_PIECES], b[MRZ_STRIPE_PIECES];
#pragma unroll
    for (int j = 0; j < MRZ_STRIPE_PIECES; j++) {
        const int64_t off = base + j * 1024 + lane * 16;
        if (off < maxf) {
            a[j] = mrz_ld16(buf + p0 + off);
            b[j] = mrz_ld16(buf + op + off);
        }
    }
    int64_t found = -1;
#pragma unroll
    for (int j = 0; j < MRZ_STRIPE_PIECES; j++) {
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
// workgroup (W x 4 KiB per round); the leader folds the per-wave results.
__device__ static int64_t mrz_long_match_len(const uint8_t *__restrict__ buf, mrz_mailbox *mb, int *mb_seq, int64_t p0,
                                             int64_t op, int64_t end, int64_t last_match, int64_t *rev_out, int lane) {
    *rev_out = 0;
    if (op >= p0) return 0;
    const int64_t maxf = end - p0;
    int64_t fwd = 0;
    if (maxf > 0) {
        for (int64_t base = 0;; base += (int64_t)MRZ_SEQ_WAVES * MRZ_STRIPE) {
            if (MRZ_SEQ_WAVES > 1) {
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
            int64_t best = mrz_wave_fwd_stripe(buf, p0, op, maxf, base, lane);  // the leader's own stripe (wave 0)
            if (MRZ_SEQ_WAVES > 1) {
                while (mrz_uni(mrz_mb_load(&mb->done)) < MRZ_SEQ_WAVES - 1) __builtin_amdgcn_s_sleep(1);
                for (int w = 1; w < MRZ_SEQ_WAVES && best < 0; w++) best = mrz_uni64(mb->res[w]);
            }
            if (best >= 0) {
                fwd = best;
                break;
            }
        }
    }
    const int64_t floor_p = last_match > 0 ? last_match : 0;
    int64_t maxb = p0 - floor_p;
    if (op < maxb) maxb = op;
    const int64_t rev = mrz_wave_bwd(buf, p0, op, maxb, lane);
    *rev_out = rev;
    const int64_t len = fwd + rev;
    return len < MRZ_MIN_MATCH ? 0 : len;
}

#define MRZ_SHORT_BYTES 128  // reach of the 8-lane short path, each direction

// Short path: up to 8 tag-equal candidates at once, 8 lanes (128 B) each,
// forwards and backwards in a single load round trip.  Per group (lane>>3):
// *len / *rev as single_match_len would return them, or *is_long when either
// direction ran through all 128 bytes (the caller then uses the striped path).
__device__ static void mrz_short_match_len(const uint8_t *__restrict__ buf, int64_t p0, int64_t op, bool valid,
                                           int64_t end, int64_t last_match, int lane, int64_t *len, int64_t *rev,
                                           bool *is_long) {
    const int i = lane & 7;
    const int g8 = lane & ~7;
    const int64_t off = (int64_t)i * 16;
    valid = valid && op < p0;
    const int64_t maxf = end - p0;
    int f_len = 0;
    bool f_full = false;
    uint4 fa, fb, ba, bb;
    const bool f_act = valid && off < maxf;
    if (f_act) {
        fa = mrz_ld16(buf + p0 + off);
        fb = mrz_ld16(buf + op + off);
    }
    const int64_t floor_p = last_match > 0 ? last_match : 0;
    int64_t maxb = p0 - floor_p;
    if (op < maxb) maxb = op;
    const bool b_act = valid && off < maxb;
    const bool b_wide = b_act && (op - off - 16 >= 0);
    if (b_wide) {
        ba = mrz_ld16(buf + p0 - off - 16);
        bb = mrz_ld16(buf + op - off - 16);
    }
    if (f_act) {
        const int64_t rem = maxf - off;
        const int lim = rem < 16 ? (int)rem : 16;
        const int d = mrz_first_diff16(fa, fb);
        f_len = d < lim ? d : lim;
        f_full = f_len == 16;
    }
    int b_len = 0;
    bool b_full = false;
    if (b_act) {
        const int64_t rem = maxb - off;
        const int lim = rem < 16 ? (int)rem : 16;
        int cnt;
        if (b_wide)
            cnt = mrz_top_equal16(ba, bb);
        else {
            cnt = 0;
            while (cnt < lim && buf[p0 - off - 1 - cnt] == buf[op - off - 1 - cnt]) cnt++;
        }
        b_len = cnt < lim ? cnt : lim;
        b_full = b_len == 16;
    }
    const mrz_u64 f_stop = __ballot(!f_full);
    const mrz_u64 b_stop = __ballot(!b_full);
    const unsigned fbits = (unsigned)(f_stop >> g8) & 0xffu;
    const unsigned bbits = (unsigned)(b_stop >> g8) & 0xffu;
    const int ffi = fbits ? __ffs((int)fbits) - 1 : 0;
    const int bfi = bbits ? __ffs((int)bbits) - 1 : 0;
    const int f_at = __shfl(f_len, g8 + ffi, MRZ_WAVE);
    const int b_at = __shfl(b_len, g8 + bfi, MRZ_WAVE);
    const int64_t fwd = ffi * 16 + f_at;
    const int64_t rv = bfi * 16 + b_at;
    *is_long = valid && (fbits == 0 || bbits == 0);
    *rev = valid ? rv : 0;
    const int64_t l = fwd + rv;
    *len = (valid && l >= MRZ_MIN_MATCH) ? l : 0;
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

__global__ __launch_bounds__(MRZ_SEQ_THREADS) void mrz_sequencer_kernel(mrz_seq_args a) {
    __shared__ int64_t pend_h[MRZ_CASCADE_MAX], pend_t[MRZ_CASCADE_MAX], pend_o[MRZ_CASCADE_MAX];
    __shared__ mrz_mailbox mbox;

    const int lane = threadIdx.x & 63;
    const int wave = mrz_uni((int)(threadIdx.x >> 6));
    const uint8_t *__restrict__ buf = a.buf;
    mrz_slot *tab = a.tab;
    mrz_seq_state *st = a.st;
    mrz_mailbox *mb = &mbox;

    if (st->finished || st->error) return;
    if (threadIdx.x == 0) {
        mb->seq = 0;
        mb->done = 0;
        mb->quit = 0;
    }
    __syncthreads();
    if (wave != 0) {
        mrz_helper_loop(buf, mb, wave, lane);
        return;
    }

    const int64_t end = st->end;
    int64_t p = st->p;
    int64_t cur_p = st->cur_p, cur_ofs = st->cur_ofs, cur_len = st->cur_len;
    int64_t last_match = st->last_match;
    int64_t min_mask = st->min_mask, tag_mask = st->tag_mask;
    int64_t count = st->count;
    const int64_t limit = st->limit;
    int64_t clean_ptr = st->clean_ptr;
    int64_t victim_round = st->victim_round;
    const int64_t max_chain = st->max_chain;
    const int64_t slot_mask = st->slot_mask;
    const int64_t nslots = slot_mask + 1;
    int64_t n_events = st->n_events;
    const int64_t event_cap = st->event_cap;
    int64_t inserts = st->inserts, tag_hits = st->tag_hits, tag_misses = st->tag_misses;

    const int64_t seg_start = a.seg_start;
    const int64_t seg_end = a.seg_start + a.seg_len;
    const int64_t lim = (end < seg_end - 1) ? end : seg_end - 1;  // last candidate position of this launch
    const int64_t nwords = (a.seg_len + 63) / 64;

    int64_t win_base = -1;
    mrz_u64 myword = 0;
    int mb_seq = 0;
    PROF_DECL;

    while (true) {
        PROF_START();
        // ---- next position > p whose bitmap bit is set -------------------
        int64_t q = -1;
        {
            int64_t pos = p + 1;
            if (pos < seg_start) pos = seg_start;
            while (pos <= lim) {
                const int64_t wb = seg_start + ((pos - seg_start) >> 12 << 12);
                if (wb != win_base) {
                    const int64_t idx = ((wb - seg_start) >> 6) + lane;
                    myword = idx < nwords ? a.bitmap[idx] : 0ull;
                    win_base = wb;
                }
                const int64_t lane_lo = wb + (int64_t)lane * 64;
                mrz_u64 w = myword;
                if (pos > lane_lo) {
                    const int64_t sh = pos - lane_lo;
                    w = sh >= 64 ? 0ull : (w >> sh) << sh;
                }
                const mrz_u64 m = __ballot(w != 0ull);
                if (!m) {
                    pos = wb + 4096;
                    continue;
                }
                const int fl = __ffsll((long long)m) - 1;
                const mrz_u64 wl = (mrz_u64)mrz_bcast64((int64_t)w, fl);
                q = wb + (int64_t)fl * 64 + (__ffsll((long long)wl) - 1);
                break;
            }
        }
        if (q < 0 || q > lim) {
            // nothing left for this launch
            if (p < lim) p = lim;
            break;
        }
        p = q;
        PROF_STOP(0);  // candidate discovery
        const int64_t t = mrz_uni64(a.tags[p - seg_start]);
        PROF_STOP(1);  // tag fetch
        PROF_COUNT(8);
        if ((t & min_mask) != min_mask) continue;  // src/rzip.c:573 with the mask reached by now

        // ---- one pass over the chain: find_best_match (:426-462) and, when this
        // position is inserted (:579), the probe walk of insert_hash ----------
        const bool do_insert = (t & tag_mask) == tag_mask;
        const int64_t better = (min_mask << 1) | 1;
        const int my_rank = mrz_ones_rank(t);
        int64_t mlen = 0, m_off = 0, m_rev = 0;
        bool ins_found = !do_insert;
        int64_t ins_slot = 0, occ_t = 0, occ_off = 0;
        int ins_kind = 0;
        if (do_insert) {
            inserts++;
            count++;
        }
        {
            const int64_t h0 = t & slot_mask;
            int64_t round = 0, victim_h = 0;
            for (int64_t b = 0;; b += MRZ_WAVE) {
                const int64_t s = (h0 + b + lane) & slot_mask;
                const mrz_slot e = tab[s];
                const bool empty = (e.off | e.t) == 0;
                const mrz_u64 m_empty = __ballot(empty);
                PROF_STOP(2);  // chain load
                const int first_empty = m_empty ? __ffsll((long long)m_empty) - 1 : MRZ_WAVE;
                if (!ins_found)
                    ins_found = mrz_insert_step(e, empty, t, my_rank, h0 + b, slot_mask, better, max_chain, &round,
                                                &victim_h, &count, &victim_round, &ins_slot, &ins_kind, &occ_t,
                                                &occ_off);
                const mrz_u64 m_same = __ballot(!empty && e.t == t) & mrz_low_mask(first_empty);
                PROF_STOP(3);  // classify
                PROF_COUNT(9);
                if (m_same) {
                    PROF_COUNT(10);
                    // tag-equal entries of this step, probe order, 8 per pass
                    mrz_u64 todo = m_same;
                    while (todo) {
                        // the g-th remaining entry goes to lane group g
                        const int g = lane >> 3;
                        mrz_u64 tmp = todo;
                        int src_lane = -1, npass = 0;
                        for (int k = 0; k < 8 && tmp; k++) {
                            const int sl = __ffsll((long long)tmp) - 1;
                            tmp &= tmp - 1;
                            if (k == g) src_lane = sl;
                            npass++;
                        }
                        const bool valid = src_lane >= 0;
                        const int rd = valid ? src_lane : 0;
                        const int lo = __shfl((int)(uint32_t)(uint64_t)e.off, rd, MRZ_WAVE);
                        const int hi = __shfl((int)(uint32_t)((uint64_t)e.off >> 32), rd, MRZ_WAVE);
                        const int64_t op = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
                        int64_t g_len, g_rev;
                        bool g_long;
                        mrz_short_match_len(buf, p, op, valid, end, last_match, lane, &g_len, &g_rev, &g_long);
                        for (int k = 0; k < npass; k++) {
                            int64_t ml = mrz_bcast64(g_len, 8 * k);
                            int64_t rv = mrz_bcast64(g_rev, 8 * k);
                            const int64_t opk = mrz_bcast64(op, 8 * k);
                            if (mrz_lane_read(g_long ? 1 : 0, 8 * k))
                                ml = mrz_long_match_len(buf, mb, &mb_seq, p, opk, end, last_match, &rv, lane);
                            if (ml) {  // first longest wins, :446-450
                                if (ml > mlen) {
                                    mlen = ml;
                                    m_off = opk - rv;
                                    m_rev = rv;
                                }
                                tag_hits++;
                            } else
                                tag_misses++;
                        }
                        todo = tmp;
                    }
                    PROF_STOP(4);  // match jobs
                }
                if (first_empty < MRZ_WAVE) break;
            }
        }

        // ---- insert + cull (:579-584) -------------------------------------
        if (do_insert) {
            int np = 0;
            int64_t it = t, io = p;
            while (true) {
                if (np >= MRZ_CASCADE_MAX) {  // cannot happen: every level has a strictly lower rank
                    if (lane == 0) st->error = 2;
                    np = 0;
                    break;
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
                                        &count, &victim_round, &ins_slot, &ins_kind, &occ_t, &occ_off))
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
            PROF_STOP(5);  // insert write-back / cascades
            if (count > limit) {
                // clean_one_from_hash (:305-328)
                while (true) {
                    const int64_t better2 = (min_mask << 1) | 1;
                    bool culled = false;
                    while (clean_ptr < nslots) {
                        const int64_t s = clean_ptr + lane;
                        mrz_slot e;
                        e.off = 0;
                        e.t = 0;
                        if (s < nslots) e = tab[s];
                        const bool hit = ((e.off | e.t) != 0) && ((e.t & better2) != better2);
                        const mrz_u64 m = __ballot(hit);
                        if (m) {
                            const int fl = __ffsll((long long)m) - 1;
                            clean_ptr += fl;
                            if (lane == fl) {
                                mrz_slot z;
                                z.off = 0;
                                z.t = 0;
                                tab[s] = z;
                            }
                            count--;
                            culled = true;
                            break;
                        }
                        clean_ptr += MRZ_WAVE;
                    }
                    if (culled) {
                        tag_mask = better2;
                        break;
                    }
                    min_mask = better2;
                    clean_ptr = 0;
                }
            }
        }

        PROF_STOP(6);  // cull
        // ---- lazy selection + emission (:586-599) -------------------------
        if (mlen > cur_len) {
            cur_p = p - m_rev;
            cur_len = mlen;
            cur_ofs = m_off;
        }
        if ((cur_len >= MRZ_GREAT_MATCH || p >= cur_p + MRZ_MIN_MATCH) && cur_len >= MRZ_MIN_MATCH) {
            if (n_events >= event_cap) {  // cannot happen: matches are >= 31 bytes and disjoint
                if (lane == 0) st->error = 1;
                break;
            }
            if (lane == 0) {
                mrz_event ev;
                ev.p = cur_p;
                ev.ofs = cur_ofs;
                ev.len = cur_len;
                a.events[n_events] = ev;
            }
            n_events++;
            last_match = cur_p + cur_len;
            cur_p = p = last_match;
            cur_len = 0;
        }
        PROF_STOP(7);  // select / emit
    }

    // release the helpers, then publish the state for the next segment's launch
    if (lane == 0) {
        mrz_mb_store(&mb->quit, 1);
        mrz_mb_store(&mb->seq, mb_seq + 1);
        st->p = p;
        st->cur_p = cur_p;
        st->cur_ofs = cur_ofs;
        st->cur_len = cur_len;
        st->last_match = last_match;
        st->min_mask = min_mask;
        st->tag_mask = tag_mask;
        st->count = count;
        st->clean_ptr = clean_ptr;
        st->victim_round = victim_round;
        st->n_events = n_events;
        st->inserts = inserts;
        st->tag_hits = tag_hits;
        st->tag_misses = tag_misses;
        st->finished = p >= end ? 1 : 0;
#ifdef MRZ_SEQ_PROFILE
        for (int k = 0; k < 16; k++) st->prof[k] += prof_acc[k];
#endif
    }
}

extern "C" hipError_t mrz_launch_sequencer(hipStream_t stream, const uint8_t *buf, mrz_slot *tab, const int64_t *tags,
                                           const mrz_u64 *bitmap, mrz_event *events, mrz_seq_state *st,
                                           int64_t seg_start, int64_t seg_len) {
    mrz_seq_args a;
    a.buf = buf;
    a.tab = tab;
    a.tags = tags;
    a.bitmap = bitmap;
    a.events = events;
    a.st = st;
    a.seg_start = seg_start;
    a.seg_len = seg_len;
    hipLaunchKernelGGL(mrz_sequencer_kernel, dim3(1), dim3(MRZ_SEQ_THREADS), 0, stream, a);
    return hipGetLastError();
}
