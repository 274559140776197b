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
//   waves 1..W-1 ("helpers") wait on an LDS mailbox.  When a look-up finds
//   tag-equal entries the leader posts their offsets and every wave (leader
//   included) runs single_match_len (:372-397) for one candidate: 64 lanes x
//   16 B x 8 pieces per step forwards (ballot + ffs for the first mismatch),
//   64 x 16 B backwards.  Results come back through LDS and are folded in
//   probe order, so ties resolve exactly as in the reference.
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
#define MRZ_FWD_UNROLL 8
#define MRZ_CASCADE_MAX 64
#define MRZ_MAX_JOBS 64

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

// LDS mailbox between the leader and the helper waves
struct mrz_mailbox {
    int64_t p0, end, last_match;    // common to all jobs of a round
    int64_t op[MRZ_MAX_JOBS];       // candidate offsets (probe order)
    int64_t len[MRZ_MAX_JOBS];      // results
    int64_t rev[MRZ_MAX_JOBS];
    int njobs;
    int seq;                        // bumped by the leader for every round; helpers wait on it
    int done;                       // helpers add the number of jobs they finished
    int quit;
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

// ---- single_match_len (src/rzip.c:372-397), one wave per candidate ----------
__device__ static int64_t mrz_wave_match_len(const uint8_t *__restrict__ buf, int64_t p0, int64_t op, int64_t end,
                                             int64_t last_match, int64_t *rev_out, int lane) {
    *rev_out = 0;
    if (op >= p0) return 0;
    // forward: while (p < end && buf[p] == buf[op])
    const int64_t maxf = end - p0;
    int64_t fwd = 0;
    if (maxf > 0) {
        // first step: one piece per lane (most candidates end within 1 KiB)
        int64_t base = 0;
        bool done = false;
        {
            const int64_t off = (int64_t)lane * 16;
            int lane_len = 0;
            bool full = false;
            if (off < maxf) {
                const int64_t rem = maxf - off;
                const int lim = rem < 16 ? (int)rem : 16;
                const int d = mrz_first_diff16(mrz_ld16(buf + p0 + off), mrz_ld16(buf + op + off));
                lane_len = d < lim ? d : lim;
                full = lane_len == 16;
            }
            const mrz_u64 stop = __ballot(!full);
            if (stop) {
                const int fl = __ffsll((long long)stop) - 1;
                fwd = (int64_t)fl * 16 + mrz_lane_read(lane_len, fl);
                done = true;
            }
            base = 1024;
        }
        while (!done) {
            uint4 a[MRZ_FWD_UNROLL], b[MRZ_FWD_UNROLL];
#pragma unroll
            for (int j = 0; j < MRZ_FWD_UNROLL; j++) {
                const int64_t off = base + j * 1024 + lane * 16;
                if (off < maxf) {
                    a[j] = mrz_ld16(buf + p0 + off);
                    b[j] = mrz_ld16(buf + op + off);
                }
            }
#pragma unroll
            for (int j = 0; j < MRZ_FWD_UNROLL; j++) {
                if (done) continue;
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
                    fwd = base + j * 1024 + (int64_t)fl * 16 + mrz_lane_read(lane_len, fl);
                    done = true;
                }
            }
            base += (int64_t)MRZ_FWD_UNROLL * 1024;
        }
    }
    // backward: while (p > max(0,last_match) && op > 0 && buf[op-1] == buf[p-1])
    const int64_t floor_p = last_match > 0 ? last_match : 0;
    int64_t maxb = p0 - floor_p;
    if (op < maxb) maxb = op;
    int64_t rev = 0;
    if (maxb > 0) {
        for (int64_t base = 0;; base += 1024) {
            const int64_t off = base + lane * 16;
            int lane_len = 0;
            bool full = false;
            if (off < maxb) {
                const int64_t rem = maxb - off;
                const int lim = rem < 16 ? (int)rem : 16;
                int cnt;
                if (op - off - 16 >= 0) {
                    const uint4 a = mrz_ld16(buf + p0 - off - 16);
                    const uint4 b = mrz_ld16(buf + op - off - 16);
                    cnt = mrz_top_equal16(a, b);
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
                rev = base + (int64_t)fl * 16 + mrz_lane_read(lane_len, fl);
                break;
            }
        }
    }
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

// helper waves: serve match-extension rounds until the leader says quit
__device__ static void mrz_helper_loop(const uint8_t *__restrict__ buf, mrz_mailbox *mb, int wave, int lane) {
    int seen = 0;
    while (true) {
        int s;
        while ((s = mrz_uni(mrz_mb_load(&mb->seq))) == seen) __builtin_amdgcn_s_sleep(2);
        seen = s;
        if (mrz_uni(mrz_mb_load(&mb->quit))) return;
        const int nj = mrz_uni(mb->njobs);
        const int64_t p0 = mrz_uni64(mb->p0), end = mrz_uni64(mb->end), lm = mrz_uni64(mb->last_match);
        int mine = 0;
        for (int j = wave; j < nj; j += MRZ_SEQ_WAVES) {
            int64_t rev = 0;
            const int64_t ml = mrz_wave_match_len(buf, p0, mrz_uni64(mb->op[j]), end, lm, &rev, lane);
            if (lane == 0) {
                mb->len[j] = ml;
                mb->rev[j] = rev;
            }
            mine++;
        }
        if (mine && lane == 0) mrz_mb_add(&mb->done, mine);
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
        mb->njobs = 0;
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

    while (true) {
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
        const int64_t t = mrz_uni64(a.tags[p - seg_start]);
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
                const int first_empty = m_empty ? __ffsll((long long)m_empty) - 1 : MRZ_WAVE;
                if (!ins_found)
                    ins_found = mrz_insert_step(e, empty, t, my_rank, h0 + b, slot_mask, better, max_chain, &round,
                                                &victim_h, &count, &victim_round, &ins_slot, &ins_kind, &occ_t,
                                                &occ_off);
                const mrz_u64 m_same = __ballot(!empty && e.t == t) & mrz_low_mask(first_empty);
                if (m_same) {
                    // post the tag-equal entries of this step, probe order
                    const int nj = __popcll(m_same);
                    const int my_idx = __popcll(m_same & mrz_low_mask(lane));
                    if ((m_same >> lane) & 1) mb->op[my_idx] = e.off;
                    if (lane == 0) {
                        mb->p0 = p;
                        mb->end = end;
                        mb->last_match = last_match;
                        mb->njobs = nj;
                        mb->done = 0;
                    }
                    // jobs 0, W, 2W, .. stay with the leader
                    const int leader_jobs = (nj + MRZ_SEQ_WAVES - 1) / MRZ_SEQ_WAVES;
                    const int helper_jobs = nj - leader_jobs;
                    if (helper_jobs) {
                        mb_seq++;
                        if (lane == 0) mrz_mb_store(&mb->seq, mb_seq);
                    }
                    for (int j = 0; j < nj; j += MRZ_SEQ_WAVES) {
                        int64_t rev = 0;
                        const int64_t op = mrz_uni64(mb->op[j]);
                        const int64_t ml = mrz_wave_match_len(buf, p, op, end, last_match, &rev, lane);
                        if (lane == 0) {
                            mb->len[j] = ml;
                            mb->rev[j] = rev;
                        }
                    }
                    if (helper_jobs)
                        while (mrz_uni(mrz_mb_load(&mb->done)) < helper_jobs) __builtin_amdgcn_s_sleep(1);
                    // fold in probe order (first longest wins, :446-450)
                    for (int j = 0; j < nj; j++) {
                        const int64_t ml = mrz_uni64(mb->len[j]);
                        if (ml) {
                            if (ml > mlen) {
                                const int64_t rv = mrz_uni64(mb->rev[j]);
                                mlen = ml;
                                m_off = mrz_uni64(mb->op[j]) - rv;
                                m_rev = rv;
                            }
                            tag_hits++;
                        } else
                            tag_misses++;
                    }
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
