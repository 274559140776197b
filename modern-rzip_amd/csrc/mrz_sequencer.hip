// mrz_sequencer.hip -- the exact, order-preserving core of the rzip stage.
//
// hash_search's main loop (src/rzip.c:548-599) is a state machine whose table
// contents at position p depend on every earlier decision (skipped inserts
// inside emitted matches :596-598, probe-order evictions :264-297, in-place
// culling without tombstones :305-328, the process-lifetime victim_round :259),
// so it has to be replayed in position order to stay bit-exact.  What the GPU
// changes is the width of every step: one wavefront walks the state machine and
// each step is 64 lanes wide --
//   * candidate discovery: the tag-scan bitmap is read 4096 positions per load
//     and the next candidate found with one ballot (the `continue` at :573 means
//     ONLY positions passing minimum_tag_mask run the loop body, emit test
//     included);
//   * find_best_match (:426-462): 64 consecutive slots (1 KiB, coalesced) per
//     probe step, ballot for first-empty and for tag-equal lanes;
//   * single_match_len (:372-397): 64 lanes x 16 B per step forwards and
//     backwards, ballot + ffs for the first mismatch;
//   * insert_hash (:256-301): the probe walk classifies 64 occupants at once
//     (empty / due-for-culling / lower-ranked / same-tag round counting via
//     popcount of the ballot), cascades of displaced occupants are collected and
//     written back innermost-first exactly like the reference's recursion;
//   * clean_one_from_hash (:305-328): 64 slots per sweep step.
// All control values are wave-uniform.  Emitted matches go to an event list;
// record encoding, literal gathering and the CRC are separate parallel kernels.
//
// Bound: latency (dependent 1 KiB probes into the 64 MiB table, which sits in
// the 256 MiB Infinity Cache / this XCD's L2), not HBM bandwidth.
#include "mrz_device.h"

#define MRZ_SEQ_THREADS 64
#define MRZ_FWD_UNROLL 4

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

// ---- single_match_len (src/rzip.c:372-397), wave-wide --------------------
__device__ static int64_t mrz_wave_match_len(const uint8_t *__restrict__ buf, int64_t p0, int64_t op, int64_t end,
                                             int64_t last_match, int64_t *rev_out, int lane) {
    *rev_out = 0;
    if (op >= p0) return 0;
    // forward: while (p < end && buf[p] == buf[op])
    const int64_t maxf = end - p0;
    int64_t fwd = 0;
    if (maxf > 0) {
        for (int64_t base = 0;; base += (int64_t)MRZ_FWD_UNROLL * 1024) {
            uint4 a[MRZ_FWD_UNROLL], b[MRZ_FWD_UNROLL];
#pragma unroll
            for (int j = 0; j < MRZ_FWD_UNROLL; j++) {
                const int64_t off = base + j * 1024 + lane * 16;
                if (off < maxf) {
                    a[j] = mrz_ld16(buf + p0 + off);
                    b[j] = mrz_ld16(buf + op + off);
                }
            }
            bool done = false;
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
            if (done) break;
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

// ---- one probe walk of insert_hash (src/rzip.c:262-297) --------------------
// Returns the slot the walk stops at; *displace = the occupant must be
// re-inserted first (the lesser_bitness case).  Mutates count / victim_round
// exactly where the reference does.
__device__ static int64_t mrz_wave_insert_walk(const mrz_slot *tab, int64_t t, int64_t slot_mask,
                                               int64_t better, int64_t max_chain, int64_t *count,
                                               int64_t *victim_round, bool *displace, int64_t *occ_t,
                                               int64_t *occ_off, int lane) {
    const int64_t h0 = t & slot_mask;
    const int my_rank = mrz_ones_rank(t);
    int64_t round = 0, victim_h = 0;
    *displace = false;
    for (int64_t b = 0;; b += MRZ_WAVE) {
        const int64_t s = (h0 + b + lane) & slot_mask;
        const mrz_slot e = tab[s];
        const bool empty = (e.off | e.t) == 0;
        const bool minbit = !empty && ((e.t & better) != better);
        const bool lesser = !empty && (mrz_ones_rank(e.t) < my_rank);
        const bool same = !empty && (e.t == t);
        const mrz_u64 m_stop = __ballot(empty || minbit || lesser);
        const int first_stop = m_stop ? __ffsll((long long)m_stop) - 1 : MRZ_WAVE;
        const mrz_u64 m_same = __ballot(same) & mrz_low_mask(first_stop);
        const int cnt = __popcll(m_same);
        // victim_h is latched at the same-tag entry whose round == victim_round
        const int64_t kv = *victim_round - round;
        if (kv >= 0 && kv < cnt) victim_h = (h0 + b + mrz_nth_set(m_same, (int)kv)) & slot_mask;
        const int64_t k = max_chain - round;  // this many more same-tag entries trip the limit
        if (k <= cnt) {
            // chain limit reached before any other stop: evict the victim (:284-291)
            *count -= 1;
            int64_t vr = *victim_round + 1;
            if (vr == max_chain) vr = 0;
            *victim_round = vr;
            return victim_h;
        }
        if (first_stop < MRZ_WAVE) {
            const int64_t hs = (h0 + b + first_stop) & slot_mask;
            const int64_t et = mrz_bcast64(e.t, first_stop);
            const int64_t eo = mrz_bcast64(e.off, first_stop);
            if ((eo | et) == 0) return hs;            // empty slot
            if ((et & better) != better) {            // due for culling: overwrite (:267-270)
                *count -= 1;
                return hs;
            }
            *displace = true;                         // outranked occupant (:275-278)
            *occ_t = et;
            *occ_off = eo;
            return hs;
        }
        round += cnt;
    }
}

#define MRZ_CASCADE_MAX 64

__global__ __launch_bounds__(MRZ_SEQ_THREADS) void mrz_sequencer_kernel(mrz_seq_args a) {
    __shared__ int64_t pend_h[MRZ_CASCADE_MAX], pend_t[MRZ_CASCADE_MAX], pend_o[MRZ_CASCADE_MAX];

    const int lane = threadIdx.x;
    const uint8_t *__restrict__ buf = a.buf;
    mrz_slot *tab = a.tab;
    mrz_seq_state *st = a.st;

    if (st->finished || st->error) return;

    const int64_t n = st->n, end = st->end;
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
    (void)n;

    const int64_t seg_start = a.seg_start;
    const int64_t seg_end = a.seg_start + a.seg_len;
    const int64_t lim = (end < seg_end - 1) ? end : seg_end - 1;  // last candidate position of this launch
    const int64_t nwords = (a.seg_len + 63) / 64;

    int64_t win_base = -1;
    mrz_u64 myword = 0;

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

        // ---- find_best_match (:426-462) -----------------------------------
        int64_t mlen = 0, m_off = 0, m_rev = 0;
        {
            const int64_t h0 = t & slot_mask;
            for (int64_t b = 0;; b += MRZ_WAVE) {
                const int64_t s = (h0 + b + lane) & slot_mask;
                const mrz_slot e = tab[s];
                const bool empty = (e.off | e.t) == 0;
                const mrz_u64 m_empty = __ballot(empty);
                const int first_empty = m_empty ? __ffsll((long long)m_empty) - 1 : MRZ_WAVE;
                mrz_u64 m_same = __ballot(!empty && e.t == t) & mrz_low_mask(first_empty);
                while (m_same) {
                    const int hl = __ffsll((long long)m_same) - 1;
                    m_same &= m_same - 1;
                    const int64_t op = mrz_bcast64(e.off, hl);
                    int64_t rev = 0;
                    const int64_t ml = mrz_wave_match_len(buf, p, op, end, last_match, &rev, lane);
                    if (ml) {
                        if (ml > mlen) {
                            mlen = ml;
                            m_off = op - rev;
                            m_rev = rev;
                        }
                        tag_hits++;
                    } else
                        tag_misses++;
                }
                if (first_empty < MRZ_WAVE) break;
            }
        }

        // ---- insert + cull (:579-584) -------------------------------------
        if ((t & tag_mask) == tag_mask) {
            inserts++;
            count++;
            {
                const int64_t better = (min_mask << 1) | 1;
                int np = 0;
                int64_t it = t, io = p;
                while (true) {
                    bool displace;
                    int64_t occ_t = 0, occ_off = 0;
                    const int64_t hs = mrz_wave_insert_walk(tab, it, slot_mask, better, max_chain, &count,
                                                            &victim_round, &displace, &occ_t, &occ_off, lane);
                    if (np >= MRZ_CASCADE_MAX) {  // cannot happen: every level has a strictly lower rank
                        if (lane == 0) st->error = 2;
                        np = 0;
                        break;
                    }
                    if (lane == 0) {
                        pend_h[np] = hs;
                        pend_t[np] = it;
                        pend_o[np] = io;
                    }
                    np++;
                    if (!displace) break;
                    it = occ_t;
                    io = occ_off;
                }
                // write back innermost-first (the recursion's return order)
                while (np-- > 0) {
                    const int64_t hs = pend_h[np];
                    if (lane == 0) {
                        mrz_slot w;
                        w.off = pend_o[np];
                        w.t = pend_t[np];
                        tab[hs] = w;
                    }
                }
            }
            if (count > limit) {
                // clean_one_from_hash (:305-328)
                while (true) {
                    const int64_t better = (min_mask << 1) | 1;
                    bool culled = false;
                    while (clean_ptr < nslots) {
                        const int64_t s = clean_ptr + lane;
                        mrz_slot e;
                        e.off = 0;
                        e.t = 0;
                        if (s < nslots) e = tab[s];
                        const bool hit = ((e.off | e.t) != 0) && ((e.t & better) != better);
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
                        if (clean_ptr > nslots) clean_ptr = nslots;
                        tag_mask = better;
                        break;
                    }
                    min_mask = better;
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

    if (lane == 0) {
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
