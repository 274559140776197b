// mrz_sequencer.hip -- the exact, order-preserving core of the rzip stage.
//
// hash_search's main loop (src/rzip.c:548-599) is a state machine whose table contents at position p depend
// on every earlier decision (skipped inserts inside emitted matches :596-598, probe-order evictions :264-297,
// in-place culling without tombstones :305-328, the process-lifetime victim_round :259), so it has to be
// replayed in position order to stay bit-exact.  What the GPU changes is the width of every step.
//
// Grid: block 0 is the SEQUENCER workgroup (MRZ_SEQ_WAVES waves, 512 threads), blocks 1.. are the helper
// workgroups of the compare farm (mrz_seq_common.h).  The sequencer alternates between two engines:
//
//   * the WIDE BATCH ENGINE (mrz_seq_wide.h): the next 512 candidates at once, one lane each -- probe walks
//     and 64-byte match probes speculatively against the table as it stands, then an in-order commit in
//     segments with workgroup-wide scans for the sequential quantities, the lazy-match fold as a prefix
//     maximum, emissions handled inside the batch.  This is what dense candidates (text, noise) run on;
//   * the COOPERATIVE PATH (mrz_seq_candidate): one candidate, wave 0 on its chain 64 slots per step, any
//     chain / match length (striped 4 KiB rounds, compare farm), cascades, evictions, sweep wrap, mask
//     promotion.  The wide engine hands over every lane it cannot prove; streams that are one long match
//     after another (every look-up finds max_chain_len entries tens of KiB long) run here directly,
//     chosen by a small predictor.
//
// Wave 0 owns the matcher state (wave-uniform, in SGPRs) and decides the mode; the other waves follow through
// LDS control words.  Emitted matches go to an event list; record encoding, literal gathering and the CRC are
// separate parallel kernels.
//
// Bound: latency -- a chain of dependent table probes, data probes and cross-CU hand-offs; DESIGN.md 4.2.
#include "mrz_seq_wide.h"

__global__ __launch_bounds__(MRZ_SEQ_THREADS) void mrz_sequencer_kernel(mrz_seq_args a) {
    __shared__ mrz_wide_lds wide;

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = mrz_uni(tid >> 6);
    mrz_seq_state *st = a.st;
    mrz_wide_lds *S = &wide;

    if (st->finished || st->error) return;
#if MRZ_HELPER_WGS > 0
    if (blockIdx.x != 0) {
        mrz_helper_wg(a.buf, (mrz_gmailbox *)a.gmailbox);
        return;
    }
#endif

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

#ifdef MRZ_SEQ_STATS
    int64_t stat[MRZ_ST_N];
    for (int k = 0; k < MRZ_ST_N; k++) stat[k] = 0;
#else
    int64_t *stat = nullptr;
#endif
    if (tid == 0) S->cmd = 0;
    __syncthreads();

    if (wave != 0) {
        // ---- waves 1..: wait for wave 0's commands; a wide batch is prepared by all waves together ------------
        int seen = 0;
        while (true) {
            int c;
            while ((c = mrz_uni(mrz_mb_load(&S->cmd))) == seen) __builtin_amdgcn_s_sleep(2);
            seen = c;
            if (mrz_uni(S->ctl[MRZ_CTL_MODE]) == 0) return;
            const mrz_lead Lw = S->Lp;
            mrz_wide_prep<MRZ_SEQ_WAVES>(C, Lw, S, a.tags, a.bitmap, seg_start, lim, nwords, mrz_uni(S->ctl[MRZ_CTL_WIDTH]),
                                         tid, lane, wave, stat);
        }
    }

    // ---- wave 0: owns the matcher state, decides the mode, commits ------------------------------------------
    int64_t win_base = -1;
    mrz_u64 myword = 0;
    bool ok = true;
    int cmd_seq = 0;
    int width = MRZ_W;         // batch width: small right after a match that swallowed the rest of a batch
    int low_yield = 0;         // consecutive batches that committed <= 2 candidates
    int seq_credit = 0;        // candidates to run through the cooperative path before batching again
    // Right after an emission the next candidate often has long matches again (repetitive input): a batch
    // would be formed, walked and probed only to stop at its first lane.  Four saturating counters, indexed by
    // the classes (short / >= GREAT_MATCH) of the last two emitted matches, learn whether that is so; when it
    // is, the first candidate after an emission goes straight through the cooperative path.
    bool after_emit = false;
    int emit_cls = 0;  // bit 0: last emitted match was great, bit 1: the one before
    int pred_long[4] = { 0, 0, 0, 0 };
    int stall = 0;  // iterations without any progress (cannot happen; keeps a logic error from hanging the GPU)
    int64_t stall_p = L.p, stall_ev = L.n_events;

    while (ok) {
        int64_t pos = L.p + 1;
        if (pos < seg_start) pos = seg_start;
        if (pos > lim) break;
        const bool first_after_emit = after_emit;
        const int64_t ev_before = L.n_events;
        long_seen = 0;
        int mode = 1;
        if (seq_credit > 0) {
            seq_credit--;  // a stretch where every candidate has long matches: one at a time is cheaper
            mode = 2;
        } else if (first_after_emit && pred_long[emit_cls] >= 2)
            mode = 2;
#ifdef MRZ_NO_BATCH
        mode = 2;
#endif
        int used = 0;
        if (mode == 1) {
            if (width <= 64) {
                // a narrow batch (right after a match that swallowed the rest of the previous one): this wave alone,
                // no workgroup barrier -- repetitive streams run on these
                mrz_wide_prep<1>(C, L, S, a.tags, a.bitmap, seg_start, lim, nwords, width, lane, lane, 0, stat);
            } else {
                if (lane == 0) {
                    S->Lp = L;
                    S->ctl[MRZ_CTL_MODE] = 1;
                    S->ctl[MRZ_CTL_WIDTH] = width;
                }
                cmd_seq++;
                if (lane == 0) mrz_mb_store(&S->cmd, cmd_seq);
                mrz_wide_prep<MRZ_SEQ_WAVES>(C, L, S, a.tags, a.bitmap, seg_start, lim, nwords, width, tid, lane, wave, stat);
            }
            mrz_wide_ret r;
            mrz_wide_commit(C, L, S, lane, stat, &r);
            used = r.used;
            ok = r.ok;
            if (r.long_seen) long_seen = 1;
            // a match that swallowed the rest of the batch: the next batch starts small and regrows
            if (r.skipped_out)
                width = 64;
            else if (width < MRZ_W)
                width = width * 4 > MRZ_W ? MRZ_W : width * 4;
            if (used <= 2 && r.coop_next) {
                if (++low_yield >= MRZ_LOW_YIELD_RUNS) {
                    seq_credit = MRZ_SEQ_CREDIT;
                    low_yield = 0;
                }
            } else
                low_yield = 0;
            if (r.coop_next && ok) mode = 2;
        }
        if (mode == 2 && ok) {
            // ---- cooperative path: the first candidate after L.p (4096-position bitmap window) ---------------
            PROF_T0();
            pos = L.p + 1;
            if (pos < seg_start) pos = seg_start;
            if (pos <= lim) {
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
                if (lim < lane_lo + 63) {
                    const int64_t keepbits = lim - lane_lo + 1;
                    w = keepbits <= 0 ? 0ull : (w & mrz_low_mask((int)keepbits));
                }
                const mrz_u64 any = __ballot(w != 0ull);
                if (!any) {
                    const int64_t nxt = wb + 4096;  // nothing left in this window
                    L.p = (nxt - 1 < lim) ? nxt - 1 : lim;
                } else {
                    ST_ADD(MRZ_ST_SEQ, 1);
                    const int fl = __ffsll((long long)any) - 1;
                    const mrz_u64 wl = (mrz_u64)mrz_bcast64((int64_t)w, fl);
                    L.p = wb + (int64_t)fl * 64 + (__ffsll((long long)wl) - 1);
                    const int64_t t = mrz_uni64(a.tags[L.p - seg_start]);
                    if ((t & L.min_mask) == L.min_mask)  // src/rzip.c:573 with the mask reached by now
                        ok = mrz_seq_candidate(C, L, &S->coop, t, lane, stat);
                    used = 1;
                }
            }
            PROF_ADD(MRZ_ST_T_SEQ);
        }
        if (first_after_emit) {
            const bool first_was_long = long_seen && used <= 1;
            int &c = pred_long[emit_cls];
            c = first_was_long ? (c < 3 ? c + 1 : 3) : (c > 0 ? c - 1 : 0);
        }
        if (L.p == stall_p && L.n_events == stall_ev) {
            if (++stall > 64) {
                if (lane == 0) st->error = 4;
                ok = false;
            }
        } else {
            stall = 0;
            stall_p = L.p;
            stall_ev = L.n_events;
        }
        after_emit = L.n_events != ev_before;
        if (after_emit) emit_cls = ((emit_cls << 1) & 2) | (L.last_len >= MRZ_GREAT_MATCH ? 1 : 0);
    }
    // send the other waves home
    if (lane == 0) S->ctl[MRZ_CTL_MODE] = 0;
    cmd_seq++;
    if (lane == 0) mrz_mb_store(&S->cmd, cmd_seq);

    // release the helpers, then publish the state for the next segment's launch
#if MRZ_HELPER_WGS > 0
    if (lane == 0 && C.gmb) mrz_g_storeu(&C.gmb->quit, 1ull);
#endif
    if (lane == 0) {
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

extern "C" hipError_t mrz_launch_sequencer(hipStream_t stream, const uint8_t *buf, mrz_slot *tab, const int64_t *tags,
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
    hipLaunchKernelGGL(mrz_sequencer_kernel, dim3(1 + a.n_helpers), dim3(MRZ_SEQ_THREADS), 0, stream, a);
    return hipGetLastError();
}

// helper workgroups a launch on `device` should carry by default: about one per CU, leaving the leader's CU and a
// few for co-resident kernels free (MRZ_FARM_WGS overrides)
extern "C" int mrz_sequencer_default_helpers(int device) {
#if MRZ_HELPER_WGS == 0
    (void)device;
    return 0;
#else
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) cus = 0;
    int n = cus > 16 ? (cus - 16) * MRZ_HELPERS_PER_CU : 0;
    const char *e = getenv("MRZ_FARM_WGS");
    if (e) n = atoi(e);
    if (n > MRZ_HELPER_WGS) n = MRZ_HELPER_WGS;
    if (n < 0) n = 0;
    return n;
#endif
}

extern "C" size_t mrz_sequencer_mailbox_size(void) { return MRZ_HELPER_WGS > 0 ? sizeof(mrz_gmailbox) : 0; }
