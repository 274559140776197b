// mrz_sequencer.hip -- the exact, order-preserving core of the rzip stage.
//
// hash_search's main loop (src/rzip.c:548-599) is a state machine whose table contents at position p depend
// on every earlier decision (skipped inserts inside emitted matches :596-598, probe-order evictions :264-297,
// in-place culling without tombstones :305-328, the process-lifetime victim_round :259), so it has to be
// replayed in position order to stay bit-exact.  What the GPU changes is the width of every step.
//
// Grid: blocks x, x + 8, x + 16, ... (x = the launch's XCD residue; up to MRZ_SEQ_WGS of them; one XCD under round-robin
// placement, checked at run time) are SEQUENCER workgroups of MRZ_SEQ_WAVES waves (512 threads); the other blocks are the helper workgroups of the
// compare farm (mrz_seq_common.h).  The sequencer workgroups run the WIDE BATCH ENGINE (mrz_seq_wide.h) and take
// turns:
//
//   * batches are fixed windows of the front end's candidate list; a workgroup PREPARES its batch -- probe walks, conflicts inside the
//     batch, overlay walks, 64-byte match probes, all read-only -- while the batches before it are being committed;
//   * at its TURN (a token in device memory) it loads the matcher state, brings the prepared lanes up to date
//     (lanes a match has covered since, table blocks written since: the write log; the cull window), commits in
//     order -- clean stretches by all waves at once, the rest by wave 0 in windows of 64 lanes with the lazy-match
//     fold as a prefix maximum -- and hands state and token on;
//   * a lane the batch cannot prove (stale, long chain, deep cascade, sweep wrap, mask promotion) goes through the
//     COOPERATIVE PATH (mrz_seq_candidate): one candidate, wave 0 on its chain 64 slots per step, any chain / match
//     length (striped 4 KiB rounds, compare farm).
//
// Streams that are one long match after another run on the narrow engine instead (mrz_seq_narrow.hip, picked per
// segment by the host).  Emitted matches go to an event list; record encoding, literal gathering and the CRC are
// separate parallel kernels.
//
// Bound: latency -- a chain of dependent table probes, data probes and cross-CU hand-offs; DESIGN.md 4.2.
#include <cstddef>
#ifdef MRZ_DBG_HITS
#define MRZ_DBG_HITS_COOP(q, h, m) atomicAdd(&mrz_dbg_hits[(q)], ((unsigned)(h) << 16) | (unsigned)(m) | (1u << 24))
extern __device__ unsigned *mrz_dbg_hits;
#endif
#include "mrz_seq_wide.h"

// ---- several sequencer workgroups taking turns ---------------------------------------------------------------
// The preparation of a batch (walks, conflicts, overlay walks, match probes) is most of its cost and only READS the
// table, so it can run while earlier batches are still being committed.  Up to MRZ_SEQ_WGS sequencer workgroups --
// all on ONE XCD, so that they share an L2 (blocks 0, 8, 16, ...: checked at run time through HW_REG_XCC_ID) -- take
// the batches of the position stream round robin: batch b is a window of positions fixed by (epoch base, b); workgroup
// j prepares its batch against the table as it sees it, then waits for its TURN (token == b in device memory),
// brings the prepared lanes up to date (mrz_wide_precommit: a lane that read a 64-slot block written since the
// preparation began is stale), commits, writes the matcher state back and passes the token on.  Commits are strictly
// in batch order, so the result is the one workgroup's result.  Whenever a batch does not end where the next window
// begins (a match swallowed the following windows, the masks moved, a window overflowed its 512 lanes) the committer
// opens a new epoch: windows are re-based at the matcher's position and batches prepared under the old epoch are
// prepared again when their turn comes.
#ifndef MRZ_SEQ_WGS
#define MRZ_SEQ_WGS 4
#endif
#ifndef MRZ_FILL_NUM
#define MRZ_FILL_NUM 14  // a window is sized for 14/16 of the lanes
#endif
#define MRZ_TURN_SPIN_LIMIT (1ll << 27)

// words of mrz_wide_shared.hand: the matcher state (mrz_lead, 16 words) and what goes with it
enum { MRZ_H_L = 0, MRZ_H_GSEQ = 16, MRZ_H_FARM, MRZ_H_GNW, MRZ_H_PW, MRZ_H_SMALL, MRZ_H_EPOCH, MRZ_H_BIDX, MRZ_H_BBATCH,
       MRZ_H_BPOS, MRZ_H_N = 32 };
static_assert(sizeof(mrz_lead) == 16 * 8, "mrz_lead is handed over as 16 words");

struct mrz_wide_shared {          // device memory, zeroed by the host before every launch
    // one 128-byte line of control words (polled with agent-scope loads)
    unsigned long long token;     // batches committed so far = the batch whose turn it is
    unsigned long long quit;      // the launch is over (segment end, error)
    unsigned long long census;    // sequencer workgroups that have reported their XCC id
    unsigned long long n_active;  // set by workgroup 0 once the state below is valid: workgroups taking part
    int xcc[8];
    int active[8];                // compact index of sequencer workgroup j, or -1
    unsigned long long pad_[4];
    // two lines handed from committer to committer: written by ONE store instruction of the committing wave (a word
    // per lane, agent scope) before it passes the token on, read by ONE agent-scope load instruction of the next
    // committer after it has seen the token -- served by the L2, never by a line some earlier load left in an L1.
    //   [0, 16)  mrz_lead    [16] farm round counter   [17] farm hint   [18] helpers seen   [19] pw: list entries per
    //   batch window   [20] batches in a row that came out less than half full   [21] epoch   [22] base_idx
    //   [23] base_batch: batch b of the epoch covers list entries [base_idx + (b - base_batch) * pw, + pw)   [24] base_pos:
    //   ... and nothing before this position (where the epoch began)
    unsigned long long hand[MRZ_H_N];
};
static_assert(offsetof(mrz_wide_shared, hand) == 128, "the hand-over block has its own cache lines");


#ifdef __HIP_DEVICE_COMPILE__
#define MRZ_ACQUIRE_AGENT() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#define MRZ_RELEASE_AGENT() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define MRZ_XCC_ID() (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15)
#define MRZ_WAIT_STORES() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define MRZ_WAIT_STORES() ((void)0)
#define MRZ_ACQUIRE_AGENT() ((void)0)
#define MRZ_RELEASE_AGENT() ((void)0)
#define MRZ_XCC_ID() 0
#endif

__device__ __forceinline__ int mrz_entries_per_batch(int64_t list_mask, int64_t min_mask) {
    // the list holds the positions that passed list_mask; 2^-(k - kl) of them still pass min_mask: aim at 7/8 of the lanes
    int d = __popcll((unsigned long long)min_mask) - __popcll((unsigned long long)list_mask);
    if (d < 0) d = 0;
    long long ent = (long long)(MRZ_W * MRZ_FILL_NUM / 16) << (d < 8 ? d : 8);
    if (ent > MRZ_RAW_PER_THREAD * MRZ_W) ent = MRZ_RAW_PER_THREAD * MRZ_W;
    return (int)(ent < 1 ? 1 : ent);
}

__global__ __launch_bounds__(MRZ_SEQ_THREADS) void mrz_sequencer_kernel(mrz_seq_args a, mrz_wide_shared *G,
                                                                        unsigned *wlog, int want_wgs) {
#ifdef MRZ_EMU_LDS_PER_BLOCK
    // (the CPU emulator of the test suite keeps `__shared__` in one static copy: one per sequencer workgroup here)
    static mrz_wide_lds wide_all[MRZ_SEQ_WGS];
    mrz_wide_lds *S = &wide_all[(blockIdx.x / 8) % MRZ_SEQ_WGS];
#else
    __shared__ mrz_wide_lds wide;
    mrz_wide_lds *S = &wide;
#endif
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = mrz_uni(tid >> 6);
    mrz_seq_state *st = a.st;
#ifdef MRZ_EMU_POISON_LDS
    // (test emulator: LDS is not zero when a workgroup starts on the GPU)
    if (tid == 0) memset((void *)S, MRZ_EMU_POISON_LDS, sizeof(*S));
    __syncthreads();
#endif

    if (st->finished || st->error) return;
    // the mask has reached the deep engine's regime (a launch queued before the host knew): nothing is sequenced here,
    // the front end scans this stretch again (under the tighter mask) for the deep engine's launch
    if (a.deep_bits > 0 && __popcll((unsigned long long)st->min_mask) >= a.deep_bits) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const int64_t pt = ((st->p + 1) >> MRZ_TILE_SHIFT) << MRZ_TILE_SHIFT;
            if (st->seg_end > st->seg_start && pt < st->scan_next) st->scan_next = pt > st->seg_start ? pt : st->seg_start;
        }
        return;
    }
    // which role: blocks 0, 8, 16, ... are sequencer workgroups (one XCD), the others compare-farm helpers
    int wgs = want_wgs < 1 ? 1 : (want_wgs > MRZ_SEQ_WGS ? MRZ_SEQ_WGS : want_wgs);
    const int bx = (int)blockIdx.x;
    const int xcd = a.xcd & 7;
    while (wgs > 1 && (int)gridDim.x < 8 * (wgs - 1) + xcd + 1) wgs--;
    const bool is_seq = (bx % 8 == xcd) && (bx / 8 < wgs) && (bx < (int)gridDim.x);
#if MRZ_HELPER_WGS > 0
    if (!is_seq) {
        if (a.gmailbox) mrz_helper_wg(a.buf, (mrz_gmailbox *)a.gmailbox);  // (no mailbox: a launch without a farm)
        return;
    }
#else
    if (!is_seq) return;
#endif
    const int j = bx / 8;

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
    C.mb = nullptr;
    C.mb_seq = nullptr;
    mrz_cands K;  // the segment the front end has laid out: its geometry comes with the matcher state
    K.cand = a.cand;
    K.tile_off = a.tile_off;
    K.bitmap = a.bitmap;
    K.seg_start = st->seg_start;
    K.seg_end = st->seg_end;
    K.n = st->n_cand;
    const int64_t seg_start = K.seg_start;
    const int64_t seg_end = K.seg_end;
    if (seg_end <= seg_start) return;  // (nothing was scanned: the chunk is done)
    const int64_t lim = (C.end < seg_end - 1) ? C.end : seg_end - 1;  // last candidate position of this launch
    const int64_t list_mask = st->list_mask;
#ifdef MRZ_SEQ_STATS
    int64_t stat[MRZ_ST_N];
    for (int k = 0; k < MRZ_ST_N; k++) stat[k] = 0;
#else
    int64_t *stat = nullptr;
#endif

    // ---- census: which sequencer workgroups share workgroup 0's XCD ----------------------------------------------
    int n_act = 1, my = 0;
    if (wgs > 1) {
        if (tid == 0) {
            G->xcc[j] = MRZ_XCC_ID();
            __hip_atomic_fetch_add(&G->census, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (j == 0 && wave == 0) {
        int n = 1;
        if (lane == 0) {
            if (wgs > 1) {
                long long spins = 0;  // the others start within microseconds; late ones are left out
                while (__hip_atomic_load(&G->census, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)wgs &&
                       spins++ < 2000)
                    __builtin_amdgcn_s_sleep(4);
                const int seen = (int)__hip_atomic_load(&G->census, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                G->active[0] = 0;
                for (int k = 1; k < wgs; k++) {
                    const bool okk = seen >= wgs && G->xcc[k] == G->xcc[0];
                    G->active[k] = okk ? n : -1;
                    if (okk) n++;
                }
            } else
                G->active[0] = 0;
        }
        // the matcher state, the first epoch: the list from the first entry behind the matcher's position
        int64_t pos = st->p + 1;
        if (pos < seg_start) pos = seg_start;
        const int64_t bidx = mrz_cand_lower_bound(K, pos, lane);
        if (lane == 0) {
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
            *(mrz_lead *)&S->hand[MRZ_H_L] = L;
            S->hand[MRZ_H_GSEQ] = 0;
            S->hand[MRZ_H_FARM] = 0;
            S->hand[MRZ_H_GNW] = 0;
            S->hand[MRZ_H_PW] = (unsigned long long)mrz_entries_per_batch(list_mask, L.min_mask);
            S->hand[MRZ_H_SMALL] = 0;
            S->hand[MRZ_H_EPOCH] = 1;
            S->hand[MRZ_H_BIDX] = (unsigned long long)bidx;
            S->hand[MRZ_H_BBATCH] = 0;
            S->hand[MRZ_H_BPOS] = (unsigned long long)pos;
            for (int k = 0; k <= MRZ_H_BPOS; k++)
                __hip_atomic_store(&G->hand[k], S->hand[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            MRZ_RELEASE_AGENT();
            __hip_atomic_store(&G->n_active, (unsigned long long)n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) {
        long long spins = 0;
        unsigned long long n = 0;
        while ((n = __hip_atomic_load(&G->n_active, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == 0 && spins++ < (1ll << 22))
            __builtin_amdgcn_s_sleep(4);
        S->ctl[0] = (int)n;
        S->ctl[1] = n ? G->active[j] : -1;
        S->ctl[8] = 0;
    }
    __syncthreads();
    n_act = mrz_uni(S->ctl[0]);
    my = mrz_uni(S->ctl[1]);
    if (n_act <= 0 || my < 0) return;  // not taking part
    const bool multi = n_act > 1;
    const int64_t hint_p0 = (int64_t)__hip_atomic_load(&G->hand[MRZ_H_L + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t hint_ev0 = (int64_t)__hip_atomic_load(&G->hand[MRZ_H_L + 10], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    mrz_lead L;  // (every wave keeps a copy, loaded at each turn; only the committing wave changes it)
    memset(&L, 0, sizeof(L));
    unsigned long long b = (unsigned long long)my;
    bool ok = true;
    while (true) {
        // ---- snapshot: token, epoch, masks (read as one: retried while a commit is writing them) ---------------
        if (tid == 0) {
            long long e = 0, bw = 0, bb = 0, mm = 0, tm = 0, bp = 0;
            int pw = 1;
            unsigned long long t0 = 0;
            for (int tries = 0; tries < 64; tries++) {
                t0 = __hip_atomic_load(&G->token, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                e = (long long)__hip_atomic_load(&G->hand[MRZ_H_EPOCH], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bw = (long long)__hip_atomic_load(&G->hand[MRZ_H_BIDX], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bb = (long long)__hip_atomic_load(&G->hand[MRZ_H_BBATCH], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bp = (long long)__hip_atomic_load(&G->hand[MRZ_H_BPOS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pw = (int)__hip_atomic_load(&G->hand[MRZ_H_PW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mm = (long long)__hip_atomic_load(&G->hand[MRZ_H_L + 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                tm = (long long)__hip_atomic_load(&G->hand[MRZ_H_L + 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long t1 = __hip_atomic_load(&G->token, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (t1 == t0) break;
            }
            S->snap64[0] = (long long)t0;
            S->snap64[1] = e;
            S->snap64[2] = bw;
            S->snap64[3] = bb;
            S->snap64[4] = pw;
            S->snap64[5] = mm;
            S->snap64[6] = tm;
            S->snap64[8] = bp;
            S->snap64[7] = (long long)__hip_atomic_load(&G->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (multi) MRZ_ACQUIRE_AGENT();  // what earlier commits wrote is not served from this CU's L1
        }
        __syncthreads();
        if (S->snap64[7]) break;
        unsigned snap = (unsigned)S->snap64[0];
        long long epoch = S->snap64[1];
        long long base_idx = S->snap64[2], base_batch = S->snap64[3];
        int pw = (int)S->snap64[4];
        long long p_min = S->snap64[5], p_tag = S->snap64[6], base_pos = S->snap64[8];
        __syncthreads();
        int64_t i0 = 0;  // this batch's window of the candidate list: entries [i0, i0 + pw)
        PROF_T0();
        bool have_prep = (long long)b >= base_batch;
        if (have_prep) {
            i0 = base_idx + ((long long)b - base_batch) * pw;
            if (i0 < K.n) mrz_wide_prep<MRZ_SEQ_WAVES>(C, S, K, lim, i0, pw, base_pos, p_min, p_tag, tid, lane, wave, stat);
        }

        PROF_ADD(MRZ_ST_T_PREP);
        // ---- wait for this batch's turn ------------------------------------------------------------------------
        // The sequencer workgroups share one XCD (the census), so its L2 is where they meet: what has to happen on this
        // side is that no line of this CU's L1 outlives the hand-over.  Nothing is loaded through the L1 between here and
        // the token (the poll is an agent-scope load, which bypasses it; the other waves sit at the barrier), so the
        // invalidate is issued now and completes while the token is awaited.
        if (wave == 0) {
            if (lane == 0) {
                if (multi) MRZ_ACQUIRE_AGENT();
                long long spins = 0;
                int verdict = 1;
                while (true) {
                    const unsigned long long tk = __hip_atomic_load(&G->token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (tk == b) break;
                    if (__hip_atomic_load(&G->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        verdict = 0;
                        break;
                    }
                    if (spins++ > MRZ_TURN_SPIN_LIMIT) {  // cannot happen: every turn ends in a token or in quit
                        verdict = -1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                MRZ_WAIT_STORES();  // (the invalidate has completed)
                S->ctl[2] = verdict;
            }
            MRZ_WAVE_SYNC();
            // the state the last committer has handed on: one load instruction, a word per lane, from the L2
            if (mrz_uni(S->ctl[2]) > 0 && lane < MRZ_H_N)
                S->hand[lane] = __hip_atomic_load(&G->hand[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const int verdict = mrz_uni(S->ctl[2]);
        PROF_ADD(MRZ_ST_T_TURN);
        if (verdict == 0) break;
        if (verdict < 0) {
            if (tid == 0) {
                st->error = 5;
                __hip_atomic_store(&G->quit, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            break;
        }
        // ---- this workgroup's turn: the matcher state is ours ----------------------------------------------------
#ifdef MRZ_SEQ_PROFILE
        const int64_t turn_t0 = (int64_t)__builtin_amdgcn_s_memtime();
#endif
        L = *(const mrz_lead *)&S->hand[MRZ_H_L];
        gseq = S->hand[MRZ_H_GSEQ];
        gnw = (int)S->hand[MRZ_H_GNW];
        farm_hint = (int64_t)S->hand[MRZ_H_FARM];
        const long long cur_epoch = (long long)S->hand[MRZ_H_EPOCH];
        if (cur_epoch != epoch || !have_prep || L.min_mask != p_min || L.tag_mask != p_tag) {
            // prepared under an older epoch (or not at all): once more, now that everything before it is committed
            epoch = cur_epoch;
            base_idx = (long long)S->hand[MRZ_H_BIDX];
            base_batch = (long long)S->hand[MRZ_H_BBATCH];
            base_pos = (long long)S->hand[MRZ_H_BPOS];
            pw = (int)S->hand[MRZ_H_PW];
            snap = (unsigned)b;
            i0 = base_idx + ((long long)b - base_batch) * pw;
            __syncthreads();
            if (i0 < K.n)
                mrz_wide_prep<MRZ_SEQ_WAVES>(C, S, K, lim, i0, pw, base_pos, L.min_mask, L.tag_mask, tid, lane, wave, stat);
        }
        bool finish = false;
        int passed_on = 0;  // the committing wave has handed state and token on already
        if (i0 >= K.n) {
            // the segment's list is used up: nothing but the end is left
            if (L.p < lim) L.p = lim;
            finish = true;
        } else {
            // commit; a batch that ends before its window does (a hand-over that cut it short, a back-jump, more
            // candidates than lanes, the cull window used up) goes on inside the window -- the rest is prepared again
            // from the matcher's position, the windows of the batches behind it stay what they are
            for (int rounds = 0;; rounds++) {
                PROF_T0R();
                mrz_wide_precommit<MRZ_SEQ_WAVES>(C, L, S, wlog, snap, (unsigned)b, tid, lane, wave, stat);
                PROF_ADD(MRZ_ST_T_PRECOMMIT);
                mrz_wide_ret r;
                r.used = 0;
                r.ok = true;
                r.whole = true;
                r.stop_batch = false;
                r.rebulk = false;
                for (;;) {
                    // wave 0 commits; when it finds a long clean stretch behind a hand-over it has all waves commit that;
                    // else it decides how the turn ends and -- if it ends here -- hands state and token on at once, before
                    // the barrier the other waves are waiting at
                    if (wave == 0) {
                        mrz_wide_commit(C, L, S, wlog, (unsigned)b, lane, stat, &r);
                        if (r.rebulk) {
                            if (lane == 0) {
                                S->ctl[4] = 1;
                                S->lead = L;
                            }
                        } else {
                            const int64_t w_end = mrz_uni64(S->w_end), adv_to = mrz_uni64(S->adv_to);
                            const int total = mrz_uni(S->total);
                            const bool masks_moved = L.min_mask != mrz_uni64(S->prep_min_mask) || L.tag_mask != mrz_uni64(S->prep_tag_mask);
                            const bool all_lanes = r.whole && !r.stop_batch;
                            if (all_lanes && adv_to > L.p) L.p = adv_to;  // no candidate is left up to there
                            const bool window_done = all_lanes && adv_to == w_end;
                            // a match has carried the matcher beyond the windows prepared ahead: their turns would be wasted
                            bool far = false;
                            if (L.p > w_end) {
                                int64_t ix = i0 + (int64_t)(n_act + 1) * pw;  // first entry behind those windows
                                if (ix > K.n) ix = K.n;
                                if (ix > i0 + pw) far = L.p >= mrz_uni64(K.cand[ix - 1].off);
                            }
                            // the window size follows what the windows turn out to hold (the list was made under an older mask)
                            constexpr int MAXPW = MRZ_RAW_PER_THREAD * MRZ_W;
                            int new_pw = 0;
                            if (rounds == 0 && all_lanes) {
                                if (!window_done)  // more candidates than lanes
                                    new_pw = (int)((long long)pw * (MRZ_W * 3 / 4) / (total > MRZ_W ? total : MRZ_W));
                                else if (total < MRZ_W * 3 / 8 && pw < MAXPW && i0 + pw <= K.n) {
                                    const int small = mrz_uni((int)S->hand[MRZ_H_SMALL]) + 1;
                                    MRZ_WAVE_SYNC();
                                    if (lane == 0) S->hand[MRZ_H_SMALL] = (unsigned long long)(small >= 4 ? 0 : small);
                                    if (small >= 4) new_pw = (int)((long long)pw * (MRZ_W * 3 / 4) / (total > 16 ? total : 16));
                                } else if (lane == 0)
                                    S->hand[MRZ_H_SMALL] = 0;
                                if (new_pw > MAXPW) new_pw = MAXPW;
                                if (new_pw < 1 && new_pw != 0) new_pw = 1;
                                if (new_pw == pw) new_pw = 0;
                            }
                            int verdict2 = 0;  // 0: pass the token on; 1: go on inside this window; 2: new epoch
                            if (!r.ok || rounds > 2 * MRZ_W) {
                                verdict2 = 0;
                                if (r.ok && lane == 0) C.st->error = 6;  // (cannot happen: every round commits or drops a lane)
                            } else if (masks_moved || far || new_pw > 0)
                                verdict2 = 2;
                            else if (window_done || L.p >= w_end)
                                verdict2 = 0;
                            else
                                verdict2 = 1;
                            if (L.p >= lim) {
                                finish = true;
                                if (verdict2 == 1) verdict2 = 0;
                            }
                            // the mask has reached the deep engine's regime: the launch ends here, the host relaunches
                            // the rest of the stretch on that engine (the front end scans it again from the matcher's
                            // position: scan_next is taken back below)
                            if (masks_moved && a.deep_bits > 0 && __popcll((unsigned long long)L.min_mask) >= a.deep_bits && !finish) {
                                finish = true;
                                if (verdict2 == 1) verdict2 = 0;
                                if (lane == 0) S->ctl[8] = 1;
                            }
                            ST_ADD(MRZ_ST_E_MORE, (all_lanes && !window_done) ? 1 : 0);
                            ST_ADD(MRZ_ST_RESET, verdict2 == 2 ? 1 : 0);
                            ST_ADD(MRZ_ST_REPREP, verdict2 == 1 ? 1 : 0);
                            int64_t nb_pos = L.p + 1, nb_idx = 0;  // a new epoch begins at the first entry behind the matcher
                            if (nb_pos < seg_start) nb_pos = seg_start;
                            if (verdict2 == 2) nb_idx = mrz_cand_lower_bound(K, nb_pos, lane);
                            if (lane == 0) {
                                if (verdict2 == 2) {
                                    S->hand[MRZ_H_EPOCH] = (unsigned long long)(cur_epoch + 1);
                                    S->hand[MRZ_H_BIDX] = (unsigned long long)nb_idx;
                                    S->hand[MRZ_H_BPOS] = (unsigned long long)nb_pos;
                                    S->hand[MRZ_H_BBATCH] = b + 1;
                                    S->hand[MRZ_H_PW] = (unsigned long long)(masks_moved ? mrz_entries_per_batch(list_mask, L.min_mask)
                                                                                          : (new_pw > 0 ? new_pw : pw));
                                }
                                S->ctl[3] = ((r.ok && rounds <= 2 * MRZ_W) ? 0 : 1) | (finish ? 2 : 0) | (verdict2 == 1 ? 4 : 0);
                                S->lead = L;
                            }
                            MRZ_WAVE_SYNC();
                            const int fl0 = mrz_uni(S->ctl[3]);
                            int passed = 0;
                            if (fl0 == 0) {  // the turn ends here, and neither the launch nor anything else does
                                if (lane == 0) {
                                    *(mrz_lead *)&S->hand[MRZ_H_L] = L;
                                    S->hand[MRZ_H_GSEQ] = gseq;
                                    S->hand[MRZ_H_GNW] = (unsigned long long)gnw;
                                    S->hand[MRZ_H_FARM] = (unsigned long long)farm_hint;
                                }
                                MRZ_WAVE_SYNC();
                                if (lane < MRZ_H_N)
                                    __hip_atomic_store(&G->hand[lane], S->hand[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                MRZ_WAIT_STORES();
                                if (lane == 0) __hip_atomic_store(&G->token, b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                passed = 1;
                            }
                            if (lane == 0) {
                                S->ctl[4] = 0;
                                S->ctl[7] = passed;
                            }
                        }
                    }
                    __syncthreads();
                    if (!mrz_uni(S->ctl[4])) break;
                    if (wave != 0) L = S->lead;
                    mrz_wide_bulk<MRZ_SEQ_WAVES>(C, L, S, wlog, (unsigned)b, mrz_uni(S->first_live), mrz_uni(S->rank0), true, tid,
                                                 lane, wave);
                }
                const int fl = mrz_uni(S->ctl[3]);
                passed_on = mrz_uni(S->ctl[7]);
                if (fl & 1) ok = false;
                finish = (fl & 2) != 0;
                if (!(fl & 4)) break;
                if (wave != 0) L = S->lead;
                snap = (unsigned)b;
                __syncthreads();
                mrz_wide_prep<MRZ_SEQ_WAVES>(C, S, K, lim, i0, pw, L.p + 1, L.min_mask, L.tag_mask, tid, lane, wave, stat);
            }
        }
        // ---- hand the state on (unless the committing wave has done so already) -----------------------------------
        if (wave == 0 && !passed_on) {
            // one store instruction, a word per lane
            if (lane == 0) {
                *(mrz_lead *)&S->hand[MRZ_H_L] = L;
                S->hand[MRZ_H_GSEQ] = gseq;
                S->hand[MRZ_H_GNW] = (unsigned long long)gnw;
                S->hand[MRZ_H_FARM] = (unsigned long long)farm_hint;
            }
            MRZ_WAVE_SYNC();
            if (lane < MRZ_H_N) __hip_atomic_store(&G->hand[lane], S->hand[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // ... and the token right behind it.  Every table / log store of this turn has completed already: the
            // bulk steps and the commit end in a wait for their own (MRZ_STORES_DONE), and the other waves have stored
            // nothing since; what is waited for here are the state words above.  The XCD's L2 need not be written back
            // for a reader on the same XCD.
            if (ok && !finish) {
                MRZ_WAIT_STORES();
                if (lane == 0) __hip_atomic_store(&G->token, b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#ifdef MRZ_SEQ_PROFILE
        stat[MRZ_ST_T_TURNWORK] += (int64_t)__builtin_amdgcn_s_memtime() - turn_t0;
#endif
        MRZ_WAIT_STORES();
        __syncthreads();
        if (!ok || finish) {
            if (tid == 0) {
                // the end of the launch: publish the state for the next segment's launch, release everybody
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
                if (S->ctl[8] && L.p < lim) {  // ended early for the deep engine: the rest of the stretch is scanned again
                    const int64_t pt = ((L.p + 1) >> MRZ_TILE_SHIFT) << MRZ_TILE_SHIFT;
                    if (pt < st->scan_next) st->scan_next = pt > seg_start ? pt : seg_start;
                }
                MRZ_RELEASE_AGENT();
                __hip_atomic_store(&G->quit, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#if MRZ_HELPER_WGS > 0
                if (C.gmb) mrz_g_storeu(&C.gmb->quit, 1ull);
#endif
            }
            break;
        }
        b += (unsigned long long)n_act;
    }
#ifdef MRZ_SEQ_STATS
    if (tid == 0)
        for (int k = 0; k < MRZ_ST_N; k++)
            if (stat[k]) __hip_atomic_fetch_add((unsigned long long *)&st->prof[k], (unsigned long long)stat[k], __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
#endif
}

extern "C" size_t mrz_sequencer_wlog_size(int64_t nslots);
extern "C" hipError_t mrz_launch_sequencer(hipStream_t stream, const uint8_t *buf, mrz_slot *tab, const mrz_cand *cand,
                                           const int *tile_off, const mrz_u64 *bitmap, mrz_event *events, mrz_seq_state *st,
                                           void *gmailbox, int n_helpers, void *wide_shared, unsigned *wlog, int64_t nslots,
                                           int seq_wgs, int xcd, int deep_bits) {
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
    a.deep_bits = deep_bits;
#if MRZ_HELPER_WGS == 0
    n_helpers = 0;
#endif
    if (n_helpers > MRZ_HELPER_WGS) n_helpers = MRZ_HELPER_WGS;
    if (n_helpers < 0 || !gmailbox) n_helpers = 0;
    a.n_helpers = n_helpers;
    if (seq_wgs < 1) seq_wgs = 1;
    if (seq_wgs > MRZ_SEQ_WGS) seq_wgs = MRZ_SEQ_WGS;
    hipError_t e = hipSuccess;
    if (gmailbox) e = hipMemsetAsync(gmailbox, 0, sizeof(mrz_gmailbox), stream);
    if (e == hipSuccess) e = hipMemsetAsync(wide_shared, 0, sizeof(mrz_wide_shared), stream);
    if (e == hipSuccess) e = hipMemsetAsync(wlog, 0, (size_t)mrz_sequencer_wlog_size(nslots), stream);
    if (e != hipSuccess) return e;
    // blocks xcd, xcd + 8, xcd + 16, ... are the sequencer workgroups (one XCD), the rest helpers: enough blocks for both
    unsigned grid = (unsigned)(seq_wgs + a.n_helpers);
    if (grid < (unsigned)(8 * (seq_wgs - 1) + a.xcd + 1)) grid = (unsigned)(8 * (seq_wgs - 1) + a.xcd + 1);
#ifdef MRZ_EMU_LDS_PER_BLOCK
    emu::request_coresident();  // (test emulator: this kernel's workgroups wait for each other)
#endif
    hipLaunchKernelGGL(mrz_sequencer_kernel, dim3(grid), dim3(MRZ_SEQ_THREADS), 0, stream, a, (mrz_wide_shared *)wide_shared,
                       wlog, seq_wgs);
    return hipGetLastError();
}

#ifdef MRZ_DBG_HITS
extern "C" int mrz_dbg_hits_set(unsigned *dev_ptr, long long n) {
    int e = (int)hipMemcpyToSymbol(HIP_SYMBOL(mrz_dbg_hits), &dev_ptr, sizeof(dev_ptr));
    if (!e) e = (int)hipMemcpyToSymbol(HIP_SYMBOL(mrz_dbg_n), &n, sizeof(n));
    return e;
}
#endif
extern "C" size_t mrz_sequencer_shared_size(void) { return sizeof(mrz_wide_shared); }
extern "C" size_t mrz_sequencer_wlog_size(int64_t nslots) { return (size_t)((nslots >> MRZ_WLOG_SHIFT) + 1) * sizeof(unsigned); }

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
