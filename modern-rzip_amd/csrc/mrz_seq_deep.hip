// mrz_seq_deep.hip -- the DEEP engine of the sequencer: long probe runs, sparse candidates (large windows).
//
// Same state machine, same matcher state (mrz_seq_state), same candidate list, same compare farm and cooperative path
// (mrz_seq_common.h) as the wide and narrow engines; the host picks one of the three kernels per segment launch
// (mrz_capi.hip).  This one is for the regime every window of more than a few hundred MiB is in:
//
//   once the cull sweeps have tightened minimum_tag_mask to k bits (src/rzip.c:234-244,305-328), every tag in the table
//   ends in k ones, so primary_hash (:232) -- the LOW hash_bits of the tag -- only ever yields slots = 2^k - 1 mod 2^k:
//   the 2.8 M entries of the table sit in 2^(22-k) contiguous RUNS of about 2/3 x 2^k slots, and find_best_match
//   (:426-462), which walks to the first EMPTY slot, reads a whole run per look-up: 21 KiB at k = 11 (a window of a
//   few GiB), 350 KiB at k = 15 (hundreds of GiB).  Candidates are sparse there (2^-k of the positions), so the cost
//   per input byte is constant: about 20 bytes of table scanned per byte of window -- streaming, cache-resident
//   (64 MiB table, 256 MB Infinity Cache), bandwidth-shaped work, not the short latency-bound walks the wide engine's
//   per-lane steps are made for (its walk budget is 96 + 128 slots; beyond it every candidate went through the
//   cooperative path one at a time: 5 us per candidate at k = 11, measured 92 s for 8 GiB of noise).
//
// One workgroup of MRZ_DEEP_WAVES waves (block `xcd` of the grid; the other blocks are the compare farm's helpers):
//
//   FORM    the next list entries that still pass minimum_tag_mask and lie behind the matcher's position: up to
//           MRZ_DEEP_LANES "lanes" of a batch, in position order;
//   SCAN    every wave takes lanes of the batch and reads their runs, 64 slots (1 KiB, coalesced) per load instruction,
//           MRZ_DEEP_GROUP instructions in flight: first empty slot, the tag-equal entries in probe order (and a 64-byte
//           probe of each, single_match_len :372-397), where insert_hash's walk stops (:264-297: empty / due for culling /
//           lower-ranked occupant to displace / the max_chain_len-th tag-equal entry => eviction) and where the displaced
//           occupant's own walk stops.  Read-only, against the table as it stands;
//   COMMIT  wave 0 replays the lanes in position order.  A lane commits as scanned unless an earlier lane of the batch
//           has changed what its scan DEPENDS ON -- which is little (mrz_deep_stale): the slot it writes, the slot its
//           occupant moves to, its first empty slot, an entry with its tag or its occupant's tag written, moved or
//           overwritten anywhere, or a cull inside what it read.  Writes that merely land somewhere in the run it read
//           do not matter: an insert only ever replaces an occupant by a higher-ranked tag (the walk passes both) or fills
//           a slot the lane would have stopped at.  Stale lanes are scanned again, all at once, and the commit resumes.
//           hash_count, victim_round, the cull sweep (a 64-slot window of failing entries ahead of tag_clean_ptr, kept in
//           registers) and the counters are the wave's scalars.  A lane with a real match among its tag-equal entries, a
//           pending lazy match (:586-599), a cascade of displacements or more tag-equal entries than a lane records goes
//           through the cooperative path (mrz_seq_candidate: exact, one candidate at a time), after which the rest of
//           the batch is formed again.
//
// What is committed is exactly what the reference's loop would have done; the result is bit-identical to the other
// engines' (tests: every chunk shape with MRZ_SEQ_ENGINE=deep pinned, the emulator tier, tools/fuzz_parity.py).
//
// Bound: table bytes scanned (cache / fabric bandwidth of the sequencer's CU) and wave 0's serial commit; DESIGN.md 4.2.
#ifndef MRZ_DEEP_WAVES
#define MRZ_DEEP_WAVES 8
#endif
#define MRZ_SEQ_WAVES MRZ_DEEP_WAVES
#include "mrz_seq_common.h"
#include <cstddef>

#ifndef MRZ_DEEP_LANES
#define MRZ_DEEP_LANES 256  // lanes of a batch
#endif
#ifndef MRZ_DEEP_GROUP
#define MRZ_DEEP_GROUP 8    // 64-slot loads a wave keeps in flight
#endif
#define MRZ_DEEP_THREADS (64 * MRZ_DEEP_WAVES)
// all of this wave's global stores have completed (acknowledged by the L2)
#ifdef __HIP_DEVICE_COMPILE__
#define MRZ_DEEP_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define MRZ_DEEP_WAIT() ((void)0)
#endif
#define MRZ_DEEP_MAP 2048      // entries of the two maps of planned writes (slots, tags)
#define MRZ_DEEP_CW_WORDS 32   // cull window: 32 x 64 slots ahead of tag_clean_ptr
#define MRZ_DEEP_XW 192        // slots written by cooperative-path candidates inside one batch (beyond: the batch is cut)

enum { MRZ_DK_NONE = 255, MRZ_DK_EMPTY = 0, MRZ_DK_OVER = 1, MRZ_DK_DISPLACE = 2, MRZ_DK_EVICT = 3 };
#define MRZ_DF_INS 1
#define MRZ_DF_CPLX 2
#define MRZ_DF_STALE 4   // an earlier lane that has committed since touched what the scan depends on: scan again
#define MRZ_DF_FE2 8     // the slot behind the first empty one is empty too
// why a lane stops a round
#define MRZ_DS_COOP 1      // needs the cooperative path (a real match, a cascade, too many tag-equal entries, a write into the cull window)
#define MRZ_DS_CONFLICT 2  // an earlier lane of the round touches what its scan depends on
#define MRZ_DS_CULLED 4    // the round's culls reach into what it has read
#define MRZ_DS_NOCULL 8    // the cull window holds no entry for it

// what a scan leaves per lane (in LDS; the scan helpers' copies travel through device memory)
struct mrz_deep_recs {
    int64_t q[MRZ_DEEP_LANES], t[MRZ_DEEP_LANES];
    int64_t occ_t[MRZ_DEEP_LANES], occ_off[MRZ_DEEP_LANES];  // the occupant a displacing lane moves
    int64_t old_t[MRZ_DEEP_LANES], old_t2[MRZ_DEEP_LANES];   // tags the lane's stores overwrite (0: none)
    int64_t cp_scan[MRZ_DEEP_LANES];                         // tag_clean_ptr when the lane was scanned
    int64_t alt_old_t[MRZ_DEEP_LANES];                       // ... and the tag a store to alt_w overwrites
    int alt_w[MRZ_DEEP_LANES];                                // where the insert's walk stops NEXT if the slot it wanted holds a
                                                              // higher-ranked tag by then (-1: unknown / something else happens)
    int h[MRZ_DEEP_LANES], fe[MRZ_DEEP_LANES], w[MRZ_DEEP_LANES], h2[MRZ_DEEP_LANES], w2[MRZ_DEEP_LANES];
    unsigned short xw_seen[MRZ_DEEP_LANES];                   // cooperative-path writes of this batch the lane's scan has seen
    unsigned char kind[MRZ_DEEP_LANES], kind2[MRZ_DEEP_LANES], nsame[MRZ_DEEP_LANES], flags[MRZ_DEEP_LANES];
    unsigned char alt_kind[MRZ_DEEP_LANES];
    int same_slot[MRZ_DEEP_LANES][MRZ_SMAX];
    int64_t same_off[MRZ_DEEP_LANES][MRZ_SMAX];
    unsigned short raw[MRZ_DEEP_LANES][MRZ_SMAX];             // mrz_deep_probe_raw of every tag-equal entry
};

struct mrz_deep_lds {
    mrz_lead lead;
    int ctl[16];
    int wsum[MRZ_DEEP_WAVES], wsum2[MRZ_DEEP_WAVES], wmin[MRZ_DEEP_WAVES], wmin2[MRZ_DEEP_WAVES], wmin3[MRZ_DEEP_WAVES];
    int n_rescan;
    mrz_deep_recs R;
    int idx[MRZ_DEEP_LANES];                                  // the lane's entry of the candidate list
    int cmin[MRZ_DEEP_LANES];                                 // first lane of the round that touches what this one depends on
    int cpos[MRZ_DEEP_LANES];                                 // where along its walk a stale lane's scan has to be taken up again
    int xw_n;                                                 // slots the cooperative path has written in this batch
    int xw_slot[MRZ_DEEP_XW];
    unsigned short rescan_list[MRZ_DEEP_LANES];
    // what the lanes of this round PLAN to write: slot -> first lane, tag -> first lane
    unsigned smap_key[MRZ_DEEP_MAP];             // slot + 1
    int smap_lane[MRZ_DEEP_MAP];
    unsigned long long tmap_key[MRZ_DEEP_MAP];   // tag (never 0)
    int tmap_lane[MRZ_DEEP_MAP];
    // the cull window: failing entries among the slots from cw_base (clean_one_from_hash, src/rzip.c:313-321)
    mrz_u64 cw[MRZ_DEEP_CW_WORDS];
    int cwcum[MRZ_DEEP_CW_WORDS + 1];
#ifdef MRZ_SEQ_STATS
    int64_t stat[MRZ_ST_N], stat_unused[MRZ_ST_N];
#endif
    mrz_coop_lds coop;
};

__device__ __forceinline__ bool mrz_deep_ranges_meet(int a, int la, int b, int lb, int smask) {
    return la > 0 && lb > 0 && ((((b - a) & smask) < la) || (((a - b) & smask) < lb));
}

// raw 64-byte probe of one (candidate, entry) pair, independent of last_match (single_match_len, src/rzip.c:372-397):
// bits 0-6 equal bytes forward (capped by end - q), bit 7 "forward runs past the reach", bits 8-14 equal bytes backward
// among the 64 before (pieces that would start before byte 0 of the chunk count as equal: mrz_deep_pair_eval knows from
// `op` which ones those are)
__device__ static unsigned mrz_deep_probe_raw(const uint8_t *__restrict__ buf, int64_t q, int64_t op, int64_t end) {
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

// single_match_len's verdict for a raw probe under the current last_match: 0 = a miss (shorter than MINIMUM_MATCH),
// 1 = a match or a compare that runs past the 64-byte reach (the cooperative path measures it)
__device__ __forceinline__ int mrz_deep_pair_eval(unsigned raw, int64_t q, int64_t op, int64_t floor_p) {
    if (op >= q) return 0;
    const int fwd = (int)(raw & 127u), rawb = (int)((raw >> 8) & 127u);
    int64_t maxb = q - floor_p;
    if (op < maxb) maxb = op;
    if (maxb < 0) maxb = 0;
    const int edge = op < 64 ? (int)(op >> 4) : 4;
    if (((raw >> 7) & 1u) || (rawb == 64 && maxb > 64) || (edge < 4 && rawb >= 16 * edge && maxb > 16 * edge)) return 1;
    const int rv = rawb < maxb ? rawb : (int)maxb;
    return fwd + rv >= MRZ_MIN_MATCH ? 1 : 0;
}

// ---- maps of the writes the lanes of a round plan (LDS): key -> FIRST lane that touches it ------------------------
__device__ __forceinline__ unsigned mrz_deep_hs(unsigned x) { return (x * 2654435761u) >> (32 - 11); }
static_assert(MRZ_DEEP_MAP == 2048, "mrz_deep_hs gives 11 bits");
static_assert(MRZ_DEEP_LANES * 4 <= MRZ_DEEP_MAP * 3 / 4 || MRZ_DEEP_LANES <= 256, "the maps hold a batch's writes");

__device__ static void mrz_deep_smap_put(mrz_deep_lds *S, int slot, int lane_no) {
    const unsigned key = (unsigned)slot + 1u;
    unsigned i = mrz_deep_hs(key);
    for (int n = 0; n < MRZ_DEEP_MAP; n++) {
        const unsigned k = atomicCAS(&S->smap_key[i], 0u, key);
        if (k == 0u || k == key) {
            atomicMin(&S->smap_lane[i], lane_no);
            return;
        }
        i = (i + 1) & (MRZ_DEEP_MAP - 1);
    }
}
__device__ static int mrz_deep_smap_get(const mrz_deep_lds *S, int slot) {  // first lane that plans to write `slot`, or INT_MAX
    if (slot < 0) return 0x7fffffff;
    const unsigned key = (unsigned)slot + 1u;
    unsigned i = mrz_deep_hs(key);
    for (int n = 0; n < MRZ_DEEP_MAP; n++) {
        const unsigned k = S->smap_key[i];
        if (k == 0u) return 0x7fffffff;
        if (k == key) return S->smap_lane[i];
        i = (i + 1) & (MRZ_DEEP_MAP - 1);
    }
    return 0;
}
__device__ static void mrz_deep_tmap_put(mrz_deep_lds *S, int64_t tag, int lane_no) {
    if (tag == 0) return;
    const unsigned long long key = (unsigned long long)tag;
    unsigned i = mrz_deep_hs((unsigned)(key ^ (key >> 23)));
    for (int n = 0; n < MRZ_DEEP_MAP; n++) {
        const unsigned long long k = atomicCAS(&S->tmap_key[i], 0ull, key);
        if (k == 0ull || k == key) {
            atomicMin(&S->tmap_lane[i], lane_no);
            return;
        }
        i = (i + 1) & (MRZ_DEEP_MAP - 1);
    }
}
__device__ static int mrz_deep_tmap_get(const mrz_deep_lds *S, int64_t tag) {
    if (tag == 0) return 0x7fffffff;
    const unsigned long long key = (unsigned long long)tag;
    unsigned i = mrz_deep_hs((unsigned)(key ^ (key >> 23)));
    for (int n = 0; n < MRZ_DEEP_MAP; n++) {
        const unsigned long long k = S->tmap_key[i];
        if (k == 0ull) return 0x7fffffff;
        if (k == key) return S->tmap_lane[i];
        i = (i + 1) & (MRZ_DEEP_MAP - 1);
    }
    return 0;
}

// slot of the failing entry of rank `r` in the cull window
__device__ __forceinline__ int mrz_deep_cw_slot(const mrz_deep_lds *S, int64_t cw_base, int r) {
    int lo = 0, hi = MRZ_DEEP_CW_WORDS - 1;
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
static_assert(MRZ_DEEP_CW_WORDS == 32, "mrz_deep_cw_slot searches 32 words in 5 steps");

// ---- SCAN of one lane by one wave ---------------------------------------------------------------------------------
// `from` > 0: a CONTINUATION -- the lane has been scanned before and all that has changed since is a slot at walk
// position `from` or behind (an earlier lane took the slot it wanted to write, or filled its first empty slot): what the
// walk found before that position stands (the precise dependency rule: no other write there matters), so the run is
// taken up again AT that slot instead of being read from its start -- one or two loads instead of 20-170 at masks of 11+
// bits, where most conflicts are several inserters of one run wanting the same slot.
__device__ static void mrz_deep_scan(const mrz_cfg &C, mrz_deep_lds *S, int i, int64_t better, int64_t tag_mask,
                                     int64_t clean_ptr, int xw_n, int lane, int from = 0) {
    const mrz_slot *tab = C.tab;
    const uint8_t *__restrict__ buf = C.buf;
    const int smask = (int)C.slot_mask;
    const int max_chain = (int)C.max_chain;
    const int64_t q = mrz_uni64(S->R.q[i]), t = mrz_uni64(S->R.t[i]);
    const bool ins = (t & tag_mask) == tag_mask;
    const int my_rank = mrz_ones_rank(t);
    const int h = (int)(t & C.slot_mask);
    int fe = -1, w = -1, kind = ins ? -1 : MRZ_DK_NONE;
    int nsame = 0, round = 0;
    bool cplx = false;
    int64_t occ_t = 0, occ_off = 0, old_t = 0;
    bool cont = false, insert_kept = false, fe2 = false;
    int nsame0 = 0;
    // the NEXT place the insert's walk would stop at, should the slot it wants be taken by a tag it passes (another inserter
    // of the same run, earlier in the batch): the second inserter then needs no second look at the run
    int alt_w = -1, alt_kind = MRZ_DK_NONE, alt_st = 0, alt_round = 0;  // alt_st: 0 not looking, 1 looking, 2 settled
    int64_t alt_old_t = 0;
    if (from > 0) {
        const int f0 = mrz_uni((int)S->R.flags[i]), k0 = mrz_uni((int)S->R.kind[i]);
        const int ns0 = mrz_uni((int)S->R.nsame[i]);
        // (evictions and displacements carry state that is not kept per position: those lanes are read from the start)
        if (!(f0 & MRZ_DF_CPLX) && k0 != MRZ_DK_EVICT && k0 != MRZ_DK_DISPLACE && from <= ((mrz_uni(S->R.fe[i]) - h) & smask)) {
            cont = true;
            // the tag-equal entries before `from` stay (they are in probe order)
            const bool before = lane < ns0 && ((S->R.same_slot[i][lane < MRZ_SMAX ? lane : 0] - h) & smask) < from;
            nsame0 = nsame = __popcll(__ballot(before));
            if (ins && (k0 == MRZ_DK_EMPTY || k0 == MRZ_DK_OVER) && ((mrz_uni(S->R.w[i]) - h) & smask) < from) {
                insert_kept = true;  // the insert's place lies before the change: it stands
                kind = k0;
                w = mrz_uni(S->R.w[i]);
                old_t = mrz_uni64(S->R.old_t[i]);
            } else
                round = nsame;  // tag-equal entries the walk has passed so far
        }
    }
    const int start = cont ? from : 0;
    const int max_groups = (int)((C.nslots + 64 * MRZ_DEEP_GROUP - 1) / (64 * MRZ_DEEP_GROUP)) + 1;
    for (int grp = 0; fe < 0; grp++) {
        if (grp >= max_groups) {  // cannot happen: the table is at most 2/3 full
            cplx = true;
            fe = h;
            break;
        }
        mrz_slot e[MRZ_DEEP_GROUP];
        const int s0 = h + start + grp * (64 * MRZ_DEEP_GROUP) + lane;
#pragma unroll
        for (int g = 0; g < MRZ_DEEP_GROUP; g++) e[g] = tab[(s0 + g * 64) & smask];
#pragma unroll
        for (int g = 0; g < MRZ_DEEP_GROUP; g++) {
            if (fe >= 0) continue;
            const int sb = s0 - lane + g * 64;  // slot of lane 0 (before masking)
            const bool empty = (e[g].off | e[g].t) == 0;
            const mrz_u64 m_empty = __ballot(empty);
            const int fe_idx = m_empty ? __ffsll((long long)m_empty) - 1 : 64;
            const mrz_u64 valid = mrz_low_mask(fe_idx);
            const mrz_u64 m_same = __ballot(!empty && e[g].t == t) & valid;
            mrz_u64 m_due = 0, m_low = 0;
            if (kind == -1 || alt_st == 1) {
                m_due = __ballot(!empty && (e[g].t & better) != better) & valid;
                m_low = __ballot(!empty && mrz_ones_rank(e[g].t) < my_rank) & valid & ~m_due;
            }
            int alt_from = 0;  // first slot of this block the search for the next stop looks at
            if (kind == -1) {
                const mrz_u64 m_stop = m_due | m_low;
                const int ks = m_stop ? __ffsll((long long)m_stop) - 1 : 64;
                const int nq = __popcll(m_same & mrz_low_mask(ks));
                if (round + nq >= max_chain) {
                    kind = MRZ_DK_EVICT;  // the victim is the victim_round-th tag-equal entry: picked at the commit
                } else {
                    round += nq;
                    if (ks < 64) {
                        w = (sb + ks) & smask;
                        if ((m_due >> ks) & 1) {
                            kind = MRZ_DK_OVER;
                            old_t = mrz_bcast64(e[g].t, ks);
                            alt_st = 1;
                            alt_from = ks + 1;
                            alt_round = round;
                        } else {
                            kind = MRZ_DK_DISPLACE;
                            occ_t = mrz_bcast64(e[g].t, ks);
                            occ_off = mrz_bcast64(e[g].off, ks);
                        }
                    } else if (fe_idx < 64) {
                        w = (sb + fe_idx) & smask;
                        kind = MRZ_DK_EMPTY;
                        // (behind a first empty slot that somebody else fills: the slot after it, if that is empty too)
                        if (fe_idx < 63 && ((m_empty >> (fe_idx + 1)) & 1)) {
                            alt_w = (sb + fe_idx + 1) & smask;
                            alt_kind = MRZ_DK_EMPTY;
                        }
                        alt_st = 2;
                    }
                }
            }
            if (alt_st == 1 && alt_from < 64) {
                const mrz_u64 beyond = ~mrz_low_mask(alt_from);
                const mrz_u64 m_stop2 = (m_due | m_low) & beyond;
                const int ks2 = m_stop2 ? __ffsll((long long)m_stop2) - 1 : 64;
                const int nq2 = __popcll(m_same & beyond & mrz_low_mask(ks2));
                if (alt_round + nq2 >= max_chain)
                    alt_st = 2;  // (the chain limit comes first: an eviction, not a plain stop)
                else {
                    alt_round += nq2;
                    if (ks2 < 64) {
                        if ((m_due >> ks2) & 1) {
                            alt_w = (sb + ks2) & smask;
                            alt_kind = MRZ_DK_OVER;
                            alt_old_t = mrz_bcast64(e[g].t, ks2);
                        }
                        alt_st = 2;  // (a lower-ranked occupant: a displacement, not recorded)
                    } else if (fe_idx < 64) {
                        alt_w = (sb + fe_idx) & smask;
                        alt_kind = MRZ_DK_EMPTY;
                        alt_st = 2;
                    }
                }
            }
            if (m_same) {
                if ((m_same >> lane) & 1) {
                    const int k = nsame + __popcll(m_same & mrz_low_mask(lane));
                    if (k < MRZ_SMAX) {
                        S->R.same_slot[i][k] = (sb + lane) & smask;
                        S->R.same_off[i][k] = e[g].off;
                    }
                }
                nsame += __popcll(m_same);
            }
            if (fe_idx < 64) {
                fe = (sb + fe_idx) & smask;
                fe2 = fe_idx < 63 && ((m_empty >> (fe_idx + 1)) & 1);
            }
        }
    }
    if (nsame > MRZ_SMAX) cplx = true;
    if (kind == MRZ_DK_EVICT && max_chain > MRZ_SMAX) cplx = true;
    // the displaced occupant's own walk (src/rzip.c:275-278; the table still holds it at `w`)
    int h2 = 0, w2 = -1, kind2 = MRZ_DK_NONE;
    int64_t old_t2 = 0;
    if (kind == MRZ_DK_DISPLACE && !cplx) {
        const int rank2 = mrz_ones_rank(occ_t);
        h2 = (int)(occ_t & C.slot_mask);
        int round2 = 0;
        bool done = false;
        for (int grp = 0; !done; grp++) {
            if (grp >= max_groups) {
                cplx = true;
                break;
            }
            mrz_slot e[MRZ_DEEP_GROUP];
            const int s0 = h2 + grp * (64 * MRZ_DEEP_GROUP) + lane;
#pragma unroll
            for (int g = 0; g < MRZ_DEEP_GROUP; g++) e[g] = tab[(s0 + g * 64) & smask];
#pragma unroll
            for (int g = 0; g < MRZ_DEEP_GROUP; g++) {
                if (done) continue;
                const int sb = s0 - lane + g * 64;
                const bool empty = (e[g].off | e[g].t) == 0;
                const bool due = !empty && (e[g].t & better) != better;
                const bool low = !empty && !due && mrz_ones_rank(e[g].t) < rank2;
                const mrz_u64 m_empty = __ballot(empty), m_due = __ballot(due), m_low = __ballot(low);
                const mrz_u64 m_stop = m_empty | m_due | m_low;
                const int ks = m_stop ? __ffsll((long long)m_stop) - 1 : 64;
                const int nq = __popcll(__ballot(!empty && !due && !low && e[g].t == occ_t) & mrz_low_mask(ks));
                if (round2 + nq >= max_chain) {
                    cplx = true;  // the occupant's own chain limit: cooperative path
                    done = true;
                } else if (ks < 64) {
                    if ((m_low >> ks) & 1)
                        cplx = true;  // second-level displacement: cooperative path
                    else {
                        w2 = (sb + ks) & smask;
                        kind2 = ((m_empty >> ks) & 1) ? MRZ_DK_EMPTY : MRZ_DK_OVER;
                        if (kind2 == MRZ_DK_OVER) old_t2 = mrz_bcast64(e[g].t, ks);
                    }
                    done = true;
                } else
                    round2 += nq;
            }
        }
    }
    // 64-byte probes of the tag-equal entries, one per lane
    const int ns = nsame < MRZ_SMAX ? nsame : MRZ_SMAX;
    if (ns > nsame0) {
        MRZ_WAVE_SYNC();
        if (lane >= nsame0 && lane < ns) S->R.raw[i][lane] = (unsigned short)mrz_deep_probe_raw(buf, q, S->R.same_off[i][lane], C.end);
    }
    if (lane == 0) {
        S->R.h[i] = h;
        S->R.fe[i] = fe;
        S->R.w[i] = w;
        S->R.kind[i] = (unsigned char)(kind < 0 ? MRZ_DK_NONE : kind);
        S->R.h2[i] = h2;
        S->R.w2[i] = w2;
        S->R.kind2[i] = (unsigned char)kind2;
        S->R.occ_t[i] = occ_t;
        S->R.occ_off[i] = occ_off;
        S->R.old_t[i] = old_t;
        S->R.old_t2[i] = old_t2;
        S->R.alt_w[i] = (cplx || kind == MRZ_DK_DISPLACE || kind == MRZ_DK_EVICT || insert_kept) ? -1 : alt_w;
        S->R.alt_kind[i] = (unsigned char)alt_kind;
        S->R.alt_old_t[i] = alt_old_t;
        S->R.nsame[i] = (unsigned char)ns;
        S->R.flags[i] = (unsigned char)((ins ? MRZ_DF_INS : 0) | (cplx ? MRZ_DF_CPLX : 0) | (fe2 ? MRZ_DF_FE2 : 0));
        if (!cont) {  // (a continuation keeps the older of the two states for what it did not read again)
            S->R.cp_scan[i] = clean_ptr;
            S->R.xw_seen[i] = (unsigned short)xw_n;
        }
    }
}

// ---- scan helpers: more CUs for the SCAN phase ------------------------------------------------------------------------
// The scan of a batch is read-only and most of what the engine costs at long runs (62 % of the kernel on the 64 GiB tar,
// at 26 GB/s of table bytes: one CU's worth of loads in flight).  Up to MRZ_DEEP_SCANNERS more workgroups of the grid --
// ANY blocks but the committer's, on all eight XCDs -- take a share of every batch's lanes: the committer posts the batch
// (positions, tags, masks) in device memory, every helper scans the lanes dealt to it into its own LDS and copies the
// records out; the committer scans its own share meanwhile, waits for the helpers' lanes and copies their records in.
// The XCDs' L2s are not coherent with each other, so everything that crosses travels release -> flag -> acquire at
// agent scope: the committer's table stores of the commits before (every wave's vmcnt(0), a barrier, ONE release fence =
// the write-back of its XCD's L2) before the batch's sequence number is stored; a helper's acquire fence (its L1 and its
// XCD's L2 give up what they held) after it has seen the number and before it reads the job or the table; the same the
// other way round for the records.  What a helper reads of the table is then the state at the post -- the state the
// committer's own scan sees --, and the table is not written again before the batch's lanes are all back.
// Rescans, conflicts and the commit stay the committer's.  A helper that is not there (not resident, another XCD, a
// launch without spare blocks) is simply not dealt any lanes: helpers check in with a ticket and the committer counts
// who has.  All spins are bounded or end with the launch (quit).
#ifndef MRZ_DEEP_SCANNERS
#define MRZ_DEEP_SCANNERS 63
#endif
#ifndef MRZ_DEEP_HELP_MIN
#define MRZ_DEEP_HELP_MIN 48  // lanes a batch must have before it is dealt out
#endif

struct mrz_deep_shared {  // device memory, zeroed by the host before every launch
    // polled control words (agent scope), one 128-byte line
    unsigned long long seq;        // number of the batch posted last (0: none yet)
    unsigned long long quit;       // the launch is over
    unsigned long long n_ok;       // helpers that have checked in (tickets handed out)
    unsigned long long xcc_plus1;  // the committer's XCC id + 1 (0: not yet posted)
    unsigned long long pad0[12];
    // the job of batch `seq`, one line
    long long nb, nscan, better, tag_mask, clean_ptr, xw_n;
    long long pad1[10];
    // lanes the helpers have finished for batch `seq`, its own line
    unsigned long long done;
    unsigned long long pad2[15];
    mrz_deep_recs R;  // q / t of every lane (committer), everything else of the helpers' lanes (helpers)
};

#ifdef __HIP_DEVICE_COMPILE__
#define MRZ_DEEP_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#define MRZ_DEEP_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define MRZ_DEEP_XCC_ID() (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15)
#else
#define MRZ_DEEP_ACQUIRE() ((void)0)
#define MRZ_DEEP_RELEASE() ((void)0)
#define MRZ_DEEP_XCC_ID() 0
#endif

// copies the scan record of lane i (everything but q / t) from `src` to `dst`; called by the lane's own thread
__device__ __forceinline__ void mrz_deep_rec_copy(mrz_deep_recs *dst, const mrz_deep_recs *src, int i) {
    dst->occ_t[i] = src->occ_t[i];
    dst->occ_off[i] = src->occ_off[i];
    dst->old_t[i] = src->old_t[i];
    dst->old_t2[i] = src->old_t2[i];
    dst->cp_scan[i] = src->cp_scan[i];
    dst->alt_old_t[i] = src->alt_old_t[i];
    dst->alt_w[i] = src->alt_w[i];
    dst->alt_kind[i] = src->alt_kind[i];
    dst->h[i] = src->h[i];
    dst->fe[i] = src->fe[i];
    dst->w[i] = src->w[i];
    dst->h2[i] = src->h2[i];
    dst->w2[i] = src->w2[i];
    dst->xw_seen[i] = src->xw_seen[i];
    dst->kind[i] = src->kind[i];
    dst->kind2[i] = src->kind2[i];
    dst->flags[i] = src->flags[i];
    const int ns = src->nsame[i];
    dst->nsame[i] = (unsigned char)ns;
    for (int k = 0; k < ns; k++) {
        dst->same_slot[i][k] = src->same_slot[i][k];
        dst->same_off[i][k] = src->same_off[i][k];
        dst->raw[i][k] = src->raw[i][k];
    }
}


// a scan helper workgroup: see above
__device__ static void mrz_deep_scan_helper(const mrz_cfg &C, mrz_deep_lds *S, mrz_deep_shared *G, int tid, int lane, int wave) {
    // check in: on the committer's XCD?
    if (tid == 0) {
        long long spins = 0;
        unsigned long long x = 0;
        while ((x = __hip_atomic_load(&G->xcc_plus1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 &&
               !__hip_atomic_load(&G->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) && spins++ < (1ll << 24))
            __builtin_amdgcn_s_sleep(8);
        int ticket = -1;
        if (x != 0)
            ticket = (int)__hip_atomic_fetch_add(&G->n_ok, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        S->ctl[0] = ticket;
    }
    __syncthreads();
    const int ticket = mrz_uni(S->ctl[0]);
    if (ticket < 0 || ticket >= MRZ_DEEP_SCANNERS) return;
    const int me = ticket + 1;  // my share: lanes i with i % nscan == me
    unsigned long long seen = 0;
    while (true) {
        if (tid == 0) {
            long long spins = 0;
            unsigned long long sq = 0;
            int v = 1;
            while (true) {
                sq = __hip_atomic_load(&G->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (sq != seen) break;
                if (__hip_atomic_load(&G->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || spins++ > MRZ_HELPER_SPIN_LIMIT) {
                    v = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            MRZ_DEEP_ACQUIRE();  // nothing of the job, nor of the table, is served from this CU's L1 from before
            S->ctl[1] = v;
            S->lead.p = (int64_t)sq;
        }
        __syncthreads();
        if (!mrz_uni(S->ctl[1])) return;
        seen = (unsigned long long)mrz_uni64(S->lead.p);
        const int nb = (int)G->nb, nscan = (int)G->nscan, xw_n = (int)G->xw_n;
        const int64_t better = G->better, tag_mask = G->tag_mask, clean_ptr = G->clean_ptr;
        int mine_n = 0;
        if (me < nscan) {
            for (int i = me + tid * nscan; i < nb; i += nscan * MRZ_DEEP_THREADS) {
                S->R.q[i] = G->R.q[i];
                S->R.t[i] = G->R.t[i];
            }
            __syncthreads();
            int k = 0;
            for (int i = me; i < nb; i += nscan, k++)
                if (k % MRZ_DEEP_WAVES == wave) mrz_deep_scan(C, S, i, better, tag_mask, clean_ptr, xw_n, lane);
            mine_n = k;
            __syncthreads();
            for (int i = me + tid * nscan; i < nb; i += nscan * MRZ_DEEP_THREADS) mrz_deep_rec_copy(&G->R, &S->R, i);
            MRZ_DEEP_WAIT();
        }
        __syncthreads();
        if (tid == 0 && mine_n) {
            MRZ_DEEP_RELEASE();
            __hip_atomic_fetch_add(&G->done, (unsigned long long)mine_n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// (out of line: the cooperative path is large and rare -- inlined, twice, it cost the scan loop 120 spilled registers)
__device__ static __attribute__((noinline)) bool mrz_deep_coop(const mrz_cfg &C, mrz_lead &L, mrz_coop_lds *B, int64_t t,
                                                               int lane, int64_t *stat) {
    return mrz_seq_candidate(C, L, B, t, lane, stat);
}

// One candidate whose scan still holds, replayed by wave 0 FROM ITS RECORD instead of walking the table again: the
// tag-equal entries are measured exactly (64-byte probes per lane, long ones by the striped path / the compare farm:
// mrz_resolve_entries) and folded in probe order (find_best_match, src/rzip.c:443-454), the insert lands where the scan
// found its place (insert_hash :256-301), the cull is the generic sweep step (:305-328), then the lazy selection and the
// emit rule (:586-599).  For lanes with a real match among their entries -- the table walk of the cooperative path is
// what costs at long runs.  Appends the slots it writes to S->xw_slot[*xw_n ...]; false on event-list overflow.
__device__ static __attribute__((noinline)) bool mrz_deep_candidate_rec(const mrz_cfg &C, mrz_lead &L, mrz_deep_lds *S, int i, int *xw_n, int lane,
                                              int64_t *stat) {
    mrz_coop_lds *B = &S->coop;
    const int64_t q = mrz_uni64(S->R.q[i]), t = mrz_uni64(S->R.t[i]);
    const int ns = mrz_uni((int)S->R.nsame[i]);
    const int f = mrz_uni((int)S->R.flags[i]);
    int64_t mlen = 0, m_off = 0, m_rev = 0;
    L.p = q;
    if (ns) {
        if (lane < ns) {
            const int64_t op = S->R.same_off[i][lane];
            B->same_off[lane] = op;
            int64_t ml, rv;
            bool lng;
            mrz_lane_match_len(C.buf, q, op, C.end, L.last_match, &ml, &rv, &lng);
            B->pair_res[lane] = lng ? -1 : (int)((ml << 8) | rv);
        }
        MRZ_WAVE_SYNC();
        int xh = 0, xm = 0;
        if (!mrz_resolve_entries(C, L, B, q, ns, lane, stat, &mlen, &m_off, &m_rev, &xh, &xm)) return false;
        L.tag_hits += xh;
        L.tag_misses += xm;
    }
    int nx = *xw_n;
    if (f & MRZ_DF_INS) {
        const int kind = mrz_uni((int)S->R.kind[i]);
        L.inserts++;
        L.count++;
        int ws = mrz_uni(S->R.w[i]);
        if (kind == MRZ_DK_OVER)
            L.count--;
        else if (kind == MRZ_DK_EVICT) {
            L.count--;
            ws = mrz_uni(S->R.same_slot[i][(int)L.victim_round]);
            L.victim_round = L.victim_round + 1 == C.max_chain ? 0 : L.victim_round + 1;
        } else if (kind == MRZ_DK_DISPLACE) {
            const int w2 = mrz_uni(S->R.w2[i]);
            if (mrz_uni((int)S->R.kind2[i]) == MRZ_DK_OVER) L.count--;
            if (lane == 0) {
                mrz_slot oc;
                oc.off = S->R.occ_off[i];
                oc.t = S->R.occ_t[i];
                C.tab[w2] = oc;
                S->xw_slot[nx] = w2;
            }
            nx++;
        }
        if (lane == 0) {
            mrz_slot nw;
            nw.off = q;
            nw.t = t;
            C.tab[ws] = nw;
            S->xw_slot[nx] = ws;
        }
        nx++;
        if (L.count > C.limit) {
            MRZ_DEEP_WAIT();  // (the stores above, before the sweep reads the table)
            mrz_cull_one(C, L, lane);
            if (lane == 0) S->xw_slot[nx] = (int)L.clean_ptr;
            nx++;
        }
    }
    *xw_n = nx;
    return mrz_select_emit(C, L, mlen, m_off, m_rev, lane);
}

// workgroup-wide inclusive prefix sum of one int per thread; *total = the sum.  One barrier; `ws` must not be reused
// before another barrier.
__device__ __forceinline__ int mrz_deep_incl(int v, int *ws, int lane, int wave, int *total) {
    const int incl = mrz_wave_incl_sum(v, lane);
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    int add = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < MRZ_DEEP_WAVES; w++) {
        const int x = ws[w];
        tot += x;
        if (w < wave) add += x;
    }
    *total = tot;
    return incl + add;
}

// first thread (lowest tid) for which `flag` holds, or `none`.  One barrier.
__device__ __forceinline__ int mrz_deep_first(bool flag, int *wm, int lane, int wave, int tid, int none) {
    const mrz_u64 m = __ballot(flag);
    if (lane == 0) wm[wave] = m ? wave * 64 + (__ffsll((long long)m) - 1) : none;
    __syncthreads();
    int r = none;
#pragma unroll
    for (int w = 0; w < MRZ_DEEP_WAVES; w++) r = wm[w] < r ? wm[w] : r;
    return r;
}

__global__ __launch_bounds__(MRZ_DEEP_THREADS) void mrz_seq_deep_kernel(mrz_seq_args a, mrz_deep_shared *G, int scanners) {
#ifdef MRZ_EMU_LDS_PER_BLOCK
    // (the CPU emulator of the test suite keeps `__shared__` in one static copy: one per workgroup that has a role here)
    static mrz_deep_lds deep_all[1 + 15];
    mrz_deep_lds *S = &deep_all[(blockIdx.x / 8) % (1 + 15)];
#else
    __shared__ mrz_deep_lds deep;
    mrz_deep_lds *S = &deep;
#endif
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = mrz_uni(tid >> 6);
    mrz_seq_state *st = a.st;
    if (st->finished || st->error) return;
    const int xcd = a.xcd & 7;
    const int bx = (int)blockIdx.x;
    const bool is_seq = bx == xcd;
#ifdef MRZ_EMU_LDS_PER_BLOCK
    const bool is_scanner = G != nullptr && !is_seq && bx % 8 == xcd && bx / 8 >= 1 && bx / 8 <= scanners && bx / 8 <= MRZ_DEEP_SCANNERS;
#else
    // (any block but the committer's: the table, the job and the records travel through release / acquire at agent scope,
    // which holds across XCDs)
    const bool is_scanner = G != nullptr && !is_seq && (bx < xcd ? bx : bx - 1) < scanners && scanners <= MRZ_DEEP_SCANNERS;
#endif
#if MRZ_HELPER_WGS > 0
    if (!is_seq && !is_scanner) {
        if (a.gmailbox) mrz_helper_wg(a.buf, (mrz_gmailbox *)a.gmailbox);
        return;
    }
#else
    if (!is_seq && !is_scanner) return;
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
    C.mb = nullptr;
    C.mb_seq = nullptr;
    mrz_cands K;
    K.cand = a.cand;
    K.tile_off = a.tile_off;
    K.bitmap = a.bitmap;
    K.seg_start = st->seg_start;
    K.seg_end = st->seg_end;
    K.n = st->n_cand;
    if (is_scanner) {
        mrz_deep_scan_helper(C, S, G, tid, lane, wave);
        return;
    }
    if (K.seg_end <= K.seg_start) {
        if (tid == 0 && G) __hip_atomic_store(&G->quit, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if MRZ_HELPER_WGS > 0
        if (tid == 0 && C.gmb) mrz_g_storeu(&C.gmb->quit, 1ull);
#endif
        return;
    }
    if (tid == 0 && G) __hip_atomic_store(&G->xcc_plus1, (unsigned long long)MRZ_DEEP_XCC_ID() + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long batch_seq = 0;
    const int64_t lim = (C.end < K.seg_end - 1) ? C.end : K.seg_end - 1;  // last candidate position of this launch
    const int smask = (int)C.slot_mask;
    const int max_chain = (int)C.max_chain;
#ifdef MRZ_SEQ_STATS
    // (diagnostic builds: the counters live in LDS, thread 0's are the ones that count)
    for (int k = tid; k < MRZ_ST_N; k += MRZ_DEEP_THREADS) S->stat[k] = 0;
    __syncthreads();
    int64_t *stat = tid == 0 ? S->stat : S->stat_unused;
#else
    int64_t *stat = nullptr;
#endif

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

    // cursor into the candidate list: first entry behind the matcher's position
    int64_t ci;
    {
        int64_t pos = L.p + 1;
        if (pos < K.seg_start) pos = K.seg_start;
        ci = mrz_cand_lower_bound(K, pos, lane);
    }
    bool ok = true;
    PROF_T0();
#ifdef MRZ_SEQ_PROFILE
    const int64_t launch_t0 = (int64_t)__builtin_amdgcn_s_memtime();
#endif
    ST_ADD(MRZ_ST_D_LAUNCHES, 1);
    while (ok && ci < K.n && L.p < lim) {
        PROF_T0R();
        // ---- FORM ------------------------------------------------------------------------------------------------
        const int64_t better = (L.min_mask << 1) | 1;
        const bool loose = L.tag_mask != better;
        int64_t my_q = 0, my_t = 0;
        bool live = false;
        if (ci + tid < K.n) {
            const mrz_cand c = K.cand[ci + tid];
            my_q = c.off;
            my_t = c.t;
            live = c.off > L.p && c.off <= lim && (c.t & L.min_mask) == L.min_mask;
        }
        int total;
        const int incl = mrz_deep_incl(live ? 1 : 0, S->wsum, lane, wave, &total);
        const int at = incl - 1;
        if (live && at < MRZ_DEEP_LANES) {
            S->R.q[at] = my_q;
            S->R.t[at] = my_t;
            S->idx[at] = (int)(ci + tid);
        }
        const int nb = total < MRZ_DEEP_LANES ? total : MRZ_DEEP_LANES;
        // entries of the list this batch covers: all MRZ_DEEP_THREADS examined, or up to its last lane
        int64_t n_examined = K.n - ci < MRZ_DEEP_THREADS ? K.n - ci : MRZ_DEEP_THREADS;
        __syncthreads();
        if (total > MRZ_DEEP_LANES) n_examined = (int64_t)S->idx[MRZ_DEEP_LANES - 1] + 1 - ci;
        if (nb == 0) {  // nothing live among them (an older list, a match that has covered them)
            ci += n_examined;
            __syncthreads();
            continue;
        }
        ST_ADD(MRZ_ST_D_BATCHES, 1);
        ST_ADD(MRZ_ST_D_LANES, nb);
        PROF_ADD(MRZ_ST_D_T_FORM);
        // ---- SCAN --------------------------------------------------------------------------------------------------
        if (!loose) {
            // how many workgroups scan this batch: the helpers that have checked in take lanes i with i % nscan != 0
            int nscan = 1;
            if (G && nb >= MRZ_DEEP_HELP_MIN) {
                int nh = (int)mrz_uni64((int64_t)__hip_atomic_load(&G->n_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (nh > MRZ_DEEP_SCANNERS) nh = MRZ_DEEP_SCANNERS;
                if (nh > scanners) nh = scanners;
                nscan = 1 + nh;
            }
            int helper_lanes = 0;
            if (nscan > 1) {
                if (tid < nb) {
                    G->R.q[tid] = S->R.q[tid];
                    G->R.t[tid] = S->R.t[tid];
                }
                if (tid == 0) {
                    G->nb = nb;
                    G->nscan = nscan;
                    G->better = better;
                    G->tag_mask = L.tag_mask;
                    G->clean_ptr = L.clean_ptr;
                    G->xw_n = 0;
                    __hip_atomic_store(&G->done, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                MRZ_DEEP_WAIT();  // every thread's job stores (and, before them, the table stores of the last commit)
                __syncthreads();
                batch_seq++;
                if (tid == 0) {
                    MRZ_DEEP_RELEASE();
                    __hip_atomic_store(&G->seq, batch_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                helper_lanes = nb - (nb + nscan - 1) / nscan;
            }
            {
                int k = 0;
                for (int i = 0; i < nb; i += nscan, k++)
                    if (k % MRZ_DEEP_WAVES == wave) mrz_deep_scan(C, S, i, better, L.tag_mask, L.clean_ptr, 0, lane);
            }
            if (nscan > 1) {
                if (tid == 0) {
                    long long spins = 0;
                    while (__hip_atomic_load(&G->done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)helper_lanes &&
                           spins++ < (1ll << 26))
                        __builtin_amdgcn_s_sleep(1);
                    S->ctl[6] = spins < (1ll << 26) ? 1 : 0;
                    MRZ_DEEP_ACQUIRE();  // the helpers' records are not served from lines this CU's L1 held before
                }
                __syncthreads();
                if (!mrz_uni(S->ctl[6])) {  // cannot happen: a helper that has checked in answers
                    if (tid == 0) C.st->error = 7;
                    ok = false;
                } else if (tid < nb && tid % nscan != 0)
                    mrz_deep_rec_copy(&S->R, &G->R, tid);
            }
        }
        __syncthreads();
        PROF_ADD(MRZ_ST_D_T_SCAN);
        // ---- COMMIT: rounds ------------------------------------------------------------------------------------------
        for (int k = tid; k < MRZ_DEEP_MAP; k += MRZ_DEEP_THREADS) {  // (every round clears the maps behind itself)
            S->smap_key[k] = 0u;
            S->smap_lane[k] = 0x7fffffff;
            S->tmap_key[k] = 0ull;
            S->tmap_lane[k] = 0x7fffffff;
        }
        __syncthreads();
        int next = 0;       // first lane not dealt with
        bool cut = false;   // the rest of the batch is void (the masks have moved): form again
        int xw_n = 0;       // slots the cooperative path has written since the batch was scanned (S->xw_slot)
        while (next < nb && !cut && ok) {
            const int i = tid;  // this thread's lane
            const bool mine = i >= next && i < nb;
            const int round_start = next;
            int coop_lane = -1;
            bool use_record = false;  // the cooperative-path candidate is replayed from its (still valid) scan record
            if (loose || L.cur_len > 0) {
                // before the first cull no lane has been scanned; a pending lazy match (src/rzip.c:586-599) is decided at
                // the next candidate: the cooperative path takes it
                coop_lane = next;
            } else {
                // the cull window's slots are asked for now (clean_one_from_hash's sweep, src/rzip.c:313-321: 32 x 64 slots
                // ahead of tag_clean_ptr) and looked at after the conflicts have been found: one trip to memory behind the
                // LDS work of R2 / R3
                const int64_t cw_base = L.clean_ptr;
                const bool load_cw = L.count + (nb - next) > C.limit;
                constexpr int CW_PER_WAVE = (MRZ_DEEP_CW_WORDS + MRZ_DEEP_WAVES - 1) / MRZ_DEEP_WAVES;
                mrz_slot cwe[CW_PER_WAVE];
#pragma unroll
                for (int b = 0; b < CW_PER_WAVE; b++) {
                    cwe[b].off = 0;
                    cwe[b].t = 0;
                    const int wi = b * MRZ_DEEP_WAVES + wave;
                    const int64_t slot = cw_base + (int64_t)wi * 64 + lane;
                    if (load_cw && wi < MRZ_DEEP_CW_WORDS && slot < C.nslots) cwe[b] = C.tab[slot];
                }
                // R2: the maps of what the lanes [next, nb) plan to write (cleared at the end of the round before)
                const int f = mine ? S->R.flags[i] : 0;
                const bool ins = mine && (f & MRZ_DF_INS), cplx = mine && (f & MRZ_DF_CPLX);
                const int kind = mine ? S->R.kind[i] : MRZ_DK_NONE, kind2 = mine ? S->R.kind2[i] : MRZ_DK_NONE;
                const int64_t q = mine ? S->R.q[i] : 0, t = mine ? S->R.t[i] : 0;
                const int w = mine ? S->R.w[i] : -1, w2 = mine ? S->R.w2[i] : -1, fe = mine ? S->R.fe[i] : -1;
                const int h = mine ? S->R.h[i] : 0, h2 = mine ? S->R.h2[i] : 0;
                const int64_t occ_t = mine ? S->R.occ_t[i] : 0;
                const bool wr_w = ins && !cplx && (kind == MRZ_DK_EMPTY || kind == MRZ_DK_OVER || kind == MRZ_DK_DISPLACE);
                const bool wr_w2 = ins && !cplx && kind == MRZ_DK_DISPLACE;
                if (ins && !cplx) {
                    if (wr_w) mrz_deep_smap_put(S, w, i);
                    if (wr_w2) mrz_deep_smap_put(S, w2, i);
                    mrz_deep_tmap_put(S, t, i);
                    if (kind == MRZ_DK_OVER) mrz_deep_tmap_put(S, S->R.old_t[i], i);
                    if (kind == MRZ_DK_DISPLACE) {
                        mrz_deep_tmap_put(S, occ_t, i);
                        if (kind2 == MRZ_DK_OVER) mrz_deep_tmap_put(S, S->R.old_t2[i], i);
                    }
                }
                __syncthreads();
                // R3a: a lane whose ONLY trouble is that an earlier lane takes the slot it wanted (the first due entry of the
                // same run -- every inserter of a run stops there) or fills its first empty slot knows from its scan what it
                // finds instead: the next stop of its walk (alt_w), one slot more for the look-up. It plans that now.
                //   res & 1: the insert goes to alt_w;  res & 2: the look-up ends one slot later (fe + 1)
                int res = 0, res_x = -1;
                int w_e = w, kind_e = kind, fe_e = fe;
                if (mine && !cplx && !(f & MRZ_DF_STALE) && kind != MRZ_DK_DISPLACE) {
                    const int m_t = mrz_deep_tmap_get(S, t);
                    const int m_fe = mrz_deep_smap_get(S, fe);
                    const int m_w = wr_w ? mrz_deep_smap_get(S, w) : 0x7fffffff;
                    const bool c_w = wr_w && m_w < i, c_fe = m_fe < i;
                    if (m_t >= i && (c_w || c_fe)) {
                        bool can = true;
                        const int aw = S->R.alt_w[i], ak = S->R.alt_kind[i];
                        if (c_w) {
                            const int x = m_w;
                            res_x = x;
                            const int kx = S->R.kind[x];
                            can = aw >= 0 && S->R.w[x] == w && (kx == MRZ_DK_EMPTY || kx == MRZ_DK_OVER) &&
                                  mrz_ones_rank(S->R.t[x]) >= mrz_ones_rank(t);
                            res |= 1;
                            if (kind == MRZ_DK_EMPTY) res |= 2;  // (w == fe: aw == fe + 1, empty)
                        }
                        if (c_fe && !(res & 2)) {
                            // somebody fills the first empty slot: the look-up passes that entry (another tag: the tag map
                            // says so) and ends at the next slot, if that is empty; an insert that was to end there cannot
                            can = can && (f & MRZ_DF_FE2) && !((res & 1) && ak == MRZ_DK_EMPTY);
                            res |= 2;
                        }
                        if (!can) res = 0;
                    }
                    if (res & 1) {
                        w_e = S->R.alt_w[i];
                        kind_e = S->R.alt_kind[i];
                        mrz_deep_smap_put(S, w_e, i);
                        if (kind_e == MRZ_DK_OVER) mrz_deep_tmap_put(S, S->R.alt_old_t[i], i);
                    }
                    if (res & 2) fe_e = (fe + 1) & smask;
                }
                __syncthreads();
                // R3b: who depends on an earlier lane's writes; who needs the cooperative path
                int cmin = 0x7fffffff;
                int cfrom = 0x7fffffff;  // walk position from which a conflicting lane's scan has to be redone (0: all of it)
                int stop = 0;
                const int cw_len = MRZ_DEEP_CW_WORDS * 64;
                if (mine) {
                    if (cplx)
                        stop |= MRZ_DS_COOP;
                    else {
                        int m;
                        bool hard = false;  // a dependence the lane has no answer to
                        m = mrz_deep_smap_get(S, fe);
                        cmin = m < cmin ? m : cmin;
                        if (m < i) {
                            cfrom = (fe - h) & smask;  // its first empty slot gets filled: the walk goes on from there
                            if (!(res & 2)) hard = true;
                        }
                        m = mrz_deep_tmap_get(S, t);
                        if (m != i) cmin = m < cmin ? m : cmin;
                        if (m < i) {
                            cfrom = 0;  // an entry of its tag somewhere in the run: all of it
                            hard = true;
                        }
                        if (wr_w) {
                            m = mrz_deep_smap_get(S, w);
                            if (m != i) cmin = m < cmin ? m : cmin;
                            if (m < i) {  // the slot it wanted is taken: the walk goes on from there
                                const int pw_ = (w - h) & smask;
                                cfrom = pw_ < cfrom ? pw_ : cfrom;
                                if (!(res & 1) || m != res_x) hard = true;  // (res_x: whose entry the lane steps over)
                            }
                        }
                        if (wr_w2) {
                            m = mrz_deep_smap_get(S, w2);
                            if (m != i) cmin = m < cmin ? m : cmin;
                            if (m < i) {
                                cfrom = 0;
                                hard = true;
                            }
                            m = mrz_deep_tmap_get(S, occ_t);
                            if (m != i) cmin = m < cmin ? m : cmin;
                            if (m < i) {
                                cfrom = 0;
                                hard = true;
                            }
                        }
                        // (what the lane plans instead must be free of earlier lanes' plans in its turn)
                        if ((res & 1) && mrz_deep_smap_get(S, w_e) < i) hard = true;
                        if ((res & 2) && mrz_deep_smap_get(S, fe_e) < i) hard = true;
                        if (hard || (f & MRZ_DF_STALE)) stop |= MRZ_DS_CONFLICT;
                        // a real match (or a compare beyond the 64-byte reach) among the tag-equal entries
                        const int ns = S->R.nsame[i];
                        const int64_t floor_p = L.last_match > 0 ? L.last_match : 0;
                        for (int k = 0; k < ns; k++)
                            if (mrz_deep_pair_eval(S->R.raw[i][k], q, S->R.same_off[i][k], floor_p)) stop |= MRZ_DS_COOP;
                        // a store that takes a failing entry out of the sweep's way changes which entries later culls find
                        if (wr_w && kind_e == MRZ_DK_OVER && w_e >= cw_base && w_e < cw_base + cw_len) stop |= MRZ_DS_COOP;
                        if (wr_w2 && kind2 == MRZ_DK_OVER && w2 >= cw_base && w2 < cw_base + cw_len) stop |= MRZ_DS_COOP;
                        // a slot the cooperative path has written since the scan, inside what the lane has read
                        for (int k = S->R.xw_seen[i]; k < xw_n; k++) {
                            const int xs = S->xw_slot[k];
                            if ((((xs - h) & smask) <= ((fe_e - h) & smask)) || (wr_w2 && (((xs - h2) & smask) <= ((w2 - h2) & smask))))
                                stop |= MRZ_DS_CULLED;
                        }
                        // culled since the scan (by earlier rounds / batches)?
                        const int64_t cp0 = S->R.cp_scan[i];
                        if (L.clean_ptr != cp0) {
                            const int lc = (int)(L.clean_ptr - cp0) + 1;
                            if (mrz_deep_ranges_meet(h, ((fe_e - h) & smask) + 1, (int)cp0, lc, smask)) stop |= MRZ_DS_CULLED;
                            if (wr_w2 && mrz_deep_ranges_meet(h2, ((w2 - h2) & smask) + 1, (int)cp0, lc, smask)) stop |= MRZ_DS_CULLED;
                        }
                    }
                    S->cmin[i] = cmin;
                }
                int e1 = mrz_deep_first(mine && stop != 0, S->wmin, lane, wave, tid, nb);
                // (every thread has read the maps: they are cleared for the next round now)
                for (int k = tid; k < MRZ_DEEP_MAP; k += MRZ_DEEP_THREADS) {
                    S->smap_key[k] = 0u;
                    S->smap_lane[k] = 0x7fffffff;
                    S->tmap_key[k] = 0ull;
                    S->tmap_lane[k] = 0x7fffffff;
                }
                // R4: the cull window's failing entries (all-zero words when this round cannot reach the limit)
#pragma unroll
                for (int b = 0; b < CW_PER_WAVE; b++) {
                    const int wi = b * MRZ_DEEP_WAVES + wave;
                    const mrz_u64 m = __ballot(((cwe[b].off | cwe[b].t) != 0) && ((cwe[b].t & better) != better));
                    if (lane == 0 && wi < MRZ_DEEP_CW_WORDS) S->cw[wi] = m;
                }
                __syncthreads();
                if (wave == 0) {
                    const int c = lane < MRZ_DEEP_CW_WORDS ? __popcll(S->cw[lane]) : 0;
                    const int cinc = mrz_wave_incl_sum(c, lane);
                    if (lane < MRZ_DEEP_CW_WORDS) S->cwcum[lane + 1] = cinc;
                    if (lane == 0) S->cwcum[0] = 0;
                }
                __syncthreads();
                const int cw_total = S->cwcum[MRZ_DEEP_CW_WORDS];
                // R5: the sequential quantities of lanes [next, e1) as prefix sums (hash_count, victim_round, cull ranks)
                const bool inb = mine && i < e1;
                const bool a_ins = inb && ins;
                const bool a_ev = a_ins && kind == MRZ_DK_EVICT;
                const int d = a_ins ? (kind_e == MRZ_DK_EMPTY ? 1 : (kind == MRZ_DK_DISPLACE ? (kind2 == MRZ_DK_EMPTY ? 1 : 0) : 0)) : 0;
                int dummy;
                const int i1 = mrz_deep_incl(d | (a_ev ? 1 << 10 : 0) | (a_ins ? 1 << 20 : 0), S->wsum, lane, wave, &dummy);
                int64_t c_before = L.count + ((i1 & 1023) - d);
                if (c_before > C.limit) c_before = C.limit;
                const bool cull = a_ins && (c_before + d > C.limit);
                const int i2 = mrz_deep_incl((cull ? 1 : 0) | ((inb ? (int)S->R.nsame[i] : 0) << 10), S->wsum2, lane, wave, &dummy);
                int cslot = -1;
                const int r_before = (i2 & 1023) - (cull ? 1 : 0);  // culls of the lanes before this one
                if (cull) {
                    if (r_before >= cw_total)
                        stop |= MRZ_DS_NOCULL;
                    else
                        cslot = mrz_deep_cw_slot(S, cw_base, r_before);
                }
                // the slots the EARLIER lanes of the round empty lie in [cw_base, slot of rank r_before - 1]: a lane that has
                // read there must not commit with them (its own cull comes after its own look-up and insert)
                if (inb && !cplx && r_before > 0) {
                    const int rr = r_before - 1 < cw_total ? r_before - 1 : cw_total - 1;
                    if (rr >= 0) {
                        const int lc = (int)(mrz_deep_cw_slot(S, cw_base, rr) - cw_base) + 1;
                        if (mrz_deep_ranges_meet(h, ((fe_e - h) & smask) + 1, (int)cw_base, lc, smask) ||
                            (wr_w2 && mrz_deep_ranges_meet(h2, ((w2 - h2) & smask) + 1, (int)cw_base, lc, smask)))
                            stop |= MRZ_DS_CULLED;
                    }
                }
                const int e2 = mrz_deep_first(inb && stop != 0, S->wmin2, lane, wave, tid, e1);
                if (mine && i == e2) S->ctl[4] = stop;
                // a lane behind the committed ones that depends on what one of them has written stays marked until it has
                // been scanned again (the round may end in the cooperative path, and the next one only knows ITS lanes' plans)
                if (mine && i >= e2 && cmin < e2) {
                    const int old = (f & MRZ_DF_STALE) ? S->cpos[i] : 0x7fffffff;
                    S->cpos[i] = cfrom < old ? cfrom : old;
                    S->R.flags[i] = (unsigned char)(f | MRZ_DF_STALE);
                }
                // R6: lanes [next, e2) commit as scanned (insert_hash + clean_one_from_hash, src/rzip.c:256-328,579-584)
                if (mine && i < e2) {
                    if (a_ins) {
                        int ws = w_e;
                        if (a_ev) {
                            const int er = ((i1 >> 10) & 1023) - 1;
                            const int vr = (int)(((unsigned)L.victim_round + (unsigned)er) % (unsigned)max_chain);
                            ws = S->R.same_slot[i][vr];
                        }
                        if (kind == MRZ_DK_DISPLACE) {
                            mrz_slot oc;
                            oc.off = S->R.occ_off[i];
                            oc.t = occ_t;
                            C.tab[w2] = oc;
                        }
                        mrz_slot nw;
                        nw.off = q;
                        nw.t = t;
                        C.tab[ws] = nw;
                        if (cslot >= 0) {
                            mrz_slot z;
                            z.off = 0;
                            z.t = 0;
                            C.tab[cslot] = z;
                        }
                    }
                }
                if (e2 > next && tid == e2 - 1) {
                    // totals of the committed lanes
                    mrz_lead N = L;
                    int64_t c_after = c_before + d;
                    if (c_after > C.limit) c_after = C.limit;
                    N.count = c_after;
                    N.inserts = L.inserts + ((i1 >> 20) & 1023);
                    N.victim_round = (int64_t)(((unsigned)L.victim_round + (unsigned)((i1 >> 10) & 1023)) % (unsigned)max_chain);
                    N.tag_misses = L.tag_misses + (i2 >> 10);
                    N.p = q;
                    S->lead = N;
                }
                // (the last culling lane below e2 leaves tag_clean_ptr at its slot)
                {
                    const bool is_c = mine && i < e2 && cslot >= 0;
                    const mrz_u64 mc = __ballot(is_c);
                    int wl = -1;
                    if (mc) wl = mrz_lane_read(cslot, 63 - __clzll((long long)mc));
                    if (lane == 0) S->wsum2[wave] = wl;
                }
                MRZ_DEEP_WAIT();  // the table stores have landed before anybody reads the table again
                __syncthreads();
                if (e2 > next) {
                    L = S->lead;
                    int lastc = -1;
#pragma unroll
                    for (int wv = 0; wv < MRZ_DEEP_WAVES; wv++) lastc = S->wsum2[wv] > lastc ? S->wsum2[wv] : lastc;
                    if (lastc >= 0) L.clean_ptr = lastc;
                }
                ST_ADD(MRZ_ST_COMMITTED, e2 - next);
                next = e2;
                PROF_ADD(MRZ_ST_D_T_COMMIT);
                if (next < nb) {
                    const int sf = mrz_uni(S->ctl[4]);
                    ST_ADD(MRZ_ST_D_S_COOP, (sf & MRZ_DS_COOP) ? 1 : 0);
                    ST_ADD(MRZ_ST_D_S_CONFLICT, (sf & MRZ_DS_CONFLICT) ? 1 : 0);
                    ST_ADD(MRZ_ST_D_S_CULLED, (sf & MRZ_DS_CULLED) ? 1 : 0);
                    ST_ADD(MRZ_ST_D_S_NOCULL, (sf & MRZ_DS_NOCULL) ? 1 : 0);
                    ST_ADD(MRZ_ST_D_S_STALE, (S->R.flags[next] & MRZ_DF_STALE) ? 1 : 0);
                    if ((sf & MRZ_DS_CONFLICT) && !(sf & MRZ_DS_COOP)) {  // what kind of plan the conflicting lane had
                        const int kk = S->R.kind[next];
                        (void)kk;
                        ST_ADD(MRZ_ST_D_C_OVER_ALT, (kk == MRZ_DK_OVER && S->R.alt_w[next] >= 0) ? 1 : 0);
                        ST_ADD(MRZ_ST_D_C_OVER_NOALT, (kk == MRZ_DK_OVER && S->R.alt_w[next] < 0) ? 1 : 0);
                        ST_ADD(MRZ_ST_D_C_EMPTY, kk == MRZ_DK_EMPTY ? 1 : 0);
                        ST_ADD(MRZ_ST_D_C_DISPLACE, kk == MRZ_DK_DISPLACE ? 1 : 0);
                        ST_ADD(MRZ_ST_D_C_OTHER, (kk == MRZ_DK_NONE || kk == MRZ_DK_EVICT) ? 1 : 0);
                    }
                    // (a lane that stops the round for the cooperative path AND whose scan no longer holds is scanned again
                    // first: its record then serves the replay)
                    bool stop_stale = (sf & (MRZ_DS_CONFLICT | MRZ_DS_CULLED)) != 0 || (S->R.flags[next] & MRZ_DF_STALE);
                    if (!stop_stale && !(S->R.flags[next] & MRZ_DF_CPLX)) {
                        const int64_t cp0 = S->R.cp_scan[next];
                        if (L.clean_ptr != cp0) {  // culled by the lanes that have just committed?
                            const int lc = (int)(L.clean_ptr - cp0) + 1;
                            const int hh = S->R.h[next], ff = S->R.fe[next];
                            stop_stale = mrz_deep_ranges_meet(hh, ((ff - hh) & smask) + 1, (int)cp0, lc, smask);
                            if (!stop_stale && (S->R.flags[next] & MRZ_DF_INS) && S->R.kind[next] == MRZ_DK_DISPLACE) {
                                const int hh2 = S->R.h2[next], ww2 = S->R.w2[next];
                                stop_stale = mrz_deep_ranges_meet(hh2, ((ww2 - hh2) & smask) + 1, (int)cp0, lc, smask);
                            }
                        }
                    }
                    if ((sf & MRZ_DS_COOP) && !stop_stale) {
                        coop_lane = next;
                        use_record = !(S->R.flags[next] & MRZ_DF_CPLX);
                    } else if ((sf & MRZ_DS_NOCULL) && next == round_start && !stop_stale) {
                        coop_lane = next;  // (the sweep has to go further than the window reaches, or to wrap: the generic step)
                        use_record = !(S->R.flags[next] & MRZ_DF_CPLX);
                    } else {
                        // lanes at or behind `next` whose scan no longer holds -- an earlier lane that HAS committed touched
                        // what they depend on, or the sweep has reached into what they read -- are scanned again, all at
                        // once, against the table as it is now
                        bool again = false, by_flag = false;
                        if (i >= next && i < nb && !(S->R.flags[i] & MRZ_DF_CPLX)) {
                            again = (S->R.flags[i] & MRZ_DF_STALE) != 0;
                            by_flag = again;
                            {
                                const int hh = S->R.h[i], ff = S->R.fe[i], hh2 = S->R.h2[i], ww2 = S->R.w2[i];
                                const bool dsp = (S->R.flags[i] & MRZ_DF_INS) && S->R.kind[i] == MRZ_DK_DISPLACE;
                                for (int k = S->R.xw_seen[i]; k < xw_n && !again; k++) {
                                    const int xs = S->xw_slot[k];
                                    again = (((xs - hh) & smask) <= ((ff - hh) & smask)) || (dsp && (((xs - hh2) & smask) <= ((ww2 - hh2) & smask)));
                                }
                            }
                            if (!again) {
                                const int64_t cp0 = S->R.cp_scan[i];
                                if (L.clean_ptr != cp0) {
                                    const int lc = (int)(L.clean_ptr - cp0) + 1;
                                    const int hh = S->R.h[i], ff = S->R.fe[i];
                                    again = mrz_deep_ranges_meet(hh, ((ff - hh) & smask) + 1, (int)cp0, lc, smask);
                                    if (!again && (S->R.flags[i] & MRZ_DF_INS) && S->R.kind[i] == MRZ_DK_DISPLACE) {
                                        const int hh2 = S->R.h2[i], ww2 = S->R.w2[i];
                                        again = mrz_deep_ranges_meet(hh2, ((ww2 - hh2) & smask) + 1, (int)cp0, lc, smask);
                                    }
                                }
                            }
                        }
                        // (a slot the cooperative path has written, or a cull, inside what the lane read: all of it again)
                        if (again && i >= next && i < nb) {
                            bool other = false;
                            {
                                const int hh = S->R.h[i], ff = S->R.fe[i], hh2 = S->R.h2[i], ww2 = S->R.w2[i];
                                const bool dsp = (S->R.flags[i] & MRZ_DF_INS) && S->R.kind[i] == MRZ_DK_DISPLACE;
                                for (int k = S->R.xw_seen[i]; k < xw_n && !other; k++) {
                                    const int xs = S->xw_slot[k];
                                    other = (((xs - hh) & smask) <= ((ff - hh) & smask)) || (dsp && (((xs - hh2) & smask) <= ((ww2 - hh2) & smask)));
                                }
                                const int64_t cp0 = S->R.cp_scan[i];
                                if (!other && L.clean_ptr != cp0) {
                                    const int lc = (int)(L.clean_ptr - cp0) + 1;
                                    other = mrz_deep_ranges_meet(hh, ((ff - hh) & smask) + 1, (int)cp0, lc, smask) ||
                                            (dsp && mrz_deep_ranges_meet(hh2, ((ww2 - hh2) & smask) + 1, (int)cp0, lc, smask));
                                }
                            }
                            if (!by_flag || other) S->cpos[i] = 0;
                        }
                        if (tid == 0) S->n_rescan = 0;
                        __syncthreads();
                        if (again) S->rescan_list[atomicAdd(&S->n_rescan, 1)] = (unsigned short)i;
                        __syncthreads();
                        const int nr = S->n_rescan;
                        ST_ADD(MRZ_ST_D_ROUNDS, 1);
                        ST_ADD(MRZ_ST_D_RESCANNED, nr);
                        for (int k = wave; k < nr; k += MRZ_DEEP_WAVES) {
                            const int li = (int)S->rescan_list[k];
                            const int cf = mrz_uni(S->cpos[li]);
                            mrz_deep_scan(C, S, li, better, L.tag_mask, L.clean_ptr, xw_n, lane, cf == 0x7fffffff ? 0 : cf);
                        }
                        __syncthreads();
                        PROF_ADD(MRZ_ST_D_T_RESCAN);
                    }
                }
            }
            if (coop_lane >= 0) {
                // one candidate through the cooperative path (wave 0): exact, any chain / match length, mask promotion.  Its
                // stores (the insert's write-back list, the cull) go on the list of slots the lanes behind it have to check
                // their scans against; a match it emits carries the matcher over the lanes it covers.
                if (wave == 0) {
                    int cl = coop_lane;
                    int verdict = 1, nx = xw_n;
                    // (before the first cull no lane has a scan to keep valid: wave 0 goes on through the batch by itself, one
                    // candidate after the other, for as long as nothing else changes -- the same call, in a loop)
                    while (true) {
                        const int64_t q = mrz_uni64(S->R.q[cl]), t = mrz_uni64(S->R.t[cl]);
                        const int64_t mm0 = L.min_mask, tm0 = L.tag_mask;
                        L.p = q;
                        int nx_rec = nx;
                        const bool rec = use_record && nx + 4 <= MRZ_DEEP_XW;
                        const bool okc = rec ? mrz_deep_candidate_rec(C, L, S, cl, &nx_rec, lane, stat)
                                             : mrz_deep_coop(C, L, &S->coop, t, lane, stat);
                        ST_ADD(MRZ_ST_D_COOP, 1);
                        ST_ADD(MRZ_ST_D_COOP_REC, rec ? 1 : 0);
                        const bool same_masks = L.min_mask == mm0 && L.tag_mask == tm0;
                        const int nwr = rec ? 0 : mrz_uni(S->coop.n_written);
                        const int64_t cs = rec ? -1 : mrz_uni64(S->coop.cull_slot);
                        const bool room = nx + nwr + 1 <= MRZ_DEEP_XW;
                        if (rec)
                            nx = nx_rec;
                        else if (okc && same_masks && room && !loose) {
                            if (lane < nwr) S->xw_slot[nx + lane] = (int)S->coop.pend_h[lane];
                            nx += nwr;
                            if (cs >= 0) {
                                if (lane == 0) S->xw_slot[nx] = (int)cs;
                                nx++;
                            }
                        }
                        // (a match that ends BEFORE the emitting position takes the loop's p back, src/rzip.c:596: the
                        // candidates behind its end run again -- the batch is formed again from there)
                        verdict = !okc ? 3 : ((same_masks && (room || loose) && L.p >= q) ? 1 : 2);
                        if (!(loose && verdict == 1)) break;
                        int j = cl + 1;
                        while (j < nb && mrz_uni64(S->R.q[j]) <= L.p) j++;  // (inside a match emitted meanwhile)
                        if (j >= nb) break;
                        cl = j;
                    }
                    if (lane == 0) {
                        S->lead = L;
                        S->ctl[0] = verdict;
                        S->ctl[5] = nx;
                    }
                    MRZ_DEEP_WAIT();
                }
                __syncthreads();
                L = S->lead;
                xw_n = mrz_uni(S->ctl[5]);
                const int v = mrz_uni(S->ctl[0]);
                if (v == 3) ok = false;
                if (v == 2) cut = true;
                // the first lane behind the matcher's position (an emitted match covers the lanes inside it)
                {
                    const int i = tid;
                    const int nf = mrz_deep_first(i > coop_lane && i < nb && S->R.q[i] > L.p, S->wmin3, lane, wave, tid, nb);
                    next = nf;
                }
                PROF_ADD(MRZ_ST_D_T_COOP);
            }
        }
        // where the next batch begins: behind the entries this one has covered -- or, when the masks have moved or a match
        // has carried the matcher beyond them, at the first entry behind its position
        if (ok) {
            if (cut || L.p > mrz_uni64(S->R.q[nb - 1])) {
                int64_t pos = L.p + 1;
                if (pos < K.seg_start) pos = K.seg_start;
                ci = mrz_cand_lower_bound(K, pos, lane);
            } else
                ci += n_examined;
        }
        __syncthreads();
    }
    if (ok && ci >= K.n && L.p < lim) L.p = lim;  // no candidate is left up to the segment's end (a pending match is
                                                  // emitted at the next candidate, whichever launch sees it)
#ifdef MRZ_SEQ_PROFILE
    stat[MRZ_ST_D_T_TOTAL] += (int64_t)__builtin_amdgcn_s_memtime() - launch_t0;
#endif
    if (tid == 0) {
        if (G) __hip_atomic_store(&G->quit, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if MRZ_HELPER_WGS > 0
        if (C.gmb) mrz_g_storeu(&C.gmb->quit, 1ull);
#endif
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

extern "C" hipError_t mrz_launch_sequencer_deep(hipStream_t stream, const uint8_t *buf, mrz_slot *tab, const mrz_cand *cand,
                                                const int *tile_off, const mrz_u64 *bitmap, mrz_event *events,
                                                mrz_seq_state *st, void *gmailbox, int n_helpers, int xcd, void *deep_shared,
                                                int scanners) {
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
    if (scanners < 0 || !deep_shared) scanners = 0;
    if (scanners > MRZ_DEEP_SCANNERS) scanners = MRZ_DEEP_SCANNERS;
    unsigned grid = (unsigned)(1 + a.n_helpers);
    if (grid < (unsigned)(a.xcd + 1)) grid = (unsigned)(a.xcd + 1);
#ifdef MRZ_EMU_LDS_PER_BLOCK
    // (test emulator: no farm, but room for the scan helpers -- blocks xcd + 8, xcd + 16, ...; they only run beside the
    // committer in the emulator's co-resident mode, and exit at once otherwise)
    if (scanners > 0) {
        grid = (unsigned)(8 * scanners + a.xcd + 1);
        emu::request_coresident();
    }
#else
    if (scanners > a.n_helpers / 2) scanners = a.n_helpers / 2;  // (the other half stays with the compare farm)
    while (scanners > 0 && grid < (unsigned)(scanners + 2)) scanners--;  // (blocks there are)
#endif
    if (deep_shared) {
        hipError_t e = hipMemsetAsync(deep_shared, 0, offsetof(mrz_deep_shared, R), stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(mrz_seq_deep_kernel, dim3(grid), dim3(MRZ_DEEP_THREADS), 0, stream, a, (mrz_deep_shared *)deep_shared,
                       scanners);
    return hipGetLastError();
}

extern "C" size_t mrz_seq_deep_shared_size(void) { return sizeof(mrz_deep_shared); }
