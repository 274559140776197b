// mrz_ctx.h -- private definition of mrz_ctx shared by the host-side translation
// units of libmrzgpu (not part of the ABI).
#pragma once
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mrzgpu.h"
#include "mrz_kernels.h"

// One segment = one front-end pass + one sequencer launch.  A pass looks at up to MRZ_SEG_POSITIONS positions
// (bitmap 1 bit, tile counters 8 B per 4096 of them) and ends early where its candidate list is full
// (MRZ_CAND_CAP entries of 16 B): under a k-bit mask a segment covers about (3/4 cap) << k positions.
#define MRZ_SEG_POSITIONS (1ll << 30)
#define MRZ_CAND_CAP (8ll << 20)
#define MRZ_SEG_AHEAD 4  // segment launches the host keeps queued ahead of the device

struct mrz_ctx {
    int device;
    int level;
    unsigned mb_used, initial_freq, max_chain;
    int hash_bits;
    int64_t nslots;
    hipStream_t stream;
    hipError_t last_err;
    int profiling;
    mrz_timings timings;

    int64_t h_index[256];
    int64_t *d_index;
    mrz_slot *d_tab;
    mrz_seq_state *d_state;
    mrz_seq_state *h_ring;  // pinned: the matcher state as every one of the last MRZ_SEG_AHEAD launches left it
    // the front end's buffers (mrz_tagscan.hip): grown to what a chunk needs
    mrz_fe_hdr *d_fe_hdr;
    uint16_t *d_bitmap;  // 1 bit per position of a pass
    int *d_tile_cnt, *d_tile_off, *d_grp_cnt;
    int64_t fe_tiles_cap;  // tiles the four arrays above hold
    mrz_cand *d_cand;
    int64_t cand_alloc;    // entries d_cand holds
    int64_t cand_cap;      // entries a pass may fill (MRZ_CAND_CAP unless a test shrinks it)
    int engine_pin;        // MRZ_SEQ_ENGINE at mrz_open: 0 per-segment choice, 1 wide, 2 narrow, 3 deep
    int deep_min_bits;     // bits of minimum_tag_mask from which a segment runs on the deep engine (MRZ_DEEP_MIN_BITS)
    int narrow_max_bits;   // ... and from which a run of long matches no longer goes to the narrow engine (MRZ_NARROW_MAX_BITS)
    int print_prof;        // MRZ_PRINT_PROF at mrz_open
    int xcd;               // block index mod 8 of this ctx's sequencer workgroups (concurrent ctxs: one XCD each)
    mrz_event *d_events;
    int64_t event_cap;
    int64_t *d_block_s0, *d_block_s1;
    int64_t block_cap, block1_cap;
    int64_t *d_lit_off;
    int64_t lit_off_cap;
    mrz_enc_totals *d_totals;
    uint8_t *d_s0, *d_s1;
    int64_t s0_cap, s1_cap;
    int64_t s0_len, s1_len;
    uint8_t *d_in;  // staging for host-resident chunks
    int64_t in_cap;
    mrz_crc_tables *d_crc_tables;
    uint32_t *d_crc_parts;
    int64_t crc_parts_cap;
    uint32_t *d_crc_out;
    void *d_gmailbox;  // mailbox of the sequencer's helper workgroups
    void *d_seq_shared;  // what the wide engine's sequencer workgroups hand to each other (token, matcher state)
    unsigned *d_wlog;    // its write log: one batch stamp per few table slots
    int seq_wgs;         // sequencer workgroups per wide launch (MRZ_SEQ_WGS in the environment; default 3)
    void *d_deep_shared; // what the deep engine's committer and its scan helpers hand to each other (mrz_seq_deep.hip)
    int deep_scanners;   // scan helper workgroups per deep launch (MRZ_DEEP_SCANNERS; default 15)
    int farm_helpers;  // helper workgroups per sequencer launch; -1 = farm_default
    int farm_default;  // the default for this ctx's device (about one per CU), fixed in mrz_open
    void *d_rs_tables;  // Reed-Solomon tables (mrz_rs.hip)
    uint8_t *d_rs_out;
    int64_t rs_out_cap;
    void *rz_scratch;  // runzip: parse tables, records, staged streams (mrz_runzip.hip)
    int64_t rz_scratch_cap;
    uint8_t *d_rz_out;
    int64_t rz_out_cap;
    unsigned *d_rz_done;
    int64_t rz_done_cap;
    int have_chunk;

    // LZ4 / BLAKE2b scratch (owned by their translation units, freed in mrz_close)
    void *lz4_scratch;
    int64_t lz4_scratch_cap;
    void *b2_scratch;
    int64_t b2_scratch_cap;
    hipStream_t side_stream;  // low-priority stream for the co-resident checksum kernels
    hipStream_t copy_stream;  // non-blocking stream for copies that must not wait for the queued segments
    mrz_progress_fn progress_fn;
    void *progress_user;
    int64_t events_final;     // matches of the chunk in flight that are final (mrz_fetch_events bound)
    int64_t seg_positions;    // positions per front-end pass at most (MRZ_SEG_POSITIONS unless a test shrinks it)
    mrz_cand_provider_fn cand_fn;  // window sharding: the rank that owns a stretch of the window scans it
    void *cand_user;
};

#define HIPCHK(ctx, expr)                     \
    do {                                      \
        hipError_t e__ = (expr);              \
        if (e__ != hipSuccess) {              \
            (ctx)->last_err = e__;            \
            return MRZ_E_HIP;                 \
        }                                     \
    } while (0)

template <typename T>
static inline int mrz_grow(mrz_ctx *ctx, T **ptr, int64_t *cap, int64_t want) {
    if (want <= *cap && *ptr) return MRZ_OK;
    if (*ptr) {
        hipFree(*ptr);
        *ptr = nullptr;
        *cap = 0;
    }
    int64_t ask = want < 16 ? 16 : want;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, (size_t)ask * sizeof(T));
    if (e != hipSuccess) {
        ctx->last_err = e;
        return MRZ_E_NOMEM;
    }
    *ptr = (T *)p;
    *cap = ask;
    return MRZ_OK;
}


// (re)allocates the front end's buffers for passes of up to `tiles` tiles and lists of up to `entries` candidates
int mrz_fe_reserve(mrz_ctx *ctx, int64_t tiles, int64_t entries);

// resolves a caller buffer to a device pointer (staging host memory on the ctx stream)
int mrz_stage_input(mrz_ctx *ctx, const void *buf, int64_t n, int where, const uint8_t **dev);
