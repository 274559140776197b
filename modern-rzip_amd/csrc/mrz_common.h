// mrz_common.h -- structures shared by the host side of libmrzgpu and its
// gfx950 kernels.  Vocabulary follows the reference (src/rzip.c): tags, slots,
// chunk, streams, matches/literals.
#pragma once
#include <stdint.h>

#define MRZ_MIN_MATCH 31      // MINIMUM_MATCH, src/rzip.c:49
#define MRZ_GREAT_MATCH 1024  // GREAT_MATCH,   src/rzip.c:48
#define MRZ_WAVE 64

// struct hash_entry {i64 offset; tag t;}  src/rzip.c:59-62.  All-zero = empty.
struct __attribute__((aligned(16))) mrz_slot {
    int64_t off;
    int64_t t;
};

// One emitted match (before 0xFFFF splitting): what hash_search hands to
// put_match at src/rzip.c:594 -- {current.p, current.ofs, current.len}.
struct mrz_event {
    int64_t p;
    int64_t ofs;
    int64_t len;
};

// Matcher state that hash_search keeps in locals / rzip_state
// (src/rzip.c:508-545, include/mrzip_private.h:388-416).  Lives in device
// memory so the per-segment launches of one chunk continue each other.
struct mrz_seq_state {
    int64_t n;            // chunk_size
    int64_t end;          // chunk_size - MINIMUM_MATCH
    int64_t p;            // last position the main loop has visited
    int64_t cur_p, cur_ofs, cur_len;  // `current`
    int64_t last_match;
    int64_t min_mask;     // st->minimum_tag_mask
    int64_t tag_mask;     // local tag_mask of hash_search
    int64_t count;        // st->hash_count
    int64_t limit;        // st->hash_limit
    int64_t clean_ptr;    // st->tag_clean_ptr
    int64_t victim_round; // static victim_round of insert_hash
    int64_t max_chain;    // level->max_chain_len
    int64_t slot_mask;    // (1 << hash_bits) - 1
    int64_t n_events;
    int64_t event_cap;
    int64_t inserts, tag_hits, tag_misses;  // stats
    int32_t finished;     // main loop has reached `end`
    int32_t error;        // nonzero: event list overflow etc.
    int64_t hint_positions;  // regime hint of the latest launch: positions it advanced over ...
    int64_t hint_events;     // ... matches it emitted ...
    int64_t hint_matched;    // ... and the bytes those matches cover (most positions matched = one long match after another)
    // the segment the front end (mrz_tagscan.hip) has laid out for the next sequencer launch: a compacted, position-ordered
    // list of the candidates of [seg_start, seg_end) -- the positions whose tag passed minimum_tag_mask (src/rzip.c:573)
    // at the time of the scan (list_mask; masks only tighten, so the list is a superset the sequencer re-checks)
    int64_t seg_start;       // first position covered (a multiple of 4096)
    int64_t seg_end;         // one past the last position covered
    int64_t n_cand;          // entries of the list
    int64_t scan_next;       // where the next front-end pass begins
    int64_t list_mask;       // minimum_tag_mask the list was made under
    int64_t prof[128];    // cycle accumulators of a -DMRZ_SEQ_PROFILE build (diagnostics only)
};

// One candidate of the front end's list: a position whose tag passes the mask.  Same 16 bytes as struct hash_entry
// (mrz_slot): off = position in the chunk, t = tag.
typedef mrz_slot mrz_cand;

// header the front-end kernels of one pass share (device memory)
struct mrz_fe_hdr {
    int64_t base;     // first position of the pass (tile-aligned)
    int64_t mask;     // minimum_tag_mask it filters with
    int64_t p_done;   // positions up to here need no candidates (the matcher has passed them)
    int32_t ntiles;   // 4096-position tiles it looks at
    int32_t T;        // ... and how many of them fit the list (set by the scan kernel)
};

#define MRZ_TILE 4096          // positions per front-end tile (one 256-thread workgroup)
#define MRZ_TILE_SHIFT 12
#define MRZ_FE_GROUP 256       // tiles per group of the offset scan

// result of the record-sizing pass
struct mrz_enc_totals {
    int64_t s0_len;       // without terminator + CRC
    int64_t s1_len;
    int64_t literals, literal_bytes, matches, match_bytes;
};
