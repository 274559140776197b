// mrz_kernels.h -- host-callable launchers of the gfx950 kernels (one per .hip
// file) used by the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mrz_common.h"

typedef unsigned long long mrz_u64;
struct mrz_crc_tables;

extern "C" {
// the front end's buffers for one pass (device memory): see mrz_tagscan.hip
struct mrz_fe_bufs {
    mrz_fe_hdr *hdr;
    uint16_t *bitmap16;  // 16 pass bits per thread of a tile: 1 bit per position, bit 0 of word 0 = the pass's first position
    int *tile_cnt;       // candidates per tile
    int *tile_off;       // list offset of every tile (+ one entry: the total)
    int *grp_cnt;        // candidates per group of MRZ_FE_GROUP tiles
    mrz_cand *cand;      // the list
};
hipError_t mrz_launch_frontend(hipStream_t stream, const uint8_t *buf, int64_t n, const int64_t *hash_index,
                               mrz_seq_state *st, int max_tiles, int64_t cap, mrz_fe_hdr *hdr, uint16_t *bitmap16,
                               int *tile_cnt, int *tile_off, int *grp_cnt, mrz_cand *cand);
// xcd: which of the 8 XCDs (block index mod 8 under round-robin placement) carries the sequencer workgroups
hipError_t mrz_launch_sequencer(hipStream_t stream, const uint8_t *buf, mrz_slot *tab, const mrz_cand *cand,
                                const int *tile_off, const mrz_u64 *bitmap, mrz_event *events, mrz_seq_state *st,
                                void *gmailbox, int n_helpers, void *wide_shared, unsigned *wlog,
                                int64_t nslots, int seq_wgs, int xcd, int deep_bits);
size_t mrz_sequencer_shared_size(void);
size_t mrz_sequencer_wlog_size(int64_t nslots);
hipError_t mrz_launch_sequencer_narrow(hipStream_t stream, const uint8_t *buf, mrz_slot *tab, const mrz_cand *cand,
                                       const int *tile_off, const mrz_u64 *bitmap, mrz_event *events, mrz_seq_state *st,
                                       void *gmailbox, int n_helpers, int xcd);
hipError_t mrz_launch_sequencer_deep(hipStream_t stream, const uint8_t *buf, mrz_slot *tab, const mrz_cand *cand,
                                     const int *tile_off, const mrz_u64 *bitmap, mrz_event *events, mrz_seq_state *st,
                                     void *gmailbox, int n_helpers, int xcd, void *deep_shared, int scanners);
size_t mrz_seq_deep_shared_size(void);
size_t mrz_sequencer_mailbox_size(void);
size_t mrz_seq_narrow_mailbox_size(void);
int mrz_sequencer_default_helpers(int device);
hipError_t mrz_launch_enc_size(hipStream_t stream, const mrz_event *ev, int64_t E, int64_t n, int cb, int64_t *block_s0,
                               int64_t *block_s1, mrz_enc_totals *totals);
hipError_t mrz_launch_enc_write(hipStream_t stream, const uint8_t *buf, const mrz_event *ev, int64_t E, int64_t n,
                                int cb, const int64_t *block_s0, const int64_t *block_s1, uint8_t *s0, uint8_t *s1,
                                int64_t s1_len, int64_t *lit_off, mrz_enc_totals *totals, uint32_t crc);
void mrz_crc_build_tables(mrz_crc_tables *tb);
size_t mrz_crc_tables_size(void);
int64_t mrz_crc32_parts_needed(int64_t n);
hipError_t mrz_launch_crc32(hipStream_t stream, const uint8_t *buf, int64_t n, const mrz_crc_tables *tb, uint32_t *parts,
                            uint32_t *crc_out);
}
