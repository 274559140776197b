// mrz_kernels.h -- host-callable launchers of the gfx950 kernels (one per .hip
// file) used by the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mrz_common.h"

typedef unsigned long long mrz_u64;
struct mrz_crc_tables;

extern "C" {
hipError_t mrz_launch_tagscan(hipStream_t stream, const uint8_t *buf, int64_t n, int64_t seg_start, int64_t seg_len,
                              const int64_t *hash_index, const mrz_seq_state *st, int64_t *tags, uint16_t *bitmap16);
hipError_t mrz_launch_sequencer(hipStream_t stream, const uint8_t *buf, mrz_slot *tab, const int64_t *tags,
                                const mrz_u64 *bitmap, mrz_event *events, mrz_seq_state *st, int64_t seg_start,
                                int64_t seg_len, void *gmailbox, int n_helpers, void *wide_shared, unsigned *wlog,
                                int64_t nslots, int seq_wgs);
size_t mrz_sequencer_shared_size(void);
size_t mrz_sequencer_wlog_size(int64_t nslots);
hipError_t mrz_launch_sequencer_narrow(hipStream_t stream, const uint8_t *buf, mrz_slot *tab, const int64_t *tags,
                                       const mrz_u64 *bitmap, mrz_event *events, mrz_seq_state *st, int64_t seg_start,
                                       int64_t seg_len, void *gmailbox, int n_helpers);
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
