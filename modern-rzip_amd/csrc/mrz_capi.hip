// mrz_capi.hip -- the C-ABI layer of libmrzgpu.so (include/mrzgpu.h): context
// and device-buffer management, chunk orchestration (HIP stream of launches),
// nothing else.  Host code only; the kernels live in the other .hip files.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mrzgpu.h"
#include "mrz_ctx.h"



// levels[] rows {mb_used, initial_freq, max_chain_len}, src/rzip.c:65-73
static const unsigned k_levels[10][3] = { { 1, 4, 1 },  { 2, 4, 2 },  { 4, 4, 2 },   { 8, 4, 2 },   { 16, 4, 3 },
                                          { 32, 4, 4 }, { 32, 2, 6 }, { 64, 1, 16 }, { 64, 1, 32 }, { 64, 1, 128 } };

// init_hash_indexes (src/rzip.c:669-673): (random() << 16) ^ random() from
// glibc's TYPE_3 additive-feedback generator at its default seed 1, computed
// here so the table does not depend on what else the host process did with
// random().
static void mrz_make_hash_index(int64_t H[256]) {
    int32_t r[31];
    int32_t word = 1;
    r[0] = 1;
    for (int i = 1; i < 31; i++) {
        const long hi = word / 127773, lo = word % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        r[i] = word = (int32_t)w;
    }
    int f = 3, b = 0;
    uint32_t draw[512];
    for (int k = -310; k < 512; k++) {
        const uint32_t s = (uint32_t)r[f] + (uint32_t)r[b];
        r[f] = (int32_t)s;
        if (++f == 31) f = 0;
        if (++b == 31) b = 0;
        if (k >= 0) draw[k] = s >> 1;
    }
    for (int i = 0; i < 256; i++) H[i] = ((int64_t)draw[2 * i] << 16) ^ (int64_t)draw[2 * i + 1];
}

extern "C" int mrz_abi_version(void) { return MRZ_ABI_VERSION; }

// Room for the emitted matches of a chunk of n bytes (24 B each).  Matches are >= 31 bytes and disjoint, so n / 31 + 2
// always suffices -- 53 GB for a 64 GiB chunk, more than the chunk for a 256 GiB window.  Beyond MRZ_EVENT_CAP entries
// (6.4 GB: one match per 256 bytes of a 64 GiB chunk; text emits one per ~400 bytes, the tar mix one per 500 KB) the
// list is bounded instead; a chunk that would overflow it fails with MRZ_E_OVERFLOW (the sequencers stop at the cap).
#define MRZ_EVENT_CAP (1ll << 28)
static int64_t mrz_event_room(int64_t n) {
    const int64_t worst = n / MRZ_MIN_MATCH + 2;
    return worst < MRZ_EVENT_CAP ? worst : MRZ_EVENT_CAP;
}

extern "C" const char *mrz_strerror(int code) {
    switch (code) {
        case MRZ_OK: return "ok";
        case MRZ_E_ARG: return "bad argument";
        case MRZ_E_NODEVICE: return "no usable HIP device (libmrzgpu has no CPU fallback)";
        case MRZ_E_NOMEM: return "out of memory";
        case MRZ_E_HIP: return "HIP runtime error";
        case MRZ_E_OVERFLOW: return "internal capacity exceeded";
        case MRZ_E_STATE: return "call order violated";
        case MRZ_E_CORRUPT: return "corrupt record stream or archive";
        case MRZ_E_UNSUPPORTED: return "block type not handled on this path (back-end codecs are host code)";
        default: return "unknown error";
    }
}

extern "C" int mrz_last_hip_error(const mrz_ctx *ctx, const char **text) {
    if (!ctx) return 0;
    if (text) *text = hipGetErrorString(ctx->last_err);
    return (int)ctx->last_err;
}

extern "C" int mrz_chunk_bytes(int64_t chunk_size) {
    int bits = 8;
    while (chunk_size >> bits > 0) bits++;
    return bits / 8 + (bits % 8 ? 1 : 0);
}

extern "C" void mrz_close(mrz_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    hipFree(ctx->d_index);
    hipFree(ctx->d_tab);
    hipFree(ctx->d_state);
    if (ctx->h_ring) hipHostFree(ctx->h_ring);
    hipFree(ctx->d_fe_hdr);
    hipFree(ctx->d_bitmap);
    hipFree(ctx->d_tile_cnt);
    hipFree(ctx->d_tile_off);
    hipFree(ctx->d_grp_cnt);
    hipFree(ctx->d_cand);
    hipFree(ctx->d_events);
    hipFree(ctx->d_block_s0);
    hipFree(ctx->d_block_s1);
    hipFree(ctx->d_lit_off);
    hipFree(ctx->d_totals);
    hipFree(ctx->d_s0);
    hipFree(ctx->d_s1);
    hipFree(ctx->d_in);
    hipFree(ctx->d_crc_tables);
    hipFree(ctx->d_crc_parts);
    hipFree(ctx->d_crc_out);
    if (ctx->d_gmailbox) hipFree(ctx->d_gmailbox);
    if (ctx->d_seq_shared) hipFree(ctx->d_seq_shared);
    if (ctx->d_deep_shared) hipFree(ctx->d_deep_shared);
    if (ctx->d_wlog) hipFree(ctx->d_wlog);
    if (ctx->rz_scratch) hipFree(ctx->rz_scratch);
    if (ctx->d_rz_out) hipFree(ctx->d_rz_out);
    if (ctx->d_rz_done) hipFree(ctx->d_rz_done);
    if (ctx->d_rs_tables) hipFree(ctx->d_rs_tables);
    if (ctx->d_rs_out) hipFree(ctx->d_rs_out);
    if (ctx->lz4_scratch) hipFree(ctx->lz4_scratch);
    if (ctx->b2_scratch) hipFree(ctx->b2_scratch);
    if (ctx->side_stream) hipStreamDestroy(ctx->side_stream);
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    free(ctx);
}

extern "C" int mrz_open(mrz_ctx **out, int device, int level, int64_t max_chunk) {
    if (!out || level < 1 || level > 9 || max_chunk < 0) return MRZ_E_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e0 = hipGetDeviceCount(&ndev);
    if (e0 != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        fprintf(stderr, "libmrzgpu: hipGetDeviceCount -> %d (%s), %d device(s), asked for %d\n", (int)e0,
                hipGetErrorString(e0), ndev, device);
        return MRZ_E_NODEVICE;
    }
    e0 = hipSetDevice(device);
    if (e0 != hipSuccess) {
        fprintf(stderr, "libmrzgpu: hipSetDevice(%d) -> %d (%s)\n", device, (int)e0, hipGetErrorString(e0));
        return MRZ_E_NODEVICE;
    }
    mrz_ctx *ctx = (mrz_ctx *)calloc(1, sizeof(mrz_ctx));
    if (!ctx) return MRZ_E_NOMEM;
    ctx->farm_helpers = -1;
    ctx->seg_positions = MRZ_SEG_POSITIONS;
    ctx->cand_cap = MRZ_CAND_CAP;
    // diagnostics / test knobs, read once per ctx: never per chunk
    {
        const char *e = getenv("MRZ_SEQ_ENGINE");
        if (e && !strcmp(e, "wide")) ctx->engine_pin = 1;
        if (e && !strcmp(e, "narrow")) ctx->engine_pin = 2;
        if (e && !strcmp(e, "deep")) ctx->engine_pin = 3;
        ctx->deep_min_bits = 6;
        if (const char *d = getenv("MRZ_DEEP_MIN_BITS")) ctx->deep_min_bits = atoi(d);
        // (at deep masks the narrow engine's one-wave table walks cost 70 us per candidate: stride-64G 20.2 s with it,
        // 8.1 s when those segments go to the deep engine too; rep64k-10G, whose mask stays below, is unchanged)
        ctx->narrow_max_bits = ctx->deep_min_bits;
        if (const char *d = getenv("MRZ_NARROW_MAX_BITS")) ctx->narrow_max_bits = atoi(d);
        e = getenv("MRZ_PRINT_PROF");
        if (e) ctx->print_prof = !strcmp(e, "narrow") ? 2 : 1;
    }
    ctx->farm_default = mrz_sequencer_default_helpers(device);
    ctx->device = device;
    ctx->level = level;
    ctx->mb_used = k_levels[level][0];
    ctx->initial_freq = k_levels[level][1];
    ctx->max_chain = k_levels[level][2];
    // table geometry, src/rzip.c:521-530
    const int64_t want = (int64_t)ctx->mb_used * (1048576 / 16);
    for (ctx->hash_bits = 0; (1ll << ctx->hash_bits) < want; ctx->hash_bits++) {
    }
    ctx->nslots = 1ll << ctx->hash_bits;
    mrz_make_hash_index(ctx->h_index);

    int rc = MRZ_OK;
    e0 = hipStreamCreate(&ctx->stream);
    if (e0 != hipSuccess) {
        fprintf(stderr, "libmrzgpu: hipStreamCreate -> %d (%s)\n", (int)e0, hipGetErrorString(e0));
        rc = MRZ_E_NODEVICE;
    }
    int64_t cap;
    if (!rc) { cap = 0; rc = mrz_grow(ctx, &ctx->d_index, &cap, 256); }
    if (!rc) { cap = 0; rc = mrz_grow(ctx, &ctx->d_tab, &cap, ctx->nslots); }
    if (!rc) { cap = 0; rc = mrz_grow(ctx, &ctx->d_state, &cap, 1); }
    if (!rc && hipHostMalloc((void **)&ctx->h_ring, MRZ_SEG_AHEAD * sizeof(mrz_seq_state)) != hipSuccess) rc = MRZ_E_NOMEM;
    if (!rc) { cap = 0; rc = mrz_grow(ctx, &ctx->d_fe_hdr, &cap, 1); }
    if (!rc) { cap = 0; rc = mrz_grow(ctx, &ctx->d_totals, &cap, 1); }
    if (!rc) { cap = 0; rc = mrz_grow(ctx, &ctx->d_crc_out, &cap, 4); }
    if (!rc) {
        void *p = nullptr;
        if (hipMalloc(&p, mrz_crc_tables_size()) != hipSuccess)
            rc = MRZ_E_NOMEM;
        else
            ctx->d_crc_tables = (mrz_crc_tables *)p;
    }
    if (!rc) {
        mrz_crc_tables *tb = (mrz_crc_tables *)malloc(mrz_crc_tables_size());
        if (!tb)
            rc = MRZ_E_NOMEM;
        else {
            mrz_crc_build_tables(tb);
            if (hipMemcpy(ctx->d_crc_tables, tb, mrz_crc_tables_size(), hipMemcpyHostToDevice) != hipSuccess)
                rc = MRZ_E_HIP;
            free(tb);
        }
    }
    if (!rc && hipMemcpy(ctx->d_index, ctx->h_index, sizeof(ctx->h_index), hipMemcpyHostToDevice) != hipSuccess)
        rc = MRZ_E_HIP;
    const size_t mbox = mrz_sequencer_mailbox_size() > mrz_seq_narrow_mailbox_size() ? mrz_sequencer_mailbox_size()
                                                                                     : mrz_seq_narrow_mailbox_size();
    if (!rc && mbox && !getenv("MRZ_NO_HELPER_WGS")) {
        void *p = nullptr;
        if (hipMalloc(&p, mbox) != hipSuccess)
            rc = MRZ_E_NOMEM;
        else
            ctx->d_gmailbox = p;
    }
    if (!rc) {
        if (hipMalloc(&ctx->d_seq_shared, mrz_sequencer_shared_size()) != hipSuccess ||
            hipMalloc((void **)&ctx->d_wlog, mrz_sequencer_wlog_size(ctx->nslots)) != hipSuccess)
            rc = MRZ_E_NOMEM;
        ctx->seq_wgs = 3;
        if (const char *e = getenv("MRZ_SEQ_WGS")) ctx->seq_wgs = atoi(e);
        if (hipMalloc(&ctx->d_deep_shared, mrz_seq_deep_shared_size()) != hipSuccess) rc = MRZ_E_NOMEM;
        ctx->deep_scanners = 63;
        if (const char *e = getenv("MRZ_DEEP_SCANNERS")) ctx->deep_scanners = atoi(e);
    }
    if (!rc && max_chunk > 0) {
        rc = mrz_grow(ctx, &ctx->d_events, &ctx->event_cap, mrz_event_room(max_chunk));
        if (!rc) rc = mrz_grow(ctx, &ctx->d_crc_parts, &ctx->crc_parts_cap, mrz_crc32_parts_needed(max_chunk));
        if (!rc) rc = mrz_fe_reserve(ctx, max_chunk / MRZ_TILE + 2, max_chunk);
    }
    if (rc) {
        mrz_close(ctx);
        return rc;
    }
    *out = ctx;
    return MRZ_OK;
}

extern "C" void *mrz_stream(const mrz_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int mrz_synchronize(mrz_ctx *ctx) {
    if (!ctx) return MRZ_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MRZ_OK;
}

extern "C" int mrz_set_profiling(mrz_ctx *ctx, int enable) {
    if (!ctx) return MRZ_E_ARG;
    ctx->profiling = enable ? 1 : 0;
    return MRZ_OK;
}

extern "C" int mrz_set_progress(mrz_ctx *ctx, mrz_progress_fn fn, void *user) {
    if (!ctx) return MRZ_E_ARG;
    ctx->progress_fn = fn;
    ctx->progress_user = user;
    return MRZ_OK;
}

extern "C" int mrz_fetch_events(mrz_ctx *ctx, int64_t first, int64_t count, mrz_match *host_dst) {
    if (!ctx || first < 0 || count < 0 || (count > 0 && !host_dst)) return MRZ_E_ARG;
    if (first + count > ctx->events_final) return MRZ_E_STATE;
    if (!count) return MRZ_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    static_assert(sizeof(mrz_match) == sizeof(mrz_event), "mrz_match mirrors mrz_event");
    HIPCHK(ctx, hipMemcpyAsync(host_dst, ctx->d_events + first, (size_t)count * sizeof(mrz_event), hipMemcpyDeviceToHost,
                               ctx->copy_stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
    return MRZ_OK;
}

extern "C" int mrz_set_cand_provider(mrz_ctx *ctx, mrz_cand_provider_fn fn, void *user) {
    if (!ctx) return MRZ_E_ARG;
    ctx->cand_fn = fn;
    ctx->cand_user = user;
    return MRZ_OK;
}

extern "C" int mrz_copy_to_device(mrz_ctx *ctx, void *dst_device, const void *src_host, int64_t n) {
    if (!ctx || n < 0 || (n > 0 && (!dst_device || !src_host))) return MRZ_E_ARG;
    if (!n) return MRZ_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(dst_device, src_host, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // the source may be reused on return
    return MRZ_OK;
}

extern "C" int mrz_copy_device(mrz_ctx *ctx, void *dst_device, const void *src_device, int64_t n) {
    if (!ctx || n < 0 || (n > 0 && (!dst_device || !src_device))) return MRZ_E_ARG;
    if (!n) return MRZ_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(dst_device, src_device, (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MRZ_OK;
}

extern "C" int mrz_set_segment_positions(mrz_ctx *ctx, int64_t positions) {
    if (!ctx || positions < MRZ_TILE || positions > MRZ_SEG_POSITIONS || positions % MRZ_TILE) return MRZ_E_ARG;
    ctx->seg_positions = positions;
    return MRZ_OK;
}

extern "C" int mrz_set_candidate_capacity(mrz_ctx *ctx, int64_t entries) {
    if (!ctx || entries < MRZ_TILE || entries > (1ll << 30)) return MRZ_E_ARG;
    ctx->cand_cap = entries;
    return MRZ_OK;
}

extern "C" int mrz_set_xcd(mrz_ctx *ctx, int xcd) {
    if (!ctx || xcd < 0 || xcd > 7) return MRZ_E_ARG;
    ctx->xcd = xcd;
    return MRZ_OK;
}

// the front end's buffers: passes of up to `tiles` tiles, lists of up to `entries` candidates
int mrz_fe_reserve(mrz_ctx *ctx, int64_t tiles, int64_t entries) {
    const int64_t max_tiles = ctx->seg_positions / MRZ_TILE;
    if (tiles > max_tiles) tiles = max_tiles;
    if (tiles < 1) tiles = 1;
    if (entries > ctx->cand_cap) entries = ctx->cand_cap;
    if (entries < MRZ_TILE) entries = MRZ_TILE;
    if (tiles > ctx->fe_tiles_cap) {
        int64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        if (ctx->d_bitmap) hipFree(ctx->d_bitmap), ctx->d_bitmap = nullptr;
        if (ctx->d_tile_cnt) hipFree(ctx->d_tile_cnt), ctx->d_tile_cnt = nullptr;
        if (ctx->d_tile_off) hipFree(ctx->d_tile_off), ctx->d_tile_off = nullptr;
        if (ctx->d_grp_cnt) hipFree(ctx->d_grp_cnt), ctx->d_grp_cnt = nullptr;
        ctx->fe_tiles_cap = 0;
        int rc = mrz_grow(ctx, &ctx->d_bitmap, &c0, tiles * (MRZ_TILE / 16) + 64);
        if (!rc) rc = mrz_grow(ctx, &ctx->d_tile_cnt, &c1, tiles + 1);
        if (!rc) rc = mrz_grow(ctx, &ctx->d_tile_off, &c2, tiles + 2);
        if (!rc) rc = mrz_grow(ctx, &ctx->d_grp_cnt, &c3, tiles / MRZ_FE_GROUP + 2);
        if (rc) return rc;
        ctx->fe_tiles_cap = tiles;
    }
    return mrz_grow(ctx, &ctx->d_cand, &ctx->cand_alloc, entries);
}

extern "C" int mrz_window_scan(mrz_ctx *ctx, const void *range_bytes, int64_t range_len, int where, int64_t range_start,
                               int64_t chunk_n, int64_t seg_start, int64_t max_span, int64_t min_mask, int64_t p_done,
                               int64_t cap, mrz_candidate *cand_out, int32_t *tile_off_out, void *bitmap_out, int out_where,
                               int64_t *scan_next, int64_t *n_cand) {
    if (!ctx || !range_bytes || !cand_out || !tile_off_out || !bitmap_out || !scan_next || !n_cand) return MRZ_E_ARG;
    if (range_len <= 0 || range_start < 0 || chunk_n <= 0 || cap < MRZ_TILE) return MRZ_E_ARG;
    if (max_span <= 0 || max_span > ctx->seg_positions || max_span % MRZ_TILE || seg_start % MRZ_TILE || seg_start < range_start)
        return MRZ_E_ARG;
    if (out_where != MRZ_MEM_HOST && out_where != MRZ_MEM_DEVICE) return MRZ_E_ARG;
    // the last position's 31-byte window must lie in the rank's bytes (or the chunk ends first); the kernels stage whole
    // 16-byte pieces: 48 bytes of halo keep every load inside the rank's bytes
    const int64_t end = chunk_n - MRZ_MIN_MATCH;
    int64_t last_pos = seg_start + max_span - 1;
    if (last_pos > end) last_pos = end;
    const int64_t need_end = last_pos + 48 < chunk_n ? last_pos + 48 : chunk_n;
    if (last_pos >= seg_start && need_end > range_start + range_len) return MRZ_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint8_t *d_range = nullptr;
    int rc = mrz_stage_input(ctx, range_bytes, range_len, where, &d_range);
    if (rc) return rc;
    const int64_t tiles = max_span / MRZ_TILE;
    const int64_t save_cap = ctx->cand_cap;
    if (cap > ctx->cand_cap) ctx->cand_cap = cap;
    rc = mrz_fe_reserve(ctx, tiles, cap);
    ctx->cand_cap = save_cap;
    if (rc) return rc;
    mrz_seq_state hs;
    memset(&hs, 0, sizeof(hs));
    hs.n = chunk_n;
    hs.end = end;
    // (a matcher position inside or beyond the stretch's first tile would move the pass's first tile: the outputs are
    // laid out from seg_start, so such a position is not used as a filter -- the list is a superset then)
    hs.p = p_done + 1 < seg_start + MRZ_TILE ? p_done : seg_start - 1;
    hs.min_mask = min_mask;
    hs.scan_next = seg_start;
    hipStream_t s = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_state, &hs, sizeof(hs), hipMemcpyHostToDevice, s));
    // the kernels index the chunk by absolute position: give them the pointer position 0 would have.  They stay
    // inside [range_start, need_end + 16) of it; the staging buffer is padded by 64 bytes.
    HIPCHK(ctx, mrz_launch_frontend(s, d_range - range_start, chunk_n, ctx->d_index, ctx->d_state, (int)tiles, cap,
                                    ctx->d_fe_hdr, ctx->d_bitmap, ctx->d_tile_cnt, ctx->d_tile_off, ctx->d_grp_cnt, ctx->d_cand));
    HIPCHK(ctx, hipMemcpyAsync(&hs, ctx->d_state, sizeof(hs), hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    int64_t T = (hs.scan_next - seg_start) / MRZ_TILE;  // tiles the pass covered (none: the matcher is past the stretch)
    if (hs.seg_start != seg_start || T < 0 || T > tiles) {
        // (p_done lies beyond the stretch: the pass began further on; report the stretch as empty)
        T = 0;
        hs.n_cand = 0;
        hs.scan_next = hs.scan_next > seg_start ? hs.scan_next : seg_start;
    }
    const hipMemcpyKind k = out_where == MRZ_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (hs.n_cand > 0) HIPCHK(ctx, hipMemcpyAsync(cand_out, ctx->d_cand, (size_t)hs.n_cand * sizeof(mrz_cand), k, s));
    if (T > 0) {
        HIPCHK(ctx, hipMemcpyAsync(tile_off_out, ctx->d_tile_off, (size_t)(T + 1) * sizeof(int), k, s));
        HIPCHK(ctx, hipMemcpyAsync(bitmap_out, ctx->d_bitmap, (size_t)T * (MRZ_TILE / 8), k, s));
    }
    HIPCHK(ctx, hipStreamSynchronize(s));
    *scan_next = hs.scan_next;
    *n_cand = hs.n_cand;
    return MRZ_OK;
}

extern "C" int mrz_set_farm_helpers(mrz_ctx *ctx, int n) {
    if (!ctx) return MRZ_E_ARG;
    ctx->farm_helpers = n < 0 ? -1 : n;
    return MRZ_OK;
}

extern "C" int mrz_get_timings(const mrz_ctx *ctx, mrz_timings *out) {
    if (!ctx || !out) return MRZ_E_ARG;
    *out = ctx->timings;
    return MRZ_OK;
}

extern "C" int64_t mrz_table_slots(const mrz_ctx *ctx) { return ctx ? ctx->nslots : 0; }

extern "C" int mrz_fetch_table(mrz_ctx *ctx, void *host_dst) {
    if (!ctx || !host_dst) return MRZ_E_ARG;
    if (!ctx->have_chunk) return MRZ_E_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpy(host_dst, ctx->d_tab, (size_t)ctx->nslots * sizeof(mrz_slot), hipMemcpyDeviceToHost));
    return MRZ_OK;
}

// resolves a caller buffer to a device pointer (staging host memory)
int mrz_stage_input(mrz_ctx *ctx, const void *buf, int64_t n, int where, const uint8_t **dev) {
    if (where == MRZ_MEM_DEVICE) {
        *dev = (const uint8_t *)buf;
        return MRZ_OK;
    }
    if (where != MRZ_MEM_HOST) return MRZ_E_ARG;
    int rc = mrz_grow(ctx, &ctx->d_in, &ctx->in_cap, n + 64);
    if (rc) return rc;
    if (n > 0) HIPCHK(ctx, hipMemcpyAsync(ctx->d_in, buf, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    *dev = ctx->d_in;
    return MRZ_OK;
}

extern "C" int mrz_crc32(mrz_ctx *ctx, const void *buf, int64_t n, int where, uint32_t *crc_out) {
    if (!ctx || !crc_out || n < 0 || (n > 0 && !buf)) return MRZ_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint8_t *d = nullptr;
    int rc = mrz_stage_input(ctx, buf, n, where, &d);
    if (rc) return rc;
    rc = mrz_grow(ctx, &ctx->d_crc_parts, &ctx->crc_parts_cap, mrz_crc32_parts_needed(n));
    if (rc) return rc;
    HIPCHK(ctx, mrz_launch_crc32(ctx->stream, d, n, ctx->d_crc_tables, ctx->d_crc_parts, ctx->d_crc_out));
    HIPCHK(ctx, hipMemcpyAsync(crc_out, ctx->d_crc_out, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MRZ_OK;
}

struct mrz_evpair {
    hipEvent_t a, b;
    int kind;  // 0 front end, 1 sequencer, 2 encode, 3 crc
};

// positions a front-end pass should look at under a mask of k bits so that its list comes out about 3/4 full
static int64_t mrz_span_for_mask(const mrz_ctx *ctx, int64_t mask) {
    const int k = __builtin_popcountll((unsigned long long)mask);
    int64_t span = (ctx->cand_cap - ctx->cand_cap / 4) << (k < 24 ? k : 24);
    if (span > ctx->seg_positions) span = ctx->seg_positions;
    span = span / MRZ_TILE * MRZ_TILE;
    return span < MRZ_TILE ? MRZ_TILE : span;
}

extern "C" int mrz_rzip_chunk(mrz_ctx *ctx, const void *chunk, int64_t n, int where, int chunk_bytes,
                              int64_t *victim_round, mrz_chunk_result *res) {
    if (!ctx || !res || !victim_round || n < 0 || (n > 0 && !chunk) || chunk_bytes < 1 || chunk_bytes > 8)
        return MRZ_E_ARG;
    if (*victim_round < 0 || *victim_round >= (int64_t)ctx->max_chain) return MRZ_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    memset(res, 0, sizeof(*res));
    memset(&ctx->timings, 0, sizeof(ctx->timings));
    ctx->have_chunk = 0;
    hipStream_t s = ctx->stream;

    const uint8_t *d_buf = nullptr;
    int rc = mrz_stage_input(ctx, chunk, n, where, &d_buf);
    if (rc) return rc;
    rc = mrz_grow(ctx, &ctx->d_events, &ctx->event_cap, mrz_event_room(n));
    if (rc) return rc;
    rc = mrz_grow(ctx, &ctx->d_crc_parts, &ctx->crc_parts_cap, mrz_crc32_parts_needed(n));
    if (rc) return rc;
    const int64_t end = n - MRZ_MIN_MATCH;  // last position that is looked up (src/rzip.c:544)
    if (end > 0) {
        rc = mrz_fe_reserve(ctx, end / MRZ_TILE + 2, end + 1);
        if (rc) return rc;
    }

    // profiling events (optional)
    mrz_evpair *evs = nullptr;
    int nev = 0, evcap = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    if (ctx->profiling) {
        hipEventCreate(&ev_begin);
        hipEventCreate(&ev_end);
        hipEventRecord(ev_begin, s);
    }
#define PROF_BEGIN(k)                                                                       \
    do {                                                                                    \
        if (ctx->profiling) {                                                               \
            if (nev == evcap) {                                                             \
                const int ncap__ = evcap ? evcap * 2 : 64;                                  \
                mrz_evpair *ne__ = (mrz_evpair *)realloc(evs, (size_t)ncap__ * sizeof(mrz_evpair)); \
                if (ne__) evs = ne__, evcap = ncap__;                                       \
            }                                                                               \
            if (nev < evcap) {                                                              \
                evs[nev].kind = (k);                                                        \
                hipEventCreate(&evs[nev].a);                                                \
                hipEventCreate(&evs[nev].b);                                                \
                hipEventRecord(evs[nev].a, s);                                              \
            }                                                                               \
        }                                                                                   \
    } while (0)
#define PROF_END()                                     \
    do {                                               \
        if (ctx->profiling && nev < evcap) {           \
            hipEventRecord(evs[nev].b, s);             \
            nev++;                                     \
        }                                              \
    } while (0)

    // hash_search prologue (src/rzip.c:518-546): zero the table, reset state
    mrz_seq_state hs;
    memset(&hs, 0, sizeof(hs));
    hs.n = n;
    hs.end = end;
    hs.min_mask = hs.tag_mask = (1ll << ctx->initial_freq) - 1;
    hs.limit = ctx->nslots / 3 * 2;
    hs.victim_round = *victim_round;
    hs.max_chain = ctx->max_chain;
    hs.slot_mask = ctx->nslots - 1;
    hs.event_cap = ctx->event_cap;
    hs.finished = end > 0 ? 0 : 1;
    hipError_t herr = hipSuccess;
    int64_t E = 0;
    uint32_t crc = 0;
    mrz_enc_totals tot;
    memset(&tot, 0, sizeof(tot));

#define STEP(expr)                       \
    do {                                 \
        if (herr == hipSuccess) herr = (expr); \
    } while (0)

    STEP(hipMemsetAsync(ctx->d_tab, 0, (size_t)ctx->nslots * sizeof(mrz_slot), s));
    STEP(hipMemcpyAsync(ctx->d_state, &hs, sizeof(hs), hipMemcpyHostToDevice, s));
    STEP(hipMemsetAsync(ctx->d_totals, 0, sizeof(mrz_enc_totals), s));

    PROF_BEGIN(3);
    STEP(mrz_launch_crc32(s, d_buf, n, ctx->d_crc_tables, ctx->d_crc_parts, ctx->d_crc_out));
    PROF_END();

    // ---- the segments: a front-end pass (candidate list of the next stretch) and a sequencer launch over it, again
    // and again until the matcher reports the end of the chunk.  WHERE a pass begins and ends is device state (it goes
    // on behind the last one, skips what an emitted match has covered, and stops where its list is full); the host
    // only bounds the span of a pass -- from the mask the matcher has reached: the tighter the mask, the more
    // positions a list of the same size covers -- and keeps MRZ_SEG_AHEAD segments queued.  Every launch leaves a
    // snapshot of the matcher state in a ring of pinned host slots (ONE copy per launch: position, masks, progress and
    // the `finished` flag belong together); the host reads a slot once the launch's event has completed.
    ctx->events_final = 0;
    const size_t snap_bytes = offsetof(mrz_seq_state, prof);
    hipEvent_t seg_ev[MRZ_SEG_AHEAD];
    int n_seg_ev = 0;
    for (int k = 0; k < MRZ_SEG_AHEAD && herr == hipSuccess; k++) {
        STEP(hipEventCreateWithFlags(&seg_ev[k], hipEventDisableTiming));
        if (herr == hipSuccess) n_seg_ev++;
    }
    int64_t launched = 0, retired = 0, n_narrow = 0, n_deep = 0;
    int64_t span_of[MRZ_SEG_AHEAD];      // positions the pass of an in-flight launch may cover
    int64_t known_next = 0;              // where the pass after the last retired launch begins, as far as the host knows
    int64_t known_mask = hs.min_mask, known_p = 0;
    int64_t hint_pos = 0, hint_matched = 0;
    bool finished = end <= 0;
    // a launch whose event has completed: its snapshot is final
    auto retire = [&]() {
        const mrz_seq_state *sn = &ctx->h_ring[retired % MRZ_SEG_AHEAD];
        retired++;
        known_p = sn->p;
        known_mask = sn->min_mask;
        if (!ctx->cand_fn)
            known_next = sn->scan_next;
        else if (sn->seg_end > sn->seg_start && sn->scan_next < sn->seg_end)
            // (window sharding, where the host drives the geometry: the wide engine has ended this launch where the mask
            // reached the deep engine's regime and taken scan_next back to its position -- the next stretch is asked for
            // from there)
            known_next = sn->scan_next;
        hint_pos = sn->hint_positions;
        hint_matched = sn->hint_matched;
        if (ctx->progress_fn) {
            ctx->events_final = sn->n_events;
            if (ctx->progress_fn(ctx->progress_user, sn->n_events, sn->last_match, 0)) rc = MRZ_E_STATE;
        }
        if (sn->finished || sn->error) finished = true;
    };
    while (!finished && herr == hipSuccess && !rc) {
        // whatever has completed meanwhile (the engine choice and the span below want the matcher's latest news)
        while (retired < launched && !finished && !rc && hipEventQuery(seg_ev[retired % MRZ_SEG_AHEAD]) == hipSuccess) retire();
        if (finished || rc) break;
        // where the queued passes will have got to if none of them is cut short
        int64_t est_next = known_next;
        for (int64_t k = retired; k < launched; k++) est_next += span_of[k % MRZ_SEG_AHEAD];
        const bool queue_full = launched - retired >= MRZ_SEG_AHEAD;
        const bool all_queued = est_next > end;  // (only a list that fills up -- or the provider -- can prove this wrong)
        if (queue_full || (all_queued && launched > retired)) {
            STEP(hipEventSynchronize(seg_ev[retired % MRZ_SEG_AHEAD]));  // wait for the oldest launch
            if (herr != hipSuccess) break;
            retire();
            continue;
        }
        if (all_queued) {
            // nothing in flight and, by the host's book-keeping, nothing left -- yet the matcher has not reported the end:
            // cannot happen (a pass always covers its span unless its list fills, and then known_next says so)
            rc = MRZ_E_OVERFLOW;
            break;
        }
        // ---- one more segment
        int64_t span = mrz_span_for_mask(ctx, known_mask);
        int64_t max_tiles = span / MRZ_TILE;
        if (max_tiles > ctx->fe_tiles_cap) max_tiles = ctx->fe_tiles_cap;
        span = max_tiles * MRZ_TILE;
        PROF_BEGIN(0);
        if (ctx->cand_fn) {
            // window sharding: the stretch's owner scans it (with the mask this rank has last heard of); the host drives
            // the geometry here, and hands it to the sequencer through the matcher state
            int64_t seg_start = known_next;
            const int64_t pt = (known_p + 1) / MRZ_TILE * MRZ_TILE;
            if (pt > seg_start) seg_start = pt;
            if (seg_start > end) {  // (the matcher is about to report the end)
                known_next = seg_start;
                span_of[launched % MRZ_SEG_AHEAD] = 0;
            } else {
                if (seg_start + span > end + 1) span = (end + 1 - seg_start + MRZ_TILE - 1) / MRZ_TILE * MRZ_TILE;
                int64_t geo[5] = { seg_start, 0, 0, 0, known_mask };  // seg_start, seg_end, n_cand, scan_next, list_mask
                int64_t nx = seg_start, nc = 0;
                if (herr == hipSuccess && ctx->cand_fn(ctx->cand_user, seg_start, span, known_mask, known_p, ctx->cand_cap,
                                                       (mrz_candidate *)ctx->d_cand, ctx->d_tile_off, ctx->d_bitmap, &nx, &nc,
                                                       (void *)s)) {
                    hipStreamSynchronize(s);
                    rc = MRZ_E_STATE;
                    break;
                }
                if (nx <= seg_start || nx > seg_start + span || nx % MRZ_TILE || nc < 0 || nc > ctx->cand_cap) {
                    hipStreamSynchronize(s);
                    rc = MRZ_E_STATE;
                    break;
                }
                geo[1] = nx < end + 1 ? nx : end + 1;
                geo[2] = nc;
                geo[3] = nx;
                STEP(hipMemcpyAsync(&ctx->d_state->seg_start, geo, sizeof(geo), hipMemcpyHostToDevice, s));
                STEP(hipStreamSynchronize(s));  // (geo lives on this stack frame)
                known_next = nx;
                span_of[launched % MRZ_SEG_AHEAD] = 0;
            }
        } else {
            STEP(mrz_launch_frontend(s, d_buf, n, ctx->d_index, ctx->d_state, (int)max_tiles, ctx->cand_cap, ctx->d_fe_hdr,
                                     ctx->d_bitmap, ctx->d_tile_cnt, ctx->d_tile_off, ctx->d_grp_cnt, ctx->d_cand));
            span_of[launched % MRZ_SEG_AHEAD] = span;
        }
        PROF_END();
        // Which engine: the wide one (512 candidates per batch) unless the segments before were one long match after
        // another (>= 80 % of the positions a launch advanced over lay inside the matches it emitted): then the
        // narrow engine's shorter chain per match wins.  The hint lags behind like everything the host knows; the first
        // two launches are waited for so that it arrives early.  MRZ_SEQ_ENGINE=wide|narrow pins the choice (tests,
        // measurements).
        const int mask_bits = __builtin_popcountll((unsigned long long)known_mask);
        bool narrow = hint_pos > 0 && hint_matched * 10 >= hint_pos * 8 && mask_bits < ctx->narrow_max_bits;
        // ... and the deep engine once the cull sweeps have tightened the mask: the table then consists of a few long
        // runs (2^(hash_bits - k) of about 2/3 x 2^k slots under a k-bit mask) that every look-up reads to the end --
        // streaming scans, not the short walks the wide engine's lanes are made for
        bool deep = !narrow && mask_bits >= ctx->deep_min_bits;
        if (ctx->engine_pin) narrow = ctx->engine_pin == 2, deep = ctx->engine_pin == 3;
        const int helpers = ctx->farm_helpers >= 0 && ctx->farm_helpers < ctx->farm_default ? ctx->farm_helpers : ctx->farm_default;
        PROF_BEGIN(1);
        if (narrow)
            STEP(mrz_launch_sequencer_narrow(s, d_buf, ctx->d_tab, ctx->d_cand, ctx->d_tile_off, (const mrz_u64 *)ctx->d_bitmap,
                                             ctx->d_events, ctx->d_state, ctx->d_gmailbox, helpers, ctx->xcd));
        else if (deep)
            STEP(mrz_launch_sequencer_deep(s, d_buf, ctx->d_tab, ctx->d_cand, ctx->d_tile_off, (const mrz_u64 *)ctx->d_bitmap,
                                           ctx->d_events, ctx->d_state, ctx->d_gmailbox, helpers, ctx->xcd, ctx->d_deep_shared,
                                           ctx->deep_scanners));
        else
            STEP(mrz_launch_sequencer(s, d_buf, ctx->d_tab, ctx->d_cand, ctx->d_tile_off, (const mrz_u64 *)ctx->d_bitmap,
                                      ctx->d_events, ctx->d_state, ctx->d_gmailbox, helpers, ctx->d_seq_shared, ctx->d_wlog,
                                      ctx->nslots, ctx->seq_wgs, ctx->xcd, ctx->engine_pin ? 0 : ctx->deep_min_bits));
        PROF_END();
        if (narrow) n_narrow++;
        if (deep) n_deep++;
        STEP(hipMemcpyAsync(&ctx->h_ring[launched % MRZ_SEG_AHEAD], ctx->d_state, snap_bytes, hipMemcpyDeviceToHost, s));
        STEP(hipEventRecord(seg_ev[launched % MRZ_SEG_AHEAD], s));
        launched++;
        if (launched <= 2 && !ctx->engine_pin && !ctx->cand_fn) STEP(hipStreamSynchronize(s));
    }
    hipStreamSynchronize(s);  // (also after a refusal of the progress hook: the queued launches run out)
    for (int k = 0; k < n_seg_ev; k++) hipEventDestroy(seg_ev[k]);
    STEP(hipMemcpyAsync(&hs, ctx->d_state, sizeof(hs), hipMemcpyDeviceToHost, s));
    STEP(hipMemcpyAsync(&crc, ctx->d_crc_out, 4, hipMemcpyDeviceToHost, s));
    STEP(hipStreamSynchronize(s));
    if (herr == hipSuccess) {
        if (!rc && (hs.error || !hs.finished)) {
            fprintf(stderr,
                    "libmrzgpu: sequencer stopped abnormally: error=%d finished=%d p=%lld end=%lld events=%lld/%lld "
                    "count=%lld min_mask=%lld scan_next=%lld launches=%lld\n",
                    hs.error, hs.finished, (long long)hs.p, (long long)hs.end, (long long)hs.n_events,
                    (long long)hs.event_cap, (long long)hs.count, (long long)hs.min_mask, (long long)hs.scan_next,
                    (long long)launched);
            rc = MRZ_E_OVERFLOW;
        }
        E = hs.n_events;
        if (!rc && ctx->progress_fn) {
            ctx->events_final = E;
            if (ctx->progress_fn(ctx->progress_user, E, hs.last_match, 1)) rc = MRZ_E_STATE;
        }
    }

    // record encoding
    if (herr == hipSuccess && !rc) {
        const int64_t nblocks = (E + 1 + 255) / 256;
        rc = mrz_grow(ctx, &ctx->d_block_s0, &ctx->block_cap, nblocks);
        if (!rc) rc = mrz_grow(ctx, &ctx->d_block_s1, &ctx->block1_cap, nblocks);
        if (!rc) rc = mrz_grow(ctx, &ctx->d_lit_off, &ctx->lit_off_cap, E + 2);
    }
    if (herr == hipSuccess && !rc) {
        PROF_BEGIN(2);
        STEP(mrz_launch_enc_size(s, ctx->d_events, E, n, chunk_bytes, ctx->d_block_s0, ctx->d_block_s1, ctx->d_totals));
        STEP(hipMemcpyAsync(&tot, ctx->d_totals, sizeof(tot), hipMemcpyDeviceToHost, s));
        STEP(hipStreamSynchronize(s));
        if (herr == hipSuccess) {
            rc = mrz_grow(ctx, &ctx->d_s0, &ctx->s0_cap, tot.s0_len + 7 + 16);
            if (!rc) rc = mrz_grow(ctx, &ctx->d_s1, &ctx->s1_cap, tot.s1_len + 16);
        }
        if (herr == hipSuccess && !rc) {
            STEP(mrz_launch_enc_write(s, d_buf, ctx->d_events, E, n, chunk_bytes, ctx->d_block_s0, ctx->d_block_s1,
                                      ctx->d_s0, ctx->d_s1, tot.s1_len, ctx->d_lit_off, ctx->d_totals, crc));
            STEP(hipMemcpyAsync(&tot, ctx->d_totals, sizeof(tot), hipMemcpyDeviceToHost, s));
        }
        PROF_END();
        if (ctx->profiling) hipEventRecord(ev_end, s);
        STEP(hipStreamSynchronize(s));
    }

    ctx->timings.n_segments = (int32_t)launched;
    ctx->timings.n_narrow = (int32_t)n_narrow;
    ctx->timings.n_deep = (int32_t)n_deep;
    if (ctx->profiling) {
        hipStreamSynchronize(s);
        for (int i = 0; i < nev; i++) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, evs[i].a, evs[i].b) == hipSuccess) {
                if (evs[i].kind == 0) ctx->timings.tagscan_ms += ms;
                if (evs[i].kind == 1) ctx->timings.sequencer_ms += ms;
                if (evs[i].kind == 2) ctx->timings.encode_ms += ms;
                if (evs[i].kind == 3) ctx->timings.crc_ms += ms;
            }
            hipEventDestroy(evs[i].a);
            hipEventDestroy(evs[i].b);
        }
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev_begin, ev_end) == hipSuccess) ctx->timings.total_ms = ms;
        hipEventDestroy(ev_begin);
        hipEventDestroy(ev_end);
        free(evs);
    }
#undef PROF_BEGIN
#undef PROF_END
#undef STEP
    if (herr != hipSuccess) {
        ctx->last_err = herr;
        return MRZ_E_HIP;
    }
    if (rc) return rc;

    *victim_round = hs.victim_round;
    ctx->s0_len = tot.s0_len + 7;
    ctx->s1_len = tot.s1_len;
    ctx->have_chunk = 1;
    res->s0_len = ctx->s0_len;
    res->s1_len = ctx->s1_len;
    res->crc32 = crc;
    res->d_s0 = ctx->d_s0;
    res->d_s1 = ctx->d_s1;
    res->stats.inserts = hs.inserts;
    res->stats.tag_hits = hs.tag_hits;
    res->stats.tag_misses = hs.tag_misses;
    res->stats.literals = tot.literals + 1;  // the zero-length terminator counts (src/rzip.c:219,664)
    res->stats.literal_bytes = tot.literal_bytes;
    res->stats.matches = tot.matches;
    res->stats.match_bytes = tot.match_bytes;
    res->min_mask = hs.min_mask;
    res->hash_count = hs.count;
    res->n_events = E;
    if (ctx->print_prof) {
        // (one list for both engines: the enum of mrz_seq_common.h)
        static const char *names[128] = { "batches", "formed", "committed", "segments", "emits", "backjump", "rewalk",
                                          "longres", "seq_cands", "cut_cplx", "cut_overflow", "skipout", "conf0", "pairs",
                                          "t_form", "t_walk", "t_conf", "t_pairs", "t_loop", "t_rewalk", "t_long", "t_seq",
                                          "farmed", "l_post", "l_stripe", "l_bwd", "l_wait", "l_rounds", "f_post", "f_wait",
                                          "f_fold", "s_tab", "s_pair", "s_ins", "ovl", "ovl_ok", "x_walk", "x_casc", "x_pool",
                                          "x_win", "x_same", "c_win", "c_evict", "c_deep", "c_many", "c_fail", "c_tie", "c_nw", "t_ovl",
                                          "h_pre", "h_cand", "h_post", "t_scan", "t_fold", "t_commit", "reprep", "w_stale", "w_drop", "reset",
                                          "t_turn", "t_prep", "t_precommit", "e_mask", "e_cull", "e_xw", "e_inwin", "e_window", "e_bulk",
                                          "e_more", "t_pc_cw", "t_pc_log", "t_pc_best", "t_pc_bulk", "t_turnwork", "t_snap", "rebulk",
                                          "batch_lanes", "cut_long", "cut_walk", "cut_conflict", "cut_cull", "batch_emits",
                                          "cut_cascade", "batch_formed", "t_walk2", "t_scans", "t_conflict", "t_window",
                                          "d_batches", "d_lanes", "d_rounds", "d_rescanned", "d_coop", "d_t_form", "d_t_scan",
                                          "d_t_commit", "d_t_rescan", "d_t_total", "d_launches", "d_t_coop", "d_coop_rec",
                                          "d_resolved", "d_s_coop", "d_s_conflict", "d_s_culled", "d_s_nocull", "d_s_stale", "d_c_over_alt", "d_c_over_noalt", "d_c_empty",
                                          "d_c_displace", "d_c_other" };
        for (int k = 0; k < 128; k++)
            if (names[k] && hs.prof[k]) fprintf(stderr, "seqstat %-12s %lld\n", names[k], (long long)hs.prof[k]);
    }
    return MRZ_OK;
}

extern "C" int mrz_fetch_streams(mrz_ctx *ctx, uint8_t *s0_host, uint8_t *s1_host) {
    if (!ctx) return MRZ_E_ARG;
    if (!ctx->have_chunk) return MRZ_E_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (s0_host && ctx->s0_len > 0)
        HIPCHK(ctx, hipMemcpyAsync(s0_host, ctx->d_s0, (size_t)ctx->s0_len, hipMemcpyDeviceToHost, ctx->stream));
    if (s1_host && ctx->s1_len > 0)
        HIPCHK(ctx, hipMemcpyAsync(s1_host, ctx->d_s1, (size_t)ctx->s1_len, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MRZ_OK;
}
