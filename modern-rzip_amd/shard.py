"""Multi-GPU sharding of the rzip stage (one process per GPU, torch.distributed).

What shards in the reference's path:

* **streams / files** -- every `mrzip` invocation starts from a zeroed table and
  `victim_round = 0`; N streams on N GPUs are N independent units: no data-path
  collective at all (this is what bench.py --gpus N measures, weak scaling);
* **chunks of one file** -- each chunk has its own table, streams and CRC
  (src/rzip.c:518-519,769,998,666), but the process-lifetime `static
  victim_round` (src/rzip.c:259) threads through them in order.  Chunks are dealt
  round-robin to ranks; the one integer is handed from the owner of chunk k to the
  owner of chunk k+1 point-to-point, and rank 0 gathers the per-chunk streams in
  chunk order.  The result is bit-identical to the single-process run; how much
  of it overlaps depends on how early a chunk's victim_round is known.

* **one window over several GPUs** (`-U`, BASELINE configs[3]) -- the front end shards by byte
  range: every rank scans the tags of the segments inside its range (31-byte window: a 30-byte
  halo) and ships them to the rank that runs the exact matcher, which takes them instead of
  scanning (mrz_set_tag_provider); CRC-32 is checksummed per range and combined over GF(2).
  The matcher itself -- one dependency chain through the hash table -- does not shard; see
  rzip_chunk_window below and DESIGN.md section 6 for what this scales and what it does not.

The functions take the rank's compute callables so that the CPU test tier can drive
them over gloo with the emulated library.
"""


def streams_of_rank(n_streams, rank, world):
    """Indices of the independent streams rank `rank` compresses (round-robin)."""
    return list(range(rank, n_streams, world))


def chunk_owner(k, world):
    return k % world


def split_chunks(total, max_chunk):
    """(offset, size) of every chunk of a file of `total` bytes (src/rzip.c:915-929);
    an empty file still has one empty chunk."""
    out, off = [], 0
    while True:
        size = min(max_chunk, total - off)
        out.append((off, size))
        off += size
        if off >= total:
            return out


def rzip_file_chunk_chain(data, max_chunk, rank, world, rzip_chunk, dist=None):
    """Runs the chunks this rank owns, chaining victim_round through the ranks.

    rzip_chunk(chunk_bytes, victim_round_in) -> (s0, s1, victim_round_out)
    Returns on rank 0 the list [(chunk_size, s0, s1), ...] in chunk order (ready for
    the stream sink), None elsewhere.
    """
    import torch
    chunks = split_chunks(len(data), max_chunk)
    mine = {}
    vr = 0
    for k, (off, size) in enumerate(chunks):
        owner = chunk_owner(k, world)
        prev_owner = chunk_owner(k - 1, world) if k else owner
        if owner == rank:
            if k and prev_owner != rank:
                t = torch.zeros(1, dtype=torch.int64)
                dist.recv(t, src=prev_owner)
                vr = int(t.item())
            s0, s1, vr = rzip_chunk(data[off:off + size], vr)
            mine[k] = (size, s0, s1)
            nxt = chunk_owner(k + 1, world)
            if k + 1 < len(chunks) and nxt != rank:
                dist.send(torch.tensor([vr], dtype=torch.int64), dst=nxt)
    if world == 1:
        return [mine[k] for k in range(len(chunks))]
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)
    if rank != 0:
        return None
    merged = {}
    for part in gathered:
        merged.update(part)
    return [merged[k] for k in range(len(chunks))]


def rzip_file_chunk_chain_speculative(data, max_chunk, rank, world, rzip_chunk, dist=None, predict=None):
    """The same result as rzip_file_chunk_chain, but the ranks do not wait for each other: every rank first runs all
    its chunks with a PREDICTED victim_round, at the same time as the other ranks; then the true value is chained
    through the chunks in order and only a chunk whose prediction was wrong is run again.  The
    integer only moves on chain-limit evictions (src/rzip.c:284-289), so on streams without heavy tag repetition the
    prediction `unchanged since the start of the file` holds and the chunks really run in parallel; on text it
    mostly fails and the cost is one wasted run per chunk, in parallel with the others.

    predict(k) -> predicted victim_round at the start of chunk k (default: 0 for every chunk).
    Returns on rank 0 ([(chunk_size, s0, s1), ...] in chunk order, number of re-runs over all ranks); None elsewhere.
    """
    import torch
    chunks = split_chunks(len(data), max_chunk)
    predict = predict or (lambda k: 0)
    spec = {}
    for k, (off, size) in enumerate(chunks):  # phase 1: everybody at once
        if chunk_owner(k, world) == rank:
            vin = predict(k)
            s0, s1, vout = rzip_chunk(data[off:off + size], vin)
            spec[k] = (vin, size, s0, s1, vout)
    mine, reruns, vr = {}, 0, 0
    for k, (off, size) in enumerate(chunks):  # phase 2: the true value, in chunk order
        owner = chunk_owner(k, world)
        prev_owner = chunk_owner(k - 1, world) if k else owner
        if owner != rank:
            continue
        if k and prev_owner != rank:
            t = torch.zeros(1, dtype=torch.int64)
            dist.recv(t, src=prev_owner)
            vr = int(t.item())
        vin, size, s0, s1, vout = spec[k]
        if vin != vr:  # mispredicted: once more with the true value
            s0, s1, vout = rzip_chunk(data[off:off + size], vr)
            reruns += 1
        mine[k] = (size, s0, s1)
        vr = vout
        nxt = chunk_owner(k + 1, world)
        if k + 1 < len(chunks) and nxt != rank:
            dist.send(torch.tensor([vr], dtype=torch.int64), dst=nxt)
    if world == 1:
        return [mine[k] for k in range(len(chunks))], reruns
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((mine, reruns), gathered, dst=0)
    if rank != 0:
        return None
    merged, total = {}, 0
    for part, r in gathered:
        merged.update(part)
        total += r
    return [merged[k] for k in range(len(chunks))], total


# ---- range-sharded front-end of ONE chunk (second row) ------------------------
# What shards inside a single window is everything except the sequencer: tag scan, CRC and literal gather work
# on contiguous byte ranges (31-byte halo for the tags).  The pieces below are the host-side glue that is
# independent of the transport: the byte ranges, and the combination of per-range CRC-32 values into the
# chunk's CRC (src/rzip.c:662) -- CRC is linear over GF(2), so crc(A || B) follows from crc(A), crc(B) and
# len(B) without touching the bytes again (the GPU kernel does the same across its tiles).

def byte_ranges(total, world):
    """Contiguous, 4 KiB-aligned byte range of every rank: [(offset, size)] * world (sizes may be 0)."""
    per = -(-total // world)
    per = -(-per // 4096) * 4096
    out = []
    for r in range(world):
        off = min(r * per, total)
        out.append((off, min(per, total - off)))
    return out


def _gf2_times(mat, vec):
    s, i = 0, 0
    while vec:
        if vec & 1:
            s ^= mat[i]
        vec >>= 1
        i += 1
    return s


def _gf2_square(mat):
    return [_gf2_times(mat, mat[i]) for i in range(32)]


def crc32_combine(crc_a, crc_b, len_b):
    """CRC-32 (IEEE, as zlib / libgcrypt GCRY_MD_CRC32) of A || B from crc(A), crc(B) and len(B)."""
    if len_b <= 0:
        return crc_a
    odd = [0xEDB88320] + [1 << i for i in range(31)]  # operator for one zero BIT
    even = _gf2_square(odd)                            # two bits
    odd = _gf2_square(even)                            # four bits
    n = len_b
    while True:  # apply len_b zero BYTES to crc_a
        even = _gf2_square(odd)
        if n & 1:
            crc_a = _gf2_times(even, crc_a)
        n >>= 1
        if not n:
            break
        odd = _gf2_square(even)
        if n & 1:
            crc_a = _gf2_times(odd, crc_a)
        n >>= 1
        if not n:
            break
    return crc_a ^ crc_b


def chunk_crc_sharded(data, rank, world, crc32_of, dist=None):
    """Every rank checksums its byte range with crc32_of(bytes) -> int; rank 0 returns the chunk's CRC-32."""
    off, size = byte_ranges(len(data), world)[rank]
    mine = (crc32_of(data[off:off + size]) & 0xFFFFFFFF, size)
    parts = [mine]
    if dist is not None and world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, mine)
    if rank != 0:
        return None
    crc = 0
    for c, n in parts:
        crc = crc32_combine(crc, c, n)
    return crc


# ---- ONE window over several ranks: range-sharded tag scan, matcher on rank 0 (SURVEY.md 8e, configs[3]) --------
def segment_ranges(total, world, seg_positions):
    """Byte range of every rank, made of whole segments (so that every segment has one owner): [(offset, size)]."""
    nseg = max(1, -(-max(total - 30, 1) // seg_positions))  # positions 0 .. total - 31 have a tag (end inclusive)
    per = -(-nseg // world)
    out = []
    for r in range(world):
        off = min(r * per * seg_positions, total)
        out.append((off, min(per * seg_positions, total - off)))
    return out


def rzip_chunk_window(ctx, my_bytes, my_off, total, rank, world, dist, seg_positions, victim_round=0, gather_bytes=True,
                      scan_ctx=None):
    """One chunk (window) of `total` bytes whose byte ranges live on the ranks: `my_bytes` = this rank's range
    [my_off, my_off + len) of segment_ranges() plus up to 48 bytes of halo from the next rank.

    Every segment's tags and candidate bitmap are computed by the rank that owns its bytes (ctx.window_scan) with the
    minimum_tag_mask the matcher reported last, and sent to rank 0 (point-to-point; RCCL send/recv between GPUs, gloo
    in the CPU test); rank 0 runs the exact matcher over them (ctx.rzip_chunk with a tag provider).  For the match
    extension rank 0 reads the chunk's bytes: on one node through peer mappings of the other GPUs' ranges; in this
    rehearsal the ranges are gathered to rank 0 first (gather_bytes).  scan_ctx: the context rank 0 scans its OWN
    range with (a second one: `ctx` is busy with the chunk); made on demand.  Returns (ChunkResult, s0, s1) on rank 0.
    """
    import torch
    ranges = segment_ranges(total, world, seg_positions)
    owner_of = lambda seg_start: max(r for r in range(world) if ranges[r][0] <= seg_start and (ranges[r][1] or r == 0))
    ctx.set_segment_positions(seg_positions)

    own_scan_ctx = None
    if rank == 0 and scan_ctx is None:
        from .binding import RzipContext
        scan_ctx = own_scan_ctx = RzipContext(level=ctx.level, max_chunk=0, device=ctx.device, lib=ctx.lib)
    sc = scan_ctx if rank == 0 else (scan_ctx or ctx)

    def scan(seg_start, seg_len, min_mask, p_done):
        return sc.window_scan(my_bytes, my_off, total, seg_start, seg_len, min_mask, p_done)

    if rank == 0:
        if world > 1 and gather_bytes:
            parts = [None] * world
            dist.gather_object(bytes(my_bytes[:ranges[0][1]]), parts, dst=0)
            chunk = b"".join(parts)
        else:
            chunk = bytes(my_bytes[:total])
        assert len(chunk) == total

        def provider(seg_index, seg_start, seg_len, min_mask, p_done):
            own = owner_of(seg_start)
            if world > 1:  # tell everybody which segment is next (segments an emitted match has covered are skipped)
                dist.broadcast(torch.tensor([seg_index, seg_start, seg_len, min_mask, p_done], dtype=torch.int64), src=0)
            if own == 0:
                return scan(seg_start, seg_len, min_mask, p_done)
            tags = torch.empty(seg_len, dtype=torch.int64)
            bitmap = torch.empty((seg_len + 63) // 64, dtype=torch.int64)
            dist.recv(tags, src=own)
            dist.recv(bitmap, src=own)
            return tags.numpy().tobytes(), bitmap.numpy().tobytes()

        ctx.set_tag_provider(provider)
        try:
            ctx.victim_round = victim_round
            out = ctx.rzip_chunk(chunk)
        finally:
            ctx.set_tag_provider(None)
            if own_scan_ctx is not None:
                own_scan_ctx.close()
            if world > 1:
                dist.broadcast(torch.tensor([-1, 0, 0, 0, 0], dtype=torch.int64), src=0)
        return out
    # ---- the other ranks: serve the segments of their range until rank 0 says it is done
    if gather_bytes:
        dist.gather_object(bytes(my_bytes[:ranges[rank][1]]), None, dst=0)
    hdr = torch.zeros(5, dtype=torch.int64)
    while True:
        dist.broadcast(hdr, src=0)
        seg_index, seg_start, seg_len, min_mask, p_done = (int(x) for x in hdr)
        if seg_index < 0:
            return None
        if owner_of(seg_start) != rank:
            continue
        tags, bitmap = scan(seg_start, seg_len, min_mask, p_done)
        dist.send(torch.frombuffer(bytearray(tags), dtype=torch.int64), dst=0)
        dist.send(torch.frombuffer(bytearray(bitmap), dtype=torch.int64), dst=0)
