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
  halo), compacts the candidates (positions whose tag passes the matcher's mask) into 16-byte
  records and ships those to the rank that runs the exact matcher, which takes them instead of
  scanning (mrz_set_cand_provider); CRC-32 is checksummed per range and combined over GF(2).
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


def rzip_file_chunks_on_ctxs(data, max_chunk, ctxs, predict=None):
    """The chunks of ONE file through several ctxs of ONE process at the same time (one host thread and one XCD per ctx:
    mrz_set_xcd), speculating on victim_round like rzip_file_chunk_chain_speculative: phase 1 runs every chunk with a
    predicted value, `len(ctxs)` at a time; phase 2 walks the chunks in order with the true value and runs a chunk again
    only where the prediction was wrong (the integer only moves on chain-limit evictions, src/rzip.c:284-289).
    Returns ([(chunk_size, s0, s1), ...] in chunk order, number of re-runs): what the stream sink frames into the archive,
    bit-identical to the one-ctx run."""
    import threading
    chunks = split_chunks(len(data), max_chunk)
    predict = predict or (lambda k: 0)
    spec = [None] * len(chunks)

    def worker(j):
        for k in range(j, len(chunks), len(ctxs)):
            off, size = chunks[k]
            c = ctxs[j]
            c.victim_round = predict(k)
            res, s0, s1 = c.rzip_chunk(data[off:off + size])
            spec[k] = (predict(k), size, s0, s1, c.victim_round)
    threads = [threading.Thread(target=worker, args=(j,)) for j in range(len(ctxs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    out, reruns, vr = [], 0, 0
    for k, (off, size) in enumerate(chunks):
        vin, size, s0, s1, vout = spec[k]
        if vin != vr:  # mispredicted: once more with the true value
            c = ctxs[0]
            c.victim_round = vr
            res, s0, s1 = c.rzip_chunk(data[off:off + size])
            vout = c.victim_round
            reruns += 1
        out.append((size, s0, s1))
        vr = vout
    return out, reruns


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


# ---- ONE window over several ranks: range-sharded front end, matcher on rank 0 (SURVEY.md 8e, configs[3]) --------
TILE = 4096  # positions per front-end tile: ranges are made of whole tiles


def window_ranges(total, world, align=TILE):
    """Byte range of every rank: contiguous, whole multiples of `align` bytes (the last one ends with the window):
    [(offset, size)] * world (sizes may be 0)."""
    per = -(-total // world)
    per = -(-per // align) * align
    out = []
    for r in range(world):
        off = min(r * per, total)
        out.append((off, min(per, total - off)))
    return out


def exchange_fds(my_fd, rank, world, dist):
    """Every rank contributes one file descriptor (its window part, mrz_window_part_create) and gets back the list of
    all ranks' descriptors, valid in ITS process.  Descriptors cannot travel through torch.distributed: rank 0 listens on
    a Unix socket (its path goes round by broadcast_object_list), the others connect and pass theirs as SCM_RIGHTS
    ancillary data, rank 0 answers each with the full set."""
    import os
    import socket
    import tempfile
    if world == 1:
        return [my_fd]
    path = [None]
    srv = None
    if rank == 0:
        path[0] = os.path.join(tempfile.mkdtemp(prefix="mrzwin"), "fds.sock")
        srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        srv.bind(path[0])
        srv.listen(world)
    dist.broadcast_object_list(path, src=0)
    if rank == 0:
        conns, fds = {}, {0: my_fd}
        try:
            for _ in range(world - 1):
                c, _ = srv.accept()
                msg, got, _, _ = socket.recv_fds(c, 16, 1)
                r = int(msg.decode())
                conns[r], fds[r] = c, got[0]
            order = [fds[r] for r in range(world)]
            for r, c in conns.items():
                socket.send_fds(c, [b"ok"], order)
                c.recv(2)  # the peer has its copies
        finally:
            for c in conns.values():
                c.close()
            srv.close()
            os.unlink(path[0])
            os.rmdir(os.path.dirname(path[0]))
        return order
    c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    c.connect(path[0])
    try:
        socket.send_fds(c, [str(rank).encode()], [my_fd])
        _, got, _, _ = socket.recv_fds(c, 16, world)
        c.send(b"ok")
    finally:
        c.close()
    return list(got)


def window_map(data_of_range, total, rank, world, dist, ctx, device=0):
    """The window's bytes over the ranks (mrz_window_part_create / mrz_window_map_create, include/mrzgpu.h): every rank
    puts its range -- data_of_range(off, size) -> bytes or a uint8 tensor -- into a shareable allocation in its own HBM,
    the descriptors go round (exchange_fds), and every rank maps all parts back to back: map.ptr + position is the
    window's byte whichever GPU holds it.  Ranges are whole allocation granules (window_ranges(total, world, granule)).
    Returns (WindowMap, WindowPart, ranges); close both when done (the map first)."""
    import os
    from .binding import WindowMap, WindowPart, window_granularity
    g = window_granularity(device, ctx.lib)
    ranges = window_ranges(total, world, align=g)
    if any(n == 0 for _, n in ranges):
        raise ValueError("a window of %d bytes does not give each of %d ranks a whole %d-byte granule" % (total, world, g))
    off, size = ranges[rank]
    mapped = -(-size // g) * g  # (the last range is padded to a whole granule)
    part = WindowPart(mapped, device=device, lib=ctx.lib)
    ctx.copy_to(part.ptr, data_of_range(off, size))
    sizes = [-(-n // g) * g for _, n in ranges]
    fds = exchange_fds(part.fd, rank, world, dist)
    try:
        wmap = WindowMap(fds, sizes, device=device, lib=ctx.lib)
    finally:
        for r, fd in enumerate(fds):
            if r != rank:
                os.close(fd)  # (imported: the map holds the allocations now)
    dist.barrier()
    return wmap, part, ranges


def _tensor_of(buf, dtype, device):
    import torch
    t = torch.frombuffer(bytearray(buf), dtype=dtype) if len(buf) else torch.empty(0, dtype=dtype)
    return t.to(device) if device is not None else t


def rzip_chunk_window(ctx, my_bytes, my_off, total, rank, world, dist, victim_round=0, window=None, scan_ctx=None,
                      device=None, cap=None, ranges=None):
    """One chunk (window) of `total` bytes whose byte ranges live on the ranks: `my_bytes` = this rank's range
    [my_off, my_off + len) of window_ranges() plus up to 48 bytes of halo from the next rank (bytes, or a cuda uint8
    tensor that stays on the device).

    The exact matcher runs on rank 0 (ctx.rzip_chunk with a candidate provider).  For every stretch of the window it is
    about to sequence it asks the rank that owns the stretch's first position for ONE front-end pass
    (ctx.window_scan: the candidates -- positions whose tag passes the minimum_tag_mask the matcher has last reported --
    compacted into 16-byte {position, tag} records, plus the tile offsets and the pass bitmap that index the list) and
    receives the result point-to-point: header [scan_next, n_cand], the records, the offsets, the bitmap.  `device`:
    a torch device -- every message is then a DEVICE tensor (RCCL send/recv moves it GPU to GPU, the scan writes its
    outputs straight into the send buffers, the matcher takes them by a device-to-device copy); None: CPU tensors (gloo).
    A request with seg_start < 0 ends the service loops.

    `window`: what rank 0 reads the window's bytes from for the match extension -- bytes / a cuda tensor / a
    (device pointer, nbytes) pair covering all `total` bytes (e.g. a peer mapping of the other ranks' ranges,
    window_map below).  None: the ranges are gathered to rank 0 first (a rehearsal: caps the window at one GPU's HBM).
    scan_ctx: the context rank 0 scans its OWN range with (a second one: `ctx` is busy with the chunk); made on demand.
    Returns (ChunkResult, s0, s1) on rank 0, None elsewhere."""
    import torch
    ranges = ranges or window_ranges(total, world)
    starts = [r[0] for r in ranges]

    def owner_of(pos):
        own = 0
        for r in range(world):
            if ranges[r][1] > 0 and starts[r] <= pos:
                own = r
        return own

    cap = cap or (1 << 20)
    own_scan_ctx = None
    if rank == 0 and scan_ctx is None:
        from .binding import RzipContext
        scan_ctx = own_scan_ctx = RzipContext(level=ctx.level, max_chunk=0, device=ctx.device, lib=ctx.lib)
    sc = scan_ctx if rank == 0 else (scan_ctx or ctx)
    max_span = max(TILE, -(-max(r[1] for r in ranges) // TILE) * TILE)
    tiles = max_span // TILE
    bufs = None
    if device is not None:  # the scan's outputs = the send buffers, on the device
        bufs = (torch.empty(cap * 16, dtype=torch.uint8, device=device),
                torch.empty((tiles + 1) * 4, dtype=torch.uint8, device=device),
                torch.empty(tiles * (TILE // 8), dtype=torch.uint8, device=device))

    def scan(seg_start, span, min_mask, p_done, capx):
        return sc.window_scan(my_bytes, my_off, total, seg_start, span, min_mask, p_done, cap=capx, out=bufs)

    def clip(seg_start, span):  # the part of a stretch that lies in the range of the rank owning its first position
        own = owner_of(seg_start)
        r_end = ranges[own][0] + ranges[own][1]
        if own == world - 1 or r_end >= total:
            r_end = -(-total // TILE) * TILE
        return own, max(TILE, min(span, r_end - seg_start, max_span))

    hdr_dev = device if device is not None else "cpu"
    if rank == 0:
        if window is None:
            mine = bytes(my_bytes[:ranges[0][1]].cpu().numpy().tobytes()) if hasattr(my_bytes, "data_ptr") else bytes(my_bytes[:ranges[0][1]])
            if world > 1:
                parts = [None] * world
                dist.gather_object(mine, parts, dst=0)
                window = b"".join(parts)
            else:
                window = mine
        served = {"n": 0, "remote": 0, "bytes": 0}

        def provider(seg_start, span, min_mask, p_done, capx):
            capx = min(capx, cap)
            own, span = clip(seg_start, span)
            served["n"] += 1
            if own == 0:
                return scan(seg_start, span, min_mask, p_done, capx)
            served["remote"] += 1
            dist.send(torch.tensor([seg_start, span, min_mask, p_done, capx], dtype=torch.int64, device=hdr_dev), dst=own)
            rep = torch.zeros(2, dtype=torch.int64, device=hdr_dev)
            dist.recv(rep, src=own)
            nx, nc = (int(x) for x in rep.tolist())
            t = (nx - seg_start) // TILE
            cand = torch.empty(nc * 16, dtype=torch.uint8, device=hdr_dev)
            toff = torch.empty((t + 1) * 4, dtype=torch.uint8, device=hdr_dev)
            bmp = torch.empty(t * (TILE // 8), dtype=torch.uint8, device=hdr_dev)
            if nc:
                dist.recv(cand, src=own)
            dist.recv(toff, src=own)
            if t:
                dist.recv(bmp, src=own)
            served["bytes"] += cand.numel() + toff.numel() + bmp.numel()
            return cand, toff, bmp, nx, nc

        ctx.set_cand_provider(provider)
        try:
            ctx.victim_round = victim_round
            out = ctx.rzip_chunk(window)
        finally:
            ctx.set_cand_provider(None)
            if own_scan_ctx is not None:
                own_scan_ctx.close()
            for r in range(1, world):
                dist.send(torch.tensor([-1, 0, 0, 0, 0], dtype=torch.int64, device=hdr_dev), dst=r)
        ctx.window_served = served
        return out
    # ---- the other ranks: serve the stretches of their range until rank 0 says it is done
    if window is None:
        mine = bytes(my_bytes[:ranges[rank][1]].cpu().numpy().tobytes()) if hasattr(my_bytes, "data_ptr") else bytes(my_bytes[:ranges[rank][1]])
        dist.gather_object(mine, None, dst=0)
    hdr = torch.zeros(5, dtype=torch.int64, device=hdr_dev)
    while True:
        dist.recv(hdr, src=0)
        seg_start, span, min_mask, p_done, capx = (int(x) for x in hdr.tolist())
        if seg_start < 0:
            return None
        cand, toff, bmp, nx, nc = scan(seg_start, span, min_mask, p_done, capx)
        t = (nx - seg_start) // TILE
        dist.send(torch.tensor([nx, nc], dtype=torch.int64, device=hdr_dev), dst=0)
        if device is None:
            cand, toff, bmp = _tensor_of(cand, torch.uint8, None), _tensor_of(toff, torch.uint8, None), _tensor_of(bmp, torch.uint8, None)
        if nc:
            dist.send(cand[:nc * 16].contiguous(), dst=0)
        dist.send(toff[:(t + 1) * 4].contiguous(), dst=0)
        if t:
            dist.send(bmp[:t * (TILE // 8)].contiguous(), dst=0)
