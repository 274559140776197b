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

There is deliberately no all-to-all here: the path has no exchange step until a
single window is split across GPUs (the `-U` 256 GiB configuration, a later row).
The functions take the rank's compute callable so that the CPU test tier can drive
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
