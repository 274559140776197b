"""Seeded synthetic workloads of SURVEY.md section 8d (no downloads: enwik8 is
not available offline).  Host generators return bytes; *_device build the large
BASELINE configurations directly in HBM with torch (plumbing only)."""
import numpy as np


def zipf_text(nbytes, seed=7, vocab=5000):
    """S1-style text: Zipf(1/rank) words from a seeded vocabulary, space
    separated, a newline every 20000 words."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(2, 11, size=vocab)
    words = [bytes(rng.integers(97, 123, size=int(k), dtype=np.uint8)) for k in lens]
    w = 1.0 / np.arange(1, vocab + 1)
    w /= w.sum()
    out = bytearray()
    while len(out) < nbytes:
        idx = rng.choice(vocab, size=20000, p=w)
        for i in idx:
            out += words[i]
            out += b" "
        out[-1:] = b"\n"
    return bytes(out[:nbytes])


def rep64k(nperiods, seed=1234, period=65536, first_period=0):
    """S2 "rep64k": a `period`-byte text block repeated; in period i the byte at
    (37*i) % period is replaced by i & 0xff."""
    base = np.frombuffer(zipf_text(period, seed=seed), dtype=np.uint8)
    arr = np.tile(base, nperiods).reshape(nperiods, period).copy()
    i = np.arange(first_period, first_period + nperiods)
    arr[np.arange(nperiods), (37 * i) % period] = (i & 0xFF).astype(np.uint8)
    return arr.tobytes()


def rep64k_device(nperiods, device, seed=1234, period=65536):
    """The same stream built in HBM: returns a uint8 torch tensor of nperiods*period bytes."""
    import torch
    base = torch.frombuffer(bytearray(zipf_text(period, seed=seed)), dtype=torch.uint8).to(device)
    out = base.repeat(nperiods).view(nperiods, period)
    i = torch.arange(nperiods, device=device, dtype=torch.int64)
    out[i, (37 * i) % period] = (i & 0xFF).to(torch.uint8)
    return out.view(-1)


def noise(nbytes, seed=99):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes()


def tar_like(nbytes, seed=5):
    """S3-style mix: text members, noise members and exact duplicates of earlier
    members, 512-byte aligned."""
    rng = np.random.default_rng(seed)
    out = bytearray()
    members = []
    while len(out) < nbytes:
        kind = rng.random()
        size = int(2 ** rng.uniform(10, 17))
        if kind < 0.6 or not members:
            mem = zipf_text(size, seed=int(rng.integers(1, 1 << 30)))
        elif kind < 0.85:
            mem = noise(size, seed=int(rng.integers(1, 1 << 30)))
        else:
            mem = members[int(rng.integers(0, len(members)))]
        members.append(mem)
        out += mem
        out += bytes((-len(out)) % 512)
    return bytes(out[:nbytes])


def tar_like_fast(nbytes, seed=5, pool_bytes=48 << 20):
    """The S3 mix at sizes of a GiB and more without generating a GiB of Zipf text word by word: text members are
    slices of one text pool at random offsets (so parts of members repeat each other at distances of up to the whole
    stream), noise members come straight from the generator, 15 % are exact duplicates of earlier members;
    512-byte aligned like tar_like."""
    rng = np.random.default_rng(seed)
    pool = np.frombuffer(zipf_text(pool_bytes, seed=seed + 1), dtype=np.uint8)
    out = np.empty(nbytes + (8 << 20), dtype=np.uint8)
    at, members = 0, []
    while at < nbytes:
        kind = rng.random()
        size = int(2 ** rng.uniform(10, 22))
        if kind < 0.6 or not members:
            off = int(rng.integers(0, len(pool) - size))
            out[at:at + size] = pool[off:off + size]
        elif kind < 0.85:
            out[at:at + size] = rng.integers(0, 256, size=size, dtype=np.uint8)
        else:
            m0, msz = members[int(rng.integers(0, len(members)))]
            size = msz
            out[at:at + size] = out[m0:m0 + size]
        members.append((at, size))
        at += size
        pad = (-at) % 512
        out[at:at + pad] = 0
        at += pad
    return out[:nbytes].tobytes()


def stride_stream(nseg, seg_bytes, copy_bytes=None, seed=99):
    """S4 "stride" shape (BASELINE configs[3], scaled): `nseg` segments of noise (seed + segment index); in every
    4th segment the first `copy_bytes` (default a quarter) repeat the segment k segments earlier, k cycling
    through 1, 3, 7 -- long matches at a distance of whole segments."""
    copy_bytes = seg_bytes // 4 if copy_bytes is None else copy_bytes
    segs = [np.frombuffer(noise(seg_bytes, seed=seed + i), dtype=np.uint8).copy() for i in range(nseg)]
    ks = (1, 3, 7)
    j = 0
    for i in range(3, nseg, 4):
        k = ks[j % 3]
        j += 1
        if i - k >= 0:
            segs[i][:copy_bytes] = segs[i - k][:copy_bytes]
    return b"".join(s.tobytes() for s in segs)


def tar_like_device(nbytes, device, seed=5, pool_bytes=48 << 20, max_member_log2=22):
    """The S3 mix of tar_like_fast built directly in HBM (BASELINE configs[2] at its full 64 GiB does not fit the host's
    generators): text members are slices of one Zipf text pool at random offsets, noise members come from torch's
    generator on the device, 15 % are exact duplicates of earlier members; 512-byte aligned.  Deterministic for a given
    (nbytes, seed, torch version), NOT byte-identical to tar_like_fast (different noise generator): the oracle is run on
    a prefix copied back from the device.  Returns a uint8 tensor of nbytes bytes."""
    import torch
    rng = np.random.default_rng(seed)
    pool = torch.frombuffer(bytearray(zipf_text(pool_bytes, seed=seed + 1)), dtype=torch.uint8).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty(nbytes + (8 << 20), dtype=torch.uint8, device=device)
    at, members = 0, []
    while at < nbytes:
        kind = rng.random()
        size = int(2 ** rng.uniform(10, max_member_log2))
        if kind < 0.6 or not members:
            off = int(rng.integers(0, pool_bytes - size))
            out[at:at + size] = pool[off:off + size]
        elif kind < 0.85:
            out[at:at + size].random_(0, 256, generator=g)
        else:
            m0, msz = members[int(rng.integers(0, len(members)))]
            size = msz
            out[at:at + size] = out[m0:m0 + size].clone() if m0 + size > at else out[m0:m0 + size]
        members.append((at, size))
        at += size
        pad = (-at) % 512
        out[at:at + pad] = 0
        at += pad
    return out[:nbytes]


def noise_device(nbytes, device, seed=99):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty(nbytes, dtype=torch.uint8, device=device)
    for a in range(0, nbytes, 1 << 30):  # (slice-wise: random_ materialises temporaries of the slice's size)
        out[a:a + (1 << 30)].random_(0, 256, generator=g)
    return out


def stride_stream_device(nseg, seg_bytes, device, copy_bytes=None, seed=99):
    """stride_stream's shape built in HBM (BASELINE configs[3]: 1 GiB segments, every 4th repeats the first quarter of the
    segment 1, 3 or 7 segments earlier)."""
    copy_bytes = seg_bytes // 4 if copy_bytes is None else copy_bytes
    out = noise_device(nseg * seg_bytes, device, seed=seed)
    ks, j = (1, 3, 7), 0
    for i in range(3, nseg, 4):
        k = ks[j % 3]
        j += 1
        if i - k >= 0:
            out[i * seg_bytes:i * seg_bytes + copy_bytes] = out[(i - k) * seg_bytes:(i - k) * seg_bytes + copy_bytes]
    return out
