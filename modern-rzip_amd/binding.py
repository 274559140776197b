"""ctypes binding of libmrzgpu.so (include/mrzgpu.h).  Plumbing only."""
import ctypes
import os

MEM_HOST = 0
MEM_DEVICE = 1

_HERE = os.path.dirname(os.path.abspath(__file__))


class MrzError(RuntimeError):
    pass


class Stats(ctypes.Structure):
    """struct rzip_state.stats (include/mrzip_private.h:407-415)."""
    _fields_ = [(n, ctypes.c_int64) for n in
                ("inserts", "literals", "literal_bytes", "matches", "match_bytes", "tag_hits", "tag_misses")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class ChunkResult(ctypes.Structure):
    _fields_ = [("s0_len", ctypes.c_int64), ("s1_len", ctypes.c_int64), ("crc32", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32), ("d_s0", ctypes.c_void_p), ("d_s1", ctypes.c_void_p),
                ("stats", Stats), ("min_mask", ctypes.c_int64), ("hash_count", ctypes.c_int64),
                ("n_events", ctypes.c_int64)]


class Timings(ctypes.Structure):
    _fields_ = [("tagscan_ms", ctypes.c_float), ("sequencer_ms", ctypes.c_float), ("encode_ms", ctypes.c_float),
                ("crc_ms", ctypes.c_float), ("total_ms", ctypes.c_float), ("n_segments", ctypes.c_int32),
                ("n_narrow", ctypes.c_int32), ("n_deep", ctypes.c_int32), ("reserved", ctypes.c_int32)]


# mrz_cand_provider_fn (include/mrzgpu.h)
CAND_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                           ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                           ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p)
TILE = 4096  # positions per front-end tile


class RsReport(ctypes.Structure):
    _fields_ = [("corrected", ctypes.c_int64), ("uncorrectable", ctypes.c_int64), ("checksum_ok", ctypes.c_int32),
                ("truncated", ctypes.c_int32)]


class Control(ctypes.Structure):
    """The rzip_control fields rzip_fd reads for `mrzip -n` (include/mrzgpu_host.h)."""
    _fields_ = [("rzip_compression_level", ctypes.c_int), ("compression_level", ctypes.c_int),
                ("window", ctypes.c_int64), ("unlimited", ctypes.c_int), ("ramsize", ctypes.c_int64),
                ("page_size", ctypes.c_int64), ("hash_code", ctypes.c_int), ("device", ctypes.c_int),
                ("lz4_test", ctypes.c_int), ("threshold", ctypes.c_int)]


class BlockInfo(ctypes.Structure):
    """mrz_block_info (include/mrzgpu_host.h)."""
    _fields_ = [("chunk_index", ctypes.c_int), ("stream", ctypes.c_int), ("chunk_bytes", ctypes.c_int),
                ("eof", ctypes.c_int), ("chunk_size", ctypes.c_int64), ("first_of_chunk", ctypes.c_int),
                ("lz4_verdict", ctypes.c_int), ("input_final", ctypes.c_int64)]


BLOCK_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(BlockInfo), ctypes.POINTER(ctypes.c_uint8),
                            ctypes.c_int64)


def lib_path():
    return os.environ.get("MRZGPU_LIB", os.path.join(_HERE, "libmrzgpu.so"))


_lib = None


def load_library(path=None):
    """Loads libmrzgpu.so; raises MrzError if it is missing (no fallback)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or lib_path()
    # One HIP runtime per process: PyTorch bundles its own libamdhip64, and whichever
    # copy is loaded first must serve both (two runtimes in one process cannot both
    # see the GPU).  Importing torch first makes libmrzgpu bind to torch's copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise MrzError(f"{p} not found: build it with __graft_entry__.build() "
                       f"(make -C modern-rzip_amd/csrc); there is no CPU fallback")
    lib = ctypes.CDLL(p)
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    lib.mrz_open.argtypes = [ctypes.POINTER(vp), ci, ci, i64]
    lib.mrz_close.argtypes = [vp]
    lib.mrz_close.restype = None
    lib.mrz_strerror.argtypes = [ci]
    lib.mrz_strerror.restype = ctypes.c_char_p
    lib.mrz_last_hip_error.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p)]
    if hasattr(lib, "mrz_rs_decode"):
        lib.mrz_rs_decode.argtypes = [vp, vp, i64, ci, vp, i64, ctypes.POINTER(i64), ctypes.POINTER(RsReport)]
    if hasattr(lib, "mrz_window_scan"):
        i64p = ctypes.POINTER(i64)
        lib.mrz_window_scan.argtypes = [vp, vp, i64, ci, i64, i64, i64, i64, i64, i64, i64, vp, vp, vp, ci, i64p, i64p]
        lib.mrz_set_cand_provider.argtypes = [vp, CAND_FN, vp]
        lib.mrz_set_segment_positions.argtypes = [vp, i64]
        lib.mrz_set_candidate_capacity.argtypes = [vp, i64]
        lib.mrz_set_xcd.argtypes = [vp, ci]
        lib.mrz_copy_to_device.argtypes = [vp, vp, vp, i64]
        lib.mrz_copy_device.argtypes = [vp, vp, vp, i64]
    if hasattr(lib, "mrz_window_map_create"):
        lib.mrz_window_granularity.argtypes = [ci]
        lib.mrz_window_granularity.restype = i64
        lib.mrz_window_part_create.argtypes = [ci, i64, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ci)]
        lib.mrz_window_part_destroy.argtypes = [vp]
        lib.mrz_window_part_destroy.restype = None
        lib.mrz_window_map_create.argtypes = [ci, ci, ctypes.POINTER(ci), ctypes.POINTER(i64), ctypes.POINTER(vp),
                                              ctypes.POINTER(vp)]
        lib.mrz_window_map_destroy.argtypes = [vp]
        lib.mrz_window_map_destroy.restype = None
    lib.mrz_stream.argtypes = [vp]
    lib.mrz_stream.restype = vp
    lib.mrz_synchronize.argtypes = [vp]
    lib.mrz_set_profiling.argtypes = [vp, ci]
    if hasattr(lib, "mrz_set_farm_helpers"):
        lib.mrz_set_farm_helpers.argtypes = [vp, ci]
    lib.mrz_get_timings.argtypes = [vp, ctypes.POINTER(Timings)]
    lib.mrz_rzip_chunk.argtypes = [vp, vp, i64, ci, ci, ctypes.POINTER(i64), ctypes.POINTER(ChunkResult)]
    lib.mrz_fetch_streams.argtypes = [vp, vp, vp]
    lib.mrz_chunk_bytes.argtypes = [i64]
    lib.mrz_table_slots.argtypes = [vp]
    lib.mrz_table_slots.restype = i64
    lib.mrz_fetch_table.argtypes = [vp, vp]
    lib.mrz_crc32.argtypes = [vp, vp, i64, ci, ctypes.POINTER(ctypes.c_uint32)]
    if hasattr(lib, "mrz_lz4_compresses_batch"):
        lib.mrz_lz4_compresses_batch.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(i64), ci, ci, ci,
                                                 ctypes.POINTER(ci)]
        lib.mrz_lz4_compresses.argtypes = [vp, vp, i64, ci, ci, ctypes.POINTER(ci)]
        lib.mrz_lz4_sizes.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ci), ci, ci, ctypes.POINTER(ci)]
    if hasattr(lib, "mrz_blake2b_batch"):
        lib.mrz_blake2b_init.argtypes = [vp, ctypes.POINTER(vp), ctypes.c_size_t]
        lib.mrz_blake2b_update.argtypes = [vp, vp, ctypes.c_size_t, ci]
        lib.mrz_blake2b_final.argtypes = [vp, vp, ctypes.c_size_t]
        lib.mrz_blake2b_batch.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(i64), ci, ci, ctypes.c_size_t, vp]
    if hasattr(lib, "mrz_rs_encode"):
        lib.mrz_rs_encoded_size.argtypes = [i64]
        lib.mrz_rs_encoded_size.restype = i64
        lib.mrz_rs_encode.argtypes = [vp, vp, i64, ci, vp, ci, i64]
    if hasattr(lib, "mrz_runzip_chunk"):
        u32p = ctypes.POINTER(ctypes.c_uint32)
        lib.mrz_runzip_chunk.argtypes = [vp, vp, i64, vp, i64, ci, ci, vp, ci, i64, ctypes.POINTER(i64), u32p, u32p]
    if hasattr(lib, "mrz_runzip_buffer"):
        lib.mrz_runzip_buffer.argtypes = [ci, vp, i64, ctypes.POINTER(vp), ctypes.POINTER(i64)]
    if hasattr(lib, "mrz_rzip_pipeline"):
        lib.mrz_rzip_pipeline.argtypes = [ctypes.POINTER(Control), vp, i64, BLOCK_FN, vp, ctypes.POINTER(Stats), vp]
    if hasattr(lib, "mrz_rzip_buffer"):
        lib.mrz_rzip_buffer.argtypes = [ctypes.POINTER(Control), vp, i64, ctypes.POINTER(vp), ctypes.POINTER(i64),
                                        ctypes.POINTER(Stats), vp]
        lib.mrz_rzip_fd.argtypes = [ctypes.POINTER(Control), ci, ci, ctypes.POINTER(Stats)]
        lib.mrz_rzip_stream.argtypes = [ctypes.POINTER(Control), ci, ci, ci, ctypes.POINTER(Stats)]
        lib.mrz_rzip_stream_buffer.argtypes = [ctypes.POINTER(Control), vp, i64, ci, ctypes.POINTER(vp),
                                               ctypes.POINTER(i64), ctypes.POINTER(Stats), ctypes.c_char_p]
        lib.mrz_free.argtypes = [vp]
        lib.mrz_free.restype = None
    if path is None:
        _lib = lib
    return lib


def _check(lib, rc, ctx=None):
    if rc == 0:
        return
    msg = lib.mrz_strerror(rc).decode()
    if ctx is not None:
        txt = ctypes.c_char_p()
        code = lib.mrz_last_hip_error(ctx, ctypes.byref(txt))
        if code:
            msg += f" (hip error {code}: {txt.value.decode() if txt.value else '?'})"
    raise MrzError(f"libmrzgpu: {msg} [{rc}]")


def chunk_bytes(chunk_size, lib=None):
    return (lib or load_library()).mrz_chunk_bytes(chunk_size)


def _as_ptr(buf):
    """(pointer, length, where, keepalive) for bytes-like / (device_ptr, n) inputs."""
    if isinstance(buf, tuple):  # (device pointer as int, nbytes)
        return ctypes.c_void_p(buf[0]), int(buf[1]), MEM_DEVICE, None
    if hasattr(buf, "data_ptr"):  # torch tensor (uint8, contiguous)
        n = buf.numel() * buf.element_size()
        where = MEM_DEVICE if buf.is_cuda else MEM_HOST
        return ctypes.c_void_p(buf.data_ptr()), n, where, buf
    b = bytes(buf) if not isinstance(buf, (bytes, bytearray)) else buf
    if isinstance(b, bytearray):
        arr = (ctypes.c_uint8 * len(b)).from_buffer(b)
        return ctypes.cast(arr, ctypes.c_void_p), len(b), MEM_HOST, arr
    return ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p), len(b), MEM_HOST, b


class WindowPart:
    """mrz_window_part: this rank's byte range of a window as a shareable allocation (ptr: where the owner writes it,
    fd: the descriptor the other ranks import; both live as long as the part)."""

    def __init__(self, nbytes, device=0, lib=None):
        self.lib = lib or load_library()
        self.handle, ptr, fd = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int(-1)
        _check(self.lib, self.lib.mrz_window_part_create(device, nbytes, ctypes.byref(self.handle), ctypes.byref(ptr),
                                                         ctypes.byref(fd)))
        self.ptr, self.fd, self.nbytes = ptr.value, fd.value, nbytes

    def close(self):
        if self.handle:
            self.lib.mrz_window_part_destroy(self.handle)
            self.handle = ctypes.c_void_p()


class WindowMap:
    """mrz_window_map: the parts of all ranks (descriptors in range order) mapped back to back; ptr + position is the
    window's byte."""

    def __init__(self, fds, sizes, device=0, lib=None):
        self.lib = lib or load_library()
        n = len(fds)
        self.handle, ptr = ctypes.c_void_p(), ctypes.c_void_p()
        _check(self.lib, self.lib.mrz_window_map_create(device, n, (ctypes.c_int * n)(*fds), (ctypes.c_int64 * n)(*sizes),
                                                        ctypes.byref(self.handle), ctypes.byref(ptr)))
        self.ptr, self.nbytes = ptr.value, sum(sizes)

    def close(self):
        if self.handle:
            self.lib.mrz_window_map_destroy(self.handle)
            self.handle = ctypes.c_void_p()


def window_granularity(device=0, lib=None):
    lib = lib or load_library()
    g = lib.mrz_window_granularity(device)
    if g <= 0:
        _check(lib, int(g))
    return int(g)


class RzipContext:
    """One mrz_ctx: the per-file half of rzip_fd (src/rzip.c:836-913)."""

    def __init__(self, level=7, max_chunk=0, device=0, lib=None):
        self.lib = lib or load_library()
        self.ctx = ctypes.c_void_p()
        _check(self.lib, self.lib.mrz_open(ctypes.byref(self.ctx), device, level, max_chunk))
        self.level = level
        self.device = device
        self.victim_round = 0  # static victim_round of insert_hash (src/rzip.c:259)

    def close(self):
        if self.ctx:
            self.lib.mrz_close(self.ctx)
            self.ctx = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return self.lib.mrz_stream(self.ctx)

    def set_farm_helpers(self, n):
        _check(self.lib, self.lib.mrz_set_farm_helpers(self.ctx, n), self.ctx)

    def set_segment_positions(self, positions):
        _check(self.lib, self.lib.mrz_set_segment_positions(self.ctx, positions), self.ctx)

    def set_candidate_capacity(self, entries):
        _check(self.lib, self.lib.mrz_set_candidate_capacity(self.ctx, entries), self.ctx)

    def set_xcd(self, xcd):
        _check(self.lib, self.lib.mrz_set_xcd(self.ctx, xcd), self.ctx)

    def copy_to(self, dst_ptr, src):
        """Copies bytes / a (cuda or cpu) uint8 tensor to device address dst_ptr on the ctx stream."""
        ptr, n, where, keep = _as_ptr(src)
        fn = self.lib.mrz_copy_device if where == MEM_DEVICE else self.lib.mrz_copy_to_device
        _check(self.lib, fn(self.ctx, ctypes.c_void_p(dst_ptr), ptr, n), self.ctx)

    def window_scan(self, range_bytes, range_start, chunk_n, seg_start, max_span, min_mask, p_done=0, cap=1 << 20,
                    out=None):
        """mrz_window_scan: one front-end pass over [seg_start, seg_start + max_span) from this rank's byte range.
        Returns (cand, tile_off, bitmap, scan_next, n_cand).  out=None: the three buffers come back as bytes objects
        trimmed to what the pass wrote; out=(cand, tile_off, bitmap) cuda tensors (uint8, at least cap * 16,
        (max_span / 4096 + 1) * 4 and max_span / 8 bytes): they are filled on the device and returned as they are."""
        ptr, n, where, keep = _as_ptr(range_bytes)
        tiles = max_span // TILE
        nx, nc = ctypes.c_int64(), ctypes.c_int64()
        if out is None:
            cand = ctypes.create_string_buffer(cap * 16)
            toff = ctypes.create_string_buffer((tiles + 1) * 4)
            bmp = ctypes.create_string_buffer(tiles * (TILE // 8))
            _check(self.lib, self.lib.mrz_window_scan(self.ctx, ptr, n, where, range_start, chunk_n, seg_start, max_span,
                                                      min_mask, p_done, cap, cand, toff, bmp, MEM_HOST, ctypes.byref(nx),
                                                      ctypes.byref(nc)), self.ctx)
            t = max(0, (nx.value - seg_start) // TILE)
            return cand.raw[:nc.value * 16], toff.raw[:(t + 1) * 4], bmp.raw[:t * (TILE // 8)], nx.value, nc.value
        cand, toff, bmp = out
        _check(self.lib, self.lib.mrz_window_scan(self.ctx, ptr, n, where, range_start, chunk_n, seg_start, max_span,
                                                  min_mask, p_done, cap, ctypes.c_void_p(cand.data_ptr()),
                                                  ctypes.c_void_p(toff.data_ptr()), ctypes.c_void_p(bmp.data_ptr()),
                                                  MEM_DEVICE, ctypes.byref(nx), ctypes.byref(nc)), self.ctx)
        return cand, toff, bmp, nx.value, nc.value

    def set_cand_provider(self, provider):
        """provider(seg_start, max_span, min_mask, p_done, cap) -> (cand, tile_off, bitmap, scan_next, n_cand) as
        window_scan returns them (bytes objects, or cuda tensors), called by rzip_chunk for every stretch instead of the
        local front end; None switches back."""
        if provider is None:
            self._cand_cb = None
            _check(self.lib, self.lib.mrz_set_cand_provider(self.ctx, ctypes.cast(None, CAND_FN), None), self.ctx)
            return

        def put(dst, src, nbytes):
            if nbytes <= 0:
                return 0
            if hasattr(src, "data_ptr"):
                if src.numel() * src.element_size() < nbytes:
                    return 1
                if src.is_cuda:
                    return self.lib.mrz_copy_device(self.ctx, dst, ctypes.c_void_p(src.data_ptr()), nbytes)
                return self.lib.mrz_copy_to_device(self.ctx, dst, ctypes.c_void_p(src.data_ptr()), nbytes)
            if len(src) < nbytes:
                return 1
            return self.lib.mrz_copy_to_device(self.ctx, dst, bytes(src), nbytes)

        def tramp(user, seg_start, max_span, min_mask, p_done, cap, d_cand, d_tile_off, d_bitmap, scan_next, n_cand, stream):
            try:
                cand, toff, bmp, nx, nc = provider(seg_start, max_span, min_mask, p_done, cap)
                t = (nx - seg_start) // TILE
                if nc < 0 or nc > cap or t < 0 or t * TILE > max_span:
                    return 1
                if put(d_cand, cand, nc * 16) or put(d_tile_off, toff, (t + 1) * 4) or put(d_bitmap, bmp, t * (TILE // 8)):
                    return 1
                scan_next[0] = nx
                n_cand[0] = nc
                return 0
            except Exception:  # noqa: BLE001 -- must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._cand_cb = CAND_FN(tramp)
        _check(self.lib, self.lib.mrz_set_cand_provider(self.ctx, self._cand_cb, None), self.ctx)

    def set_profiling(self, on=True):
        _check(self.lib, self.lib.mrz_set_profiling(self.ctx, 1 if on else 0), self.ctx)

    def timings(self):
        t = Timings()
        _check(self.lib, self.lib.mrz_get_timings(self.ctx, ctypes.byref(t)), self.ctx)
        return t

    def rzip_chunk(self, chunk, chunk_bytes_=None, fetch=True):
        """hash_search over one chunk.  Returns (ChunkResult, s0 bytes, s1 bytes)."""
        ptr, n, where, keep = _as_ptr(chunk)
        cb = chunk_bytes_ or self.lib.mrz_chunk_bytes(n)
        res = ChunkResult()
        vr = ctypes.c_int64(self.victim_round)
        _check(self.lib, self.lib.mrz_rzip_chunk(self.ctx, ptr, n, where, cb, ctypes.byref(vr), ctypes.byref(res)),
               self.ctx)
        self.victim_round = vr.value
        if not fetch:
            return res, None, None
        s0 = ctypes.create_string_buffer(max(1, res.s0_len))
        s1 = ctypes.create_string_buffer(max(1, res.s1_len))
        _check(self.lib, self.lib.mrz_fetch_streams(self.ctx, s0, s1), self.ctx)
        return res, s0.raw[:res.s0_len], s1.raw[:res.s1_len]

    def fetch_table(self):
        n = self.lib.mrz_table_slots(self.ctx)
        buf = ctypes.create_string_buffer(n * 16)
        _check(self.lib, self.lib.mrz_fetch_table(self.ctx, buf), self.ctx)
        return buf.raw

    def crc32(self, data):
        ptr, n, where, keep = _as_ptr(data)
        out = ctypes.c_uint32()
        _check(self.lib, self.lib.mrz_crc32(self.ctx, ptr, n, where, ctypes.byref(out)), self.ctx)
        return out.value

    # ---- LZ4 gate (src/stream.c:1685-1733) ----
    def lz4_compresses(self, blocks, threshold=100):
        single = isinstance(blocks, (bytes, bytearray)) or hasattr(blocks, "data_ptr")
        blks = [blocks] if single else list(blocks)
        prep = [_as_ptr(b) for b in blks]
        if not prep:
            return []
        where = prep[0][2]
        ptrs = (ctypes.c_void_p * len(prep))(*[p[0] for p in prep])
        lens = (ctypes.c_int64 * len(prep))(*[p[1] for p in prep])
        out = (ctypes.c_int * len(prep))()
        _check(self.lib, self.lib.mrz_lz4_compresses_batch(self.ctx, ptrs, lens, len(prep), where, threshold, out),
               self.ctx)
        return out[0] if single else list(out)

    def lz4_sizes(self, blocks):
        prep = [_as_ptr(b) for b in blocks]
        if not prep:
            return []
        ptrs = (ctypes.c_void_p * len(prep))(*[p[0] for p in prep])
        lens = (ctypes.c_int * len(prep))(*[p[1] for p in prep])
        out = (ctypes.c_int * len(prep))()
        _check(self.lib, self.lib.mrz_lz4_sizes(self.ctx, ptrs, lens, len(prep), prep[0][2], out), self.ctx)
        return list(out)

    # ---- rs-mrzip encoder (rs-mrzip/rs-mrzip.c:119-158) ----
    def rs_encode(self, data):
        ptr, n, where, keep = _as_ptr(data)
        total = self.lib.mrz_rs_encoded_size(n)
        out = ctypes.create_string_buffer(total)
        _check(self.lib, self.lib.mrz_rs_encode(self.ctx, ptr, n, where, out, MEM_HOST, total), self.ctx)
        return out.raw

    def rs_decode(self, data):
        """mrz_rs_decode -> (bytes, dict(corrected, uncorrectable, checksum_ok, truncated))."""
        ptr, n, where, keep = _as_ptr(data)
        cap = (n // 2084880) * 1823248
        out = ctypes.create_string_buffer(max(cap, 1))
        out_len = ctypes.c_int64()
        rep = RsReport()
        _check(self.lib, self.lib.mrz_rs_decode(self.ctx, ptr, n, where, out, cap, ctypes.byref(out_len), ctypes.byref(rep)),
               self.ctx)
        return out.raw[:out_len.value], dict(corrected=rep.corrected, uncorrectable=rep.uncorrectable,
                                             checksum_ok=bool(rep.checksum_ok), truncated=bool(rep.truncated))

    # ---- runzip (src/runzip.c:120-207,277-308) ----
    def runzip_chunk(self, s0, s1, chunk_bytes_, out_cap, out=None):
        """Decodes the two streams of one chunk.  Returns (bytes or None, out_len, crc_calc, crc_stored);
        with `out` = (device pointer, nbytes) or a cuda tensor the bytes stay on the device."""
        p0, n0, w0, k0 = _as_ptr(s0)
        p1, n1, w1, k1 = _as_ptr(s1)
        if w0 != w1:
            raise MrzError("runzip_chunk: both streams must live in the same memory space")
        got = ctypes.c_int64()
        cc, cs = ctypes.c_uint32(), ctypes.c_uint32()
        if out is None:
            buf = ctypes.create_string_buffer(max(1, out_cap))
            rc = self.lib.mrz_runzip_chunk(self.ctx, p0, n0, p1, n1, w0, chunk_bytes_, buf, MEM_HOST, out_cap,
                                           ctypes.byref(got), ctypes.byref(cc), ctypes.byref(cs))
            _check(self.lib, rc, self.ctx)
            return buf.raw[:got.value], got.value, cc.value, cs.value
        po, no, wo, ko = _as_ptr(out)
        rc = self.lib.mrz_runzip_chunk(self.ctx, p0, n0, p1, n1, w0, chunk_bytes_, po, wo, min(no, out_cap),
                                       ctypes.byref(got), ctypes.byref(cc), ctypes.byref(cs))
        _check(self.lib, rc, self.ctx)
        return None, got.value, cc.value, cs.value

    # ---- BLAKE2b (common/blake2b.h:47-49) ----
    def blake2b(self, data, outlen=64, pieces=None):
        st = ctypes.c_void_p()
        _check(self.lib, self.lib.mrz_blake2b_init(self.ctx, ctypes.byref(st), outlen), self.ctx)
        parts = pieces if pieces is not None else [data]
        for part in parts:
            ptr, n, where, keep = _as_ptr(part)
            _check(self.lib, self.lib.mrz_blake2b_update(st, ptr, n, where), self.ctx)
        out = ctypes.create_string_buffer(outlen)
        _check(self.lib, self.lib.mrz_blake2b_final(st, out, outlen), self.ctx)
        return out.raw

    def blake2b_batch(self, msgs, outlen=64):
        prep = [_as_ptr(m) for m in msgs]
        if not prep:
            return []
        ptrs = (ctypes.c_void_p * len(prep))(*[p[0] for p in prep])
        lens = (ctypes.c_int64 * len(prep))(*[p[1] for p in prep])
        out = ctypes.create_string_buffer(outlen * len(prep))
        _check(self.lib, self.lib.mrz_blake2b_batch(self.ctx, ptrs, lens, len(prep), prep[0][2], outlen, out),
               self.ctx)
        return [out.raw[i * outlen:(i + 1) * outlen] for i in range(len(prep))]


def rzip_buffer(data, level=7, window=0, unlimited=False, ramsize=60 << 30, device=0, lib=None):
    """`mrzip -n -L<level>` of an in-memory file through the C host driver
    (mrz_rzip_buffer: rzip_fd + the -n stream sink + write_magic).
    Returns (archive bytes, Stats, md5 bytes)."""
    lib = lib or load_library()
    ctl = Control(level, level, window, 1 if unlimited else 0, ramsize, 4096, 1, device)
    ptr, n, where, keep = _as_ptr(data)
    if where != MEM_HOST:
        raise MrzError("rzip_buffer takes host memory")
    out = ctypes.c_void_p()
    out_len = ctypes.c_int64()
    st = Stats()
    md5 = ctypes.create_string_buffer(16)
    _check(lib, lib.mrz_rzip_buffer(ctypes.byref(ctl), ptr, n, ctypes.byref(out), ctypes.byref(out_len),
                                    ctypes.byref(st), md5))
    try:
        return ctypes.string_at(out, out_len.value), st, md5.raw
    finally:
        lib.mrz_free(out)


def rzip_stream_buffer(data, to_stdout=False, level=7, window=0, ramsize=60 << 30, device=0, lib=None):
    """`mrzip -n` reading STDIN (mrz_rzip_stream_buffer: the STDIN form of rzip_fd's chunk loop, `data` standing
    for what read() delivers).  Returns (archive bytes, Stats, md5 bytes)."""
    lib = lib or load_library()
    ctl = Control(level, level, window, 0, ramsize, 4096, 1, device)
    ptr, n, where, keep = _as_ptr(data)
    if where != MEM_HOST:
        raise MrzError("rzip_stream_buffer takes host memory")
    out = ctypes.c_void_p()
    out_len = ctypes.c_int64()
    st = Stats()
    md5 = ctypes.create_string_buffer(16)
    _check(lib, lib.mrz_rzip_stream_buffer(ctypes.byref(ctl), ptr, n, 1 if to_stdout else 0, ctypes.byref(out),
                                           ctypes.byref(out_len), ctypes.byref(st), md5))
    try:
        return ctypes.string_at(out, out_len.value), st, md5.raw
    finally:
        lib.mrz_free(out)


def rzip_fd(fd_in, fd_out, level=7, window=0, unlimited=False, ramsize=60 << 30, device=0, lib=None):
    """mrz_rzip_fd on two open file descriptors (a regular file, or a pipe = the STDIN form).  Returns Stats."""
    lib = lib or load_library()
    ctl = Control(level, level, window, 1 if unlimited else 0, ramsize, 4096, 1, device)
    st = Stats()
    _check(lib, lib.mrz_rzip_fd(ctypes.byref(ctl), fd_in, fd_out, ctypes.byref(st)))
    return st


def runzip_buffer(mrz, device=0, lib=None):
    """`mrzip -d` of a -n archive held in memory (mrz_runzip_buffer).  Returns the original bytes."""
    lib = lib or load_library()
    ptr, n, where, keep = _as_ptr(mrz)
    if where != MEM_HOST:
        raise MrzError("runzip_buffer takes host memory")
    out = ctypes.c_void_p()
    out_len = ctypes.c_int64()
    _check(lib, lib.mrz_runzip_buffer(device, ptr, n, ctypes.byref(out), ctypes.byref(out_len)))
    try:
        return ctypes.string_at(out, out_len.value)
    finally:
        lib.mrz_free(out)


def rzip_pipeline(data, on_block, level=7, window=0, unlimited=False, ramsize=60 << 30, device=0, lib=None,
                  lz4_test=False, threshold=100):
    """mrz_rzip_pipeline: the GPU rzip stage over the chunks of `data`; `on_block(info_dict, payload_bytes)` is
    called once per stream block in the reference's flush order, while the rest of the chunk is still being
    sequenced (return None/0 to go on).  lz4_test: every block is put to the LZ4 gate first (info["lz4_verdict"]).
    Returns (Stats, md5)."""
    lib = lib or load_library()
    ctl = Control(level, level, window, 1 if unlimited else 0, ramsize, 4096, 1, device, 1 if lz4_test else 0,
                  threshold)
    ptr, n, where, keep = _as_ptr(data)
    if where != MEM_HOST:
        raise MrzError("rzip_pipeline takes host memory")

    def trampoline(user, info, payload, length):
        i = info.contents
        d = dict(chunk_index=i.chunk_index, stream=i.stream, chunk_bytes=i.chunk_bytes, eof=i.eof,
                 chunk_size=i.chunk_size, first_of_chunk=i.first_of_chunk, lz4_verdict=i.lz4_verdict,
                 input_final=i.input_final)
        return int(on_block(d, ctypes.string_at(payload, length)) or 0)

    fn = BLOCK_FN(trampoline)
    st = Stats()
    md5 = ctypes.create_string_buffer(16)
    _check(lib, lib.mrz_rzip_pipeline(ctypes.byref(ctl), ptr, n, fn, None, ctypes.byref(st), md5))
    return st, md5.raw
