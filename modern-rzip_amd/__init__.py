"""modern-rzip_amd -- MI355X-native rzip stage (long-range dedup pre-processor).

The product is ``libmrzgpu.so`` (hand-written gfx950 HIP kernels behind the C
ABI of ``include/mrzgpu.h``).  This package is only the Python-side plumbing
used by tests and ``bench.py``: a ctypes binding whose names mirror the C ABI,
which in turn mirrors the reference's ``rzip_fd`` / ``lz4_compresses`` /
``blake2b_*`` interfaces.  There is no CPU fallback: using any operation
without a loadable ``libmrzgpu.so`` and a HIP device raises.
"""
import os as _os

# several RzipContexts of one process only overlap on the GPU if the HIP runtime may open enough hardware queues;
# it reads this when it initialises (see mrz_set_farm_helpers in include/mrzgpu.h)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .binding import (  # noqa: E402,F401
    MrzError,
    RzipContext,
    WindowPart,
    WindowMap,
    window_granularity,
    ChunkResult,
    Stats,
    Timings,
    Control,
    lib_path,
    load_library,
    chunk_bytes,
    rzip_buffer,
    rzip_stream_buffer,
    rzip_fd,
    runzip_buffer,
    rzip_pipeline,
    MEM_HOST,
    MEM_DEVICE,
)

__all__ = [
    "MrzError", "RzipContext", "WindowPart", "WindowMap", "window_granularity", "ChunkResult", "Stats", "Timings", "Control", "lib_path", "load_library",
    "chunk_bytes", "rzip_buffer", "rzip_stream_buffer", "rzip_fd", "runzip_buffer", "rzip_pipeline", "MEM_HOST", "MEM_DEVICE",
]
