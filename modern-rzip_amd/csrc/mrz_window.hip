// mrz_window.hip -- ONE window whose byte ranges live on several GPUs (BASELINE configs[3]: `-U`, 256 GiB over 8 GPUs).
//
// The reference maps the whole file and lets single_match_len run anywhere in it (src/rzip.c:372-397, -U at :881-882).
// Here every rank keeps its byte range of the window in its own HBM, as a SHAREABLE physical allocation (HIP virtual
// memory management); the allocations are exported as POSIX file descriptors, handed between the processes (SCM_RIGHTS,
// the host program's business), and every rank that needs the window maps them back to back into ONE virtual address
// range: positions are addresses again, the kernels stay what they are, and a load that falls into another rank's range
// is served over xGMI (peer access) -- the compare farm of the matcher's rank, the 30-byte halo of a range's tag scan,
// the literal gather and the CRC read their bytes where they lie; nothing is gathered.
//
// No kernel here; host-side HIP runtime calls only.
#include <hip/hip_runtime.h>
#include <string.h>
#include <unistd.h>

#include "mrz_ctx.h"

struct mrz_window_part {
    int device;
    int64_t bytes;  // physical size (a multiple of the granularity)
    hipMemGenericAllocationHandle_t handle;
    void *va;       // the owner's own mapping
    int fd;         // the exported descriptor (closed with the part)
};

struct mrz_window_map {
    int device;
    int n_parts;
    int64_t total;  // bytes reserved
    void *va;
    hipMemGenericAllocationHandle_t *handles;  // imported (released with the map)
    int64_t *sizes;
};

static hipMemAllocationProp mrz_window_prop(int device) {
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    return prop;
}

extern "C" int64_t mrz_window_granularity(int device) {
    if (hipSetDevice(device) != hipSuccess) return MRZ_E_NODEVICE;
    hipMemAllocationProp prop = mrz_window_prop(device);
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || g == 0) return MRZ_E_HIP;
    return (int64_t)g;
}

static int mrz_window_access(void *va, size_t bytes, int device) {
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    return hipMemSetAccess(va, bytes, &acc, 1) == hipSuccess ? MRZ_OK : MRZ_E_HIP;
}

extern "C" void mrz_window_part_destroy(mrz_window_part *p) {
    if (!p) return;
    hipSetDevice(p->device);
    if (p->va) {
        hipMemUnmap(p->va, (size_t)p->bytes);
        hipMemAddressFree(p->va, (size_t)p->bytes);
    }
    if (p->fd >= 0) close(p->fd);
    hipMemRelease(p->handle);
    free(p);
}

extern "C" int mrz_window_part_create(int device, int64_t bytes, mrz_window_part **out, void **dptr, int *fd) {
    if (!out || !dptr || !fd || bytes <= 0) return MRZ_E_ARG;
    const int64_t g = mrz_window_granularity(device);
    if (g <= 0) return (int)g;
    if (bytes % g) return MRZ_E_ARG;  // ranges are made of whole granules (mrz_window_granularity)
    mrz_window_part *p = (mrz_window_part *)calloc(1, sizeof(*p));
    if (!p) return MRZ_E_NOMEM;
    p->device = device;
    p->bytes = bytes;
    p->fd = -1;
    hipMemAllocationProp prop = mrz_window_prop(device);
    if (hipMemCreate(&p->handle, (size_t)bytes, &prop, 0) != hipSuccess) {
        free(p);
        return MRZ_E_NOMEM;
    }
    int rc = MRZ_OK;
    if (hipMemAddressReserve(&p->va, (size_t)bytes, (size_t)g, nullptr, 0) != hipSuccess) {
        p->va = nullptr;
        rc = MRZ_E_NOMEM;
    }
    if (!rc && hipMemMap(p->va, (size_t)bytes, 0, p->handle, 0) != hipSuccess) {
        hipMemAddressFree(p->va, (size_t)bytes);
        p->va = nullptr;
        rc = MRZ_E_HIP;
    }
    if (!rc) rc = mrz_window_access(p->va, (size_t)bytes, device);
    if (!rc && hipMemExportToShareableHandle(&p->fd, p->handle, hipMemHandleTypePosixFileDescriptor, 0) != hipSuccess) {
        p->fd = -1;
        rc = MRZ_E_HIP;
    }
    if (rc) {
        mrz_window_part_destroy(p);
        return rc;
    }
    *out = p;
    *dptr = p->va;
    *fd = p->fd;
    return MRZ_OK;
}

extern "C" void mrz_window_map_destroy(mrz_window_map *m) {
    if (!m) return;
    hipSetDevice(m->device);
    int64_t off = 0;
    for (int i = 0; i < m->n_parts; i++) {
        if (m->handles[i]) {
            hipMemUnmap((char *)m->va + off, (size_t)m->sizes[i]);
            hipMemRelease(m->handles[i]);
        }
        off += m->sizes[i];
    }
    if (m->va) hipMemAddressFree(m->va, (size_t)m->total);
    free(m->handles);
    free(m->sizes);
    free(m);
}

extern "C" int mrz_window_map_create(int device, int n_parts, const int *fds, const int64_t *sizes, mrz_window_map **out,
                                     void **dptr) {
    if (!out || !dptr || !fds || !sizes || n_parts < 1 || n_parts > 64) return MRZ_E_ARG;
    const int64_t g = mrz_window_granularity(device);
    if (g <= 0) return (int)g;
    int64_t total = 0;
    for (int i = 0; i < n_parts; i++) {
        if (sizes[i] <= 0 || sizes[i] % g || fds[i] < 0) return MRZ_E_ARG;
        total += sizes[i];
    }
    mrz_window_map *m = (mrz_window_map *)calloc(1, sizeof(*m));
    if (!m) return MRZ_E_NOMEM;
    m->device = device;
    m->n_parts = n_parts;
    m->total = total;
    m->handles = (hipMemGenericAllocationHandle_t *)calloc((size_t)n_parts, sizeof(*m->handles));
    m->sizes = (int64_t *)calloc((size_t)n_parts, sizeof(int64_t));
    if (!m->handles || !m->sizes) {
        free(m->handles);
        free(m->sizes);
        free(m);
        return MRZ_E_NOMEM;
    }
    for (int i = 0; i < n_parts; i++) m->sizes[i] = sizes[i];
    int rc = MRZ_OK;
    if (hipMemAddressReserve(&m->va, (size_t)total, (size_t)g, nullptr, 0) != hipSuccess) {
        m->va = nullptr;
        rc = MRZ_E_NOMEM;
    }
    int64_t off = 0;
    for (int i = 0; i < n_parts && !rc; i++) {
        hipMemGenericAllocationHandle_t h;
        // (the runtime takes the descriptor through a pointer; CUDA's convention is the value cast to a pointer: a
        // runtime that wants the value sees an address here, which is no open descriptor, and fails cleanly -- then the
        // value form is tried; the other order would make a pointer-taking runtime dereference a small integer)
        int fd_i = fds[i];
        if (hipMemImportFromShareableHandle(&h, (void *)&fd_i, hipMemHandleTypePosixFileDescriptor) != hipSuccess &&
            hipMemImportFromShareableHandle(&h, (void *)(uintptr_t)fds[i], hipMemHandleTypePosixFileDescriptor) != hipSuccess) {
            (void)hipGetLastError();
            rc = MRZ_E_HIP;
            break;
        }
        if (hipMemMap((char *)m->va + off, (size_t)sizes[i], 0, h, 0) != hipSuccess) {
            hipMemRelease(h);
            rc = MRZ_E_HIP;
            break;
        }
        m->handles[i] = h;
        off += sizes[i];
    }
    if (!rc) rc = mrz_window_access(m->va, (size_t)total, device);
    if (rc) {
        mrz_window_map_destroy(m);
        return rc;
    }
    *out = m;
    *dptr = m->va;
    return MRZ_OK;
}
