// mg_mem.h — where the observation buffers live (host code; HIP virtual memory management).
//
// Measured on MI355X (profiles/r02/README.md section 3, profiles/r03): the rate at which HBM absorbs the obs
// raster's write pattern — thousands of waves, each streaming its own env — depends on the ALLOCATION it
// writes into (5.3 vs 6.6-6.8 TB/s), while a dense fill of the same buffers is flat.  A buffer that comes out
// of hipMalloc is backed by whatever physical blocks the VRAM manager had at hand; what the raster is
// sensitive to is how that backing is cut up.  So the engine builds its observation buffers itself: one
// virtual range (hipMemAddressReserve), backed by physical handles of a chosen size (hipMemCreate, 2 MiB
// granules), mapped back to back (hipMemMap) — a CONSTRUCTION with a known layout instead of a search among
// allocations of unknown layout.  `chunk_bytes` = 0 falls back to a plain hipMalloc.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace mg {

struct ObsBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;        // what the caller asked for
    size_t mapped = 0;       // what is reserved and mapped (a multiple of the chunk size)
    size_t chunk = 0;        // bytes per physical handle; 0: plain hipMalloc
    int device = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};

inline size_t obs_granularity(int device, bool recommended) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &prop, recommended ? hipMemAllocationGranularityRecommended
                                                              : hipMemAllocationGranularityMinimum) != hipSuccess)
        return 0;
    return g;
}

inline void obs_free(ObsBuffer* b) {
    if (!b) return;
    if (b->chunk == 0) {
        if (b->ptr) (void)hipFree(b->ptr);
    } else {
        if (b->ptr) {   // mapping by mapping, as they were made
            for (size_t i = 0; i < b->handles.size(); i++) (void)hipMemUnmap(static_cast<char*>(b->ptr) + i * b->chunk, b->chunk);
            (void)hipMemAddressFree(b->ptr, b->mapped);
        }
        for (auto h : b->handles) (void)hipMemRelease(h);
    }
    delete b;
}

// chunk_bytes: 0 = hipMalloc; > 0 = one physical handle per chunk (rounded up to the 2 MiB granule);
// < 0 = ONE handle for the whole buffer.  Returns nullptr on failure (nothing is left allocated).
inline ObsBuffer* obs_alloc(size_t bytes, int device, long long chunk_bytes) {
    if (bytes == 0) return nullptr;
    ObsBuffer* b = new ObsBuffer();
    b->bytes = bytes;
    b->device = device;
    if (chunk_bytes == 0) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        const hipError_t e = hipMalloc(&b->ptr, bytes);
        if (cur != device) (void)hipSetDevice(cur);
        if (e != hipSuccess) { (void)hipGetLastError(); delete b; return nullptr; }
        b->mapped = bytes;
        return b;
    }
    size_t gran = obs_granularity(device, true);
    if (gran == 0) gran = 2u << 20;
    if (gran < (2u << 20)) gran = 2u << 20;
    const size_t total = (bytes + gran - 1) / gran * gran;
    size_t chunk = chunk_bytes < 0 ? total : ((size_t)chunk_bytes + gran - 1) / gran * gran;
    if (chunk > total) chunk = total;
    b->chunk = chunk;
    b->mapped = (total + chunk - 1) / chunk * chunk;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    if (hipMemAddressReserve(&b->ptr, b->mapped, chunk < (1u << 30) ? chunk : (1u << 30), nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        b->ptr = nullptr;
        delete b;
        return nullptr;
    }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    bool ok = true;
    size_t n_mapped = 0;
    for (size_t off = 0; off < b->mapped && ok; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        char* va = static_cast<char*>(b->ptr) + off;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { ok = false; break; }
        b->handles.push_back(h);
        if (hipMemMap(va, chunk, 0, h, 0) != hipSuccess) { ok = false; break; }
        n_mapped++;
        if (hipMemSetAccess(va, chunk, &acc, 1) != hipSuccess) { ok = false; break; }
    }
    if (!ok) {
        (void)hipGetLastError();
        for (size_t i = 0; i < n_mapped; i++) (void)hipMemUnmap(static_cast<char*>(b->ptr) + i * chunk, chunk);
        (void)hipMemAddressFree(b->ptr, b->mapped);
        for (auto h : b->handles) (void)hipMemRelease(h);
        delete b;
        return nullptr;
    }
    return b;
}

}  // namespace mg
