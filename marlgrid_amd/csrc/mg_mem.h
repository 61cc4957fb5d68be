// mg_mem.h — MEASUREMENT BUILD ONLY (libmarlgrid_hip_ab.so; tools/placement_vmm*.py, placement_va.py,
// vmm_reuse_check.py): observation buffers taken apart with HIP virtual memory management.
//
// Why: the rate at which HBM absorbs the obs raster's write pattern — thousands of waves, each streaming its own
// env — depends on the BUFFER it writes into (0.160 vs 0.205 ms for the same 925 MB), reproducibly per buffer,
// while a dense fill of the same buffers is flat (profiles/r02/README.md section 3).  Round 3 built buffers from
// 2 MiB physical handles (hipMemCreate) behind a reserved virtual range (hipMemAddressReserve / hipMemMap) to
// find out what the class belongs to — profiles/r03/README.md section 2:
//   * a buffer keeps its class when its handles are permuted, and when half of them are traded with a buffer of
//     the other class: not the physical memory;
//   * the same physical memory behind 40 fresh virtual ranges (obs_rebase) gives 40 times one and the same —
//     slow — time: remapping does not draw a new class the way a new allocation does;
//   * and the decisive one for the product: on ROCm 7.2 a virtual range that was freed and is handed out again
//     ALIASES STALE TRANSLATIONS — writes through the reused range land in the previous owner's memory
//     (tools/vmm_reuse_check.py: two buffers corrupt each other; an env built after another one was destroyed
//     returned wrong observations).  A construction that is not safe to free cannot ship.
// So the product allocates observation buffers with hipMalloc (mg_obs_alloc) and chooses among candidates by
// timing the raster into each (MultiGridEnv._place_obs_buffers); this file stays as the record of the experiment.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>
#include <vector>

namespace mg {

struct ObsBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;        // what the caller asked for
    size_t mapped = 0;       // what is reserved and mapped (a multiple of the chunk size)
    size_t chunk = 0;        // bytes per physical handle; 0: plain hipMalloc
    int device = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<void*> ranges;   // every virtual range reserved for this buffer; `ptr` is the one the handles are mapped in
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
        if (b->ptr)   // mapping by mapping, as they were made
            for (size_t i = 0; i < b->handles.size(); i++) (void)hipMemUnmap(static_cast<char*>(b->ptr) + i * b->chunk, b->chunk);
        for (void* r : b->ranges) (void)hipMemAddressFree(r, b->mapped);
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
    b->ranges.push_back(b->ptr);
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

inline bool obs_map_all(ObsBuffer* b, void* base) {
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = b->device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t i = 0; i < b->handles.size(); i++) {
        char* va = static_cast<char*>(base) + i * b->chunk;
        if (hipMemMap(va, b->chunk, 0, b->handles[i], 0) != hipSuccess) return false;
        if (hipMemSetAccess(va, b->chunk, &acc, 1) != hipSuccess) return false;
    }
    return true;
}
inline void obs_unmap_all(ObsBuffer* b) {
    for (size_t i = 0; i < b->handles.size(); i++) (void)hipMemUnmap(static_cast<char*>(b->ptr) + i * b->chunk, b->chunk);
}

// The same physical memory behind another virtual range.  The caller must have drained every stream that uses
// the buffer.  obs_rebase reserves a NEW range (the ones tried before stay reserved, so every call explores a
// different one) and moves the mapping there; obs_select moves it back to the i-th range tried (0 = the one the
// buffer was built with); obs_trim gives every range but the current one back.  Contents survive (same memory).
inline bool obs_move_to(ObsBuffer* b, void* range) {
    if (range == b->ptr) return true;
    obs_unmap_all(b);
    if (!obs_map_all(b, range)) return false;
    b->ptr = range;
    return true;
}
inline void* obs_rebase(ObsBuffer* b) {
    if (!b || b->chunk == 0) return nullptr;
    void* range = nullptr;
    if (hipMemAddressReserve(&range, b->mapped, b->chunk < (1u << 30) ? b->chunk : (1u << 30), nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    b->ranges.push_back(range);
    return obs_move_to(b, range) ? range : nullptr;
}
inline bool obs_select(ObsBuffer* b, int i) {
    if (!b || b->chunk == 0 || i < 0 || (size_t)i >= b->ranges.size()) return false;
    return obs_move_to(b, b->ranges[(size_t)i]);
}
inline void obs_trim(ObsBuffer* b) {
    if (!b || b->chunk == 0) return;
    for (void* r : b->ranges)
        if (r != b->ptr) (void)hipMemAddressFree(r, b->mapped);
    b->ranges.assign(1, b->ptr);
}

#if defined(MG_AB_VARIANTS)
// Measurement build (tools/placement_vmm2.py): the same physical handles behind the same virtual range in
// another ORDER, and handles traded between two buffers — is a buffer's class a property of its handles
// (additive), or of how the raster's streams fall onto them?  (Answer: neither moves it; profiles/r03.)
inline bool obs_permute(ObsBuffer* b, const int32_t* order) {
    if (!b || b->chunk == 0) return false;
    (void)hipDeviceSynchronize();
    obs_unmap_all(b);
    std::vector<hipMemGenericAllocationHandle_t> h(b->handles.size());
    for (size_t i = 0; i < h.size(); i++) h[i] = b->handles[(size_t)order[i]];
    b->handles = h;
    return obs_map_all(b, b->ptr);
}
inline bool obs_exchange(ObsBuffer* a, ObsBuffer* b, const int32_t* slots, int n) {
    if (!a || !b || a->chunk == 0 || a->chunk != b->chunk) return false;
    (void)hipDeviceSynchronize();
    obs_unmap_all(a);
    obs_unmap_all(b);
    for (int i = 0; i < n; i++) {
        const size_t s = (size_t)slots[i];
        if (s < a->handles.size() && s < b->handles.size()) std::swap(a->handles[s], b->handles[s]);
    }
    return obs_map_all(a, a->ptr) && obs_map_all(b, b->ptr);
}
#endif

}  // namespace mg
