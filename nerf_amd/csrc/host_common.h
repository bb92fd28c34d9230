// Host-side launch helpers shared by the kernel translation units.  Everything here is keyed by the CURRENT DEVICE and guarded by a
// mutex: hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies per device, so one process driving several GPUs (or several host
// threads) must opt in once per (device, kernel function), and the CU count that sizes the persistent grids is a per-device fact.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>

namespace nerf_host {

inline std::mutex& table_mutex() { static std::mutex m; return m; }

inline int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return dev;
}

// Dynamic LDS above 64 KiB is an opt-in per kernel function AND device; done once for each pair (and again if a larger size is asked).
inline int allow_dynamic_lds(const void* fn, size_t lds) {
    struct Entry { int dev; const void* fn; size_t lds; };
    static std::vector<Entry> done;
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(table_mutex());
    for (auto& e : done)
        if (e.dev == dev && e.fn == fn) {
            if (e.lds >= lds) return 0;
            hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (err != hipSuccess) return (int)err;
            e.lds = lds;
            return 0;
        }
    hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return (int)err;
    done.push_back({dev, fn, lds});
    return 0;
}

// compute units of the current device (256 on MI355X); persistent kernels launch one workgroup per CU
inline int cu_count() {
    static int n_cu[64] = {0};
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(table_mutex());
    int& slot = n_cu[dev & 63];
    if (!slot) {
        hipDeviceProp_t p;
        slot = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
    return slot;
}

}  // namespace nerf_host
