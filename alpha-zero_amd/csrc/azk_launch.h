// azk_launch.h - host-side launch helpers shared by the translation units of libazk.so.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (function, device): keep, per pair, the largest size set so far and set it again
// when a launch wants more or runs on another device (a process-wide "set once" flag served only the first device and the first size).
inline hipError_t azk_set_max_lds(const void *fn, int bytes) {
    struct Rec { const void *fn; int dev, bytes; };
    static Rec recs[1024];
    static int n = 0;
    static std::mutex mu;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    int at = -1;
    for (int i = 0; i < n; i++)
        if (recs[i].fn == fn && recs[i].dev == dev) { at = i; break; }
    if (at >= 0 && recs[at].bytes >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    if (at < 0 && n < 1024) at = n++;
    if (at >= 0) recs[at] = Rec{fn, dev, bytes};
    return hipSuccess;
}
