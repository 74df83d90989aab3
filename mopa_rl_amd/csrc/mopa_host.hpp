// mopa_host.hpp -- host-side plumbing shared by the translation units of libmopa_hip.so (mopa_hip.hip: scene, K1-K3, K5,
// path post-processing; mopa_envdyn.hip: K4 env.step, K6 servo dynamics, K7 contacts): error state, HIP call check,
// device guard, grow-only device buffers.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>

#include "../../include/mopa_hip.h"

constexpr int kMaxLdsBytes = 160 * 1024;   // LDS per CU on gfx950
int mopa_fail(int code, const std::string &msg);      // sets the thread's last error (mopa_last_error) and returns `code`
static inline int fail(int code, const std::string &msg) { return mopa_fail(code, msg); }
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(MOPA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));          \
    } while (0)

// Every entry point runs on its object's device and leaves the caller's current device as it found it (a torch process
// that touches several GPUs keeps its own notion of "current").
class DeviceGuard {
    int prev_ = -1;
    bool ok_ = false, switched_ = false;
public:
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev_) != hipSuccess) return;
        if (prev_ != device) {
            if (hipSetDevice(device) != hipSuccess) return;
            switched_ = true;
        }
        ok_ = true;
    }
    ~DeviceGuard() {
        if (switched_) (void)hipSetDevice(prev_);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
    bool ok() const { return ok_; }
};
#define ON_DEVICE(dev)                                                        \
    DeviceGuard _guard(dev);                                                  \
    if (!_guard.ok()) return fail(MOPA_ERR_HIP, "cannot switch to the object's HIP device")

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;     // bytes
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// bounding radius of a primitive geom about its origin
static inline double rbound_of(int type, const double *sz) {
    switch (type) {
        case 2: return sz[0];                                                            // sphere
        case 3: return sz[0] + sz[1];                                                    // capsule
        case 5: return sqrt(fma(sz[1], sz[1], sz[0] * sz[0]));                           // cylinder
        case 6: return sqrt(fma(sz[2], sz[2], fma(sz[1], sz[1], sz[0] * sz[0])));        // box
        default: return 0.0;
    }
}
