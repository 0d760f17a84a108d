// common.h - error plumbing and small device helpers shared by every translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/diffab_hip.h"

namespace diffab {

void set_error(const char* fmt, ...);

#define DIFFAB_REQUIRE(cond, code, ...)  \
  do {                                   \
    if (!(cond)) {                       \
      ::diffab::set_error(__VA_ARGS__);  \
      return (code);                     \
    }                                    \
  } while (0)

#define DIFFAB_HIP_CHECK(expr)                                                              \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      ::diffab::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return DIFFAB_ERR_HIP;                                                                \
    }                                                                                       \
  } while (0)

#define DIFFAB_LAUNCH_CHECK()                                                               \
  do {                                                                                      \
    hipError_t e_ = hipGetLastError();                                                      \
    if (e_ != hipSuccess) {                                                                 \
      ::diffab::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
      return DIFFAB_ERR_HIP;                                                                \
    }                                                                                       \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// (The library never calls getenv; the kernel variants that were measured and not adopted, and the environment switches used to A/B
// them, live under experiments/ as patches on top of these sources.)

// Cross-stream ordering guard, first statement of every stream-taking entry point (include/diffab_hip.h, "Streams").  OFF by default since
// round 6 (the miscompute it worked around was the v_pk_*_f32 op_sel:[0,1] hazard, removed from every kernel: profiles/r06_lanes_48_63.md);
// then the constructor is one relaxed atomic load.  When switched on (diffab_set_stream_guard(1)): work the library enqueued on a DIFFERENT
// stream before is ordered in front of this call (hipEventRecord on the previous stream + hipStreamWaitEvent on this one).
class StreamOrder {
 public:
  explicit StreamOrder(void* stream);
  ~StreamOrder();
  StreamOrder(const StreamOrder&) = delete;
  StreamOrder& operator=(const StreamOrder&) = delete;

 private:
  int dev_;
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's workspace (256-B aligned carves).
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t n) {
    off = align_up(off, 256);
    T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return r;
  }
  size_t bytes() const { return align_up(off, 256); }
};

}  // namespace diffab
