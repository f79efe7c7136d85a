// Library-level entry points: version / device probe used by the Python loader.
#include "common.h"

namespace mtvaf {
static const uint64_t* g_rng_epoch = nullptr;
const uint64_t* rng_epoch_ptr() { return g_rng_epoch; }
__global__ void rng_epoch_advance_kernel(uint64_t* p) { *p += 1; }
}  // namespace mtvaf

extern "C" {

// Device-side dropout epoch for captured launches (see common.h).  `dev_word` is a device uint64 owned by the caller;
// NULL switches back to host-fed masks.  Process-global (one Python thread per process, one process per GPU).
int mtvaf_rng_set_epoch_ptr(const uint64_t* dev_word) {
  mtvaf::g_rng_epoch = dev_word;
  return MTVAF_OK;
}

// *dev_word += 1 on `stream`: the first node of a captured training step.
int mtvaf_rng_epoch_advance(uint64_t* dev_word, hipStream_t stream) {
  if (!dev_word) return MTVAF_ERR_ARG;
  hipLaunchKernelGGL(mtvaf::rng_epoch_advance_kernel, dim3(1), dim3(1), 0, stream, dev_word);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_version(void) { return 100; }  // 0.1.0

// Number of compute units of the current device (0 if no HIP device is usable).
int mtvaf_device_cus(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
  return p.multiProcessorCount;
}

}  // extern "C"
