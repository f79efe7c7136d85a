// Library-level entry points: version / device probe used by the Python loader.
#include "common.h"

extern "C" {

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
