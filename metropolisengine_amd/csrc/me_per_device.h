// Values that must be resolved once PER DEVICE of the process (a __device__ function pointer read from a code object,
// a function attribute that raises the dynamic-LDS limit, a CU count): an engine's device is per engine
// (me_config.device_id), so a process-wide `static const` would hand the first device's answer to every other one.
// No HIP dependency: the device id is passed in, so the host-only tests can exercise it (tests/native/per_device_host.cpp).
#pragma once

#include <mutex>

namespace me {

constexpr int kMaxDevices = 64;

template <typename T>
class PerDevice {
 public:
  // make() runs once for each distinct device id (ids outside [0, kMaxDevices) are never cached)
  template <class Make>
  T get(int device, Make &&make) {
    if (device < 0 || device >= kMaxDevices) return make();
    std::lock_guard<std::mutex> guard(mutex_);
    if (!have_[device]) {
      value_[device] = make();
      have_[device] = true;
    }
    return value_[device];
  }

 private:
  std::mutex mutex_;
  T value_[kMaxDevices] = {};
  bool have_[kMaxDevices] = {};
};

}  // namespace me
