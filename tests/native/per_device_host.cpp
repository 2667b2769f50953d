// Host check of me_per_device.h (tests/test_capi_cpu.py): the cache is keyed by the device id.
#include "../../metropolisengine_amd/csrc/me_per_device.h"

extern "C" int me_test_per_device() {
  static me::PerDevice<int> cache;
  int calls = 0;
  auto make = [&] { return 100 + ++calls; };
  const int a0 = cache.get(0, make), b0 = cache.get(1, make), a1 = cache.get(0, make), b1 = cache.get(1, make);
  const int out = cache.get(me::kMaxDevices + 3, make);      // out of range: never cached
  // two distinct devices -> two make() calls with distinct results, repeated lookups hit the cache
  return (a0 == 101 && b0 == 102 && a1 == 101 && b1 == 102 && out == 103 && calls == 3) ? 0 : 1;
}
