// TEST INFRASTRUCTURE ONLY: the vendor's Philox4x32-10 engine (rocRAND, the generator behind hipRAND's
// HIPRAND_RNG_PSEUDO_PHILOX4_32_10) evaluated on the HOST -- rocrand_philox4x32_10.h is __host__ __device__ -- so that
// tests/test_oracle_philox.py can pin the build's stream contract (oracle/philox.py, csrc/me_device.h) to it:
//     block `b` of step `s` of chain `c` under seed `k`  ==  engine(seed = k, subsequence = s_lo | (w3 << 32),
//                                                                  offset = 4 c).next4(),   w3 = ((s >> 32) & 0xffff) << 16 | b
// (the engine's counter is (offset / 4, subsequence) and its key is the seed).
#include <rocrand/rocrand_philox4x32_10.h>

extern "C" void me_rocrand_philox_block(unsigned long long seed, unsigned long long chain, unsigned long long step,
                                        unsigned int block, unsigned int *out4) {
  const unsigned long long w3 = (((step >> 32) & 0xffffull) << 16) | (unsigned long long)block;
  const unsigned long long subsequence = (step & 0xffffffffull) | (w3 << 32);
  rocrand_device::philox4x32_10_engine engine(seed, subsequence, 4ull * chain);
  const uint4 r = engine.next4();
  out4[0] = r.x;
  out4[1] = r.y;
  out4[2] = r.z;
  out4[3] = r.w;
}
