// Test helper (not reference code): draws std::uniform_int_distribution<int32_t>(0, n-1) from a
// std::mt19937 with the C++ standard library of this toolchain, so that the restatement of
// libstdc++'s algorithm in the oracle and on the device can be compared with the library itself.
#include <cstdint>
#include <random>

extern "C" void uid_draws(uint32_t seed, const int32_t *ns, int32_t count, int32_t *out, uint32_t *next_raw) {
  std::mt19937 gen(seed);
  for (int32_t i = 0; i < count; ++i) {
    std::uniform_int_distribution<int32_t> dist(0, ns[i] - 1);
    out[i] = dist(gen);
  }
  *next_raw = (uint32_t)gen();  // position of the stream after the draws
}
