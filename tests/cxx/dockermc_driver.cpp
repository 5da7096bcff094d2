// dockermc_driver.cpp -- TEST: the web app's move chooser (corintho_ai/docker/choose_move.pyx:88-133
// search + :199-221 result) in C++ against `class DockerMC` of corintho_ai_amd/cpp/dockermc.h, with the
// stand-in network of tests/harness.py hash_net.  Prints one line per position:
//   move done drawn nodes eval_bits legal0 legal1 legal2   (or "pre draw" / "pre win")
//   dockermc_driver <max_searches> <searches_per_eval> < positions   (lines: seed to_play p0..p5 b0..b63)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../corintho_ai_amd/cpp/dockermc.cpp"

namespace {
constexpr int GS = 70, NM = 96;
uint64_t mix(uint64_t h) {
  h = (h ^ (h >> 33)) * 0xFF51AFD7ED558CCDull;
  h = (h ^ (h >> 33)) * 0xC4CEB9FE1A85EC53ull;
  return h ^ (h >> 33);
}
void hash_net(const float *states, int n, float *evals, float *probs) {
  for (int r = 0; r < n; ++r) {
    const float *s = states + (size_t)r * GS;
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int j = 0; j < GS; ++j) h = mix(h ^ ((uint64_t)std::nearbyint((double)s[j] * 4.0) + (uint64_t)j * 0x100000001B3ull + 1ull));
    evals[r] = (float)((double)(h >> 40) / (double)(1 << 24) * 2.0 - 1.0);
    for (int m = 0; m < NM; ++m)
      probs[(size_t)r * NM + m] = (float)(((double)(mix(h + (uint64_t)(m + 1) * 0x9E3779B97F4A7C15ull) >> 40) + 1.0) / (double)((1 << 24) + 1));
  }
}
}  // namespace

int main(int argc, char **argv) {
  if (argc != 3) return 2;
  const int S = atoi(argv[1]), spe = atoi(argv[2]);
  int seed, tp, pieces[6], board[64];
  try {
    while (scanf("%d %d", &seed, &tp) == 2) {
      for (int i = 0; i < 6; ++i) if (scanf("%d", &pieces[i]) != 1) return 3;
      for (int i = 0; i < 64; ++i) if (scanf("%d", &board[i]) != 1) return 3;
      DockerMC mc(seed, S, spe, 1.0f, 0.25f, board, tp, pieces);
      if (mc.done()) {  // get_pre_result, choose_move.pyx:75-86
        printf("pre %s\n", mc.drawn() ? "draw" : "win");
        continue;
      }
      std::vector<float> ev(spe), pr((size_t)spe * NM), gs((size_t)spe * GS);
      for (;;) {  // choose_move.pyx:110-133 without the time limit
        if (mc.doIteration(ev.data(), pr.data())) break;
        const int n = mc.num_requests();
        if (n == 0) break;
        mc.writeRequests(gs.data());
        hash_net(gs.data(), n, ev.data(), pr.data());
      }
      const int move = mc.chooseMove();
      int32_t legal[96];
      mc.getLegalMoves(legal);
      uint32_t m[3] = {0, 0, 0};
      for (int i = 0; i < 96; ++i) if (legal[i]) m[i >> 5] |= 1u << (i & 31);
      const float e = mc.eval();
      uint32_t eb;
      memcpy(&eb, &e, 4);
      printf("%d %d %d %d %u %u %u %u\n", move, mc.done() ? 1 : 0, mc.drawn() ? 1 : 0, mc.num_nodes(), eb, m[0], m[1], m[2]);
    }
  } catch (const std::exception &e) {
    fprintf(stderr, "dockermc_driver: %s\n", e.what());
    return 1;
  }
  return 0;
}
