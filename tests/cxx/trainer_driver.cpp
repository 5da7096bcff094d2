// trainer_driver.cpp -- TEST: the reference's play loop (corintho_ai/python/main.pyx:123-219,
// play_games + get_samples) written in C++ against `class Trainer` of corintho_ai_amd/cpp/trainer.h
// -- the source-level boundary the reference's Cython module consumes (main.pyx:17-38).  The
// network is a deterministic stand-in coded here (the integer hash of tests/harness.py hash_net),
// so the run needs nothing but the engine library.  Every request batch and the three sample
// arrays are dumped for the Python test, which compares them with the CPU oracle bit for bit.
//
//   trainer_driver <out_prefix> <num_games> <seed> <max_searches> <searches_per_eval> <testing>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../corintho_ai_amd/cpp/trainer.cpp"

namespace {
constexpr int GS = 70, NM = 96, NSYM = 8;

uint64_t mix(uint64_t h) {
  h = (h ^ (h >> 33)) * 0xFF51AFD7ED558CCDull;
  h = (h ^ (h >> 33)) * 0xC4CEB9FE1A85EC53ull;
  return h ^ (h >> 33);
}

// tests/harness.py hash_net: value in (-1, 1) and 96 positive priors from the 70-float row only
void hash_net(const float *states, int n, uint64_t salt, float *evals, float *probs) {
  for (int r = 0; r < n; ++r) {
    const float *s = states + (size_t)r * GS;
    uint64_t h = 0x9E3779B97F4A7C15ull + salt;
    for (int j = 0; j < GS; ++j) {
      uint64_t q = (uint64_t)std::nearbyint((double)s[j] * 4.0);  // entries are k/4
      h = mix(h ^ (q + (uint64_t)j * 0x100000001B3ull + 1ull));
    }
    evals[r] = (float)((double)(h >> 40) / (double)(1 << 24) * 2.0 - 1.0);
    for (int m = 0; m < NM; ++m) {
      uint64_t hm = mix(h + (uint64_t)(m + 1) * 0x9E3779B97F4A7C15ull);
      probs[(size_t)r * NM + m] = (float)(((double)(hm >> 40) + 1.0) / (double)((1 << 24) + 1));
    }
  }
}

void dump(FILE *f, const void *p, size_t bytes) {
  if (bytes && fwrite(p, 1, bytes, f) != bytes) throw std::runtime_error("short write");
}
}  // namespace

int main(int argc, char **argv) {
  if (argc != 7) {
    fprintf(stderr, "usage: %s out_prefix num_games seed max_searches searches_per_eval testing\n", argv[0]);
    return 2;
  }
  const std::string prefix = argv[1];
  const int G = atoi(argv[2]), seed = atoi(argv[3]), S = atoi(argv[4]), spe = atoi(argv[5]);
  const bool testing = atoi(argv[6]) != 0;
  try {
    // main.pyx:299-310: Trainer(num_games, log_folder, seed, max_searches, searches_per_eval, c_puct,
    // epsilon, num_logged, num_threads, testing)
    // two logged games (trainer.cpp:243-250): <prefix>.logs/game_0.txt, game_1.txt
    Trainer trainer(G, prefix + ".logs", seed, S, spe, 1.0f, 0.25f, 2, 1, testing);
    const size_t cap = (size_t)G * spe;
    std::vector<float> evals(cap, 0.0f), probs(cap * NM, 0.0f), game_states(cap * GS, 0.0f);  // main.pyx:132-134
    int to_play = testing ? 0 : -1;
    FILE *req = fopen((prefix + ".requests.bin").c_str(), "wb");
    if (!req) throw std::runtime_error("cannot open output");
    int32_t iterations = 0;
    for (;;) {  // main.pyx:142-168
      const bool res = trainer.doIteration(evals.data(), probs.data(), to_play);
      ++iterations;
      if (res) break;
      const int32_t n = trainer.num_requests(to_play);
      if (n == 0) {
        if (to_play != -1) {
          to_play = 1 - to_play;
          continue;
        }
        throw std::runtime_error("No requests during training");
      }
      trainer.writeRequests(game_states.data(), to_play);
      // get_predictions (main.pyx:70-83): new model when to_play == 0, else the best model
      hash_net(game_states.data(), n, testing ? (to_play == 0 ? 1 : 2) : 0, evals.data(), probs.data());
      const int32_t hdr[2] = {to_play, n};
      dump(req, hdr, sizeof hdr);
      dump(req, game_states.data(), (size_t)n * GS * sizeof(float));
    }
    fclose(req);
    // get_samples, main.pyx:189-198
    const int32_t ns = trainer.num_samples();
    std::vector<float> sgs((size_t)ns * NSYM * GS), sev((size_t)ns * NSYM), spr((size_t)ns * NSYM * NM);
    if (ns > 0) trainer.writeSamples(sgs.data(), sev.data(), spr.data());
    FILE *smp = fopen((prefix + ".samples.bin").c_str(), "wb");
    if (!smp) throw std::runtime_error("cannot open output");
    const float score = trainer.score(), mate = trainer.avg_mate_length();
    const int32_t hdr[2] = {ns, iterations};
    dump(smp, hdr, sizeof hdr);
    dump(smp, &score, 4);
    dump(smp, &mate, 4);
    dump(smp, sgs.data(), sgs.size() * 4);
    dump(smp, sev.data(), sev.size() * 4);
    dump(smp, spr.data(), spr.size() * 4);
    fclose(smp);
    trainer.writeScores(prefix + ".scores.txt");
  } catch (const std::exception &e) {
    fprintf(stderr, "trainer_driver: %s\n", e.what());
    return 1;
  }
  return 0;
}
