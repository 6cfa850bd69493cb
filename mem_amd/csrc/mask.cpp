// Host-side mask generators with CPython `random` semantics (reference:
// mem/masking_generator.py:18-116).  Bit-exact parity needs the MT19937 stream, CPython's
// random()/uniform()/randint()/sample() draw recipes, Python's round-half-even and glibc's
// exp/log/sqrt -- all sequential host arithmetic, so this lives in host C++ inside the same
// library (the rejection loop consumes a data-dependent number of draws: masks of one stream
// cannot be produced in parallel; parallelism is across streams).
#include <cmath>
#include <cstring>
#include <vector>
#include "common.h"

namespace {

constexpr int N = 624, M = 397;

struct MT {
  uint32_t* s;  // 624 words + index at s[624]
  explicit MT(uint32_t* st) : s(st) {}
  uint32_t u32() {
    uint32_t& idx = s[N];
    if (idx >= (uint32_t)N) {
      int k;
      auto mix = [](uint32_t u, uint32_t v) {
        uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
        return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
      };
      for (k = 0; k < N - M; ++k) s[k] = s[k + M] ^ mix(s[k], s[k + 1]);
      for (; k < N - 1; ++k) s[k] = s[k + (M - N)] ^ mix(s[k], s[k + 1]);
      s[N - 1] = s[M - 1] ^ mix(s[N - 1], s[0]);
      idx = 0;
    }
    uint32_t y = s[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  // random.random(): 53-bit float from two draws
  double random() {
    uint32_t a = u32() >> 5, b = u32() >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
  }
  double uniform(double a, double b) { return a + (b - a) * random(); }
  // Random._randbelow_with_getrandbits (n >= 1, n < 2^32)
  uint32_t randbelow(uint32_t n) {
    int k = 32 - __builtin_clz(n);   // n.bit_length()
    uint32_t r = u32() >> (32 - k);
    while (r >= n) r = u32() >> (32 - k);
    return r;
  }
  long randint(long a, long b) { return a + (long)randbelow((uint32_t)(b - a + 1)); }
};

void init_genrand(uint32_t* s, uint32_t seed) {
  s[0] = seed;
  for (int i = 1; i < N; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
  s[N] = N;
}

inline long py_round(double x) { return (long)std::nearbyint(x); }  // round-half-even (default FE mode)

}  // namespace

extern "C" int memhip_mt_seed(uint32_t* st, const uint32_t* key, int key_len) {
  MEMHIP_REQUIRE(st && key && key_len > 0, "mt_seed: bad arguments");
  // init_by_array, as random.seed(int) does with the 32-bit limbs of abs(seed)
  init_genrand(st, 19650218u);
  int i = 1, j = 0;
  int k = N > key_len ? N : key_len;
  for (; k; --k) {
    st[i] = (st[i] ^ ((st[i - 1] ^ (st[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
    ++i; ++j;
    if (i >= N) { st[0] = st[N - 1]; i = 1; }
    if (j >= key_len) j = 0;
  }
  for (k = N - 1; k; --k) {
    st[i] = (st[i] ^ ((st[i - 1] ^ (st[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
    ++i;
    if (i >= N) { st[0] = st[N - 1]; i = 1; }
  }
  st[0] = 0x80000000u;
  st[N] = N;
  return MEMHIP_OK;
}

extern "C" double memhip_mt_random(uint32_t* st) { return MT(st).random(); }

extern "C" int memhip_mask_blockwise(uint32_t* st, int H, int W, int num_masking_patches,
                                     int min_num_patches, int max_num_patches, double log_lo,
                                     double log_hi, int n_masks, uint8_t* out) {
  MEMHIP_REQUIRE(st && out && H > 0 && W > 0 && n_masks >= 0, "mask_blockwise: bad arguments");
  MEMHIP_REQUIRE(st[N] <= (uint32_t)N, "mask_blockwise: corrupt MT19937 position");
  MT rng(st);
  const int cap = max_num_patches < 0 ? num_masking_patches : max_num_patches;
  for (int m = 0; m < n_masks; ++m) {
    uint8_t* mask = out + (size_t)m * H * W;
    std::memset(mask, 0, (size_t)H * W);
    int count = 0;
    while (count < num_masking_patches) {
      int budget = num_masking_patches - count;
      if (budget > cap) budget = cap;
      int delta = 0;
      for (int attempt = 0; attempt < 10; ++attempt) {
        const double area = rng.uniform((double)min_num_patches, (double)budget);
        const double ar = std::exp(rng.uniform(log_lo, log_hi));
        const long h = py_round(std::sqrt(area * ar));
        const long w = py_round(std::sqrt(area / ar));
        if (w < W && h < H) {
          const long top = rng.randint(0, H - h);
          const long left = rng.randint(0, W - w);
          long already = 0;
          for (long i = top; i < top + h; ++i)
            for (long j = left; j < left + w; ++j) already += mask[i * W + j];
          const long fresh = h * w - already;
          if (0 < fresh && fresh <= budget) {
            for (long i = top; i < top + h; ++i)
              for (long j = left; j < left + w; ++j) mask[i * W + j] = 1;
            delta = (int)fresh;
          }
          if (delta > 0) break;
        }
      }
      if (delta == 0) break;
      count += delta;
    }
  }
  return MEMHIP_OK;
}

extern "C" int memhip_mask_random_location(uint32_t* st, int H, int W, int num_masking_patches,
                                           int n_masks, uint8_t* out) {
  MEMHIP_REQUIRE(st && out && H > 0 && W > 0 && n_masks >= 0, "mask_random_location: bad arguments");
  const int n = H * W - 1;   // reference off-by-one: the last patch is never maskable
  const int k = num_masking_patches;
  MEMHIP_REQUIRE(k >= 0 && k <= n, "mask_random_location: sample larger than population");
  MT rng(st);
  // random.sample(): pool branch when n <= setsize, set-rejection branch otherwise
  long setsize = 21;
  if (k > 5) setsize += (long)std::pow(4.0, std::ceil(std::log((double)k * 3) / std::log(4.0)));
  std::vector<int> pool(n);
  std::vector<uint8_t> seen(n);
  for (int m = 0; m < n_masks; ++m) {
    uint8_t* mask = out + (size_t)m * H * W;
    std::memset(mask, 0, (size_t)H * W);
    if (n <= setsize) {
      for (int i = 0; i < n; ++i) pool[i] = i;
      for (int i = 0; i < k; ++i) {
        const uint32_t j = rng.randbelow((uint32_t)(n - i));
        mask[pool[j]] = 1;
        pool[j] = pool[n - i - 1];
      }
    } else {
      std::fill(seen.begin(), seen.end(), 0);
      for (int i = 0; i < k; ++i) {
        uint32_t j = rng.randbelow((uint32_t)n);
        while (seen[j]) j = rng.randbelow((uint32_t)n);
        seen[j] = 1;
        mask[j] = 1;
      }
    }
  }
  return MEMHIP_OK;
}
