/* oracle/mt19937.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * Restates the RNG the reference samples with: std::mt19937 rng_ seeded by
 * rng_.seed(seed) (rela/prioritized_replay.h:183,346) and drawn through a fresh
 * std::uniform_real_distribution<float>(0, segment) per call (:267,279).
 * The generator is a third-party algorithm (libstdc++ 11.4, <bits/random.tcc>):
 *   seed(v):  x[0]=v; x[i] = 1812433253*(x[i-1]^(x[i-1]>>30)) + i
 *   twist/temper: MT19937 (n=624,m=397,a=0x9908b0df,u=11,s=7,b=0x9d2c5680,t=15,
 *                 c=0xefc60000,l=18)
 *   generate_canonical<float,24>: one 32-bit draw; float(u)/float(2^32); >=1 -> nextafter(1,0)
 *   uniform_real<float>(a,b): canonical*(b-a)+a
 */
#include <math.h>

#include "oracle.h"

void oracle_mt_seed(oracle_mt19937* g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i) {
    uint32_t x = g->mt[i - 1];
    g->mt[i] = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
  }
  g->idx = 624;
}

static void mt_twist(oracle_mt19937* g) {
  for (int i = 0; i < 624; ++i) {
    uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
    uint32_t v = g->mt[(i + 397) % 624] ^ (y >> 1);
    if (y & 1u) v ^= 0x9908b0dfu;
    g->mt[i] = v;
  }
  g->idx = 0;
}

uint32_t oracle_mt_next(oracle_mt19937* g) {
  if (g->idx >= 624) mt_twist(g);
  uint32_t y = g->mt[g->idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

float oracle_canonical_from_u32(uint32_t u) {
  volatile float sum = (float)u; /* round-to-nearest-even to 24 bits */
  volatile float ret = sum / 4294967296.0f;
  if (ret >= 1.0f) ret = nextafterf(1.0f, 0.0f);
  return ret;
}

float oracle_uniform_float(oracle_mt19937* g, float a, float b) {
  volatile float c = oracle_canonical_from_u32(oracle_mt_next(g));
  volatile float scaled = c * (b - a);
  return scaled + a;
}
