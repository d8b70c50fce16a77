// sleef_powf_core.h -- restatement of SLEEF's single-precision pow with 1.0 ULP bound (Sleef_powf*_u10), the
// function behind `torch::pow(priority, alpha)` in the reference (rela/prioritized_replay.h:188,239 -> ATen's
// vectorised CPU pow -> Vectorized<float>::pow -> Sleef_powf16_u10 / Sleef_powf8_u10; SLEEF 3.x as bundled
// with PyTorch 2.10, FMA builds).  Third-party algorithm, restated from its published structure:
//     pow(x, y) = expk( logk(|x|) * y )    in double-float ("df") arithmetic
//     logk: m * 2^e = x with m in [0.75, 1.5);  t = (m - 1) / (m + 1);  log x = e*ln2 + 2t + t^3 * P(t^2)
//     expk: q = rint(d / ln2);  s = d - q*ln2 (two-part ln2);  e^s = 1 + s + s^2 * Q(s);  result * 2^q
// Every operation is an IEEE single operation (add, mul, fma, div), so a GPU can reproduce the CPU bits:
// the replay stores exactly the weights the reference stores for alpha != 1 (tests/golden/*_a06*.json
// `stored_w`, tests/golden/sleef_powf_vectors.json).  Only the domain the replay uses is restated: x >= 0
// finite, y finite (the special cases x = 0, x = 1, y = 0 follow SLEEF; negative bases are not needed).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef RELA_HD
#ifdef __HIPCC__
#define RELA_HD __host__ __device__ __forceinline__
#else
#define RELA_HD inline
#endif
#endif

namespace rela_amd {
namespace sleef {

struct f2 {
  float x, y;
};

#ifdef __HIP_DEVICE_COMPILE__
RELA_HD float fadd(float a, float b) { return __fadd_rn(a, b); }
RELA_HD float fsub(float a, float b) { return __fsub_rn(a, b); }
RELA_HD float fmul(float a, float b) { return __fmul_rn(a, b); }
RELA_HD float ffma(float a, float b, float c) { return __fmaf_rn(a, b, c); }
RELA_HD float fdiv(float a, float b) { return __fdiv_rn(a, b); }
RELA_HD uint32_t f2u(float x) { return __float_as_uint(x); }
RELA_HD float u2f(uint32_t u) { return __uint_as_float(u); }
#else
// host build (tests/cpu_shims): compile with -ffp-contract=off; fmaf is the correctly rounded libm one
RELA_HD float fadd(float a, float b) { volatile float r = a + b; return r; }
RELA_HD float fsub(float a, float b) { volatile float r = a - b; return r; }
RELA_HD float fmul(float a, float b) { volatile float r = a * b; return r; }
RELA_HD float ffma(float a, float b, float c) { return fmaf(a, b, c); }
RELA_HD float fdiv(float a, float b) { volatile float r = a / b; return r; }
RELA_HD uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
RELA_HD float u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
#endif

RELA_HD f2 mk(float x, float y) { f2 r; r.x = x; r.y = y; return r; }

// ---- double-float helpers, FMA forms (sleef: src/common/df.h, ENABLE_FMA_SP) --------------------------
RELA_HD f2 df_normalize(f2 t) {
  const float sx = fadd(t.x, t.y);
  return mk(sx, fadd(fsub(t.x, sx), t.y));
}
RELA_HD f2 df_scale(f2 d, float s) { return mk(fmul(d.x, s), fmul(d.y, s)); }
RELA_HD f2 df_add2_f_f(float x, float y) {  // dfadd2_vf2_vf_vf
  const float rx = fadd(x, y);
  const float v = fsub(rx, x);
  return mk(rx, fadd(fsub(x, fsub(rx, v)), fsub(y, v)));
}
RELA_HD f2 df_add2_f2_f(f2 x, float y) {  // dfadd2_vf2_vf2_vf
  const float rx = fadd(x.x, y);
  const float v = fsub(rx, x.x);
  const float ry = fadd(fsub(x.x, fsub(rx, v)), fsub(y, v));
  return mk(rx, fadd(ry, x.y));
}
RELA_HD f2 df_add2_f2_f2(f2 x, f2 y) {  // dfadd2_vf2_vf2_vf2
  const float rx = fadd(x.x, y.x);
  const float v = fsub(rx, x.x);
  const float ry = fadd(fsub(x.x, fsub(rx, v)), fsub(y.x, v));
  return mk(rx, fadd(ry, fadd(x.y, y.y)));
}
RELA_HD f2 df_add_f2_f2(f2 x, f2 y) {  // dfadd_vf2_vf2_vf2: |x| >= |y|
  const float rx = fadd(x.x, y.x);
  return mk(rx, fadd(fadd(fadd(fsub(x.x, rx), y.x), x.y), y.y));
}
RELA_HD f2 df_add_f_f2(float x, f2 y) {  // dfadd_vf2_vf_vf2
  const float rx = fadd(x, y.x);
  return mk(rx, fadd(fadd(fsub(x, rx), y.x), y.y));
}
RELA_HD f2 df_mul_f2_f(f2 x, float y) {  // dfmul_vf2_vf2_vf
  const float rx = fmul(x.x, y);
  return mk(rx, ffma(x.y, y, ffma(x.x, y, -rx)));
}
RELA_HD f2 df_mul_f2_f2(f2 x, f2 y) {  // dfmul_vf2_vf2_vf2
  const float rx = fmul(x.x, y.x);
  return mk(rx, ffma(x.x, y.y, ffma(x.y, y.x, ffma(x.x, y.x, -rx))));
}
RELA_HD f2 df_squ(f2 x) {  // dfsqu_vf2_vf2
  const float rx = fmul(x.x, x.x);
  return mk(rx, ffma(fadd(x.x, x.x), x.y, ffma(x.x, x.x, -rx)));
}
RELA_HD f2 df_div(f2 n, f2 d) {  // dfdiv_vf2_vf2_vf2
  const float t = fdiv(1.0f, d.x);
  const float q0 = fmul(n.x, t);
  const float u = ffma(t, n.x, -q0);
  float q1 = ffma(-d.y, t, ffma(-d.x, t, 1.0f));
  q1 = ffma(q0, q1, ffma(n.y, t, u));
  return mk(q0, q1);
}

// vgetexp / vgetmant of the AVX-512 build (_mm512_getexp_ps, _mm512_getmant_ps(_MM_MANT_NORM_p75_1p5)) for
// positive finite non-zero inputs; subnormals are treated as normalised numbers, as the instructions do
RELA_HD void split_pos(float d, int* e, uint32_t* mant) {
  uint32_t u = f2u(d);
  int ex = (int)(u >> 23);
  uint32_t m = u & 0x7fffffu;
  if (ex == 0) {  // subnormal: normalise
    ex = 1;
    while (!(m & 0x800000u)) {
      m <<= 1;
      ex -= 1;
    }
    m &= 0x7fffffu;
  }
  *e = ex - 127;
  *mant = m;
}
RELA_HD float getexp_pos(float d) {
  int e;
  uint32_t m;
  split_pos(d, &e, &m);
  return (float)e;
}
RELA_HD float getmant_p75_1p5(float d) {
  int e;
  uint32_t m;
  split_pos(d, &e, &m);
  // [1, 2) mantissa; values >= 1.5 move to [0.75, 1) (exponent - 1)
  const uint32_t one = 0x3f800000u | m;
  return (m >= 0x400000u) ? u2f((0x3f000000u) | m) : u2f(one);
}

RELA_HD f2 logk(float d) {  // logkf, d > 0 finite
  const float sc = fmul(d, 1.0f / 0.75f);
  const float e = getexp_pos(sc);
  const float m = getmant_p75_1p5(d);
  const f2 x = df_div(df_add2_f_f(-1.0f, m), df_add2_f_f(1.0f, m));
  const f2 x2 = df_squ(x);
  float t = 0.240320354700088500976562f;
  t = ffma(t, x2.x, 0.285112679004669189453125f);
  t = ffma(t, x2.x, 0.400007992982864379882812f);
  const f2 c = mk(0.66666662693023681640625f, 3.69183861259614332084311e-09f);
  f2 s = df_mul_f2_f(mk(0.69314718246459960938f, -1.904654323148236017e-09f), e);
  s = df_add_f2_f2(s, df_scale(x, 2.0f));
  s = df_add_f2_f2(s, df_mul_f2_f2(df_mul_f2_f2(x2, x), df_add2_f2_f2(df_mul_f2_f(x2, t), c)));
  return s;
}

RELA_HD float ldexp2k(float x, int q) {  // vldexp_vf_vf_vi2
  int m = q >> 31;
  m = (((m + q) >> 6) - m) << 4;
  q = q - (m << 2);
  m += 0x7f;
  m = m < 0 ? 0 : m;
  m = m > 0xff ? 0xff : m;
  float u = u2f((uint32_t)m << 23);
  x = fmul(fmul(fmul(fmul(x, u), u), u), u);
  u = u2f((uint32_t)(q + 0x7f) << 23);
  return fmul(x, u);
}

RELA_HD float expk(f2 d) {  // expkf
  float u = fmul(fadd(d.x, d.y), 1.442695040888963407359924681001892137426645954152985934135449406931f);
  const int q = (int)rintf(u);  // vrint_vi2_vf: round to nearest even
  f2 s = df_add2_f2_f(d, fmul((float)q, -0.693145751953125f));
  s = df_add2_f2_f(s, fmul((float)q, -1.428606765330187045e-06f));
  s = df_normalize(s);
  u = 0.00136324646882712841033936f;
  u = ffma(u, s.x, 0.00836596917361021041870117f);
  u = ffma(u, s.x, 0.0416710823774337768554688f);
  u = ffma(u, s.x, 0.166665524244308471679688f);
  u = ffma(u, s.x, 0.499999850988388061523438f);
  f2 t = df_add_f2_f2(s, df_mul_f2_f(df_squ(s), u));
  t = df_add_f_f2(1.0f, t);
  u = fadd(t.x, t.y);
  u = ldexp2k(u, q);
  if (d.x < -104.0f) u = 0.0f;
  return u;
}

// Sleef_powf_u10 for x >= 0 finite and finite y
RELA_HD float powf_u10(float x, float y) {
  if (y == 0.0f || x == 1.0f) return 1.0f;
  if (x == 0.0f) return y < 0.0f ? INFINITY : 0.0f;
  float r = expk(df_mul_f2_f(logk(x), y));
  if (r != r) r = INFINITY;
  return r;
}

}  // namespace sleef
}  // namespace rela_amd
