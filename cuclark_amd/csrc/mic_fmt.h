// mic_fmt.h — the text printf("%g") gives for (double)num / den, produced with integer arithmetic only, for host and
// device.  The CSV writer (CuCLARK_hh.hh:2127-2135) prints gamma = total / (len - k + 1) and confidence =
// best / (best + second) with "%g"; the device CSV kernel (mic_ingest.hip) must emit the same bytes without a C
// library.  "%g" = 6 significant digits, correctly rounded from the EXACT binary value of the double (ties to even),
// trailing zeros removed, scientific notation when the decimal exponent is < -4.
//
// The double is v = m * 2^-s (m: 53-bit integer); its first six digits are N = round_half_even(m * 10^P / 2^s) with
// P = 5 - floor(log10 v): one 64 x 64 -> 128-bit multiply, one shift, one exact remainder comparison.
#ifndef MIC_FMT_H
#define MIC_FMT_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define MIC_HD __host__ __device__
#else
#define MIC_HD
#endif

MIC_HD static inline uint64_t mic_fmt_mulhi(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

MIC_HD static inline uint64_t mic_fmt_bits(double v) {
  uint64_t b;
#if defined(__HIP_DEVICE_COMPILE__)
  b = (uint64_t)__double_as_longlong(v);
#else
  memcpy(&b, &v, 8);
#endif
  return b;
}

// first six significant digits of v (0 < v <= 1, normal): N in [100000, 999999] and the decimal exponent X of the
// leading digit, after rounding (0.9999996 gives N = 100000, X = 0)
MIC_HD static inline uint32_t mic_fmt_digits6(double v, int* x_out) {
  const uint64_t bits = mic_fmt_bits(v);
  const int e = (int)((bits >> 52) & 0x7FF);
  const uint64_t m = (bits & 0xFFFFFFFFFFFFFULL) | (1ULL << 52);
  const int s = 1075 - e;                       // v = m * 2^-s, s >= 52
  int X = 0;
  { double t = v; while (t < 1.0 && X > -40) { t *= 10.0; --X; } }   // a guess; corrected below
  uint64_t N = 0;
  for (int it = 0; it < 4; ++it) {
    const int P = 5 - X;                        // N = v * 10^P rounded
    uint64_t p10 = 1;
    for (int i = 0; i < P; ++i) p10 *= 10;      // P <= 19 here: fits 64 bits
    const uint64_t lo = m * p10, hi = mic_fmt_mulhi(m, p10);
    uint64_t q, rhi, rlo, hhi, hlo;             // quotient, remainder, half of the divisor 2^s
    if (s >= 64) {
      const int t = s - 64;
      q = t < 64 ? hi >> t : 0;
      rhi = t == 0 ? 0 : (t < 64 ? hi & ((1ULL << t) - 1) : hi);
      rlo = lo;
      if (t == 0) { hhi = 0; hlo = 1ULL << 63; } else { hhi = 1ULL << (t - 1); hlo = 0; }
    } else {
      q = (hi << (64 - s)) | (lo >> s);
      rhi = 0; rlo = lo & ((1ULL << s) - 1);
      hhi = 0; hlo = 1ULL << (s - 1);
    }
    const bool gt = rhi > hhi || (rhi == hhi && rlo > hlo), eq = rhi == hhi && rlo == hlo;
    if (gt || (eq && (q & 1))) ++q;
    N = q;
    if (N >= 1000000) { ++X; continue; }
    if (N < 100000) { --X; continue; }
    break;
  }
  *x_out = X;
  return (uint32_t)N;
}

// "%g" of v = (double)num / den for 0 < num <= den; returns the number of characters written (<= 13, no terminator)
MIC_HD static inline int mic_fmt_g_unit(double v, char* out) {
  if (v >= 1.0) { out[0] = '1'; return 1; }
  int X;
  uint32_t N = mic_fmt_digits6(v, &X);
  int nd = 6;
  while (nd > 1 && N % 10 == 0) { N /= 10; --nd; }
  char d[6];
  for (int i = nd - 1; i >= 0; --i) { d[i] = (char)('0' + N % 10); N /= 10; }
  int n = 0;
  if (X >= 0) {                                 // rounded up to 1
    out[n++] = d[0];
    if (nd > 1) { out[n++] = '.'; for (int i = 1; i < nd; ++i) out[n++] = d[i]; }
  } else if (X >= -4) {
    out[n++] = '0'; out[n++] = '.';
    for (int i = 0; i < -X - 1; ++i) out[n++] = '0';
    for (int i = 0; i < nd; ++i) out[n++] = d[i];
  } else {
    out[n++] = d[0];
    if (nd > 1) { out[n++] = '.'; for (int i = 1; i < nd; ++i) out[n++] = d[i]; }
    out[n++] = 'e'; out[n++] = '-';
    const int ax = -X;
    if (ax >= 100) out[n++] = (char)('0' + ax / 100);
    out[n++] = (char)('0' + (ax / 10) % 10);
    out[n++] = (char)('0' + ax % 10);
  }
  return n;
}

// gamma field: total / ((double)norm - k + 1.0) as the host computes it on x86-64 (CuCLARK_hh.hh:2127): a read shorter
// than k has no k-mer (total = 0) and a non-positive denominator: 0 / negative = "-0", 0 / 0 = "-nan" (the x86 default
// NaN carries the sign bit).  Returns -1 for a combination this formatter does not cover (total > denominator).
MIC_HD static inline int mic_fmt_gamma(uint32_t total, uint32_t norm, int k, char* out) {
  const double den = ((double)norm - (double)k) + 1.0;
  if (total == 0) {
    if (den > 0) { out[0] = '0'; return 1; }
    if (den < 0) { out[0] = '-'; out[1] = '0'; return 2; }
    out[0] = '-'; out[1] = 'n'; out[2] = 'a'; out[3] = 'n'; return 4;
  }
  if (!(den >= (double)total)) return -1;
  return mic_fmt_g_unit((double)total / den, out);
}

// confidence field: best / (best + second), 0 when both are 0 (CuCLARK_hh.hh:2128-2129)
MIC_HD static inline int mic_fmt_conf(uint32_t best, uint32_t second, char* out) {
  const uint32_t sum = best + second;
  if (sum == 0 || best == 0) { out[0] = '0'; return 1; }
  return mic_fmt_g_unit((double)best / (double)sum, out);
}

MIC_HD static inline int mic_fmt_u32(uint32_t v, char* out) {
  char t[10]; int n = 0;
  do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  for (int i = 0; i < n; ++i) out[i] = t[n - 1 - i];
  return n;
}

#endif
