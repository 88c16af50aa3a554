/* ccmp_detmath.h — bit-reproducible elementary functions (sincos, atan) and the FMA macro that
 * defines this project's canonical rounding model.
 *
 * Why this file exists
 * --------------------
 * The reference projector (include/closed_chain_motion_planner/base/constraints/
 * ConstraintFunction.h:57-82) differentiates its residual by finite differences with a step of
 * sqrt(DBL_EPSILON) and stops at the first iterate under tolerance.  A 1-ulp difference in one
 * sin() is amplified ~7e7 times in the Jacobian and then grows along the Newton trajectory
 * (measured: the same algorithm run with glibc sin/cos and with the sin/cos below ends > 1e-6 rad
 * apart on ~20 % of uniform samples and stops at a different iteration on ~11 %).  Agreement "to
 * 1e-6 rad" between a GPU kernel and a CPU restatement therefore needs *bitwise identical*
 * arithmetic.  Vendor libraries (glibc libm on the host, ocml on the device) cannot give that.
 * This header can: only IEEE-754 double + - * / sqrt fma and comparisons, in a fixed order.
 *
 * Rounding model
 * --------------
 * CCMP_FMA(a,b,c) is a*b+c.  With CCMP_USE_FMA it is one fused operation (v_fma_f64 on gfx950,
 * vfmadd on x86 with -mfma, libm fma() otherwise — all exact); without it, a rounded multiply
 * followed by a rounded add.  Every translation unit that must agree bitwise is compiled with
 * -ffp-contract=off so that only the FMAs written here and in ccmp_kin.h exist.
 * The "fast" analytic kernel compiles the same source with contraction on and no bitwise claim.
 *
 * Algorithms: the classical Sun fdlibm scheme (published algorithm and minimax coefficients):
 * Cody-Waite reduction by pi/2 split into 33-bit pieces (exact products for |x| < 2^20*pi/2),
 * degree-13/14 kernels on [-pi/4, pi/4] in branch-free form, 4-interval atan reduction with one
 * division.  Accuracy (tests/test_detmath.py, against glibc): <= 1 ulp sin/cos, <= 2 ulp atan.
 */
#ifndef CCMP_DETMATH_H
#define CCMP_DETMATH_H

#if defined(__HIPCC__)
#define CCMP_HD __host__ __device__ __forceinline__
#else
#define CCMP_HD static inline
#endif

#if defined(CCMP_USE_FMA)
#define CCMP_FMA(a, b, c) __builtin_fma((a), (b), (c))
#else
#define CCMP_FMA(a, b, c) ((a) * (b) + (c))
#endif

/* CCMP_K(i, v): the i-th FP64 literal of the elementary functions below.  By default the literal itself.  A device
 * translation unit may define CCMP_K before including this header to fetch the constant from a table instead —
 * ccmp_flat_newton.h keeps one in LDS, filled by ccmp_fill_ktab() below, because a lone wavefront pays two move
 * instructions per FP64 literal and use, and a 16-byte LDS read brings two constants for one.  Same values either way. */
#ifndef CCMP_K
#define CCMP_K(i, v) (v)
#endif
#define CCMP_K_COUNT 36

/* Largest |x| the pi/2 reduction is exact for (2^20 * pi/2).  Beyond it sincos returns NaN:
 * a Newton iterate that large is a diverged sample, never a valid projection. */
#define CCMP_SINCOS_MAX 1647099.0

/* IEEE correctly rounded square root on both sides.  On gfx950 the compiler expands __builtin_sqrt into v_rsq_f64, nine
 * fused steps and — for arguments below 2^-767, whose intermediates would go subnormal — a scaling by 2^256 / 2^-128
 * around them, plus a pass-through for zero and infinity: 20 instructions, 8 of them spent on cases the kernels meet
 * only at an exact zero.  CCMP_LEAN_SQRT runs the same nine steps directly when NO lane of the wavefront holds such an
 * argument (wave-uniform test on the exponent field: 2 instructions) — without the scaling the steps are the very same
 * operations on the very same values, hence the same bits — and the compiler's full expansion otherwise. */
#if defined(__HIP_DEVICE_COMPILE__) && defined(CCMP_LEAN_SQRT)
static __device__ __forceinline__ double ccmp_sqrt(double x)
{
  const unsigned hi = (unsigned)__double2hiint(x);
  /* 2^-767 <= x < inf  <=>  0x10000000 <= high dword < 0x7ff00000 (sign bit clear) */
  const bool plain = (hi - 0x10000000u) < (0x7ff00000u - 0x10000000u);
  if (__builtin_amdgcn_ballot_w64(!plain) == 0ull) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = CCMP_FMA(-h, g, 0.5);
    g = CCMP_FMA(g, r, g);
    h = CCMP_FMA(h, r, h);
    double d = CCMP_FMA(-g, g, x);
    g = CCMP_FMA(d, h, g);
    d = CCMP_FMA(-g, g, x);
    return CCMP_FMA(d, h, g);
  }
  return __builtin_sqrt(x);
}
#else
CCMP_HD double ccmp_sqrt(double x) { return __builtin_sqrt(x); }
#endif
CCMP_HD double ccmp_abs(double x) { return __builtin_fabs(x); }

/* sin(x+y) and cos(x+y), |x| <= pi/4 (+ a hair), y the tail of the reduced argument. */
CCMP_HD double ccmp_kernel_sin(double x, double y)
{
  const double S1 = CCMP_K(4, -1.66666666666666324348e-01), S2 = CCMP_K(5, 8.33333333332248946124e-03),
               S3 = CCMP_K(6, -1.98412698298579493134e-04), S4 = CCMP_K(7, 2.75573137070700676789e-06),
               S5 = CCMP_K(8, -2.50507602534068634195e-08), S6 = CCMP_K(9, 1.58969099521155010221e-10);
  double z = x * x;
  double w = z * z;
  double p1 = CCMP_FMA(z, S4, S3);
  p1 = CCMP_FMA(z, p1, S2);
  double p2 = CCMP_FMA(z, S6, S5);
  double r = CCMP_FMA(z * w, p2, p1);
  double v = z * x;
  double in = CCMP_FMA(-v, r, 0.5 * y);
  double t2 = CCMP_FMA(z, in, -y);
  double t3 = CCMP_FMA(-v, S1, t2);
  return x - t3;
}

CCMP_HD double ccmp_kernel_cos(double x, double y)
{
  const double C1 = CCMP_K(10, 4.16666666666666019037e-02), C2 = CCMP_K(11, -1.38888888888741095749e-03),
               C3 = CCMP_K(12, 2.48015872894767294178e-05), C4 = CCMP_K(13, -2.75573143513906633035e-07),
               C5 = CCMP_K(14, 2.08757232129817482790e-09), C6 = CCMP_K(15, -1.13596475577881948265e-11);
  double z = x * x;
  double w = z * z;
  double q1 = CCMP_FMA(z, C3, C2);
  q1 = CCMP_FMA(z, q1, C1);
  q1 = z * q1;
  double q2 = CCMP_FMA(z, C6, C5);
  q2 = CCMP_FMA(z, q2, C4);
  double r = CCMP_FMA(w * w, q2, q1);
  double hz = 0.5 * z;
  double w1 = 1.0 - hz;
  return w1 + (((1.0 - w1) - hz) + CCMP_FMA(z, r, -(x * y)));
}

/* Simultaneous sine and cosine.  NaN for non-finite or |x| >= CCMP_SINCOS_MAX. */
CCMP_HD void ccmp_sincos(double x, double *s_out, double *c_out)
{
  const double invpio2 = CCMP_K(0, 6.36619772367581382433e-01);
  const double pio2_1 = CCMP_K(1, 1.57079632673412561417e+00);  /* first 33 bits of pi/2 */
  const double pio2_2 = CCMP_K(2, 6.07710050630396597660e-11);  /* second 33 bits */
  const double pio2_2t = CCMP_K(3, 2.02226624879595063154e-21); /* pi/2 - (pio2_1 + pio2_2) */
  double t = ccmp_abs(x);
  /* out of range (or not finite): the reduction below runs on zero instead and its result is replaced at the end —
   * one rarely-taken region instead of an if/else around the whole function */
  const int bad = !(t < CCMP_SINCOS_MAX);
  if (bad) t = 0.0;
  /* n = nearest multiple of pi/2; fn*pio2_1 is exact (33 + 20 bits) */
  int n = (int)CCMP_FMA(t, invpio2, 0.5);
  double fn = (double)n;
  double r = CCMP_FMA(-fn, pio2_1, t);
  double w = fn * pio2_2;
  double tt = r;
  r = tt - w;
  w = CCMP_FMA(fn, pio2_2t, -((tt - r) - w));
  double y0 = r - w;
  double y1 = (r - y0) - w;
  double ks = ccmp_kernel_sin(y0, y1);
  double kc = ccmp_kernel_cos(y0, y1);
  /* quadrant of |x|; then sin is odd, cos is even */
  double s = (n & 1) ? kc : ks;
  double c = (n & 1) ? ks : kc;
  if (n & 2) s = -s;
  if ((n + 1) & 2) c = -c;
  s = x < 0.0 ? -s : s;
  if (bad) {
    double nan = (x - x) / (x - x); /* NaN for inf, NaN and out-of-range finite x alike */
    s = nan;
    c = nan;
  }
  *s_out = s;
  *c_out = c;
}

/* atan(x) for any double (NaN propagates). */
CCMP_HD double ccmp_atan(double x)
{
  const double hi0 = CCMP_K(16, 4.63647609000806093515e-01), lo0 = CCMP_K(17, 2.26987774529616870924e-17); /* atan(0.5) */
  const double hi1 = CCMP_K(18, 7.85398163397448278999e-01), lo1 = CCMP_K(19, 3.06161699786838301793e-17); /* atan(1)   */
  const double hi2 = CCMP_K(20, 9.82793723247329054082e-01), lo2 = CCMP_K(21, 1.39033110312309984516e-17); /* atan(1.5) */
  const double hi3 = CCMP_K(22, 1.57079632679489655800e+00), lo3 = CCMP_K(23, 6.12323399573676603587e-17); /* atan(inf) */
  const double a0 = CCMP_K(24, 3.33333333333329318027e-01), a1 = CCMP_K(25, -1.99999999998764832476e-01),
               a2 = CCMP_K(26, 1.42857142725034663711e-01), a3 = CCMP_K(27, -1.11111104054623557880e-01),
               a4 = CCMP_K(28, 9.09088713343650656196e-02), a5 = CCMP_K(29, -7.69187620504482999495e-02),
               a6 = CCMP_K(30, 6.66107313738753120669e-02), a7 = CCMP_K(31, -5.83357013379057348645e-02),
               a8 = CCMP_K(32, 4.97687799461593236017e-02), a9 = CCMP_K(33, -3.65315727442169155270e-02),
               a10 = CCMP_K(34, 1.62858201153657823623e-02);
  int neg = x < 0.0;
  double ax = ccmp_abs(x);
  if (ax >= 7.378697629483821e19) { /* 2^66: atan saturates (also catches +-inf) */
    double z = hi3 + lo3;
    return neg ? -z : z;
  }
  /* interval selection expressed as one quotient num/den so every lane runs one divide */
  double num, den, hi, lo;
  int direct = 0;
  if (ax < 0.4375) {
    num = ax; den = 1.0; hi = 0.0; lo = 0.0; direct = 1;
  } else if (ax < 0.6875) {
    num = CCMP_FMA(2.0, ax, -1.0); den = 2.0 + ax; hi = hi0; lo = lo0;
  } else if (ax < 1.1875) {
    num = ax - 1.0; den = ax + 1.0; hi = hi1; lo = lo1;
  } else if (ax < 2.4375) {
    num = ax - 1.5; den = CCMP_FMA(1.5, ax, 1.0); hi = hi2; lo = lo2;
  } else { /* also the NaN path: every comparison above is false */
    num = -1.0; den = ax; hi = hi3; lo = lo3;
  }
  double t = num / den;
  double z = t * t;
  double w = z * z;
  double s1 = CCMP_FMA(w, a10, a8);
  s1 = CCMP_FMA(w, s1, a6);
  s1 = CCMP_FMA(w, s1, a4);
  s1 = CCMP_FMA(w, s1, a2);
  s1 = CCMP_FMA(w, s1, a0);
  s1 = z * s1;
  double s2 = CCMP_FMA(w, a9, a7);
  s2 = CCMP_FMA(w, s2, a5);
  s2 = CCMP_FMA(w, s2, a3);
  s2 = CCMP_FMA(w, s2, a1);
  s2 = w * s2;
  double res;
  if (direct)
    res = CCMP_FMA(-t, s1 + s2, t);
  else
    res = hi - (CCMP_FMA(t, s1 + s2, -lo) - t);
  return neg ? -res : res;
}

/* the table CCMP_K may be redirected to: entry i = the literal CCMP_K(i, .) names */
CCMP_HD void ccmp_fill_ktab(double *t)
{
  t[0] = 6.36619772367581382433e-01; /* invpio2 */
  t[1] = 1.57079632673412561417e+00; /* pio2_1 */
  t[2] = 6.07710050630396597660e-11; /* pio2_2 */
  t[3] = 2.02226624879595063154e-21; /* pio2_2t */
  t[4] = -1.66666666666666324348e-01; /* S1 */
  t[5] = 8.33333333332248946124e-03; /* S2 */
  t[6] = -1.98412698298579493134e-04; /* S3 */
  t[7] = 2.75573137070700676789e-06; /* S4 */
  t[8] = -2.50507602534068634195e-08; /* S5 */
  t[9] = 1.58969099521155010221e-10; /* S6 */
  t[10] = 4.16666666666666019037e-02; /* C1 */
  t[11] = -1.38888888888741095749e-03; /* C2 */
  t[12] = 2.48015872894767294178e-05; /* C3 */
  t[13] = -2.75573143513906633035e-07; /* C4 */
  t[14] = 2.08757232129817482790e-09; /* C5 */
  t[15] = -1.13596475577881948265e-11; /* C6 */
  t[16] = 4.63647609000806093515e-01; /* hi0 */
  t[17] = 2.26987774529616870924e-17; /* lo0 */
  t[18] = 7.85398163397448278999e-01; /* hi1 */
  t[19] = 3.06161699786838301793e-17; /* lo1 */
  t[20] = 9.82793723247329054082e-01; /* hi2 */
  t[21] = 1.39033110312309984516e-17; /* lo2 */
  t[22] = 1.57079632679489655800e+00; /* hi3 */
  t[23] = 6.12323399573676603587e-17; /* lo3 */
  t[24] = 3.33333333333329318027e-01; /* a0 */
  t[25] = -1.99999999998764832476e-01; /* a1 */
  t[26] = 1.42857142725034663711e-01; /* a2 */
  t[27] = -1.11111104054623557880e-01; /* a3 */
  t[28] = 9.09088713343650656196e-02; /* a4 */
  t[29] = -7.69187620504482999495e-02; /* a5 */
  t[30] = 6.66107313738753120669e-02; /* a6 */
  t[31] = -5.83357013379057348645e-02; /* a7 */
  t[32] = 4.97687799461593236017e-02; /* a8 */
  t[33] = -3.65315727442169155270e-02; /* a9 */
  t[34] = 1.62858201153657823623e-02; /* a10 */
  t[35] = 0.0;
}

/* atan2(y, x) restricted to y >= 0, x >= 0 — the only quadrant the residual's
 * angularDistance needs (2*atan2(|vec|, |w|), ConstraintFunction.h:94-97 via Eigen).
 * atan2(0,0) = 0 as in C. */
CCMP_HD double ccmp_atan2_nn(double y, double x)
{
  if (x == 0.0 && y == 0.0) return 0.0;
  return ccmp_atan(y / x); /* x == 0 < y gives +inf -> pi/2 */
}

/* Natural logarithm for finite normal x > 0 (the only use: Box-Muller on u in (0,1]).  fdlibm scheme:
 * x = 2^k (1+f), sqrt(2)/2 <= 1+f < sqrt(2), log(1+f) from s = f/(2+f) with a degree-14 minimax. */
CCMP_HD double ccmp_log(double x)
{
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  unsigned long long u;
  __builtin_memcpy(&u, &x, 8);
  int k = (int)((u >> 52) & 0x7ff) - 1023;
  unsigned long long m = (u & 0x000fffffffffffffULL);
  /* mantissa >= sqrt(2): use the next binade so that 1+f stays in [sqrt(2)/2, sqrt(2)) */
  if (m >= 0x6a09e667f3bcdULL) { k += 1; m |= 0x3fe0000000000000ULL; }
  else m |= 0x3ff0000000000000ULL;
  double xm;
  __builtin_memcpy(&xm, &m, 8);
  const double f = xm - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  double t1 = CCMP_FMA(w, Lg6, Lg4);
  t1 = CCMP_FMA(w, t1, Lg2);
  t1 = w * t1;
  double t2 = CCMP_FMA(w, Lg7, Lg5);
  t2 = CCMP_FMA(w, t2, Lg3);
  t2 = CCMP_FMA(w, t2, Lg1);
  t2 = z * t2;
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  return CCMP_FMA(dk, ln2_hi, -((hfsq - CCMP_FMA(s, hfsq + R, dk * ln2_lo)) - f));
}

#endif /* CCMP_DETMATH_H */
