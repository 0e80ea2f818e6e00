/* Double-precision log2 / exp2 / log built from IEEE +, -, *, / only - compiled, unchanged, into BOTH the HIP kernels
 * (kernels.hpp) and the CPU oracle (oracle/ludwig_oracle.c), always with -ffp-contract=off.
 *
 * Why: the wall model (reference src/physics_kernels.jl:206-236) is the only place the hot path leaves +,-,*,/,sqrt:
 * `x^(1/7)` and `log(y+)`. Julia evaluates Float32 `^` as Float32(exp2(log2(Float64(x)) * y)) and Float32 `log` through a
 * Float64 kernel, i.e. both are "an accurate double result, rounded once to Float32". The device library (ocml) and glibc
 * each provide such functions, but their last double bit differs now and then, and once in ~1e8 calls that survives the
 * rounding to Float32 and then grows in a turbulent run. With one shared implementation the device and the oracle
 * produce the SAME bits by construction, so wall-model cases are compared for equality like everything else.
 * Accuracy (tests/test_jl_math.py, against glibc): <= 4 ulp in double; the Float32-rounded results of pow/log agree with glibc's on
 * all but ~1e-8 of random inputs - the same standing as ocml's or Julia's own kernels relative to each other.
 *
 * Domain used by the hot path: x positive, finite. Zero, negative, infinite and NaN arguments follow IEEE conventions. */
#ifndef LW_JL_MATH_H
#define LW_JL_MATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define LW_HD __host__ __device__ static inline
#else
#define LW_HD static inline
#endif

LW_HD uint64_t lw_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
LW_HD double lw_from_bits(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }

/* x = m * 2^e with m in [sqrt(1/2), sqrt(2)); returns log(m) by the atanh series, 2 s (1 + z/3 + z^2/5 + ...), z = s^2 <= 0.0295 */
LW_HD double lw_log_mantissa(double x, int *e_out)
{
    uint64_t u = lw_bits(x);
    int e = (int)((u >> 52) & 0x7FF);
    if (e == 0) {                                    /* subnormal double: scale up by 2^54 (never reached from Float32 input) */
        u = lw_bits(x * 18014398509481984.0);
        e = (int)((u >> 52) & 0x7FF) - 54;
    }
    e -= 1023;
    double m = lw_from_bits((u & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL);   /* [1, 2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    *e_out = e;
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z;                                       /* z/3 + z^2/5 + ... */
    return 2.0 * s + 2.0 * s * p;
}

LW_HD int lw_log_special(double x, double *out)
{
    const uint64_t u = lw_bits(x);
    if (x != x) { *out = x; return 1; }
    if (x == 0.0) { *out = lw_from_bits(0xFFF0000000000000ULL); return 1; }          /* -Inf */
    if (u >> 63) { *out = lw_from_bits(0x7FF8000000000000ULL); return 1; }           /* NaN  */
    if (u == 0x7FF0000000000000ULL) { *out = x; return 1; }                          /* +Inf */
    return 0;
}

LW_HD double lw_log(double x)
{
    double r;
    if (lw_log_special(x, &r)) return r;
    int e;
    const double lm = lw_log_mantissa(x, &e);
    /* ln 2 split in two so that e * hi is exact for |e| < 2^11 */
    return (double)e * 0.693147180369123816490 + ((double)e * 1.90821492927058770002e-10 + lm);
}

LW_HD double lw_log2(double x)
{
    double r;
    if (lw_log_special(x, &r)) return r;
    int e;
    const double lm = lw_log_mantissa(x, &e);
    /* 1/ln 2 split: hi has 32 significant bits */
    const double hi = 1.44269504072144627571, lo = 1.67517131648865118353e-10;
    return (double)e + (lm * hi + lm * lo);
}

LW_HD double lw_exp2(double x)
{
    if (x != x) return x;
    if (x >= 1024.0) return lw_from_bits(0x7FF0000000000000ULL);
    if (x < -1022.0) return 0.0;                     /* no subnormal results: flushed (outside the hot path's range) */
    /* n = nearest integer, r = x - n in [-0.5, 0.5] (exact) */
    const double t = x + 6755399441055744.0;         /* 1.5 * 2^52: round to nearest integer in the low bits */
    const int n = (int)(int32_t)(lw_bits(t) & 0xFFFFFFFFULL);
    const double r = x - (t - 6755399441055744.0);
    const double y = r * 0.693147180559945309417;    /* |y| <= 0.3466 */
    /* exp(y) by its Taylor series to y^14 / 14!  (0.3466^15 / 15! = 1e-19) */
    double p = 1.0 / 87178291200.0;
    p = p * y + 1.0 / 6227020800.0;
    p = p * y + 1.0 / 479001600.0;
    p = p * y + 1.0 / 39916800.0;
    p = p * y + 1.0 / 3628800.0;
    p = p * y + 1.0 / 362880.0;
    p = p * y + 1.0 / 40320.0;
    p = p * y + 1.0 / 5040.0;
    p = p * y + 1.0 / 720.0;
    p = p * y + 1.0 / 120.0;
    p = p * y + 1.0 / 24.0;
    p = p * y + 1.0 / 6.0;
    p = p * y + 0.5;
    p = p * y + 1.0;
    p = p * y + 1.0;
    if (n > 1023) return (p * lw_from_bits((uint64_t)2046 << 52)) * 2.0;   /* x in [1023.5, 1024) */
    return p * lw_from_bits((uint64_t)(n + 1023) << 52);
}

/* Base.^(::Float32, ::Float32) = Float32(exp2(log2(abs(widen(x))) * y)) (Julia base/math.jl pow_body); x > 0 here */
LW_HD float lw_powf(float x, float y) { return (float)lw_exp2(lw_log2((double)x) * (double)y); }
/* Base.log(::Float32): a Float64 kernel rounded once */
LW_HD float lw_logf(float x) { return (float)lw_log((double)x); }

#endif
