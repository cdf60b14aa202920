/*
 * portable_math.h -- bit-reproducible fp64 tanh and log for the product-sum check update (row a5).
 *
 * The reference's product-sum pass calls the platform libm (`tanh(b2c / 2)`, `log((1 + x) / (1 - x))`, SURVEY.md
 * Appendix A.3), whose last-bit behaviour differs between glibc versions, CPU architectures (FMA or not) and the
 * device math library; belief propagation amplifies such last-bit differences until iteration counts and hard
 * decisions change on a few percent of shots (tests/test_gpu_parity.py::test_config2_product_sum_cs60_vs_golden).
 * The kernels therefore evaluate both functions with the routines below: only IEEE-754 correctly rounded +, -, *, /
 * in a fixed order (build with -ffp-contract=off), integer bit manipulation and comparisons, so the GPU and any CPU
 * compile of this file produce identical bits.  Algorithms: the classic table-free reductions
 *   log:   x = 2^k (1 + f), sqrt(2)/2 <= 1 + f < sqrt(2);  s = f / (2 + f);  log(1 + f) = f - (f^2/2 - s (f^2/2 + R(s^2)))
 *   expm1: x = k ln2 + r, |r| <= ln2 / 2;  expm1(r) from a rational approximation in r^2;  rescale by 2^k
 *   tanh:  1 - 2 / (expm1(2|x|) + 2) for |x| >= 1,  -t / (t + 2) with t = expm1(-2|x|) below
 * These are the reductions and the minimax coefficients (Lg1..Lg7, Q1..Q5, ln2_hi / ln2_lo) of FreeBSD's msun / Sun's
 * fdlibm -- e_log.c, s_expm1.c, s_tanh.c -- restated here without their table look-ups, special-case branches and
 * extended-precision tricks that do not matter for the argument ranges of the check update.  fdlibm's licence line:
 *   "Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.  Developed at SunPro, a Sun Microsystems, Inc.
 *    business.  Permission to use, copy, modify, and distribute this software is freely granted, provided that this
 *    notice is preserved."
 * Errors stay below 1 ulp (tests/test_portable_math.py measures them against libm).  Plain C99; usable from host C,
 * host C++ and HIP device code.  fdlibm is third-party, permissively licensed code; it is NOT part of the reference
 * repository (/root/reference contains no arithmetic for this path).
 */
#ifndef BPOSD_PORTABLE_MATH_H
#define BPOSD_PORTABLE_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PM_FN __host__ __device__ static inline
#else
#define PM_FN static inline
#endif

PM_FN uint64_t pm_bits(double x) {
    uint64_t u;
    memcpy(&u, &x, sizeof(u));
    return u;
}
PM_FN double pm_from_bits(uint64_t u) {
    double x;
    memcpy(&x, &u, sizeof(x));
    return x;
}

/* natural logarithm */
PM_FN double pm_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01; /* 0x3fe62e42fee00000 */
    const double ln2_lo = 1.90821492927058770002e-10; /* 0x3dea39ef35793c76 */
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                 L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    uint64_t u = pm_bits(x);
    int k = 0;
    if (x != x) return x;                                  /* NaN */
    if (u >> 63) {                                         /* negative (or -0) */
        if ((u << 1) == 0) return -1.0 / 0.0;              /* log(-0) = -inf */
        return (x - x) / 0.0;                              /* NaN */
    }
    if (u == 0) return -1.0 / 0.0;                         /* log(+0) = -inf */
    if (u == 0x7ff0000000000000ull) return x;              /* +inf */
    if (u < 0x0010000000000000ull) {                       /* subnormal: scale by 2^54 */
        x *= 18014398509481984.0;
        u = pm_bits(x);
        k -= 54;
    }
    /* x = 2^e * m, m in [1, 2); fold m >= sqrt(2) into the next binade so that 1 + f lies in [sqrt(2)/2, sqrt(2)) */
    k += (int)(u >> 52) - 1023;
    u = (u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m = pm_from_bits(u);
    if (m >= 1.4142135623730951) {
        m *= 0.5;
        k += 1;
    }
    const double f = m - 1.0;
    const double dk = (double)k;
    const double s = f / (2.0 + f);
    const double s2 = s * s;
    const double s4 = s2 * s2;
    const double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    const double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

/* exp(x) - 1 */
PM_FN double pm_expm1(double x) {
    const double o_threshold = 7.09782712893383973096e+02;
    const double ln2x56 = 3.88162421113569373274e+01;
    const double ln2halfx3 = 1.03972077083991796413e+00;
    const double ln2half = 3.46573590279972654709e-01;
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const double tiny = 5.5511151231257827e-17; /* 2^-54 */
    const double Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03, Q3 = -7.93650757867487942473e-05,
                 Q4 = 4.00821782732936239552e-06, Q5 = -2.01099218183624371326e-07;
    if (x != x) return x;
    double absx = x;
    int sign = 0;
    if (x < 0.0) {
        absx = -absx;
        sign = 1;
    }
    if (absx >= ln2x56) { /* |x| >= 56 ln2 */
        if (sign) return -1.0;
        if (absx >= o_threshold) return 1.0 / 0.0;
    }
    double c = 0.0;
    int k = 0;
    if (absx > ln2half) {
        double hi, lo;
        if (absx < ln2halfx3) {
            if (!sign) {
                hi = x - ln2_hi;
                lo = ln2_lo;
                k = 1;
            } else {
                hi = x + ln2_hi;
                lo = -ln2_lo;
                k = -1;
            }
        } else {
            k = (int)(sign ? invln2 * x - 0.5 : invln2 * x + 0.5);
            const double t = (double)k;
            hi = x - t * ln2_hi; /* t * ln2_hi is exact here */
            lo = t * ln2_lo;
        }
        x = hi - lo;
        c = (hi - x) - lo;
    } else if (absx < tiny) {
        return x;
    }
    /* x is now in the primary range */
    const double hfx = 0.5 * x;
    const double hxs = x * hfx;
    const double r1 = 1.0 + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
    double t = 3.0 - r1 * hfx;
    double e = hxs * ((r1 - t) / (6.0 - x * t));
    if (k == 0) return x - (x * e - hxs);
    e = x * (e - c) - c;
    e -= hxs;
    if (k == -1) return 0.5 * (x - e) - 0.5;
    if (k == 1) {
        if (x < -0.25) return -2.0 * (e - (x + 0.5));
        return 1.0 + 2.0 * (x - e);
    }
    if (k <= -2 || k > 56) { /* suffices to return exp(x) - 1 */
        double y = 1.0 - (e - x);
        y = pm_from_bits(pm_bits(y) + ((uint64_t)(int64_t)k << 52)); /* add k to y's exponent */
        return y - 1.0;
    }
    if (k < 20) {
        t = pm_from_bits(0x3ff0000000000000ull - (0x0020000000000000ull >> k)); /* 1 - 2^-k */
        double y = t - (e - x);
        return pm_from_bits(pm_bits(y) + ((uint64_t)k << 52));
    }
    t = pm_from_bits((uint64_t)(0x3ff - k) << 52); /* 2^-k */
    double y = x - (e + t);
    y += 1.0;
    return pm_from_bits(pm_bits(y) + ((uint64_t)k << 52));
}

/* hyperbolic tangent.  One expm1 evaluation and one division serve both ranges (lanes of a GPU wave that fall on
 * different sides of |x| = 1 would otherwise run the expm1 code twice); per argument the operations and their order are
 * exactly those of the two-branch form:  |x| >= 1: 1 - 2 / (expm1(2|x|) + 2),  below: -t / (t + 2) with t = expm1(-2|x|). */
PM_FN double pm_tanh(double x) {
    if (x != x) return x;
    const uint64_t u = pm_bits(x);
    const uint64_t a = u & 0x7fffffffffffffffull;
    const double ax = pm_from_bits(a);
    if (a < 0x3c80000000000000ull) return x; /* |x| < 2^-55 (also +-0) */
    double z = 1.0;                          /* +-inf, and |x| >= 22: 1 - tiny rounds to 1 */
    if (ax < 22.0) {
        const int big = ax >= 1.0;
        const double t = pm_expm1(big ? 2.0 * ax : -2.0 * ax);
        const double q = (big ? 2.0 : t) / (t + 2.0);
        z = big ? 1.0 - q : -q;
    }
    return (u >> 63) ? -z : z;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Round 4: the check update with TWO divisions per edge instead of four, in the reference's own operation order.
 *
 *   t_j = tanh(b2c_j / 2)                -- pm_tanh_half: exp(-|x|) from a division-free polynomial, then ONE division
 *                                           ((1 - e) / (1 + e) below |x| = 2, 1 - 2 e / (1 + e) above: the same two branches
 *                                           and the same error level as pm_tanh);
 *   X = prod t_j,  A = 1 + X,  B = 1 - X -- formed and rounded exactly as the reference forms them;
 *   log(A / B)                           -- pm_log_quot: the quotient is folded into the logarithm's own reduction,
 *                                           log(A / B) = k ln2 + 2 atanh(s),  s = (ma - mb) / (ma + mb)  with A = 2^ka ma,
 *                                           B = 2^kb mb: ONE division instead of two.
 * (A form with ONE division -- tanh kept as a fraction n / d, numerator and denominator products apart,
 * log((PD + PN) / (PD - PN)) -- was built and measured first: 1.73 x faster than the four-division form, but it changes
 * WHERE the roundings fall in the ill-conditioned quantity 1 - X, and against the libm oracle twice as many shots of the
 * clipped configs[2] run changed an integer output, 18 % against 9 %.  Not kept: the reference's operation order is part
 * of the parity contract.)
 * Saturation as in the reference formula: tanh(x / 2) rounds to +-1 from |x| = 38.2 on, B = 0 gives +inf, A = 0 gives
 * -inf, NaN propagates.
 */
PM_FN double pm_tanh_half(double x) {
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    if (x != x) return x;
    const uint64_t u = pm_bits(x);
    const uint64_t ua = u & 0x7fffffffffffffffull;
    const double a = pm_from_bits(ua);
    if (ua < 0x3c90000000000000ull) return 0.5 * x; /* |x| < 2^-54 (also +-0): tanh(x / 2) = x / 2 */
    double z = 1.0;                                  /* |x| >= 40 (and +-inf) */
    if (a < 40.0) {
        /* |x| = k ln2 + r, |r| <= ln2 / 2;  exp(-|x|) = 2^-k exp(y), y = -r */
        const int k = (int)(a * invln2 + 0.5);
        const double t = (double)k;
        const double y = t * ln2_lo - (a - t * ln2_hi); /* (t * ln2_hi is exact) */
        /* q = expm1(y) = y + y^2 (1/2 + y/6 + ... + y^11/13!): truncation below 4e-18 relative for |y| <= 0.3466 */
        double p = 1.6059043836821613e-10;       /* 1/13! */
        p = 2.0876756987868100e-09 + y * p;      /* 1/12! */
        p = 2.5052108385441720e-08 + y * p;      /* 1/11! */
        p = 2.7557319223985888e-07 + y * p;      /* 1/10! */
        p = 2.7557319223985893e-06 + y * p;      /* 1/9!  */
        p = 2.4801587301587302e-05 + y * p;      /* 1/8!  */
        p = 1.9841269841269841e-04 + y * p;      /* 1/7!  */
        p = 1.3888888888888889e-03 + y * p;      /* 1/6!  */
        p = 8.3333333333333332e-03 + y * p;      /* 1/5!  */
        p = 4.1666666666666664e-02 + y * p;      /* 1/4!  */
        p = 1.6666666666666666e-01 + y * p;      /* 1/3!  */
        p = 0.5 + y * p;
        const double q = y + (y * y) * p;
        if (a < 2.0) { /* (1 - e) / (1 + e) = -em1 / (2 + em1) with em1 = e - 1 */
            const double em1 = (k == 0) ? q : ((1.0 + q) * pm_from_bits((uint64_t)(1023 - k) << 52) - 1.0);
            z = (0.0 - em1) / (2.0 + em1);
        } else {       /* 1 - 2 e / (1 + e) */
            const double e = (1.0 + q) * pm_from_bits((uint64_t)(1023 - k) << 52); /* k <= 58 */
            z = 1.0 - (e + e) / (1.0 + e);
        }
    }
    return (u >> 63) ? -z : z;
}

/* log(A / B) for A = 1 + X, B = 1 - X, |X| <= 1 (or NaN): one division */
PM_FN double pm_log_quot(double A, double B) {
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                 L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    if (A != A || B != B) return A + B;  /* NaN */
    if (!(B > 0.0)) return (A > 0.0) ? 1.0 / 0.0 : (A - A) / (B - B); /* X = 1: log(2 / 0) = +inf;  0 / 0: NaN */
    if (!(A > 0.0)) return -1.0 / 0.0;                                  /* X = -1: log(0 / 2) */
    uint64_t ua = pm_bits(A), ub = pm_bits(B);
    int k = 0;
    if (ua < 0x0010000000000000ull) { /* subnormal (cannot arise from 1 +- X; kept for the function's own sake) */
        ua = pm_bits(A * 18014398509481984.0);
        k -= 54;
    }
    if (ub < 0x0010000000000000ull) {
        ub = pm_bits(B * 18014398509481984.0);
        k += 54;
    }
    k += (int)(ua >> 52) - (int)(ub >> 52);
    double ma = pm_from_bits((ua & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    double mb = pm_from_bits((ub & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    /* ma / mb in (1/2, 2): fold into [sqrt(1/2), sqrt(2)] */
    if (ma > mb * 1.4142135623730951) {
        mb = mb * 2.0;
        k += 1;
    } else if (ma * 1.4142135623730951 < mb) {
        ma = ma * 2.0;
        k -= 1;
    }
    const double s = (ma - mb) / (ma + mb); /* the numerator is exact; |s| <= 0.1716 */
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (L2 + w * (L4 + w * L6));
    const double t2 = z * (L1 + w * (L3 + w * (L5 + w * L7)));
    const double R = t2 + t1; /* log(ma / mb) = 2 s + s R */
    const double dk = (double)k;
    return dk * ln2_hi + (2.0 * s + (s * R + dk * ln2_lo));
}


/* The product-sum check update in its two evaluation orders (C-ABI bposd_config.ps_math_form, oracle ps_math 2 / 1):
 *   form 0, "reference order": tanh(b2c / 2) as fdlibm evaluates it (a division inside expm1, one in the tanh quotient) and
 *           log of the rounded quotient (1 + x) / (1 - x) (the quotient, then a division inside the logarithm's reduction) --
 *           four divisions per edge, the operation order of the reference's formula; the default: against the platform libm
 *           188 of 2048 clipped configs[2] shots differ in an integer output;
 *   form 1, "two divisions": pm_tanh_half / pm_log_quot above -- 1.4x the throughput on configs[2], 207 of 2048. */
PM_FN double pm_ps_tanh_half(double v, int form) { return form ? pm_tanh_half(v) : pm_tanh(v * 0.5); }
PM_FN double pm_ps_log_ratio(double x, int form) { return form ? pm_log_quot(1 + x, 1 - x) : pm_log((1 + x) / (1 - x)); }

#endif /* BPOSD_PORTABLE_MATH_H */
