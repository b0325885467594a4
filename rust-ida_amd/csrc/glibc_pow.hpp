// pow(x, y) with the bits of glibc 2.35's pow on x86-64 with FMA (`__pow_fma`, the variant libm's ifunc resolver selects on
// every CPU that has FMA and AVX2 -- this container's and any MI355X host's).
//
// Why: the reference's step-size and order controller calls f64::powf = the platform's pow
// (/root/reference/src/lib.rs:1163-1169, src/impl_complete_step.rs:128-132, src/ida_nls.rs:249-253), and its result
// feeds h directly; a pow that is merely accurate to an ulp changes step sequences. The device-resident controller
// therefore needs glibc's result bit for bit. glibc's algorithm is Szabolcs Nagy's (ARM optimized-routines, MIT): table-driven
// log in double-double, product with y, table-driven exp. Which operations are fused is NOT visible in the C source -- glibc is
// built with -ffp-contract=fast, so gcc contracted some a*b+c into FMAs -- so this file restates the *machine code* of
// `__pow_fma` (objdump of /lib/x86_64-linux-gnu/libm.so.6, Ubuntu GLIBC 2.35-0ubuntu3.11, function at file offset 0x768b0):
// one line of C per arithmetic instruction, every vfmadd written as __builtin_fma, every vmulsd/vaddsd as a plain product or
// sum, in a translation unit compiled with -ffp-contract=off. The two lookup tables are read from the same libm by
// tools/extract_glibc_pow_tables.py (glibc_pow_tables.hpp).
//
// Checked: tests/test_glibc_pow.py compares this function, compiled for the host, against libm's pow on 10^8 random
// arguments of the controller's domain and on the special cases (CPU), and the device build against the committed fixture
// tests/golden/glibc_pow_2p20.npz and the host's pow (GPU).
#pragma once
#include <cstdint>
#include <cstring>

#include "glibc_pow_tables.hpp"

#if defined(__HIPCC__)
#define GLIBC_POW_HD __host__ __device__
#else
#define GLIBC_POW_HD
#endif

namespace glibc_pow {

GLIBC_POW_HD inline uint64_t as_u64(double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
}
GLIBC_POW_HD inline double as_f64(uint64_t u) {
    double x;
    memcpy(&x, &u, 8);
    return x;
}

// table access: host code reads the host arrays, device code their __device__ copies
#if defined(__HIP_DEVICE_COMPILE__)
#define GLIBC_POW_LOG(i) as_f64(glibc_pow_tables::LOG_DATA_DEV[i])
#define GLIBC_POW_EXPD(i) as_f64(glibc_pow_tables::EXP_DATA_DEV[i])
#define GLIBC_POW_EXPU(i) (glibc_pow_tables::EXP_DATA_DEV[i])
#else
#define GLIBC_POW_LOG(i) as_f64(glibc_pow_tables::LOG_DATA[i])
#define GLIBC_POW_EXPD(i) as_f64(glibc_pow_tables::EXP_DATA[i])
#define GLIBC_POW_EXPU(i) (glibc_pow_tables::EXP_DATA[i])
#endif

// 0 = not an integer, 1 = odd, 2 = even
GLIBC_POW_HD inline int checkint(uint64_t iy) {
    const int e = (int)(iy >> 52) & 0x7ff;
    if (e < 0x3ff) return 0;
    if (e > 0x3ff + 52) return 2;
    if (iy & ((1ULL << (0x3ff + 52 - e)) - 1)) return 0;
    if (iy & (1ULL << (0x3ff + 52 - e))) return 1;
    return 2;
}
GLIBC_POW_HD inline bool zeroinfnan(uint64_t i) { return 2 * i - 1 >= 2 * 0x7ff0000000000000ULL - 1; }

GLIBC_POW_HD inline double pow(double x, double y) {
    uint32_t sign_bias = 0;
    uint64_t ix = as_u64(x);
    const uint64_t iy = as_u64(y);
    uint32_t topx = (uint32_t)(ix >> 52);
    const uint32_t topy = (uint32_t)(iy >> 52);
    if (topx - 0x001 >= 0x7ff - 0x001 || (topy & 0x7ff) - 0x3be >= 0x43e - 0x3be) {
        // special cases: x < 0x1p-1022, x infinite or NaN, x < 0; |y| < 0x1p-65, |y| >= 0x1p63, y NaN
        if (zeroinfnan(iy)) {
            if (2 * iy == 0) return 1.0;  // (signalling NaNs are not distinguished: x + y and 1.0 have the same bits for quiet arguments)
            if (ix == 0x3ff0000000000000ULL) return 1.0;
            if (2 * ix > 2 * 0x7ff0000000000000ULL || 2 * iy > 2 * 0x7ff0000000000000ULL) return x + y;
            if (2 * ix == 2 * 0x3ff0000000000000ULL) return 1.0;
            if ((2 * ix < 2 * 0x3ff0000000000000ULL) == !(iy >> 63)) return 0.0;  // |x| < 1 && y == inf, or |x| > 1 && y == -inf
            return y * y;
        }
        if (zeroinfnan(ix)) {
            double x2 = x * x;
            if ((ix >> 63) && checkint(iy) == 1) x2 = -x2;
            return (iy >> 63) ? 1.0 / x2 : x2;  // (pow(+-0, y < 0): the division gives the infinity __math_divzero returns)
        }
        if (ix >> 63) {  // finite x < 0
            const int yint = checkint(iy);
            if (yint == 0) return (x - x) / (x - x);  // __math_invalid: NaN
            if (yint == 1) sign_bias = 0x800u << 7;
            ix &= 0x7fffffffffffffffULL;
            topx &= 0x7ff;
        }
        if ((topy & 0x7ff) - 0x3be >= 0x43e - 0x3be) {
            if (ix == 0x3ff0000000000000ULL) return 1.0;
            if ((topy & 0x7ff) < 0x3be) return ix > 0x3ff0000000000000ULL ? 1.0 + y : 1.0 - y;  // |y| < 2^-65
            return (ix > 0x3ff0000000000000ULL) == (topy < 0x800) ? as_f64(0x7ff0000000000000ULL) /* overflow */ : 0.0 /* underflow */;
        }
        if (topx == 0) {  // subnormal x: normalised, the exponent becomes negative
            ix = as_u64(x * 0x1p52);
            ix &= 0x7fffffffffffffffULL;
            ix -= 52ULL << 52;
        }
    }
    // ---- log_inline: hi + lo ~ log(x), instruction for instruction (0x768fb .. 0x769f8)
    const uint64_t tmp = ix - 0x3fe6955500000000ULL;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & 0xfff0000000000000ULL);
    const double z = as_f64(iz);
    const double kd = (double)k;
    const double invc = GLIBC_POW_LOG(9 + 4 * i), logc = GLIBC_POW_LOG(9 + 4 * i + 2), logctail = GLIBC_POW_LOG(9 + 4 * i + 3);
    const double ln2hi = GLIBC_POW_LOG(0), ln2lo = GLIBC_POW_LOG(1);
    const double A0 = GLIBC_POW_LOG(2), A1 = GLIBC_POW_LOG(3), A2 = GLIBC_POW_LOG(4), A3 = GLIBC_POW_LOG(5), A4 = GLIBC_POW_LOG(6),
                 A5 = GLIBC_POW_LOG(7), A6 = GLIBC_POW_LOG(8);
    const double t1 = __builtin_fma(kd, ln2hi, logc);      // vfmadd213sd 0x18(%rdx),%xmm2,%xmm5
    const double r = __builtin_fma(z, invc, -1.0);         // vfmadd132sd 0x8(%rdx),%xmm3,%xmm0
    const double ar = r * A0;                              // vmulsd A[0]
    const double lo1 = __builtin_fma(kd, ln2lo, logctail); // vfmadd213sd 0x20(%rdx),%xmm2,%xmm4
    const double p12 = __builtin_fma(r, A2, A1);           // A[1] + r*A[2]
    const double p34 = __builtin_fma(r, A4, A3);           // A[3] + r*A[4]
    const double t2 = r + t1;                              // vaddsd %xmm5,%xmm0,%xmm6
    const double ar2 = r * ar;                             // vmulsd %xmm9,%xmm0,%xmm2
    const double d12 = t1 - t2;                            // vsubsd %xmm6,%xmm5,%xmm5
    const double ar3 = r * ar2;                            // vmulsd %xmm2,%xmm0,%xmm7
    const double lo3 = __builtin_fma(ar, r, -ar2);         // vfmsub132sd %xmm0,%xmm2,%xmm9
    const double lo2 = d12 + r;                            // vaddsd %xmm0,%xmm5,%xmm8
    const double p56 = __builtin_fma(r, A6, A5);           // A[5] + r*A[6]
    const double hi0 = t2 + ar2;                           // vaddsd %xmm2,%xmm6,%xmm5
    const double d2h = t2 - hi0;                           // vsubsd %xmm5,%xmm6,%xmm6
    const double q1 = __builtin_fma(p56, ar2, p34);        // vfmadd132sd %xmm2,%xmm11,%xmm0
    const double lo4 = d2h + ar2;                          // vaddsd %xmm2,%xmm6,%xmm6
    const double q2 = __builtin_fma(ar2, q1, p12);         // vfmadd132sd %xmm0,%xmm10,%xmm2
    double los = lo1 + lo2;                                // vaddsd %xmm8,%xmm4,%xmm0
    los = los + lo3;                                       // vaddsd %xmm9,%xmm0,%xmm0
    los = los + lo4;                                       // vaddsd %xmm6,%xmm0,%xmm0
    const double lo = __builtin_fma(ar3, q2, los);         // vfmadd231sd %xmm2,%xmm7,%xmm0
    const double hi = hi0 + lo;                            // vaddsd %xmm0,%xmm5,%xmm4
    const double lotail = (hi0 - hi) + lo;                 // vsubsd; vaddsd
    // ---- y * log(x) in double-double (0x769fc .. 0x76a1f)
    const double ehi = y * hi;
    const double ehl = __builtin_fma(hi, y, -ehi);         // vfmsub132sd %xmm1,%xmm0,%xmm4
    const double elo = __builtin_fma(y, lotail, ehl);      // vfmadd132sd %xmm2,%xmm4,%xmm1
    // ---- exp_inline(ehi, elo, sign_bias) (0x76a2d .. 0x76acb)
    uint32_t abstop = (uint32_t)(as_u64(ehi) >> 52) & 0x7ff;
    if (abstop - 0x3c9 >= 0x3f) {
        if (abstop - 0x3c9 >= 0x80000000u) {  // tiny: 1 + x
            const double one = 1.0 + ehi;
            return sign_bias ? -one : one;
        }
        if (abstop >= 0x409) {  // |y log x| >= 1024
            const double inf = as_f64(0x7ff0000000000000ULL);
            if (as_u64(ehi) >> 63) return sign_bias ? -0.0 : 0.0;  // __math_uflow
            return sign_bias ? -inf : inf;                          // __math_oflow
        }
        abstop = 0;  // large: handled after the polynomial
    }
    const double InvLn2N = GLIBC_POW_EXPD(0), Shift = GLIBC_POW_EXPD(1), NegLn2hiN = GLIBC_POW_EXPD(2), NegLn2loN = GLIBC_POW_EXPD(3);
    const double C2 = GLIBC_POW_EXPD(4), C3 = GLIBC_POW_EXPD(5), C4 = GLIBC_POW_EXPD(6), C5 = GLIBC_POW_EXPD(7);
    const double kds = __builtin_fma(ehi, InvLn2N, Shift);  // z + Shift, contracted
    const uint64_t ki = as_u64(kds);
    const double kde = kds - Shift;
    const double r0 = __builtin_fma(kde, NegLn2hiN, ehi);   // vfmadd231sd
    const double r1 = __builtin_fma(kde, NegLn2loN, r0);    // vfmadd132sd
    const double rr = elo + r1;                             // r += xtail
    const int idx = 2 * (int)(ki & 127);
    const uint64_t top = (ki + sign_bias) << 45;
    const double tail = GLIBC_POW_EXPD(14 + idx);
    uint64_t sbits = GLIBC_POW_EXPU(14 + idx + 1) + top;
    const double c23 = __builtin_fma(rr, C3, C2);           // C2 + r*C3
    const double tr = rr + tail;                            // tail + r
    const double r2 = rr * rr;
    const double c45 = __builtin_fma(rr, C5, C4);           // C4 + r*C5
    const double s1 = __builtin_fma(c23, r2, tr);           // vfmadd132sd %xmm0,%xmm4,%xmm2
    const double r4 = r2 * r2;
    const double tmpv = __builtin_fma(c45, r4, s1);         // vfmadd132sd %xmm0,%xmm2,%xmm1
    if (abstop == 0) {
        // specialcase(): the scale 2^(k/N) alone would over- or underflow
        if ((ki & 0x80000000ULL) == 0) {  // k > 0
            sbits -= 1009ULL << 52;
            const double scale = as_f64(sbits);
            return 0x1p1009 * __builtin_fma(scale, tmpv, scale);
        }
        sbits += 1022ULL << 52;
        const double scale = as_f64(sbits);
        const double st = scale * tmpv;
        double yv = scale + st;
        if ((yv < 0.0 ? -yv : yv) < 1.0) {
            const double one = yv < 0.0 ? -1.0 : 1.0;
            double lo_ = (scale - yv) + st;
            const double hi_ = one + yv;
            lo_ = ((one - hi_) + yv) + lo_;
            yv = (hi_ + lo_) - one;
            if (yv == 0.0) yv = as_f64(sbits & 0x8000000000000000ULL);
        }
        return 0x1p-1022 * yv;
    }
    const double scale = as_f64(sbits);
    return __builtin_fma(tmpv, scale, scale);               // scale + scale*tmp, contracted
}

}  // namespace glibc_pow
