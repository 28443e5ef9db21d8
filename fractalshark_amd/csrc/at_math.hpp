// at_math.hpp -- ATInfo::PerformAT, the approximation-term iteration in front of the LA stages (ATInfo.h:155-188):
// at_perform (device; the tuned wave-uniform loop the LAv2 kernels call) and at_perform_literal (host and device; the
// reference's loop operation by operation -- what at_perform falls back to and what the known-answer harness runs on the
// host).  Moved out of kernels.hip in round 4 so that tests/kat can call the kernels' own function.
#pragma once

#include "hdr_math.hpp"

namespace fs {

// The reference's loop, literally: i = number of iterations z = z*z + c taken before |z|^2 > SqrEscapeRadius.
template <class F, class IterT = uint32_t>
FS_HD void at_perform_literal(const hcplx<F> c, const hreal<F> esc, const IterT ATMaxIt, hcplx<F> &z_out, IterT &i_out)
{
    hcplx<F> z = hc_zero<F>();
    IterT i = 0;
    for (; i < ATMaxIt; i++) {
        hreal<F> nsq = hc_norm2(z);
        hr_reduce(nsq);
        if (hr_cmp_pos(nsq, esc) > 0)
            break;
        z = hc_add(hc_mul(z, z), c);
    }
    z_out = z;
    i_out = i;
}

#if defined(__HIPCC__)
// ATInfo::PerformAT (ATInfo.h:155-188): i = number of iterations z = z*z + c taken before |z|^2 > SqrEscapeRadius.
// The literal loop is HDRFloatComplex arithmetic with a 4-way exponent alignment per add and a Reduce + lexicographic
// compare per norm.  After the first iteration (0*0 + c = c) z carries c's exponent k, and for -120 < k <= 0 every later
// iteration takes the same alignment branch (z*z has exponent 2k, gap k to c): z' = (z*z) * 2^k + c with exponent k
// again.  The steady state below executes exactly those IEEE operations -- same operands, same order; re*im + im*re is
// computed as ri + ri, which is the same value -- on bare mantissas, and replaces Reduce + compare by a value comparison
// against T = esc.m * 2^(esc.e - 2k) (exact power-of-two scaling; +inf when it overflows), which is equivalent for a
// positive normal |z|^2.  Anything else (k outside the window, a zero / denormal norm) runs the literal loop.
template <class F, class IterT = uint32_t>
__device__ __forceinline__ void at_perform(const hcplx<F> c, const hreal<F> esc, const IterT ATMaxIt, hcplx<F> &z_out,
                                           IterT &i_out)
{
    hcplx<F> z = hc_zero<F>();
    IterT i = 0;
    const int k = c.e;
    if (k <= 0 && k > -kExpDiffIgnored && ATMaxIt > 1) {
        // iteration 0 literally: the norm of the zero start never exceeds the radius; z becomes c
        z = hc_add(hc_mul(z, z), c);
        i = 1;
        if (z.e == k) {
            F re = z.re, im = z.im;
            const F P = pow2_normal<F>(k);
            const int te = esc.e - 2 * k;
            F T;
            if (te >= fbits<F>::kMaxMulExp)
                T = type_max<F>() * F(2); // +inf: the scaled radius is >= 2^128 (2^1024), above every finite norm
            else
                T = esc.m * pow2_normal<F>(te < -fbits<F>::kBias + 2 ? -fbits<F>::kBias + 2 : te);
            const F min_normal = pow2_normal<F>(-fbits<F>::kBias + 1);
            bool literal = te < -fbits<F>::kBias + 2;
            // Loop shape.  The lanes that are in this loop all entered it at i = 1 and take one iteration per trip, so the
            // iteration number is ONE wave-uniform counter on the scalar unit.  A lane that is done (its norm left the
            // normal range or passed the radius) is NOT masked off: the wave keeps iterating all its lanes -- a finished
            // lane's z runs on into infinity or NaN, which nothing reads -- and the lane's state is recorded once, on the
            // trip it finishes (a wave-uniform branch on the vote of the lanes finishing now: at most 64 such trips per
            // wave against thousands of iterations).  Per iteration that leaves 8 arithmetic instructions, two compares and
            // a handful of scalar ones; the first form of this loop -- the two tests as divergent breaks, a per-lane
            // counter -- spent 17 scalar instructions per iteration on EXEC bookkeeping.
            if (!literal && i < ATMaxIt) {
                IterT it = 1;                                               // wave-uniform
                uint64_t pending = __builtin_amdgcn_ballot_w64(true);       // lanes still iterating (a lane mask, scalar)
                F xre = re, xim = im, xm = min_normal;
                IterT xi = ATMaxIt;
#define FS_AT_LOOP(SCALE)                                                                                           \
    for (;;) {                                                                                                      \
        const F rr = re * re, ii = im * im;                                                                         \
        const F m = rr + ii;                                                                                        \
        const uint64_t fin = (__builtin_amdgcn_ballot_w64(!(m >= min_normal)) | __builtin_amdgcn_ballot_w64(m > T)) \
                             & pending;                                                                             \
        if (fin != 0ull) {                                                                                          \
            if (__builtin_amdgcn_inverse_ballot_w64(fin))                                                           \
                xre = re, xim = im, xm = m, xi = it;                                                                \
            pending &= ~fin;                                                                                        \
            if (pending == 0ull)                                                                                    \
                break;                                                                                              \
        }                                                                                                           \
        const F ri = re * im;                                                                                       \
        re = (rr - ii) SCALE + c.re;                                                                                \
        im = (ri + ri) SCALE + c.im;                                                                                \
        if (++it >= ATMaxIt) {                                                                                      \
            if (__builtin_amdgcn_inverse_ballot_w64(pending))                                                       \
                xre = re, xim = im, xi = ATMaxIt; /* took its last iteration (xm stays normal: not literal) */      \
            break;                                                                                                  \
        }                                                                                                           \
    }
                if (__builtin_amdgcn_ballot_w64(k != 0) == 0ull) {
                    // every lane of the wave has 1 <= |c| < 2: P is 1 and x * 1 is x, bit for bit -- two multiplications less
                    FS_AT_LOOP()
                } else {
                    FS_AT_LOOP(*P)
                }
#undef FS_AT_LOOP
                re = xre, im = xim, i = xi;
                // finished through the norm test with a norm that is not a normal number: the literal loop continues from here
                literal = i < ATMaxIt && !(xm >= min_normal);
            }
            z = hcplx<F>{re, im, k};
            if (!literal) {
                z_out = z;
                i_out = i;
                return;
            }
        }
    }
    for (; i < ATMaxIt; i++) {
        hreal<F> nsq = hc_norm2(z);
        hr_reduce(nsq);
        if (hr_cmp_pos(nsq, esc) > 0)
            break;
        z = hc_add(hc_mul(z, z), c);
    }
    z_out = z;
    i_out = i;
}

#endif // __HIPCC__

} // namespace fs
