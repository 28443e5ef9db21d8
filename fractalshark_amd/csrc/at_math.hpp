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
                                           IterT &i_out, IterT *executed = nullptr, IterT *own_cost = nullptr)
{
    hcplx<F> z = hc_zero<F>();
    IterT i = 0;
    IterT skipped = 0; // iterations the cycle search spared this lane (counting builds: executed = i_out - skipped)
    IterT own = 0;     // != 0: what this lane alone would have run (where ITS cycle showed + its remainder), for the pixel order
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
            // Round 4, F = double: the loop in TRUE values.  With Z = z 2^k (the value z represents) and C = c 2^k the
            // iteration z' = (z*z) 2^k + c is Z' = Z*Z + C: a power-of-two scaling commutes with every IEEE operation
            // that neither underflows nor overflows, so the same bits come out with the two multiplications by P gone
            // (10 instead of 12 vector instructions per iteration).  What can differ is an underflow in ONE of the two
            // domains: a product below 2^-1022 in true values (rr, ii, ri), or (z*z) 2^k below it in the reference's.
            // rr / ii: the iteration is only taken while |Z|^2 >= 2^-782 (`floor_t`; the reference's own norm is then
            // normal too), so an underflowed square is 2^239 below the other one and absorbed by both sums that read
            // it.  ri: added to C.im, which absorbs it when |c.im| >= 2^-800 (the ratio is the same in both domains);
            // c.im == 0 keeps im at exactly 0 (z starts at 0).  Hence the wave-uniform precondition below; a wave with
            // a lane outside it (a pixel within 2^-800 of the real axis in relative terms, an escape radius outside
            // 2^-700 .. 2^900) takes the loop in the reference's units.  Below the floor the literal loop continues, as before.
            bool true_domain = false;
            F thr = min_normal, cre = c.re, cim = c.im;
            if constexpr (sizeof(F) == 8) {
                const F acim = c.im < F(0) ? -c.im : c.im;
                const bool lane_ok = !literal && te < fbits<F>::kMaxMulExp && (c.im == F(0) || acim >= pow2_normal<F>(-800)) &&
                                     esc.e > -700 && esc.e < 900;
                if (__builtin_amdgcn_ballot_w64(!lane_ok) == 0ull) {
                    true_domain = true;
                    re *= P, im *= P, cre *= P, cim *= P; // exact: |parts| are 0 or >= 2^-920
                    T = esc.m * pow2_normal<F>(esc.e);
                    thr = pow2_normal<F>(-782);
                }
            }
            if (!literal && i < ATMaxIt) {
                IterT it = 1;                                               // wave-uniform
                uint64_t pending = __builtin_amdgcn_ballot_w64(true);       // lanes still iterating (a lane mask, scalar)
                F xre = re, xim = im, xm = thr;
                IterT xi = ATMaxIt;
                // (cycle search of FS_AT_LOOP: the kept state as bit patterns -- all ones is a NaN no state equals)
                constexpr uint32_t kAtLoopChunk = 16u;
                typename fbits<F>::U g_sre = ~(typename fbits<F>::U)0, g_sim = ~(typename fbits<F>::U)0;
                IterT g_sit = 0, g_snext = (IterT)kAtLoopChunk, g_cyc_p = 0, g_cyc_at = 0;
#define FS_AT_LOOP(SCALE)                                                                                           \
    for (;;) {                                                                                                      \
        const F rr = re * re, ii = im * im;                                                                         \
        const F m = rr + ii;                                                                                        \
        const uint64_t fin = (__builtin_amdgcn_ballot_w64(!(m >= thr)) | __builtin_amdgcn_ballot_w64(m > T))      \
                             & pending;                                                                             \
        if (fin != 0ull) {                                                                                          \
            if (__builtin_amdgcn_inverse_ballot_w64(fin))                                                           \
                xre = re, xim = im, xm = m, xi = it;                                                                \
            pending &= ~fin;                                                                                        \
            if (pending == 0ull)                                                                                    \
                break;                                                                                              \
        }                                                                                                           \
        const F ri = re * im;                                                                                       \
        re = (rr - ii) SCALE + cre;                                                                                 \
        im = (ri + ri) SCALE + cim;                                                                                 \
        if (++it >= ATMaxIt) {                                                                                      \
            if (__builtin_amdgcn_inverse_ballot_w64(pending))                                                       \
                xre = re, xim = im, xi = ATMaxIt; /* took its last iteration (xm stays normal: not literal) */      \
            break;                                                                                                  \
        }                                                                                                           \
        if ((it & (IterT)(kAtLoopChunk - 1u)) == 0) { /* the cycle search (see the hand-written loop below) */       \
            if (g_cyc_p == 0) {                                                                                     \
                if (to_bits<F>(re) == g_sre && to_bits<F>(im) == g_sim) {                                           \
                    g_cyc_p = it - g_sit;                                                                           \
                    g_cyc_at = it;                                                                                  \
                } else if (it >= g_snext) {                                                                         \
                    g_sre = to_bits<F>(re), g_sim = to_bits<F>(im), g_sit = it, g_snext = it + it;                   \
                }                                                                                                   \
            }                                                                                                       \
            if ((__builtin_amdgcn_ballot_w64(g_cyc_p == 0) & pending) == 0ull) {                                    \
                /* every lane still iterating is on its cycle: each walks its remainder, in this loop's arithmetic */ \
                if (__builtin_amdgcn_inverse_ballot_w64(pending)) {                                                 \
                    IterT r_ = (ATMaxIt - it) % g_cyc_p;                                                            \
                    skipped = ATMaxIt - it - r_;                                                                    \
                    own = g_cyc_at + (ATMaxIt - g_cyc_at) % g_cyc_p;                                                \
                    for (; r_ != 0; r_--) {                                                                         \
                        const F rr_ = re * re, ii_ = im * im, ri_ = re * im;                                        \
                        re = (rr_ - ii_) SCALE + cre;                                                               \
                        im = (ri_ + ri_) SCALE + cim;                                                               \
                    }                                                                                               \
                    xre = re, xim = im, xi = ATMaxIt;                                                               \
                }                                                                                                   \
                break;                                                                                              \
            }                                                                                                       \
        }                                                                                                           \
    }
                if constexpr (sizeof(F) == 8) {
                    if (true_domain) {
                        // The true-value loop by hand: what the compiler makes of FS_AT_LOOP() is 10 vector and 11 scalar
                        // instructions per iteration, and the scalar unit (shared by the CU's four SIMDs) is then as busy as
                        // the vector pipes.  Here: seven arithmetic operations -- (ri + ri) + C.im is ONE fused multiply-add by 2:
                        // the doubling is exact, so the single rounding of the fma is the rounding of the sum --, ONE test -- the high word of |Z|^2 inside
                        // [high word of the floor, high word of T): two 32-bit instructions that flag a superset of the
                        // lanes that finish (equal high words are decided exactly, outside) -- and four scalar instructions
                        // (vote, branch, count, branch).  The statement leaves when a pending lane is flagged, with the state
                        // NOT advanced, or after its iteration budget; recording, the exact test and that one iteration are
                        // C++ (a lane finishes once, a wave leaves the statement a few dozen times in thousands of iterations).
                        const uint32_t thr_hi = (uint32_t)(to_bits<F>(thr) >> 32);
                        const uint32_t range = (uint32_t)(to_bits<F>(T) >> 32) - thr_hi; // (T >= 2^-700 > floor: precondition)
                        // CYCLE SEARCH (round 5).  The loop is a pure function Z -> Z*Z + C of the lane's state, so a state
                        // that comes back bit for bit means the lane repeats itself for good: it will never finish through
                        // the norm test (it has just gone once round its cycle without doing so), and its state after ATMaxIt
                        // iterations is the state (ATMaxIt - it) mod P iterations further round the cycle, P = the distance
                        // of the two equal states.  A pixel inside the set is exactly that: its AT orbit converges to an
                        // attracting cycle and, in binary64, locks into it -- View 14: after 320 iterations (median; 9 267 for
                        // the slowest percent) of the 18 402 the iteration limit asks of it, and those pixels are 9 % of the
                        // frame and 97 % of its AT iterations (tools/at_cycle_potential.py).  Brent's scheme, sampled where the
                        // statement's budget ends (every kAtCycleChunk = 16 iterations): compare with a kept state, keep a new one
                        // at doubling distances.  When every lane still iterating has found its cycle the wave leaves the loop
                        // and each lane walks its remainder.  Same states, same iteration count, same results.
#ifndef FS_AT_CYCLE_CHUNK
#define FS_AT_CYCLE_CHUNK 16 /* measured on C4 (ms per frame): 8: 46.0, 16: 45.8, 32: 46.1, 64: 46.7, 128: 48.1, 256: 51.3 */
#endif
                        constexpr uint32_t kAtCycleChunk = FS_AT_CYCLE_CHUNK;
                        uint64_t s_re = ~0ull, s_im = ~0ull; // the kept state (bit patterns; all ones = a NaN no state equals)
                        IterT s_it = 0, s_next = (IterT)kAtCycleChunk, cyc_p = 0, cyc_at = 0; // cyc_p != 0: this lane has found its cycle (at iteration cyc_at)
                        bool all_cyclic = false;
                        for (;;) {
                            const IterT left = ATMaxIt - it; // >= 1
                            const uint32_t n = left > (IterT)kAtCycleChunk ? kAtCycleChunk : (uint32_t)left;
                            uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n - 1u));
                            F rr, ii, ri, m;
                            uint32_t t_;
                            uint64_t tmp_;
                            int st;
                            asm volatile("s_mov_b32 %[st], 0\n"
                                         ".Lat_loop_%=:\n\t"
                                         "v_mul_f64 %[rr], %[re], %[re]\n\t"
                                         "v_mul_f64 %[ii], %[im], %[im]\n\t"
                                         "v_mul_f64 %[ri], %[re], %[im]\n\t"
                                         "v_add_f64 v[62:63], %[rr], %[ii]\n\t"
                                         "v_add_f64 %[rr], %[rr], -%[ii]\n\t"
                                         "v_subrev_u32_e32 %[t], %[thi], v63\n\t"
                                         "v_cmp_le_u32_e32 vcc, %[rng], %[t]\n\t"
                                         "s_and_b64 %[tmp], vcc, %[pend]\n\t"
                                         "s_cbranch_scc1 .Lat_flag_%=\n\t"
                                         "v_add_f64 %[re], %[rr], %[cre]\n\t"
                                         "v_fma_f64 %[im], %[ri], 2.0, %[cim]\n\t" /* (ri + ri) + cim: doubling is exact */
                                         "s_sub_u32 %[cnt], %[cnt], 1\n\t"
                                         "s_cbranch_scc0 .Lat_loop_%=\n\t"
                                         "s_branch .Lat_end_%=\n"
                                         ".Lat_flag_%=:\n\t"
                                         "s_mov_b32 %[st], 1\n"
                                         ".Lat_end_%=:"
                                         : [re] "+v"(re), [im] "+v"(im), [rr] "=&v"(rr), [ii] "=&v"(ii), [ri] "=&v"(ri),
                                           "={v[62:63]}"(m), [t] "=&v"(t_), [tmp] "=&s"(tmp_), [st] "=&s"(st), [cnt] "+s"(cnt)
                                         : [cre] "v"(cre), [cim] "v"(cim), [rng] "v"(range), [pend] "s"(pending),
                                           [thi] "s"(thr_hi)
                                         : "vcc", "scc");
                            st = __builtin_amdgcn_readfirstlane(st);
                            cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
                            if (st == 0) {
                                it += (IterT)n;
                                if (it >= ATMaxIt) {
                                    if (__builtin_amdgcn_inverse_ballot_w64(pending))
                                        xre = re, xim = im, xi = ATMaxIt;
                                    break;
                                }
                                // (the cycle search, see above)
                                const uint64_t b_re = to_bits<F>(re), b_im = to_bits<F>(im);
                                if (cyc_p == 0) {
                                    if (b_re == s_re && b_im == s_im) {
                                        cyc_p = it - s_it;
                                        cyc_at = it;
                                    } else if (it >= s_next) {
                                        s_re = b_re, s_im = b_im, s_it = it;
                                        s_next = it + it; // (it <= ATMaxIt / 2 matters only: beyond it no cycle can pay)
                                    }
                                }
                                if ((__builtin_amdgcn_ballot_w64(cyc_p == 0) & pending) == 0ull) {
                                    all_cyclic = true;
                                    break;
                                }
                                continue;
                            }
                            it += (IterT)(n - 1u - cnt);
                            // the state is at iteration `it`, its norm in m, rr - ii and re * im in rr / ri: the exact test
                            const uint64_t fin =
                                (__builtin_amdgcn_ballot_w64(!(m >= thr)) | __builtin_amdgcn_ballot_w64(m > T)) & pending;
                            if (fin != 0ull) {
                                if (__builtin_amdgcn_inverse_ballot_w64(fin))
                                    xre = re, xim = im, xm = m, xi = it;
                                pending &= ~fin;
                                if (pending == 0ull)
                                    break;
                            }
                            re = rr + cre;
                            im = (ri + ri) + cim;
                            if (++it >= ATMaxIt) {
                                if (__builtin_amdgcn_inverse_ballot_w64(pending))
                                    xre = re, xim = im, xi = ATMaxIt;
                                break;
                            }
                        }
                        if (all_cyclic) {
                            // every lane still iterating is on its cycle at iteration `it`: the state after ATMaxIt
                            // iterations is (ATMaxIt - it) mod P iterations ahead -- the statement's own operations
                            if (__builtin_amdgcn_inverse_ballot_w64(pending)) {
                                IterT r = (ATMaxIt - it) % cyc_p;
                                skipped = ATMaxIt - it - r;
                                own = cyc_at + (ATMaxIt - cyc_at) % cyc_p;
                                for (; r != 0; r--) {
                                    const F rr = re * re, ii = im * im, ri = re * im;
                                    re = (rr - ii) + cre;
                                    im = (ri + ri) + cim;
                                }
                                xre = re, xim = im, xi = ATMaxIt;
                            }
                        }
                    }
                }
                if (true_domain) {
                    // (done above)
                } else if (__builtin_amdgcn_ballot_w64(k != 0) == 0ull) {
                    // every lane of the wave has 1 <= |c| < 2: P is 1 and x * 1 is x, bit for bit -- two multiplications less
                    FS_AT_LOOP()
                } else {
                    FS_AT_LOOP(*P)
                }
#undef FS_AT_LOOP
                re = xre, im = xim, i = xi;
                // finished through the norm test with a norm below the floor (not a normal number in the reference's units,
                // or too close to it for the true-value loop): the literal loop continues from here
                literal = i < ATMaxIt && !(xm >= thr);
            }
            if (true_domain) {
                const F Pinv = pow2_normal<F>(-k); // back to the reference's units: exact
                re *= Pinv, im *= Pinv;
            }
            z = hcplx<F>{re, im, k};
            if (!literal) {
                z_out = z;
                i_out = i;
                if (executed)
                    *executed = i - skipped;
                if (own_cost)
                    *own_cost = own != 0 ? own : i - skipped;
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
    if (executed)
        *executed = i - skipped;
    if (own_cost)
        *own_cost = own != 0 ? own : i - skipped;
}

#endif // __HIPCC__

} // namespace fs
