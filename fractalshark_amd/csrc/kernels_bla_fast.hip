// kernels_bla_fast.hip -- the HDRFloat<float> perturbation + BLA kernel (GpuHDRx32PerturbedBLA <-> Cpu32PerturbedBLAHDR,
// Fractal.cpp:2266-2470; BLAKernels.cuh:193-434; BLAS::LookupBackwards, BLAS.cpp:256-310) with its hot loop written by hand.
//
// Round 5.  The compiled kernel (k_perturb_scalar<float, kBla, kNat>, kernels_perturb.hip) issues 329 vector and 133 scalar
// instructions per outer trip of a wave on C5 (rocprofv3, profiles/r05a_c5_*) and is bound by vector issue; a third of
// them are the structurizer's copies between the exits of its two nested loops, exponent alignment through factor +
// multiply, votes lowered to v_cndmask + v_cmp, and 64-bit level pointers.  Here the per-pixel state lives in sixteen named
// vector registers across ONE asm statement that holds the whole loop:
//
//   lookup   vote "some lane sits at an index = 1 (mod 4)"; one 16-byte load of Q[(m - 1) / 4] = {pre-test key, heap position of
//            the walk's first element, its level}; one compare; ladder rounds of two 16-byte loads and four 64-bit compares.
//            The table is numbered like a binary heap -- element ix of level L sits at 2^(H - L) + ix -- so the element one
//            level down is at twice the position: no level offsets, no LDS, no count-trailing-zeros per lookup.
//   jump     BLA::getValue in the alignment-free form of kernels_perturb.hip (every product scaled to the maximum of the four
//            exponents with v_ldexp_f32 on the product itself), the quiet form only (both parts of the new dz four binades
//            below the orbit value it arrives at: 99.95 % of C5's jumps, profiles/r05c_c5_actions.json); what a record must
//            satisfy for it is decided when the table is made and poisons the record's arrival exponent.
//   step     the straight-line step; quiet form (67 % of C5's steps) or with z = Z + dz, the two norms, the escape and the
//            rebase test (33 %): an escaping lane just leaves the wave's running mask with its count, a rebasing lane
//            rewrites its dz under EXEC.
//
// Everything else -- a lane that needs the reference's operation order (exact zeros, sums outside 2^+-30), a jump that is
// not quiet, a walk that starts at an element the heap numbering cannot name -- leaves the statement with the wave's state
// intact and a status; the C++ around it performs that ONE action for the lanes concerned in the literal order of
// Fractal.cpp (hdr_math.hpp) and re-enters.  Acceptance tests are those of the compiled kernel's fast forms (never weaker),
// so the set of pixels and their counts are the same: tests/test_gpu_parity.py, test_gpu_goldens.py (the reference's own
// CRC-64s), the 25-view sweep and the oracle frame CRCs run through this kernel by default.
//
// The step-counting build of the compiled kernel keeps serving fs_enable_step_count (the work per frame is a function of the
// inputs, not of the kernel), the 64-bit counting instantiation, the refill variant, and variant 2 (A/B reference).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <cstring>

#include "../../include/fs_layout.h"
#include "hdr_math.hpp"
#include "kernels.h"
#include "kernel_common.hpp"

using namespace fs;

namespace {

constexpr int32_t kPoisonExp = -(1 << 28);       // arrival exponent of a record / entry the fast forms must not use
constexpr int32_t kPoisonExpHigh = 1 << 27;      // true-exponent slot of an entry no step may arrive at (past the orbit)

struct HeapGeom {
    uint32_t level_off[kBlaMaxLevels];
    uint32_t level_n[kBlaMaxLevels];
    int32_t n_levels;
    int32_t H;      // heap position of element ix of level L = (1 << (H - L)) + ix
    uint32_t total; // old positions
};

// One thread per element of the native table (FsBlaRec + ladder, k_bla_make_native): its copy at the heap position.
// A record whose jump the quiet form may not take is marked in its arrival exponent: mantissas beyond 2^30 or an exponent at
// or below -2^26 (the compiled kernel tests both per jump), an arrival entry outside 2^-40 .. 2^1, an arrival at the orbit's
// last entry or beyond.  l = 0xFFFFFFFF for a jump that would leave the orbit (the reference then takes a step instead,
// exactly as when the jump would pass the iteration cap: the kernel's saturating iter + l covers both with one compare).
__global__ void __launch_bounds__(256) k_bla_make_heap(const FsBlaRec *__restrict__ rec, const int4 *__restrict__ lad, HeapGeom G,
                                                       uint32_t orbit_count, FsBlaRec *__restrict__ hrec, int4 *__restrict__ hlad)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= G.total)
        return;
    int32_t L = 2;
    for (int32_t l = 3; l < G.n_levels; l++)
        if (G.level_n[l] != 0u && p >= G.level_off[l])
            L = l;
    const uint32_t ix = p - G.level_off[L];
    const uint32_t h = (1u << (G.H - L)) + ix;
    FsBlaRec o = rec[p];
    const uint64_t arrive = ((uint64_t)ix << L) + 1u + (uint64_t)o.l;
    const float amx = fmaxf(fmaxf(fabsf(o.Axm), fabsf(o.Aym)), fmaxf(fabsf(o.Bxm), fabsf(o.Bym)));
    const int32_t emin = min(min(o.Axe, o.Aye), min(o.Bxe, o.Bye));
    const bool eligible = amx <= 0x1p30f && emin > -(1 << 26) && o.Ze <= 1 && o.Ze >= -40 && arrive + 1u < orbit_count;
    if (!eligible)
        o.Ze = kPoisonExp;
    if (arrive >= orbit_count)
        o.l = 0xFFFFFFFFu;
    hrec[h] = o;
    hlad[2 * (size_t)h] = lad[2 * (size_t)p];
    hlad[2 * (size_t)h + 1] = lad[2 * (size_t)p + 1];
}

// Q[q], one per orbit index m = 4 q + 1, 48 bytes: {pre-test key (k_bla_make_kmax; for q = 0 also bounded by the key of the first
// element of level 2, the gate of BLAS.cpp:270-281), heap position of the element the walk starts at, its level -- 0 when the
// walk cannot be expressed here (an element index outside its level: the compiled lookup serves)} and that element's ladder
// entry (the four keys of the walk's first round), so that a lookup that ends in its first round -- 15 of 16 -- reads nothing
// else before the record: the dependent chain of a lookup is Q -> record instead of key -> ladder -> record, and Q itself is
// requested by the action that arrives at the index (a step or a jump), well before the lookup needs it.
__global__ void __launch_bounds__(256) k_bla_make_q(const long long *__restrict__ kmax, const int4 *__restrict__ lad, HeapGeom G,
                                                    int32_t lm2, int4 *__restrict__ hq, uint32_t n)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n)
        return;
    const uint32_t k = q << 2;
    const int32_t zeros = k == 0u ? 32 : (int32_t)__ffs((int)k) - 1;
    const uint32_t ix = k == 0u ? 0u : k >> zeros;
    const int32_t L = zeros <= lm2 ? zeros : lm2;
    long long key = kmax[q];
    uint32_t pos = 0;
    int32_t lvl = 0;
    if (L < 2 || L >= G.n_levels) {
        key = (long long)0x8000000000000000ull; // no walk at this index: never passes
    } else if (ix < G.level_n[L] && key != 0x7FFFFFFFFFFFFFFFll) {
        pos = (1u << (G.H - L)) + ix;
        lvl = L;
        if (q == 0u) {
            const int4 e20 = lad[2u * (size_t)G.level_off[2]];
            const long long key20 = (long long)(((unsigned long long)(unsigned)e20.y << 32) | (unsigned)e20.x);
            key = key20 < key ? key20 : key;
        }
    }
    hq[3 * (size_t)q] = make_int4((int)(unsigned long long)key, (int)((unsigned long long)key >> 32), (int)pos, lvl);
    const int4 never = make_int4(0, (int)0x80000000u, 0, (int)0x80000000u);
    hq[3 * (size_t)q + 1] = lvl ? lad[2 * ((size_t)G.level_off[L] + ix)] : never;
    hq[3 * (size_t)q + 2] = lvl ? lad[2 * ((size_t)G.level_off[L] + ix) + 1] : never;
}

// The orbit as this kernel's steps read it: {re, im, exponent, exponent - 4 where the QUIET step may arrive}.  .w is poisoned
// unless 2^-40 <= |Z| < 2^2 and the entry is not the orbit's last; .z is the true exponent (the step with z needs it) unless
// no step may arrive here at all (beyond the orbit, or an exponent at or below -2^26: zeros), where a huge one makes the
// step's window test fail.
__global__ void __launch_bounds__(256) k_bla_make_zb(const float4 *__restrict__ zref, uint32_t count, float4 *__restrict__ zb,
                                                     uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float4 v = i < count ? zref[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int32_t e = __float_as_int(v.z);
    const bool arrivable = i < count && e > -(1 << 26);
    const bool quiet_ok = arrivable && e <= 1 && e >= -40 && i + 1u < count;
    v.z = __int_as_float(arrivable ? e : kPoisonExpHigh);
    v.w = __int_as_float(quiet_ok ? e - 4 : kPoisonExp);
    zb[i] = v;
}

// ---- literal actions (the reference's operation order; what the asm statement hands back)
struct PixelState {
    hreal32 dX, dY;   // DeltaSubN
    hreal32 cX, cY;   // DeltaSub0
    hreal32 dn;       // DeltaNormSquared
    uint32_t ref;     // RefIteration
    uint32_t iter;
};

__device__ __forceinline__ hcplx32 zref_entry(const float4 *__restrict__ z, uint32_t i)
{
    const float4 v = z[i];
    return hcplx32{v.x, v.y, __float_as_int(v.z)};
}

// BLAS::LookupBackwards on the native table (kernels_perturb.hip bla_lookup_native, per lane, without its wave votes): position or ~0u.
__device__ __forceinline__ uint32_t lookup_literal(const FsBlaArgsT<float> &A, uint32_t m, hreal32 z2)
{
    if (m == 0u)
        return 0xFFFFFFFFu;
    const int32_t k = (int32_t)m - 1;
    if ((k & 1) != 0)
        return 0xFFFFFFFFu;
    const long long zkey = (long long)(((unsigned long long)(unsigned)z2.e << 32) | (unsigned)__float_as_int(z2.m));
    const bool first = k == 0;
    const int32_t zeros = first ? 32 : (int32_t)__ffs(k) - 1;
    uint32_t ix = first ? 0u : (uint32_t)k >> (zeros & 31);
    int32_t L = zeros <= A.lm2 ? zeros : A.lm2;
    if (L < 2)
        return 0xFFFFFFFFu;
    if (first) {
        const int4 e20 = A.nlad[2u * (size_t)A.level_off[2]];
        const long long key20 = (long long)(((unsigned long long)(unsigned)e20.y << 32) | (unsigned)e20.x);
        if (!(zkey < key20))
            return 0xFFFFFFFFu;
    }
    while (L >= 2) {
        const uint32_t p = A.level_off[L] + ix;
        const int4 a = A.nlad[2u * (size_t)p], b = A.nlad[2u * (size_t)p + 1u];
        const long long k0 = (long long)(((unsigned long long)(unsigned)a.y << 32) | (unsigned)a.x);
        const long long k1 = (long long)(((unsigned long long)(unsigned)a.w << 32) | (unsigned)a.z);
        const long long k2 = (long long)(((unsigned long long)(unsigned)b.y << 32) | (unsigned)b.x);
        const long long k3 = (long long)(((unsigned long long)(unsigned)b.w << 32) | (unsigned)b.z);
        int32_t nf = zkey < k3 ? 3 : 4;
        nf = zkey < k2 ? 2 : nf;
        nf = zkey < k1 ? 1 : nf;
        nf = zkey < k0 ? 0 : nf;
        if (nf < 4)
            return A.level_off[L - nf] + (ix << nf);
        L -= 4;
        ix <<= 4;
    }
    return 0xFFFFFFFFu;
}

// One round of the reference's `while (a table entry applies)` loop for this lane (Fractal.cpp:2288-2340 through
// BLAKernels.cuh:262-330): true = a jump was applied and the loop goes on (look up again), false = the lane's next action is a
// step (nothing applied, or the jump's escape test fired).
__device__ __forceinline__ bool lookup_jump_literal(const FsBlaArgsT<float> &A, PixelState &s, uint32_t count, uint32_t n_iterations)
{
    const uint32_t pos = lookup_literal(A, s.ref, s.dn);
    if (pos == 0xFFFFFFFFu)
        return false;
    const FsBlaRec b = A.nrec[pos];
    const uint32_t l = b.l;
    if (s.ref + l >= count)
        return false;
    if (s.iter + l >= n_iterations)
        return false;
    s.iter += l;
    const hreal32 Ax{b.Axm, b.Axe}, Ay{b.Aym, b.Aye}, Bx{b.Bxm, b.Bxe}, By{b.Bym, b.Bye};
    const hreal32 nx = hr_sub(hr_add(hr_sub(hr_mul(Ax, s.dX), hr_mul(Ay, s.dY)), hr_mul(Bx, s.cX)), hr_mul(By, s.cY));
    const hreal32 ny = hr_add(hr_add(hr_add(hr_mul(Ax, s.dY), hr_mul(Ay, s.dX)), hr_mul(Bx, s.cY)), hr_mul(By, s.cX));
    s.dX = nx;
    s.dY = ny;
    s.ref += l;
    const hcplx32 Z = zref_entry(A.zref, s.ref);
    const hreal32 tempZX = hr_add(hc_re(Z), s.dX);
    const hreal32 tempZY = hr_add(hc_im(Z), s.dY);
    const hreal32 normSquared = hr_reduced(hr_add(hr_mul(tempZX, tempZX), hr_mul(tempZY, tempZY)));
    s.dn = hr_reduced(hr_add(hr_mul(s.dX, s.dX), hr_mul(s.dY, s.dY)));
    if (hr_cmp_pos(normSquared, hreal32{1.0f, 8}) > 0)
        return false;
    if (hr_cmp_pos(normSquared, s.dn) < 0 || s.ref >= count - 1) {
        s.dX = tempZX;
        s.dY = tempZY;
        s.dn = normSquared;
        s.ref = 0;
    }
    return true;
}

// One perturbation step in the literal order of Fractal.cpp:2342-2466.  false = the pixel is finished (escaped, or the orbit
// ran out): its count is s.iter as it stands.
__device__ __forceinline__ bool step_literal(const FsBlaArgsT<float> &A, PixelState &s, uint32_t count)
{
    const hcplx32 Z = zref_entry(A.zref, s.ref);
    const hreal32 OX = s.dX, OY = s.dY;
    const hreal32 T4 = hr_add(hr_mul2(hc_re(Z)), OX);
    const hreal32 T3 = hr_add(hr_mul2(hc_im(Z)), OY);
    const hreal32 TermB1 = hr_mul(OX, T4);
    const hreal32 TermB2 = hr_mul(OY, T3);
    s.dX = hr_sub(TermB1, TermB2);
    s.dX = hr_add(s.dX, s.cX);
    hr_reduce(s.dX);
    s.dY = hr_add(hr_mul(OX, T3), hr_mul(OY, T4));
    s.dY = hr_add(s.dY, s.cY);
    hr_reduce(s.dY);
    ++s.ref;
    if (s.ref >= count)
        return false;
    const hcplx32 Z2 = zref_entry(A.zref, s.ref);
    const hreal32 tempZX = hr_add(hc_re(Z2), s.dX);
    const hreal32 tempZY = hr_add(hc_im(Z2), s.dY);
    const hreal32 normSquared = hr_reduced(hr_add(hr_mul(tempZX, tempZX), hr_mul(tempZY, tempZY)));
    s.dn = hr_reduced(hr_add(hr_mul(s.dX, s.dX), hr_mul(s.dY, s.dY)));
    if (hr_cmp_pos(normSquared, hreal32{1.0f, 8}) > 0)
        return false;
    if (hr_cmp_pos(normSquared, s.dn) < 0 || s.ref >= count - 1) {
        s.dX = tempZX;
        s.dY = tempZY;
        s.dn = normSquared;
        s.ref = 0;
    }
    ++s.iter;
    return true;
}

// ------------------------------------------------------------------------------------------------
// Register map of the statement (vector registers are named: the halves of a packed pair have no operand syntax).
//   state   v0,v1   dz mantissas (X, Y)          v2,v3   dz exponents            v4,v5   dc mantissas     v6,v7  dc exponents
//           v8      RefIteration                 v9      iter                    v10,v11 |dz|^2 {mantissa bits, exponent} = the
//           v12,v13 orbit entry at RefIteration (re, im)   v14 its exponent      lookup's 64-bit key      v15 min(dc exponents)
//   temps   v16..v47 (clobbered)
//   scalar  s[36:37] c0  s[38:39] c1  s[40:41] c2  s[42:43] c3  s[44:45] hit lanes of the walk  s[46:47] t0  s[48:49] EXEC on entry
//           s50 0x807FFFFF  s51 -127  s52 2^-30  s53 2^30  s54 -2^26  s55 -400  s56 kMinBigExp
// Status on the way out: 0 = no lane is running any more; 1 = every running lane needs ONE literal step; 2 = the lanes in
// %[M] need one literal round of the lookup loop (the others of their group go on to the step).
// mode on the way in: 0 = a trip starts (every running lane looks up), 1 = the lanes in %[J] continue their lookup loop, the
// others have finished theirs.
#ifdef FS_BLA_FAST_PROBE
#define FS_CNT(R) "s_add_u32 " R ", " R ", 1\n\t"
#define FS_LANE_STEP "v_add_u32_e32 v48, 1, v48\n\t" /* steps this lane has taken (written out INSTEAD of the count) */
#define FS_CNT_ZERO "s_mov_b32 s58, 0\n\ts_mov_b32 s59, 0\n\ts_mov_b32 s60, 0\n\ts_mov_b32 s61, 0\n\ts_mov_b32 s62, 0\n\ts_mov_b32 s63, 0\n\ts_mov_b32 s64, 0\n\ts_mov_b32 s65, 0\n\ts_mov_b32 s66, 0\n\t"
// (round 6) passes whose lanes all hold the same value in V -- do the lanes of a pass read ONE orbit entry / ONE table record?
#define FS_UNI(V, R) "v_readfirstlane_b32 s67, " V "\n\ts_nop 1\n\tv_cmp_eq_u32_e32 vcc, s67, " V "\n\ts_cmp_eq_u64 vcc, exec\n\ts_cselect_b32 s67, 1, 0\n\ts_add_u32 " R ", " R ", s67\n\t"
#else
#define FS_UNI(V, R) ""
#define FS_CNT(R) ""
#define FS_LANE_STEP ""
#define FS_CNT_ZERO ""
#endif
// Q[(m - 1) / 4] into v[22:33] for the lanes that have just arrived at an index m = 1 (mod 4) -- the only ones whose next
// lookup can find anything; s57 = "requested" (the lookup then only waits).  48 bytes per record: (m - 1) * 12.
// (round 6: FS_BLA_Q_SPLIT -- the kernel keeps the CU's texture-address unit at 0.88 accesses per cycle, and three lookups of five end at the first
// key test: the record's first 16 bytes -- key, position, level -- are read ahead, the 32 bytes of ladder keys only by the lanes that pass)
#ifndef FS_BLA_Q_SPLIT
#define FS_BLA_Q_SPLIT 0 /* (A/B build; measured on C5: 133.8 against 133.5 ms -- the second round trip of the lanes that pass costs what the unread bytes saved) */
#endif
#if FS_BLA_Q_SPLIT
#define FS_Q_LOADS                                                                                                      \
    "v_mad_u32_u24 v16, v8, 12, -12\n\t"                                                                               \
    "global_load_dwordx4 v[22:25], v16, %[hq]\n\t"
#define FS_Q_LOADS_REST                                                                                                 \
    "v_mad_u32_u24 v16, v8, 12, -12\n\t"                                                                               \
    "global_load_dwordx4 v[26:29], v16, %[hq] offset:16\n\t"                                                            \
    "global_load_dwordx4 v[30:33], v16, %[hq] offset:32\n\t"                                                            \
    "s_waitcnt vmcnt(0)\n\t"
#else
#define FS_Q_LOADS                                                                                                      \
    "v_mad_u32_u24 v16, v8, 12, -12\n\t"                                                                               \
    "global_load_dwordx4 v[22:25], v16, %[hq]\n\t"                                                                      \
    "global_load_dwordx4 v[26:29], v16, %[hq] offset:16\n\t"                                                            \
    "global_load_dwordx4 v[30:33], v16, %[hq] offset:32\n\t"
#define FS_Q_LOADS_REST ""
#endif
#define FS_Q_PREFETCH(N)                                                                                                \
    "v_and_b32_e32 v16, 3, v8\n\t"                                                                                      \
    "v_cmp_eq_u32_e32 vcc, 1, v16\n\t"                                                                                  \
    "s_and_saveexec_b64 s[46:47], vcc\n\t"                                                                              \
    "s_cbranch_execz .Lbf_pf" N "_%=\n\t" FS_Q_LOADS "s_mov_b32 s57, 1\n"                                               \
    ".Lbf_pf" N "_%=:\n\t"                                                                                              \
    "s_mov_b64 exec, s[46:47]\n\t"
#define FS_BLA_ASM                                                                                                      \
    FS_CNT_ZERO "s_mov_b32 s57, 0\n\t" "s_mov_b64 s[48:49], exec\n\t"                                                                                      \
    "s_mov_b32 s50, 0x807fffff\n\t"                                                                                     \
    "s_mov_b32 s51, 0xffffff81\n\t"                                                                                     \
    "s_mov_b32 s52, 0x30800000\n\t"                                                                                     \
    "s_mov_b32 s53, 0x4e800000\n\t"                                                                                     \
    "s_mov_b32 s54, 0xfc000000\n\t"                                                                                     \
    "s_mov_b32 s55, 0xfffffe70\n\t"                                                                                     \
    "s_mov_b32 s56, 0xf0000000\n\t"                                                                                     \
    "s_cmp_lg_u32 %[mode], 0\n\t"                                                                                       \
    "s_cbranch_scc1 .Lbf_lk_%=\n"                                                                                       \
    ".Lbf_top_%=:\n\t"                                                                                                  \
    "s_mov_b64 %[J], %[R]\n"                                                                                            \
    /* ---------------- lookup: lanes in J */                                                                           \
    ".Lbf_lk_%=:\n\t"                                                                                                   \
    "s_and_b64 exec, %[J], %[R]\n\t"                                                                                    \
    "s_cbranch_scc0 .Lbf_step_%=\n\t" FS_CNT("s58")                                                                                   \
    "v_and_b32_e32 v16, 3, v8\n\t"                                                                                      \
    "v_cmp_eq_u32_e32 vcc, 1, v16\n\t"                                                                                  \
    "s_and_b64 exec, exec, vcc\n\t"                                                                                     \
    "s_cbranch_scc0 .Lbf_step_%=\n\t"                                                                                   \
    "s_cmp_lg_u32 s57, 0\n\t"                                                                                           \
    "s_cbranch_scc1 .Lbf_haveq_%=\n\t" FS_Q_LOADS                                                                       \
    ".Lbf_haveq_%=:\n\t"                                                                                                \
    "s_mov_b32 s57, 0\n\t"                                                                                              \
    "s_mov_b64 s[44:45], 0\n\t"                                                                                         \
    "s_waitcnt vmcnt(0)\n\t"                                                                                            \
    "v_cmp_lt_i64_e32 vcc, v[10:11], v[22:23]\n\t"                                                                      \
    "s_and_b64 exec, exec, vcc\n\t"                                                                                     \
    "s_cbranch_scc0 .Lbf_step_%=\n\t" FS_Q_LOADS_REST                                                                   \
    "v_cmp_eq_u32_e32 vcc, 0, v25\n\t"                                                                                  \
    "s_cbranch_vccnz .Lbf_slowlk_%=\n\t"                                                                                \
    "s_branch .Lbf_cmp_%=\n"                     /* (the first round's keys came with Q) */                            \
    ".Lbf_round_%=:\n\t" FS_CNT("s60")                                                                                  \
    "v_lshlrev_b32_e32 v16, 5, v24\n\t"                                                                                 \
    "global_load_dwordx4 v[26:29], v16, %[hlad]\n\t"                                                                    \
    "global_load_dwordx4 v[30:33], v16, %[hlad] offset:16\n\t"                                                          \
    "s_waitcnt vmcnt(0)\n"                                                                                              \
    ".Lbf_cmp_%=:\n\t"                                                                                                  \
    "v_mov_b32_e32 v17, 4\n\t"                                                                                          \
    "v_cmp_lt_i64_e64 s[42:43], v[10:11], v[32:33]\n\t"                                                                 \
    "v_cmp_lt_i64_e64 s[40:41], v[10:11], v[30:31]\n\t"                                                                 \
    "v_cmp_lt_i64_e64 s[38:39], v[10:11], v[28:29]\n\t"                                                                 \
    "v_cmp_lt_i64_e64 s[36:37], v[10:11], v[26:27]\n\t"                                                                 \
    "v_cndmask_b32_e64 v17, v17, 3, s[42:43]\n\t"                                                                       \
    "v_cndmask_b32_e64 v17, v17, 2, s[40:41]\n\t"                                                                       \
    "v_cndmask_b32_e64 v17, v17, 1, s[38:39]\n\t"                                                                       \
    "v_cndmask_b32_e64 v17, v17, 0, s[36:37]\n\t"                                                                       \
    "s_or_b64 s[42:43], s[42:43], s[40:41]\n\t"                                                                         \
    "s_or_b64 s[38:39], s[38:39], s[36:37]\n\t"                                                                         \
    "s_or_b64 s[42:43], s[42:43], s[38:39]\n\t" /* lanes that found their element this round */                         \
    "v_lshlrev_b32_e32 v24, v17, v24\n\t"       /* its position -- or, for the others, the next round's start */         \
    "v_cmp_le_i32_e32 vcc, 6, v25\n\t"          /* four more levels down is still level 2 or above */                   \
    "v_add_u32_e32 v25, -4, v25\n\t"                                                                                    \
    "s_or_b64 s[44:45], s[44:45], s[42:43]\n\t"                                                                         \
    "s_andn2_b64 exec, vcc, s[42:43]\n\t"                                                                               \
    "s_cbranch_scc1 .Lbf_round_%=\n\t"                                                                                  \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                      \
    "s_cbranch_execz .Lbf_step_%=\n\t" FS_CNT("s61") FS_UNI("v24", "s66")                                              \
    /* ---------------- jump: lanes that found an element (position v20) */                                             \
    "v_mul_u32_u24_e32 v16, 48, v24\n\t"                                                                                \
    "global_load_dwordx4 v[30:33], v16, %[hrec] offset:32\n\t" /* Z.re, Z.im, Z.exp (poisoned: not quiet), l */          \
    "global_load_dwordx4 v[22:25], v16, %[hrec]\n\t"           /* Ax, Ay, Bx, By mantissas */                            \
    "global_load_dwordx4 v[26:29], v16, %[hrec] offset:16\n\t" /* ... exponents */                                       \
    "s_waitcnt vmcnt(2)\n\t"                                                                                            \
    "v_add_u32_e64 v17, v9, v33 clamp\n\t"      /* iter + l, saturating (l = ~0: the jump would leave the orbit) */      \
    "v_cmp_gt_u32_e32 vcc, %[n], v17\n\t"                                                                               \
    "s_and_b64 exec, exec, vcc\n\t"                                                                                     \
    "s_cbranch_scc0 .Lbf_jnone_%=\n\t"                                                                                  \
    "s_waitcnt vmcnt(0)\n\t"                                                                                            \
    "v_add_u32_e32 v34, v26, v2\n\t" /* exponents of the eight products */                                              \
    "v_add_u32_e32 v26, v26, v3\n\t"                                                                                    \
    "v_add_u32_e32 v35, v27, v3\n\t"                                                                                    \
    "v_add_u32_e32 v27, v27, v2\n\t"                                                                                    \
    "v_add_u32_e32 v36, v28, v6\n\t"                                                                                    \
    "v_add_u32_e32 v28, v28, v7\n\t"                                                                                    \
    "v_add_u32_e32 v37, v29, v7\n\t"                                                                                    \
    "v_add_u32_e32 v29, v29, v6\n\t"                                                                                    \
    "v_max3_i32 v42, v34, v35, v36\n\t"                                                                                 \
    "v_max3_i32 v43, v26, v27, v28\n\t"                                                                                 \
    "v_max_i32_e32 v42, v42, v37\n\t" /* Ex */                                                                          \
    "v_max_i32_e32 v43, v43, v29\n\t" /* Ey */                                                                          \
    "v_sub_u32_e32 v34, v34, v42\n\t"                                                                                   \
    "v_sub_u32_e32 v35, v35, v42\n\t"                                                                                   \
    "v_sub_u32_e32 v36, v36, v42\n\t"                                                                                   \
    "v_sub_u32_e32 v37, v37, v42\n\t"                                                                                   \
    "v_sub_u32_e32 v26, v26, v43\n\t"                                                                                   \
    "v_sub_u32_e32 v27, v27, v43\n\t"                                                                                   \
    "v_sub_u32_e32 v28, v28, v43\n\t"                                                                                   \
    "v_sub_u32_e32 v29, v29, v43\n\t"                                                                                   \
    "v_pk_mul_f32 v[38:39], v[22:23], v[0:1] op_sel:[0,0] op_sel_hi:[0,1]\n\t" /* (Ax dX, Ax dY) */                     \
    "v_ldexp_f32 v40, v38, v34\n\t"                                                                                     \
    "v_ldexp_f32 v41, v39, v26\n\t"                                                                                     \
    "v_pk_mul_f32 v[38:39], v[22:23], v[0:1] op_sel:[1,1] op_sel_hi:[1,0]\n\t" /* (Ay dY, Ay dX) */                     \
    "v_ldexp_f32 v38, v38, v35\n\t"                                                                                     \
    "v_ldexp_f32 v39, v39, v27\n\t"                                                                                     \
    "v_pk_add_f32 v[40:41], v[40:41], v[38:39] neg_lo:[0,1] neg_hi:[0,0]\n\t"                                           \
    "v_min_f32_e64 v44, |v40|, |v41|\n\t"                                                                               \
    "v_pk_mul_f32 v[38:39], v[24:25], v[4:5] op_sel:[0,0] op_sel_hi:[0,1]\n\t" /* (Bx cX, Bx cY) */                     \
    "v_ldexp_f32 v38, v38, v36\n\t"                                                                                     \
    "v_ldexp_f32 v39, v39, v28\n\t"                                                                                     \
    "v_pk_add_f32 v[40:41], v[40:41], v[38:39]\n\t"                                                                     \
    "v_min3_f32 v44, |v40|, |v41|, v44\n\t"                                                                             \
    "v_pk_mul_f32 v[38:39], v[24:25], v[4:5] op_sel:[1,1] op_sel_hi:[1,0]\n\t" /* (By cY, By cX) */                     \
    "v_ldexp_f32 v38, v38, v37\n\t"                                                                                     \
    "v_ldexp_f32 v39, v39, v29\n\t"                                                                                     \
    "v_pk_add_f32 v[40:41], v[40:41], v[38:39] neg_lo:[0,1] neg_hi:[0,0]\n\t" /* the new dz under (Ex, Ey) */           \
    "v_cmp_lt_f32_e32 vcc, 0, v44\n\t"          /* no partial sum is an exact zero */                                   \
    "v_min_f32_e64 v45, |v40|, |v41|\n\t"                                                                               \
    "v_max_f32_e64 v46, |v40|, |v41|\n\t"                                                                               \
    "v_cmp_le_f32_e64 s[36:37], s52, v45\n\t"                                                                           \
    "v_cmp_ge_f32_e64 s[38:39], s53, v46\n\t"                                                                           \
    "v_min3_i32 v45, v2, v3, v15\n\t"                                                                                   \
    "v_cmp_lt_i32_e64 s[40:41], s54, v45\n\t"                                                                           \
    "v_bfe_u32 v45, v40, 23, 8\n\t"                                                                                     \
    "v_bfe_u32 v46, v41, 23, 8\n\t"                                                                                     \
    "v_add3_u32 v45, v42, v45, s51\n\t"                                                                                 \
    "v_add3_u32 v46, v43, v46, s51\n\t"                                                                                 \
    "v_max_i32_e32 v45, v45, v46\n\t"                                                                                   \
    "v_add_u32_e32 v46, -4, v32\n\t"                                                                                    \
    "v_cmp_le_i32_e64 s[42:43], v45, v46\n\t"   /* quiet: both parts four binades below the arrival entry */            \
    "s_and_b64 s[36:37], s[36:37], vcc\n\t"                                                                             \
    "s_and_b64 s[38:39], s[38:39], s[40:41]\n\t"                                                                        \
    "s_and_b64 s[36:37], s[36:37], s[42:43]\n\t"                                                                        \
    "s_and_b64 s[36:37], s[36:37], s[38:39]\n\t"                                                                        \
    "s_xor_b64 s[36:37], s[36:37], exec\n\t"                                                                            \
    "s_cbranch_scc1 .Lbf_slowlk_%=\n\t"         /* some lane's jump is not the quiet form: nothing committed */          \
    "v_pk_mov_b32 v[0:1], v[40:41], v[40:41] op_sel:[0,1]\n\t"                                                         \
    "v_pk_mov_b32 v[2:3], v[42:43], v[42:43] op_sel:[0,1]\n\t"                                                         \
    "v_add_u32_e32 v8, v8, v33\n\t"                                                                                     \
    "v_mov_b32_e32 v9, v17\n\t"                                                                                         \
    "v_pk_mov_b32 v[12:13], v[30:31], v[30:31] op_sel:[0,1]\n\t"                                                       \
    "v_mov_b32_e32 v14, v32\n\t"                                                                                        \
    "v_pk_mul_f32 v[38:39], v[40:41], v[40:41]\n\t" /* |dz|^2 for the lookup that follows */                            \
    "v_max_i32_e32 v45, v42, v43\n\t"                                                                                   \
    "v_sub_u32_e32 v46, v42, v45\n\t"                                                                                   \
    "v_sub_u32_e32 v47, v43, v45\n\t"                                                                                   \
    "v_lshlrev_b32_e32 v46, 1, v46\n\t"                                                                                 \
    "v_lshlrev_b32_e32 v47, 1, v47\n\t"                                                                                 \
    "v_ldexp_f32 v38, v38, v46\n\t"                                                                                     \
    "v_ldexp_f32 v39, v39, v47\n\t"                                                                                     \
    "v_add_f32_e32 v38, v38, v39\n\t"                                                                                   \
    "v_bfe_u32 v39, v38, 23, 8\n\t"                                                                                     \
    "v_and_or_b32 v10, v38, s50, 1.0\n\t"                                                                               \
    "v_lshl_add_u32 v45, v45, 1, v39\n\t"                                                                               \
    "v_add_u32_e32 v11, s51, v45\n\t"                                                                                   \
    "s_mov_b64 %[J], exec\n\t" FS_Q_PREFETCH("j")                                                                       \
    "s_branch .Lbf_lk_%=\n"                                                                                             \
    ".Lbf_jnone_%=:\n\t"                                                                                                \
    "s_waitcnt vmcnt(0)\n"                                                                                              \
    /* ---------------- step: every running lane */                                                                     \
    ".Lbf_step_%=:\n\t"                                                                                                 \
    "s_mov_b64 exec, %[R]\n\t"                                                                                          \
    "s_waitcnt vmcnt(0)\n\t"                    /* (a requested Q that no lookup consumed) */                           \
    "s_mov_b32 s57, 0\n\t"                                                                                              \
    "v_lshl_add_u32 v16, v8, 4, 16\n\t" FS_CNT("s62") FS_LANE_STEP FS_UNI("v8", "s65")                                                                                 \
    "global_load_dwordx4 v[18:21], v16, %[zb]\n\t" /* the entry the step arrives at: re, im, exponent, quiet bound */    \
    "v_add_u32_e32 v17, 1, v14\n\t"                                                                                     \
    "v_max3_i32 v42, v17, v2, v3\n\t"           /* T = 2Z + dz under eT */                                              \
    "v_sub_u32_e32 v17, v17, v42\n\t"                                                                                   \
    "v_sub_u32_e32 v34, v2, v42\n\t"                                                                                    \
    "v_sub_u32_e32 v35, v3, v42\n\t"                                                                                    \
    "v_ldexp_f32 v38, v12, v17\n\t"                                                                                     \
    "v_ldexp_f32 v39, v13, v17\n\t"                                                                                     \
    "v_ldexp_f32 v40, v0, v34\n\t"                                                                                      \
    "v_ldexp_f32 v41, v1, v35\n\t"                                                                                      \
    "v_pk_add_f32 v[38:39], v[38:39], v[40:41]\n\t"                                                                     \
    "v_pk_mul_f32 v[22:23], v[0:1], v[38:39] op_sel:[0,0] op_sel_hi:[0,1]\n\t" /* dX (T.x, T.y) */                       \
    "v_pk_mul_f32 v[24:25], v[0:1], v[38:39] op_sel:[1,1] op_sel_hi:[1,0]\n\t" /* dY (T.y, T.x) */                       \
    "v_max_i32_e32 v43, v2, v3\n\t"                                                                                     \
    "v_sub_u32_e32 v34, v2, v43\n\t"                                                                                    \
    "v_sub_u32_e32 v35, v3, v43\n\t"                                                                                    \
    "v_ldexp_f32 v22, v22, v34\n\t"                                                                                     \
    "v_ldexp_f32 v23, v23, v34\n\t"                                                                                     \
    "v_ldexp_f32 v24, v24, v35\n\t"                                                                                     \
    "v_ldexp_f32 v25, v25, v35\n\t"                                                                                     \
    "v_pk_add_f32 v[40:41], v[22:23], v[24:25] neg_lo:[0,1] neg_hi:[0,0]\n\t" /* N under E */                           \
    "v_add_u32_e32 v43, v43, v42\n\t"                                                                                   \
    "v_max3_i32 v42, v43, v6, v7\n\t"           /* Q = N + dc under EQ */                                               \
    "v_sub_u32_e32 v43, v43, v42\n\t"                                                                                   \
    "v_sub_u32_e32 v34, v6, v42\n\t"                                                                                    \
    "v_sub_u32_e32 v35, v7, v42\n\t"                                                                                    \
    "v_ldexp_f32 v22, v40, v43\n\t"                                                                                     \
    "v_ldexp_f32 v23, v41, v43\n\t"                                                                                     \
    "v_ldexp_f32 v24, v4, v34\n\t"                                                                                      \
    "v_ldexp_f32 v25, v5, v35\n\t"                                                                                      \
    "v_pk_add_f32 v[22:23], v[22:23], v[24:25]\n\t"                                                                     \
    "v_bfe_u32 v34, v22, 23, 8\n\t"                                                                                     \
    "v_bfe_u32 v35, v23, 23, 8\n\t"                                                                                     \
    "v_add3_u32 v36, v42, v34, s51\n\t"         /* exponents of the reduced parts */                                    \
    "v_add3_u32 v37, v42, v35, s51\n\t"                                                                                 \
    "v_max3_f32 v44, |v38|, |v39|, |v40|\n\t"   /* every sum inside 2^+-30 */                                           \
    "v_min3_f32 v45, |v38|, |v39|, |v40|\n\t"                                                                           \
    "v_max3_f32 v44, |v41|, |v22|, v44\n\t"                                                                             \
    "v_min3_f32 v45, |v41|, |v22|, v45\n\t"                                                                             \
    "v_max_f32_e64 v44, |v23|, v44\n\t"                                                                                 \
    "v_min_f32_e64 v45, |v23|, v45\n\t"                                                                                 \
    "v_cmp_ge_f32_e64 s[36:37], s53, v44\n\t"                                                                           \
    "v_cmp_le_f32_e32 vcc, s52, v45\n\t"                                                                                \
    "s_and_b64 s[36:37], s[36:37], vcc\n\t"                                                                             \
    "v_min_i32_e32 v46, v2, v3\n\t"                                                                                     \
    "v_cmp_lt_i32_e32 vcc, s54, v46\n\t"                                                                                \
    "s_and_b64 s[36:37], s[36:37], vcc\n\t"     /* ok_dz */                                                             \
    "v_max_i32_e32 v46, v36, v37\n\t"                                                                                   \
    "s_waitcnt vmcnt(0)\n\t"                                                                                            \
    "v_cmp_le_i32_e32 vcc, v46, v21\n\t"        /* quiet */                                                             \
    "s_and_b64 vcc, vcc, s[36:37]\n\t"                                                                                  \
    "s_xor_b64 s[46:47], vcc, exec\n\t"                                                                                 \
    "s_cbranch_scc1 .Lbf_wz_%=\n\t"                                                                                     \
    /* quiet commit */                                                                                                  \
    "v_and_or_b32 v0, v22, s50, 1.0\n\t"                                                                                \
    "v_and_or_b32 v1, v23, s50, 1.0\n\t"                                                                                \
    "v_pk_mov_b32 v[2:3], v[36:37], v[36:37] op_sel:[0,1]\n\t"                                                         \
    "v_pk_mov_b32 v[12:13], v[18:19], v[18:19] op_sel:[0,1]\n\t"                                                       \
    "v_mov_b32_e32 v14, v20\n\t"                                                                                        \
    "v_add_u32_e32 v8, 1, v8\n\t"                                                                                       \
    "v_add_u32_e32 v9, 1, v9\n\t"                                                                                       \
    "v_cmp_gt_u32_e32 vcc, %[n], v9\n\t"        /* lanes at the cap leave with it */                                    \
    "s_mov_b64 %[R], vcc\n\t"                                                                                           \
    "v_and_b32_e32 v16, 3, v8\n\t"                                                                                      \
    "v_cmp_eq_u32_e32 vcc, 1, v16\n\t"                                                                                  \
    "s_cbranch_vccz .Lbf_next_%=\n\t"           /* no lane arrives where a lookup can find anything: |dz|^2 not needed */ \
    "v_pk_mul_f32 v[24:25], v[0:1], v[0:1]\n\t"                                                                         \
    "v_max_i32_e32 v43, v2, v3\n\t"                                                                                     \
    "v_sub_u32_e32 v34, v2, v43\n\t"                                                                                    \
    "v_sub_u32_e32 v35, v3, v43\n\t"                                                                                    \
    "v_lshlrev_b32_e32 v34, 1, v34\n\t"                                                                                 \
    "v_lshlrev_b32_e32 v35, 1, v35\n\t"                                                                                 \
    "v_ldexp_f32 v24, v24, v34\n\t"                                                                                     \
    "v_ldexp_f32 v25, v25, v35\n\t"                                                                                     \
    "v_add_f32_e32 v24, v24, v25\n\t"                                                                                   \
    "v_bfe_u32 v25, v24, 23, 8\n\t"                                                                                     \
    "v_and_or_b32 v10, v24, s50, 1.0\n\t"                                                                               \
    "v_lshl_add_u32 v43, v43, 1, v25\n\t"                                                                               \
    "v_add_u32_e32 v11, s51, v43\n\t" FS_Q_PREFETCH("q")                                                                \
    "s_branch .Lbf_next_%=\n"                                                                                           \
    /* the step with z = Z' + dz' */                                                                                    \
    ".Lbf_wz_%=:\n\t" FS_CNT("s63")                                                                                                   \
    "s_xor_b64 s[46:47], s[36:37], exec\n\t"                                                                            \
    "s_cbranch_scc1 .Lbf_slowstep_%=\n\t"                                                                               \
    "v_and_or_b32 v22, v22, s50, 1.0\n\t"       /* reduced mantissas of the new dz */                                   \
    "v_and_or_b32 v23, v23, s50, 1.0\n\t"                                                                               \
    "v_pk_mul_f32 v[24:25], v[22:23], v[22:23]\n\t" /* |dz'|^2 */                                                       \
    "v_max_i32_e32 v43, v36, v37\n\t"                                                                                   \
    "v_sub_u32_e32 v34, v36, v43\n\t"                                                                                   \
    "v_sub_u32_e32 v35, v37, v43\n\t"                                                                                   \
    "v_lshlrev_b32_e32 v34, 1, v34\n\t"                                                                                 \
    "v_lshlrev_b32_e32 v35, 1, v35\n\t"                                                                                 \
    "v_ldexp_f32 v24, v24, v34\n\t"                                                                                     \
    "v_ldexp_f32 v25, v25, v35\n\t"                                                                                     \
    "v_add_f32_e32 v24, v24, v25\n\t"           /* dnm, exponent 2 max(nxe, nye) */                                     \
    "v_lshlrev_b32_e32 v43, 1, v43\n\t"         /* dne */                                                               \
    "v_max3_i32 v47, v20, v36, v37\n\t"         /* ez */                                                                \
    "v_sub_u32_e32 v34, v20, v47\n\t"                                                                                   \
    "v_sub_u32_e32 v35, v36, v47\n\t"                                                                                   \
    "v_sub_u32_e32 v46, v37, v47\n\t"                                                                                   \
    "v_ldexp_f32 v38, v18, v34\n\t"                                                                                     \
    "v_ldexp_f32 v39, v19, v34\n\t"                                                                                     \
    "v_ldexp_f32 v40, v22, v35\n\t"                                                                                     \
    "v_ldexp_f32 v41, v23, v46\n\t"                                                                                     \
    "v_pk_add_f32 v[38:39], v[38:39], v[40:41]\n\t" /* z under ez */                                                    \
    "v_pk_mul_f32 v[40:41], v[38:39], v[38:39]\n\t"                                                                     \
    "v_max_f32_e64 v44, |v38|, |v39|\n\t"                                                                               \
    "v_min_f32_e64 v45, |v38|, |v39|\n\t"                                                                               \
    "v_add_f32_e32 v40, v40, v41\n\t"           /* |z|^2, exponent 2 ez */                                              \
    "v_cmp_ge_f32_e64 s[38:39], s53, v44\n\t"                                                                           \
    "v_cmp_le_f32_e32 vcc, s52, v45\n\t"                                                                                \
    "s_and_b64 s[38:39], s[38:39], vcc\n\t"                                                                             \
    "s_xor_b64 s[46:47], s[38:39], exec\n\t"                                                                            \
    "s_cbranch_scc1 .Lbf_slowstep_%=\n\t"                                                                               \
    "v_pk_mov_b32 v[0:1], v[22:23], v[22:23] op_sel:[0,1]\n\t" /* commit */                                            \
    "v_pk_mov_b32 v[2:3], v[36:37], v[36:37] op_sel:[0,1]\n\t"                                                         \
    "v_pk_mov_b32 v[12:13], v[18:19], v[18:19] op_sel:[0,1]\n\t"                                                       \
    "v_mov_b32_e32 v14, v20\n\t"                                                                                        \
    "v_add_u32_e32 v8, 1, v8\n\t"                                                                                       \
    "v_bfe_u32 v25, v24, 23, 8\n\t"                                                                                     \
    "v_and_or_b32 v10, v24, s50, 1.0\n\t"                                                                               \
    "v_add3_u32 v11, v43, v25, s51\n\t"                                                                                 \
    "v_lshlrev_b32_e32 v42, 1, v47\n\t"         /* 2 ez */                                                              \
    "v_add_u32_e32 v34, -8, v42\n\t"                                                                                    \
    "v_max_i32_e32 v34, s55, v34\n\t"                                                                                   \
    "v_ldexp_f32 v35, v40, v34\n\t"                                                                                     \
    "v_cmp_lt_f32_e32 vcc, 1.0, v35\n\t"        /* |z|^2 > 256: the pixel is finished with the count it has */           \
    "s_andn2_b64 exec, exec, vcc\n\t"                                                                                   \
    "s_cbranch_scc0 .Lbf_wzdone_%=\n\t"         /* every lane of the wave escaped */                                    \
    "v_sub_u32_e32 v34, v43, v42\n\t"                                                                                   \
    "v_max_i32_e32 v34, s55, v34\n\t"                                                                                   \
    "v_ldexp_f32 v35, v24, v34\n\t"                                                                                     \
    "v_cmp_lt_f32_e32 vcc, v40, v35\n\t"        /* |z|^2 < |dz|^2 */                                                    \
    "v_cmp_le_u32_e64 s[38:39], %[cm1], v8\n\t" /* ... or the orbit ends */                                             \
    "s_or_b64 vcc, vcc, s[38:39]\n\t"                                                                                   \
    "s_and_saveexec_b64 s[46:47], vcc\n\t"                                                                              \
    "s_cbranch_execz .Lbf_norebase_%=\n\t" FS_CNT("s64")                                                                              \
    "v_max_i32_e32 v2, v20, v36\n\t"            /* rebase: dz = z, each part under max(exponent of Z', of its own dz') */ \
    "v_max_i32_e32 v3, v20, v37\n\t"                                                                                    \
    "v_sub_u32_e32 v34, v47, v2\n\t"                                                                                    \
    "v_sub_u32_e32 v35, v47, v3\n\t"                                                                                    \
    "v_ldexp_f32 v0, v38, v34\n\t"                                                                                      \
    "v_ldexp_f32 v1, v39, v35\n\t"                                                                                      \
    "v_bfe_u32 v34, v40, 23, 8\n\t"                                                                                     \
    "v_and_or_b32 v10, v40, s50, 1.0\n\t"                                                                               \
    "v_add3_u32 v11, v42, v34, s51\n\t"                                                                                 \
    "v_mov_b32_e32 v8, 0\n\t"                                                                                           \
    "v_mov_b32_e32 v12, 0\n\t"                  /* orbit entry 0 */                                                     \
    "v_mov_b32_e32 v13, 0\n\t"                                                                                          \
    "v_mov_b32_e32 v14, s56\n"                                                                                          \
    ".Lbf_norebase_%=:\n\t"                                                                                             \
    "s_mov_b64 exec, s[46:47]\n\t"                                                                                      \
    "v_add_u32_e32 v9, 1, v9\n"                                                                                           \
    ".Lbf_wzdone_%=:\n\t"                                                                                               \
    "v_cmp_gt_u32_e32 vcc, %[n], v9\n\t"        /* (EXEC = the lanes that did not escape) */                            \
    "s_mov_b64 %[R], vcc\n\t" FS_Q_PREFETCH("z")                                                                        \
    "s_nop 0\n"                                                                                             \
    ".Lbf_next_%=:\n\t"                                                                                                 \
    "s_cmp_lg_u64 %[R], 0\n\t"                                                                                          \
    "s_cbranch_scc0 .Lbf_done_%=\n\t"                                                                                   \
    "s_add_u32 %[bud], %[bud], -1\n\t"          /* trips until the workgroup's next checkpoint (pooling) */             \
    "s_cmp_eq_u32 %[bud], 0\n\t"                                                                                        \
    "s_cbranch_scc0 .Lbf_top_%=\n\t"                                                                                    \
    "s_mov_b32 %[st], 3\n\t"                                                                                            \
    "s_branch .Lbf_end_%=\n"                                                                                             \
    ".Lbf_done_%=:\n\t"                                                                                                 \
    "s_mov_b32 %[st], 0\n\t"                                                                                            \
    "s_branch .Lbf_end_%=\n"                                                                                            \
    ".Lbf_slowstep_%=:\n\t"                                                                                             \
    "s_mov_b32 %[st], 1\n\t"                                                                                            \
    "s_branch .Lbf_end_%=\n"                                                                                            \
    ".Lbf_slowlk_%=:\n\t"                                                                                               \
    "s_mov_b64 %[M], exec\n\t"                                                                                          \
    "s_mov_b32 %[st], 2\n"                                                                                              \
    ".Lbf_end_%=:\n\t"                                                                                                  \
    "s_waitcnt vmcnt(0)\n\t"                    /* nothing stays in flight into registers the compiler owns again */    \
    "s_mov_b64 exec, s[48:49]"

__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// kPool: the running pixels of a workgroup's four waves (four 8 x 8 tiles side by side) are re-packed into as few waves as possible.
// A wave runs until its slowest pixel is done -- 1038 trips on C5 against a mean of 741 steps per pixel -- and a pass costs the
// same whatever EXEC holds (tools/microbench/exec_mask_rate.hip), so the tail of every tile runs at a fraction of the lanes.
// Every kPoolEvery trips the waves of a workgroup meet: finished pixels are stored, the running lanes are counted, and when
// they fit in fewer waves than hold them, every running lane writes its 17 words of state (the statement's sixteen registers and
// the pixel it belongs to) to LDS at its rank and the lowest waves read them back densely; waves left without a lane end.
constexpr uint32_t kPoolEvery = 32;
constexpr uint32_t kPoolWords = 17;

template <bool kPool> __global__ void __launch_bounds__(256) k_bla_hdr32_fast(FsBlaArgsT<float> A)
{
    __shared__ uint32_t s_cnt[kPool ? 8 : 1];
    __shared__ uint32_t s_slot[kPool ? 256 * kPoolWords : 1];
    const uint32_t count = A.orbit_count;
    const uint32_t n_iterations = A.n_iterations;
    uint32_t X, L;
    tile_pixel(X, L);
    const uint32_t Y = global_row(A.frame, L);
    const bool have = X < A.frame.width && L < A.frame.local_rows && Y < A.frame.height;
    // per-pixel state in the registers the statement names
    float dXm = 0.0f, dYm = 0.0f, cXm = 0.0f, cYm = 0.0f, dnm = 0.0f, Zre = 0.0f, Zim = 0.0f;
    int dXe = kMinBigExp, dYe = kMinBigExp, cXe = kMinBigExp, cYe = kMinBigExp, dne = kMinBigExp, Ze = kMinBigExp, cemin = kMinBigExp;
    uint32_t ref = 0, iter = 0;
    if (have) {
        hreal32 a, b;
        // Pixel -> delta c, Fractal.cpp:2272-2281 (lav2_common.hpp pixel_delta)
        a = hr_mul(A.coords.dx, hr_from_mant<float>((float)X));
        hr_reduce(a);
        a = hr_sub(a, A.coords.centerX);
        b = hr_mul(hr_neg(A.coords.dy), hr_from_mant<float>((float)Y));
        hr_reduce(b);
        b = hr_sub(b, A.coords.centerY);
        hr_reduce(a);
        hr_reduce(b);
        cXm = a.m, cXe = a.e, cYm = b.m, cYe = b.e;
        cemin = cXe < cYe ? cXe : cYe;
        const float4 z0 = A.zref[0];
        Zre = z0.x, Zim = z0.y, Ze = __float_as_int(z0.z);
    }
    uint64_t R = __builtin_amdgcn_ballot_w64(have && n_iterations != 0u);
    uint64_t J = 0;
    uint32_t mode = 0;
    uint32_t budget = kPool ? kPoolEvery : 0x7FFFFFFFu;
    uint32_t pix = threadIdx.x; // which pixel of the workgroup's 32 x 8 block this lane's state belongs to
    bool mine = have;           // ... and whether it holds one at all
    uint32_t round = 0;
#ifdef FS_BLA_FAST_PROBE
    uint32_t n_enter = 0, n_slow_step = 0, n_slow_lk = 0; // (measurement build: how often the statement is left, per wave)
    uint32_t lane_steps = 0;
    uint32_t pc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, pacc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // passes: lookup, pre-test, ladder round, jump, step, step with z, rebase
#endif
    while (kPool || R != 0ull) {
        uint32_t st = 0;
        uint64_t M = 0;
        // (scalar operands of the statement must be provably uniform where they enter it)
        R = uniform64(R), J = uniform64(J);
        budget = (uint32_t)__builtin_amdgcn_readfirstlane((int)budget);
        if (!kPool || R != 0ull)
        asm volatile(FS_BLA_ASM
                     : "+{v0}"(dXm), "+{v1}"(dYm), "+{v2}"(dXe), "+{v3}"(dYe), "+{v4}"(cXm), "+{v5}"(cYm), "+{v6}"(cXe),
                       "+{v7}"(cYe), "+{v8}"(ref), "+{v9}"(iter), "+{v10}"(dnm), "+{v11}"(dne), "+{v12}"(Zre), "+{v13}"(Zim),
                       "+{v14}"(Ze), "+{v15}"(cemin), [R] "+s"(R), [J] "+s"(J), [st] "=&s"(st), [M] "=&s"(M), [bud] "+s"(budget)
#ifdef FS_BLA_FAST_PROBE
                       , "+{v48}"(lane_steps), "={s58}"(pc[0]), "={s59}"(pc[1]), "={s60}"(pc[2]), "={s61}"(pc[3]), "={s62}"(pc[4]), "={s63}"(pc[5]), "={s64}"(pc[6]), "={s65}"(pc[7]), "={s66}"(pc[8])
#endif
                     : [mode] "s"(__builtin_amdgcn_readfirstlane((int)mode)), [zb] "s"(A.zb), [hq] "s"(A.hq), [hlad] "s"(A.hlad), [hrec] "s"(A.hrec),
                       [n] "s"(n_iterations), [cm1] "s"(count - 1u)
                     : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30",
                       "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45",
                       "v46", "v47", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48",
                       "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s67", "vcc", "scc", "memory");
        st = (uint32_t)__builtin_amdgcn_readfirstlane((int)st);
        R = uniform64(R), J = uniform64(J);
        budget = (uint32_t)__builtin_amdgcn_readfirstlane((int)budget);
#ifdef FS_BLA_FAST_PROBE
        n_enter++;
        for (int i = 0; i < 9; i++)
            pacc[i] += (uint32_t)__builtin_amdgcn_readfirstlane((int)pc[i]);
        n_slow_step += st == 1u;
        n_slow_lk += st == 2u;
#endif
        if constexpr (kPool) {
            if (st == 0u || st == 3u) {
                // ---- checkpoint of the workgroup (every wave that is still alive passes here the same number of times)
                const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
                bool running = ((R >> lane) & 1ull) != 0ull;
                if (mine && !running) { // this lane's pixel is finished: its count goes out, the lane is free
                    const uint32_t px = (blockIdx.x * 4u + (pix >> 6)) * 8u + (pix & 7u), pl = blockIdx.y * 8u + ((pix & 63u) >> 3);
                    store_iter(A.out, A.frame, pl, px, iter);
                    mine = false;
                }
                uint32_t *cnt = s_cnt + ((round & 1u) << 2); // (two sets in rotation: one barrier per checkpoint suffices)
                round++;
                if (lane == 0u)
                    cnt[wave] = (uint32_t)__popcll(R);
                __syncthreads();
                const uint32_t c0 = cnt[0], c1 = cnt[1], c2 = cnt[2], c3 = cnt[3];
                const uint32_t total = c0 + c1 + c2 + c3;
                if (total == 0u)
                    break;
                const uint32_t holding = (c0 != 0u) + (c1 != 0u) + (c2 != 0u) + (c3 != 0u), needed = (total + 63u) >> 6;
                if (needed < holding) {
                    const uint32_t base = (wave > 0u ? c0 : 0u) + (wave > 1u ? c1 : 0u) + (wave > 2u ? c2 : 0u);
                    const uint32_t rank = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(R >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)R, 0u));
                    if (running) {
                        uint32_t *d = s_slot + rank * kPoolWords;
                        d[0] = __float_as_uint(dXm), d[1] = __float_as_uint(dYm), d[2] = (uint32_t)dXe, d[3] = (uint32_t)dYe;
                        d[4] = __float_as_uint(cXm), d[5] = __float_as_uint(cYm), d[6] = (uint32_t)cXe, d[7] = (uint32_t)cYe;
                        d[8] = ref, d[9] = iter, d[10] = __float_as_uint(dnm), d[11] = (uint32_t)dne;
                        d[12] = __float_as_uint(Zre), d[13] = __float_as_uint(Zim), d[14] = (uint32_t)Ze, d[15] = (uint32_t)cemin;
                        d[16] = pix;
                    }
                    __syncthreads();
                    running = threadIdx.x < total;
                    if (running) {
                        const uint32_t *q = s_slot + threadIdx.x * kPoolWords;
                        dXm = __uint_as_float(q[0]), dYm = __uint_as_float(q[1]), dXe = (int)q[2], dYe = (int)q[3];
                        cXm = __uint_as_float(q[4]), cYm = __uint_as_float(q[5]), cXe = (int)q[6], cYe = (int)q[7];
                        ref = q[8], iter = q[9], dnm = __uint_as_float(q[10]), dne = (int)q[11];
                        Zre = __uint_as_float(q[12]), Zim = __uint_as_float(q[13]), Ze = (int)q[14], cemin = (int)q[15];
                        pix = q[16];
                    }
                    mine = running;
                    R = __builtin_amdgcn_ballot_w64(running);
                    __syncthreads(); // (the slots are free for the next compaction)
                    // (a wave left without a lane stays and keeps attending the checkpoints -- it waits at the barrier, which
                    // costs nothing -- until the whole workgroup is done: letting it end here made the frame wrong on gfx950,
                    // although a barrier is documented to wait for the surviving waves only)
                }
                budget = kPoolEvery;
                mode = 0;
                continue;
            }
        } else if (st == 0u) {
            break;
        }
        M = uniform64(M);
        const uint32_t lane = threadIdx.x & 63u;
        const bool running = ((R >> lane) & 1ull) != 0ull;
        PixelState s{hreal32{dXm, dXe}, hreal32{dYm, dYe}, hreal32{cXm, cXe}, hreal32{cYm, cYe}, hreal32{dnm, dne}, ref, iter};
        if (st == 1u) {
            bool still = false;
            if (running) {
                still = step_literal(A, s, count);
                if (still && s.iter >= n_iterations)
                    still = false;
            }
            R = __builtin_amdgcn_ballot_w64(still);
            mode = 0;
        } else {
            bool again = false;
            if (running && ((M >> lane) & 1ull) != 0ull)
                again = lookup_jump_literal(A, s, count, n_iterations);
            J = __builtin_amdgcn_ballot_w64(again);
            mode = 1;
        }
        if (running) {
            dXm = s.dX.m, dXe = s.dX.e, dYm = s.dY.m, dYe = s.dY.e, dnm = s.dn.m, dne = s.dn.e, ref = s.ref, iter = s.iter;
            // the entry at RefIteration, true exponent (the statement's invariant)
            const float4 z = A.zref[ref];
            Zre = z.x, Zim = z.y, Ze = __float_as_int(z.z);
        }
    }
#ifdef FS_BLA_FAST_PROBE
    if (A.probe_pitch == 0x57E9u) // (probe switch FSMI355_BLA_STEPS_OUT=1: the buffer receives the STEPS each pixel took)
        iter = lane_steps;
#endif
    if (!kPool && have)
        store_iter(A.out, A.frame, L, X, iter);
#ifdef FS_BLA_FAST_PROBE
    if ((threadIdx.x & 63u) == 0u && A.stats) {
        atomicAdd((unsigned long long *)&A.stats[20], (unsigned long long)n_enter);
        atomicAdd((unsigned long long *)&A.stats[21], (unsigned long long)n_slow_step);
        atomicAdd((unsigned long long *)&A.stats[22], (unsigned long long)n_slow_lk);
        atomicAdd((unsigned long long *)&A.stats[23], 1ull);
        for (int i = 0; i < 9; i++) // (words 31, 32: step passes with all lanes at one orbit entry, jump passes with all lanes at one record)
            atomicAdd((unsigned long long *)&A.stats[24 + i], (unsigned long long)pacc[i]);
    }
#endif
}

} // namespace

// Heap geometry of a table with these level sizes: H = max over levels of (L + ceil(log2(elements of L))); 0 = not usable
// (positions must stay below 2^24: the record address is one 24-bit multiply).
static int heap_height(const uint64_t *epl, int n_levels)
{
    int H = 0;
    for (int l = 2; l < n_levels && l < kBlaMaxLevels; l++) {
        if (epl[l] == 0)
            continue;
        int lg = 0;
        while (((uint64_t)1 << lg) < epl[l])
            lg++;
        H = H > l + lg ? H : l + lg;
    }
    if (H < 3 || H - 1 > 24)
        return 0;
    return H;
}

uint64_t fsk_bla_heap_positions(const uint64_t *epl, int n_levels)
{
    const int H = heap_height(epl, n_levels);
    return H ? (uint64_t)1 << (H - 1) : 0;
}

void fsk_bla_make_heap(const FsBlaRec *rec, const int4 *lad, const long long *kmax, uint32_t n_kmax, const uint32_t *level_off,
                       const uint64_t *epl, int n_levels, int32_t lm2, const float4 *zref, uint32_t orbit_count, FsBlaRec *hrec,
                       int4 *hlad, int4 *hq, float4 *zb, hipStream_t s)
{
    HeapGeom G;
    memset(&G, 0, sizeof(G));
    G.n_levels = n_levels < kBlaMaxLevels ? n_levels : kBlaMaxLevels;
    G.H = heap_height(epl, n_levels);
    uint32_t total = 0;
    for (int l = 2; l < G.n_levels; l++) {
        G.level_off[l] = level_off[l];
        G.level_n[l] = (uint32_t)epl[l];
        total = level_off[l] + (uint32_t)epl[l];
    }
    G.total = total;
    if (G.H == 0 || total == 0)
        return;
    hipLaunchKernelGGL(k_bla_make_heap, dim3((total + 255u) / 256u), dim3(256), 0, s, rec, lad, G, orbit_count, hrec, hlad);
    hipLaunchKernelGGL(k_bla_make_q, dim3((n_kmax + 255u) / 256u), dim3(256), 0, s, kmax, lad, G, lm2, hq, n_kmax); // 3 int4 each
    const uint32_t nz = orbit_count + 2u;
    hipLaunchKernelGGL(k_bla_make_zb, dim3((nz + 255u) / 256u), dim3(256), 0, s, zref, orbit_count, zb, nz);
}

void fsk_bla_hdr32_fast(const FsBlaArgsT<float> &A_in, bool pool, hipStream_t s)
{
    FsBlaArgsT<float> A = A_in;
#ifdef FS_BLA_FAST_PROBE
    if (getenv("FSMI355_BLA_STEPS_OUT"))
        A.probe_pitch = 0x57E9u;
#endif
    // pool (FS_VARIANT_BLA_POOL, A/B, off by default): the running pixels of a workgroup's four waves re-packed every 32 trips.
    // Measured on C5: 140.9 against 132.6 ms at 7680x4320, 39.4 against 37.4 ms at 3840x2160 -- the passes it saves (18 % of the
    // step passes if it were free and perfect, tools/c5_pooling_potential.py) cost less than the checkpoints and the mixed waves
    // (lanes of four tiles at unrelated orbit phases: more lookup and jump passes per trip, fewer all-quiet steps) add.
    const dim3 g((A.frame.width + 31) / 32, (A.frame.local_rows + 7) / 8, 1), b(256);
    if (pool)
        hipLaunchKernelGGL(k_bla_hdr32_fast<true>, g, b, 0, s, A);
    else
        hipLaunchKernelGGL(k_bla_hdr32_fast<false>, g, b, 0, s, A);
}
