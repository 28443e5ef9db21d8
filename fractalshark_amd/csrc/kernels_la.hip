// kernels_la.hip -- the LAv2 table built ON THE DEVICE from the uploaded reference orbit (SURVEY.md section 8(f) row 1;
// reference: LAReference::GenerateApproximationData -> CreateLAFromOrbit / CreateNewLAStage / CreateATFromLA,
// FractalSharkLib/LAReference.cpp:28-210,774-966,971-1013,1050-1074, records LAInfoDeep.h:108-391).
//
// The reference builds every stage with one sequential scan: accumulate a record (Step / Composite) until the period
// detector fires or the stage's period cap is reached, push it, start the next one.  That scan is a state machine whose
// state at a segment boundary is (index b, flavour f) -- f says whether the new record starts from element b alone or
// from b combined with b+1 (the reference decides that with DetectPeriod on the element after the boundary) -- and, given
// the stage's period, EVERYTHING that decides where the segment (b, f) ends and which flavour follows is a running
// minimum of Chebyshev norms inside that segment (detection method 1, the default: LAInfoDeep.h:133-155,178-246,279-369).
// The coefficient products are not needed to find the boundaries.  Hence, per stage:
//   1. one lane per element: Chebyshev norm of the element's reference value, its MinMag, its step length;
//      an exclusive scan of the step lengths gives each element's orbit position;
//   2. the first detection of the stage's prologue (an uncapped scan from element 0: LAReference.cpp:92-134,811-852) as a
//      two-pass prefix-minimum over 1024 contiguous chunks -- the only long dependence of the algorithm;
//   3. one lane per (b, f): walk the segment (at most one period long), emit next(b, f);
//   4. the segments the sequential scan would actually visit are the chain x0 -> next(x0) -> ...: marked by pointer
//      doubling (log2 rounds over all states), ranked by an exclusive scan (chain order = index order);
//   5. one lane per marked segment: fold its elements with the reference's own Step / Composite (csrc/la_math.hpp, the
//      source the golden-pinned host builder compiles too) and write the record at its rank.
// The host (renderer.cpp, fs_build_la) keeps only the scalar decisions of LAReference.cpp (period from the prologue, the
// low-bound rules, when the stage loop stops) and reads a few words back per stage.  Result: the table of the reference's
// single-threaded builder, bit for bit.  Its multi-threaded stage-0 variant (CreateLAFromOrbitMT, :215-770) scans the orbit
// in pieces -- each piece a stretch of one of the chains x -> next(x) above, begun where a worker's two uncapped trackers
// first detect a period (k_la_first_from) and ended where it meets the next worker's published start -- and stitches them:
// fs_build_la_mt computes next() and the first detections here, walks and stitches the chains on the host (indices only) and
// folds the records of the resulting segment list with k_la_records_list.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fs_layout.h"
#include "hdr_math.hpp"
#include "kernels.h"
#include "la_math.hpp"

using namespace fs;
using namespace fs::la;

namespace {

constexpr uint32_t kTerm = 0xFFFFFFFFu;

template <class F> __device__ __forceinline__ hcplx<F> z_at(const void *zref, uint32_t i);
template <> __device__ __forceinline__ hcplx<float> z_at<float>(const void *zref, uint32_t i)
{
    const float4 v = ((const float4 *)zref)[i];
    return hcplx<float>{v.x, v.y, __float_as_int(v.z)};
}
template <> __device__ __forceinline__ hcplx<double> z_at<double>(const void *zref, uint32_t i)
{
    const FsZ64 v = ((const FsZ64 *)zref)[i];
    return hcplx<double>{v.re, v.im, v.e};
}

template <class F> __device__ __forceinline__ hreal<F> shifted(hreal<F> a, int exp2) { return hr_mul(a, hreal<F>{F(1), exp2}); }
template <class F> __device__ __forceinline__ bool lt(hreal<F> a, hreal<F> b) { return hr_cmp_pos(a, b) < 0; }

// ---- 1. per-element sources
template <class F> __global__ void k_la_src_orbit(const void *zref, uint32_t n, hreal<F> *chebv)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        chebv[i] = hc_cheb(z_at<F>(zref, i));
}
template <class F>
__global__ void k_la_src_stage(const LAInfo<F> *P, uint32_t n, hreal<F> *chebv, hreal<F> *mm, uint32_t *steps)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) {
        chebv[j] = hc_cheb(P[j].Ref);
        mm[j] = P[j].MinMag;
        steps[j] = P[j].StepLength;
    }
}

// exclusive prefix sum, one workgroup of 1024 lanes over contiguous chunks; out[n] = total
__device__ __forceinline__ void scan_u32_body(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *part)
{
    const uint32_t t = threadIdx.x, chunk = (n + 1023u) / 1024u;
    const uint32_t a = t * chunk, b = a + chunk < n ? a + chunk : n;
    uint32_t s = 0;
    for (uint32_t i = a; i < b; i++)
        s += in[i];
    part[t] = s;
    __syncthreads();
    // inclusive scan of the 1024 partials in ten rounds (round 4: lane 0 alone took 1024 dependent LDS round trips here --
    // 10 to 57 us of every stage of fs_build_la)
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        const uint32_t v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    if (t == 1023u)
        out[n] = part[t];
    uint32_t run = t == 0u ? 0u : part[t - 1u];
    for (uint32_t i = a; i < b; i++) {
        const uint32_t v = in[i];
        out[i] = run;
        run += v;
    }
}
__global__ void __launch_bounds__(1024) k_scan_u32(const uint32_t *in, uint32_t *out, uint32_t n)
{
    __shared__ uint32_t part[1024];
    scan_u32_body(in, out, n, part);
}

// What one element contributes to the running minimum once it has been folded into the record.
template <class F, bool kStage0> __device__ __forceinline__ hreal<F> contrib(const hreal<F> *chebv, const hreal<F> *mm, uint32_t j)
{
    if (kStage0)
        return chebv[j]; // Step: MinMag = min(cheb(z), MinMag)
    return hr_min_pos(chebv[j], mm[j]); // Composite: MinMag = min(min(cheb(Ref), MinMag), LA.MinMag)
}
// Running minimum of the record that starts at b with flavour f.
template <class F, bool kStage0> __device__ __forceinline__ hreal<F> start_min(const hreal<F> *chebv, const hreal<F> *mm, uint32_t b, uint32_t f)
{
    hreal<F> r = kStage0 ? hr_from_number<F>(F(4)) : mm[b]; // LAInfoDeep(z).MinMag = 4; a copied record keeps its own
    if (f)
        r = hr_min_pos(contrib<F, kStage0>(chebv, mm, b + 1), r);
    return r;
}

// ---- 2. first detection of the prologue: the uncapped scan from (base, 1), test elements base+2 .. limit-1 (base = 0 for
// a stage's prologue; the workers of the multi-threaded stage 0 begin elsewhere, see k_la_first_from).
// out[0] = index of the first detection (kTerm if none), out[1] = the flavour that follows it by the DetectPeriod rule.
template <class F, bool kStage0>
__device__ __forceinline__ void la_first_body(const hreal<F> *chebv, const hreal<F> *mm, uint32_t limit, int shift, uint32_t *out,
                                              hreal<F> *part, uint32_t &found, uint32_t base = 0u)
{
    const uint32_t t = threadIdx.x;
    const uint32_t n = limit > base + 2u ? limit - (base + 2u) : 0u, chunk = (n + 1023u) / 1024u;
    const uint32_t a0 = base + 2u + t * chunk, a = a0 < limit ? a0 : limit, b = (a + chunk < limit ? a + chunk : limit);
    const hreal<F> big = hreal<F>{F(1), 1 << 28};
    hreal<F> m = big;
    for (uint32_t j = a; j < b; j++)
        m = hr_min_pos(contrib<F, kStage0>(chebv, mm, j), m);
    part[t] = m;
    if (t == 0)
        found = kTerm;
    __syncthreads();
    // exclusive prefix minimum over the 1024 chunk minima, seeded with the record's start value: ten rounds instead of lane 0
    // walking all of them (round 4: that walk was 58 us of every stage).  min over a total order: associative, and equal
    // values are identical, so the order of the operands changes nothing
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        hreal<F> v = part[t];
        if (t >= off)
            v = hr_min_pos(part[t - off], v);
        __syncthreads();
        part[t] = v;
        __syncthreads();
    }
    {
        const hreal<F> start = start_min<F, kStage0>(chebv, mm, base, 1u);
        const hreal<F> excl = t == 0u ? start : hr_min_pos(part[t - 1u], start);
        __syncthreads();
        part[t] = excl;
    }
    __syncthreads();
    hreal<F> run = part[t];
    for (uint32_t j = a; j < b; j++) {
        if (lt(chebv[j], shifted(run, shift))) {
            atomicMin(&found, j);
            break;
        }
        run = hr_min_pos(contrib<F, kStage0>(chebv, mm, j), run);
    }
    __syncthreads();
    if (t == 0) {
        out[0] = found;
        out[1] = 0;
    }
    __syncthreads();
    // the flavour after the boundary needs the minimum INCLUDING the detecting element: recomputed by the lane that owns it
    if (found != kTerm && found >= a && found < b) {
        hreal<F> r2 = part[t];
        for (uint32_t j = a; j < found; j++)
            r2 = hr_min_pos(contrib<F, kStage0>(chebv, mm, j), r2);
        const hreal<F> nm = hr_min_pos(contrib<F, kStage0>(chebv, mm, found), r2);
        const bool detect = found + 1u < limit + 1u && lt(chebv[found + 1u], shifted(nm, -3)); // element limit exists (sentinel)
        out[1] = (detect || found + 1u >= limit) ? 0u : 1u;
    }
}
template <class F, bool kStage0>
__global__ void __launch_bounds__(1024) k_la_first(const hreal<F> *chebv, const hreal<F> *mm, uint32_t limit, int shift, uint32_t *out)
{
    __shared__ hreal<F> part[1024];
    __shared__ uint32_t found;
    la_first_body<F, kStage0>(chebv, mm, limit, shift, out, part, found);
}
// CreateLAFromOrbitMT's workers (LAReference.cpp:486-560): each begins at a fixed orbit index with two trackers one element apart
// and runs both, uncapped, until one detects a period.  A tracker is the scan from state (base, 1): one workgroup per tracker,
// out[q] = index of its first detection (kTerm: none before the end of the orbit).  Which tracker wins is the host's decision.
template <class F>
__global__ void __launch_bounds__(1024) k_la_first_from(const hreal<F> *chebv, const uint32_t *bases, uint32_t limit, int shift,
                                                        uint32_t *out)
{
    __shared__ hreal<F> part[1024];
    __shared__ uint32_t found;
    __shared__ uint32_t two[2];
    la_first_body<F, true>(chebv, nullptr, limit, shift, two, part, found, bases[blockIdx.x]);
    __syncthreads();
    if (threadIdx.x == 0)
        out[blockIdx.x] = found;
}
// A higher stage's prologue in ONE launch (round 4): the exclusive scan of the step lengths (orbit positions), the first
// detection, and the words the host decides the stage's period from next to it (the first element's step length, the orbit
// position of the detecting element, whether its record's LAThreshold is zero: LAReference.cpp:811-852) -- three launches and
// three read-backs before.
template <class F>
__global__ void __launch_bounds__(1024) k_la_stage_prologue(const LAInfo<F> *P, const hreal<F> *chebv, const hreal<F> *mm,
                                                            const uint32_t *steps, uint32_t *pos, uint32_t count, int shift,
                                                            uint32_t *out)
{
    __shared__ hreal<F> part[1024];
    __shared__ uint32_t upart[1024];
    __shared__ uint32_t found;
    scan_u32_body(steps, pos, count + 1u, upart);
    __syncthreads();
    la_first_body<F, false>(chebv, mm, count, shift, out, part, found);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t jd = found;
        out[2] = steps[0];
        out[3] = jd != kTerm ? pos[jd] : 0u;
        out[4] = jd != kTerm && P[jd].LAThreshold.m == F(0) ? 1u : 0u;
    }
}

// ---- 3. next(b, f) for every state; shift = the stage's in-loop detection threshold exponent (-6 stage 0, -3 above)
template <class F, bool kStage0>
__global__ void k_la_next(const hreal<F> *chebv, const hreal<F> *mm, const uint32_t *pos, uint32_t limit, uint32_t period,
                          int shift, uint32_t *next, uint32_t *reach, uint32_t x_start)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= 2u * limit)
        return;
    reach[x] = x == x_start ? 1u : 0u; // the chain's start mark (round 4: was a memset and a 4-byte upload per stage)
    const uint32_t b = x >> 1, f = x & 1u;
    if (f && b + 1u >= limit) {
        next[x] = kTerm; // never reached: flavour 1 needs the element after the boundary
        return;
    }
    hreal<F> run = start_min<F, kStage0>(chebv, mm, b, f);
    const uint64_t period_end = (uint64_t)(kStage0 ? b : pos[b]) + period;
    uint32_t j = b + 1u + f;
    uint32_t res = kTerm;
    for (; j < limit; j++) {
        const bool detected = lt(chebv[j], shifted(run, shift));
        if (detected || (uint64_t)(kStage0 ? j : pos[j]) >= period_end) {
            const hreal<F> nm = hr_min_pos(contrib<F, kStage0>(chebv, mm, j), run);
            const bool detect2 = lt(chebv[j + 1u], shifted(nm, -3)); // DetectPeriod(NewLA, element j + 1)
            res = 2u * j + ((detect2 || j + 1u >= limit) ? 0u : 1u);
            break;
        }
        run = hr_min_pos(contrib<F, kStage0>(chebv, mm, j), run);
    }
    next[x] = res;
}

// ---- 4. chain marking by pointer doubling: after round r every state within 2^(r+1) hops of the start is marked
// (marks are only ever set, and everything reachable from a marked state is on the chain, so marking in place is safe)
__global__ void k_la_reach(const uint32_t *jin, uint32_t *jout, uint32_t *reach, uint32_t nstates)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nstates)
        return;
    const uint32_t j = jin[x];
    if (reach[x] && j != kTerm)
        reach[j] = 1u; // idempotent stores from several lanes
    jout[x] = j == kTerm ? kTerm : jin[j];
}

// The same for a stage of at most 2^16 states, all rounds in ONE launch (round 4): one workgroup walks the states in strides,
// a barrier between rounds (a workgroup-scope fence: the waves of a workgroup share their CU's L1, what one round stored is
// what the next one loads).  The launches of the per-round form -- fourteen for View 5's first stage, ~9 us apart -- were
// most of what fs_build_la took on a small orbit.  Returns with the final jump table in whichever buffer the last round wrote.
__global__ void __launch_bounds__(1024) k_la_reach_all(const uint32_t *next, uint32_t *bufB, uint32_t *bufC, uint32_t *reach,
                                                       uint32_t nstates, uint32_t rounds)
{
    const uint32_t *jin = next;
    uint32_t *jout = bufB;
    for (uint32_t r = 0; r < rounds; r++) {
        for (uint32_t x = threadIdx.x; x < nstates; x += 1024u) {
            const uint32_t j = jin[x];
            if (reach[x] && j != kTerm)
                reach[j] = 1u;
            jout[x] = j == kTerm ? kTerm : jin[j];
        }
        __syncthreads();
        jin = jout;
        jout = jout == bufB ? bufC : bufB;
    }
}

// A few words to the host without a stream synchronisation (round 4): `mail` is page-locked, coherent host memory that the
// device writes directly; the words first, a system-scope fence, then the sequence number the host is spinning on.  A
// hipStreamSynchronize behind a 4-byte copy costs 20 - 30 us of interrupt latency, and fs_build_la needs two such reads per
// stage: on a small orbit they were half of its time.
__global__ void k_la_mail(const uint32_t *src, uint32_t n, volatile uint32_t *mail, uint32_t seq)
{
    if (blockIdx.x != 0 || threadIdx.x != 0)
        return;
    for (uint32_t i = 0; i < n; i++)
        mail[i] = src[i];
    __threadfence_system();
    mail[31] = seq;
}

// ---- 5. records
template <class F, bool kStage0>
__global__ void k_la_records(const void *zref, const LAInfo<F> *P, const uint32_t *pos, const uint32_t *next,
                             const uint32_t *reach, const uint32_t *rank, uint32_t limit, uint32_t nstates,
                             uint32_t rank_offset, LAInfo<F> *out, LAInfo<F> *tail_out, uint32_t max_ref)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const LAParams p{};
    if (x == 0u && tail_out) // the stage's tail record (k_la_tail), in the same launch
        *tail_out = la_init<F>(p, z_at<F>(zref, max_ref));
    if (x >= nstates || !reach[x])
        return;
    const uint32_t b = x >> 1;
    const uint32_t e = next[x] == kTerm ? limit : next[x] >> 1;
    LAInfo<F> LA;
    if (kStage0) {
        LA = la_init<F>(p, z_at<F>(zref, b));
        for (uint32_t t = b + 1u; t < e; t++)
            LA = la_step_new<F>(p, LA, z_at<F>(zref, t));
        LA.StepLength = e - b;
    } else {
        LA = P[b];
        for (uint32_t t = b + 1u; t < e; t++)
            LA = la_composite_new<F>(p, LA, P[t]);
        LA.StepLength = pos[e] - pos[b];
    }
    LA.NextStageLAIndex = b;
    out[rank_offset + rank[x]] = LA;
}

// ... and from an explicit list of segments [b, e) of the orbit (the stitched chains of the multi-threaded stage 0: the host has
// walked them, see build_la in renderer.cpp): record k = init(z[b]) stepped through z[b+1 .. e-1], StepLength e - b.
template <class F>
__global__ void k_la_records_list(const void *zref, const uint32_t *seg, uint32_t n, LAInfo<F> *out, LAInfo<F> *tail_out,
                                  uint32_t max_ref)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const LAParams p{};
    if (k == 0u && tail_out)
        *tail_out = la_init<F>(p, z_at<F>(zref, max_ref));
    if (k >= n)
        return;
    const uint32_t b = seg[2u * k], e = seg[2u * k + 1u];
    LAInfo<F> LA = la_init<F>(p, z_at<F>(zref, b));
    for (uint32_t t = b + 1u; t < e; t++)
        LA = la_step_new<F>(p, LA, z_at<F>(zref, t));
    LA.StepLength = e - b;
    LA.NextStageLAIndex = b;
    out[k] = LA;
}

// one explicit segment [0, e) (the prologue's first record, or the single record of the last stage)
template <class F, bool kStage0>
__global__ void k_la_one_record(const void *zref, const LAInfo<F> *P, uint32_t e, uint32_t step_length, LAInfo<F> *out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0)
        return;
    const LAParams p{};
    LAInfo<F> LA;
    if (kStage0) {
        LA = la_init<F>(p, z_at<F>(zref, 0));
        for (uint32_t t = 1u; t < e; t++)
            LA = la_step_new<F>(p, LA, z_at<F>(zref, t));
    } else {
        LA = P[0];
        for (uint32_t t = 1u; t < e; t++)
            LA = la_composite_new<F>(p, LA, P[t]);
    }
    LA.StepLength = step_length;
    LA.NextStageLAIndex = 0;
    *out = LA;
}

// the record every stage ends with: LAInfoDeep(z[maxRef]); also answers isZCoeffZero of the very first step
template <class F> __global__ void k_la_tail(const void *zref, uint32_t max_ref, LAInfo<F> *out, uint32_t *zcoeff_zero)
{
    if (blockIdx.x != 0 || threadIdx.x != 0)
        return;
    const LAParams p{};
    if (out)
        *out = la_init<F>(p, z_at<F>(zref, max_ref));
    if (zcoeff_zero) {
        const LAInfo<F> first = la_step_new<F>(p, la_init<F>(p, mk<F>::czero()), z_at<F>(zref, 1));
        *zcoeff_zero = (first.ZCoeff.re == F(0) && first.ZCoeff.im == F(0)) ? 1u : 0u;
    }
}

// CreateATFromLA, LAReference.cpp:1050-1074: last stage first
template <class F>
__global__ void k_la_at(const LAInfo<F> *las, const uint32_t *stage_la_index, uint32_t stage_count, hreal<F> radius,
                        int use_small_exponents, ATInfoT<F> *at_out, uint32_t *use_at)
{
    if (blockIdx.x != 0 || threadIdx.x != 0)
        return;
    const hreal<F> SqrRadius = hr_reduced(hr_square(radius));
    ATInfoT<F> at;
    *use_at = 0;
    for (uint32_t Stage = stage_count; Stage > 0;) {
        Stage--;
        const uint32_t LAIndex = stage_la_index[Stage];
        la_create_at<F>(las[LAIndex], at, las[LAIndex + 1], use_small_exponents != 0);
        at.StepLength = las[LAIndex].StepLength;
        if (at.StepLength > 0 && at_usable<F>(at, SqrRadius)) {
            *use_at = 1;
            break;
        }
    }
    *at_out = at;
}

// LAInfo<F> -> the ABI record the render kernels read (LAInfoDeep layout, fs_layout.h)
__global__ void k_la_pack32(const LAInfo<float> *in, fs_la_hdr32_u32 *out, uint32_t n)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    const LAInfo<float> s = in[k];
    fs_la_hdr32_u32 r;
    r.Ref = fs_cplx_hdr32{s.Ref.re, s.Ref.im, s.Ref.e};
    r.ZCoeff = fs_cplx_hdr32{s.ZCoeff.re, s.ZCoeff.im, s.ZCoeff.e};
    r.CCoeff = fs_cplx_hdr32{s.CCoeff.re, s.CCoeff.im, s.CCoeff.e};
    r.LAThreshold = fs_real_hdr32{s.LAThreshold.m, s.LAThreshold.e};
    r.LAThresholdC = fs_real_hdr32{s.LAThresholdC.m, s.LAThresholdC.e};
    r.MinMag = fs_real_hdr32{s.MinMag.m, s.MinMag.e};
    r.StepLength = s.StepLength;
    r.NextStageLAIndex = s.NextStageLAIndex;
    out[k] = r;
}
__global__ void k_la_pack64(const LAInfo<double> *in, fs_la_hdr64_u32 *out, uint32_t n)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    const LAInfo<double> s = in[k];
    fs_la_hdr64_u32 r;
    memset(&r, 0, sizeof(r));
    r.Ref = fs_cplx_hdr64{s.Ref.re, s.Ref.im, s.Ref.e, 0};
    r.ZCoeff = fs_cplx_hdr64{s.ZCoeff.re, s.ZCoeff.im, s.ZCoeff.e, 0};
    r.CCoeff = fs_cplx_hdr64{s.CCoeff.re, s.CCoeff.im, s.CCoeff.e, 0};
    r.LAThreshold = fs_real_hdr64{s.LAThreshold.m, s.LAThreshold.e, 0};
    r.LAThresholdC = fs_real_hdr64{s.LAThresholdC.m, s.LAThresholdC.e, 0};
    r.MinMag = fs_real_hdr64{s.MinMag.m, s.MinMag.e, 0};
    r.StepLength = s.StepLength;
    r.NextStageLAIndex = s.NextStageLAIndex;
    out[k] = r;
}

inline unsigned nblk(uint32_t n) { return (n + 255u) / 256u; }

} // namespace

// ------------------------------------------------------------------------------------------------ launch entry points
template <class F> struct LaDev {
    using Rec = LAInfo<F>;
};

template <class F> void fsk_la_src_orbit(const void *zref, uint32_t n, void *chebv, hipStream_t s)
{
    hipLaunchKernelGGL((k_la_src_orbit<F>), dim3(nblk(n)), dim3(256), 0, s, zref, n, (hreal<F> *)chebv);
}
template <class F> void fsk_la_src_stage(const void *P, uint32_t n, void *chebv, void *mm, uint32_t *steps, hipStream_t s)
{
    hipLaunchKernelGGL((k_la_src_stage<F>), dim3(nblk(n)), dim3(256), 0, s, (const LAInfo<F> *)P, n, (hreal<F> *)chebv,
                       (hreal<F> *)mm, steps);
}
void fsk_scan_u32(const uint32_t *in, uint32_t *out, uint32_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, s, in, out, n);
}
template <class F> void fsk_la_first(bool stage0, const void *chebv, const void *mm, uint32_t limit, uint32_t *out, hipStream_t s)
{
    if (stage0)
        hipLaunchKernelGGL((k_la_first<F, true>), dim3(1), dim3(1024), 0, s, (const hreal<F> *)chebv, (const hreal<F> *)mm,
                           limit, LAParams{}.stage0PeriodDetectionThreshold2Exp, out);
    else
        hipLaunchKernelGGL((k_la_first<F, false>), dim3(1), dim3(1024), 0, s, (const hreal<F> *)chebv,
                           (const hreal<F> *)mm, limit, LAParams{}.periodDetectionThreshold2Exp, out);
}
template <class F>
void fsk_la_first_from(const void *chebv, const uint32_t *bases, uint32_t n_queries, uint32_t limit, uint32_t *out, hipStream_t s)
{
    hipLaunchKernelGGL((k_la_first_from<F>), dim3(n_queries), dim3(1024), 0, s, (const hreal<F> *)chebv, bases, limit,
                       LAParams{}.stage0PeriodDetectionThreshold2Exp, out);
}
template <class F>
void fsk_la_records_list(const void *zref, const uint32_t *seg, uint32_t n, void *out, void *tail_out, uint32_t max_ref,
                         hipStream_t s)
{
    hipLaunchKernelGGL((k_la_records_list<F>), dim3(nblk(n ? n : 1u)), dim3(256), 0, s, zref, seg, n, (LAInfo<F> *)out,
                       (LAInfo<F> *)tail_out, max_ref);
}
template <class F>
void fsk_la_next(bool stage0, const void *chebv, const void *mm, const uint32_t *pos, uint32_t limit, uint32_t period,
                 uint32_t *next, uint32_t *reach, uint32_t x_start, hipStream_t s)
{
    if (stage0)
        hipLaunchKernelGGL((k_la_next<F, true>), dim3(nblk(2u * limit)), dim3(256), 0, s, (const hreal<F> *)chebv,
                           (const hreal<F> *)mm, pos, limit, period, LAParams{}.stage0PeriodDetectionThreshold2Exp, next, reach,
                           x_start);
    else
        hipLaunchKernelGGL((k_la_next<F, false>), dim3(nblk(2u * limit)), dim3(256), 0, s, (const hreal<F> *)chebv,
                           (const hreal<F> *)mm, pos, limit, period, LAParams{}.periodDetectionThreshold2Exp, next, reach, x_start);
}
void fsk_la_reach(const uint32_t *jin, uint32_t *jout, uint32_t *reach, uint32_t nstates, hipStream_t s)
{
    hipLaunchKernelGGL(k_la_reach, dim3(nblk(nstates)), dim3(256), 0, s, jin, jout, reach, nstates);
}
void fsk_la_mail(const uint32_t *src, uint32_t n, uint32_t *mail, uint32_t seq, hipStream_t s)
{
    hipLaunchKernelGGL(k_la_mail, dim3(1), dim3(1), 0, s, src, n, (volatile uint32_t *)mail, seq);
}
void fsk_la_reach_all(const uint32_t *next, uint32_t *bufB, uint32_t *bufC, uint32_t *reach, uint32_t nstates, uint32_t rounds,
                      hipStream_t s)
{
    hipLaunchKernelGGL(k_la_reach_all, dim3(1), dim3(1024), 0, s, next, bufB, bufC, reach, nstates, rounds);
}
template <class F>
void fsk_la_records(bool stage0, const void *zref, const void *P, const uint32_t *pos, const uint32_t *next,
                    const uint32_t *reach, const uint32_t *rank, uint32_t limit, uint32_t rank_offset, void *out, void *tail_out,
                    uint32_t max_ref, hipStream_t s)
{
    const uint32_t nstates = 2u * limit;
    if (stage0)
        hipLaunchKernelGGL((k_la_records<F, true>), dim3(nblk(nstates)), dim3(256), 0, s, zref, (const LAInfo<F> *)P, pos,
                           next, reach, rank, limit, nstates, rank_offset, (LAInfo<F> *)out, (LAInfo<F> *)tail_out, max_ref);
    else
        hipLaunchKernelGGL((k_la_records<F, false>), dim3(nblk(nstates)), dim3(256), 0, s, zref, (const LAInfo<F> *)P, pos,
                           next, reach, rank, limit, nstates, rank_offset, (LAInfo<F> *)out, (LAInfo<F> *)tail_out, max_ref);
}
template <class F>
void fsk_la_stage_prologue(const void *P, const void *chebv, const void *mm, const uint32_t *steps, uint32_t *pos, uint32_t count,
                           uint32_t *out, hipStream_t s)
{
    hipLaunchKernelGGL((k_la_stage_prologue<F>), dim3(1), dim3(1024), 0, s, (const LAInfo<F> *)P, (const hreal<F> *)chebv,
                       (const hreal<F> *)mm, steps, pos, count, LAParams{}.periodDetectionThreshold2Exp, out);
}
template <class F>
void fsk_la_one_record(bool stage0, const void *zref, const void *P, uint32_t e, uint32_t step_length, void *out, hipStream_t s)
{
    if (stage0)
        hipLaunchKernelGGL((k_la_one_record<F, true>), dim3(1), dim3(64), 0, s, zref, (const LAInfo<F> *)P, e, step_length,
                           (LAInfo<F> *)out);
    else
        hipLaunchKernelGGL((k_la_one_record<F, false>), dim3(1), dim3(64), 0, s, zref, (const LAInfo<F> *)P, e, step_length,
                           (LAInfo<F> *)out);
}
template <class F> void fsk_la_tail(const void *zref, uint32_t max_ref, void *out, uint32_t *zcoeff_zero, hipStream_t s)
{
    hipLaunchKernelGGL((k_la_tail<F>), dim3(1), dim3(64), 0, s, zref, max_ref, (LAInfo<F> *)out, zcoeff_zero);
}
template <class F>
void fsk_la_at(const void *las, const uint32_t *stage_la_index, uint32_t stage_count, const void *radius,
               int use_small_exponents, void *at_out, uint32_t *use_at, hipStream_t s)
{
    hipLaunchKernelGGL((k_la_at<F>), dim3(1), dim3(64), 0, s, (const LAInfo<F> *)las, stage_la_index, stage_count,
                       *(const hreal<F> *)radius, use_small_exponents, (ATInfoT<F> *)at_out, use_at);
}
void fsk_la_pack(bool is64, const void *in, void *out, uint32_t n, hipStream_t s)
{
    if (is64)
        hipLaunchKernelGGL(k_la_pack64, dim3(nblk(n)), dim3(256), 0, s, (const LAInfo<double> *)in, (fs_la_hdr64_u32 *)out, n);
    else
        hipLaunchKernelGGL(k_la_pack32, dim3(nblk(n)), dim3(256), 0, s, (const LAInfo<float> *)in, (fs_la_hdr32_u32 *)out, n);
}

#define FS_LA_INSTANTIATE(F)                                                                                        \
    template void fsk_la_src_orbit<F>(const void *, uint32_t, void *, hipStream_t);                                 \
    template void fsk_la_src_stage<F>(const void *, uint32_t, void *, void *, uint32_t *, hipStream_t);             \
    template void fsk_la_first<F>(bool, const void *, const void *, uint32_t, uint32_t *, hipStream_t);             \
    template void fsk_la_first_from<F>(const void *, const uint32_t *, uint32_t, uint32_t, uint32_t *, hipStream_t); \
    template void fsk_la_records_list<F>(const void *, const uint32_t *, uint32_t, void *, void *, uint32_t, hipStream_t); \
    template void fsk_la_next<F>(bool, const void *, const void *, const uint32_t *, uint32_t, uint32_t, uint32_t *, \
                                 uint32_t *, uint32_t, hipStream_t);                                                \
    template void fsk_la_records<F>(bool, const void *, const void *, const uint32_t *, const uint32_t *,            \
                                    const uint32_t *, const uint32_t *, uint32_t, uint32_t, void *, void *, uint32_t,  \
                                    hipStream_t);                                                                   \
    template void fsk_la_stage_prologue<F>(const void *, const void *, const void *, const uint32_t *, uint32_t *,   \
                                           uint32_t, uint32_t *, hipStream_t);                                      \
    template void fsk_la_one_record<F>(bool, const void *, const void *, uint32_t, uint32_t, void *, hipStream_t);  \
    template void fsk_la_tail<F>(const void *, uint32_t, void *, uint32_t *, hipStream_t);                          \
    template void fsk_la_at<F>(const void *, const uint32_t *, uint32_t, const void *, int, void *, uint32_t *, hipStream_t);
FS_LA_INSTANTIATE(float)
FS_LA_INSTANTIATE(double)
#undef FS_LA_INSTANTIATE
