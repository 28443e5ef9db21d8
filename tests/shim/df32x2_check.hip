// GPU unit test of df32x2 (fractalshark_amd/csrc/df32_math.hpp): the packed pair operations must give, in each half, the
// bits of the scalar df32 operation on the corresponding operands -- the 2x32 AT loop and LA step rely on it.
// Random normalised double-floats over 60 binades, special cases (zeros, equal magnitudes, cancelling sums) included.
// Prints the number of mismatching results per operation; exit code 0 when all are zero.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../fractalshark_amd/csrc/df32_math.hpp"

using namespace fs;

__device__ inline bool same(df32 a, df32 b)
{
    return __float_as_uint(a.head) == __float_as_uint(b.head) && __float_as_uint(a.tail) == __float_as_uint(b.tail);
}

__global__ void k_check(const float4 *__restrict__ in, size_t n, unsigned long long *__restrict__ bad)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n)
        return;
    const float4 u = in[2 * i], v = in[2 * i + 1]; // two operand pairs: (a0, b0) = u, (a1, b1) = v
    const df32 a0(u.x, u.y), b0(u.z, u.w), a1(v.x, v.y), b1(v.z, v.w);
    const df32x2 A(a0, a1), B(b0, b1);
    const df32x2 S = A + B, P = A * B, D = A + B.neg_lo(), W = A.swapped();
    if (!same(S.lo(), a0 + b0) || !same(S.hi(), a1 + b1))
        atomicAdd(&bad[0], 1ull);
    if (!same(P.lo(), a0 * b0) || !same(P.hi(), a1 * b1))
        atomicAdd(&bad[1], 1ull);
    if (!same(D.lo(), a0 - b0) || !same(D.hi(), a1 + b1))
        atomicAdd(&bad[2], 1ull);
    if (!same(W.lo(), a1) || !same(W.hi(), a0))
        atomicAdd(&bad[3], 1ull);
    const df32x2 D2 = sub_lo_add_hi(A, B);
    if (!same(D2.lo(), a0 - b0) || !same(D2.hi(), a1 + b1))
        atomicAdd(&bad[4], 1ull);
    // mul_by_float: a * {2^k, 0} with the two fused operations on the zero tail left out == the full product
    {
        const float m0 = __builtin_amdgcn_ldexpf(1.0f, (int)((__float_as_uint(u.w) >> 2) % 121) - 120);
        const float m1 = __builtin_amdgcn_ldexpf(1.0f, (int)((__float_as_uint(v.y) >> 2) % 121) - 120);
        const df32x2 M = mul_by_float(A, (df32x2::f2){m0, m1});
        if (!same(M.lo(), a0 * df32(m0)) || !same(M.hi(), a1 * df32(m1)))
            atomicAdd(&bad[7], 1ull);
    }
    // hr_add2: two HDRFloat<CudaDblflt> additions (or a subtraction and an addition) side by side, against hr_add / hr_sub
    // on each pair.  Exponents from the operands' low mantissa bits: gaps of -130 .. 130, mostly small.  Where the packed
    // form reports `rare` nothing is claimed (the caller runs the literal code); everywhere else every bit must agree.
    const int32_t k = (int32_t)(__float_as_uint(u.y) >> 3), q = (int32_t)(__float_as_uint(v.w) >> 5);
    const int32_t ea0 = (k % 41) - 20, eb0 = ea0 + (((k >> 8) & 7) == 0 ? ((k >> 11) % 261) - 130 : ((k >> 11) % 49) - 24);
    const int32_t ea1 = (q % 33) - 16, eb1 = ea1 + (((q >> 8) & 7) == 0 ? ((q >> 11) % 261) - 130 : ((q >> 11) % 49) - 24);
    const hreal<df32> x0{a0, ea0}, y0{b0, eb0}, x1{a1, ea1}, y1{b1, eb1};
    for (int sub = 0; sub < 2; sub++) {
        bool rare = false;
        const hreal2 r = sub ? hr_add2<true>(hreal2(x0, x1), hreal2(y0, y1), rare) : hr_add2<false>(hreal2(x0, x1), hreal2(y0, y1), rare);
        if (rare) {
            atomicAdd(&bad[6], 1ull);
            continue;
        }
        const hreal<df32> w0 = sub ? hr_sub(x0, y0) : hr_add(x0, y0), w1 = hr_add(x1, y1);
        if (!same(r.x().m, w0.m) || r.ex != w0.e || !same(r.y().m, w1.m) || r.ey != w1.e)
            atomicAdd(&bad[5], 1ull);
    }
}

static void two_sum(float a, float b, float &s, float &e)
{
    s = a + b;
    const float bb = s - a;
    e = (a - (s - bb)) + (b - bb);
}

int main()
{
    const size_t n = 1 << 20;
    std::vector<float> h(4 * n);
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> mant(1.0, 2.0);
    std::uniform_int_distribution<int> ex(-30, 30), sgn(0, 1), kind(0, 15);
    auto mk = [&](float &head, float &tail) {
        const double x = (sgn(rng) ? -1.0 : 1.0) * std::ldexp(mant(rng), ex(rng));
        head = (float)x;
        tail = (float)(x - (double)head);
        float s, e;
        two_sum(head, tail, s, e); // normalised: |tail| <= ulp(head) / 2
        head = s, tail = e;
    };
    for (size_t i = 0; i < n; i++) {
        mk(h[4 * i], h[4 * i + 1]);
        mk(h[4 * i + 2], h[4 * i + 3]);
        switch (kind(rng)) {
        case 0: h[4 * i] = 0.0f, h[4 * i + 1] = 0.0f; break;                                   // zero operand
        case 1: h[4 * i + 2] = -h[4 * i], h[4 * i + 3] = -h[4 * i + 1]; break;                  // exact cancellation
        case 2: h[4 * i + 2] = h[4 * i], h[4 * i + 3] = h[4 * i + 1]; break;                    // equal operands
        case 3: h[4 * i + 2] = -h[4 * i], h[4 * i + 3] = h[4 * i + 1] * 0.5f; break;            // near cancellation
        case 4: h[4 * i + 1] = 0.0f; break;                                                     // no tail
        default: break;
        }
    }
    float4 *d_in;
    unsigned long long *d_bad, bad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMalloc((void **)&d_in, h.size() * sizeof(float)) != hipSuccess || hipMalloc((void **)&d_bad, sizeof(bad)) != hipSuccess)
        return 2;
    hipMemcpy(d_in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    hipMemset(d_bad, 0, sizeof(bad));
    hipLaunchKernelGGL(k_check, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, 0, d_in, n, d_bad);
    if (hipDeviceSynchronize() != hipSuccess)
        return 2;
    hipMemcpy(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
    printf("{\"pairs\": %zu, \"add_mismatch\": %llu, \"mul_mismatch\": %llu, \"sub_mismatch\": %llu, \"swap_mismatch\": %llu, "
           "\"sub_lo_add_hi_mismatch\": %llu, \"hr_add2_mismatch\": %llu, \"hr_add2_rare\": %llu, \"mul_by_float_mismatch\": %llu}\n",
           n / 2, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5], bad[6], bad[7]);
    return (bad[0] | bad[1] | bad[2] | bad[3] | bad[4] | bad[5] | bad[7]) ? 1 : 0;
}
