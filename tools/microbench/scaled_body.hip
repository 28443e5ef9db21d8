// What bounds the scaled quiet step of k_lav2_hdr32_fast?  The step's instruction stream as it is generated (vector ALU,
// one 12-byte load from an L2-resident table, the wait, the scalar ORs and the branch), with pieces removed one at a time.
// Build: hipcc --offload-arch=gfx950 -O3 -o scaled_body scaled_body.hip ; run on the box (8 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f3 __attribute__((ext_vector_type(3)));

// VARIANT bits: 1 = vector loads, 2 = scalar ORs + compare + branch, 4 = checks (max/add/cmp ...), 8 = per-lane different
// addresses, 16 = entries through the scalar cache (four s_load_dwordx4 per body, one wait) instead of vector loads
template <int V> __global__ void __launch_bounds__(256) k(const float4 *__restrict__ tab, float *out, int iters, int n)
{
    f2 w = {1.0f + threadIdx.x * 1e-3f, 0.5f}, z = {0.3f, -0.2f};
    const f2 sE2 = {1e-3f, 1e-3f}, dcs = {1e-4f, 2e-4f};
    const int Esh = -(10 << 23);
    uint32_t lane_off = (V & 8) ? ((threadIdx.x * 37u) % 1024u) * 16u : 16u;
    uint32_t c = 0;
    const float4 *zp = tab + (blockIdx.x % 7) * 64;
    const float4 *zp0 = zp;
    uint64_t acc = 0;
    for (int it = 0; it < iters; it++) {
#define STEP(W_, Z_, NW_, NZ_, OFS, FULL)                                                                               \
    {                                                                                                                   \
        f3 ent = {Z_.x, Z_.y, 1.0f};                                                                                    \
        if (V & 1)                                                                                                      \
            asm volatile("global_load_dwordx3 %0, %2, %3 offset:" OFS : "=v"(ent), "+v"(W_) : "v"(lane_off), "s"(zp));  \
        const f2 s_ = __builtin_elementwise_fma(W_, sE2, Z_);                                                           \
        const f2 pa_ = W_.xx * s_;                                                                                      \
        const f2 pb_ = W_.yy * s_.yx;                                                                                   \
        f2 p_;                                                                                                          \
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(p_) : "v"(pa_), "v"(pb_));                       \
        NW_ = p_ + dcs;                                                                                                 \
        float mx = __builtin_fmaxf(__builtin_fabsf(NW_.x), __builtin_fabsf(NW_.y));                                     \
        if (V & 1)                                                                                                      \
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(ent), "+v"(mx));                                                   \
        NZ_ = (f2){ent.x, ent.y};                                                                                       \
        if (V & 4) {                                                                                                    \
            viol |= __builtin_amdgcn_ballot_w64(__float_as_int(mx) + Esh > __float_as_int(ent.z));                      \
            if (FULL) {                                                                                                 \
                const float mn = __builtin_fminf(__builtin_fabsf(NW_.x), __builtin_fabsf(NW_.y));                       \
                viol |= __builtin_amdgcn_ballot_w64(!(mn >= mx * 0x1p-40f)) |                                           \
                        __builtin_amdgcn_ballot_w64((uint32_t)(__float_as_int(mx) - (7 << 23)) >= (uint32_t)(240 << 23)); \
            }                                                                                                           \
        }                                                                                                               \
    }
        f2 t1, u1, w2, z2, t3, u3;
        uint64_t viol = 0;
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 ua, ub, uc, ud;
        if (V & 16) {
            asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(ua) : "s"(zp));
            asm volatile("s_load_dwordx4 %0, %1, 0x10" : "=s"(ub) : "s"(zp));
            asm volatile("s_load_dwordx4 %0, %1, 0x20" : "=s"(uc) : "s"(zp));
            asm volatile("s_load_dwordx4 %0, %1, 0x30" : "=s"(ud) : "s"(zp));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ua), "+s"(ub), "+s"(uc), "+s"(ud), "+v"(w));
            z = (f2){z.x + ua.x * 1e-30f, z.y + ub.y * 1e-30f + uc.x * 1e-30f + ud.x * 1e-30f};
        }
        STEP(w, z, t1, u1, "0", false)
        STEP(t1, u1, w2, z2, "16", true)
        if ((V & 2) && viol != 0ull) {
            acc += viol;
            break;
        }
        STEP(w2, z2, t3, u3, "32", false)
        STEP(t3, u3, w, z, "48", true)
        if ((V & 2) && viol != 0ull) {
            acc += viol;
            break;
        }
        if (!(V & 2))
            acc |= viol;
        c += 4;
        zp += 4;
        if (c >= 16384u) { // walk 256 KB so that scalar-cache lines are new ones, as in the kernel
            c = 0;
            zp = zp0;
            // keep the values in range
            w = (f2){1.0f + w.x * 1e-30f, 0.5f + w.y * 1e-30f};
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = w.x + w.y + z.x + z.y + (float)acc;
}

template <int V> void run(const char *name, int waves_per_simd = 8)
{
    float *out;
    float4 *tab;
    const int blocks = 256 * waves_per_simd, n = 1 << 16;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&tab, n * sizeof(float4));
    float4 *h = new float4[n];
    for (int i = 0; i < n; i++)
        h[i] = make_float4(0.3f + 1e-4f * (i % 17), -0.2f, 1.0f, 0.0f);
    hipMemcpy(tab, h, n * sizeof(float4), hipMemcpyHostToDevice);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, tab, out, 10, n);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, tab, out, iters, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double steps_per_simd = (double)iters * 4 * waves_per_simd; // 4 steps per trip
    printf("{\"variant\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"ns_per_wave_step_per_simd\": %.2f, "
           "\"cycles_at_2.4GHz\": %.1f}\n",
           name, waves_per_simd, ms, ms * 1e6 / steps_per_simd, ms * 1e-3 * 2.4e9 / steps_per_simd);
    hipFree(out);
    hipFree(tab);
    delete[] h;
}

int main()
{
    run<0>("arithmetic only (5 packed + max)");
    run<4>("+ checks");
    run<6>("+ checks + scalar OR / branch");
    run<5>("+ checks + loads (same address in every lane)");
    run<7>("everything, same address in every lane");
    run<15>("everything, per-lane addresses");
    run<9>("arithmetic + loads, per-lane addresses");
    run<22>("checks + scalar OR / branch + scalar-cache entries");
    // one wave per SIMD: the dependent chain, nothing to hide it behind
    run<0>("arithmetic only (5 packed + max)", 1);
    run<4>("+ checks", 1);
    run<6>("+ checks + scalar OR / branch", 1);
    run<7>("everything, vector loads, same address in every lane", 1);
    run<22>("checks + scalar OR / branch + scalar-cache entries", 1);
    return 0;
}
