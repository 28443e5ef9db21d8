// Lone-wave latency probe for gfx950: what one wave that is alone on its SIMD pays per instruction -- the regime that
// bounds C2 (interior pixels' 4.7 M-step chains) and every kernel's drain.  Each kind runs with 1 wave per SIMD and, for
// comparison, 8.  Time per repetition from the 100 MHz wall clock read inside the kernel (lane 0 of block 0) and from HIP
// events.  Build: hipcc --offload-arch=gfx950 -O3 -o lone_wave lone_wave.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(X) X X X X
#define REP16(X) REP4(REP4(X))
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND> __global__ void __launch_bounds__(256) k(float *out, int iters, unsigned long long *ticks)
{
    f2 w = {1.0f + threadIdx.x * 1e-3f, 0.5f}, z = {0.25f, -0.125f}, sE = {0x1p-20f, 0x1p-20f}, dc = {1e-3f, 2e-3f};
    f2 p0 = w, p1 = z, p2 = dc, p3 = sE;
    float a0 = w.x, a1 = w.y, a2 = z.x, a3 = z.y;
    int esh = -(20 << 23), i0 = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) { // 16 dependent v_pk_fma_f32
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(p3), "v"(p1));)
        } else if (KIND == 1) { // 16 v_pk_fma_f32 in 4 independent chains
            REP4(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(w) : "v"(p3), "v"(z));)
        } else if (KIND == 2) { // 16 dependent v_add_f32
            REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(a1));)
        } else if (KIND == 3) { // 16 v_add_f32 in 4 independent chains
            REP4(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w.x));)
        } else if (KIND == 4) { // 16 x the step's arithmetic chain alone: fma -> 2 mul -> add(neg) -> add   (5 instr, 4 levels)
            REP16(asm volatile("v_pk_fma_f32 %1, %0, %3, %2\n"
                               "v_pk_mul_f32 %4, %0, %1 op_sel_hi:[0,1]\n v_pk_mul_f32 %1, %0, %1 op_sel:[1,1] op_sel_hi:[1,0]\n"
                               "s_nop 0\n v_pk_add_f32 %1, %4, %1 neg_lo:[0,1] neg_hi:[0,0]\n s_nop 0\n v_pk_add_f32 %0, %5, %1\n s_nop 0"
                               : "+v"(w), "+v"(p0) : "v"(z), "v"(sE), "v"(p1), "v"(dc));)
        } else if (KIND == 5 || KIND == 6) { // 8 trips of the kernel's loop (two steps + tests + votes + branch), in C++ like the
                                             // kernel; KIND 6 branches on a trip's tests one trip late (they overlap the next chain)
            uint64_t vprev = 0;
#define STEP(W_, Z_, NW_, T, V, FULL, EB)                                                                            \
    const f2 s_##T = __builtin_elementwise_fma(W_, sE, Z_);                                                          \
    const f2 pa_##T = W_.xx * s_##T;                                                                                 \
    const f2 pb_##T = W_.yy * s_##T.yx;                                                                              \
    f2 p_##T;                                                                                                        \
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(p_##T) : "v"(pa_##T), "v"(pb_##T));               \
    NW_ = p_##T + dc;                                                                                                \
    float mx_##T = __builtin_fmaxf(__builtin_fabsf(NW_.x), __builtin_fabsf(NW_.y));                                  \
    V |= __builtin_amdgcn_ballot_w64(__float_as_int(mx_##T) + esh > EB);                                             \
    if (FULL) {                                                                                                      \
        const float mn_##T = __builtin_fminf(__builtin_fabsf(NW_.x), __builtin_fabsf(NW_.y));                        \
        V |= __builtin_amdgcn_ballot_w64(!(mn_##T >= mx_##T * 0x1p-40f)) |                                           \
             __builtin_amdgcn_ballot_w64((uint32_t)(__float_as_int(mx_##T) - (7 << 23)) >= (uint32_t)(240 << 23));   \
    }
#define TRIP(T1, T2)                                                                                                 \
    {                                                                                                                \
        f2 t_, n_;                                                                                                   \
        uint64_t v = 0;                                                                                              \
        STEP(w, z, t_, T1, v, false, eb)                                                                             \
        STEP(t_, z, n_, T2, v, true, eb)                                                                             \
        if (KIND == 5) {                                                                                             \
            if (v != 0ull)                                                                                           \
                break;                                                                                               \
        } else {                                                                                                     \
            if (vprev != 0ull)                                                                                       \
                break;                                                                                               \
            vprev = v;                                                                                               \
        }                                                                                                            \
        w = n_;                                                                                                      \
    }
            const int eb = 0x7f000000 + (it & 1);
            TRIP(a, b) TRIP(c, d) TRIP(e, f) TRIP(g, h) TRIP(i, j) TRIP(k_, l) TRIP(m, n) TRIP(o, p)
            // keep the iterates bounded so that no test fires: fold the state back
            w = (f2){1.0f, 0.5f} + w * 0.0f;
        } else if (KIND == 7) { // 16 x (v_add_f32 dependent + s_add_u32): does scalar work ride along in a lone wave?
            REP16(asm volatile("v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1" : "+v"(a0) : "v"(a1) : "s20", "scc");)
        } else if (KIND == 8) { // 16 dependent non-packed fma: v_fma_f32
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2));)
        } else if (KIND == 9) { // the step with NON-packed arithmetic: 2 fma, 4 mul, 2 add/sub, 2 add  (10 instr, 4 levels)
            REP16(asm volatile("v_fma_f32 %4, %0, %8, %2\n v_fma_f32 %5, %1, %8, %3\n"
                               "v_mul_f32 %6, %0, %4\n v_mul_f32 %7, %1, %5\n v_mul_f32 %4, %1, %4\n v_mul_f32 %5, %0, %5\n"
                               "v_sub_f32 %6, %6, %7\n v_add_f32 %4, %5, %4\n v_add_f32 %0, %9, %6\n v_add_f32 %1, %10, %4"
                               : "+v"(a0), "+v"(a1)
                               : "v"(z.x), "v"(z.y), "v"(p0.x), "v"(p0.y), "v"(p1.x), "v"(p1.y), "v"(sE.x), "v"(dc.x), "v"(dc.y));)
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *ticks = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p2.x + w.x + w.y + i0;
}

template <int KIND> void run(const char *name, int insts_per_iter, int steps_per_iter)
{
    float *out;
    unsigned long long *ticks;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipHostMalloc((void **)&ticks, 8);
    for (int waves_per_simd : {1, 8}) {
        const int blocks = 256 * waves_per_simd;
        const int iters = waves_per_simd == 1 ? 20000 : 4000;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100, ticks);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, ticks);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double ns_wave = (double)*ticks * 10.0 / iters; // 100 MHz wall clock
        printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ns_per_iter_wallclock\": %.2f, \"ns_per_iter_events\": %.2f, "
               "\"ns_per_instruction\": %.3f%s%.2f}\n",
               name, waves_per_simd, ns_wave, ms * 1e6 / iters, ns_wave / insts_per_iter,
               steps_per_iter ? ", \"ns_per_step\": " : ", \"_\": ", steps_per_iter ? ns_wave / steps_per_iter : 0.0);
    }
    hipFree(out);
    hipHostFree(ticks);
}

int main()
{
    run<0>("v_pk_fma_f32 dependent x16", 16, 0);
    run<1>("v_pk_fma_f32 4 chains x16", 16, 0);
    run<2>("v_add_f32 dependent x16", 16, 0);
    run<3>("v_add_f32 4 chains x16", 16, 0);
    run<8>("v_fma_f32 dependent x16", 16, 0);
    run<7>("v_add_f32 dependent + s_add_u32 x16", 32, 0);
    run<4>("step arithmetic only (5 packed + 3 s_nop) x16", 16 * 8, 16);
    run<9>("step arithmetic non-packed (10 VOP2/VOP3) x16", 16 * 10, 16);
    run<5>("kernel trip (2 steps + tests + votes + branch) x8", 8 * 28, 16);
    run<6>("kernel trip, branch one trip late x8", 8 * 28, 16);
    return 0;
}
