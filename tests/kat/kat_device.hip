// kat_device.hip -- the known-answer stack machine in ONE gfx950 kernel: a full wave runs the same program in every lane
// (the kernels' tuned at_perform votes across the wave), lane 0 reports.  C ABI for ctypes:
// kat_run_device(program, n, checks, max) -> assertions evaluated, or a negative HIP error.
#include <hip/hip_runtime.h>

#include "kat_vm.hpp"

namespace {
__global__ void __launch_bounds__(64) k_kat(const kat::Instr *prog, int n, kat::Check *out, int max_checks, int *count,
                                            kat::Machine *machines)
{
    kat::Machine &m = machines[threadIdx.x];
    m.sp = 0, m.n_bla = 0, m.n_at = 0, m.n_res = 0;
    const int nc = m.run(prog, n, threadIdx.x == 0 ? out : nullptr, threadIdx.x == 0 ? max_checks : 0);
    if (threadIdx.x == 0)
        *count = nc;
}
} // namespace

extern "C" int kat_run_device(const kat::Instr *prog, int n, kat::Check *out, int max_checks)
{
    kat::Instr *dp = nullptr;
    kat::Check *dc = nullptr;
    kat::Machine *dm = nullptr;
    int *dn = nullptr;
    int nc = 0;
#define KAT_TRY(e)                                                                                                  \
    do {                                                                                                            \
        const hipError_t err_ = (e);                                                                                \
        if (err_ != hipSuccess)                                                                                     \
            return -(int)err_;                                                                                      \
    } while (0)
    KAT_TRY(hipMalloc(&dp, sizeof(kat::Instr) * (size_t)n));
    KAT_TRY(hipMalloc(&dc, sizeof(kat::Check) * (size_t)max_checks));
    KAT_TRY(hipMalloc(&dm, sizeof(kat::Machine) * 64));
    KAT_TRY(hipMalloc(&dn, sizeof(int)));
    KAT_TRY(hipMemcpy(dp, prog, sizeof(kat::Instr) * (size_t)n, hipMemcpyHostToDevice));
    KAT_TRY(hipMemset(dc, 0, sizeof(kat::Check) * (size_t)max_checks));
    KAT_TRY(hipMemset(dm, 0, sizeof(kat::Machine) * 64));
    hipLaunchKernelGGL(k_kat, dim3(1), dim3(64), 0, 0, dp, n, dc, max_checks, dn, dm);
    KAT_TRY(hipGetLastError());
    KAT_TRY(hipDeviceSynchronize());
    KAT_TRY(hipMemcpy(out, dc, sizeof(kat::Check) * (size_t)max_checks, hipMemcpyDeviceToHost));
    KAT_TRY(hipMemcpy(&nc, dn, sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(dp), (void)hipFree(dc), (void)hipFree(dm), (void)hipFree(dn);
    return nc;
}
