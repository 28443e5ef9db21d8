// Compile/link check of fractalshark_amd/csrc/gpu_render_shim.hpp WITHOUT the reference tree: minimal stand-ins
// for the reference types the shim touches (same names, same member signatures as GPU_Render.h:20-227,
// GPU_Types.h, LAReference.h:217-262, BLAS.h:13-24, HDRFloat.h getters; tests/shim/standin_types.hpp), then the explicit
// instantiations a maintainer's GPU_Render_hip.cpp would contain.  Compile + link only; the members are EXECUTED by
// tests/shim/shim_exec.cpp (-m gpu) and type-checked against the real headers by tests/test_shim_real_headers.py.
#include "shim/standin_types.hpp"

#include "../fractalshark_amd/csrc/gpu_render_shim.hpp"

// What GPU_Render_hip.cpp instantiates (subset of GPU_Render.cu:227-230,409-429,503-537,1204-1300,1610-1692).
using HDR32 = HDRFloat<float>;
template uint32_t GPURenderer::InitializeMemory<uint32_t>(uint32_t, uint32_t, uint32_t, const Color16 *, uint32_t,
                                                          uint32_t, uint64_t, bool);
template void GPURenderer::ClearMemory<uint32_t>();
template uint32_t GPURenderer::InitializePerturb<uint32_t, HDR32, float, PerturbExtras::Disable, HDR32>(
    size_t, const GPUPerturbResults<uint32_t, HDR32, PerturbExtras::Disable> *, size_t,
    const GPUPerturbResults<uint32_t, HDR32, PerturbExtras::Disable> *,
    const LAReference<uint32_t, HDR32, float, PerturbExtras::Disable> *);
template uint32_t GPURenderer::RenderPerturbLAv2<uint32_t, HDR32, float, LAv2Mode::Full, PerturbExtras::Disable>(
    RenderAlgorithm, HDR32, HDR32, HDR32, HDR32, HDR32, HDR32, uint32_t);
template uint32_t GPURenderer::RenderPerturbLAv2<uint32_t, HDR32, float, LAv2Mode::PO, PerturbExtras::Disable>(
    RenderAlgorithm, HDR32, HDR32, HDR32, HDR32, HDR32, HDR32, uint32_t);
template uint32_t GPURenderer::RenderPerturbLAv2<uint32_t, HDR32, float, LAv2Mode::LAO, PerturbExtras::Disable>(
    RenderAlgorithm, HDR32, HDR32, HDR32, HDR32, HDR32, HDR32, uint32_t);
template uint32_t GPURenderer::RenderPerturbBLA<uint32_t, HDR32>(RenderAlgorithm,
                                                                 const GPUPerturbResults<uint32_t, HDR32, PerturbExtras::Disable> *,
                                                                 BLAS<uint32_t, HDR32> *, HDR32, HDR32, HDR32, HDR32, HDR32,
                                                                 HDR32, uint32_t, int);
template uint32_t GPURenderer::Render<uint32_t, double>(RenderAlgorithm, double, double, double, double, uint32_t, int);
using HDR64 = HDRFloat<double>;
template uint32_t GPURenderer::Render<uint32_t, HDR32>(RenderAlgorithm, HDR32, HDR32, HDR32, HDR32, uint32_t, int);
template uint32_t GPURenderer::RenderPerturbLAv2<uint32_t, HDR64, double, LAv2Mode::Full, PerturbExtras::Disable>(
    RenderAlgorithm, HDR64, HDR64, HDR64, HDR64, HDR64, HDR64, uint32_t);
template uint32_t GPURenderer::RenderPerturbBLA<uint32_t, HDR64>(RenderAlgorithm,
                                                                 const GPUPerturbResults<uint32_t, HDR64, PerturbExtras::Disable> *,
                                                                 BLAS<uint32_t, HDR64> *, HDR64, HDR64, HDR64, HDR64, HDR64,
                                                                 HDR64, uint32_t, int);
template uint32_t GPURenderer::RenderCurrent<uint32_t>(uint32_t, uint32_t *, Color16 *, ReductionResults *, bool);
// SimpleCompression orbits (Gpu*RC* algorithms), the 2x32 type and IterType = uint64_t
template uint32_t GPURenderer::InitializePerturb<uint32_t, HDR32, float, PerturbExtras::SimpleCompression, HDR32>(
    size_t, const GPUPerturbResults<uint32_t, HDR32, PerturbExtras::SimpleCompression> *, size_t,
    const GPUPerturbResults<uint32_t, HDR32, PerturbExtras::SimpleCompression> *,
    const LAReference<uint32_t, HDR32, float, PerturbExtras::SimpleCompression> *);
template uint32_t GPURenderer::RenderPerturbLAv2<uint32_t, HDR32, float, LAv2Mode::Full, PerturbExtras::SimpleCompression>(
    RenderAlgorithm, HDR32, HDR32, HDR32, HDR32, HDR32, HDR32, uint32_t);
using HDR2x32 = HDRFloat<CudaDblflt<MattDblflt>>;
template uint32_t GPURenderer::InitializePerturb<uint32_t, HDR2x32, CudaDblflt<MattDblflt>, PerturbExtras::Disable, HDR2x32>(
    size_t, const GPUPerturbResults<uint32_t, HDR2x32, PerturbExtras::Disable> *, size_t,
    const GPUPerturbResults<uint32_t, HDR2x32, PerturbExtras::Disable> *,
    const LAReference<uint32_t, HDR2x32, CudaDblflt<MattDblflt>, PerturbExtras::Disable> *);
template uint32_t GPURenderer::RenderPerturbLAv2<uint32_t, HDR2x32, CudaDblflt<MattDblflt>, LAv2Mode::Full, PerturbExtras::Disable>(
    RenderAlgorithm, HDR2x32, HDR2x32, HDR2x32, HDR2x32, HDR2x32, HDR2x32, uint32_t);
// the non-HDR LAv2 types: Gpu1x32 / Gpu1x64 / Gpu2x32 PerturbedLAv2* (GPU_Render.cu:511-537,1204-1300)
using P2x32 = CudaDblflt<MattDblflt>;
#define FS_INST_PLAIN(T, IT)                                                                                            \
    template uint32_t GPURenderer::InitializePerturb<IT, T, T, PerturbExtras::Disable, T>(                              \
        size_t, const GPUPerturbResults<IT, T, PerturbExtras::Disable> *, size_t,                                       \
        const GPUPerturbResults<IT, T, PerturbExtras::Disable> *, const LAReference<IT, T, T, PerturbExtras::Disable> *); \
    template uint32_t GPURenderer::RenderPerturbLAv2<IT, T, T, LAv2Mode::Full, PerturbExtras::Disable>(                 \
        RenderAlgorithm, T, T, T, T, T, T, IT);                                                                         \
    template uint32_t GPURenderer::RenderPerturbLAv2<IT, T, T, LAv2Mode::PO, PerturbExtras::Disable>(                   \
        RenderAlgorithm, T, T, T, T, T, T, IT);                                                                         \
    template uint32_t GPURenderer::RenderPerturbLAv2<IT, T, T, LAv2Mode::LAO, PerturbExtras::Disable>(                  \
        RenderAlgorithm, T, T, T, T, T, T, IT)
FS_INST_PLAIN(float, uint32_t);
FS_INST_PLAIN(double, uint32_t);
FS_INST_PLAIN(P2x32, uint32_t);
FS_INST_PLAIN(float, uint64_t);
#undef FS_INST_PLAIN
// ... and their SimpleCompression forms (Gpu1x32 / Gpu1x64 / Gpu2x32 / GpuHDRx2x32 PerturbedRCLAv2*)
#define FS_INST_RC(T, S)                                                                                                \
    template uint32_t GPURenderer::InitializePerturb<uint32_t, T, S, PerturbExtras::SimpleCompression, T>(              \
        size_t, const GPUPerturbResults<uint32_t, T, PerturbExtras::SimpleCompression> *, size_t,                       \
        const GPUPerturbResults<uint32_t, T, PerturbExtras::SimpleCompression> *,                                       \
        const LAReference<uint32_t, T, S, PerturbExtras::SimpleCompression> *);                                         \
    template uint32_t GPURenderer::RenderPerturbLAv2<uint32_t, T, S, LAv2Mode::Full, PerturbExtras::SimpleCompression>( \
        RenderAlgorithm, T, T, T, T, T, T, uint32_t)
FS_INST_RC(float, float);
FS_INST_RC(double, double);
FS_INST_RC(P2x32, P2x32);
FS_INST_RC(HDR2x32, P2x32);
#undef FS_INST_RC
template uint32_t GPURenderer::RenderPerturbBLAScaled<uint32_t, HDR32>(
    RenderAlgorithm, const GPUPerturbResults<uint32_t, HDR32, PerturbExtras::Bad> *,
    const GPUPerturbResults<uint32_t, float, PerturbExtras::Bad> *, HDR32, HDR32, HDR32, HDR32, HDR32, HDR32, uint32_t, int);
template uint32_t GPURenderer::Render<uint32_t, float>(RenderAlgorithm, float, float, float, float, uint32_t, int);
template uint32_t GPURenderer::Render<uint32_t, MattDblflt>(RenderAlgorithm, MattDblflt, MattDblflt, MattDblflt, MattDblflt,
                                                            uint32_t, int);
template uint32_t GPURenderer::Render<uint32_t, MattQFltflt>(RenderAlgorithm, MattQFltflt, MattQFltflt, MattQFltflt,
                                                             MattQFltflt, uint32_t, int);
template uint32_t GPURenderer::Render<uint32_t, MattQDbldbl>(RenderAlgorithm, MattQDbldbl, MattQDbldbl, MattQDbldbl,
                                                             MattQDbldbl, uint32_t, int);
template uint32_t GPURenderer::Render<uint32_t, MattDbldbl>(RenderAlgorithm, MattDbldbl, MattDbldbl, MattDbldbl, MattDbldbl,
                                                            uint32_t, int);
template uint32_t GPURenderer::RenderPerturbBLAScaled<uint32_t, double>(
    RenderAlgorithm, const GPUPerturbResults<uint32_t, double, PerturbExtras::Bad> *,
    const GPUPerturbResults<uint32_t, float, PerturbExtras::Bad> *, double, double, double, double, double, double, uint32_t, int);
template uint32_t GPURenderer::InitializeMemory<uint64_t>(uint32_t, uint32_t, uint32_t, const Color16 *, uint32_t,
                                                          uint32_t, uint64_t, bool);
template void GPURenderer::ClearMemory<uint64_t>();
template uint32_t GPURenderer::InitializePerturb<uint64_t, HDR32, float, PerturbExtras::Disable, HDR32>(
    size_t, const GPUPerturbResults<uint64_t, HDR32, PerturbExtras::Disable> *, size_t,
    const GPUPerturbResults<uint64_t, HDR32, PerturbExtras::Disable> *,
    const LAReference<uint64_t, HDR32, float, PerturbExtras::Disable> *);
template uint32_t GPURenderer::RenderPerturbLAv2<uint64_t, HDR32, float, LAv2Mode::Full, PerturbExtras::Disable>(
    RenderAlgorithm, HDR32, HDR32, HDR32, HDR32, HDR32, HDR32, uint64_t);
template uint32_t GPURenderer::RenderCurrent<uint64_t>(uint64_t, uint64_t *, Color16 *, ReductionResults *, bool);

int main()
{
    // Never executed by the CPU test (it only compiles and links); on a GPU box it is a tiny end-to-end call.
    GPURenderer r;
    return (int)r.SyncComputeStream();
}
