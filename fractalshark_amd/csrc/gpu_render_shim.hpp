// gpu_render_shim.hpp -- the reference-side binding: `GPURenderer` member definitions that forward to the
// C ABI of libfsmi355.so (include/fsmi355.h).
//
// How a FractalShark maintainer uses it:
//   * keep FractalSharkLib/GPU_Render.h exactly as it is (class declaration, NB_THREADS_W/H constants);
//   * replace FractalSharkGpuLib/GPU_Render.cu in the link by ONE translation unit that does
//         #include "GPU_Render.h"
//         #include "LAReference.h"
//         #include "PerturbationResults.h"
//         #include "gpu_render_shim.hpp"
//     and link libfsmi355.so instead of FractalSharkGpuLib + cudart.
//
// The reference's class keeps its state in CUDA-typed private members (cudaStream_t == void* outside nvcc,
// GPU_Types.h:26-28).  The shim stores the opaque `fs_renderer*` in `m_ComputeStream` (a void* it owns) and leaves
// the other members untouched, so the class layout seen by Fractal.cpp does not change.
//
// Every template below is a *definition of a member the reference declares* (GPU_Render.h:25-158) with the
// reference's explicit-instantiation list (GPU_Render.cu:227-230,409-429,503-537,583-594,849-991,1192-1300,
// 1380-1436,1610-1692,1807-1818) reduced to the numeric types this library implements; calling an
// instantiation that is not built returns FS_ERR_UNSUPPORTED (the caller then disables the GPU exactly as it
// does for a CUDA error, Fractal.cpp:153-155).
//
// This header is also compiled stand-alone by tests/test_shim_compile.py against minimal stand-ins of the
// reference types (FS_SHIM_SELFTEST), to keep the signatures honest without the reference tree.
#pragma once

#include <stdint.h>

#include <type_traits>

#include "../../include/fsmi355.h"

#ifndef FS_SHIM_SELFTEST
// Provided by the including translation unit: GPU_Render.h, LAReference.h, PerturbationResults.h, BLAS.h.
#endif

namespace fsmi355_shim {

inline fs_renderer *&handle(cudaStream_t &slot) { return reinterpret_cast<fs_renderer *&>(slot); }

// Map the reference's numeric type T to a C-ABI tag (types this library implements; -1 = FS_ERR_UNSUPPORTED).
template <class T> struct type_tag {
    static constexpr int value = -1;
};
template <> struct type_tag<double> {
    static constexpr int value = FS_T_F64;
};
template <> struct type_tag<float> {
    static constexpr int value = FS_T_F32;
};
template <> struct type_tag<::CudaDblflt<::MattDblflt>> {
    static constexpr int value = FS_T_2X32;
};
template <> struct type_tag<::HDRFloat<float>> {
    static constexpr int value = FS_T_HDR32;
};
template <> struct type_tag<::HDRFloat<double>> {
    static constexpr int value = FS_T_HDR64;
};
template <> struct type_tag<::HDRFloat<::CudaDblflt<::MattDblflt>>> {
    static constexpr int value = FS_T_HDR2X32;
};

// HDRFloat<float> has the layout {float mantissa; int32 exp} = fs_real_hdr32 (HDRFloat.h:61-69);
// HDRFloat<double> = {double mantissa; int32 exp; pad} = fs_real_hdr64.
inline fs_real_hdr32 to_abi(const ::HDRFloat<float> &v)
{
    fs_real_hdr32 r;
    r.m = v.getMantissa();
    r.e = v.getExp();
    return r;
}
inline fs_real_hdr64 to_abi(const ::HDRFloat<double> &v)
{
    fs_real_hdr64 r;
    r.m = v.getMantissa();
    r.e = v.getExp();
    r.pad_ = 0;
    return r;
}
// HDRFloat<CudaDblflt<MattDblflt>> = {head, tail, exp} = fs_real_2x32 (CudaDblflt.h:24-28, dblflt.h:5-62).
inline fs_real_2x32 to_abi(const ::HDRFloat<::CudaDblflt<::MattDblflt>> &v)
{
    fs_real_2x32 r;
    r.head = v.getMantissa().head();
    r.tail = v.getMantissa().tail();
    r.e = v.getExp();
    return r;
}
// the plain (non-HDR) types cross the ABI as themselves; CudaDblflt<MattDblflt> = {head, tail} = fs_real_p2x32
inline float to_abi(float v) { return v; }
inline double to_abi(double v) { return v; }
inline fs_real_p2x32 to_abi(const ::CudaDblflt<::MattDblflt> &v)
{
    fs_real_p2x32 r;
    r.head = v.head();
    r.tail = v.tail();
    return r;
}
template <class T> struct abi_real;
template <> struct abi_real<float> {
    using type = float;
};
template <> struct abi_real<double> {
    using type = double;
};
template <> struct abi_real<::CudaDblflt<::MattDblflt>> {
    using type = fs_real_p2x32;
};
template <> struct abi_real<::HDRFloat<::CudaDblflt<::MattDblflt>>> {
    using type = fs_real_2x32;
};
template <> struct abi_real<::HDRFloat<float>> {
    using type = fs_real_hdr32;
};
template <> struct abi_real<::HDRFloat<double>> {
    using type = fs_real_hdr64;
};

} // namespace fsmi355_shim

inline GPURenderer::GPURenderer()
{
    // ClearLocals() equivalent: the handle is created lazily on the first InitializeMemory so that
    // constructing the std::array<GPURenderer,4> in Fractal never touches the device (Fractal.h:496-508).
    m_ComputeStream = nullptr;
    m_DisplayStream = nullptr;
    OutputIterMatrix = nullptr;
    m_Width = m_Height = 0;
}

inline GPURenderer::~GPURenderer()
{
    if (m_ComputeStream)
        fs_destroy(fsmi355_shim::handle(m_ComputeStream));
    m_ComputeStream = nullptr;
}

inline uint32_t GPURenderer::TestCudaIsWorking() { return fs_test_device_is_working(); }
inline const char *GPURenderer::ConvertErrorToString(uint32_t err) { return fs_error_string(err); }

template <typename IterType>
uint32_t GPURenderer::InitializeMemory(uint32_t w, uint32_t h, uint32_t antialiasing, const Color16 *palInterleaved,
                                       uint32_t palIters, uint32_t paletteAuxDepth, uint64_t paletteGeneration,
                                       bool expectedReuse)
{
    if (!m_ComputeStream) {
        fsmi355_shim::handle(m_ComputeStream) = fs_create(0); // the reference hard-codes device 0
        if (!m_ComputeStream)
            return FS_ERR_1;
    }
    static_assert(sizeof(Color16) == sizeof(fs_color16), "Color16 layout");
    const uint32_t err =
        fs_init_memory(fsmi355_shim::handle(m_ComputeStream), w, h, antialiasing, (uint32_t)sizeof(IterType),
                       reinterpret_cast<const fs_color16 *>(palInterleaved), palIters, paletteAuxDepth,
                       paletteGeneration, expectedReuse ? 1 : 0);
    if (err == 0) {
        m_Width = w;
        m_Height = h;
        m_Antialiasing = antialiasing;
        m_IterTypeSize = sizeof(IterType);
    }
    return err;
}

template <typename IterType> void GPURenderer::ClearMemory()
{
    if (m_ComputeStream)
        fs_clear(fsmi355_shim::handle(m_ComputeStream));
}

template <typename IterType, class T1, class SubType, PerturbExtras PExtras, class T2>
uint32_t GPURenderer::InitializePerturb(size_t GenerationNumber1, const GPUPerturbResults<IterType, T1, PExtras> *Perturb1,
                                        size_t GenerationNumber2, const GPUPerturbResults<IterType, T2, PExtras> *Perturb2,
                                        const LAReference<IterType, T1, SubType, PExtras> *LaReferenceHost)
{
    (void)GenerationNumber2;
    (void)Perturb2; // second orbit: scaled kernels only (not built)
    if (!m_ComputeStream)
        return 0;
    constexpr int tag = fsmi355_shim::type_tag<T1>::value;
    if constexpr (tag < 0 || (PExtras != PerturbExtras::Disable && PExtras != PerturbExtras::SimpleCompression)) {
        return FS_ERR_UNSUPPORTED;
    } else {
    fs_renderer *r = fsmi355_shim::handle(m_ComputeStream);
    uint32_t err;
    if constexpr (PExtras == PerturbExtras::SimpleCompression) {
        // the waypoints + the constant c of the runtime decompressor (Perturb.cuh:300-326)
        const auto xlow = fsmi355_shim::to_abi(Perturb1->GetOrbitXLow());
        const auto ylow = fsmi355_shim::to_abi(Perturb1->GetOrbitYLow());
        err = fs_upload_orbit_compressed(r, GenerationNumber1, tag, (uint32_t)sizeof(IterType), Perturb1->GetFullOrbit(),
                                         Perturb1->GetCompressedSize(), Perturb1->GetUncompressedSize(),
                                         Perturb1->GetPeriodMaybeZero(), &xlow, &ylow);
    } else {
        err = fs_upload_orbit(r, GenerationNumber1, tag, (uint32_t)sizeof(IterType), Perturb1->GetFullOrbit(),
                              Perturb1->GetCompressedSize(), Perturb1->GetUncompressedSize(),
                              Perturb1->GetPeriodMaybeZero());
    }
    if (err || !LaReferenceHost)
        return err;
    // LAReference keeps LAInfoDeep / LAStageInfo in GrowableVectors whose element layouts are the fs_la_* /
    // fs_la_stage_* records of fs_layout.h (the reference static-asserts host == device layout at
    // GPU_LAReference.h:118-133); sizeof(IterType) selects the _u32 or _u64 family.
    const auto &at = LaReferenceHost->GetAT();
    static_assert(sizeof(IterType) != 4 || !std::is_same<T1, ::HDRFloat<float>>::value || sizeof(at) == sizeof(fs_at_hdr32_u32),
                  "ATInfo layout");
    static_assert(sizeof(IterType) != 8 || !std::is_same<T1, ::HDRFloat<float>>::value || sizeof(at) == sizeof(fs_at_hdr32_u64),
                  "ATInfo layout (uint64_t IterType)");
    return fs_upload_la(r, GenerationNumber1, tag, (uint32_t)sizeof(IterType), LaReferenceHost->GetLAs().GetData(),
                        (uint32_t)LaReferenceHost->GetLAs().GetSize(), LaReferenceHost->GetLAStages().GetData(),
                        (uint32_t)LaReferenceHost->GetLAStageCount(), LaReferenceHost->IsValid() ? 1 : 0,
                        LaReferenceHost->UseAT() ? 1 : 0, &at);
    }
}

template <typename IterType, class T, class SubType, LAv2Mode Mode, PerturbExtras PExtras>
uint32_t GPURenderer::RenderPerturbLAv2(RenderAlgorithm /*algorithm*/, T /*cx*/, T /*cy*/, T dx, T dy, T centerX,
                                        T centerY, IterType n_iterations)
{
    if (!m_ComputeStream)
        return 0; // "memory not initialised" is silent, GPU_Render.cu:1007-1009
    constexpr int tag = fsmi355_shim::type_tag<T>::value;
    if constexpr (tag < 0) {
        return FS_ERR_UNSUPPORTED;
    } else {
    // PExtras only changes how the orbit was uploaded (InitializePerturb): SimpleCompression waypoints are expanded on
    // the device at upload, for every numeric type
    if (PExtras != PerturbExtras::Disable && PExtras != PerturbExtras::SimpleCompression)
        return FS_ERR_UNSUPPORTED;
    const typename fsmi355_shim::abi_real<T>::type co[4] = {fsmi355_shim::to_abi(dx), fsmi355_shim::to_abi(dy),
                                                            fsmi355_shim::to_abi(centerX), fsmi355_shim::to_abi(centerY)};
    const int mode = Mode == LAv2Mode::Full ? FS_LAV2_FULL : (Mode == LAv2Mode::PO ? FS_LAV2_PO : FS_LAV2_LAO);
    return fs_render_lav2(fsmi355_shim::handle(m_ComputeStream), tag, mode, FS_PARITY_CPU, co, (uint64_t)n_iterations);
    }
}

template <typename IterType, class T>
uint32_t GPURenderer::RenderPerturbBLA(RenderAlgorithm /*algorithm*/,
                                       const GPUPerturbResults<IterType, T, PerturbExtras::Disable> *results,
                                       BLAS<IterType, T> *blas, T /*cx*/, T /*cy*/, T dx, T dy, T centerX, T centerY,
                                       IterType n_iterations, int /*iteration_precision*/)
{
    if (!m_ComputeStream)
        return 0;
    constexpr int tag = fsmi355_shim::type_tag<T>::value;
    // GPU_Render.cu:1610-1692: T = double (Gpu1x64PerturbedBLA), HDRFloat<float>, HDRFloat<double>
    if constexpr (tag != FS_T_HDR32 && tag != FS_T_HDR64 && tag != FS_T_F64) {
        return FS_ERR_UNSUPPORTED;
    } else {
    fs_renderer *r = fsmi355_shim::handle(m_ComputeStream);
    // The reference re-uploads orbit and table on every call (GPU_Render.cu:1464-1479); so do we.
    uint32_t err = fs_upload_orbit(r, 0, tag, (uint32_t)sizeof(IterType), results->GetFullOrbit(),
                                   results->GetCompressedSize(), results->GetUncompressedSize(),
                                   results->GetPeriodMaybeZero());
    if (err)
        return err;
    const size_t n_levels = blas->m_B.size();
    const void *levels[64] = {};
    uint64_t sizes[64] = {};
    for (size_t l = 0; l < n_levels && l < 64; l++) {
        levels[l] = blas->m_B[l].empty() ? nullptr : blas->m_B[l].data();
        sizes[l] = blas->m_B[l].size();
    }
    static_assert(sizeof(BLA<::HDRFloat<float>>) == sizeof(fs_bla_hdr32) || !std::is_same<T, ::HDRFloat<float>>::value,
                  "BLA layout");
    static_assert(sizeof(BLA<double>) == sizeof(fs_bla_f64) || !std::is_same<T, double>::value, "BLA layout (double)");
    err = fs_upload_bla(r, tag, levels, sizes, (int32_t)n_levels, blas->m_LM2);
    if (err)
        return err;
    const typename fsmi355_shim::abi_real<T>::type co[4] = {fsmi355_shim::to_abi(dx), fsmi355_shim::to_abi(dy),
                                                            fsmi355_shim::to_abi(centerX), fsmi355_shim::to_abi(centerY)};
    return fs_render_bla(r, tag, co, (uint64_t)n_iterations);
    }
}

template <typename IterType, class T>
uint32_t GPURenderer::RenderPerturbBLAScaled(RenderAlgorithm /*algorithm*/,
                                             const GPUPerturbResults<IterType, T, PerturbExtras::Bad> *double_perturb,
                                             const GPUPerturbResults<IterType, float, PerturbExtras::Bad> *float_perturb,
                                             T /*cx*/, T /*cy*/, T dx, T dy, T centerX, T centerY, IterType n_iterations,
                                             int /*iteration_precision*/)
{
    if (!m_ComputeStream)
        return 0; // GPU_Render.cu:1317-1319
    constexpr int tag = fsmi355_shim::type_tag<T>::value;
    if constexpr (tag != FS_T_HDR32 && tag != FS_T_F64) {
        return FS_ERR_UNSUPPORTED;
    } else {
    fs_renderer *r = fsmi355_shim::handle(m_ComputeStream);
    // both orbits are uploaded inside the call, like the reference (GPU_Render.cu:1324-1345)
    uint32_t err = fs_upload_orbit_scaled(r, tag, (uint32_t)sizeof(IterType), double_perturb->GetFullOrbit(),
                                          float_perturb->GetFullOrbit(), double_perturb->GetUncompressedSize(),
                                          float_perturb->GetPeriodMaybeZero());
    if (err)
        return err;
    if constexpr (tag == FS_T_F64) { // Gpu1x32PerturbedScaled
        const double co[4] = {(double)dx, (double)dy, (double)centerX, (double)centerY};
        return fs_render_scaled(r, tag, co, (uint64_t)n_iterations);
    } else { // GpuHDRx32PerturbedScaled
        const fs_real_hdr32 co[4] = {fsmi355_shim::to_abi(dx), fsmi355_shim::to_abi(dy), fsmi355_shim::to_abi(centerX),
                                     fsmi355_shim::to_abi(centerY)};
        return fs_render_scaled(r, tag, co, (uint64_t)n_iterations);
    }
    }
}

template <typename IterType, class T>
uint32_t GPURenderer::Render(RenderAlgorithm /*algorithm*/, T cx, T cy, T dx, T dy, IterType n_iterations,
                             int iteration_precision)
{
    if (!m_ComputeStream)
        return 0;
    fs_renderer *r = fsmi355_shim::handle(m_ComputeStream);
    // Fractal::FillGpuCoords hands over the view's MIN corner (Fractal.cpp:1833-1844).
    if constexpr (std::is_same<T, float>::value) { // Gpu1x32
        const float co[4] = {cx, cy, dx, dy};
        return fs_render_direct_lp(r, FS_T_F32, co, (uint64_t)n_iterations, iteration_precision);
    } else if constexpr (std::is_same<T, ::MattDblflt>::value) { // Gpu2x32
        const float co[8] = {cx.head, cx.tail, cy.head, cy.tail, dx.head, dx.tail, dy.head, dy.tail};
        return fs_render_direct_lp(r, FS_T_2X32, co, (uint64_t)n_iterations, iteration_precision);
    } else if constexpr (std::is_same<T, ::MattDbldbl>::value) { // Gpu2x64
        const double co[8] = {cx.head, cx.tail, cy.head, cy.tail, dx.head, dx.tail, dy.head, dy.tail};
        return fs_render_direct_lp(r, FS_T_2X64, co, (uint64_t)n_iterations, iteration_precision);
    } else if constexpr (std::is_same<T, ::MattQFltflt>::value) { // Gpu4x32
        const float co[16] = {cx.x, cx.y, cx.z, cx.w, cy.x, cy.y, cy.z, cy.w, dx.x, dx.y, dx.z, dx.w, dy.x, dy.y, dy.z, dy.w};
        return fs_render_direct_lp(r, FS_T_4X32, co, (uint64_t)n_iterations, iteration_precision);
    } else if constexpr (std::is_same<T, ::MattQDbldbl>::value) { // Gpu4x64
        const double co[16] = {cx.x, cx.y, cx.z, cx.w, cy.x, cy.y, cy.z, cy.w, dx.x, dx.y, dx.z, dx.w, dy.x, dy.y, dy.z, dy.w};
        return fs_render_direct_lp(r, FS_T_4X64, co, (uint64_t)n_iterations, iteration_precision);
    } else if constexpr (std::is_same<T, double>::value) {
        // Gpu1x64 is built as the twin of the CPU algorithm Cpu64 (rows from maxY downwards, Fractal.cpp:2148-2183).
        // maxY is not part of this interface: it is rebuilt from the min corner, which rounds differently from
        // T(ptz.GetMaxY()) in the last place; pass maxY through fs_render_direct directly for bit-equality with Cpu64.
        const double maxY = (double)cy + (double)dy * (double)m_Height;
        const double co[4] = {(double)dx, (double)dy, (double)cx, maxY};
        return fs_render_direct(r, FS_T_F64, co, (uint64_t)n_iterations);
    } else if constexpr (fsmi355_shim::type_tag<T>::value == FS_T_HDR32 || fsmi355_shim::type_tag<T>::value == FS_T_HDR64) {
        const T maxY = cy + dy * T((float)m_Height);
        const typename fsmi355_shim::abi_real<T>::type co[4] = {fsmi355_shim::to_abi(dx), fsmi355_shim::to_abi(dy),
                                                                fsmi355_shim::to_abi(cx), fsmi355_shim::to_abi(maxY)};
        return fs_render_direct(r, fsmi355_shim::type_tag<T>::value, co, (uint64_t)n_iterations);
    } else if constexpr (std::is_same<T, ::CudaDblflt<::MattDblflt>>::value) {
        // instantiated by the reference (GPU_Render.cu:905-912,977-984) but no branch of its Render launches anything
        // for this T: it returns cudaSuccess (:843)
        return 0;
    } else {
        return FS_ERR_UNSUPPORTED;
    }
}

template <typename IterType>
uint32_t GPURenderer::RenderCurrent(IterType n_iterations, IterType *iter_buffer, Color16 *color_buffer,
                                    ReductionResults *reduction_results, bool progressive)
{
    if (!m_ComputeStream)
        return 0;
    static_assert(sizeof(ReductionResults) == sizeof(fs_reduction), "ReductionResults layout");
    return fs_render_current(fsmi355_shim::handle(m_ComputeStream), (uint64_t)n_iterations, iter_buffer,
                             reinterpret_cast<fs_color16 *>(color_buffer),
                             reinterpret_cast<fs_reduction *>(reduction_results), progressive ? 1 : 0);
}

inline uint32_t GPURenderer::SyncComputeStream()
{
    return m_ComputeStream ? fs_sync_compute(fsmi355_shim::handle(m_ComputeStream)) : 0;
}
inline uint32_t GPURenderer::SyncDisplayStream()
{
    return m_ComputeStream ? fs_sync_display(fsmi355_shim::handle(m_ComputeStream)) : 0;
}
inline uint32_t GPURenderer::QueryComputeStream()
{
    return m_ComputeStream ? fs_query_compute(fsmi355_shim::handle(m_ComputeStream)) : 0;
}
inline uint32_t GPURenderer::EnqueueComputeDoneCallback()
{
    if (!m_ComputeStream)
        return 0;
    // GPU_Render.cu:608-615: SignalComputeDone() runs on a runtime thread once prior compute work is done.
    return fs_enqueue_done_callback(
        fsmi355_shim::handle(m_ComputeStream), [](void *self) { static_cast<GPURenderer *>(self)->SignalComputeDone(); },
        this);
}
