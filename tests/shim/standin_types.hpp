// tests/shim/standin_types.hpp -- minimal stand-ins of the reference types gpu_render_shim.hpp touches (same names,
// same member signatures as GPU_Render.h:20-227, GPU_Types.h, LAReference.h:217-262, BLAS.h:13-24, HDRFloat.h getters),
// for the two shim tests that must work WITHOUT the reference tree (the GPU box has none): tests/shim_selftest.cpp
// (compile + link of the instantiation list) and tests/shim/shim_exec.cpp (the members executed on the GPU).  The real
// headers are used by tests/test_shim_real_headers.py where /root/reference exists.  Unlike the first version the
// containers here carry data (pointer + size), so that the executed test hands real inputs through the shim.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <vector>

#define FS_SHIM_SELFTEST 1

using cudaStream_t = void *;
enum class PerturbExtras { Disable, Bad, SimpleCompression, MaxCompression };
enum class LAv2Mode { Full, PO, LAO };
struct RenderAlgorithm {
    int Algorithm;
};
struct Color16 {
    uint16_t r, g, b, a;
};
struct ReductionResults {
    uint64_t Min, Max, Sum;
};
struct AntialiasedColors {
    Color16 *aa_colors;
};
struct Palette {
    Color16 *local_pal;
};
struct PerturbResultsCollection {
};
template <class T> class HDRFloat {
public:
    T mantissa;
    int32_t exp;
    HDRFloat() = default;
    explicit HDRFloat(float v) : mantissa(T(v)), exp(0) {}
    HDRFloat(T m, int32_t e) : mantissa(m), exp(e) {}
    T getMantissa() const { return mantissa; }
    int32_t getExp() const { return exp; }
    // arithmetic is the reference's; the stand-in only needs the operators to exist
    friend HDRFloat operator+(HDRFloat a, const HDRFloat &) { return a; }
    friend HDRFloat operator*(HDRFloat a, const HDRFloat &) { return a; }
};
struct MattDbldbl {
    double head, tail;
};
struct MattDblflt {
    float head, tail;
};
struct MattQFltflt {
    float x, y, z, w;
};
struct MattQDbldbl {
    double x, y, z, w;
};
template <class T = MattDblflt> class CudaDblflt {
public:
    T d;
    float head() const { return d.head; }
    float tail() const { return d.tail; }
};
template <class T, PerturbExtras P> struct GPUReferenceIter {
    T x, y;
};
template <typename IterType, class T, PerturbExtras PExtras> class GPUPerturbResults {
public:
    const GPUReferenceIter<T, PExtras> *GetFullOrbit() const { return orb; }
    IterType GetCompressedSize() const { return n; }
    IterType GetUncompressedSize() const { return n; }
    IterType GetPeriodMaybeZero() const { return period; }
    T GetOrbitXLow() const { return xlow; }
    T GetOrbitYLow() const { return ylow; }
    const GPUReferenceIter<T, PExtras> *orb = nullptr;
    IterType n = 0, period = 0;
    T xlow{}, ylow{};
};
template <class E> struct GrowableVector {
    E *GetData() const { return data; }
    size_t GetSize() const { return size; }
    E *data = nullptr;
    size_t size = 0;
};
template <typename IterType, class F, class S> struct ATInfo { // opaque bytes, large enough for every numeric type
    unsigned char bytes[256];
};
template <typename IterType, class S> struct ATInfo<IterType, HDRFloat<float>, S> { // the size the shim static-asserts
    unsigned char bytes[sizeof(IterType) == 4 ? 116 : 120];
};
template <typename IterType, class F, class S, PerturbExtras P> struct LAInfoDeep {
    unsigned char bytes[68];
};
template <typename IterType> struct LAStageInfo {
    IterType LAIndex, MacroItCount;
};
template <typename IterType, class Float, class SubType, PerturbExtras PExtras> class LAReference {
public:
    bool IsValid() const { return valid; }
    bool UseAT() const { return use_at; }
    const ATInfo<IterType, Float, SubType> &GetAT() const { return at; }
    IterType GetLAStageCount() const { return (IterType)stages.size; }
    bool valid = true, use_at = true;
    const GrowableVector<LAInfoDeep<IterType, Float, SubType, PExtras>> &GetLAs() const { return las; }
    const GrowableVector<LAStageInfo<IterType>> &GetLAStages() const { return stages; }
    ATInfo<IterType, Float, SubType> at;
    GrowableVector<LAInfoDeep<IterType, Float, SubType, PExtras>> las;
    GrowableVector<LAStageInfo<IterType>> stages;
};
template <class T> struct BLA {
    T r2, Ax, Ay, Bx, By;
    int l;
};
template <typename IterType, class T, PerturbExtras PExtras = PerturbExtras::Disable> class BLAS {
public:
    std::vector<std::vector<BLA<T>>> m_B;
    int32_t m_LM2 = 0;
};
// Member list of the reference class (declarations only).
class GPURenderer {
public:
    GPURenderer();
    ~GPURenderer();
    static uint32_t TestCudaIsWorking();
    template <typename IterType, class T>
    uint32_t Render(RenderAlgorithm algorithm, T cx, T cy, T dx, T dy, IterType n_iterations, int iteration_precision);
    template <typename IterType, class T>
    uint32_t RenderPerturbBLA(RenderAlgorithm algorithm,
                              const GPUPerturbResults<IterType, T, PerturbExtras::Disable> *results,
                              BLAS<IterType, T> *blas, T cx, T cy, T dx, T dy, T centerX, T centerY,
                              IterType n_iterations, int iteration_precision);
    template <typename IterType, class T>
    uint32_t RenderPerturbBLAScaled(RenderAlgorithm algorithm,
                                    const GPUPerturbResults<IterType, T, PerturbExtras::Bad> *double_perturb,
                                    const GPUPerturbResults<IterType, float, PerturbExtras::Bad> *float_perturb, T cx, T cy,
                                    T dx, T dy, T centerX, T centerY, IterType n_iterations, int iteration_precision);
    template <typename IterType, class T, class SubType, LAv2Mode Mode, PerturbExtras PExtras>
    uint32_t RenderPerturbLAv2(RenderAlgorithm algorithm, T cx, T cy, T dx, T dy, T centerX, T centerY,
                               IterType n_iterations);
    template <typename IterType>
    uint32_t InitializeMemory(uint32_t w, uint32_t h, uint32_t antialiasing, const Color16 *palInterleaved,
                              uint32_t palIters, uint32_t paletteAuxDepth, uint64_t paletteGeneration,
                              bool expectedReuse);
    template <typename IterType, class T1, class SubType, PerturbExtras PExtras, class T2>
    uint32_t InitializePerturb(size_t GenerationNumber1, const GPUPerturbResults<IterType, T1, PExtras> *Perturb1,
                               size_t GenerationNumber2, const GPUPerturbResults<IterType, T2, PExtras> *Perturb2,
                               const LAReference<IterType, T1, SubType, PExtras> *LaReferenceHost);
    template <typename IterType> void ClearMemory();
    static const char *ConvertErrorToString(uint32_t err);
    static const int32_t NB_THREADS_W = 16;
    static const int32_t NB_THREADS_H = 8;
    template <typename IterType>
    uint32_t RenderCurrent(IterType n_iterations, IterType *iter_buffer, Color16 *color_buffer,
                           ReductionResults *reduction_results, bool progressive = false);
    uint32_t SyncComputeStream();
    uint32_t SyncDisplayStream();
    uint32_t QueryComputeStream();
    uint32_t EnqueueComputeDoneCallback();
    void SignalComputeDone() { m_ComputeDoneFlag.store(true, std::memory_order_release); }
    void ResetComputeDoneFlag() { m_ComputeDoneFlag.store(false, std::memory_order_release); }
    bool IsComputeDone() const { return m_ComputeDoneFlag.load(std::memory_order_acquire); }

private:
    void *OutputIterMatrix;
    uint32_t m_Width, m_Height, m_Antialiasing, m_IterTypeSize;
    cudaStream_t m_ComputeStream;
    cudaStream_t m_DisplayStream;
    std::atomic<bool> m_ComputeDoneFlag{false};
};

