// Compile/link check of fractalshark_amd/csrc/gpu_render_shim.hpp WITHOUT the reference tree: minimal stand-ins
// for the reference types the shim touches (same names, same member signatures as GPU_Render.h:20-227,
// GPU_Types.h, LAReference.h:217-262, BLAS.h:13-24, HDRFloat.h getters), then the explicit instantiations a
// maintainer's GPU_Render_hip.cpp would contain.  No arithmetic, nothing is executed on the CPU-only box.
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
    E *GetData() const { return nullptr; }
    size_t GetSize() const { return 0; }
};
template <typename IterType, class F, class S> struct ATInfo {
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
    bool IsValid() const { return true; }
    bool UseAT() const { return true; }
    const ATInfo<IterType, Float, SubType> &GetAT() const { return at; }
    IterType GetLAStageCount() const { return 0; }
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

private:
    void *OutputIterMatrix;
    uint32_t m_Width, m_Height, m_Antialiasing, m_IterTypeSize;
    cudaStream_t m_ComputeStream;
    cudaStream_t m_DisplayStream;
    std::atomic<bool> m_ComputeDoneFlag{false};
};

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
