// tests/shim/shim_exec.cpp -- the GPURenderer members of gpu_render_shim.hpp EXECUTED on the GPU (tests/test_gpu_shim.py,
// -m gpu), the way Fractal.cpp drives them (Fractal.cpp:2693-2930): InitializeMemory -> InitializePerturb ->
// RenderPerturbLAv2 / RenderPerturbBLA / Render -> RenderCurrent -> SyncComputeStream, plus the done callback and the
// error codes.  The reference tree does not exist on the GPU box, so the reference types are the stand-ins of
// standin_types.hpp (the real headers are compiled against on the CPU side, tests/test_shim_real_headers.py); the
// inputs are raw record files written by the Python test from the golden-pinned host builders, and every iteration
// buffer the members return is written back for the test to compare with the fixtures / the oracle.  A member that
// forwards a wrong argument (swapped coordinate, wrong count, wrong mode, wrong type tag) changes a buffer or a code.
//
//   shim_exec <dir>     reads <dir>/meta.txt + *.bin, writes <dir>/out_*.bin and <dir>/result.txt
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "standin_types.hpp"

#include "../../fractalshark_amd/csrc/gpu_render_shim.hpp"

using HDR32 = HDRFloat<float>;
using HDR64 = HDRFloat<double>;

static std::vector<unsigned char> slurp(const std::string &path)
{
    std::vector<unsigned char> v;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path.c_str());
        exit(2);
    }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize((size_t)n);
    if (n && fread(v.data(), 1, (size_t)n, f) != (size_t)n)
        exit(2);
    fclose(f);
    return v;
}

static void dump(const std::string &path, const void *p, size_t n)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f || fwrite(p, 1, n, f) != n)
        exit(2);
    fclose(f);
}

struct Meta {
    unsigned w, h, n_iter, orbit_n, period, n_las, n_stages, use_at, la_valid, n_levels, lm2, pal_n, w0, h0, n_iter0;
    unsigned long long level_sizes[64];
};

int main(int argc, char **argv)
{
    if (argc < 2)
        return 2;
    const std::string d = std::string(argv[1]) + "/";
    Meta m{};
    {
        FILE *f = fopen((d + "meta.txt").c_str(), "r");
        if (!f)
            return 2;
        if (fscanf(f, "%u %u %u %u %u %u %u %u %u %u %u %u %u %u %u", &m.w, &m.h, &m.n_iter, &m.orbit_n, &m.period, &m.n_las,
                   &m.n_stages, &m.use_at, &m.la_valid, &m.n_levels, &m.lm2, &m.pal_n, &m.w0, &m.h0, &m.n_iter0) != 15)
            return 2;
        for (unsigned l = 0; l < m.n_levels; l++)
            if (fscanf(f, "%llu", &m.level_sizes[l]) != 1)
                return 2;
        fclose(f);
    }
    FILE *res = fopen((d + "result.txt").c_str(), "w");
    if (!res)
        return 2;
    auto orbit = slurp(d + "orbit.bin");   // GPUReferenceIter<HDRFloat<float>, Disable>[orbit_n]
    auto las = slurp(d + "las.bin");       // LAInfoDeep<uint32_t, HDRFloat<float>, float, Disable>[n_las]
    auto stages = slurp(d + "stages.bin"); // LAStageInfo<uint32_t>[n_stages]
    auto at = slurp(d + "at.bin");         // ATInfo<uint32_t, HDRFloat<float>, float>
    auto coords = slurp(d + "coords.bin"); // {dx, dy, centerX, centerY} as {float mantissa, int32 exp}
    auto pal = slurp(d + "palette.bin");   // Color16[pal_n]
    auto direct = slurp(d + "direct.bin"); // doubles {dx, dy, minX, minY}
    struct Pair {
        float m;
        int32_t e;
    };
    const Pair *co = (const Pair *)coords.data();
    const HDR32 dx(co[0].m, co[0].e), dy(co[1].m, co[1].e), cX(co[2].m, co[2].e), cY(co[3].m, co[3].e);
    // cx / cy are unused by the perturbation kernels; hand over poison so that a swapped argument shows
    const HDR32 poison(12345.0f, 77);
    const RenderAlgorithm alg{0};

    fprintf(res, "working %u\n", GPURenderer::TestCudaIsWorking());
    GPURenderer r;
    // before InitializeMemory every call is silent (GPU_Render.cu:564-566,1007-1009)
    fprintf(res, "uninit_render %u\n",
            r.RenderPerturbLAv2<uint32_t, HDR32, float, LAv2Mode::Full, PerturbExtras::Disable>(alg, poison, poison, dx, dy, cX,
                                                                                               cY, m.n_iter));
    fprintf(res, "bad_aa %u\n", r.InitializeMemory<uint32_t>(m.w, m.h, 5, nullptr, 0, 0, 0, false));
    fprintf(res, "init %u\n", r.InitializeMemory<uint32_t>(m.w, m.h, 1, (const Color16 *)pal.data(), m.pal_n, 0, 1, false));

    GPUPerturbResults<uint32_t, HDR32, PerturbExtras::Disable> pr;
    pr.orb = (const GPUReferenceIter<HDR32, PerturbExtras::Disable> *)orbit.data();
    pr.n = m.orbit_n;
    pr.period = m.period;
    LAReference<uint32_t, HDR32, float, PerturbExtras::Disable> la;
    la.las.data = (LAInfoDeep<uint32_t, HDR32, float, PerturbExtras::Disable> *)las.data();
    la.las.size = m.n_las;
    la.stages.data = (LAStageInfo<uint32_t> *)stages.data();
    la.stages.size = m.n_stages;
    la.valid = m.la_valid != 0;
    la.use_at = m.use_at != 0;
    memcpy(&la.at, at.data(), at.size() < sizeof(la.at) ? at.size() : sizeof(la.at));

    // render before any upload: Error6 (GPU_Render.cu:1015-1022)
    fprintf(res, "no_orbit %u\n",
            r.RenderPerturbLAv2<uint32_t, HDR32, float, LAv2Mode::Full, PerturbExtras::Disable>(alg, poison, poison, dx, dy, cX,
                                                                                               cY, m.n_iter));
    fprintf(res, "init_perturb %u\n",
            r.InitializePerturb<uint32_t, HDR32, float, PerturbExtras::Disable, HDR32>(7, &pr, 0, nullptr, &la));

    const size_t rw = (m.w + 15) / 16 * 16, rh = (m.h + 7) / 8 * 8;
    std::vector<uint32_t> iters(rw * rh);
    std::vector<Color16> colors(((m.w + 15) / 16 * 16) * ((m.h + 7) / 8 * 8));
    ReductionResults red{};

    r.ClearMemory<uint32_t>();
    fprintf(res, "lav2_full %u\n",
            r.RenderPerturbLAv2<uint32_t, HDR32, float, LAv2Mode::Full, PerturbExtras::Disable>(alg, poison, poison, dx, dy, cX,
                                                                                               cY, m.n_iter));
    fprintf(res, "done_cb %u\n", r.EnqueueComputeDoneCallback());
    fprintf(res, "current %u\n", r.RenderCurrent<uint32_t>(m.n_iter, iters.data(), colors.data(), &red, false));
    fprintf(res, "sync %u\n", r.SyncComputeStream());
    fprintf(res, "query %u\n", r.QueryComputeStream());
    dump(d + "out_lav2_full.bin", iters.data(), iters.size() * 4);
    dump(d + "out_colors.bin", colors.data(), colors.size() * sizeof(Color16));
    fprintf(res, "reduction %llu %llu %llu\n", (unsigned long long)red.Min, (unsigned long long)red.Max,
            (unsigned long long)red.Sum);

    r.ClearMemory<uint32_t>();
    fprintf(res, "lav2_lao %u\n",
            r.RenderPerturbLAv2<uint32_t, HDR32, float, LAv2Mode::LAO, PerturbExtras::Disable>(alg, poison, poison, dx, dy, cX,
                                                                                              cY, m.n_iter));
    fprintf(res, "current %u\n", r.RenderCurrent<uint32_t>(m.n_iter, iters.data(), nullptr, nullptr, false));
    fprintf(res, "sync %u\n", r.SyncComputeStream());
    dump(d + "out_lav2_lao.bin", iters.data(), iters.size() * 4);

    r.ClearMemory<uint32_t>();
    fprintf(res, "lav2_po %u\n",
            r.RenderPerturbLAv2<uint32_t, HDR32, float, LAv2Mode::PO, PerturbExtras::Disable>(alg, poison, poison, dx, dy, cX, cY,
                                                                                             m.n_iter));
    fprintf(res, "current %u\n", r.RenderCurrent<uint32_t>(m.n_iter, iters.data(), nullptr, nullptr, false));
    fprintf(res, "sync %u\n", r.SyncComputeStream());
    dump(d + "out_lav2_po.bin", iters.data(), iters.size() * 4);

    // RenderPerturbBLA: orbit + table are handed over inside the call (GPU_Render.cu:1464-1479)
    BLAS<uint32_t, HDR32> blas;
    blas.m_LM2 = (int32_t)m.lm2;
    blas.m_B.resize(m.n_levels);
    static_assert(sizeof(BLA<HDR32>) == sizeof(fs_bla_hdr32), "stand-in BLA record");
    for (unsigned l = 0; l < m.n_levels; l++) {
        if (!m.level_sizes[l])
            continue;
        auto raw = slurp(d + "bla_" + std::to_string(l) + ".bin");
        blas.m_B[l].resize(m.level_sizes[l]);
        memcpy((void *)blas.m_B[l].data(), raw.data(), raw.size());
    }
    r.ClearMemory<uint32_t>();
    fprintf(res, "bla %u\n", r.RenderPerturbBLA<uint32_t, HDR32>(alg, &pr, &blas, poison, poison, dx, dy, cX, cY, m.n_iter, 1));
    fprintf(res, "current %u\n", r.RenderCurrent<uint32_t>(m.n_iter, iters.data(), nullptr, nullptr, false));
    fprintf(res, "sync %u\n", r.SyncComputeStream());
    dump(d + "out_bla.bin", iters.data(), iters.size() * 4);

    // Render<uint32_t, double> (Gpu1x64) on its own geometry: cx / cy = the view's MIN corner (Fractal.cpp:1833-1844)
    const double *dc = (const double *)direct.data();
    fprintf(res, "init0 %u\n", r.InitializeMemory<uint32_t>(m.w0, m.h0, 1, nullptr, 0, 0, 0, false));
    fprintf(res, "direct %u\n", r.Render<uint32_t, double>(alg, dc[2], dc[3], dc[0], dc[1], m.n_iter0, 1));
    const size_t rw0 = (m.w0 + 15) / 16 * 16, rh0 = (m.h0 + 7) / 8 * 8;
    std::vector<uint32_t> it0(rw0 * rh0);
    fprintf(res, "current %u\n", r.RenderCurrent<uint32_t>(m.n_iter0, it0.data(), nullptr, nullptr, false));
    fprintf(res, "sync %u\n", r.SyncComputeStream());
    dump(d + "out_direct.bin", it0.data(), it0.size() * 4);

    // the done callback of the first frame has fired by now (the compute stream has been synchronised since)
    fprintf(res, "compute_done %d\n", r.IsComputeDone() ? 1 : 0);
    fprintf(res, "errstr %s\n", GPURenderer::ConvertErrorToString(10002));
    fclose(res);
    return 0;
}
