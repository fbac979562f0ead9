// prof_driver.cpp -- torch-free frame loop for rocprofv3 runs (PMC collection serialises every dispatch, so the
// profiled process must launch nothing but the hot-path kernels).  Loads the planes and constant buffers that
// `python bench.py --dump-scene DIR` wrote and replays crychic_draw_hot_path.
//   tools/prof_driver DIR [frames=20] [warmup=3]
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>
#include "crychic_hip.h"

#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
#define CHK(x) do { int st_ = (x); if (st_ < 0) { std::fprintf(stderr, "%s: %s\n", #x, crychic_last_error()); std::exit(1); } } while (0)

static void* load(const std::string& path, size_t expect)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); std::exit(2); }
    size_t n = (size_t)f.tellg();
    if (expect && n != expect) { std::fprintf(stderr, "%s: %zu bytes, expected %zu\n", path.c_str(), n, expect); std::exit(2); }
    std::vector<char> h(n);
    f.seekg(0); f.read(h.data(), (std::streamsize)n);
    void* d = nullptr;
    HIPCHK(hipMalloc(&d, n));
    HIPCHK(hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice));
    return d;
}
template <class T> static T load_struct(const std::string& path)
{
    T t;
    std::ifstream f(path, std::ios::binary);
    if (!f.read(reinterpret_cast<char*>(&t), sizeof t)) { std::fprintf(stderr, "cannot read %s\n", path.c_str()); std::exit(2); }
    return t;
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: prof_driver DIR [frames] [warmup]\n"); return 2; }
    const std::string dir = argv[1];
    const int frames = argc > 2 ? std::atoi(argv[2]) : 20, warmup = argc > 3 ? std::atoi(argv[3]) : 3;
    unsigned W, H, SD, CD; int blur, lights; float radius; unsigned flags;
    {
        std::ifstream m(dir + "/meta.txt");
        if (!(m >> W >> H >> SD >> CD >> blur >> lights >> radius >> flags)) { std::fprintf(stderr, "bad meta.txt\n"); return 2; }
    }
    crychic_ctx* ctx = nullptr;
    CHK(crychic_ctx_create(0, &ctx));
    const size_t N = (size_t)W * H;
    crychic_frame_desc f = {};
    f.W = W; f.H = H; f.blurCount = blur; f.numDirLights = lights; f.pcfSearchRadius = radius; f.flags = flags; f.row0 = 0; f.rows = H;
    f.normal_dev = load(dir + "/normal.bin", N * 8);
    f.depth_dev = (const uint32_t*)load(dir + "/depth.bin", N * 4);
    f.randvec_dev = (const uint8_t*)load(dir + "/randvec.bin", 256 * 256 * 4);
    f.g0_dev = (const float*)load(dir + "/g0.bin", N * 16);
    f.g1_dev = (const float*)load(dir + "/g1.bin", N * 16);
    f.g2_dev = (const float*)load(dir + "/g2.bin", N * 16);
    for (int i = 0; i < 4; ++i) f.shadow_dev[i] = (const uint32_t*)load(dir + "/shadow" + std::to_string(i) + ".bin", (size_t)SD * SD * 4);
    f.shadowDim = SD;
    f.cube_dev = (const uint8_t*)load(dir + "/cube.bin", (size_t)6 * CD * CD * 4);
    f.cubeDim = CD;
    void *a0, *a1, *edge, *out;
    HIPCHK(hipMalloc(&a0, N / 2)); HIPCHK(hipMalloc(&a1, N / 2));
    HIPCHK(hipMalloc(&edge, crychic_edge_plane_bytes(W, H)));
    HIPCHK(hipMalloc(&out, N * 4));
    f.ambient0_dev = (uint16_t*)a0; f.ambient1_dev = (uint16_t*)a1; f.edge_dev = edge; f.out_rgba8_dev = (uint8_t*)out;
    const crychic_ssao_constants scb = load_struct<crychic_ssao_constants>(dir + "/ssao_cb.bin");
    const crychic_pass_constants pcb = load_struct<crychic_pass_constants>(dir + "/pass_cb.bin");
    hipStream_t s;
    HIPCHK(hipStreamCreate(&s));
    for (int i = 0; i < warmup; ++i) CHK(crychic_draw_hot_path(ctx, &scb, &pcb, &f, s));
    HIPCHK(hipStreamSynchronize(s));
    CHK(crychic_ctx_set_profiling(ctx, 1));
    crychic_pass_times acc = {};
    for (int i = 0; i < frames; ++i) {
        CHK(crychic_draw_hot_path(ctx, &scb, &pcb, &f, s));
        crychic_pass_times t;
        CHK(crychic_ctx_last_pass_times(ctx, &t));
        acc.ssao_ms += t.ssao_ms / frames; acc.blur_ms += t.blur_ms / frames; acc.light_ms += t.light_ms / frames; acc.total_ms += t.total_ms / frames;
    }
    std::vector<unsigned char> host(N * 4);
    HIPCHK(hipMemcpy(host.data(), out, N * 4, hipMemcpyDeviceToHost));
    unsigned long long sum = 0;
    for (unsigned char c : host) sum += c;
    std::printf("{\"W\": %u, \"H\": %u, \"frames\": %d, \"ssao_ms\": %.4f, \"blur_ms\": %.4f, \"light_ms\": %.4f, \"total_ms\": %.4f, \"checksum\": %llu}\n",
                W, H, frames, acc.ssao_ms, acc.blur_ms, acc.light_ms, acc.total_ms, sum);
    crychic_ctx_destroy(ctx);
    return 0;
}
