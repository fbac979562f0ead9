// veneer_driver.cpp -- exercises the C++ veneer (include/crychic/*.h) exactly as reference call sites would:
// CRYCHIC::Initialize / Update / Draw, UploadBuffer::CopyData, Ssao::ComputeSsao ...  Input planes come from raw
// files written by the pytest side (tests/test_cpp_veneer.py); outputs go back as raw files for comparison with
// the oracle.  Usage: veneer_driver <dir> <W> <H> <shadowDim> <cubeDim> <blurCount> <numDirLights> [scene]
// With the trailing word `scene` the application also runs its own producer passes on its built-in scene (100 boxes +
// grid) instead of loading depth / normal / G-buffer / shadow planes from <dir>.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>
#include "crychic/CRYCHIC.h"

static std::vector<char> slurp(const std::string& p)
{
    std::ifstream f(p, std::ios::binary);
    if (!f) { std::fprintf(stderr, "cannot open %s\n", p.c_str()); std::exit(2); }
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void dump(const std::string& p, const void* d, size_t n)
{
    std::ofstream f(p, std::ios::binary);
    f.write(static_cast<const char*>(d), (std::streamsize)n);
}
static void put(ID3D12Resource* r, const std::string& path, hipStream_t s)
{
    auto b = slurp(path);
    if (b.size() != r->Bytes()) { std::fprintf(stderr, "%s: %zu bytes, resource has %zu\n", path.c_str(), b.size(), r->Bytes()); std::exit(2); }
    r->Upload(b.data(), b.size(), s);
    CrychicHipThrowIfFailed(hipStreamSynchronize(s));
}

int main(int argc, char** argv)
{
    if (argc < 8) { std::fprintf(stderr, "usage\n"); return 2; }
    const std::string dir = argv[1];
    const UINT W = std::atoi(argv[2]), H = std::atoi(argv[3]), SD = std::atoi(argv[4]), CD = std::atoi(argv[5]);
    try {
        CRYCHIC app(0, W, H);
        app.mShadowMapSize = SD;
        app.mBlurCount = std::atoi(argv[6]);
        app.mNumDirLights = std::atoi(argv[7]);
        app.mSkyEnabled = true;
        const bool sceneMode = argc > 8 && std::string(argv[8]) == "scene";
        app.mRunProducerPasses = sceneMode;
        if (!app.Initialize()) return 3;
        hipStream_t s = app.CommandList()->Stream();
        if (!sceneMode) {
            put(app.DepthStencilBuffer(), dir + "/depth.bin", s);
            put(app.mSsao->NormalMap(), dir + "/normal.bin", s);
            for (int i = 0; i < 3; ++i) put(app.mDeferred->Resource(i), dir + "/g" + std::to_string(i) + ".bin", s);
            for (int i = 0; i < 4; ++i) put(app.mShadowMap->Resource(i), dir + "/shadow" + std::to_string(i) + ".bin", s);
        }
        auto cube = std::make_unique<ID3D12Resource>((size_t)6 * CD * CD * 4, ID3D12Resource::DEFAULT_HEAP);
        put(cube.get(), dir + "/cube.bin", s);
        app.SetCubeMap(std::move(cube), CD);
        if (argc > 9) app.LoadTextures(argv[9], true);       // CRYCHIC::LoadTextures: six material textures + the sky cube map from a directory

        GameTimer gt;
        for (int frame = 0; frame < 5; ++frame) {  // cycles the 3-deep frame-resource ring and its fences
            gt.Tick(1.0f / 60.0f);
            app.Update(gt);
            app.Draw(gt);
        }
        app.CommandList()->Flush();

        std::vector<uint8_t> out((size_t)W * H * 4);
        app.CurrentBackBuffer()->Download(out.data(), out.size(), s);
        std::vector<uint16_t> ao((size_t)(W / 2) * (H / 2));
        app.mSsao->AmbientMap()->Download(ao.data(), ao.size() * 2, s);
        app.CommandList()->Flush();
        {   // the reference's literal pass sequence (one SSAO pass, then BlurAmbientMap sweep by sweep) gives the same frame
            app.mSsao->mLiteralSequence = true;
            gt.Tick(1.0f / 60.0f);
            app.Update(gt);
            app.Draw(gt);
            std::vector<uint8_t> out2(out.size());
            std::vector<uint16_t> ao2(ao.size());
            app.CurrentBackBuffer()->Download(out2.data(), out2.size(), s);
            app.mSsao->AmbientMap()->Download(ao2.data(), ao2.size() * 2, s);
            app.CommandList()->Flush();
            app.mSsao->mLiteralSequence = false;
            if (out2 != out || ao2 != ao) { std::fprintf(stderr, "literal pass sequence differs from crychic_ssao_compute\n"); return 6; }
        }
        std::vector<uint8_t> rv(256 * 256 * 4);
        app.mSsao->RandomVectorMap()->Download(rv.data(), rv.size(), s);
        app.CommandList()->Flush();
        if (sceneMode) {   // the planes the producer passes rendered
            auto grab = [&](ID3D12Resource* r, const std::string& name) {
                std::vector<char> h(r->Bytes());
                r->Download(h.data(), h.size(), s);
                app.CommandList()->Flush();
                dump(dir + "/" + name, h.data(), h.size());
            };
            grab(app.DepthStencilBuffer(), "depth_out.bin");
            grab(app.mSsao->NormalMap(), "normal_out.bin");
            for (int i = 0; i < 3; ++i) grab(app.mDeferred->Resource(i), "g" + std::to_string(i) + "_out.bin");
            for (int i = 0; i < 4; ++i) grab(app.mShadowMap->Resource(i), "shadow" + std::to_string(i) + "_out.bin");
        }
        dump(dir + "/out.bin", out.data(), out.size());
        dump(dir + "/ao.bin", ao.data(), ao.size() * 2);
        dump(dir + "/randvec.bin", rv.data(), rv.size());
        dump(dir + "/pass_cb.bin", &app.mCurrFrameResource->PassCB->Element(0), sizeof(PassConstants));
        dump(dir + "/ssao_cb.bin", &app.mCurrFrameResource->SsaoCB->Element(0), sizeof(SsaoConstants));

        // error behaviour mirrors ThrowIfFailed -> DxException
        bool threw = false;
        try { app.mSsao->CalcGaussWeights(3.0f); } catch (const CrychicException& e) { threw = e.Status == CRYCHIC_E_INVALID_ARG; }
        if (!threw) { std::fprintf(stderr, "CalcGaussWeights(3.0) did not throw\n"); return 4; }
        auto w = app.mSsao->CalcGaussWeights(2.5f);
        if (w.size() != 11) return 5;
        std::printf("veneer ok %ux%u\n", W, H);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
