// mgpu_driver.cpp -- a C++ host rendering ONE frame on several GPUs through the veneer and the C ABI, no Python, no torch:
// every rank runs CRYCHIC::Initialize / Update / Draw on its own scene (producer passes on the device), joined by
// CRYCHIC::JoinNode, so Draw renders the rank's row strip and crychic_allgather_frame (RCCL) completes the back buffer.
//
//   mgpu_driver rank <nranks> <rank> <idfile> <dir> <W> <H> [ragged] [parts=N]   one process per GPU (rank r uses device r); rank 0 writes
//                                                                      the 128-byte rendezvous id to <idfile>, the others wait for it
//   mgpu_driver all <nranks> <dir> <W> <H>                             one process, one thread, every GPU (crychic_comm_create_all)
//
// Every rank writes <dir>/frame_<rank>.bin (the gathered frame); rank 0 also renders the whole frame alone into
// <dir>/frame_single.bin.  The pytest side compares them byte for byte.  `ragged` uses strips of different heights.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
#include "crychic/CRYCHIC.h"

static void dump(const std::string& p, const void* d, size_t n)
{
    std::ofstream f(p, std::ios::binary);
    f.write(static_cast<const char*>(d), (std::streamsize)n);
}

static std::unique_ptr<CRYCHIC> make_app(int device, UINT W, UINT H)
{
    auto app = std::make_unique<CRYCHIC>(device, W, H);
    app->mShadowMapSize = 512;
    app->mBlurCount = 3;
    app->mNumDirLights = 1;
    if (!app->Initialize()) throw std::runtime_error("Initialize failed");
    const UINT CD = 16;   // a small procedural cube map: sky gradient by face
    std::vector<uint8_t> cube((size_t)6 * CD * CD * 4);
    for (size_t i = 0; i < cube.size(); i += 4) {
        const size_t face = i / ((size_t)CD * CD * 4);
        cube[i] = (uint8_t)(60 + 30 * face); cube[i + 1] = (uint8_t)(120 + 10 * face); cube[i + 2] = 230; cube[i + 3] = 255;
    }
    auto res = std::make_unique<ID3D12Resource>(cube.size(), ID3D12Resource::DEFAULT_HEAP);
    res->Upload(cube.data(), cube.size(), app->CommandList()->Stream());
    app->CommandList()->Flush();
    app->SetCubeMap(std::move(res), CD);
    return app;
}

static std::vector<uint8_t> frame_of(CRYCHIC& app, UINT W, UINT H)
{
    std::vector<uint8_t> out((size_t)W * H * 4);
    app.CurrentBackBuffer()->Download(out.data(), out.size(), app.CommandList()->Stream());
    app.CommandList()->Flush();
    return out;
}

static std::vector<uint32_t> ragged_bounds(int nranks, UINT H)
{
    // rank r gets weight r + 1 (in row pairs), the last rank the remainder: heights differ, every strip keeps >= 2 rows
    std::vector<uint32_t> b;
    const uint32_t pairs = H / 2;
    uint32_t total = 0, at = 0;
    for (int r = 0; r < nranks; ++r) total += (uint32_t)r + 1;
    for (int r = 0; r < nranks; ++r) {
        uint32_t n = r == nranks - 1 ? pairs - at : std::max<uint32_t>(1, pairs * ((uint32_t)r + 1) / total);
        b.push_back(2 * at); b.push_back(2 * n);
        at += n;
    }
    return b;
}

int main(int argc, char** argv)
{
    try {
        const std::string mode = argc > 1 ? argv[1] : "";
        GameTimer gt;
        if (mode == "rank" && argc >= 8) {
            const int nranks = std::atoi(argv[2]), rank = std::atoi(argv[3]);
            const std::string idfile = argv[4], dir = argv[5];
            const UINT W = std::atoi(argv[6]), H = std::atoi(argv[7]);
            bool ragged = false;
            UINT parts = 1;
            for (int a = 8; a < argc; ++a) {
                const std::string o = argv[a];
                if (o == "ragged") ragged = true;
                else if (o.rfind("parts=", 0) == 0) parts = (UINT)std::atoi(o.c_str() + 6);      // CRYCHIC::SetExchangeParts
            }
            uint8_t id[CRYCHIC_COMM_ID_BYTES];
            if (rank == 0) {
                CrychicThrowIfFailed(crychic_comm_unique_id(id));
                dump(idfile + ".tmp", id, sizeof id);
                std::rename((idfile + ".tmp").c_str(), idfile.c_str());
            } else {
                bool got = false;
                for (int t = 0; t < 1200 && !got; ++t) {          // up to 60 s for rank 0
                    std::ifstream f(idfile, std::ios::binary);
                    got = f && f.read(reinterpret_cast<char*>(id), sizeof id);
                    if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(50));
                }
                if (!got) { std::fprintf(stderr, "rank %d: no rendezvous id in %s\n", rank, idfile.c_str()); return 3; }
            }
            auto app = make_app(rank, W, H);
            if (rank == 0) {           // the frame one GPU renders alone
                gt.Tick(1.0f / 60.0f); app->Update(gt); app->Draw(gt);
                dump(dir + "/frame_single.bin", frame_of(*app, W, H).data(), (size_t)W * H * 4);
            }
            app->JoinNode(nranks, rank, id, ragged ? ragged_bounds(nranks, H) : std::vector<uint32_t>{});
            app->SetExchangeParts(parts);
            for (int frame = 0; frame < 3; ++frame) { gt.Tick(1.0f / 60.0f); app->Update(gt); app->Draw(gt); }
            dump(dir + "/frame_" + std::to_string(rank) + ".bin", frame_of(*app, W, H).data(), (size_t)W * H * 4);
            app->LeaveNode();
            std::printf("mgpu rank %d/%d ok\n", rank, nranks);
            return 0;
        }
        if (mode == "all" && argc >= 6) {
            const int nranks = std::atoi(argv[2]);
            const std::string dir = argv[3];
            const UINT W = std::atoi(argv[4]), H = std::atoi(argv[5]);
            std::vector<std::unique_ptr<CRYCHIC>> apps;
            for (int r = 0; r < nranks; ++r) apps.push_back(make_app(r, W, H));
            gt.Tick(1.0f / 60.0f);
            apps[0]->Update(gt); apps[0]->Draw(gt);
            dump(dir + "/frame_single.bin", frame_of(*apps[0], W, H).data(), (size_t)W * H * 4);
            // one thread owns every rank: strips via SetStrip, the exchange via the *_all entry points
            std::vector<crychic_ctx*> ctxs;
            std::vector<crychic_comm*> comms((size_t)nranks, nullptr);
            for (auto& a : apps) ctxs.push_back(a->Device()->Ctx());
            CrychicThrowIfFailed(crychic_comm_create_all(ctxs.data(), nranks, comms.data()));
            std::vector<uint8_t*> frames;
            std::vector<void*> streams;
            for (int r = 0; r < nranks; ++r) {
                uint32_t r0, rn;
                CrychicThrowIfFailed(crychic_strip_rows(H, nranks, r, &r0, &rn));
                apps[(size_t)r]->SetStrip(r0, rn);
                apps[(size_t)r]->Update(gt);
                apps[(size_t)r]->Draw(gt);
                frames.push_back(static_cast<uint8_t*>(apps[(size_t)r]->CurrentBackBuffer()->Data()));
                streams.push_back(apps[(size_t)r]->CommandList()->Stream());
            }
            CrychicThrowIfFailed(crychic_allgather_frame_all(comms.data(), nranks, frames.data(), W, H, nullptr, streams.data()));
            for (int r = 0; r < nranks; ++r) dump(dir + "/frame_" + std::to_string(r) + ".bin", frame_of(*apps[(size_t)r], W, H).data(), (size_t)W * H * 4);
            for (auto c : comms) crychic_comm_destroy(c);
            std::printf("mgpu all %d ok\n", nranks);
            return 0;
        }
        std::fprintf(stderr, "usage: mgpu_driver rank <nranks> <rank> <idfile> <dir> <W> <H> [ragged] [parts=N] | all <nranks> <dir> <W> <H>\n");
        return 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
}
