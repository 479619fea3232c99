// crt_render: command-line front end.  Replaces main.cu:8-21 (hard-coded scene +
// viewport) and, with --sequence, Sequence::DoRender (renderer/sequence.h:63-99).
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "renderer.h"

static void usage() {
    printf("usage: crt_render <scene.xml|scene.crtscene> [-o out.png] [--spp N] [--passes N] [--width W --height H]\n"
           "                  [--assets DIR] [--device N] [--tlas]\n"
           "           --tlas: walk a BVH over the scene's object nodes instead of scanning the node list; for scenes of hundreds\n"
           "           of nodes and more, images are then no longer the reference's bit for bit\n"
           "       crt_render --sequence <dir> [--out-dir DIR] [--devices N] [--workers-per-device K] [same options]\n"
           "           every *.xml (or *.crtscene) of <dir>, one frame per worker: frames are handed out from a queue to\n"
           "           N x K workers, worker w renders on device w %% N (K > 1 overlaps one frame's scene load and BVH\n"
           "           build with another frame's kernels on the same GPU)\n"
           "       crt_render --bake <scene.xml> <out.crtscene> [--assets DIR] [--width W --height H]\n");
}

static std::mutex g_print;

// One frame from scratch, as Sequence::DoRender does it (renderer/sequence.h:79-88): load, render every pass, save;
// the report line carries the frame's load, render and save times.
static bool render_one(crt::Renderer& r, const crt::RenderSettings& st, const std::string& scene, const std::string& out) {
    const auto t0 = std::chrono::steady_clock::now();
    if (!r.LoadScene(scene.c_str())) return false;
    const auto t1 = std::chrono::steady_clock::now();
    r.BeginRendering(true);
    const auto t2 = std::chrono::steady_clock::now();
    if (!r.LastRenderOk()) {      // no image file for a frame that is not complete
        std::lock_guard<std::mutex> guard(g_print);
        printf("ERROR: %s was not rendered: %s\n", scene.c_str(), r.LastError().c_str());
        return false;
    }
    const bool saved = r.GetImage().SaveImage(out.c_str());
    const auto t3 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::lock_guard<std::mutex> guard(g_print);
    if (!saved) { printf("ERROR: cannot write %s\n", out.c_str()); return false; }
    printf("%s -> %s: %ux%u, %d x %d spp, device %d, load %.1f ms, render %.2f ms (kernels %.2f ms), save %.1f ms, %.1f Mrays/s\n", scene.c_str(), out.c_str(),
           r.GetImage().width, r.GetImage().height, st.passes, st.samples, st.device, ms(t0, t1), ms(t1, t2), r.LastRenderKernelMs(), ms(t2, t3),
           r.LastRenderKernelMs() > 0 ? (double)r.LastRenderRays() / r.LastRenderKernelMs() * 1e-3 : 0.0);
    fflush(stdout);
    return true;
}

static bool has_suffix(const std::string& n, const char* suffix) {
    const size_t k = strlen(suffix);
    return n.size() > k && n.compare(n.size() - k, k, suffix) == 0;
}

int main(int argc, char** argv) {
    crt::RenderSettings st;
    st.samples = 64; st.passes = 1;
    std::string scene, out = "output.png", seq_dir, out_dir = "output", bake_in, bake_out;
    int devices = 1, workers_per_device = 1;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { usage(); exit(2); } return argv[++i]; };
        if (a == "-o") out = val();
        else if (a == "--spp") st.samples = atoi(val());
        else if (a == "--passes") st.passes = atoi(val());
        else if (a == "--width") st.width = (uint32_t)atoi(val());
        else if (a == "--height") st.height = (uint32_t)atoi(val());
        else if (a == "--assets") st.asset_root = val();
        else if (a == "--device") st.device = atoi(val());
        else if (a == "--tlas") st.tlas = true;
        else if (a == "--sequence") seq_dir = val();
        else if (a == "--out-dir") out_dir = val();
        else if (a == "--devices") devices = std::max(1, atoi(val()));
        else if (a == "--workers-per-device") workers_per_device = std::max(1, atoi(val()));
        else if (a == "--bake") { bake_in = val(); bake_out = val(); }
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else scene = a;
    }
    if (!bake_in.empty()) {
        crt_scene* s = nullptr;
        if (crt_scene_load_xml(bake_in.c_str(), st.asset_root.empty() ? nullptr : st.asset_root.c_str(), &s) != CRT_OK) { printf("ERROR: %s\n", crt_host_last_error()); return 1; }
        if (st.width && st.height) crt_scene_set_resolution(s, st.width, st.height);
        if (crt_scene_save_file(s, bake_out.c_str()) != CRT_OK) { printf("ERROR: %s\n", crt_host_last_error()); return 1; }
        crt_scene_free(s);
        return 0;
    }
    if (!seq_dir.empty()) {
        std::vector<std::string> files;
        if (DIR* d = opendir(seq_dir.c_str())) {
            while (dirent* e = readdir(d)) {
                const std::string n = e->d_name;
                if (has_suffix(n, ".xml") || has_suffix(n, ".crtscene")) files.push_back(n);
            }
            closedir(d);
        } else {
            printf("ERROR: cannot read directory %s\n", seq_dir.c_str());
            return 1;
        }
        std::sort(files.begin(), files.end());
        // a worker on a device that does not exist would take frames from the queue only to fail them
        int have = 0;
        if (crt_device_count(&have) != CRT_OK || st.device < 0 || st.device + devices > have) {
            printf("ERROR: --device %d --devices %d asks for GPUs %d..%d, the machine has %d\n", st.device, devices, st.device, st.device + devices - 1, have);
            return 2;
        }
        mkdir(out_dir.c_str(), 0777);      // (an existing directory is fine)
        // Frames are independent: a queue of frames, one worker per (device, slot); each worker keeps one renderer
        // (one device context) and renders whole frames from scratch.
        const auto t0 = std::chrono::steady_clock::now();
        std::atomic<size_t> next{0};
        std::atomic<int> failed{0};
        const int workers = devices * workers_per_device;
        std::vector<std::thread> pool;
        for (int w = 0; w < workers; w++)
            pool.emplace_back([&, w]() {
                crt::RenderSettings mine = st;
                mine.device = st.device + w % devices;
                crt::Renderer r(mine);
                for (size_t i = next++; i < files.size(); i = next++) {
                    const std::string& f = files[i];
                    const std::string stem = f.substr(0, f.rfind('.'));
                    if (!render_one(r, mine, seq_dir + "/" + f, out_dir + "/" + stem + ".png")) failed++;
                }
            });
        for (auto& t : pool) t.join();
        const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("Total render time is %.3f s for %zu frames on %d device(s) x %d worker(s): %.2f frames/s (%d failed).\n",
               total, files.size(), devices, workers_per_device, total > 0 ? files.size() / total : 0.0, failed.load());
        return failed ? 1 : 0;
    }
    if (scene.empty()) { usage(); return 2; }
    crt::Renderer r(st);
    return render_one(r, st, scene, out) ? 0 : 1;
}
