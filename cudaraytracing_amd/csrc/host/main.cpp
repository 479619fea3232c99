// crt_render: command-line front end.  Replaces main.cu:8-21 (hard-coded scene +
// viewport) and, with --sequence, Sequence::DoRender (renderer/sequence.h:63-99).
#include <dirent.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "renderer.h"

static void usage() {
    printf("usage: crt_render <scene.xml|scene.crtscene> [-o out.png] [--spp N] [--passes N] [--width W --height H]\n"
           "                  [--assets DIR] [--device N]\n"
           "       crt_render --sequence <dir> [--out-dir DIR] [same options]\n"
           "       crt_render --bake <scene.xml> <out.crtscene> [--assets DIR] [--width W --height H]\n");
}

static bool render_one(const crt::RenderSettings& st, const std::string& scene, const std::string& out) {
    crt::Renderer r(st);
    if (!r.LoadScene(scene.c_str())) return false;
    const auto t0 = std::chrono::steady_clock::now();
    r.BeginRendering(true);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (!r.GetImage().SaveImage(out.c_str())) { printf("ERROR: cannot write %s\n", out.c_str()); return false; }
    printf("%s -> %s: %ux%u, %d x %d spp, frame %.2f ms (kernels %.2f ms), %.1f Mrays/s\n", scene.c_str(), out.c_str(),
           r.GetImage().width, r.GetImage().height, st.passes, st.samples, ms, r.LastRenderKernelMs(),
           r.LastRenderKernelMs() > 0 ? (double)r.LastRenderRays() / r.LastRenderKernelMs() * 1e-3 : 0.0);
    return true;
}

int main(int argc, char** argv) {
    crt::RenderSettings st;
    st.samples = 64; st.passes = 1;
    std::string scene, out = "output.png", seq_dir, out_dir = "output", bake_in, bake_out;
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
        else if (a == "--sequence") seq_dir = val();
        else if (a == "--out-dir") out_dir = val();
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
                if (n.size() > 4 && n.substr(n.size() - 4) == ".xml") files.push_back(n);
            }
            closedir(d);
        }
        std::sort(files.begin(), files.end());
        const auto t0 = std::chrono::steady_clock::now();
        int failed = 0;
        for (const std::string& f : files)
            if (!render_one(st, seq_dir + "/" + f, out_dir + "/" + f.substr(0, f.size() - 4) + ".png")) failed++;
        printf("Total render time is %.3f s for %zu frames (%d failed).\n",
               std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), files.size(), failed);
        return failed ? 1 : 0;
    }
    if (scene.empty()) { usage(); return 2; }
    return render_one(st, scene, out) ? 0 : 1;
}
