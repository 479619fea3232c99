#include "renderer.h"

#include <cstdio>
#include <cstring>

namespace crt {

void RenderedImage::Init(int w, int h) {
    delete[] pixels;
    width = (unsigned)w; height = (unsigned)h;
    pixels = new Color24[(size_t)w * h]();
    passes = 0;
}

bool RenderedImage::SaveImage(char const* filename) const {
    return pixels && crt_png_write_rgb8(filename, &pixels[0].r, width, height) == CRT_OK;
}

RenderedImage::~RenderedImage() { delete[] pixels; }

Renderer::Renderer(const RenderSettings& s) : settings(s) {}

Renderer::~Renderer() {
    if (worker.joinable()) worker.join();
    if (ctx) crt_destroy(ctx);
    if (scene) crt_scene_free(scene);
}

bool Renderer::LoadScene(const char* filename) {
    if (worker.joinable()) worker.join();
    if (scene) { crt_scene_free(scene); scene = nullptr; }
    const size_t len = strlen(filename);
    const bool flat = len > 9 && strcmp(filename + len - 9, ".crtscene") == 0;
    const int rc = flat ? crt_scene_load_file(filename, &scene)
                        : crt_scene_load_xml(filename, settings.asset_root.empty() ? nullptr : settings.asset_root.c_str(), &scene);
    if (rc != CRT_OK) {
        lastError = crt_host_last_error();
        printf("ERROR: %s\n", lastError.c_str());
        return false;
    }
    if (settings.width && settings.height && crt_scene_set_resolution(scene, settings.width, settings.height) != CRT_OK) {
        lastError = crt_host_last_error();
        printf("ERROR: %s\n", lastError.c_str());
        return false;
    }
    crt_scene_desc desc;
    crt_scene_get_desc(scene, &desc);
    srgb = desc.camera.srgb;
    if (!ctx && crt_create(settings.device, &ctx) != CRT_OK) {
        lastError = crt_host_last_error();
        printf("ERROR: %s\n", lastError.c_str());
        return false;
    }
    if (crt_set_tlas(ctx, settings.tlas ? 1 : 0) != CRT_OK || crt_upload_scene(ctx, &desc) != CRT_OK) {
        lastError = crt_last_error(ctx);
        printf("ERROR: %s\n", lastError.c_str());
        return false;
    }
    sceneFile = filename;
    image.Init((int)desc.camera.width, (int)desc.camera.height);
    return true;
}

void Renderer::BeginRendering(const bool wait) {
    if (worker.joinable()) worker.join();
    rendering = true;
    worker = std::thread(&Renderer::DoRendering, this);
    if (wait) worker.join();
}

void Renderer::DoRendering() {
    lastKernelMs = 0; lastRays = 0;
    image.passes = 0;
    lastOk = false;
    if (!ctx || !scene || crt_clear(ctx) != CRT_OK) {
        lastError = ctx ? crt_last_error(ctx) : "no scene loaded";
        printf("ERROR: %s\n", lastError.c_str());
        rendering = false;
        return;
    }
    bool ok = true;
    for (int pass = 0; pass < settings.passes; pass++) {
        if (crt_render_pass(ctx, pass, settings.samples, nullptr, 0) != CRT_OK ||
            crt_convert(ctx, 1.0f / static_cast<float>(pass + 1), srgb, nullptr, nullptr, 0) != CRT_OK ||
            crt_read_image(ctx, &image.pixels[0].r) != CRT_OK) {
            lastError = crt_last_error(ctx);
            printf("ERROR: %s\n", lastError.c_str());
            ok = false;
            break;
        }
        float ms = 0;
        if (crt_last_kernel_ms(ctx, &ms) == CRT_OK) lastKernelMs += ms;
        image.passes++;
    }
    crt_stats st;
    if (crt_get_stats(ctx, &st) == CRT_OK) lastRays = st.closest_rays + st.shadow_rays;
    else if (ok) {
        // (a kernel that gave up shows here at the latest: the counters carry its error count)
        lastError = crt_last_error(ctx);
        printf("ERROR: %s\n", lastError.c_str());
        ok = false;
    }
    lastOk = ok;
    image.passes++;
    rendering = false;
}

}  // namespace crt
