// C++ front end with the reference's public shape (class Renderer and struct
// RenderedImage, renderer/renderer.cuh:26-98), implemented on the C ABI of
// include/crt_amd.h only.  SAMPLES and PASSES (renderer/settings.cuh:12,15)
// are run-time settings here, and the image size can be overridden.
#pragma once
#include <stdint.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/crt_amd.h"

namespace crt {

struct Color24 {
    uint8_t r, g, b;
};

struct RenderedImage {
    unsigned int width = 0, height = 0;
    Color24* pixels = nullptr;
    std::atomic<unsigned int> passes{0};

    void Init(int w, int h);
    bool SaveImage(char const* filename) const;
    ~RenderedImage();
};

struct RenderSettings {
    int samples = 256;        // samples per pixel per pass
    int passes = 8;
    uint32_t width = 0;       // 0 = keep the scene's size
    uint32_t height = 0;
    int device = 0;
    std::string asset_root;   // where relative mesh / texture paths resolve; empty = current directory
    bool tlas = false;        // walk a BVH over the scene's object nodes instead of scanning the node list (crt_set_tlas: faster on scenes of hundreds of nodes, images no longer the reference's bit for bit)
};

class Renderer {
public:
    explicit Renderer(const RenderSettings& settings = RenderSettings());
    ~Renderer();

    // Loads an XML scene (or a flat ".crtscene" file).  Returns false and prints the
    // reason on failure, as the reference does (lib/xmlload.cu:106-149).
    bool LoadScene(const char* filename);

    const RenderedImage& GetImage() const { return image; }
    std::string SceneFileName() const { return sceneFile; }
    bool IsRendering() const { return rendering.load(); }

    // Renders every pass on a worker thread; joins it when wait is true
    // (renderer/renderer.cu:7-14).  The image is refreshed after each pass and
    // image.passes ends at settings.passes + 1 like the reference's counter.
    void BeginRendering(bool wait = false);

    // last render: kernel milliseconds summed over passes and rays traced
    double LastRenderKernelMs() const { return lastKernelMs; }
    uint64_t LastRenderRays() const { return lastRays; }
    const std::string& LastError() const { return lastError; }
    // false when the last BeginRendering did not complete every pass (the image is then incomplete)
    bool LastRenderOk() const { return lastOk.load(); }

private:
    void DoRendering();

    RenderSettings settings;
    std::string sceneFile;
    RenderedImage image;
    std::atomic<bool> rendering{false};
    std::thread worker;
    crt_scene* scene = nullptr;
    crt_ctx* ctx = nullptr;
    int srgb = 0;
    double lastKernelMs = 0;
    uint64_t lastRays = 0;
    std::string lastError;
    std::atomic<bool> lastOk{false};
};

}  // namespace crt
