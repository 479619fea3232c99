/* crt_amd.h - C ABI of the MI355X renderer core (libcrt_amd.so).
 *
 * The reference has no FFI: its per-pixel trace path sits behind a C++ class
 * and one global (`class Renderer`, renderer/renderer.cuh:50-98;
 * `__managed__ Scene theScene`, renderer/renderer.cuh:45).  This header is the
 * boundary a maintainer binds instead: plain pointers and sizes, every call
 * returns 0 on success or a negative CRT_E_* code, and crt_last_error() gives
 * the message.  Calls on one context must come from one thread at a time; all
 * calls block until their GPU work has finished unless noted.
 */
#ifndef CRT_AMD_H
#define CRT_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRT_OK 0
#define CRT_E_INVALID (-1)   /* bad argument / shape mismatch */
#define CRT_E_IO (-2)        /* file could not be read / written / parsed */
#define CRT_E_HIP (-3)       /* a HIP runtime call failed */
#define CRT_E_NOSCENE (-4)   /* render call before crt_upload_scene */
#define CRT_E_LIMIT (-5)     /* BVH deeper than 64 levels (bvh.cuh:6) or similar */

/* ---- flat scene description (POD; replaces the pointer graph under
 *      `Scene`, scene/scene.cuh:84-114) ---------------------------------- */

enum { CRT_OBJ_NONE = 0, CRT_OBJ_SPHERE = 1, CRT_OBJ_PLANE = 2, CRT_OBJ_MESH = 3 };
enum { CRT_LIGHT_AMBIENT = 0, CRT_LIGHT_DIRECT = 1, CRT_LIGHT_POINT = 2 };

/* Camera + RenderInfo (scene.cuh:25-79; derived fields as computed by
 * RenderInfo::Load, lib/xmlload.cu:324-354). */
typedef struct crt_camera {
    uint32_t width, height;
    float position[3];
    float dof;
    int32_t srgb;
    float pixel_size;
    float cX[3], cY[3], cZ[3];
    float top_left_pixel[3];
    /* inputs kept so the image size can be changed and the frame re-derived */
    float target[3], up[3], fov;
} crt_camera;

/* Node (scene.cuh:137-189), depth-first order, node 0 is the object-less root. */
typedef struct crt_node {
    uint32_t child_count;
    int32_t object_kind; /* CRT_OBJ_* */
    int32_t mesh_id;     /* index into meshes when object_kind == CRT_OBJ_MESH */
    int32_t material_id;
    float box_min[3], box_max[3];
    float tm[12];  /* local -> world, column major 3x4 */
    float itm[12]; /* world -> local */
} crt_node;

/* Material (scene.cuh:116-132); textures are indices into the texture table. */
typedef struct crt_material {
    int32_t diffuse, specular, refraction, emission;
    float glossiness;
    float ior;
    float absorption[3];
} crt_material;

/* Light (scene/lights.cuh:27-72). `vec` is the position (point) or the
 * direction (direct); ambient ignores it. */
typedef struct crt_light {
    int32_t kind; /* CRT_LIGHT_* */
    float intensity[3];
    float vec[3];
    float size;
} crt_light;

/* 8-bit RGB texture (scene/texture.cuh:5-16), row major, 3 bytes per texel. */
typedef struct crt_texture {
    uint32_t width, height;
    const uint8_t* rgb;
} crt_texture;

/* Triangle mesh (scene/mesh.cuh:6-20). Index triples are 0-based. */
typedef struct crt_mesh {
    uint32_t nv, nf, nvn, nvt;
    const float* v;     /* nv * 3 */
    const uint32_t* f;  /* nf * 3 */
    const float* vn;    /* nvn * 3 or NULL */
    const uint32_t* fn; /* nf * 3 or NULL */
    const float* vt;    /* nvt * 3 or NULL */
    const uint32_t* ft; /* nf * 3 or NULL */
    float box_min[3], box_max[3];
} crt_mesh;

typedef struct crt_scene_desc {
    crt_camera camera;
    const crt_node* nodes;
    uint32_t node_count;
    const crt_material* materials;
    uint32_t material_count;
    const crt_light* lights;
    uint32_t light_count;
    const crt_texture* textures;
    uint32_t texture_count;
    const crt_mesh* meshes;
    uint32_t mesh_count;
    int32_t env_texture;
} crt_scene_desc;

/* ---- host scene model --------------------------------------------------- */

typedef struct crt_scene crt_scene; /* owns every array a crt_scene_desc points at */

/* Parse a scene XML and everything it references (OBJ meshes, PNG textures).
 * Replaces Renderer::LoadScene (lib/xmlload.cu:106-149).  Relative asset paths
 * are resolved against asset_root (NULL = current directory, as the
 * reference does).  A missing texture falls back to 1x1 white times the
 * colour attributes (lib/xmlload.cuh:101-117). */
int crt_scene_load_xml(const char* xml_path, const char* asset_root, crt_scene** out);

/* Flat binary container of a crt_scene (this project's own format). */
int crt_scene_load_file(const char* path, crt_scene** out);
int crt_scene_save_file(const crt_scene* scene, const char* path);

void crt_scene_free(crt_scene* scene);

/* Fill *out with pointers into the scene; valid until crt_scene_free. */
int crt_scene_get_desc(const crt_scene* scene, crt_scene_desc* out);

/* Change the image size and re-derive pixel_size / top_left_pixel the way
 * RenderInfo::Load does (lib/xmlload.cu:324-354). */
int crt_scene_set_resolution(crt_scene* scene, uint32_t width, uint32_t height);

/* Synthetic stress scene: `copies` scaled, z-rotated copies of mesh
 * `mesh_index` of `source` on a jittered 3-D grid, flattened into ONE mesh (one
 * BVH), over a plane floor with one point light; the placement is a fixed
 * function of `seed` (see csrc/host/scene_synth.cpp).  The reference ships no
 * such scene; 1582 copies of its 6320-triangle teapot give the ten-million-
 * triangle workload that no cache holds.  The BVH over the result is built by
 * the same deterministic builder as any other mesh (renderer/bvh.cu:7-35). */
int crt_scene_make_instanced(const crt_scene* source, uint32_t mesh_index, uint32_t copies, uint32_t seed,
                             uint32_t width, uint32_t height, crt_scene** out);

/* Recompute the derived camera fields of `cam` from position/target/up/fov/
 * width/height (same arithmetic as crt_scene_set_resolution). */
int crt_camera_derive(crt_camera* cam);

/* Message for the last failing crt_scene_ / crt_bvh_ / crt_png_ call on the
 * calling thread. */
const char* crt_host_last_error(void);

/* ---- BVH builder (replaces BVH::ConstructBVH, renderer/bvh.cu:7-35) ------ */

typedef struct crt_bvh crt_bvh;

/* The node array in the reference's own shape (BVHNode, renderer/bvh.cuh:13-35)
 * split per field, plus the element permutation; used for bit-exact parity
 * checks of the builder. */
typedef struct crt_bvh_arrays {
    uint32_t node_count;
    uint32_t elem_count;
    uint32_t depth;
    const uint8_t* leaf;        /* node_count */
    const uint32_t* elem_start; /* node_count */
    const uint32_t* elem_count_per_node;
    const uint32_t* left;
    const uint32_t* right;
    const uint32_t* next;
    const float* box;           /* node_count * 6: pmin xyz, pmax xyz */
    const uint32_t* elements;   /* elem_count */
} crt_bvh_arrays;

int crt_bvh_build(const float* v, uint32_t nv, const uint32_t* f, uint32_t nf,
                  const float* mesh_box_min, const float* mesh_box_max, crt_bvh** out);
int crt_bvh_get_arrays(const crt_bvh* bvh, crt_bvh_arrays* out);
void crt_bvh_free(crt_bvh* bvh);

/* ---- PNG (replaces RenderedImage::SaveImage, renderer/renderer.cuh:37-40,
 *      and the lodepng decode in lib/xmlload.cuh:65-91) -------------------- */

int crt_png_write_rgb8(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height);
/* *rgb is malloc'ed; release with crt_free. */
int crt_png_read_rgb8(const char* path, uint8_t** rgb, uint32_t* width, uint32_t* height);
void crt_free(void* p);

/* ---- device context ----------------------------------------------------- */

typedef struct crt_ctx crt_ctx;

/* Number of GPUs crt_create accepts (device = 0 .. count - 1).  The reference renders on whatever device the
 * CUDA runtime selects (renderer/renderer.cu:21-24, no cudaSetDevice); the sequence front end uses this to
 * refuse a --devices count the machine does not have. */
int crt_device_count(int* count);

int crt_create(int device, crt_ctx** out);
void crt_destroy(crt_ctx* ctx);

/* Opt-in acceleration that does NOT reproduce the reference's images bit for bit: from the next crt_upload_scene
 * on, ray queries walk a BVH over the world boxes of the scene's object nodes instead of scanning the node list
 * from end to end (renderer/trace.cu:51 closest hit, :166 shadow rays).  Objects are then met in the order of that
 * walk rather than in node order: among hits at exactly the same distance another one may win, and a ray goes
 * through the ToLocal / FromLocal round trips of other nodes, so last bits differ.  For scenes of many nodes
 * (instances); off by default.  The environment variable CRT_TLAS=0/1, read at upload, overrides the call. */
int crt_set_tlas(crt_ctx* ctx, int enable);
const char* crt_last_error(const crt_ctx* ctx);

/* Copy the scene to the device: builds one BVH per mesh on the host, re-lays
 * nodes / triangles out for the GPU and allocates the three W*H float3
 * accumulators (results, normals, albedos: renderer/renderer.cu:21-24), which
 * start zeroed.  The caller keeps ownership of everything in *scene. */
int crt_upload_scene(crt_ctx* ctx, const crt_scene_desc* scene);

/* Restrict this context to the image rows it owns: rows are cut into strips of
 * strip_rows rows and strip s belongs to part (s % parts).  parts = 1 selects
 * the whole image.  Pixels outside the part are never touched. */
int crt_set_partition(crt_ctx* ctx, uint32_t part, uint32_t parts, uint32_t strip_rows);

/* Zero the accumulators and the work counters (blocking). */
int crt_clear(crt_ctx* ctx);

/* Same, enqueued on `stream` (a hipStream_t) without waiting. */
int crt_clear_async(crt_ctx* ctx, void* stream);

/* One pass of the trace kernel: TracePrimaryRays<<<...>>>(pass_id)
 * (renderer/trace.cu:7-41, launch at renderer/renderer.cu:44) with `samples`
 * in the role of the SAMPLES constant (renderer/settings.cuh:12).  Adds into
 * the accumulators.  `stream` is a hipStream_t (NULL = default stream); the
 * call returns after the kernel has been enqueued when async != 0. */
int crt_render_pass(crt_ctx* ctx, int pass_id, int samples, void* stream, int async);

/* crt_render_pass for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
 * It has no counterpart in the reference.  Same arguments, same accumulators
 * and ray counters as crt_render_pass, bit for bit; the kernel's workspace
 * (pixel-queue counter, sequence-jump matrix, per-lane path state) is a fixed
 * 32 MiB array inside the library's code object, one copy per GPU, instead of
 * buffers from hipMalloc; the jump matrices of passes 1 .. 63 are written into
 * it when the library first meets the GPU (crt_create), so the call itself
 * allocates nothing and copies nothing: it enqueues the kernel and nothing
 * else.  With async != 0 it makes no call that stream capture forbids
 * and can be recorded into a HIP graph (async == 0 waits for the stream, which
 * capture does not allow, and then reports a kernel that gave up like
 * crt_render_pass does).  pass_id >= 64 fails with CRT_E_LIMIT.  The grid
 * shrinks to the lanes whose state fits; a scene whose BVH is so deep that one
 * block's state does not fit fails with CRT_E_LIMIT.  Tiles are handed out in
 * storage order and no kernel time is recorded.  Every call on one GPU, from
 * any context, shares that GPU's array: calls that overlap on one GPU race and
 * must be ordered by the caller (replays of a recorded graph included: wait for
 * one launch of the graph before the next). */
int crt_render_pass_noalloc(crt_ctx* ctx, int pass_id, int samples, void* stream, int async);

/* ConvertColors<<<...>>>(results, out, N, sRGB, pass_mul) (math/color.cu:19-23,
 * launch at renderer/renderer.cu:51).  rgb_out_device is W*H*3 bytes of device
 * memory, or NULL to convert into the context's own buffer. */
int crt_convert(crt_ctx* ctx, float pass_mul, int srgb, void* rgb_out_device, void* stream,
                int async);

/* Denoiser hand-off.  crt_prepare_denoise replaces PrepareForDenoise
 * (math/color.cu:32-51, launched from renderer/renderer.cu:89-92): the three
 * accumulators times pass_mul (= 1 / passes) into three caller-owned device
 * buffers of width * height * 3 floats, the beauty / normal / albedo images a
 * denoiser takes.  crt_convert_floats replaces the float overload of
 * ConvertColors (math/color.cu:25-30, renderer/renderer.cu:146): a float3
 * device image, e.g. the denoiser's output, to the 8-bit image (sRGB curve
 * when srgb != 0), into rgb_out_device or the context's own image when NULL.
 * The filter between the two is the caller's (the reference links OIDN). */
int crt_prepare_denoise(crt_ctx* ctx, float pass_mul, float* beauty_device, float* normal_device, float* albedo_device,
                        void* stream, int async);
int crt_convert_floats(crt_ctx* ctx, const float* rgb_in_device, int srgb, void* rgb_out_device, void* stream, int async);

/* Copy the context's converted image to host memory (renderer/renderer.cu:57). */
int crt_read_image(crt_ctx* ctx, uint8_t* rgb_host);

/* Copy the float accumulators to host memory (W*H*3 floats each; any pointer
 * may be NULL). */
int crt_read_accumulators(crt_ctx* ctx, float* results, float* normals, float* albedos);

/* Device addresses of the accumulators and the converted image, for callers
 * that exchange them between GPUs without a host copy. */
int crt_device_buffers(crt_ctx* ctx, void** results, void** normals, void** albedos,
                       void** rgb8);

/* Parity hook: the first camera ray of pass pass_id for every owned pixel,
 * traced with TraceThroughScene (renderer/trace.cu:44-81).  Host outputs, W*H
 * entries each (any may be NULL): node = index of the hit node, -1 for a miss,
 * -2 - light index for a light; face = original face index for mesh hits else
 * -1; t = hit distance in ray-parameter units (FLT_MAX on a miss). */
int crt_primary_hits(crt_ctx* ctx, int pass_id, int32_t* node, int32_t* face, float* t);

/* Work counters of everything rendered since crt_clear / crt_upload_scene.
 * A ray is one closest-hit query or one shadow query toward a point light. */
typedef struct crt_stats {
    uint64_t closest_rays;
    uint64_t shadow_rays;
    uint64_t paths;
    /* traversal detail, filled only by crt_render_pass_counted */
    uint64_t node_box_tests;   /* scene-node slab tests */
    uint64_t node_enters;      /* object intersections after the box passed */
    uint64_t bvh_interior;     /* interior BVH nodes expanded (two child slabs each) */
    uint64_t bvh_leaves;       /* leaves visited */
    uint64_t tri_tests;        /* ray/triangle tests */
    uint64_t tri_hits;         /* accepted closest-hit triangles */
    uint64_t sphere_tests;
    uint64_t plane_tests;
    uint64_t texel_quads;      /* bilinear texture evaluations */
    /* scheduler detail of the trace kernel (counted build): how often each operation
     * (0 box, 1 triangle, 2 transform, 3 shading, 4 exchange of results for new queries,
     * 5 pixel fetch) was executed by a wave, and how many lanes took part in total; 6, 7: census at the
     * votes of the traversal waves; 8 votes that moved queries between waves (lanes: queries given up),
     * 9 queries taken over from other waves (turns: attempts, lanes: queries), 10, 11 spare */
    uint64_t sched_turns[12];
    uint64_t sched_lanes[12];
    uint64_t sched_cycles[12];  /* shader clock cycles waves spent in turns of each operation */
    uint64_t path_cycles[8];    /* cycles inside the path operation: lights, hit attributes, textures,
                                   light sampling, BSDF sampling, next ray; [6] summed wave
                                   lifetimes, [7] the part of them after the pixel queue ran dry */
    uint64_t scan_checks;       /* scene-node boxes the counted pass found hit, each checked against the */
    uint64_t scan_violations;   /* candidate mask the plain pass scans by; a violation would be a parity bug */
    uint64_t errors;            /* waves of the trace kernel that gave up waiting on their block's queues: 0 unless
                                   the kernel's hand-off protocol failed (crt_get_stats then returns an error) */
} crt_stats;

int crt_get_stats(crt_ctx* ctx, crt_stats* out);

/* Same image as crt_render_pass, additionally counting traversal work into
 * crt_stats (slower; used to price the roofline, not for timing). */
int crt_render_pass_counted(crt_ctx* ctx, int pass_id, int samples);

/* Parity hook for the seeded sequence (curand_init + RandomFloat,
 * renderer/trace.cu:22-23, renderer/rng.cuh:8-12): generate on the GPU, for
 * each of n (seed, subsequence) pairs, the six state words after init followed
 * by `draws` raw 32-bit outputs.  out is n * (6 + draws) host words. */
int crt_rng_selftest(crt_ctx* ctx, const uint32_t* seeds, const uint32_t* subsequences,
                     uint32_t n, uint32_t draws, uint32_t* out);

/* Parity hook for the pinned transcendentals: evaluate function `fn`
 * (0 sin, 1 cos, 2 exp, 3 pow(x, y), 4 atan2(x, y), 5 asin, 6 acos, 7 tan,
 * 8 the kernels' reciprocal, which stands for the division 1.0f / x)
 * on n host floats on the GPU.  fn = 100 reads no input: it runs all 2^32
 * float bit patterns through that reciprocal and through the division on
 * the device and stores in out[0] the number of patterns on which they differ. */
int crt_math_selftest(crt_ctx* ctx, int fn, const float* x, const float* y, uint32_t n,
                      float* out);

/* How the library reads a numeric environment override (CRT_SCHED_BOX_MIN,
 * CRT_TRAV_WAVES, ...): the decimal value of `text` when it is a whole number
 * inside [lo, hi], otherwise `fallback` (NULL, empty, non-numeric or out of
 * range text never reaches the kernels). */
uint32_t crt_knob_value(const char* text, uint32_t fallback, uint32_t lo, uint32_t hi);

/* Milliseconds the most recent crt_render_pass kernel took, measured with HIP
 * events on its stream (blocks until it has finished). */
int crt_last_kernel_ms(crt_ctx* ctx, float* ms);

/* Durations of the trace-kernel launches since the previous call of this
 * function (at most 256 are kept, oldest first), each measured with a HIP event
 * pair on the launch's own stream.  Blocks until those launches have finished.
 * Writes min(*count, cap) values to ms and the number of launches to *count. */
int crt_kernel_times(crt_ctx* ctx, float* ms, uint32_t cap, uint32_t* count);

#ifdef __cplusplus
}
#endif
#endif /* CRT_AMD_H */
