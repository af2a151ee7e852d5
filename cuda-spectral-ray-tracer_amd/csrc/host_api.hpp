// host_api.hpp -- C++ host mirror of the reference's interface for the render path, header-only, on top of the
// C-ABI (include/srt_c_api.h).  Same class and method names, argument meaning and "print and ignore" error
// behaviour as the reference host classes, so that main.cpp-style drivers keep working:
//
//   camera / camera_builder      rendering/camera.cuh:6-104, rendering/camera_builder.cuh:13-71
//   frame_buffer / image_channels rendering/frame_buffer.cuh:6-71
//   scene_manager                scene/scene.cuh:103-176   (device-heap world -> flattened srt_scene)
//   renderer                     rendering/rendering.cuh:39-155
//   render_manager               rendering/render_manager.cuh:37-224, rendering/render_manager.cu
//
// Differences forced by the boundary: the first two constructor arguments of renderer / render_manager
// (bvh** dev_bvh, material* dev_mat_list) become one `const srt_scene*` (a device-heap pointer tree is not
// portable); dim3 becomes a plain struct; CUDA errors become return codes instead of exit(99).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <iostream>
#include <semaphore>
#include <string>
#include <thread>
#include <vector>

#include "../../include/srt_c_api.h"

namespace srt_host {

typedef unsigned int uint;
struct dim3 { uint x = 1, y = 1, z = 1; dim3() {} dim3(uint x_, uint y_ = 1, uint z_ = 1) : x(x_), y(y_), z(z_) {} };
struct vec3 { float e[3] = {0.f, 0.f, 0.f}; vec3() {} vec3(float a, float b, float c) : e{a, b, c} {} float operator[](int i) const { return e[i]; } };
using point3 = vec3;
using color = vec3;

// rendering/camera.cuh:6-104
class camera {
public:
    camera() {}
    camera(float ar, int w, int h, float _vfov, point3 _lookfrom, point3 _lookat, vec3 _vup, float _da, float _fd, color bg)
        : aspect_ratio(ar), image_width(w), image_height(h), background(bg) {
        srt_camera_init(w, h, _vfov, _lookfrom.e, _lookat.e, _vup.e, _da, _fd, &data);   // camera::initialize, camera.cu:7-58
        num_pixels = (uint)(w * h);
    }
    // camera whose derived quantities were already computed (e.g. by srt_scene_default_camera)
    static camera fromData(const srt_camera_data &d, float ar = 1.f) {
        camera c; c.aspect_ratio = ar; c.image_width = (int)d.width; c.image_height = (int)d.height; c.num_pixels = d.width * d.height; c.data = d;
        return c;
    }
    const int &getImageWidth() const { return image_width; }
    const int &getImageHeight() const { return image_height; }
    const uint &getNumPixels() const { return num_pixels; }
    vec3 getCenter() const { return v(data.camera_center); }
    vec3 getPixel00Loc() const { return v(data.pixel00_loc); }
    vec3 getPixelDeltaU() const { return v(data.pixel_delta_u); }
    vec3 getPixelDeltaV() const { return v(data.pixel_delta_v); }
    float getDefocusAngle() const { return data.defocus_angle; }
    vec3 getDefocusDiskU() const { return v(data.defocus_disk_u); }
    vec3 getDefocusDiskV() const { return v(data.defocus_disk_v); }
    const color &getBackground() const { return background; }
    const srt_camera_data &getCameraData() const { return data; }   // camera_data, rendering.cuh:19-37
private:
    static vec3 v(const float *p) { return vec3(p[0], p[1], p[2]); }
    float aspect_ratio = 1.f;
    int image_width = 0, image_height = 0;
    uint num_pixels = 0;
    color background;
    srt_camera_data data{};
};

// rendering/camera_builder.cuh:13-71 (image size comes from the caller instead of the param_manager singleton)
class camera_builder {
public:
    camera_builder setVfov(float v) { vfov = v; return *this; }
    camera_builder setLookfrom(point3 p) { lookfrom = p; return *this; }
    camera_builder setLookat(point3 p) { lookat = p; return *this; }
    camera_builder setVup(vec3 p) { vup = p; return *this; }
    camera_builder setDefocusAngle(float a) { defocus_angle = a; return *this; }
    camera_builder setFocusDist(float d) { focus_dist = d; return *this; }
    camera_builder setBackground(color c) { background = c; return *this; }
    camera_builder setImageSize(int xres, int yres) { width = xres; height = yres; return *this; }
    camera getCamera() const { return camera((float)width / (float)height, width, height, vfov, lookfrom, lookat, vup, defocus_angle, focus_dist, background); }
private:
    float vfov = 90.0f;
    point3 lookfrom = point3(0, 0, -1), lookat = point3(0, 0, 0);
    vec3 vup = vec3(0, 1, 0);
    float defocus_angle = 0, focus_dist = 10;
    color background = color(0, 0, 0);
    int width = 600, height = 600;   // io/params.h:204-222 defaults (xres 600, AR 1)
};

// rendering/frame_buffer.cuh:6-44: planar float RGB, row-major, values already 0..255
class frame_buffer {
public:
    explicit frame_buffer(size_t img_size) : channel_size(img_size) { r = new float[img_size](); g = new float[img_size](); b = new float[img_size](); }
    ~frame_buffer() { delete[] r; delete[] g; delete[] b; }
    frame_buffer(const frame_buffer &) = delete;
    size_t single_channel_byte_size() const { return channel_size * sizeof(float); }
    void split_channels(unsigned char *const _r, unsigned char *const _g, unsigned char *const _b) const {
        for (size_t i = 0; i < channel_size; i++) { _r[i] = (unsigned char)r[i]; _g[i] = (unsigned char)g[i]; _b[i] = (unsigned char)b[i]; }
    }
    size_t channel_size;
    float *r, *g, *b;
};

// rendering/frame_buffer.cuh:46-71
struct image_channels {
    explicit image_channels(frame_buffer &fb) : n(fb.channel_size) { r = new unsigned char[n]; g = new unsigned char[n]; b = new unsigned char[n]; fb.split_channels(r, g, b); }
    image_channels &operator=(const frame_buffer &fb) { fb.split_channels(r, g, b); return *this; }
    ~image_channels() { delete[] r; delete[] g; delete[] b; }
    size_t n;
    unsigned char *r, *g, *b;
};

// scene/scene.cuh:103-176: owns the world; getWorld()/getMaterials() collapse into getScene()
class scene_manager {
public:
    explicit scene_manager(int scene_id, int xres, int yres, int bvh_mode = SRT_BVH_REFERENCE, uint64_t seed = SRT_DEFAULT_SEED) {
        s = srt_scene_builtin(scene_id, 0);
        if (!s) { res_msg = srt_last_error(nullptr); return; }
        if (srt_scene_build_bvh(s, bvh_mode, seed) != SRT_OK) { res_msg = "Error building BVH\n"; return; }   // scene.cu:413-416
        srt_camera_data cd;
        srt_scene_default_camera(s, xres, yres, &cd);
        cam_data = cd; xr = xres; yr = yres;
        world_inited = true; res_msg = "World created";
    }
    ~scene_manager() { srt_scene_destroy(s); }
    scene_manager(const scene_manager &) = delete;
    bool isWorldInited() const { return world_inited; }
    const std::string &getResultMsg() const { return res_msg; }
    const srt_scene *getScene() const { return s; }
    const srt_camera_data &getCameraData() const { return cam_data; }
    int getXres() const { return xr; }
    int getYres() const { return yr; }
    size_t getWorldSize() const { return srt_scene_tri_count(s); }
    size_t getNumMaterials() const { return srt_scene_material_count(s); }
    // Tree tuning for a THROUGHPUT-bound render of the whole image on n_gpus devices (no reference counterpart: the tree is an
    // input of bvh::hit, bvh/bvh.cu:98-166): when a rank's launch has at least 6 pixels per persistent lane, the
    // SAH tree is post-optimised by reinsertion (up to 8 192 triangles) and its child order is measured on one instrumented
    // probe frame of the scene's camera at a quarter of the size, 8 samples per pixel (srt_order_children_by_profile, which undoes itself when
    // the probe frame did not get cheaper).  Launches with fewer pixels per lane are bound by their longest pixel and keep the
    // tree as built.  Call before the renderer is created (it uploads the scene).  Returns what was done, for the log.
    std::string tune_tree_for_throughput(uint bounce_limit, int device = 0, int n_gpus = 1) {
        if (!world_inited) return "no scene";
        srt_ctx *probe = nullptr;
        if (srt_create(device, &probe) != SRT_OK) return std::string("no tuning: ") + srt_last_error(nullptr);
        std::string what = "tree as built";
        int waves = 0;
        if (srt_upload_scene(probe, s) == SRT_OK && srt_launch_plan(probe, &waves, nullptr, nullptr, nullptr) == SRT_OK) {
            const double lanes = (double)srt_ctx_cu_count(probe) * waves * 64.0;
            const double per_lane = (double)xr * yr / (double)(n_gpus < 1 ? 1 : n_gpus) / (lanes > 0 ? lanes : 1.0);
            if (per_lane >= 6.0) {
                what = "";
                if (srt_scene_tri_count(s) <= 8192) {
                    int resident = 0, still = 0;
                    srt_launch_plan(probe, nullptr, nullptr, &resident, nullptr);
                    if (srt_scene_optimise_bvh(s, 3) == SRT_OK && srt_upload_scene(probe, s) == SRT_OK &&
                        srt_launch_plan(probe, nullptr, nullptr, &still, nullptr) == SRT_OK) {
                        if (resident && !still) {      // (a deeper tree needs deeper LDS stacks: it no longer fits LDS -- not worth it)
                            srt_scene_build_bvh(s, SRT_BVH_SAH, SRT_DEFAULT_SEED);
                            what = "reinsertion undone (the deeper tree would no longer be LDS resident); ";
                        } else what = "3 reinsertion passes; ";
                    }
                }
                // ONE probe recipe for every front end (bench.py / the Python binding's profile_child_order use the same): the scene's
                // camera at a quarter of the frame's size (at least 32 x 32), 8 samples per pixel, nodes with at least 16 deciding rays --
                // so srt_render --sah and bench.py traverse the same tree for the same workload
                const int pw = std::max(xr / 4, 32), ph = std::max(yr / 4, 32);
                srt_camera_data probe_cam{};
                uint32_t swapped = 0;
                if (srt_scene_default_camera(s, pw, ph, &probe_cam) == SRT_OK && srt_set_camera(probe, &probe_cam) == SRT_OK &&
                    srt_order_children_by_profile(probe, s, (uint32_t)pw, (uint32_t)ph, 8, bounce_limit, 16, &swapped) == SRT_OK)
                    what += (swapped ? "child order profiled (" + std::to_string(pw) + "x" + std::to_string(ph) + " x 8 spp probe): " + std::to_string(swapped) + " nodes swapped"
                                     : std::string("builder's child order kept"));
                else what += std::string("child order not profiled: ") + srt_last_error(probe);
            } else what = "tree as built (chain-bound launch: fewer than 6 pixels per lane)";
        }
        srt_destroy(probe);
        return what;
    }
private:
    srt_scene *s = nullptr;
    srt_camera_data cam_data{};
    int xr = 0, yr = 0;
    bool world_inited = false;
    std::string res_msg;
};

// rendering/rendering.cuh:39-155.  One GPU (srt_ctx) or, when several devices are given, a communicator over them
// (srt_comm: the chunk's 8x8-pixel tiles are interleaved over the GPUs, one RCCL gather brings them to the first device,
// whose context then answers getDevFB* / srt_read_fb exactly like the single-GPU renderer).
class renderer {
public:
    renderer() {}
    renderer(const srt_scene *scene, uint _samples_per_pixel, const srt_camera_data &cam, uint _bounce_limit, int device = 0)
        : samples_per_pixel(_samples_per_pixel), bounce_limit(_bounce_limit) {
        if (srt_create(device, &ctx) != SRT_OK) { note(SRT_ERR_NO_DEVICE, srt_last_error(nullptr)); return; }
        int rc = srt_upload_scene(ctx, scene);
        if (rc == SRT_OK) rc = srt_set_camera(ctx, &cam);
        if (rc != SRT_OK) note(rc, srt_last_error(ctx));
    }
    renderer(const srt_scene *scene, uint _samples_per_pixel, const srt_camera_data &cam, uint _bounce_limit, const std::vector<int> &devices)
        : samples_per_pixel(_samples_per_pixel), bounce_limit(_bounce_limit) {
        int rc = srt_comm_init_all(devices.data(), (int)devices.size(), &comm);
        if (rc != SRT_OK) { note(rc, srt_comm_last_error(nullptr)); return; }
        ctx = srt_comm_root_ctx(comm);
        rc = srt_comm_upload_scene(comm, scene);
        if (rc == SRT_OK) rc = srt_comm_set_camera(comm, &cam);
        if (rc != SRT_OK) note(rc, srt_comm_last_error(comm));
    }
    ~renderer() { release(); }
    renderer(const renderer &) = delete;
    renderer &operator=(renderer &&o) noexcept {
        release();
        ctx = o.ctx; comm = o.comm; o.ctx = nullptr; o.comm = nullptr;
        samples_per_pixel = o.samples_per_pixel; bounce_limit = o.bounce_limit;
        max_chunk_width = o.max_chunk_width; max_chunk_height = o.max_chunk_height; device_inited = o.device_inited;
        first_error.store(o.first_error.load());
        return *this;
    }
    void init_device_params(dim3 _threads, dim3 _blocks, uint _max_chunk_width, uint _max_chunk_height) {   // rendering.cu:279-357
        threads = _threads; blocks = _blocks; max_chunk_width = _max_chunk_width; max_chunk_height = _max_chunk_height;
        int rc = SRT_ERR_INVALID;
        if (comm) rc = srt_comm_init_device_params(comm, threads.x, threads.y, blocks.x, blocks.y, max_chunk_width, max_chunk_height, samples_per_pixel, bounce_limit, SRT_DEFAULT_SEED);
        else if (ctx) rc = srt_init_device_params(ctx, threads.x, threads.y, blocks.x, blocks.y, max_chunk_width, max_chunk_height, samples_per_pixel, bounce_limit, SRT_DEFAULT_SEED);
        device_inited = rc == SRT_OK;
        if (!device_inited) note(rc, comm ? srt_comm_last_error(comm) : srt_last_error(ctx));
    }
    void render(uint offset_x, uint offset_y) { call_render_kernel(max_chunk_width, max_chunk_height, offset_x, offset_y); }     // rendering.cuh:57-61
    void render(uint width, uint height, uint offset_x, uint offset_y) { call_render_kernel(width, height, offset_x, offset_y); }   // :63-66
    uint getMaxChunkWidth() const { return max_chunk_width; }
    uint getMaxChunkHeight() const { return max_chunk_height; }
    const float *getDevFBr() const { return plane(0); }   // rendering.cuh:87-97 (device pointers)
    const float *getDevFBg() const { return plane(1); }
    const float *getDevFBb() const { return plane(2); }
    srt_ctx *getContext() const { return ctx; }
    bool isDeviceInited() const { return device_inited; }
    // dynamic LDS bytes of one workgroup of the render launch: this build's counterpart of shared_mem_size (rendering.cu:290-301)
    size_t getSharedMemSize() const { size_t b = 0; if (ctx) (void)srt_launch_lds_bytes(ctx, &b); return b; }
    // first failing status of any call made through this object (0 = none): the reference exits the process from
    // checkCudaErrors (utils/cuda_utility.cu:8-18); here the caller decides
    int getError() const { return first_error.load(); }
    uint64_t lastRays() const {   // closest-hit queries of the last chunk, all GPUs
        uint64_t rays = 0;
        if (comm) { if (srt_comm_stats(comm, &rays, nullptr, nullptr) != SRT_OK) rays = 0; }
        else if (ctx) { srt_stats st; if (srt_get_stats(ctx, &st) == SRT_OK) rays = st.rays; }
        return rays;
    }
private:
    void call_render_kernel(uint width, uint height, uint offset_x, uint offset_y) {   // rendering.cu:244-277
        if (!device_inited) { std::cerr << "Device parameters were not initialized, render aborted" << std::endl; if (!first_error.load()) first_error.store(SRT_ERR_INVALID); return; }
        int rc;
        if (comm) {
            rc = srt_render_frame_multi(comm, width, height, offset_x, offset_y);
            if (rc == SRT_OK) rc = srt_comm_synchronize(comm);
            if (rc != SRT_OK) note(rc, srt_comm_last_error(comm));
        } else {
            rc = srt_render_chunk(ctx, width, height, offset_x, offset_y, nullptr);
            if (rc == SRT_OK) rc = srt_scatter_tiles(ctx, nullptr, nullptr);
            if (rc == SRT_OK) rc = srt_synchronize(ctx);   // the reference's cudaDeviceSynchronize after the launch (rendering.cu:275-276)
            if (rc != SRT_OK) note(rc, srt_last_error(ctx));
        }
    }
    void note(int rc, const char *msg) {
        std::cerr << "renderer: " << (msg ? msg : "error") << std::endl;
        int expected = 0;
        first_error.compare_exchange_strong(expected, rc);
    }
    void release() {
        if (comm) { srt_comm_destroy(comm); comm = nullptr; ctx = nullptr; }
        else if (ctx) { srt_destroy(ctx); ctx = nullptr; }
    }
    const float *plane(int k) const { void *p[3] = {nullptr, nullptr, nullptr}; size_t n = 0; if (ctx) srt_dev_fb(ctx, &p[0], &p[1], &p[2], &n); return (const float *)p[k]; }
    srt_ctx *ctx = nullptr;      // single GPU: owned; communicator: rank 0's context (owned by the communicator)
    srt_comm *comm = nullptr;
    uint samples_per_pixel = 0, bounce_limit = 0;
    uint max_chunk_width = 0, max_chunk_height = 0;
    dim3 blocks, threads;
    bool device_inited = false;
    std::atomic<int> first_error{0};
};

// rendering/render_manager.cuh:9-35: one of the two hand-off slots between the render thread and the caller
struct render_step_data {
    render_step_data() : empty(1), full(0) {}
    void alloc_buffer(size_t buffer_size) { fb_r.assign(buffer_size, 0.f); fb_g.assign(buffer_size, 0.f); fb_b.assign(buffer_size, 0.f); }
    std::vector<float> fb_r, fb_g, fb_b;      // block-linear, grid sized (what the three D2H copies of step() deliver)
    uint starting_offset_x = 0, starting_offset_y = 0, chunk_width = 0, chunk_height = 0;
    std::counting_semaphore<1> empty, full;   // utils/multithread.cuh:3
    bool is_last = false;
};

// rendering/render_manager.cuh:37-224 + render_manager.cu.  Same structure as the reference: render_cycle() starts a
// render thread that produces chunks into two slots (step), the caller consumes them with update_fb(), which un-swizzles
// the block-linear planes into the row-major frame_buffer; `empty` / `full` semaphores order the two sides.
class render_manager {
public:
    render_manager(const srt_scene *_scene, camera *_cam, frame_buffer *_fb) {
        if (_scene != nullptr && _cam != nullptr && _fb != nullptr) {
            scene = _scene; cam = _cam; fb = _fb;
            image_width = (uint)cam->getImageWidth(); image_height = (uint)cam->getImageHeight();
            scene_inited = true;
        }
    }
    ~render_manager() { end_render(); }

    void init_renderer(uint bounce_limit, uint samples_per_pixel, int device = 0) {   // render_manager.cu:121-133
        if (scene_inited) { r = renderer(scene, samples_per_pixel, cam->getCameraData(), bounce_limit, device); renderer_inited = true; }
        else std::cerr << "Scene not yet initialized" << std::endl;
    }
    // the same over several GPUs of one node (tile-interleaved chunks, one RCCL gather per chunk: srt_render_frame_multi)
    void init_renderer(uint bounce_limit, uint samples_per_pixel, const std::vector<int> &devices) {
        if (!scene_inited) { std::cerr << "Scene not yet initialized" << std::endl; return; }
        if (devices.size() <= 1) { init_renderer(bounce_limit, samples_per_pixel, devices.empty() ? 0 : devices[0]); return; }
        r = renderer(scene, samples_per_pixel, cam->getCameraData(), bounce_limit, devices);
        renderer_inited = true;
    }
    void init_device_params(dim3 _threads, dim3 _blocks, uint _chunk_width, uint _chunk_height) {   // render_manager.cu:68-89
        if (!renderer_inited) { std::cerr << "Initialize renderer before assigning device parameters" << std::endl; return; }
        threads = _threads; blocks = _blocks; chunk_width = _chunk_width; chunk_height = _chunk_height;
        x_chunks = (uint)std::ceil(float(image_width) / float(chunk_width));
        uint y_chunks = (uint)std::ceil(float(image_height) / float(chunk_height));
        n_iterations = x_chunks * y_chunks;
        r.init_device_params(threads, blocks, chunk_width, chunk_height);
        for (auto &rd : render_data_container) rd.alloc_buffer((size_t)threads.x * blocks.x * threads.y * blocks.y);   // render_manager.cu:82-83
        device_inited = r.isDeviceInited();   // reference sets true unconditionally; here a missing GPU must not look ready
    }
    void init_device_params(uint _chunk_width, uint _chunk_height) {   // render_manager.cu:91-102
        if (!renderer_inited) { std::cerr << "Init renderer before assigning device parameters" << std::endl; return; }
        const uint tx = 28, ty = 16;
        init_device_params(dim3(tx, ty), dim3(_chunk_width / tx + 1, _chunk_height / ty + 1), _chunk_width, _chunk_height);
    }
    void init_device_params() { init_device_params(image_width, image_height); }   // render_manager.cu:104-119

    bool step() {   // render_manager.cu:3-66
        const uint endpoint_x = chunk_width + offset_x, endpoint_y = chunk_height + offset_y;
        last_chunk_width = endpoint_x > image_width ? chunk_width - (endpoint_x - image_width) : chunk_width;
        last_chunk_height = endpoint_y > image_height ? chunk_height - (endpoint_y - image_height) : chunk_height;
        render_step_data *rd = &render_data_container[next_write_render_data_index];
        next_write_render_data_index = (next_write_render_data_index + 1) % 2;
        rd->empty.acquire();
        rd->chunk_width = last_chunk_width; rd->chunk_height = last_chunk_height;
        rd->starting_offset_x = offset_x; rd->starting_offset_y = offset_y;
        r.render(last_chunk_width, last_chunk_height, offset_x, offset_y);
        int rc = r.getError();
        if (rc == SRT_OK) rc = srt_read_fb(r.getContext(), rd->fb_r.data(), rd->fb_g.data(), rd->fb_b.data());   // the three D2H copies, render_manager.cu:41-45
        if (rc == SRT_OK) total_rays += r.lastRays();
        bool last_step = false;
        i++;
        if (rc != SRT_OK) {      // a failed chunk ends the render: the consumer sees is_last, the caller reads getError()
            int expected = 0;
            first_error.compare_exchange_strong(expected, rc);
            rd->is_last = true; last_step = true;
        }
        if (i == n_iterations) { rd->is_last = true; last_step = true; }
        rd->full.release();
        offset_x = (i % x_chunks) * chunk_width;
        offset_y = (i / x_chunks) * chunk_height;
        return !last_step;
    }
    bool update_fb() {   // render_manager.cuh:68-142
        render_step_data *rd = &render_data_container[next_read_render_data_index];
        next_read_render_data_index = (next_read_render_data_index + 1) % 2;
        rd->full.acquire();
        // un-swizzle: lane idx = thread_y * threads.x + thread_x + block_size * (block_y * blocks.x + block_x) holds chunk pixel
        // (threads.x * block_x + thread_x, threads.y * block_y + thread_y); pixels outside the chunk were never written
        const uint block_size = threads.x * threads.y;
        for (uint y = 0; y < rd->chunk_height && y < threads.y * blocks.y; y++) {
            const uint block_y = y / threads.y, thread_y = y % threads.y;
            float *dr = fb->r + (size_t)(y + rd->starting_offset_y) * image_width + rd->starting_offset_x;
            float *dg = fb->g + (size_t)(y + rd->starting_offset_y) * image_width + rd->starting_offset_x;
            float *db = fb->b + (size_t)(y + rd->starting_offset_y) * image_width + rd->starting_offset_x;
            for (uint x = 0; x < rd->chunk_width && x < threads.x * blocks.x; x++) {
                const uint block_x = x / threads.x, thread_x = x % threads.x;
                const size_t idx = (size_t)thread_y * threads.x + thread_x + (size_t)block_size * ((size_t)block_y * blocks.x + block_x);
                dr[x] = rd->fb_r[idx]; dg[x] = rd->fb_g[idx]; db[x] = rd->fb_b[idx];
            }
        }
        const bool last_read = rd->is_last;
        rd->empty.release();
        return !last_read;
    }
    bool isDone() const { return done.load(); }
    int getError() const { return first_error.load() ? first_error.load() : r.getError(); }   // 0 = every chunk rendered
    size_t getLdsBytes() const { return r.getSharedMemSize(); }
    uint64_t getTotalRays() const { return total_rays; }   // closest-hit queries of all chunks rendered so far
    uint getImWidth() const { return image_width; }
    uint getImHeight() const { return image_height; }
    bool isReadyToRender() const { return device_inited && i < n_iterations; }
    void render_cycle() { end_render(); done = false; render_worker = std::thread(&render_manager::render_loop, this); worker_started = true; }   // :160-166
    void end_render() { if (worker_started) { render_worker.join(); worker_started = false; } }                                                // :168-173
private:
    void render_loop() { while (step()) {} done = true; }
    const srt_scene *scene = nullptr;
    camera *cam = nullptr;
    frame_buffer *fb = nullptr;
    uint image_width = 0, image_height = 0;
    bool scene_inited = false, renderer_inited = false, device_inited = false, worker_started = false;
    std::atomic<bool> done{true};
    std::atomic<int> first_error{0};
    renderer r;
    render_step_data render_data_container[2];
    size_t next_write_render_data_index = 0, next_read_render_data_index = 0;
    uint i = 0, chunk_width = 0, chunk_height = 0, n_iterations = 0, x_chunks = 1;
    uint offset_x = 0, offset_y = 0, last_chunk_width = 0, last_chunk_height = 0;
    uint64_t total_rays = 0;
    std::thread render_worker;
    dim3 threads, blocks;
};

}  // namespace srt_host
