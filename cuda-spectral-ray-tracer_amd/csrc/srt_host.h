// srt_host.h -- host-side flattened scene (replaces the reference's device-heap scene graph,
// scene/scene.cuh:103-176).  No HIP in here: everything below runs before upload.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/srt_c_api.h"

namespace srt {

struct F3 { float x, y, z; };

// Host XORWOW (cuRAND core, restated from the published definition; used by create_bvh_kernel,
// scene/scene.cu:12-14, to pick split axes).
struct HostRng {
    uint32_t d, v[5];
    explicit HostRng(uint64_t seed);
    uint32_t next();
    float uniform();                         // (0,1]   utils/cuda_utility.cu:19-26
    float range(float mn, float mx);         // utils/cuda_utility.cu:28-41
    int rand_int(int mn, int mx);            // utils/cuda_utility.cu:44-49
};

// tri::init products (primitives/tri.cuh:88-94)
struct TriRecord {
    float n[3];
    float D;
    bool clockwise;
    uint32_t aa_plane;
    float box[6];   // xmin xmax ymin ymax zmin zmax (padded, aabb.cuh:92-102)
};

struct BvhNode {
    int32_t left = -1, right = -1;   // node indices
    int32_t prim = -1;               // triangle index for leaves
    float box[6] = {0, 0, 0, 0, 0, 0};
};

struct CameraSetup {   // what the scene's camera_builder holds (scene/scene.cu:259-320)
    float vfov = 40.f;
    float lookfrom[3] = {278.f, 278.f, -800.f}, lookat[3] = {278.f, 278.f, 0.f}, vup[3] = {0.f, 1.f, 0.f};
    float defocus_angle = 0.f, focus_dist = 10.f;
};

}  // namespace srt

struct srt_scene {
    std::vector<srt_tri_in> raw;
    std::vector<srt::TriRecord> rec;
    std::vector<srt_material> mats;
    float background[SRT_N_CIE_SAMPLES];
    std::vector<srt::BvhNode> nodes;
    int32_t root = -1;
    int depth = 0;            // max number of internal nodes on a root-to-leaf path
    bool bvh_valid = false;
    srt::CameraSetup cam;     // the scene's DEFAULT camera (srt_scene_default_camera); never modified after construction
    // viewpoint the SAH builder / srt_scene_order_children put the nearer child on the left for; unset = the default camera's
    float order_eye[3] = {0.f, 0.f, 0.f};
    bool has_order_eye = false;
    int scene_id = -1;
    std::string name;
};

namespace srt {

void set_global_error(const std::string &msg);
const char *global_error();

// tri::init (primitives/tri.cu:47-84); aa_plane_in = member value before init (Q12)
void tri_precompute(const srt_tri_in &t, TriRecord &out);
int build_bvh_reference(srt_scene &s, uint64_t seed);   // bvh/bvh.cu:206-346
int build_bvh_sah(srt_scene &s);
void scene_builtin(srt_scene &s, int scene_id, uint64_t seed);
bool scene_builtin_known(int scene_id);

// GPU images of the scene (layout: srt_device.h)
bool tree_is_paired(const srt_scene &s);      // every internal node has two leaf children or none

struct FlatScene {
    std::vector<float> nodes;    // 16 floats per INNER record (both children internal)
    std::vector<float> nodes_sw; // 20 floats per INNER record, pre-swizzled for the L2-served assembly visit (see flatten_scene)
    std::vector<float> fringe;   // 24 floats per FRINGE record (a leaf child; triangle data inline, (left, right) pairs), record index - n_inner
    std::vector<float> tris;     // 12 floats per triangle
    std::vector<float> mat_sd;   // 192 floats per material, then 192 for the background (table index n_materials)
    std::vector<float> shade;    // 12 floats per triangle: normal, material index, material scalars (shading record)
    std::vector<float> mat_par;  // 8 floats per material
    int root_ref = 0;
    int stack_depth = 1;
    int n_inner = 0;             // records [0, n_inner) have two internal children (BFS order); the rest have a leaf child
    int n_records = 0;
};
int flatten_scene(const srt_scene &s, FlatScene &out);
void cmf_rows(float *rows96x4);   // { x_bar, y_bar, z_bar, D65n } per 5 nm sample

}  // namespace srt
