/*
 * pbrt_hip.h — C ABI of the MI355X (gfx950) ray/path-tracing hot path.
 *
 * The reference (lazytiger/pbrt-rs, /root/reference) has no FFI for this path; its boundary
 * is a set of Rust trait objects. Each entry point below states the trait
 * method(s) it stands in for (file:line relative to /root/reference). A Rust caller wraps
 * them as `impl Integrator for HipIntegrator` / `impl Primitive for HipAggregate`
 * (INTEGRATION.md shows the binding).
 *
 * Conventions: plain C, int status returns (0 = PBRT_HIP_OK), no exceptions cross the
 * boundary; host buffers are borrowed for the duration of the call; the library owns all
 * device memory behind the opaque handles; handles may be shared between host threads (the
 * reference's Primitive is Sync + Send, src/core/primitive.rs:179): calls that reach the device
 * through one context are serialised inside the library; one device per context. Float = f32 everywhere (src/core/pbrt.rs:16).
 */
#ifndef PBRT_HIP_H
#define PBRT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PbrtHipContext PbrtHipContext;
typedef struct PbrtHipScene PbrtHipScene;

enum PbrtHipStatus {
    PBRT_HIP_OK = 0,
    PBRT_HIP_ERR_INVALID = 1,   /* bad argument (null pointer, negative count, index out of range) */
    PBRT_HIP_ERR_DEVICE = 2,    /* a HIP runtime call failed; see pbrt_hip_last_error */
    PBRT_HIP_ERR_NO_DEVICE = 3, /* no gfx950 device visible */
    PBRT_HIP_ERR_OOM = 4        /* a device allocation, or a host allocation inside the library (a tree being built, a scene re-laid
                                 * out, the tiles of a frame being dealt), failed. No C++ exception ever crosses this boundary: every
                                 * status-returning entry point catches what its body throws (csrc/abi_guard.h) */
};

/* src/accelerators/bvh.rs:129-135 LinearBVHNode, packed to pbrt-v3's 32 bytes (the reference's
 * 48 B is usize padding). Interior: first child = self + 1, `offset` = second child.
 * Leaf (n_primitives > 0): `offset` = first slot in the leaf-ordered primitive list. */
typedef struct PbrtLinearBVHNode {
    float bounds_min[3];
    float bounds_max[3];
    int32_t offset;
    uint16_t n_primitives;
    uint8_t axis;
    uint8_t pad;
} PbrtLinearBVHNode;

/* src/core/geometry.rs:757-763 Ray {o, d, t_max, time} (medium dropped: handle_media is false
 * on every call of this path, path.rs:117, directlighting.rs:114). */
typedef struct PbrtRay {
    float o[3];
    float d[3];
    float t_max;
    float time;
} PbrtRay;

/* What Primitive::intersect leaves behind that the shade stage needs (SurfaceInteraction,
 * src/core/interaction.rs:224-245, is rebuilt from it): the shrunk ray.t_max
 * (src/core/primitive.rs:70), the watertight barycentrics (src/shapes/triangle.rs:134-137)
 * and the primitive. prim_id = -1 and t = +inf on a miss. prim_id indexes the caller's
 * triangle order (not the leaf order). */
typedef struct PbrtHit {
    float t;
    float b0, b1, b2;
    int32_t prim_id;
    int32_t instance_id; /* TransformedPrimitive that was entered (caller's order), -1 = none */
    int32_t pad[2];
} PbrtHit;

/* The material files under src/materials/ are empty stubs; the three materials are pbrt-v3's on the reference's
 * BxDFs (src/core/reflection.rs:821-855, 614-659, 733-819). */
enum PbrtMaterialType { PBRT_MAT_NONE = 0, PBRT_MAT_MATTE = 1, PBRT_MAT_MIRROR = 2, PBRT_MAT_GLASS = 3 };
typedef struct PbrtMaterial {
    int32_t type;
    float kd[3]; /* matte Kd; mirror / glass Kr */
    float kt[3]; /* glass Kt */
    float eta;   /* glass index of refraction */
} PbrtMaterial;

/* src/lights/diffuse.rs:19-27 DiffuseAreaLight on one triangle; src/lights/infinite.rs:23-31
 * InfiniteAreaLight with a constant (1x1) map; the delta lights src/lights/point.rs:17-40,
 * src/lights/spot.rs:18-64 and src/lights/distant.rs:18-47. The caller passes what the reference's
 * constructors compute: p_light = light_to_world * (0,0,0), cos_total_width / cos_falloff_start =
 * cos(radians(..)), the upper 3x3 of world_to_light (spot.rs:50-51), w_light = normalize(light_to_world * w). */
enum PbrtLightType {
    PBRT_LIGHT_DIFFUSE_AREA = 0,
    PBRT_LIGHT_INFINITE = 1,
    PBRT_LIGHT_POINT = 2,
    PBRT_LIGHT_SPOT = 3,
    PBRT_LIGHT_DISTANT = 4
};
typedef struct PbrtLight {
    int32_t type;
    float L[3];        /* area / infinite / distant: radiance L; point / spot: intensity I */
    int32_t prim;      /* area light: triangle index (caller's order) */
    int32_t two_sided; /* src/lights/diffuse.rs:25 */
    int32_t n_samples; /* src/core/light.rs:76 (delta lights: 1) */
    int32_t pad;
    float pos[3];      /* point / spot: p_light (world); distant: w_light (world, unit length) */
    float cos_total_width, cos_falloff_start; /* spot */
    float world_to_light[9];                  /* spot: rows of the upper 3x3, row-major */
    int32_t pad2[2];
} PbrtLight;

/* src/core/primitive.rs:105-123 TransformedPrimitive with a static transform: one instance of the
 * object-space aggregate. Row-major 4x4, last row (0,0,0,1); to_object = inverse of to_world
 * (the reference keeps both, src/core/transform.rs:195-198). material >= 0 overrides the
 * triangles' own materials for this instance. Instanced primitives cannot be area lights. */
typedef struct PbrtInstance {
    float to_world[16];
    float to_object[16];
    int32_t material;
    int32_t pad[3];
} PbrtInstance;

/* src/cameras/perspective.rs:19-32: the two matrices the ray generator applies. Row-major. kind selects
 * PerspectiveCamera::generate_ray (perspective.rs:90-112), OrthographicCamera::generate_ray
 * (src/cameras/orthographic.rs:82-104; raster_to_camera of the orthographic projection) or
 * EnvironmentCamera::generate_ray (src/cameras/environment.rs:37-56; only camera_to_world is used). */
enum PbrtCameraKind { PBRT_CAMERA_PERSPECTIVE = 0, PBRT_CAMERA_ORTHOGRAPHIC = 1, PBRT_CAMERA_ENVIRONMENT = 2 };
typedef struct PbrtCamera {
    float camera_to_world[16];
    float raster_to_camera[16];
    float lens_radius;
    float focal_distance;
    float shutter_open;
    float shutter_close;
    int32_t kind; /* PbrtCameraKind */
    int32_t pad[3];
} PbrtCamera;

enum PbrtIntegratorKind {
    PBRT_INTEGRATOR_PATH = 0,    /* src/integrators/path.rs */
    PBRT_INTEGRATOR_DIRECT = 1,  /* src/integrators/directlighting.rs */
    PBRT_INTEGRATOR_WHITTED = 2, /* src/integrators/whitted.rs:47-98 */
    PBRT_INTEGRATOR_AO = 3       /* src/integrators/ao.rs:55-104 (D51: the unoccluded directions contribute) */
};

/* Constructor arguments of PathIntegrator (src/integrators/path.rs:31-46) /
 * DirectLightingIntegrator (src/integrators/directlighting.rs:33-46) plus the sampler and film
 * plumbing of SamplerIntegrator::render (src/core/integrator.rs:399-480). */
enum PbrtTileOrder { PBRT_TILE_ORDER_MORTON = 0, PBRT_TILE_ORDER_ROW_MAJOR = 1 };
enum PbrtSamplerKind { PBRT_SAMPLER_RANDOM = 0, PBRT_SAMPLER_STRATIFIED = 1, PBRT_SAMPLER_ZEROTWO = 2, PBRT_SAMPLER_HALTON = 3 };
typedef struct PbrtRenderParams {
    int32_t integrator;     /* PbrtIntegratorKind */
    int32_t max_depth;
    float rr_threshold;     /* path only */
    int32_t light_strategy; /* path: 0 "uniform", 1 "power", 2 "spatial" (src/core/lightdistrib.rs:222-232);
                             * direct: 0 UniformSampleAll, 1 UniformSampleOne;
                             * ao: cos_sample (0 uniform hemisphere, 1 cosine-weighted) */
    int32_t spp;            /* RandomSampler samples per pixel */
    int32_t width, height;  /* film full_resolution */
    int32_t x0, y0, x1, y1; /* pixel_bounds [x0,x1) x [y0,y1) */
    uint64_t seed;          /* stream of (pixel, sample) = seed ^ (pixel_number * spp + s), pixel_number = the pixel's
                             * row-major number over the film's sample bounds: y * width + x with the 0.5 box filter */
    int32_t tile_rank;      /* this GPU renders the 16x16 tiles whose position in the dealing order (tile_order below)
                             * is tile_rank modulo tile_world */
    int32_t tile_world;     /* 1 = all tiles */
    int32_t spp_per_pass;   /* 0 = library default; samples of one pixel traced concurrently. Queue entries are 32 bits
                             * (path << 2 | ray slot): pixels x spp_per_pass of one pass must stay below 2^30, larger values
                             * are refused with PBRT_HIP_ERR_INVALID (the default never gets there: it is sized from free HBM) */
    int32_t ao_samples;     /* AOIntegrator::n_samples (ao.rs:21); other integrators ignore it */
    /* Reconstruction filter (src/core/filter.rs:10-15, src/filters/): radius in pixels and Film's 16x16
     * table of filter.evaluate over the positive quadrant (src/core/film.rs:52-63; build it with
     * pbrt_hip_filter_table). filter_table == NULL or radius 0 = the 0.5 box filter. With a wider filter
     * the pixel bounds may reach outside the film by the sample bounds (pbrt_hip_sample_bounds). */
    float filter_radius[2];
    const float* filter_table;
    /* Sampler (src/core/sampler.rs): PBRT_SAMPLER_RANDOM (src/samplers/random.rs), PBRT_SAMPLER_STRATIFIED
     * (src/samplers/stratified.rs:22-40: sampler_x * sampler_y samples per pixel replace `spp`, sampler_jitter,
     * sampler_dims = n_sampled_dimensions) or PBRT_SAMPLER_ZEROTWO (src/samplers/zerotwosequence.rs:17-23:
     * `spp` is rounded up to a power of two, sampler_dims) or PBRT_SAMPLER_HALTON (src/samplers/halton.rs:63-98 over
     * the film's sample bounds, sample_at_pixel_center = false; a render whose sample arrays reach past the 1000 tabulated
     * dimensions, where the reference panics on PRIME_SUMS (halton.rs:100-108), is refused with PBRT_HIP_INVALID).
     * All zero = the random sampler. */
    int32_t sampler;
    int32_t sampler_x, sampler_y;
    int32_t sampler_jitter;
    int32_t sampler_dims;
    float max_sample_luminance; /* Film::max_sample_luminance (src/core/film.rs:24, 253-255); 0 = infinity (no clamp) */
    /* Order in which the path integrator's shading kernel takes the hits of a wavefront — where the reference dispatches per
     * hit to the material (src/core/interaction.rs:318-329 -> src/core/material.rs:16-55): 0 = queue order, 1 = by material
     * inside every block of 256 queue entries (LDS counting sort), 2 = the whole queue sorted by material. The film does not
     * depend on it. Other integrators ignore it. */
    int32_t shade_order;
    int32_t ray_order; /* 0: the path integrator's ray queues are put into Morton order of the ray origins from the second bounce on
                        * (cache order: the rays of a wave walk the same part of the tree); 1: queue order. Same film either way. */
    int32_t tile_order; /* PbrtTileOrder: the order in which the 16x16 tiles (src/core/integrator.rs:404-409) are dealt to the
                         * tile_world ranks. 0 = Morton order of the tile grid (every rank's tiles are spread over the
                         * whole frame in both directions), 1 = row-major (with 120 or 240 tiles per row and 2 / 4 / 8 ranks a rank
                         * then owns whole tile COLUMNS). A rank renders its own tiles in row-major order either way (with one rank
                         * the two settings are the same job). The merged film does not depend on it. */
    int32_t samples_per_wave; /* How the paths of a pass are laid out over the 64-lane waves: a wave holds samples_per_wave consecutive
                         * samples of each of 64 / samples_per_wave neighbouring pixels. 0 = the library's choice: the samples of ONE
                         * pixel per wave (64, or the largest power of two that divides the pass's samples per pixel; the stratified
                         * and (0,2) samplers, whose tables are read per pixel, keep 1); 1 = 64 pixels of one sample index (rounds
                         * 1-4); a power of two up to 64. The camera rays of a wave — and the first bounce, which keeps their order
                         * — then start from the same pixel: +2.5 % of a config-3 frame, +4 % on config 5
                         * (profiles/r05_path_layout.txt). Streams are keyed by (pixel, sample): the film is the same bits. */
} PbrtRenderParams;

typedef struct PbrtRenderStats {
    uint64_t camera_samples;
    uint64_t rays_closest;   /* Scene::intersect calls (camera, bounce and MIS rays) */
    uint64_t rays_shadow;    /* Scene::intersect_p calls */
    uint64_t trace_launches; /* launches of the traversal kernel */
    double trace_ms;         /* HIP-event time spent in the traversal kernel */
    double total_ms;         /* HIP-event time of the whole render (scene resident, film on device) */
} PbrtRenderStats;

/* ---- context ----
 * Order of destruction: scenes and communicators (pbrt_hip_comm_destroy takes the context's lock and device) before
 * their context. A call that gives up on a kernel (the 120 s wavefront deadline of pbrt_hip_render / pbrt_hip_li)
 * leaves the context LOST: every later call on it fails with PBRT_HIP_ERR_DEVICE, its device buffers are never
 * reused, and pbrt_hip_context_destroy then releases the host side only. What a lost context held on the device — its
 * block cache (tens of GB after a large frame), the buffers of the abandoned call, its scenes (pbrt_hip_scene_destroy and
 * pbrt_hip_comm_destroy after the loss release their host side only; the order above still holds: before the context) — goes back
 * with the PROCESS, not before: a long-lived host should ask pbrt_hip_context_is_lost after a PBRT_HIP_ERR_DEVICE and,
 * if so, finish in a fresh child process or exit non-zero. The caller's own device buffers of the abandoned *_device
 * call (d_film, d_rgb, d_rays, d_stream_keys) may still be written or read by the kernel that never finished: they must
 * not be freed (hipFree waits for the device) or reused either. */
int pbrt_hip_context_create(int device_id, PbrtHipContext** out);
void pbrt_hip_context_destroy(PbrtHipContext* ctx);
/* 1 when a call on this context ran into the wavefront deadline (see above), else 0; -1 for NULL. Takes no lock (the flag is
 * atomic): safe to poll from any thread while another one is inside a call on the context. */
int pbrt_hip_context_is_lost(const PbrtHipContext* ctx);
/* How long pbrt_hip_render / pbrt_hip_li wait for one wavefront before they give the context up for lost (default 120 s; a
 * profiler or a debug build may need more, a service less). seconds > 0. */
int pbrt_hip_context_set_deadline(PbrtHipContext* ctx, double seconds);
/* Last error text for this context; ctx == NULL: why the CALLING THREAD's last pbrt_hip_context_create failed (contexts may
 * be created from several host threads at once; the text is thread-local and lives until that thread's next failed creation). */
const char* pbrt_hip_last_error(const PbrtHipContext* ctx);

/* ---- host side: BVHAccel::new (src/accelerators/bvh.rs:216-271) ----
 * Builds the flat DFS node array and the leaf order over world-space triangles
 * (split_method: 0 SAH, 1 HLBVH, 2 Middle, 3 EqualCounts: the order of bvh.rs:200-205).
 * Outputs are allocated by the library; release with pbrt_hip_free. */
int pbrt_hip_bvh_build(const float* positions, int32_t n_verts, const int32_t* indices, int32_t n_tris,
                       int32_t max_prims_in_node, int32_t split_method, PbrtLinearBVHNode** nodes_out,
                       int32_t* n_nodes_out, int32_t** prim_order_out);
/* Same builder over caller-supplied primitive bounds (n x 3 floats min, n x 3 floats max), e.g. the
 * world bounds of instances for the top-level aggregate. */
int pbrt_hip_bvh_build_boxes(const float* bounds_min, const float* bounds_max, int32_t n, int32_t max_prims_in_node,
                             int32_t split_method, PbrtLinearBVHNode** nodes_out, int32_t* n_nodes_out,
                             int32_t** prim_order_out);
/* BVHAccel::new with SplitMethod::HLBVH built on the GPU (src/accelerators/bvh.rs:475-568: Morton codes
 * :137-156 — the loop the reference marks "TODO parallel" at :489 — radix sort :158-197, treelets :509-528,
 * emit_lbvh :570-676, build_upper_sah :678-772, flatten :774-811). Same outputs, byte for byte, as
 * pbrt_hip_bvh_build(..., split_method = 1). build_ms (may be NULL): HIP-event time of the build with
 * the mesh resident on the device (upload of the mesh and download of the tree excluded). */
int pbrt_hip_bvh_build_hlbvh_device(PbrtHipContext* ctx, const float* positions, int32_t n_verts,
                                    const int32_t* indices, int32_t n_tris, int32_t max_prims_in_node,
                                    PbrtLinearBVHNode** nodes_out, int32_t* n_nodes_out, int32_t** prim_order_out,
                                    double* build_ms);
/* Scene::new (src/core/scene.rs:18-34) over BVHAccel::new(HLBVH) with the tree built AND re-laid out for
 * the traversal kernels on the device: the mesh is uploaded once, nothing of the tree crosses PCIe. The
 * scene is identical to pbrt_hip_scene_create over pbrt_hip_bvh_build(..., split_method = 1).
 * build_ms / layout_ms (may be NULL): HIP-event times of the tree build and of the re-layout. */
int pbrt_hip_scene_create_hlbvh(PbrtHipContext* ctx, const float* positions, int32_t n_verts, const int32_t* indices,
                                int32_t n_tris, const int32_t* tri_material, const PbrtMaterial* materials,
                                int32_t n_materials, const int32_t* tri_light, const PbrtLight* lights, int32_t n_lights,
                                int32_t max_prims_in_node, PbrtHipScene** out, double* build_ms, double* layout_ms);
/* TransformedPrimitive::world_bound (src/core/primitive.rs:126-134, src/core/transform.rs:568-607):
 * the 8 transformed corners of the object aggregate's bounds, per instance. Host only. */
int pbrt_hip_instance_bounds(const float object_min[3], const float object_max[3], const PbrtInstance* instances,
                             int32_t n_instances, float* bounds_min, float* bounds_max);
void pbrt_hip_free(void* p);

/* ---- scene: Scene::new (src/core/scene.rs:18-34) over GeometricPrimitive triangles
 * (src/core/primitive.rs:33-55) inside a BVHAccel ----
 * nodes / prim_order come from the host BVH builder in the reference's DFS order;
 * prim_order[slot] = caller's triangle index held at leaf slot `slot`.
 * tri_material[i] indexes materials; tri_light[i] indexes lights or is -1. */
int pbrt_hip_scene_create(PbrtHipContext* ctx, const float* positions, int32_t n_verts, const int32_t* indices,
                          int32_t n_tris, const int32_t* tri_material, const PbrtMaterial* materials,
                          int32_t n_materials, const int32_t* tri_light, const PbrtLight* lights, int32_t n_lights,
                          const PbrtLinearBVHNode* nodes, int32_t n_nodes, const int32_t* prim_order,
                          PbrtHipScene** out);
/* Optional per-vertex data of the TriangleMesh (src/shapes/triangle.rs:17-26): shading normals `n`
 * (n_verts x 3, used at triangle.rs:252-312 and :337-341), tangents `s` (n_verts x 3, :265-275) and texture
 * coordinates `uv` (n_verts x 2, used for dpdu / dpdv at :60-72, 197-216). positions / indices are the arrays
 * the scene was created from. Call once, after scene creation; any of the three pointers may be NULL. */
int pbrt_hip_scene_set_shading_data(PbrtHipScene* scene, const float* positions, int32_t n_verts, const int32_t* indices,
                                    int32_t n_tris, const float* normals, const float* tangents, const float* uvs);
/* Scene with spheres next to the triangles (src/shapes/sphere.rs:38-92, 228-284 with src/core/efloat.rs; BASELINE
 * config 1): n_spheres full spheres {centre.xyz, radius}, i.e. Sphere::new with object_to_world = translate(centre).
 * Sphere i is primitive n_tris + i in prim_order; nodes come from pbrt_hip_bvh_build_boxes over the primitives' world
 * bounds (triangles first). sphere_light[i] = index of the sphere's DiffuseAreaLight (whose `prim` is n_tris + i) or -1
 * (Sphere::sample2 / pdf2, sphere.rs:123-192). Hit records of spheres carry the refined object-space hit point in
 * (b0, b1, b2). */
int pbrt_hip_scene_create_with_spheres(PbrtHipContext* ctx, const float* positions, int32_t n_verts, const int32_t* indices,
                                       int32_t n_tris, const int32_t* tri_material, const PbrtMaterial* materials,
                                       int32_t n_materials, const int32_t* tri_light, const PbrtLight* lights,
                                       int32_t n_lights, const float* spheres, const int32_t* sphere_material,
                                       const int32_t* sphere_light, int32_t n_spheres, const PbrtLinearBVHNode* nodes, int32_t n_nodes,
                                       const int32_t* prim_order, PbrtHipScene** out);
/* Two-level scene of BASELINE config 5's shape: `n_instances` TransformedPrimitives of ONE object-space triangle
 * aggregate (pbrt_hip_scene_create_two_level with one object and no world-space triangles). blas_* = BVHAccel over the
 * triangles (object space); tlas_* = BVHAccel over the instances' world bounds, tlas_order[slot] = instance index.
 * No area lights (there is no world-space triangle to carry one). */
int pbrt_hip_scene_create_instanced(PbrtHipContext* ctx, const float* positions, int32_t n_verts,
                                    const int32_t* indices, int32_t n_tris, const int32_t* tri_material,
                                    const PbrtMaterial* materials, int32_t n_materials, const PbrtLight* lights,
                                    int32_t n_lights, const PbrtLinearBVHNode* blas_nodes, int32_t n_blas_nodes,
                                    const int32_t* blas_order, const PbrtInstance* instances, int32_t n_instances,
                                    const PbrtLinearBVHNode* tlas_nodes, int32_t n_tlas_nodes,
                                    const int32_t* tlas_order, PbrtHipScene** out);
/* The general two-level scene (src/core/primitive.rs:105-159, 33-103): a top-level BVHAccel whose primitives are
 *   0 .. n_instances - 1                 TransformedPrimitives, instance i of object aggregate instance_object[i]
 *                                        (NULL = all of object 0), each object a triangle mesh in its own BVHAccel;
 *   n_instances .. + n_world_tris - 1    plain GeometricPrimitive triangles in world space beside them; only these can
 *                                        be area lights (PbrtLight.prim then counts the world triangles; pbrt-v3:
 *                                        "area lights not supported with object instancing").
 * tlas_nodes / tlas_order: BVHAccel over the world bounds of those primitives in that order (pbrt_hip_bvh_build_boxes
 * over pbrt_hip_instance_bounds of each instance's object bounds, then the triangles' bounds). Hit records name a
 * triangle by its index inside its own object (instance_id >= 0) or among the world triangles (instance_id = -1). */
typedef struct PbrtObject {
    const float* positions;
    int32_t n_verts;
    const int32_t* indices;
    int32_t n_tris;
    const int32_t* tri_material; /* may be NULL (material 0) */
    const PbrtLinearBVHNode* nodes; /* BVHAccel over the object-space triangles */
    int32_t n_nodes;
    const int32_t* prim_order;
} PbrtObject;
int pbrt_hip_scene_create_two_level(PbrtHipContext* ctx, const PbrtObject* objects, int32_t n_objects,
                                    const PbrtInstance* instances, const int32_t* instance_object, int32_t n_instances,
                                    const float* world_positions, int32_t n_world_verts, const int32_t* world_indices,
                                    int32_t n_world_tris, const int32_t* world_tri_material, const int32_t* world_tri_light,
                                    const PbrtMaterial* materials, int32_t n_materials, const PbrtLight* lights, int32_t n_lights,
                                    const PbrtLinearBVHNode* tlas_nodes, int32_t n_tlas_nodes, const int32_t* tlas_order,
                                    PbrtHipScene** out);
void pbrt_hip_scene_destroy(PbrtHipScene* scene);
/* Traversal layout of a scene. Beside the 64-B child-pair records of the reference's tree a scene may carry 4-wide,
 * 48-byte records with 8-bit conservative boxes laid over the same tree (two levels of BVHAccel's nodes per record,
 * src/accelerators/bvh.rs:129-135): the traversal kernels then walk those and decide every leaf visit with
 * Bounds3f::intersect_p (src/core/geometry.rs:709-751) on the exact leaf box, so results are the reference's bit for
 * bit. Two-level scenes carry them for the top-level tree and for every object aggregate. n_records = number of wide
 * records (all trees of the scene), 0 when the tree is a single leaf, -1 when the scene has none; *reason then says why
 * (spheres, a leaf with more than 4 primitives, coordinates beyond 2^20, PBRT_WIDE_BUILD_NONE ...). The string lives as long
 * as the scene. */
int pbrt_hip_scene_wide_records(const PbrtHipScene* scene, int32_t* n_records, const char** reason);
/* Diagnostic (not needed by a renderer; the test that the device builder and the host builder of those records produce the
 * same bytes reads them through it): copies a single-level scene's wide records back to the host — n_records x 12 dwords,
 * n_slots x 12 floats (triangles in the records' order), n_slots x 8 floats (exact leaf boxes); n_slots must be the scene's
 * triangle count, any of the three pointers may be NULL. PBRT_HIP_ERR_INVALID for scenes without wide records and for
 * two-level scenes. */
int pbrt_hip_debug_wide_export(PbrtHipScene* scene, uint32_t* nodes, float* tris, float* boxes, int32_t n_slots);
/* Where the scenes created on this context from now on get those records:
 *   PBRT_WIDE_BUILD_DEVICE  laid out on the device from the flat tree (csrc/wide_gpu.hip; two-level scenes: on the host) — the default;
 *   PBRT_WIDE_BUILD_HOST    by the host builder (csrc/host_wide.cpp): the same bytes, tens of ms per million triangles slower
 *                           (kept for the test of exactly that);
 *   PBRT_WIDE_BUILD_NONE    not at all: the scene is traced over the binary records. */
enum { PBRT_WIDE_BUILD_DEVICE = 0, PBRT_WIDE_BUILD_HOST = 1, PBRT_WIDE_BUILD_NONE = 2 };
int pbrt_hip_context_set_wide_build(PbrtHipContext* ctx, int where);
/* How the scenes created on this context from now on keep those records and their triangles in HBM: PBRT_WIDE_LAYOUT_AUTO (the
 * default) = packed, 48 bytes apart, while records + triangles fit 8 MiB (a tree that L2 holds), else one 64-byte line each (no
 * record straddles two lines: +1 % of a config-3 frame for a third more memory); _PACKED / _LINES force one or the other (the
 * tests run every scene both ways: same hits, same exported bytes). Two-level scenes are always packed (the trees of a scene of
 * instances are small). pbrt_hip_scene_wide_stride: 48 or 64, 0 without wide records. */
enum { PBRT_WIDE_LAYOUT_AUTO = 0, PBRT_WIDE_LAYOUT_PACKED = 1, PBRT_WIDE_LAYOUT_LINES = 2 };
int pbrt_hip_context_set_wide_layout(PbrtHipContext* ctx, int layout);
int pbrt_hip_scene_wide_stride(const PbrtHipScene* scene);

/* ---- batch Primitive::intersect / intersect_p (src/core/primitive.rs:17-30 via
 * Scene::intersect / intersect_p, src/core/scene.rs:40-46) ----
 * rays[i].t_max is the input upper bound; out[i].t <= it on a hit. Host buffers. */
int pbrt_hip_intersect(PbrtHipScene* scene, const PbrtRay* rays, int64_t n, PbrtHit* out);
int pbrt_hip_intersect_p(PbrtHipScene* scene, const PbrtRay* rays, int64_t n, uint8_t* out);
/* Same with device-resident buffers on the context's stream; returns after enqueueing.
 * `out_flags` (any-hit) may alias nothing else. */
int pbrt_hip_intersect_device(PbrtHipScene* scene, const PbrtRay* d_rays, int64_t n, PbrtHit* d_out);
int pbrt_hip_intersect_p_device(PbrtHipScene* scene, const PbrtRay* d_rays, int64_t n, uint8_t* d_out);
/* Blocks until everything enqueued on the context's stream has finished. */
int pbrt_hip_synchronize(PbrtHipContext* ctx);
/* Average duration (ms, HIP events on the context's stream) of the traversal-kernel launches
 * since the last call with reset != 0, and their count. */
int pbrt_hip_trace_timing(PbrtHipContext* ctx, int reset, double* total_ms, uint64_t* launches);

/* Which traversal kernel the context's intersect / render calls launch (src/accelerators/bvh.rs:828-932 in every case: same
 * visiting order, same tests, same hits):
 *   PBRT_TRAVERSAL_AUTO      the 4-wide quantised records where the scene has them, else the binary records with a stack;
 *   PBRT_TRAVERSAL_STACK     the binary child-pair records with the per-lane stack (the reference's own form);
 *   PBRT_TRAVERSAL_STACKLESS the binary records with parent links and a 64-bit trail instead of a stack — single-level
 *                            triangle scenes only: calls on other scenes fail with PBRT_HIP_ERR_INVALID.
 * AUTO is the fast one (DESIGN.md section 4.6 has all three on BASELINE config 3). */
enum { PBRT_TRAVERSAL_AUTO = 0, PBRT_TRAVERSAL_STACK = 1, PBRT_TRAVERSAL_STACKLESS = 2 };
int pbrt_hip_context_set_traversal(PbrtHipContext* ctx, int traversal);

/* Instrumentation for the roofline accounting: when enabled, traversal launches
 * run an instrumented variant that counts the box tests (src/accelerators/bvh.rs:841-842) and
 * triangle tests (src/shapes/triangle.rs:74) the reference's loops perform for the same rays.
 * counters = {rays, node_tests, prim_tests, instance_tests (src/core/primitive.rs:136 calls)}.
 * Slower; never enabled in a timed region. */
int pbrt_hip_set_counting(PbrtHipContext* ctx, int enable);
int pbrt_hip_get_counters(PbrtHipContext* ctx, int reset, uint64_t counters[4]);
/* enable = 2: the kernels over the 4-wide records (pbrt_hip_scene_wide_records) count what THEY fetch instead:
 * counters = {records stepped (48 B each), candidate leaves, triangles loaded (48 B each), rays left to the binary
 * kernel}. These are the product path's own figures (bench.py's gather-rate roofline), not the reference's. */
int pbrt_hip_get_wide_counters(PbrtHipContext* ctx, int reset, uint64_t counters[4]);
/* Measured ceiling for that roofline: every lane of a fully resident grid (waves_per_simd = 4, 5, 6 or 8 waves per
 * SIMD) walks `iters` dependent fetches of record_bytes-sized records (48: three 16-B loads, 64: four) from a table of
 * table_bytes of pseudo-random data, the next index computed from the bytes just loaded, and does nothing else.
 * Returns records per second. Measurement only. */
int pbrt_hip_probe_gather(PbrtHipContext* ctx, int64_t table_bytes, int32_t record_bytes, int32_t waves_per_simd, int32_t iters,
                          double* records_per_second);
/* The shading kernel's access pattern with a known byte count, for calibrating a profiler's memory counters on it
 * (profiles/r04_fetch_size_calibration_shade.txt): a shade queue of ascending path numbers (density_permille of n_paths) and,
 * per queued path, `parts` of: 1 nine 16-B SoA records, 2 two 32-B ray records (slot-major, 32-B stride), 4 one 8-B and two
 * 4-B scalars, 8 one random 48-B record of a gather_table_bytes table, 16 the stores (five 16-B, three 32-B, 8 + 4 + 4 B and
 * seven 4-B queue words). Three launches; bytes_read / bytes_written = what the lanes of ONE launch ask for. Measurement only. */
int pbrt_hip_probe_state_stream(PbrtHipContext* ctx, int64_t n_paths, int32_t density_permille, int64_t gather_table_bytes,
                                int32_t parts, int64_t* bytes_read, int64_t* bytes_written, double* ms);

/* ---- Integrator::render (src/core/integrator.rs:29-42, 399-480) for this GPU's tile set ----
 * film_xyzw: width*height*4 floats {xyz[3], filter_weight_sum} = the first 16 bytes of the
 * reference's Pixel (src/core/film.rs:9-15); pixels outside this GPU's tiles are zero — but for the footprint of its own
 * samples across a tile border (wider filters; with the 0.5 box the few samples whose film position rounds onto a pixel
 * border, which add_sample gives to both neighbours) — so the per-GPU films sum to the frame: bit for bit inside the tiles, up
 * to the order of the float additions on their borders. pbrt_hip_render writes a host buffer; _device leaves the
 * film in a caller-supplied device buffer (e.g. for an RCCL reduce). stats may be NULL. */
int pbrt_hip_render(PbrtHipScene* scene, const PbrtCamera* camera, const PbrtRenderParams* params,
                    float* film_xyzw, PbrtRenderStats* stats);
int pbrt_hip_render_device(PbrtHipScene* scene, const PbrtCamera* camera, const PbrtRenderParams* params,
                           float* d_film_xyzw, PbrtRenderStats* stats);

/* ---- Integrator::li (src/core/integrator.rs:29-42) in batch form ----
 * What SamplerIntegrator::render calls per camera sample (integrator.rs:452): `li(&mut ray, scene, sampler, 0) -> Spectrum`.
 * The host keeps its own Camera, Sampler and Film (north_star) and hands over, per call: the camera ray and the random
 * stream the integrator is to draw from. stream_keys[i] is the argument of RNG::set_sequence (src/core/rng.rs:21-35)
 * of ray i's RandomSampler (the reference clones one sampler per tile, sampler.rs:452 / integrator.rs:414-415; here one
 * stream per call keeps calls independent, as pbrt_hip_render's (pixel, sample) streams do); draws_before_li values
 * have already been drawn from that stream by the caller when li starts (render draws the CameraSample's 5 first,
 * integrator.rs:430: with 5 and pbrt_hip_camera_rays' rays and keys the results are pbrt_hip_render's samples bit for
 * bit). rgb[3 i ..] = the returned Spectrum, unguarded (the NaN / negative test is render's, integrator.rs:455).
 * Re-entrant like the reference's li: calls on one context are serialised inside. n = 0 is a no-op. */
typedef struct PbrtLiParams {
    int32_t integrator;     /* PbrtIntegratorKind */
    int32_t max_depth;
    float rr_threshold;     /* path */
    int32_t light_strategy; /* as PbrtRenderParams.light_strategy */
    int32_t ao_samples;     /* ao */
    int32_t draws_before_li;
} PbrtLiParams;
int pbrt_hip_li(PbrtHipScene* scene, const PbrtLiParams* params, const PbrtRay* rays, const uint64_t* stream_keys, int64_t n,
                float* rgb, PbrtRenderStats* stats);
int pbrt_hip_li_device(PbrtHipScene* scene, const PbrtLiParams* params, const PbrtRay* d_rays, const uint64_t* d_stream_keys,
                       int64_t n, float* d_rgb, PbrtRenderStats* stats);
/* The camera-ray stage of pbrt_hip_render on its own: Sampler::get_camera_sample (src/core/sampler.rs:27-33) +
 * Camera::generate_ray (src/cameras/perspective.rs:90-112 ...) for every sample of this GPU's tiles (params as for
 * pbrt_hip_render; whole 16x16 tiles, all samples of a pixel in one pass: params->spp of them, or what the sampler makes of that
 * count — sampler_x * sampler_y, the next power of two — and n_out says so). Per path: the ray, its stream key, the
 * film position p_film (x, y) and {pixel x, pixel y, sample index}; pixels of border tiles outside the pixel bounds
 * carry pixel = (-1, -1) and a ray with t_max < 0. n_out = number of paths; PBRT_HIP_ERR_INVALID with n_out set if
 * capacity is too small. A host without a Camera of its own (tests, the C example) feeds pbrt_hip_li from this. */
int pbrt_hip_camera_rays(PbrtHipScene* scene, const PbrtCamera* camera, const PbrtRenderParams* params, int64_t capacity,
                         PbrtRay* rays, uint64_t* stream_keys, float* p_film, int32_t* pixel_sample, int64_t* n_out);

/* A film on the context's device ({xyz[3], filter_weight_sum} per pixel, zero-filled) for a host that has no HIP binding of
 * its own (a Rust or C caller of this header): pbrt_hip_render_device renders into it, pbrt_hip_film_reduce merges the ranks'
 * films in it, pbrt_hip_film_download (which first waits for the context's stream) copies it to the host. Destroy before the
 * context. PBRT_HIP_ERR_OOM when the device has no room. A host WITH a device allocator (bench.py: a torch tensor) may pass
 * its own pointer to the same calls instead. */
int pbrt_hip_film_create(PbrtHipContext* ctx, int64_t n_pixels, float** d_film_out);
int pbrt_hip_film_download(PbrtHipContext* ctx, const float* d_film_xyzw, int64_t n_pixels, float* film_xyzw);
void pbrt_hip_film_destroy(PbrtHipContext* ctx, float* d_film_xyzw);

/* ---- multi-GPU film merge (replaces the cross-tile part of Film::merge_film_tile,
 * src/core/film.rs:93-123, and parallel_for_2d's join, src/core/parallel.rs:4-21) ----
 * One process per GPU. Rank 0 draws a communicator id and hands it to the other ranks out of band (the host
 * renderer's launcher: a file, MPI, torch.distributed ...); every rank then creates its communicator on its own
 * context (ncclCommInitRank over xGMI) and, after pbrt_hip_render_device, sums the per-rank films in place with one
 * RCCL reduce on the context's stream: root >= 0 leaves the frame on that rank, root < 0 on every rank.
 * RCCL is loaded on first use; pbrt_hip_comm_last_error gives the text when no context is at hand. */
typedef struct PbrtHipComm PbrtHipComm;
#define PBRT_HIP_COMM_ID_BYTES 128
int pbrt_hip_comm_unique_id(uint8_t id[PBRT_HIP_COMM_ID_BYTES]);
int pbrt_hip_comm_create(PbrtHipContext* ctx, int32_t world, int32_t rank, const uint8_t id[PBRT_HIP_COMM_ID_BYTES],
                         PbrtHipComm** out);
void pbrt_hip_comm_destroy(PbrtHipComm* comm); /* before pbrt_hip_context_destroy of the context it was created on */
int pbrt_hip_film_reduce(PbrtHipComm* comm, float* d_film_xyzw, int64_t n_pixels, int32_t root);
const char* pbrt_hip_comm_last_error(void);

/* Tile partition used by pbrt_hip_render (host only, no GPU needed) — what replaces the tile loop of
 * parallel_for_2d! (src/core/parallel.rs:4-21, src/core/integrator.rs:402-416) across GPUs: the 16x16 tiles of the pixel
 * bounds are put into `order` (PbrtTileOrder: Morton order of the tile grid, or row-major) and the k-th tile of that order
 * belongs to rank k % world. Writes this rank's tile origins (x, y pairs, in row-major order) to origins_xy (capacity in tiles)
 * and their count to n_out; returns PBRT_HIP_ERR_INVALID if the capacity is too small (n_out then holds the need).
 * pbrt_hip_tile_partition is the Morton deal (PbrtRenderParams.tile_order = 0, the default). */
int pbrt_hip_tile_partition(int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t rank, int32_t world,
                            int32_t* origins_xy, int32_t capacity, int32_t* n_out);
int pbrt_hip_tile_partition_order(int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t rank, int32_t world, int32_t order,
                                  int32_t* origins_xy, int32_t capacity, int32_t* n_out);

/* Reconstruction filters of src/filters/{boxf,gaussian,mitchell,sinc,triangle}.rs tabulated as Film::new
 * does (src/core/film.rs:52-63). a, b: gaussian alpha / mitchell B, C / lanczos tau. Host only. */
enum PbrtFilterType { PBRT_FILTER_BOX = 0, PBRT_FILTER_GAUSSIAN = 1, PBRT_FILTER_MITCHELL = 2, PBRT_FILTER_LANCZOS = 3,
                      PBRT_FILTER_TRIANGLE = 4 };
int pbrt_hip_filter_table(int32_t type, float radius_x, float radius_y, float a, float b, float table256[256]);
/* Film::get_sample_bounds (src/core/film.rs:76-81) of the whole film: {x0, y0, x1, y1}. Host only. */
int pbrt_hip_sample_bounds(int32_t width, int32_t height, float radius_x, float radius_y, int32_t bounds[4]);
/* Image output: Film::write_image ends in a file writer that is todo!() in the reference
 * (src/core/imageio.rs:3-5); this writes the RGB image as a little-endian PFM (top row first in memory). */
int pbrt_hip_write_pfm(const char* path, const float* rgb, int32_t width, int32_t height);
/* The same image as an 8-bit sRGB PNG (pbrt-v3's WriteImage for ".png": gamma-corrected, 255 v + 0.5 clamped). */
int pbrt_hip_write_png(const char* path, const float* rgb, int32_t width, int32_t height);
/* ... and as OpenEXR, pbrt-v3's default (".exr"): scanline file, float32 B / G / R channels, uncompressed, linear. */
int pbrt_hip_write_exr(const char* path, const float* rgb, int32_t width, int32_t height);

/* Film::write_image's per-pixel arithmetic (src/core/film.rs:153-178, without the file
 * writer, which is todo!() in the reference): rgb = max(0, xyz_to_rgb(xyz) / filter_weight_sum). Host. */
void pbrt_hip_film_to_rgb(const float* film_xyzw, int64_t n_pixels, float* rgb);

#ifdef __cplusplus
}
#endif
#endif /* PBRT_HIP_H */
