/*
 * hrt.h -- C ABI of the MI355X ray-trace path (libhrt.so).
 *
 * This is the drop-in boundary for the reference's 'r'-triggered render:
 *   key 'r'                         /root/reference/main.cpp:321-325
 *   void ray_trace_from_camera()    /root/reference/main.cpp:200-263
 *     -> trace_line()               /root/reference/main.cpp:183-198
 *        -> Scene::rayTrace()       /root/reference/src/Scene.h:345-350
 * The reference has no FFI of its own; a maintainer replaces
 * the body of ray_trace_from_camera() with: flatten scene -> hrt_scene_create
 * -> hrt_render -> PPM dump.
 *
 * Everything here is plain C: pointers, sizes, PODs.  No torch / HIP types.
 * All functions return 0 on success or a negative hrt_status; they never
 * throw.  hrt_last_error() gives the text of the last failure on the calling
 * thread.  Calls are blocking unless a stream is given.
 */
#ifndef HRT_H
#define HRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libhrt.so is built with -fvisibility=hidden: the functions declared here are everything it exports. */
#define HRT_API __attribute__((visibility("default")))

/* ---- compile-time constants of the path (reference src/Constants.h) ---- */
#define HRT_MAXBOUNCES 6             /* Constants.h:11  MAXBOUNCES            */
#define HRT_NB_ECH 10                /* Constants.h:12  shadow rays per light  */
#define HRT_EPSILON 0.00001          /* Constants.h:18  (a double literal)     */
#define HRT_TRIANGLE_SCALING 1.000001f /* Mesh.h:23                             */

typedef enum hrt_status {
    HRT_OK = 0,
    HRT_ERR_INVALID = -1,   /* bad argument / inconsistent scene description   */
    HRT_ERR_DEVICE = -2,    /* HIP runtime failure (no GPU, OOM, launch error) */
    HRT_ERR_STATE = -3,     /* library not initialised / scene destroyed       */
    HRT_ERR_IO = -4         /* file could not be read / written                */
} hrt_status;

/* Material.h:11-21 */
enum { HRT_MAT_DIFFUSE = 0, HRT_MAT_GLASS = 1, HRT_MAT_MIRROR = 2 };
enum { HRT_TEX_NONE = 0, HRT_TEX_CHECKER = 1, HRT_TEX_IMAGE = 2 };
/* Mesh.h:64-68 */
enum { HRT_COLOR_VERTEX = 0, HRT_COLOR_FACE = 1, HRT_COLOR_NONE = 2 };

/* The fields of reference `struct Material` (Material.h:23-50) that the hot
 * path reads.  ambient/specular/shininess are never read by the integrator
 * (Scene.h:317 is commented out) and are not carried. */
typedef struct hrt_material {
    float albedo[3];          /* diffuse_material                              */
    float transparency;
    float index_medium;
    int32_t type;             /* HRT_MAT_*                                      */
    int32_t texture_type;     /* HRT_TEX_*                                      */
    float checker1[3];
    float checker2[3];
    float tex_scale_x, tex_scale_y;
    int32_t emissive;         /* 0/1; undefined in the reference => 0 (N10)     */
    float light_color[3];
    float light_intensity;
    int32_t image;            /* index into images[], -1 = none                 */
    int32_t normal_map;       /* index into images[], -1 = no normal map        */
    float motion[3];          /* motion_blur_translation                        */
} hrt_material;

/* ppmLoader::ImageRGB (imageLoader.h:18-22): tightly packed RGB8, row-major. */
typedef struct hrt_image {
    int32_t w, h;
    const uint8_t *rgb;
} hrt_image;

/* Sphere.h:45-47 */
typedef struct hrt_sphere {
    float center[3];
    float radius;
    int32_t material;
} hrt_sphere;

/* Square: intersect() reads vertices[0],[1],[3] (Square.h:68-70); the normal
 * map frame is m_right_vector / m_up_vector as left by setQuad (Square.h:35-45,
 * Scene.h:284), which later transforms do NOT update. */
typedef struct hrt_quad {
    float v0[3], v1[3], v3[3];
    float tangent[3], bitangent[3];
    int32_t material;
} hrt_quad;

/* Scene.h:28-41; only pos/radius/material are read (Scene.h:306-333). */
typedef struct hrt_light {
    float pos[3];
    float radius;
    float color[3];
} hrt_light;

/* Flattened KD-tree with ropes, in 16-byte units ("nodelets").
 *   ref = unit index | HRT_KD_LEAF (leaf) ; HRT_KD_NIL = no neighbour.
 *   inner nodelet (1 unit):  { f32 split, u32 axis, u32 left_ref, u32 right_ref }
 *   leaf  nodelet (4 units): { bmin.xyz, u32 tri_first | bmax.xyz, u32 tri_count |
 *                              rope[-x,+x,-y,+y] | rope[-z,+z], 0, 0 }
 * tri_first/tri_count index leaf_tris[] (triangle ids of the mesh).
 * Any numbering is valid: hrt_scene_create re-lays the reachable part of the tree for its walk (two-level treelets,
 * breadth-first, so that a prefix of ITS array is the top of the tree -- what the kernels stage into LDS).  The host
 * builder keeps a node and its inner children inside one 64-byte line and starts leaves on 64-byte boundaries (unused
 * padding units are zero), which is what the CPU-side walks of the same array like. */
#define HRT_KD_LEAF 0x80000000u
#define HRT_KD_NIL 0xFFFFFFFFu
typedef struct hrt_kdunit {
    uint32_t w[4];
} hrt_kdunit;

/* The SPLIT SEARCH of the tree build as a replaceable step ("GPU (or parallel host) flattened KD build with
 * SAH", replacing KDTree::buildTree, KDTree.cpp:87-151).  The host layer prepares the triangle references of a mesh
 * (id + bounds, the padded root cell, the heuristic's constants), hands them to a builder, and turns the builder's nodes
 * into the rope tree of hrt_kdunit above.  Two builders exist and give the SAME nodes: the host's own (threaded, the
 * default) and hrt_kd_build_gpu in libhrt.so (level by level on the device); hrt_host_scene_set_kd_builder selects. */
typedef struct hrt_kd_build_input {
    uint32_t n_refs;
    const uint32_t *ids;   /* triangle id of each reference                                                  */
    const float *lo, *hi;  /* 3 floats per reference: bounds of the triangle (of the part inside the cell)   */
    float cell_lo[3], cell_hi[3];  /* the root cell (padded hull)                                            */
    uint32_t leaf_max, max_depth;  /* a node of <= leaf_max references, or at depth max_depth, is a leaf     */
    float cost_traverse, cost_intersect, empty_bonus;  /* surface-area heuristic                              */
} hrt_kd_build_input;
typedef struct hrt_kd_build_node {
    int32_t axis;          /* 0..2: inner node, -1: leaf                                                     */
    float split;
    int32_t left, right;   /* inner: indices into nodes[]                                                    */
    float lo[3], hi[3];    /* the node's cell                                                                */
    uint32_t first_tri, n_tris;  /* leaf: its triangle ids are tris[first_tri .. first_tri + n_tris), ascending */
} hrt_kd_build_node;
typedef struct hrt_kd_build_output {   /* arrays allocated by the builder with malloc(); the caller free()s them */
    hrt_kd_build_node *nodes;
    uint32_t n_nodes;
    uint32_t *tris;
    uint32_t n_tris;
    int32_t root;
    uint32_t depth;        /* deepest node */
} hrt_kd_build_output;
typedef int (*hrt_kd_builder_fn)(const hrt_kd_build_input *in, hrt_kd_build_output *out, void *user);

/* IRREGULAR triangles (hai719-raytracing_amd/host/ref_tree.h).  The reference only finds a triangle through the leaves of
 * its own KD-tree (KDTree.cpp:31-69): a ray tests it when it passes the box of a leaf that holds it.  That is
 * unobservable except for triangles its builder drops below depth 100 (KDTree.cpp:101) and for
 * near-degenerate slivers, whose barycentric test (Triangle.h:62-75) accepts phantom points far outside the triangle.
 * Those triangles are kept OUT of the flattened tree (not listed in leaf_tris) and tested exactly when the reference
 * would: when AABB::intersects (AABB.h:48-65) passes for the box of one of the reference leaves that hold them.
 * One entry per (triangle, reference leaf, box) -- a leaf is one box, or a few when ancestors stick in (`group`) -- in any
 * order; hrt_scene_create groups them by triangle under a small bounding hierarchy: a ray tests such a triangle at most once,
 * and asks its boxes only for a hit closer than the best. */
typedef struct hrt_tri_exception {
    uint32_t triangle;           /* triangle id of the mesh */
    float box_min[3], box_max[3];
    uint32_t group;              /* which reference leaf this box belongs to: the entries of one triangle with the same group are the
                                    boxes the ray must ALL pass to reach that leaf -- the leaf's own box, and those of its ancestors
                                    that do not contain it (the reference cuts a node at the median of UNCLIPPED triangle bounds,
                                    KDTree.cpp:87-98: the plane can lie outside the node and a child then sticks out of its parent).
                                    The triangle is tested when some group passes entirely.                                        */
} hrt_tri_exception;

typedef struct hrt_mesh {
    uint32_t n_vertices, n_triangles;
    const float *positions;      /* 3*n_vertices, world space, NOT yet scaled by
                                    HRT_TRIANGLE_SCALING (KDTree.cpp:38-40)      */
    const uint32_t *indices;     /* 3*n_triangles                               */
    int32_t color_type;          /* HRT_COLOR_*                                 */
    const float *vert_colors;    /* 3*n_vertices or NULL                        */
    const float *face_colors;    /* 3*n_triangles or NULL                       */
    float aabb_min[3], aabb_max[3]; /* Mesh::computeAABB (Mesh.h:143-157)       */
    int32_t material;
    /* flattened KD-tree (built by the host layer, hrt_host.h) */
    uint32_t kd_root;            /* ref of the root                             */
    float kd_min[3], kd_max[3];  /* root cell of the tree (scaled-triangle hull, padded) */
    uint32_t n_kd_units;
    const hrt_kdunit *kd_units;
    uint32_t n_leaf_tris;
    const uint32_t *leaf_tris;
    /* irregular triangles (may be 0 / NULL: then every triangle must be in the tree) */
    uint32_t n_exceptions;
    const hrt_tri_exception *exceptions;
} hrt_mesh;

typedef struct hrt_scene_desc {
    uint32_t n_materials;  const hrt_material *materials;
    uint32_t n_spheres;    const hrt_sphere *spheres;
    uint32_t n_quads;      const hrt_quad *quads;
    uint32_t n_meshes;     const hrt_mesh *meshes;
    uint32_t n_lights;     const hrt_light *lights;
    uint32_t n_images;     const hrt_image *images;
    int32_t dark_sky;      /* Scene.h:65                                        */
    int32_t skybox_image;  /* index into images[] or -1 (Scene.h:149-161)       */
} hrt_scene_desc;

/* Replaces the GL read-back of matrixUtilities.h:33-74: eye + orthonormal
 * basis + the gluPerspective parameters of Camera.cpp:24-28,41-50.
 * Reference default: eye (0,0,6.1), right +X, up +Y, forward -Z, fovy 45,
 * znear 4.1, zfar 1e4, aspect = w/h. */
typedef struct hrt_camera {
    float eye[3];
    float right[3], up[3], forward[3];
    float fovy_deg;
    float aspect;
    float znear, zfar;
} hrt_camera;

/* Image-tile partition of one frame across ranks (one process per GPU).
 * Tiles are HRT_TILE x HRT_TILE pixels, numbered row-major; rank r renders
 * tiles r, r+world, r+2*world, ... and writes them densely, tile-major, into
 * its own buffer (hrt_tiles_owned() tiles of HRT_TILE*HRT_TILE*3 floats). */
#define HRT_TILE 8

enum {
    HRT_FLAG_GAMMA = 1u,       /* apply pow(c,1/2.2) (main.cpp:196)             */
    HRT_FLAG_NO_LDS_TREE = 2u, /* debug: fetch every nodelet from global memory */
    HRT_FLAG_WAVE_KERNEL = 4u, /* force the one-pixel-per-lane kernel                                             */
    HRT_FLAG_STREAM_KERNEL = 8u, /* force the workgroup-streaming kernel (the default for scenes with meshes or lights) */
    HRT_FLAG_NO_SHADOW_CULL = 16u, /* debug: shadow rays test every sphere (the reference's loop) instead of the culled groups */
    HRT_FLAG_DUAL_KERNEL = 32u, /* force the two-streams-per-lane kernel (mesh scenes; all kernel forms give identical pixels) */
    /* Proof builds of the lane-per-pixel and streaming kernels: no filter and no reciprocal approximation anywhere in
     * front of the reference arithmetic.  Every square goes through Square::intersect's arithmetic in index order
     * (Square.h:65-126, Scene.h:214-221), every mesh gate through AABB::intersects' fp64 form (AABB.h:48-65,
     * KDTree.cpp:82), shadow rays test every sphere (Scene.h:235-255), the camera quotient is a true fp64 division
     * (matrixUtilities.h:66-68).  Slow; exists so tests can show that the default path's filters never change a pixel. */
    HRT_FLAG_EXACT_ONLY = 64u,
    /* With HRT_FLAG_EXACT_ONLY: meshes are not walked through the KD-tree at all -- every triangle of a gated mesh is
     * tested (Mesh::intersectOld, Mesh.h:257-277).  Shows that the rope walk never skips the closest triangle. */
    HRT_FLAG_MESH_BRUTE = 128u
};

typedef struct hrt_stats {
    double kernel_ms;          /* HIP-event time of the trace kernel(s)         */
    double total_ms;           /* wall time of the call                         */
    uint64_t samples;          /* pixels * spp rendered by this call            */
    uint32_t vgprs, sgprs, lds_bytes, waves_launched;
} hrt_stats;

typedef struct hrt_scene hrt_scene;   /* opaque: device-resident SoA scene */

/* Prepares `device_ordinal` (once) and makes it the current device of the library: scenes are created on the current
 * device and stay there.  May be called for several devices.  Entry points that take an hrt_scene or an hrt_multi switch
 * the calling thread to that scene's device (every time: HIP's current device is per thread) and leave it current.
 * Entry points that only take device POINTERS -- hrt_assemble_frame, hrt_finalize_tiles, hrt_encode_ppm -- and the
 * scene-less debug calls (hrt_debug_kat, hrt_debug_path_stream) do not switch: they run on the calling thread's current
 * device, which must be the one the pointers live on. */
HRT_API int hrt_init(int device_ordinal);
HRT_API void hrt_shutdown(void);
HRT_API const char *hrt_last_error(void);
HRT_API int hrt_device_count(void);

/* Upload: repack the description into device SoA arrays.  The description
 * (and everything it points to) may be freed after the call returns. */
HRT_API int hrt_scene_create(const hrt_scene_desc *desc, hrt_scene **out);
HRT_API void hrt_scene_destroy(hrt_scene *scene);

/* Whole frame on the current device into a HOST buffer out_rgb[h*w*3]
 * (row-major x + y*w, as main.cpp:193).  Value = mean over spp of
 * Scene::rayTrace, gamma-corrected when HRT_FLAG_GAMMA. */
HRT_API int hrt_render(hrt_scene *scene, const hrt_camera *cam, uint32_t w, uint32_t h,
               uint32_t spp, uint64_t seed, uint32_t flags, float *out_rgb,
               hrt_stats *stats /* may be NULL */);

/* Several GPUs from ONE process -- the multi-GPU form of the reference's single caller ray_trace_from_camera()
 * (main.cpp:200-263).  Slot i of `device_ordinals` holds a replica of the scene on that device, renders the image tiles
 * i, i + n, i + 2n, ... on a stream of its own (all slots run at once), and its dense tile buffer travels device to
 * device (xGMI between GPUs) into its block of a gather buffer on slot 0's device: ONE gather step -- an RCCL gather, or
 * peer copies, see hrt_multi_gather -- and no reduction (slots own disjoint pixels).  Slot 0 then de-interleaves the tiles and copies the frame to out_rgb (host, h*w*3).  The pixels
 * are bit-identical to hrt_render's for any number of slots.  An ordinal may be repeated (several slots share a GPU),
 * which makes the path testable on a one-GPU machine.  hrt_multi_create prepares every listed device (hrt_init is not
 * needed first) and leaves slot 0's device current; stats: kernel_ms = the slowest slot's kernel.
 * hrt_render_multi = create + render + destroy in one call. */
typedef struct hrt_multi hrt_multi;
HRT_API int hrt_multi_create(const hrt_scene_desc *desc, uint32_t n_devices, const int *device_ordinals, hrt_multi **out);
HRT_API int hrt_multi_render(hrt_multi *m, const hrt_camera *cam, uint32_t w, uint32_t h, uint32_t spp, uint64_t seed,
                     uint32_t flags, float *out_rgb, hrt_stats *stats /* may be NULL */);
HRT_API void hrt_multi_destroy(hrt_multi *m);
/* Which gather this handle runs: "rccl" -- one ncclGather (rccl.h:745) to slot 0 over the communicators hrt_multi_create
 * made with ncclCommInitAll, the default whenever the ordinals are distinct (one slot included) -- or "peer" --
 * hipMemcpyPeerAsync per slot, used when an ordinal repeats or when HRT_MULTI_GATHER=peer is set in the environment
 * (HRT_MULTI_GATHER=rccl makes a missing librccl.so or a failed communicator an error instead of a fallback).  After a
 * successful hrt_multi_create, hrt_last_error() holds a note about anything that was fallen back from (else ""). */
HRT_API const char *hrt_multi_gather(const hrt_multi *m);
HRT_API int hrt_render_multi(const hrt_scene_desc *desc, const hrt_camera *cam, uint32_t w, uint32_t h, uint32_t spp,
                     uint64_t seed, uint32_t flags, uint32_t n_devices, const int *device_ordinals, float *out_rgb,
                     hrt_stats *stats /* may be NULL */);

/* Multi-GPU building blocks (device pointers; `stream` is a hipStream_t cast
 * to void*, NULL = the default stream).  Asynchronous w.r.t. the host. */
HRT_API uint32_t hrt_tiles_total(uint32_t w, uint32_t h);
HRT_API uint32_t hrt_tiles_owned(uint32_t w, uint32_t h, uint32_t rank, uint32_t world);
HRT_API int hrt_render_tiles(hrt_scene *scene, const hrt_camera *cam, uint32_t w,
                     uint32_t h, uint32_t spp, uint64_t seed, uint32_t flags,
                     uint32_t rank, uint32_t world,
                     float *d_tiles /* device, hrt_tiles_owned()*HRT_TILE^2*3 */,
                     void *stream);
/* Rank 0 after the gather: d_gathered holds world blocks of
 * tiles_per_rank_padded tiles (rank-major); writes the row-major frame. */
HRT_API int hrt_assemble_frame(const float *d_gathered, uint32_t tiles_per_rank_padded,
                       uint32_t w, uint32_t h, uint32_t world,
                       float *d_frame /* device, h*w*3 */, void *stream);
/* One hrt_scene carries one launch at a time (it owns the work-queue head, the path pool and the camera block of the
 * launch).  Launches of the same scene on ONE stream are ordered by the stream; a launch on a different stream is made
 * to wait for the previous one.  Two scenes never interfere.
 *
 * hrt_check_last_launch: waits for the last launch of this scene and returns HRT_ERR_DEVICE when the trace kernel gave
 * up (its scheduler has a cycle bound so that a bug can never spin the GPU): the tiles of that launch are then
 * incomplete and must not be used.  hrt_render and hrt_last_kernel_ms call it themselves; callers of the asynchronous
 * entry points (hrt_render_tiles, hrt_render_accumulate) call it before they consume or ship the tiles. */
HRT_API int hrt_check_last_launch(hrt_scene *scene);
/* Timing of the last hrt_render_tiles on this scene (after a sync). */
HRT_API int hrt_last_kernel_ms(hrt_scene *scene, double *ms);
HRT_API int hrt_kernel_info(hrt_stats *out);

/* Parity instruments (deterministic, no RNG): first-hit AOVs through pixel
 * centres at time 0.  which: 0 = (t, kind, index) with kind 1 sphere / 2 square /
 * 3 mesh and index = object or triangle id (t = 0, index = -1 on a miss),
 * 1 = shading normal, 2 = albedo, 3 = emission.  out_rgb: host, h*w*3. */
HRT_API int hrt_render_aov(hrt_scene *scene, const hrt_camera *cam, uint32_t w, uint32_t h,
                   uint32_t which, float *out_rgb);
/* Draws 0..n-1 of the per-path RNG stream (seed, pixel, sample) as the kernel
 * produces them (DESIGN.md "RNG stream").  out: host, n floats. */
HRT_API int hrt_debug_path_stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float *out);

/* Known-answer instrument: runs the DEVICE functions of the trace path on caller vectors (one lane each), so that a test
 * can compare the device arithmetic bit for bit with vectors produced by the reference's own code instead of inferring
 * it from pixels.  `in` holds n rows, `out` receives n rows (host memory); rays are {origin, direction, time} and get
 * the Ray constructor's normalisation (Line.h:13-16).
 *   which             cam / prim                                in row   out row
 *   HRT_KAT_CAMERA    cam                                       u, v     12: Ray origin + direction (main.cpp:189-192,
 *                                                                        matrixUtilities.h:53-74), then the same from
 *                                                                        the exact-division build (must be equal)
 *   HRT_KAT_TRIANGLE  prim = c0, c1, c2 (Triangle ctor)         ray 7    8: hit, t, w0, w1, w2, normal (Triangle.h:77-126)
 *   HRT_KAT_AABB      prim = lo, hi                             ray 7    2: AABB::intersects (AABB.h:48-65), shipped gate
 *   HRT_KAT_SPHERE    prim = centre, radius, motion             ray 7    9: hit, t, theta, phi, normal, p.x, p.y (Sphere.h:91-132)
 *   HRT_KAT_QUAD      prim = v0, v1, v3, motion, glass          ray 7    8: hit, t, u, v, normal (Square.h:65-126), filter bit
 *   HRT_KAT_OPTICS    -                                         d, n, eta, cosine   8: reflect, refract, reflectance, gamma(|cosine|)
 *   HRT_KAT_NORMALIZE -                                         v        3: v / |v| (Vec3.h:46) */
enum { HRT_KAT_CAMERA = 0, HRT_KAT_TRIANGLE = 1, HRT_KAT_AABB = 2, HRT_KAT_SPHERE = 3, HRT_KAT_QUAD = 4, HRT_KAT_OPTICS = 5,
       HRT_KAT_NORMALIZE = 6 };
HRT_API int hrt_debug_kat(uint32_t which, const hrt_camera *cam, const float *prim, const float *in, uint32_t n, float *out);

/* Cycle counters per kernel stage of the last launch; all zero unless libhrt.so was built with
 * -DHRT_STAMPS (diagnostic build, tools/variants.sh).  out: 16 values. */
HRT_API int hrt_debug_read_stamps(hrt_scene *scene, uint64_t out[16]);

/* Output stage of main.cpp:252-262: P3 ASCII with (int)(255*min(1,c)). */
/* The tree build's split search and partition on the GPU (a hrt_kd_builder_fn; `user` is ignored): level by level, every
 * candidate plane of every open node evaluated in parallel with the host builder's arithmetic and tie-breaking, so the
 * nodes -- and the flattened tree -- are identical to the host builder's (tests compare the arrays).  Exhaustive in the
 * candidates (references x candidates per node): meant for meshes up to a few hundred thousand triangles.  Needs hrt_init. */
HRT_API int hrt_kd_build_gpu(const hrt_kd_build_input *in, hrt_kd_build_output *out, void *user);

HRT_API int hrt_write_ppm(const char *path, const float *rgb, uint32_t w, uint32_t h);

/* ---- progressive rendering / resume (replaces the all-or-nothing sample loop main.cpp:188-195)
 * d_sum_tiles (device, hrt_tiles_owned()*HRT_TILE^2*3 floats, zeroed by the caller before the first call) holds
 * the running per-pixel SUMS of samples [0, first_sample); the call adds samples [first_sample, first_sample +
 * n_samples) in sample order.  Because a sample's random numbers depend only on (seed, pixel, sample index) and
 * the sum continues in the same order, k calls covering [0, N) leave exactly the bits one hrt_render_tiles of N
 * samples would have summed: a render can be previewed, stopped, checkpointed (copy the buffer) and resumed.
 * HRT_FLAG_GAMMA is ignored here; hrt_finalize_tiles applies it. */
HRT_API int hrt_render_accumulate(hrt_scene *scene, const hrt_camera *cam, uint32_t w, uint32_t h,
                          uint32_t first_sample, uint32_t n_samples, uint64_t seed, uint32_t flags,
                          uint32_t rank, uint32_t world, float *d_sum_tiles, void *stream);
/* sums -> pixel means (`image[i] /= nsamples`, main.cpp:195) and, with HRT_FLAG_GAMMA, gamma_correct
 * (main.cpp:196).  d_tiles may alias d_sum_tiles.  The result is what hrt_render_tiles(total_samples) writes. */
HRT_API int hrt_finalize_tiles(const float *d_sum_tiles, uint32_t n_tiles, uint32_t total_samples, uint32_t flags,
                       float *d_tiles, void *stream);
/* The PPM file of main.cpp:252-262 encoded ON THE DEVICE from a row-major frame (device, h*w*3 floats).
 * format 3: the reference's ASCII file byte for byte ("P3\n<w> <h>\n255\n", then "r g b " per pixel, "\n");
 * format 6: the same integers as bytes (binary PPM; negative values, which P3 prints with a sign, clamp to 0).
 * d_out: device buffer of `capacity` bytes (16*w*h + 64 always suffices for non-negative frames; the call
 * fails with the needed size otherwise); *bytes = size of the file.  Synchronises the stream. */
HRT_API int hrt_encode_ppm(const float *d_frame, uint32_t w, uint32_t h, int format, unsigned char *d_out,
                   size_t capacity, size_t *bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HRT_H */
