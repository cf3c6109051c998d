// Device-side scene layout of the MI355X trace path (gfx950 only).
//
// hrt_scene_create() repacks the hrt_scene_desc into the arrays below, all
// resident in HBM for the life of the scene (they total a few MB, so after the
// first touch they live in the per-XCD L2 / Infinity Cache; the top of every
// KD-tree is additionally staged into LDS by each workgroup).
//
//   prim records  float4 rows, read with wave-uniform indices (scalar loads):
//     sphere  2 rows: {c.xyz, r} {motion.xyz, material}
//     quad    5 rows: {p0.xyz, D0} {n.xyz, flags} {R.xyz, |R|} {U.xyz, |U|} {motion.xyz, material}
//             + 2 rows used only when shading: {T.xyz,0} {B.xyz,0}
//   quad filter   DScene::qfilter, wave-uniform rows of the no-division filter (hrt_kernels.hip quad_filter):
//               static squares (nearly) in an axis plane, by normal axis K: {sgn D, centre_I, centre_J, half_I} {half_J, bits, par, cq}
//               all others, (R, U) and (n.y, n.z) components side by side in aligned SGPR pairs:
//               {p0.xyz, D0} {n.y, n.z, n.x, flags | index << 8} {R.x, U.x, R.y, U.y} {R.z, U.z, |R|, |U|}
//   materials     8 float4 rows per material, read per lane at the closest hit (rows 6, 7: geometry of its texture / normal map)
//   meshes        DMesh records (wave-uniform)
//   kd units      uint4 nodelets (include/hrt.h), refs rebased to the global array
//   exceptions    irregular triangles (include/hrt.h hrt_tri_exception), one leaf entry each: {cull lo, soup slot} {cull hi, nb} followed by
//                 its nb reference leaf boxes {lo, 0} {hi, 0} (its rows sit behind the mesh's leaf-ordered soup); bounding entry
//                 {lo, HRT_EXC_INNER} {hi, skip}; threaded depth-first
//   triangles     leaf-ordered soup in two arrays by slot: planes {n, D} (16 B: the planes of a leaf's triangles share a cache line) and
//                 rows {c0, d11} {e1, d00} {e2, d01} {id, -, -, -} (one 64-byte line; three rows read when the plane is hit in front, the id at shading)
//                 (Triangle.h:32-37, 62-75 constants folded on the host in the reference's arithmetic)
//   colours       float4 per face / per vertex (+ uint4 vertex ids per triangle)
//   texels        RGBA8 packed in a u32, one table entry {offset, w, h} per image
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hrt.h"

#define HRT_QUAD_ROWS 7
#define HRT_SPHERE_ROWS 2
#define HRT_MAT_ROWS 8
#define HRT_TRI_ROWS 4   // {c0, d11} {e1, d00} {e2, d01} {id}: one 64-byte line per triangle; the plane {n, D} lives in DScene::tri_planes
#define HRT_EXC_INNER 0xFFFFFFFFu  // first word of a bounding entry of a mesh's exception list (DScene::exceptions)
#define HRT_QUAD_FLAG_GLASS 1u
#define HRT_QUAD_FLAG_MOVING 2u

struct DMesh {
    float aabb_lo[3], aabb_hi[3];  // Mesh::computeAABB box (the reference's gate test)
    float kd_lo[3], kd_hi[3];      // root cell of the flattened tree
    uint32_t root;                 // rebased ref
    uint32_t tri_base;             // first row-triple of this mesh in the soup (in triangles, not rows)
    uint32_t material;
    int32_t color_type;
    uint32_t color_base;           // into colours (face) or into vert ids / vertex colours
    uint32_t vcolor_base;
    uint32_t n_soup;               // rows-of-five in this mesh's leaf-ordered soup (straddlers repeated)
    uint32_t exc_base, n_exc;      // irregular triangles (hrt_tri_exception): entries [exc_base, exc_base + n_exc) of DScene::exceptions
    uint32_t pad1, pad2, pad3;     // 96 bytes
};

struct DImage {
    uint32_t offset;  // into texels
    int32_t w, h;
    uint32_t pad;
};

struct DScene {
    const float4 *spheres;
    const float4 *quads;
    const float4 *materials;
    const DMesh *meshes;
    const uint4 *kd_units;
    const float4 *tris;        // HRT_TRI_ROWS rows per soup slot
    const float4 *tri_planes;  // {n, D} per soup slot
    const float4 *colors;
    const uint4 *tri_vids;
    const DImage *images;
    const uint32_t *texels;
    const float4 *lights;  // 2 rows: {pos.xyz, radius} {color.xyz, 0}
    const float4 *tabs;        // squares | materials | spheres | mesh records | sphere pair-filter rows (| short exception lists) in ONE array (what `quads`, `materials`, `spheres`, `meshes` above
                               // point into): the streaming kernel stages it in LDS for per-lane row fetches (hrt_kernels.hip CtxT)
    uint32_t tab_quads, tab_mats, tab_spheres, tab_meshes, tab_exc, tab_rows;  // row offsets of the tables, rows in all
    uint32_t exc_in_tabs;      // 1: the meshes' exception lists are short (<= 512 rows) and sit in `tabs` at tab_exc; 0: in `exceptions`
    const float4 *qfilter;     // rows of the squares' no-division filter: squares in an axis plane by normal axis x, y, z (2 rows each), then the rest (4 rows each)
    uint32_t qf_n[4];          // squares per section
    uint32_t tab_sfilter;      // row offset in `tabs` of the spheres' pair-filter rows (hrt_kernels.hip sphere_filter), 4 per PAIR of spheres
                               // (A = 2p, B = 2p + 1; an odd last sphere is paired with itself), the two spheres' values side by
                               // side: {c.x A, c.x B, c.y A, c.y B} {c.z A, c.z B, r^2 A, r^2 B}
                               // {motion.x A, B, motion.y A, B} {motion.z A, B, |r| A, |r| B}
    uint32_t sf_pairs;         // pairs; sf_psize consecutive pairs share one bit of the shadow rays' 64-bit group mask
    uint32_t sf_psize;
    const float4 *exceptions;  // 2 rows per entry (see above)
    uint32_t n_spheres, n_quads, n_meshes, n_lights, n_images;
    uint32_t n_kd_units;
    int32_t dark_sky, skybox_image;
    uint32_t any_motion;       // some material has a motion vector != 0: only then does a ray's time matter (hrt_stream.hip recomputes it per hit visit)
    uint32_t prune_ok;         // every colour a path's throughput or radiance is multiplied by or added to is finite (materials, lights, mesh
                               // colours; texels are bytes): then throughput x value == 0 whenever the throughput is 0, which the streaming
                               // kernel's exact path pruning (hrt_stream.hip HRT_SP_PRUNE) relies on
};

struct DCamera {
    // The two fp64 mat-vecs of matrixUtilities.h:60-68 with their structural zeros removed.  The host inverts the GL
    // matrices with the reference's own method (hrt_api.hip host_invert4 == gluInvertMatrix, term by term) and checks
    // the sparsity  P^-1 = [pi0 . . .; . pi5 . .; . . . pi14; . . pi11 pi15],  MV^-1 = [* * * *; * * * *; * * * *; 0 0 0 m15].
    // With z = GL_DEPTH_RANGE[0] = 0 the reference's left-to-right sums then reduce EXACTLY (a zero coefficient adds a
    // signed zero, and x + (+-0) == x) to
    //   ri = (pi0*x, pi5*y, pi14, pi15)      r_k = ((mx[k]*ri0 + my[k]*ri1) + c1[k]) + c2[k]      r_3 = fl(m15*pi15)
    double pi0, pi5;
    double pi15, inv15;            // r_3, the divisor of matrixUtilities.h:66-68, and 1/r_3 (fast path of the division, see camera_ray)
    double mx[3], my[3];           // MV^-1 columns 0 and 1
    double c1[3];                  // fl(MV^-1 column 2 * pi14)   (one rounding each, as the reference)
    double c2[3];                  // fl(MV^-1 column 3 * pi15)
    float eye[3];                  // cameraSpaceToWorldSpace(0,0,0), matrixUtilities.h:53-58
    float pad;
};

// Kernel argument block: small on purpose.  The scene and the camera live in device memory and
// are read through constant-address-space pointers (scalar loads at the point of use), which keeps
// the SGPR file for exec masks and primitive rows instead of pinning 60+ SGPRs of arguments.
struct DRender {
    const DScene *scene;
    const DCamera *cam;
    uint32_t w, h, spp;
    uint32_t seed_lo, seed_hi;
    uint32_t flags;
    uint32_t rank, world;
    uint32_t tiles_x, tiles_total, tiles_owned;
    uint32_t s0;               // index of the first sample of this launch (progressive rendering), else 0
    uint32_t accumulate;       // 1: out_tiles holds running sums of samples [0, s0): add this launch's samples, store sums
    uint32_t lds_units;        // leading kd units each workgroup stages into LDS
    float err_abs;             // 2e-6 * (largest |coordinate| of scene + camera): margin of the no-division filters
    float *out_tiles;          // tiles_owned * 64 * 3 floats, tile-major
    uint32_t *tile_counter;    // work queue head, zeroed before every launch
    unsigned long long *stamps;  // 16 cycle counters, written only by -DHRT_STAMPS diagnostic builds
    float *sp_scratch;           // streaming kernel: per-workgroup [sample][pixel][rgb] scratch of one sample chunk
    uint32_t sp_group_log2;      // streaming kernel: log2 of the tiles per work unit
    uint32_t sp_band_log2;       // streaming kernel: a work unit is one ROW BAND of a tile, 8 x (8 >> this) pixels (then one tile per unit): finer
                                 // items on the tile queue when tiles are few and heavy (many samples per pixel)
    uint32_t *sp_pool;           // streaming kernel built with HRT_SP_GLOBAL: per-workgroup path records
};
