/*
 * lidarhip.h -- C ABI of liblidarhip.so, the MI355X (gfx950) virtual-LiDAR ray-cast path.
 *
 * Drop-in boundary for PRBonn/lidar_transfer's native raytracer: every entry point below names
 * the reference interface it replaces (paths relative to the reference repository).  Plain C
 * types only -- pointers, sizes, an opaque handle -- so the library can be bound from ctypes,
 * Cython, cgo, JNI ... exactly where the reference binds `ctrace`
 * (auxiliary/raytracer/RayTracerCython.pyx:5-7).
 *
 * Conventions
 *   - every function returns an int status: LT_OK (0) or a negative LT_ERR_* code; the message
 *     of the last failure on the calling thread is available from lt_last_error();
 *   - "host" pointers are ordinary process memory, "dev" pointers are HIP device pointers on the
 *     scene's device (e.g. torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream);
 *   - array layouts are the reference's: all arrays flat and C-contiguous, rays / endpoints /
 *     endcolors at 3*(W*j + i), range / endrem / tri at W*j + i, W = n_rays / height
 *     (auxiliary/raytracer/RayTracer.cpp:56, :65, :86-89); `colors` holds 3 ints per VERTEX and
 *     only vertex 0 of the hit face is reported (RayTracer.cpp:75, Triangle.h:56-61).
 */
#ifndef LIDARHIP_H
#define LIDARHIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LT_OK 0
#define LT_ERR_INVALID_ARG (-1)  /* NULL pointer, negative size, height <= 0 ...                      */
#define LT_ERR_NO_MEMORY (-2)    /* hipMalloc failed                                                  */
#define LT_ERR_HIP (-3)          /* any other HIP runtime error (message has the HIP error string)    */
#define LT_ERR_BAD_INDEX (-4)    /* a face references a vertex outside [0, n_verts)                   */
#define LT_ERR_NOT_BUILT (-5)    /* trace requested before lt_scene_build                             */
#define LT_ERR_TOO_LARGE (-6)    /* more than LT_MAX_FACES triangles                                  */

#define LT_MAX_FACES (1 << 28)

/* trace flags */
#define LT_TRACE_WRITE_MISSES 1u /* write 0 / tri = -1 for rays that hit nothing (otherwise outputs   */
                                 /* are left untouched for misses, as in RayTracer.cpp:73)            */
#define LT_TRACE_COUNT 2u        /* accumulate nodes visited / triangles tested into the scene stats  */
#define LT_TRACE_NORM_EXACT 4u   /* seed the direction normalisation (Vector3.h:73-89) with a correctly */
                                 /* rounded 1/sqrt instead of the replayed x86 RSQRTSS seed (default:   */
                                 /* the Intel table); vendor independent, last-ulp different            */
#define LT_TRACE_NORM_AMD 16u    /* replay the RSQRTSS seed of an AMD host (measured on EPYC 9575F, Zen 5) instead */
                                 /* of the Intel one: the reference as it runs on the MI355X box's own CPU      */
#define LT_TRACE_LABEL_IMAGE 8u  /* `endcolors` receives [n_rays] ints: colour channel 2 only = the      */
                                 /* semantic label image `deform` unpacks (label_image = ray_colors[:, :, */
                                 /* 2], laserscan.py:912) instead of [n_rays, 3]; device-pointer calls    */

/* Per-phase timings (hipEvent, milliseconds) and work counters of the last build / trace of a scene. */
typedef struct lt_stats {
  float ms_bounds;     /* scene bounds reduction                                        */
  float ms_morton;     /* centroid Morton codes                                         */
  float ms_sort;       /* LSD radix sort (3 passes of 10-bit digits)                    */
  float ms_gather;     /* sorted triangle records + padded leaf boxes                   */
  float ms_segtree;    /* min/max segment tree over the leaf boxes                      */
  float ms_hierarchy;  /* Karras topology into 4-wide nodes + child boxes               */
  float ms_build;      /* whole build, first to last kernel                             */
  float ms_trace;      /* ray-cast kernel(s): k_trace4, or the scatter sequence         */
  int n_faces;
  int n_nodes;         /* node slots (n_faces - 1)                                      */
  int n_rays;
  int n_hits;          /* valid after a LT_TRACE_COUNT trace                            */
  unsigned long long nodes_visited; /* 4-wide nodes fetched (lbvh) / candidate bins (scatter) */
  unsigned long long tris_tested;   /* Moller-Trumbore evaluations, summed over rays    */
  unsigned long long stack_overflows; /* rays that spilled past the LDS stack           */
  unsigned long long entries_culled;  /* lbvh built with -DLT_TRACE_CULL (an A/B, not the default): stack entries dropped
                                         at pop because their box begins behind the best hit found since they were pushed
                                         (BVH.cpp:41); otherwise 0 */
} lt_stats;

typedef struct lt_scene lt_scene; /* opaque: device workspace + BVH of one mesh */

/* ---- one-call drop-in ----------------------------------------------------------------------- */

/*
 * lt_ctrace -- same 14 parameters, order and meaning as the reference's
 *   extern "C" void ctrace(float* rays, float* origin, float* verts, int* faces, int* colors,
 *                          float* rem, int n_rays, int n_verts, int n_faces, int height,
 *                          float* endpoints, int* endcolors, float* range, float* endrem)
 * (auxiliary/raytracer/RayTracer.cpp:116-124), but returns a status instead of void.
 * All pointers are HOST pointers; the call uploads the mesh and rays to the current HIP device,
 * casts the rays and copies the four outputs back.  State (workspace, stream, the binned ray set of the previous
 * call) is kept per calling THREAD: concurrent callers do not serialise each other.  As in the reference, outputs
 * are written only for rays that hit (the caller pre-zeroes them, fusion_lidar.py:440-447).
 */
int lt_ctrace(const float* rays, const float* origin, const float* verts, const int* faces,
              const int* colors, const float* rem, int n_rays, int n_verts, int n_faces, int height,
              float* endpoints, int* endcolors, float* range, float* endrem);

/* As lt_ctrace, plus the hit-triangle image `tri` (face index, -1 = miss; may be NULL) and
 * optional statistics (may be NULL).  The reference has no triangle output (IntersectionInfo.h:9-13
 * keeps the object pointer internal). */
int lt_ctrace_ex(const float* rays, const float* origin, const float* verts, const int* faces,
                 const int* colors, const float* rem, int n_rays, int n_verts, int n_faces, int height,
                 float* endpoints, int* endcolors, float* range, float* endrem, int* tri,
                 lt_stats* stats);

/* ---- host buffers, pipelined: a sequence of scans with their transfers overlapped -------------------------------- */

typedef struct lt_hostpipe lt_hostpipe; /* opaque: `depth` scans in flight, uploader + launcher threads, HIP streams */

/*
 * The work of lt_ctrace for a SEQUENCE of scans that share one ray set (one target sensor model, as in the reference's
 * batch loop lidar_deform.py:393-462, which calls throw_rays_at_mesh -> ctrace once per output scan,
 * fusion_lidar.py:434-451): while scan i renders, scan i + 1 uploads and scan i - 1 downloads.
 *   rays    HOST [n_rays,3] f32, uploaded and binned once;  depth = scans in flight (3 overlap; from 4 on two uploader threads share the link: +7 %)
 *   flags   LT_TRACE_LABEL_IMAGE | LT_TRACE_NORM_EXACT | LT_TRACE_NORM_AMD
 * Unlike lt_ctrace the pipe writes EVERY cell of the output images (misses: 0, tri -1) -- what the reference's
 * pre-zeroed arrays (fusion_lidar.py:440-447) contain after ctrace -- so nothing is uploaded for the outputs.
 */
int lt_hostpipe_create(lt_hostpipe** pipe, const float* rays, int n_rays, int height, int depth, unsigned flags,
                       int device);

/* Queue one scan: mesh in HOST arrays with the layouts of ctrace (RayTracer.cpp:116-124), except that `colors` may be
 * the uint8 [n_verts,3] array get_mesh returns (colors_are_u8 != 0; fusion_lidar.py:423) instead of its int32 copy
 * (:435) -- a quarter of the bytes on the link.  Output HOST arrays as for lt_ctrace_ex (any may be NULL).  Returns
 * immediately; every input and output array must stay valid and untouched until lt_hostpipe_wait(ticket),
 * lt_hostpipe_flush, or until `depth` further scans have been submitted (a submit that needs the slot of an older
 * scan completes that scan first -- its status is what this call returns).  Pageable and pinned memory both run at
 * the wire rate here; pinned (lt_host_alloc) input additionally frees the worker thread during the transfer. */
int lt_hostpipe_submit(lt_hostpipe* pipe, const float* origin, const float* verts, const int* faces,
                       const void* colors, int colors_are_u8, const float* rem, int n_verts, int n_faces,
                       float* endpoints, int* endcolors, float* range, float* endrem, int* tri, int* ticket);

/* Block until the scan with this ticket is complete: its images are in the caller's arrays. */
int lt_hostpipe_wait(lt_hostpipe* pipe, int ticket);
/* Complete everything submitted so far; reports deferred device-side errors (LT_ERR_BAD_INDEX). */
int lt_hostpipe_flush(lt_hostpipe* pipe);
int lt_hostpipe_destroy(lt_hostpipe* pipe);

/* Page-locked host memory (hipHostMalloc) for callers that want their arrays DMA-able without staging. */
int lt_host_alloc(void** p, size_t bytes);
int lt_host_free(void* p);

/* ---- split API: build once, trace many, device-resident buffers ----------------------------- */

/* Create an empty scene on HIP device `device` (-1 = current device).  Replaces the per-call
 * `vector<Object*> objects` + `BVH bvh(&objects)` of RayTracer.cpp:22, :54. */
int lt_scene_create(lt_scene** scene, int device);

/* Attach a mesh given as DEVICE pointers (borrowed: they must stay valid until the last trace of
 * this mesh has completed).  Replaces the triangle de-indexing loop RayTracer.cpp:32-51. */
int lt_scene_set_mesh_dev(lt_scene* scene, const float* verts, const int* faces, const int* colors,
                          const float* rem, int n_verts, int n_faces);

/* Same from HOST pointers: the scene copies the mesh into device buffers it owns. */
int lt_scene_set_mesh_host(lt_scene* scene, const float* verts, const int* faces, const int* colors,
                           const float* rem, int n_verts, int n_faces, void* stream);

/* Build the linear BVH (Morton codes -> radix sort -> Karras topology -> child boxes) on `stream`.
 * Asynchronous unless `stats` is non-NULL (then it synchronises and fills the ms_* fields).
 * Replaces BVH::BVH / BVH::build (auxiliary/raytracer/BVH.cpp:116-126, :143-243). */
int lt_scene_build(lt_scene* scene, void* stream, lt_stats* stats);

/* Cast n_rays rays (DEVICE pointer, [n_rays,3] f32, normalised inside like Vector3.h:73-89) from
 * `origin` (HOST pointer to 3 floats) against the built scene on `stream`; outputs are DEVICE
 * pointers, any of them may be NULL.  Asynchronous unless `stats` is non-NULL.
 * Replaces the OpenMP ray loop RayTracer.cpp:62-92 incl. BVH::getIntersection (BVH.cpp:19-110). */
int lt_scene_trace_dev(lt_scene* scene, const float* rays, const float* origin, int n_rays, int height,
                       float* endpoints, int* endcolors, float* range, float* endrem, int* tri,
                       unsigned flags, void* stream, lt_stats* stats);

/* ---- single-origin fast path: ray set + triangle scatter ------------------------------------------ */

typedef struct lt_rayset lt_rayset; /* opaque: normalised directions of one ray batch, binned by direction */

/* Prepare a ray batch (DEVICE pointer rays[n_rays,3] f32) for lt_scene_render_dev: normalise the
 * directions exactly as the trace kernel does (Vector3.h:73-89; flags & LT_TRACE_NORM_EXACT selects
 * the seed) and bin them by azimuth x elevation.  A sensor model's rays do not change from scan to
 * scan (create_rays, laserscan.py:1092-1119, depends only on the YAML), so one rayset serves a whole
 * sequence -- it is read-only once created (the state of a render lives in the scene), so ONE rayset can be used by
 * any number of scenes and streams at the same time.  The call returns when the ray set is ready (it synchronises
 * `stream` once -- a ray set is built once per sensor model); `rays` is not referenced afterwards. */
int lt_rayset_create_dev(lt_rayset** rayset, const float* rays, int n_rays, int height, unsigned flags,
                         void* stream);
int lt_rayset_destroy(lt_rayset* rayset);

/* Closest hit of every ray of `rayset`, all cast from `origin` (HOST pointer to 3 floats), against the
 * scene's CURRENT mesh -- no BVH needed: the triangles are streamed once, each visits only the ray
 * bins inside its angular bounds and merges hits by atomic min over (t, face).  Same outputs, flags and
 * bit-identical results as lt_scene_build + lt_scene_trace_dev; replaces BVH::build + the ray loop
 * (BVH.cpp:143-243, RayTracer.cpp:62-92) for the reference's only call pattern, one origin per call
 * (RayTracer.cpp:58).  A scene renders one scan at a time (its mesh and its z-min image are per scene); the
 * rayset may be shared. */
int lt_scene_render_dev(lt_scene* scene, lt_rayset* rayset, const float* origin, float* endpoints,
                        int* endcolors, float* range, float* endrem, int* tri, unsigned flags, void* stream,
                        lt_stats* stats);

/* The same for up to 8 scans in ONE call -- n_scans (scene, rayset) pairs, the scenes distinct and each with
 * its own current mesh, the raysets normally all the same one (one sensor model: its bin grid then stays in
 * every L2); origins[3 * i ..] is scan i's origin (HOST), the output arguments are HOST arrays
 * of n_scans DEVICE pointers (an array, or single entries of it, may be NULL).  The three kernels of the
 * scatter strategy are launched once for the whole batch instead of once per scan: with tens of thousands of
 * scans per second, the time a hardware queue spends between small kernels is what limits throughput
 * (DESIGN.md section 9).  Results are those of n_scans separate lt_scene_render_dev calls.  Asynchronous;
 * LT_TRACE_COUNT is not accepted. */
int lt_scene_render_batch_dev(int n_scans, lt_scene* const* scenes, lt_rayset* const* raysets,
                              const float* origins, float* const* endpoints, int* const* endcolors,
                              float* const* range, float* const* endrem, int* const* tri, unsigned flags,
                              void* stream);

/* Measurement hook: record the caller's two hipEvent_t (passed as void*) on the launch stream immediately
 * before and after the dominant kernel (k_sc_tris / k_trace4) of the NEXT lt_scene_render_dev /
 * lt_scene_trace_dev call; one shot, no synchronisation.  bench.py uses it for the roofline figure. */
int lt_scene_set_probe(lt_scene* scene, void* ev_start, void* ev_stop);

/* Synchronise the scene's last stream and report deferred device-side errors (LT_ERR_BAD_INDEX). */
int lt_scene_status(lt_scene* scene);

/* Free the workspace.  Replaces the delete loop RayTracer.cpp:110-113 and BVH::~BVH (BVH.cpp:112-114). */
int lt_scene_destroy(lt_scene* scene);

/* ---- scan model: rays and spherical projection ------------------------------------------------ */

/* Unit ray direction per (beam, azimuth) cell into a DEVICE buffer rays[H*W*3] (f32, row-major
 * h*W + w).  Replaces MultiSemLaserScan.create_rays (auxiliary/laserscan.py:1092-1119), quirks
 * included: linspace(0, 360, W) holds both end points, beam_angles are ignored, float64
 * trigonometry is cast to float32 last.  fov_up / fov_down in degrees. */
int lt_create_rays_dev(double fov_up, double fov_down, int H, int W, float* rays, void* stream);

#define LT_PROJ_REMOVE 1u /* `remove=True`: drop depth == 0 and points whose proj_y is outside [0, 1]   */
#define LT_PROJ_NEW 2u    /* do_range_projection_new: depth == 0 is always dropped (laserscan.py:306-309) */

/*
 * lt_range_projection_dev -- point cloud -> H x W spherical image, closest point per cell (atomic
 * z-min, lowest point index among equal depths).  Replaces LaserScan.do_range_projection
 * (auxiliary/laserscan.py:202-292), do_range_projection_new(method="depth") (:294-391) and
 * SemLaserScan.do_label_projection[_new] (:645-649, :672-676).
 *
 *   points      [n,3] f32 (is_f64 = 0) or f64 (is_f64 = 1) -- arithmetic is done in that dtype, as
 *               numpy does for the array the reference holds; rem [n] f32 and label [n] u32 may be NULL
 *   fov_up/down degrees (down negative); beam_angles: HOST array of n_beams radians or NULL
 *   color_lut   DEVICE [lut_len,3] f32 or NULL (SemLaserScan.color_lut)
 *   *_kept      per-point outputs COMPACTED to the points that survive the removals, in input
 *               order (what remove_points leaves in self.points / remissions / label, plus depth
 *               = unproj_range, integer and float pixel coordinates); capacity n; any may be NULL
 *   *_img       [H*W] images: idx = index into the compacted arrays (-1 empty), range, xyz [H*W,3],
 *               remission, label, color [H*W,3], mask = (idx > 0) as the reference computes it;
 *               empty cells receive range_init / rem_init / xyz_init (reference: -1 for the old
 *               variant, 0 / -1 for the new one), label 0, color 0
 *   n_kept      HOST int: number of surviving points (the call synchronises `stream`)
 * All array pointers are DEVICE pointers except beam_angles and n_kept.
 */
int lt_range_projection_dev(const void* points, int is_f64, const float* rem, const unsigned* label, int n,
                            double fov_up, double fov_down, int H, int W, const double* beam_angles,
                            int n_beams, unsigned flags, const float* color_lut, int lut_len,
                            void* points_kept, float* rem_kept, unsigned* label_kept, void* depth_kept,
                            int* proj_x_kept, int* proj_y_kept, void* proj_xf_kept, void* proj_yf_kept,
                            int* idx_img, float* range_img, float* xyz_img, float* rem_img, int* label_img,
                            float* color_img, float* mask_img, float range_init, float rem_init,
                            float xyz_init, int* n_kept, void* stream);

/* Same with HOST pointers throughout (uploads, runs lt_range_projection_dev, downloads). */
int lt_range_projection(const void* points, int is_f64, const float* rem, const unsigned* label, int n,
                        double fov_up, double fov_down, int H, int W, const double* beam_angles, int n_beams,
                        unsigned flags, const float* color_lut, int lut_len, void* points_kept,
                        float* rem_kept, unsigned* label_kept, void* depth_kept, int* proj_x_kept,
                        int* proj_y_kept, void* proj_xf_kept, void* proj_yf_kept, int* idx_img,
                        float* range_img, float* xyz_img, float* rem_img, int* label_img, float* color_img,
                        float* mask_img, float range_init, float rem_init, float xyz_init, int* n_kept);

/*
 * lt_range_projection_batch_dev -- the same projection for SEVERAL clouds in one launch sequence, nothing read back by the
 * host, no memsets between calls: the `number_of_scans` observations of one output scan (MultiSemLaserScan.deform,
 * auxiliary/laserscan.py:874-881: do_range_projection_new + do_label_projection_new per source scan; :841-843 for the merged
 * cloud of the `cp` adaption).  Same per-point arithmetic and the same winner per cell as lt_range_projection_dev; the
 * per-point COMPACTED outputs of that call are not produced -- what the callers downstream consume are images:
 *
 *   lt_projector   owns the z-min workspace (one per caller thread / HIP stream; calls on one projector are serialised)
 *   clouds[k]      DEVICE pointers of cloud k: points [n,3] f32 / f64 (is_f64, one dtype per call), rem [n] f32 or NULL,
 *                  label [n] u32 or NULL
 *   out[k]         [H*W] DEVICE images of cloud k, every member may be NULL:
 *                    idx            index of the winning point among the KEPT points (remove_points numbering), -1 empty
 *                    range, xyz [H*W,3], rem, label, color [H*W,3], mask    as lt_range_projection_dev
 *                    label_folded   float32 floor(label * 256 * 256): the colour image TSDFVolume.integrate folds from
 *                                   proj_label3 (laserscan.py:893-895, fusion_lidar.py:260-264) -- feed it to
 *                                   lt_tsdf_integrate_dev / lt_fusion_scan_dev as color_im
 *                    proj_x, proj_y int32 pixel of the winner; proj_xf, proj_yf its unclamped coordinates in the points'
 *                                   dtype (do_range_projection_new's proj_x / proj_y / proj_x_float / proj_y_float,
 *                                   laserscan.py:384-388; an EMPTY cell holds the values of the last kept point, numpy's
 *                                   index -1)
 *                    n_kept         DEVICE int: number of points that survived the removals
 * Asynchronous on `stream`; more than 8 clouds are processed in groups of 8.
 */
typedef struct lt_projector lt_projector;
typedef struct lt_cloud {
  const void* points;
  const float* rem;
  const unsigned* label;
  int n;
} lt_cloud;
typedef struct lt_proj_images {
  int* idx;
  float* range;
  float* xyz;
  float* rem;
  int* label;
  float* color;
  float* mask;
  float* label_folded;
  int* proj_x;
  int* proj_y;
  void* proj_xf;
  void* proj_yf;
  int* n_kept;
  double* bnds; /* [6] {xmin, xmax, ymin, ymax, zmin, zmax} of the KEPT points: SemLaserScan.get_bnds() after the projection's
                 * remove_points (laserscan.py:678-681, read by deform('mergemesh') :957); (+inf, -inf) when none is kept */
} lt_proj_images;
int lt_projector_create(lt_projector** projector, int device);
int lt_projector_destroy(lt_projector* projector);
int lt_range_projection_batch_dev(lt_projector* projector, int n_clouds, const lt_cloud* clouds, int is_f64,
                                  double fov_up, double fov_down, int H, int W, const double* beam_angles, int n_beams,
                                  unsigned flags, const float* color_lut, int lut_len, const lt_proj_images* out,
                                  float range_init, float rem_init, float xyz_init, void* stream);

/* ---- before the render: class-aware TSDF fusion of range images (device-resident volumes) -------- */

typedef struct lt_tsdf lt_tsdf; /* opaque: one (tsdf, weight, colour, rem) float32 record per voxel, [dim_x][dim_y][dim_z] */

#define LT_TSDF_MERGE 1u /* class-aware update, the branch the reference runs (fusion_lidar.py:177, :191-228) */
/* the reference's numpy branch of `integrate` (FUSION_GPU_MODE == 0: what it runs without pycuda, fusion_lidar.py:290-388):
 * float64 voxel projection, plain running average (float64 sum and quotient), per-channel colour average with
 * round-half-even, remissions NOT integrated.  Excludes LT_TSDF_MERGE (the numpy branch has no class-aware update). */
#define LT_TSDF_HOST_MODE 2u

/* vol_bnds = {xmin, xmax, ymin, ymax, zmin, zmax} (metres), fov in degrees.  Replaces TSDFVolume.__init__
 * (auxiliary/fusion_lidar.py:23-63): dims = ceil(extent / voxel_size), truncation = 5 voxels,
 * tsdf = 1, weight = colour = remission = 0. */
int lt_tsdf_create(lt_tsdf** vol, const double* vol_bnds, double voxel_size, double fov_up, double fov_down,
                   int device);
/* Back to the initial state (the reference builds a new TSDFVolume per output scan, laserscan.py:886-887); only the
 * voxel columns written since the last reset are re-initialised. */
int lt_tsdf_reset(lt_tsdf* vol, void* stream);
/* Integrate one spherical observation: color_im = labels folded into one float per pixel as
 * fusion_lidar.py:262-264, depth_im, rem_im -- DEVICE pointers [im_h * im_w] f32.  Replaces the pycuda kernel
 * `integrate` (fusion_lidar.py:66-229) and its launch loop (:267-287); like that kernel it ignores cam_pose. */
int lt_tsdf_integrate_dev(lt_tsdf* vol, const float* color_im, const float* depth_im, const float* rem_im,
                          int im_h, int im_w, float obs_weight, unsigned flags, void* stream);
/* `n_obs` observations in order -- the `number_of_scans` range images the reference's `mesh` adaption fuses into ONE new
 * volume, all re-projected into the primary pose (laserscan.py:874-897).  Same result, bit for bit, as n_obs calls of
 * lt_tsdf_integrate_dev; on a volume that holds no observation yet (after lt_tsdf_reset) the class-aware update of up to 8
 * observations runs as ONE pass: every observation projects a voxel into the same pixel, so a voxel's geometry is
 * evaluated once and the updates are applied in order on its state in registers.  color_ims / depth_ims / rem_ims: HOST
 * arrays of n_obs DEVICE image pointers [im_h * im_w] f32 (colour folded as for lt_tsdf_integrate_dev). */
int lt_tsdf_integrate_multi_dev(lt_tsdf* vol, int n_obs, const float* const* color_ims, const float* const* depth_ims,
                                const float* const* rem_ims, int im_h, int im_w, float obs_weight, unsigned flags,
                                void* stream);

/* Device pointers of the volumes (TSDFVolume.get_volume, fusion_lidar.py:395-400, without the copies): the four fields of
 * voxel 0.  The volume is ONE array of (tsdf, weight, colour, remission) records, [x][y][z]: voxel v's field is
 * lt_tsdf_volume_stride() (= 4) floats x v behind the pointer.  (Four separate arrays until ABI 6; one record per voxel
 * makes an update one 16-byte access and hands marching cubes a vertex's samples and attributes in the lines it reads
 * anyway.) */
int lt_tsdf_volumes(lt_tsdf* vol, int* dims, float* origin, float** tsdf, float** weight, float** color,
                    float** rem);
int lt_tsdf_volume_stride(void);
/* Tell the volume that its fields were modified through the pointers above (the library tracks which (x, y) columns
 * its own integrate calls wrote, so that reset and marching cubes skip the untouched ones). */
int lt_tsdf_touch(lt_tsdf* vol);
int lt_tsdf_destroy(lt_tsdf* vol);

/* ---- between fusion and render: marching cubes on the device, the mesh is born in HBM -------------------------- */

typedef struct lt_mesh lt_mesh; /* opaque: an indexed triangle mesh in device memory + the extraction workspace */

int lt_mesh_create(lt_mesh** mesh, int device);
int lt_mesh_destroy(lt_mesh* mesh);

/* Extract the level-0 surface of the volume's current state into `mesh`.  Replaces TSDFVolume.get_mesh
 * (auxiliary/fusion_lidar.py:403-424): the three device-to-host volume copies of get_volume (:395-400),
 * skimage.measure.marching_cubes_lewiner on the CPU (:407), the vertex attribute look-ups (:409-423: nearest-voxel
 * colour / remission, voxel -> world coordinates, colour unfolding incl. the uint8 wrap of labels 256..259) and
 * the later upload of the mesh for the ray cast (:433-451).  The mesh is scikit-image 0.18's: Lewiner's cases with their
 * face / interior tests and centre vertices, one vertex per sign-changing lattice edge shared by the cells around it,
 * positions by its centre-of-mass rule -- the same vertices (bit for bit) and the same FACE STREAM as the reference's
 * get_mesh returns: face k has the same three vertices in the same order (golden F10 of the real scikit-image,
 * tests/test_pin_f10_f11_gpu.py).  Only the NUMBERING of the vertices is this library's (by word of 64 voxels, owner voxel,
 * edge axis; scikit-image numbers them by first use in its serial face stream).
 * The call synchronises `stream` once (the sizes of the mesh are needed on the host).  ms: NULL, or two floats that
 * receive the duration of the sign pass (the one stream over the float field) and of everything else. */
int lt_tsdf_extract_mesh_dev(lt_tsdf* vol, lt_mesh* mesh, void* stream, float* ms);

/* The same on caller-owned DEVICE fields [nx][ny][nz] f32 (z fastest); origin: HOST pointer to 3 floats. */
int lt_marching_cubes_dev(const float* tsdf, const float* color_vol, const float* rem_vol, int nx, int ny, int nz,
                          float voxel_size, const float* origin, lt_mesh* mesh, void* stream, float* ms);

/* Sizes and DEVICE pointers of the last extraction: verts [V,3] f32 (world), faces [F,3] i32, colors [V,3] i32
 * (r, g, b; the label in channel 2, as get_mesh + throw_rays_at_mesh hand it to ctrace), rem [V] f32.  The
 * pointers stay valid until the next extraction into this mesh or lt_mesh_destroy.  Any argument may be NULL. */
int lt_mesh_get(lt_mesh* mesh, int* n_verts, int* n_faces, float** verts, int** faces, int** colors, float** rem);

/* Number the vertices of the last extraction as scikit-image numbers them -- by first use in the face stream -- and rewrite
 * the face array accordingly: verts / colors / rem / faces then EQUAL the arrays the reference's get_mesh returns (golden
 * F10), not only up to the vertices' numbers.  For host-facing consumers of the arrays (TSDFVolume.get_mesh); the ray cast
 * does not care.  Call it BEFORE lt_scene_set_mesh / lt_mesh_get: the vertex arrays move to other buffers.  Synchronises
 * `stream`. */
int lt_mesh_renumber_dev(lt_mesh* mesh, void* stream);

/* lt_scene_set_mesh_dev with the arrays of `mesh` (borrowed until the next extraction): the render reads the
 * mesh where marching cubes wrote it -- no PCIe traffic between fusion and range image. */
int lt_scene_set_mesh(lt_scene* scene, lt_mesh* mesh);

/* One OUTPUT SCAN of the reference's `mesh` adaption in one call, nothing leaving HBM: a fresh volume (lt_tsdf_reset) <-
 * n_obs observations (lt_tsdf_integrate_dev each; color_ims / depth_ims / rem_ims are HOST arrays of n_obs DEVICE
 * pointers, images [im_h * im_w] f32) -> marching cubes into `mesh` -> lt_scene_set_mesh -> lt_scene_render_dev of
 * `rayset` from `origin` (HOST, 3 floats) into the DEVICE output images (any may be NULL), and -- with sync != 0 -- a wait
 * for `stream`.  Replaces the body of MultiSemLaserScan.deform's mesh branch per output scan (auxiliary/laserscan.py:
 * 874-914: TSDFVolume(...) :886, integrate per scan :896-899, throw_rays_at_mesh :907 = get_mesh + C_Trace).  One call
 * per scan is what lets several scans be in flight from several host threads of a Python caller (the interpreter lock is
 * released for the whole chain): lidar_transfer_amd.pipeline.FusionScanPipeline.  Returns the first error of the chain. */
int lt_fusion_scan_dev(lt_tsdf* vol, lt_mesh* mesh, lt_scene* scene, lt_rayset* rayset, int n_obs,
                       const float* const* color_ims, const float* const* depth_ims, const float* const* rem_ims,
                       int im_h, int im_w, float obs_weight, unsigned tsdf_flags, const float* origin,
                       float* endpoints, int* endcolors, float* range, float* endrem, int* tri, unsigned trace_flags,
                       void* stream, int sync);

/* The same from the POINT CLOUDS of the source scans (MultiSemLaserScan.deform's mesh branch, laserscan.py:874-914, in one
 * call): do_range_projection_new(fov, remove=True) + do_label_projection_new per cloud (lt_range_projection_batch_dev, into
 * images the projector owns) -> lt_fusion_scan_dev.  fov_up / fov_down (degrees) and H x W are the SOURCE sensor's, i.e. the
 * volume's; clouds / is_f64 / beam_angles as for lt_range_projection_batch_dev; the rest as for lt_fusion_scan_dev.  At most
 * 64 clouds. */
int lt_deform_scan_dev(lt_projector* projector, lt_tsdf* vol, lt_mesh* mesh, lt_scene* scene, lt_rayset* rayset,
                       int n_clouds, const lt_cloud* clouds, int is_f64, double fov_up, double fov_down, int H, int W,
                       const double* beam_angles, int n_beams, float obs_weight, unsigned tsdf_flags, const float* origin,
                       float* endpoints, int* endcolors, float* range, float* endrem, int* tri, unsigned trace_flags,
                       void* stream, int sync);

/* ---- deform('mergemesh'): the volume geometry as device-resident state ---------------------------------------------
 *
 * The reference's default adaption clips ONE voxel_bounds array output scan after output scan (auxiliary/laserscan.py:957-962
 * on the array lidar_deform.py:321 hands every MultiSemLaserScan; auxiliary/fusion_lidar.py:33-37 then re-derives its upper
 * bounds): a scan's volume depends on every earlier scan of the SEQUENCE.  lt_mm_state holds that array in HBM;
 * lt_mm_geometry_dev applies the five numpy statements to it in stream order, from the kept points' bounds the projection
 * left on the device (lt_proj_images.bnds), and mirrors the outcome into a pinned host record behind an event -- no
 * stream synchronisation between projection and fusion.  lt_mm_geometry_get waits for that event only (passed already
 * when the caller has read the chain's mesh sizes) and returns the record. */
typedef struct lt_mm_state lt_mm_state;
typedef struct lt_mm_geometry {
  double bnds_given[6]; /* the bounds after the clip (:961-962): what TSDFVolume(vol_bnds, ...) is constructed from */
  double bnds_after[6]; /* ... after fusion_lidar.py:36: what the reference leaves in the caller's array            */
  int dim[3];           /* ceil((max - min) / voxel_size), fusion_lidar.py:34                                         */
  int status;           /* 0 ok; 1 no point survived the projection (state untouched; numpy raises in amin);          */
                        /* 2 the clipped volume is empty (state holds the clip; the reference dies in np.ones)        */
  int ticket;           /* the call's ordinal on its state (set by lt_mm_geometry_get): later calls hold later state  */
  int reserved;
} lt_mm_geometry;
/* vol_bnds = {xmin, xmax, ymin, ymax, zmin, zmax}; bounds_are_int: the caller's array has an integer dtype (the YAML's
 * ints: fusion_lidar.py:36 then truncates). */
int lt_mm_state_create(lt_mm_state** state, const double* vol_bnds, int bounds_are_int, double voxel_size, int device);
int lt_mm_state_destroy(lt_mm_state* state);
/* New bounds for a new sequence (the reference starts one process per sequence, experiments/run_lidar_deform.sh). */
int lt_mm_state_reset(lt_mm_state* state, const double* vol_bnds, void* stream);
/* point_bnds: DEVICE [6] as lt_proj_images.bnds.  *ticket names the record.  The kernels of successive calls on one state
 * run in the order of the calls, whatever their streams (each waits for the previous call's event).  seq >= 0: the call is
 * scan `seq` of its sequence (numbered from 0 since create / reset) and waits on the HOST until scans 0 .. seq - 1 have made
 * theirs -- several chains (threads, streams) of one sequence; seq < 0: the caller keeps the order itself.  point_bnds ==
 * NULL with seq >= 0: the scan gives up its turn (it failed before its projection); *ticket = -1. */
int lt_mm_geometry_dev(lt_mm_state* state, const double* point_bnds, int seq, int* ticket, void* stream);
int lt_mm_geometry_get(lt_mm_state* state, int ticket, lt_mm_geometry* out);

/* deform('mergemesh') of ONE output scan in ONE call (auxiliary/laserscan.py:921-1012; the interpreter lock of a Python
 * caller is released for all of it): `cloud` -- the merged source scans -- through do_range_projection_new(fov, remove=True)
 * into images the projector owns (fov_up / fov_down: the TARGET's, H x W: the SOURCE's, laserscan.py:929-931, :952-954) ->
 * lt_mm_geometry_dev(mm, seq) on the kept points' bounds -> with `vol` != NULL (the volume of the geometry the caller
 * EXPECTS, i.e. the previous scan's): lt_fusion_scan_dev on it, then lt_mm_geometry_get -> *geo; *done = 1 when vol was
 * built from exactly geo->bnds_given (the images are this scan's), else 0: the caller makes the volume of geo->bnds_given
 * and calls lt_mergemesh_rerun_dev, which runs the chain on the images the projector still holds.  With vol == NULL the
 * call waits for the record and returns *done = 0.  geo->status != 0: *done = 0, nothing else to do. */
int lt_mergemesh_scan_dev(lt_projector* projector, lt_mm_state* mm, int seq, lt_tsdf* vol, lt_mesh* mesh, lt_scene* scene,
                          lt_rayset* rayset, const lt_cloud* cloud, int is_f64, double fov_up, double fov_down, int H, int W,
                          const double* beam_angles, int n_beams, float obs_weight, unsigned tsdf_flags, const float* origin,
                          float* endpoints, int* endcolors, float* range, float* endrem, int* tri, unsigned trace_flags,
                          void* stream, lt_mm_geometry* geo, int* done);
int lt_mergemesh_rerun_dev(lt_projector* projector, lt_tsdf* vol, lt_mesh* mesh, lt_scene* scene, lt_rayset* rayset, int H, int W,
                           float obs_weight, unsigned tsdf_flags, const float* origin, float* endpoints, int* endcolors,
                           float* range, float* endrem, int* tri, unsigned trace_flags, void* stream);

/* ---- after the render: back-projection, scan packing, comparison ------------------------------- */

/* xyz of every cell from its range and pixel coordinates; replaces LaserScan.do_reverse_projection_new
 * (auxiliary/laserscan.py:475-501).  range_img [H*W] f32, proj_x / proj_y [H*W] int32 (or float64 when
 * coords_are_f64: the `preserve_float` branch), back_points [H*W,3] f64 -- all DEVICE pointers. */
int lt_reverse_projection_dev(const float* range_img, const void* proj_x, const void* proj_y,
                              int coords_are_f64, double fov_up, double fov_down, int H, int W,
                              double* back_points, void* stream);

/* Pack a rendered scan in SemanticKITTI layout; replaces the filtering and the per-point struct.pack loops
 * of MultiSemLaserScan.write (auxiliary/laserscan.py:1133-1178).  Keeps, in order, the cells with
 * (index == NULL or index[i] > 0) and label[i] >= 0 and x + y + z != 0.  points [n,3] f32 or f64,
 * rem [n] f32, label [n] i32, out_bin [n,4] f32 (x, y, z, remission), out_label [n] u32 -- DEVICE
 * pointers; *n_out (HOST) = number of points written (the call synchronises `stream`). */
int lt_pack_scan_dev(const void* points, int is_f64, const float* rem, const int* label, const int* index, int n,
                     float* out_bin, unsigned* out_label, int* n_out, void* stream);

/* Masked comparison of a source and a target image; replaces the array part of compare()
 * (auxiliary/laserscan.py:1181-1301) and iouEval.addBatch (auxiliary/np_ioueval.py:31-47): cells whose
 * source colour is black or whose source label is 0 are background in both images;
 * conf[target * n_labels + source] counts raw label pairs (u64, n_labels^2); range_diff / rem_diff are the
 * squared differences [n] (may be NULL); src_masked / tgt_masked the masked label images (may be NULL);
 * *sq_sum (DEVICE double) = sum of range_diff (MSE = sq_sum / n).  All DEVICE pointers, asynchronous. */
int lt_compare_dev(const int* src_label, const float* src_color, const int* tgt_label, const float* src_range,
                   const float* tgt_range, const float* src_rem, const float* tgt_rem, int n, int n_labels,
                   unsigned long long* conf, float* range_diff, float* rem_diff, int* src_masked, int* tgt_masked,
                   double* sq_sum, void* stream);

/* ---- misc ------------------------------------------------------------------------------------ */

/* Message of the last error raised on the calling thread ("" if none). */
const char* lt_last_error(void);

/* Library version string, e.g. "lidarhip 0.1 (gfx950)". */
const char* lt_version(void);

/* Layout version of the structs this header declares (lt_proj_images, lt_stats, lt_mm_geometry ...): a caller compiled
 * against another header must not pass them.  lt_abi_version() returns the library's; the Python binding compares at load. */
#define LT_ABI_VERSION 7
int lt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LIDARHIP_H */
