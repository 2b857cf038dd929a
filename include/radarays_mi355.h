/*
 * radarays_mi355.h -- C ABI of libradarays_mi355.so
 *
 * MI355X (gfx950) implementation of ONE path of uos/radarays_ros: the
 * per-azimuth multi-bounce radar ray loop of RadarCPU::simulate
 * (src/radarays_ros/RadarCPU.cpp:155-548), behind the reference's own seam
 *     virtual sensor_msgs::ImagePtr Radar::simulate(ros::Time)      (include/radarays_ros/Radar.hpp:64)
 * The reference has no FFI: backends are C++ subclasses of `Radar` chosen at
 * start-up (src/radar_simulator.cpp:118-176).  A third subclass `RadarHIP`
 *  marshals the protected state `simulate()` reads
 * (Radar.hpp:66-105) into the calls below -- plain pointers and sizes only.
 *
 * Threading: one rr_ctx is used by one thread at a time; one ctx per GPU.  Several GPUs of one node behind ONE
 * object: rr_multi (below) -- one process, one ctx per device, one RCCL collective per call.
 * Errors: every call returns 0 on success, <0 on error; rr_last_error() gives
 * the text.  No exceptions cross this boundary.  There is NO CPU fallback: if
 * no HIP device is usable rr_create() fails.
 */
#ifndef RADARAYS_MI355_H
#define RADARAYS_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RR_ABI_VERSION 6
#define RR_MAX_BATCH 64   /* frames (poses or material sets) one call renders in one set of launches */

typedef struct rr_ctx rr_ctx;

/* msg/RadarMaterial.msg:1-4 -- m_params.materials.data[] (Radar.hpp:84) */
typedef struct rr_material {
    float velocity;   /* m/ns; 0.3 = air; 0 = nothing is transmitted */
    float ambient;    /* A in  E * (A + B * cos(theta)^C)  (RadarCPU.cpp:310-316) */
    float diffuse;    /* B */
    float specular;   /* C */
} rr_material;

/* The RadarModelConfig fields (cfg/RadarModel.cfg:11-85) and RadarModel fields
 * (msg/RadarModel.msg:1-3) that RadarCPU::simulate reads, plus the constants
 * Radar::Radar fixes (src/radarays_ros/Radar.cpp:22-32). */
typedef struct rr_config {
    int32_t n_cells;                 /* RadarModel.cfg:16   rows of the polar image */
    int32_t n_angles;                /* Radar.cpp:29        400 azimuth columns */
    int32_t n_reflections;           /* RadarModel.cfg:29   number of ray-cast passes (RadarCPU.cpp:220) */
    int32_t signal_denoising;        /* RadarModel.cfg:44   0 none 1 triangular 2 gaussian 3 maxwell-boltzmann */
    int32_t signal_denoising_triangular_width;   /* :46 */
    int32_t signal_denoising_gaussian_width;     /* :48 */
    int32_t signal_denoising_mb_width;           /* :50 */
    int32_t ambient_noise;           /* RadarModel.cfg:60   0 none 1 uniform 2 perlin */
    int32_t scroll_image;            /* :81 */
    int32_t record_multi_reflection; /* :83 */
    int32_t record_multi_path;       /* :82 */
    int32_t max_waves_per_azimuth;   /* build's own: capacity of the per-azimuth wave queue per pass;
                                        0 = n_samples * 2^(n_reflections-1) clamped to 65536 */
    int32_t brdf_model;              /* build's own: 0 = the checkout's A + B cos^C (radar_algorithms.h:168-187);
                                        1 = Cook-Torrance lobe, A + B * D_GGX * G_Smith normalised to 1 at normal
                                        incidence, alpha^2 = 2 / (C + 2) (BASELINE.json configs[4]; the reference's own
                                        version lives on its dev/flex branch, outside the checkout: PARITY UNPINNED) */
    int32_t reserved_;
    double  resolution;              /* :15  m per range bin */
    double  energy_max;              /* :32 */
    double  signal_max;              /* :33 */
    double  signal_denoising_triangular_mode;    /* :47 */
    double  signal_denoising_gaussian_mode;      /* :49 */
    double  signal_denoising_mb_mode;            /* :51 */
    double  ambient_noise_at_signal_0;           /* :61 */
    double  ambient_noise_at_signal_1;           /* :62 */
    double  ambient_noise_energy_max;            /* :63 */
    double  ambient_noise_energy_min;            /* :64 */
    double  ambient_noise_energy_loss;           /* :65 */
    double  multipath_threshold;                 /* :84 */
    float   wave_energy_threshold;   /* Radar.cpp:24  0.001 */
    float   theta_min;               /* Radar.cpp:28  0 */
    float   theta_inc;               /* Radar.cpp:27  -(2 pi)/400 */
    float   range_max;               /* radar_algorithms.cpp:158  1000 (ray tfar) */
} rr_config;

/* Counters of the last simulated frame (optional; reading them synchronises). */
typedef struct rr_stats {
    uint64_t wave_passes;     /* waves ray-cast, all passes */
    uint64_t hits;
    uint64_t signals;
    uint64_t nodes_visited;   /* only counted when the ctx was put in stats mode */
    uint64_t tris_tested;
    uint32_t overflow;        /* 1: wave queue capacity exceeded (frame invalid) */
    uint32_t pad_;
} rr_stats;

/* Fills *cfg with the defaults of cfg/RadarModel.cfg + Radar.cpp:22-32. */
void rr_default_config(rr_config* cfg);

/* Context bound to HIP device `device` (what the RadarCPU/RadarGPU constructor
 * does with its map handle, RadarCPU.hpp:21-28).  NULL on failure. */
rr_ctx* rr_create(int device);
void    rr_destroy(rr_ctx* ctx);
const char* rr_last_error(const rr_ctx* ctx);   /* ctx may be NULL: create error */
int     rr_abi_version(void);

/* Replaces rm::import_embree_map (src/radar_simulator.cpp:149): triangle soup
 * + per-face object id (index into object_materials; NULL -> all 0).  Builds
 * the BVH on the host (SAH over references with spatial splits: a face larger than its neighbours may be cut and
 * then has one triangle record per leaf that holds a part of it) and uploads it.  For meshes of up to 2M triangles
 * the tree is CHOSEN by measurement: the plain SAH tree (no spatial splits) is built as well, both trace the same sample
 * of radar-like rays on the GPU, the one with fewer traversal steps stays (RR_BVH_CHOOSE).  Inputs are copied.  Size limit:
 * 8 x BVH4 nodes + 3 x triangle records < 2^28 (child references are 28-bit offsets), i.e. about 50M triangles; a
 * mesh whose split parts would exceed it is built without spatial splits. */
int rr_set_mesh(rr_ctx* ctx, const float* verts /*[nv][3]*/, size_t nv,
                const uint32_t* faces /*[nf][3]*/, size_t nf,
                const uint32_t* face_object_id /*[nf] or NULL*/);

/* Same contract, but the BVH is built ON THE GPU (early split clipping of oversized faces + Morton codes + rocprim
 * radix sort + Karras radix tree + refit + 4-wide collapse): 0.35 s instead of 1.9 s for 10M triangles, tree of
 * lower quality (rays traverse about 1.2x slower than through the host builder's SAH tree with spatial splits).
 * Images are bit-identical whichever builder made the tree: the nearest hit is defined independently of
 * traversal order. */
int rr_set_mesh_gpu(rr_ctx* ctx, const float* verts /*[nv][3]*/, size_t nv,
                    const uint32_t* faces /*[nf][3]*/, size_t nf,
                    const uint32_t* face_object_id /*[nf] or NULL*/);

/* The finished tree of `src` (either builder), copied device to device into `ctx` -- `ctx` may sit on another GPU
 * (hipMemcpyPeer over xGMI) or on the same one.  What rr_multi_set_mesh uses to replicate the map after ONE build
 * (radar_simulator.cpp:149 loads the map once per process).  Both contexts are drained first; `src` keeps its tree. */
int rr_copy_mesh(rr_ctx* ctx, rr_ctx* src);

/* Radar::loadParams (Radar.cpp:220-226): materials, object_materials,
 * material_id_air. */
int rr_set_materials(rr_ctx* ctx, const rr_material* materials, size_t n_materials,
                     const int32_t* object_materials, size_t n_objects,
                     int32_t material_id_air);

/* Radar::updateDynCfg (Radar.cpp:188-218). */
int rr_set_config(rr_ctx* ctx, const rr_config* cfg);

/* m_waves_start (RadarCPU.cpp:136-145): beam sample directions in the local
 * azimuth frame, as sample_cone_local (radar_algorithms.cpp:248-294) returns
 * them.  The reference draws them from std::random_device, so they are an
 * input here. */
int rr_set_beam_samples(rr_ctx* ctx, const float* dirs /*[n][3]*/, size_t n);

/* per-azimuth `random_begin` of the ambient-noise stage (RadarCPU.cpp:472);
 * [n_angles].  Needed only when ambient_noise != 0.  The reference draws fresh offsets for every frame
 * (RadarCPU.cpp:461-472): a caller of the single-frame entry points sets a new row before each frame; for
 * the batch entry points n may be k * n_angles (k >= 2 rows), frame f of a batch then uses row f % k. */
int rr_set_noise_offsets(rr_ctx* ctx, const float* rnd, size_t n);

/* include_motion = true (RadarCPU.cpp:190-196, cfg/RadarModel.cfg:85 -- the .cfg default): the reference looks
 * Tsm up once PER AZIMUTH.  poses = [n_angles][7] (qx,qy,qz,qw,tx,ty,tz); while set, every
 * rr_simulate* call uses poses[azimuth] and ignores its own pose argument (which must still
 * be a valid pose).  n = 0 switches back to one pose per frame.  For the batch entry points n may be
 * k * n_angles (k tables, one sweep of the antenna each): frame f of a batch then uses table f % k, like the
 * rows of rr_set_noise_offsets -- the reference's default mode through the batched / multi-GPU path.  A pose batch of
 * more than one frame while exactly ONE table is set is refused (-3): every frame would be the same sweep and the call's
 * poses would be ignored silently.  (A parameter batch renders every set with table 0.) */
int rr_set_motion_poses(rr_ctx* ctx, const float* poses, size_t n);

/* RadarCPU::simulate for azimuths [az_begin, az_end) with sensor pose
 * Tsm = {quaternion x,y,z,w ; translation x,y,z} (Radar::updateTsm,
 * Radar.cpp:80-132).  Host buffers, synchronous:
 *   out_u8  [n_cells][n_angles] row-major, step n_angles  == the mono8
 *           sensor_msgs::Image of RadarCPU.cpp:555-561; only the columns of
 *           the simulated azimuths are written.
 *   out_f32 optional, same layout: the float slice before convertTo(CV_8U).
 *   stats   optional. */
int rr_simulate(rr_ctx* ctx, const float pose_qxyzw_t[7], int az_begin, int az_end,
                uint8_t* out_u8, float* out_f32, rr_stats* stats);

/* Same, asynchronous on `stream` (a hipStream_t; NULL = the ctx's own stream)
 * with DEVICE buffers.  d_cols_u8 receives the simulated columns column-major:
 * [az_end-az_begin][n_cells] (this is the block a rank contributes to the
 * multi-GPU gather).  d_cols_f32 optional, same layout. */
int rr_simulate_columns_device(rr_ctx* ctx, const float pose_qxyzw_t[7], int az_begin, int az_end,
                               uint8_t* d_cols_u8, float* d_cols_f32, void* stream);

/* Frame batch (multi-GPU weak scaling, offline generation): the same azimuth block
 * [az_begin, az_end) of n_frames (1..RR_MAX_BATCH) different poses in ONE set of launches.
 * poses = [n_frames][7]; d_cols_u8 = [n_frames][az_end-az_begin][n_cells].  Kernels then see
 * n_frames x block segments, i.e. a rank that owns 1/N of the azimuths of N frames does the
 * same amount of work per launch as a single GPU does for one whole frame. */
int rr_simulate_batch_columns_device(rr_ctx* ctx, const float* poses, int n_frames, int az_begin, int az_end,
                                     uint8_t* d_cols_u8, void* stream);

/* The same, carrying a device -> host copy of the CALLER's: carry_bytes from d_carry_src (device memory written by earlier
 * work on `stream`, e.g. the images this rank assembled from its previous batch on this stream) to h_carry_dst (page-locked:
 * rr_host_alloc) travel on this batch's later-pass trace launches -- a few waves with one 1-KB store in flight each, the
 * route rr_simulate_batch_host_async uses, which keeps the stores of the running kernels from queueing behind a bulk
 * copy.  The bytes are complete once `stream` has passed this call's launches.  With a single ray-cast pass, a pageable
 * destination or sizes that are not multiples of 16 the copy is a plain hipMemcpyAsync ahead of the batch.  This is how
 * the sharded step loop (one process per GPU, radarays_ros_amd/dist.py) leaves every frame in HOST memory like the
 * reference's simulate() does (RadarCPU.cpp:542,555-561) without paying a copy per batch. */
int rr_simulate_batch_columns_carry_device(rr_ctx* ctx, const float* poses, int n_frames, int az_begin, int az_end,
                                           uint8_t* d_cols_u8, void* stream,
                                           const void* d_carry_src, void* h_carry_dst, size_t carry_bytes);

/* Whole frames of n_frames (1..RR_MAX_BATCH) poses in one set of launches, everything on `stream` (no internal
 * streams: callers that want several batches in flight issue them on several streams, 4 is the measured
 * optimum): d_imgs_u8 = [n_frames][n_cells][n_angles].  The throughput entry point for offline generation
 * from C/C++ (tools/cpp_bench.cpp: 41k images/s at config 2 with 4 poses per call on 4 streams). */
int rr_simulate_batch_device(rr_ctx* ctx, const float* poses, int n_frames, uint8_t* d_imgs_u8, void* stream);

/* The reference leaves every frame in HOST memory (m_polar_image -> sensor_msgs::Image, RadarCPU.cpp:542,555-561).
 * Whole frames of n_frames poses like rr_simulate_batch_device, delivered to the caller's host buffer
 * h_imgs_u8 = [n_frames][n_cells][n_angles].  Returns at once; the images are COMPLETE ONLY after rr_wait_host(ctx,
 * h_imgs_u8) (NULL: every outstanding buffer) or rr_synchronize() -- until then the buffer must stay valid and must not be
 * read.  How the bytes travel is the library's business.  By default they leave at once over the SDMA engines, submitted
 * through ROCr by worker threads of the context behind the batch's last kernel (csrc/rr_sdma.cpp; two image buffers per
 * frame lane, so the host does not wait for a copy before it issues the lane's next batch): no shader core stores a byte of
 * them, and it is the same engine whichever HIP runtime serves the process (a ROCm 7.0.2 runtime, e.g. the one a Python ML
 * wheel bundles, would carry a hipMemcpyAsync as a blit kernel: 27-35k images/s on config 2 where SDMA delivers the link's
 * 39k).  Where that path is not available (RR_HOST_SDMA=0, a pageable buffer, no reachable ROCr) a batch's images wait in
 * device memory and ride out on the trace launches of the next batch that uses the same frame lane (a few waves trickle them over PCIe with one store in
 * flight each, which keeps the stores of the running kernels from queueing behind them: within 1 % of the rate with the
 * images left in HBM, where a plain copy behind each batch costs 7 %); rr_wait_host / rr_synchronize / any other use of
 * the lane send what is still waiting with a plain copy.  Issue batches on up to four streams (HIP maps streams onto
 * four hardware queues) and hand the buffers out from a ring twice as deep as the batches in flight.  h_imgs_u8
 * should be page-locked (rr_host_alloc / rr_host_free = hipHostMalloc); a pageable buffer works through the plain copy.
 * rr_destroy drops images that nobody waited for.
 * `stream` must stay valid until rr_wait_host() / rr_synchronize() has returned for this buffer: a copy that did not
 * ride on a later batch is issued on it then.  "Returns at once" has two exceptions: a lane whose two previous
 * deliveries are both still in flight makes the call wait for the older one, and so does a buffer reallocation. */
int rr_simulate_batch_host_async(rr_ctx* ctx, const float* poses, int n_frames, uint8_t* h_imgs_u8, void* stream);
int rr_wait_host(rr_ctx* ctx, const void* h_imgs_u8);
void* rr_host_alloc(size_t bytes);
void  rr_host_free(void* p);

/* bytes of device memory -> host memory, asynchronous on `stream`, by the library's own copy kernel (one-wave workgroups,
 * 1 KB per wave and store, a bounded number of stores outstanding per wave) when h_dst is page-locked and both pointers
 * and the size are multiples of 16 -- else a plain hipMemcpyAsync.  Why an entry point: which engine carries a
 * hipMemcpyAsync to page-locked memory is the choice of the HIP runtime in the caller's process (a ROCm 7.0.2 runtime
 * launches a blit kernel per copy and reads 27-36k images/s on config 2, the image's own ROCm 7.2 uses SDMA: 39.4k);
 * the library's deliveries that must be STREAM-ordered take this route (the fallback copies of rr_simulate_batch_host_async
 * where SDMA is not available); measured, a shader-core copy is no faster than the runtime's blit kernel (24-31k on config 2)
 * -- callers that can fence with rr_wait_host use rr_deliver_to_host_async below.  Replaces nothing in the reference
 * (cv_bridge deep-copies m_polar_image on the host, RadarCPU.cpp:555-558). */
int rr_copy_to_host_async(rr_ctx* ctx, const void* d_src, void* h_dst, size_t bytes, void* stream);
/* The same for a caller that can fence with rr_wait_host instead of the stream: bytes of a caller-owned device buffer leave
 * over the SDMA engines (the route of rr_simulate_batch_host_async, csrc/rr_sdma.cpp) once the work enqueued on `stream` so
 * far has completed.  Returns at once; h_dst is complete -- and d_src may be overwritten -- after rr_wait_host(ctx, h_dst)
 * (NULL: everything outstanding) or rr_synchronize().  Where the SDMA path is not available it is rr_copy_to_host_async plus
 * an event.  rr_multi's root and the sharded step loop's flush deliver this way. */
int rr_deliver_to_host_async(rr_ctx* ctx, const void* d_src, void* h_dst, size_t bytes, void* stream);
/* Which route the host deliveries of this context take: 2 = SDMA through ROCr (csrc/rr_sdma.cpp) is in use; 1 = it will be
 * tried by the first delivery; 0 = stream-ordered copies (RR_HOST_SDMA=0, or the path was not available / was switched off --
 * RR_HOST_SDMA_VERBOSE=1 says why).  bench.py prints it on its line, the GPU tests assert it. */
int rr_host_delivery_route(rr_ctx* ctx);

/* Assemble the mono8 image from column-major columns, applying scroll_image
 * (RadarCPU.cpp:457): d_img[c][(scroll + a) % n_angles] = d_cols[a][c].
 * Device buffers, asynchronous on `stream`. */
int rr_assemble_image_device(rr_ctx* ctx, const uint8_t* d_cols_u8 /*[n_angles][n_cells]*/,
                             uint8_t* d_img_u8 /*[n_cells][n_angles]*/, void* stream);

/* Same for columns that arrive in blocks of n_loc azimuths `block_stride` BYTES apart (what a rank
 * holds after the all_to_all of a multi-frame step: [source rank][frame][n_loc][n_cells]):
 * azimuth a is read at d_cols + (a / n_loc) * block_stride + (a % n_loc) * n_cells. */
int rr_assemble_blocks_device(rr_ctx* ctx, const uint8_t* d_cols_u8, int n_loc, size_t block_stride,
                              uint8_t* d_img_u8, void* stream);

/* Parameter batch: n_sets (1..RR_MAX_BATCH) material tables, ONE pose -> n_sets images.
 * Replaces n_sets round trips of the reference's optimisation loop, where every objective
 * evaluation sends one RadarParams goal to the gen_radar_image action and waits for one image
 * (action/GenRadarImage.action, scripts/radaray_opti.py:170-200; the server side sets
 * m_params.materials and calls simulate(), Radar.hpp:52-53).  `sets` is [n_sets][n_materials]
 * with n_materials as given to rr_set_materials (object -> material map, air id, config and beam
 * samples stay as set).  Pass 0 does not depend on the materials and is traced once for all sets;
 * image k is bit-identical to rr_set_materials(sets[k]) + rr_simulate_device(pose).
 * d_imgs_u8: [n_sets][n_cells][n_angles] in HBM, stream-ordered like rr_simulate_device. */
int rr_simulate_material_sets_device(rr_ctx* ctx, const float pose[7], const rr_material* sets, int n_sets,
                                     size_t n_materials /* per set; must equal rr_set_materials' count */,
                                     uint8_t* d_imgs_u8, void* stream);
/* Same with a host output buffer (synchronous). */
int rr_simulate_material_sets(rr_ctx* ctx, const float pose[7], const rr_material* sets, int n_sets,
                              size_t n_materials, uint8_t* out_imgs_u8);

/* The whole parameter vector of the optimiser (scripts/radaray_opti.py:36-113: beam_width, n_reflections, 2 x 4 material
 * values) batched the same way: set k = {material table, beam sample directions, number of ray-cast passes}, ONE pose,
 * n_sets (1..RR_MAX_BATCH) images in one set of launches.
 *   materials      [n_materials] (as many as rr_set_materials got), or NULL: the table of rr_set_materials
 *   beam_dirs      [n_beam][3] with n_beam as given to rr_set_beam_samples (what sample_cone_local returns for this set's
 *                  beam_width: rr_sample_cone_local), or NULL: the samples of rr_set_beam_samples.  Sets with the SAME
 *                  directions (same pointer or same bytes) form a group: pass 0 does not depend on the materials and is
 *                  traced once per group
 *   n_reflections  0..16 passes for this set, negative: the config's.  A set with fewer passes than the others simply
 *                  stops early (no live waves in the later launches); wave queues are sized for the largest
 * Image k is bit-identical to rr_set_materials / rr_set_beam_samples / rr_set_config(n_reflections) of set k followed
 * by rr_simulate_device(pose); every set sees the same noise realisation (row 0 of rr_set_noise_offsets).  One
 * difference in ERROR behaviour: with max_waves_per_azimuth left at 0 the queues of a batch are sized for its largest
 * number of passes (and a lane that already holds larger ones is reused as it is), so a set that alone would have run
 * into the 65,536-wave clamp of its own nominal capacity (n_samples x 2^(passes-1) beyond 65,536) may render completely
 * here where the one-by-one call returns -7; a user-set max_waves_per_azimuth is honoured exactly. */
typedef struct rr_param_set {
    const rr_material* materials;
    const float* beam_dirs;
    int32_t n_reflections;
    int32_t reserved_;
} rr_param_set;
int rr_simulate_param_sets_device(rr_ctx* ctx, const float pose[7], const rr_param_set* sets, int n_sets, size_t n_materials,
                                  uint8_t* d_imgs_u8 /* [n_sets][n_cells][n_angles] in HBM */, void* stream);
/* The objective of that optimiser is ONE float per evaluation -- minus the PSNR of the simulated image against one real
 * radar image (radaray_opti.py:170-211, skimage.metrics.peak_signal_noise_ratio on mono8) -- so an evaluation need not
 * ship 1.37 MB per set to the host: rr_score_images_device returns, for n images in HBM against one reference image in
 * HBM, psnr[k] = 10 log10(255^2 / mean((img_k - ref)^2)) in f64 (+inf for identical images; the device part is the
 * exact integer sum of squared differences, optionally returned in out_sse) to HOST arrays; synchronous on `stream`. */
int rr_score_images_device(rr_ctx* ctx, const uint8_t* d_imgs_u8, int n_images, const uint8_t* d_ref_u8,
                           double* out_psnr /* host [n_images], or NULL */, uint64_t* out_sse /* host [n_images], or NULL */, void* stream);
/* Host-buffer form (synchronous) of the parameter batch: out_imgs_u8 [n_sets][n_cells][n_angles] or NULL; with
 * ref_img_u8 (host, [n_cells][n_angles]) and out_psnr ([n_sets]) the scores come back as well -- with out_imgs_u8 NULL
 * an evaluation of n_sets parameter vectors returns n_sets doubles and no image leaves the GPU. */
int rr_simulate_param_sets(rr_ctx* ctx, const float pose[7], const rr_param_set* sets, int n_sets, size_t n_materials,
                           uint8_t* out_imgs_u8, const uint8_t* ref_img_u8, double* out_psnr);

/* All frames of a multi-frame step in ONE launch: frame j reads its columns frame_stride bytes after
 * frame j-1 (block addressing as above) and writes image j of d_imgs_u8 [n_frames][n_cells][n_angles]. */
int rr_assemble_frames_device(rr_ctx* ctx, const uint8_t* d_cols_u8, int n_loc, size_t block_stride,
                              int n_frames, size_t frame_stride, uint8_t* d_imgs_u8, void* stream);

/* Convenience: rr_simulate_columns_device for all azimuths into the ctx's own
 * column buffer + rr_assemble_image_device into d_img_u8.  Asynchronous. */
int rr_simulate_device(rr_ctx* ctx, const float pose_qxyzw_t[7], uint8_t* d_img_u8, void* stream);

/* Blocks until `stream` (NULL = ctx stream) and the ctx's own frame lanes -- and, because batches may
 * have been issued on further caller streams, everything else on the device -- are idle, then reports what
 * the asynchronous (*_device) entry points could not: -7 if any frame enqueued since the last call
 * exceeded its wave/signal queue capacity, -8 if one met an object/material id outside the tables
 * (such a frame is truncated; the synchronous rr_simulate returns the same codes itself). */
int rr_synchronize(rr_ctx* ctx, void* stream);

/* Pipelined callers that must not drain the device to learn about an error: enqueues on `stream` (the one the LAST
 * *_device / *_host_async call ran on) a 4-byte copy of the error bits of the frame lane that call used (bit 0: wave /
 * signal queue overflow, bit 1: object / material id outside the tables; accumulated since the last rr_synchronize, which
 * reports and clears them) into *h_bits.  h_bits should be page-locked (rr_host_alloc); it is valid once `stream` has
 * reached this point.  rr_multi uses it per batch. */
int rr_peek_error_bits_async(rr_ctx* ctx, uint32_t* h_bits, void* stream);

/* Counters of the last frame (synchronises the ctx stream). */
int rr_get_stats(rr_ctx* ctx, rr_stats* stats);

/* stats mode: traversal counters (nodes_visited, tris_tested) on/off; off by
 * default because the counting kernel variant is slower. */
int rr_set_stats_mode(rr_ctx* ctx, int enable);
/* stats mode only: the wave-level shape of the traversal loop over the last frame (all k_trace launches), what
 * separates the kernel's instruction ISSUE rate from useful work (bench.py: roofline.useful_issue_frac).  A wave holds
 * 16 rays (one per quad of lanes) and iterates until its slowest ray is done; an iteration issues the node path if any
 * quad holds a node and the leaf path if any holds a leaf.
 *   out[0] waves   out[1] wave iterations   out[2] iterations that issued the node path   out[3] ... the leaf path
 *   out[4] live quad-steps (ray steps that did work; 16 x out[1] were issued)   out[5] longest wave (iterations)
 *   out[6] node steps of all rays (= nodes_visited)   out[7] leaf steps of all rays (= out[4] - out[6]) */
int rr_get_traversal_shape(rr_ctx* ctx, uint64_t out[8]);

/* ---- introspection used by tests / bench ---- */
/* nearest-hit query for rays given in map coordinates (device traversal). */
int rr_debug_trace(rr_ctx* ctx, const float* origs /*[n][3]*/, const float* dirs /*[n][3]*/, size_t n,
                   float* out_t /*[n], <0 = miss*/, uint32_t* out_face /*[n]*/);
/* BVH facts: nodes, leaf triangle records (>= faces: the host builder may cut a face by spatial splits and keeps
 * one record per part, at most twice the faces), depth, stack entries needed. */
/* Test hook: the kernels' own Fresnel split (k_shade's fresnel_split = radar_algorithms.h:55-139; the incidence angle by acosf of
 * the f32 dot product as k_shade forms it, the angle of total reflection by the function that fills the material table) on n
 * independent inputs: v1 = the wave's velocity (0.3 in the frame path, RadarCPU.cpp:107-110), v2 = the material's.  Host arrays
 * in, host arrays out; tests compare them with the oracle on the reference-derived cases of tests/golden/pyref_dense_*. */
int rr_debug_fresnel(rr_ctx* ctx, size_t n, const float* normals /*[n][3]*/, const float* dirs /*[n][3]*/, const double* energy /*[n]*/,
                     const double* v1 /*[n]*/, const float* v2 /*[n]*/,
                     float* out_refl_dir /*[n][3]*/, double* out_refl_energy /*[n]*/, float* out_refr_dir /*[n][3]*/, double* out_refr_energy /*[n]*/);
/* Test hook: the kernels' own back_reflection_shader (radar_algorithms.h:168-187 as RadarCPU.cpp:310-316,347-353 call it) on n
 * independent inputs, in5 = [n][5] (incidence angle, energy, ambient, diffuse, specular); brdf_model as in rr_config.  Tests compare
 * it with the oracle on the cases of tests/golden/pyref_brdf.npy (outputs of the reference's scripts/radarays_snell_fresnel_brdf.py). */
int rr_debug_brdf(rr_ctx* ctx, size_t n, const float* in5 /*[n][5]*/, int brdf_model, float* out /*[n]*/);
int rr_get_bvh_info(rr_ctx* ctx, uint64_t* n_nodes, uint64_t* n_tris, uint32_t* depth, uint32_t* stack_need);
/* How the later-pass trace launches are sized (round 5).  A segment holds at most n_beam * 2^pass waves in pass `pass`;
 * instead of a row of 16-ray workgroups up to that bound per segment, a row is as long as earlier batches of this context
 * needed (the largest per-segment count seen per pass, + 1/16 + 32 rays), and a segment that exceeds its row anyway is
 * finished by a small repair launch -- images never depend on the history.  Synchronises the device.
 *   out_rows[p]  workgroups per segment row of pass p in the last call's launches (0: the doubling bound)
 *   out_hist[p]  the largest per-segment wave count seen in pass p since mesh / materials / beam / config last changed
 *   *repaired_groups  16-ray groups the repair launches had to trace since then (0 once the history has settled) */
int rr_get_trace_grid(rr_ctx* ctx, uint32_t out_rows[24], uint32_t out_hist[24], uint64_t* repaired_groups);
/* Launch graphs (round 5): the launch chain of a pose batch (rr_simulate_batch_*_device, rr_simulate_columns_device,
 * rr_multi's device entries) that has been issued before with the same shape -- azimuth block, frames, output buffer,
 * trace rows -- is captured in a hipGraph on its second use and replayed from then on: one hipGraphLaunch instead of
 * 4..20 kernel launches (host time per chain 46 -> ~11 us at 4 passes); the poses of a replay travel as the parameters of the
 * graph's first node.  Chains that carry a host copy, parameter batches and instrumented runs (timing / statistics /
 * roctx) are issued kernel by kernel.  Any setter, mesh change or buffer reallocation drops the captured graphs.
 * Returns how many chains this context has captured / replayed. */
int rr_get_graph_stats(rr_ctx* ctx, uint64_t* captures, uint64_t* replays);
/* average duration (ms) of the trace kernel launches since the last call with
 * reset!=0, measured with hipEvents on the launch stream when timing mode is
 * on; also returns the number of launches.  Used by bench.py for roofline. */
int rr_set_timing_mode(rr_ctx* ctx, int enable /* 0 off, 1 every kernel, 2 k_trace only */);
int rr_get_kernel_time(rr_ctx* ctx, const char* kernel /* "trace0"|"trace"|"trace_repair"|"shade"|"scan"|"column"|"assemble";
                                                            "trace" = the later-pass launches WITHOUT the k_trace_repair launch that
                                                            follows a tightened row, which is "trace_repair" */,
                       double* total_ms, uint64_t* launches, int reset);

/* every launch duration (ms) recorded for `kernel` since the last reset ("trace0" = pass 0, "trace" = later
 * passes, ...): *n_out = count, the first min(count, capacity) values go to out_ms (may be NULL). */
int rr_get_kernel_samples(rr_ctx* ctx, const char* kernel, float* out_ms, size_t capacity, size_t* n_out);
/* pre-creates n timing events so that a timed region never calls hipEventCreate */
int rr_reserve_timing_events(rr_ctx* ctx, size_t n);

/* ---- several GPUs of one node behind one object (SURVEY.md §8b "Threading", §8e) -------------------------
 * The reference creates ONE backend object per process (src/radar_simulator.cpp:145-176) and fans out inside it
 * (OpenMP over azimuths, RadarCPU.cpp:155).  rr_multi is that object for n GPUs: one rr_ctx per device, mesh and
 * parameters replicated, device i renders the contiguous azimuth block rr_partition(n_angles, n, i) of every frame
 * of a call in one set of launches, ONE RCCL collective per call over xGMI gathers the blocks on device 0 (one
 * group of send / recv pairs to the root: one piece per device for equal blocks, one per device and frame for ragged
 * ones; no other device receives anything), which transposes them into the mono8 images and copies them to the
 * caller's host buffer.  With one device no collective runs and the images are byte-identical to rr_simulate's.
 * RCCL (librccl.so.1) is loaded at run time when n > 1. */
/* (Test switch RR_MULTI_LOOPBACK=1: a device may be listed several times; the collective is then replaced by
 * device-to-device copies along the same plan -- the n > 1 path on a one-GPU box, minus the RCCL calls.) */
typedef struct rr_multi rr_multi;
rr_multi* rr_create_multi(const int* devices, int n_devices);    /* NULL on failure: rr_multi_last_error(NULL) */
void rr_destroy_multi(rr_multi* m);
const char* rr_multi_last_error(const rr_multi* m);
int rr_multi_device_count(const rr_multi* m);
/* the NCCL version code of the RCCL library this object's communicator was made with (ncclGetVersion: 2.27.7 -> 22707); 0: no
 * communicator (one device, loopback) or a library that does not say.  rr_create_multi refuses a library outside [2.7, 3.0)
 * -- the prototypes it calls through are declared by hand -- and, before the first frame, has every rank send 16 bytes to rank
 * 0 with guard bytes behind them (the constant taken for ncclUint8 must move exactly 16). */
int rr_multi_rccl_version(const rr_multi* m);
rr_ctx* rr_multi_ctx(rr_multi* m, int i);                        /* the context of device i (stats, tuning) */
/* azimuth block [*begin, *end) of `rank` among `world`: contiguous, sizes differ by at most one column */
void rr_partition(int n_angles, int world, int rank, int* begin, int* end);
/* the data plan of one rr_multi_simulate_batch call (pure arithmetic, no GPU): whether the blocks are equal (all-gather
 * of bytes_per_device per device) and, for the ragged case, for device r and frame f (index r * n_frames + f) the byte
 * offset of the piece in r's block buffer, its offset in the root's [n_frames][n_angles][n_cells] buffer and its size */
int rr_multi_plan(int n_angles, int n_cells, int n_devices, int n_frames, int* equal_blocks, size_t* bytes_per_device,
                  size_t* send_off, size_t* recv_off, size_t* piece_bytes);
/* replicated setters: same contracts as the rr_set_* calls above, applied to every device */
/* rr_multi_set_mesh / _gpu: the tree is built ONCE (on the host / on device 0) and copied to the other devices (rr_copy_mesh) */
int rr_multi_set_mesh(rr_multi* m, const float* verts, size_t nv, const uint32_t* faces, size_t nf, const uint32_t* face_object_id);
int rr_multi_set_mesh_gpu(rr_multi* m, const float* verts, size_t nv, const uint32_t* faces, size_t nf, const uint32_t* face_object_id);
int rr_multi_set_materials(rr_multi* m, const rr_material* materials, size_t n_materials,
                           const int32_t* object_materials, size_t n_objects, int32_t material_id_air);
int rr_multi_set_config(rr_multi* m, const rr_config* cfg);
int rr_multi_set_beam_samples(rr_multi* m, const float* dirs, size_t n);
int rr_multi_set_noise_offsets(rr_multi* m, const float* rnd, size_t n);
int rr_multi_set_motion_poses(rr_multi* m, const float* poses, size_t n);
/* Radar::simulate on all devices: one frame / n_frames (1..RR_MAX_BATCH) frames, host output
 * [n_frames][n_cells][n_angles], synchronous; -7 / -8 like rr_simulate when a device reports an overflow / bad id */
int rr_multi_simulate(rr_multi* m, const float pose_qxyzw_t[7], uint8_t* out_u8);
int rr_multi_simulate_batch(rr_multi* m, const float* poses, int n_frames, uint8_t* out_imgs_u8);
/* Pipelined form (what a node that streams frames calls, radar_simulator.cpp:197-212): enqueues the batch and returns;
 * the images are complete only after rr_multi_wait(m, h_imgs_u8) (NULL: every batch in flight), which also reports a
 * -7 / -8 of that batch.  Up to RR_MULTI_SLOTS (default 4, 1..8) batches are in flight -- the render of batch k+1
 * overlaps the collective, the transpose and the D2H copy of batch k; a call that finds its slot still busy waits for
 * that older batch first.  h_imgs_u8 should be page-locked (rr_host_alloc) and handed out from a ring at least as deep
 * as the slots; it must stay valid and unread until waited for.  After any error return nothing of the object is in
 * flight any more (every device drained, error bits cleared): the caller may free its buffers.  AN ERROR INVALIDATES EVERY
 * BATCH IN FLIGHT: error bits are kept per frame lane, not per batch, and the drain reads and clears them all, so the other
 * batches that were in flight report the same code from the rr_multi_wait for their own buffer (or from the call that next
 * uses their slot), whatever their images look like; rr_multi_wait(m, NULL) reports the error once for all of them.
 * With ONE device a batch takes rr_simulate_batch_host_async's route (deferred, trickled host copy). */
int rr_multi_simulate_batch_async(rr_multi* m, const float* poses, int n_frames, uint8_t* h_imgs_u8);
int rr_multi_wait(rr_multi* m, const void* h_imgs_u8);

/* ---- host side of the seam: what the reference does on the CPU around simulate() (no GPU involved) -----------------
 * Beam samples = sample_cone_local (src/radarays_ros/radar_algorithms.cpp:248-294; erfinvf radar_math.h:13-44):
 * RadarCPU::simulate re-draws m_waves_start whenever a dynamic reconfigure changed beam_width / n_samples /
 * beam_sample_dist / p_in_cone (RadarCPU.cpp:136-145).  rr_cone_dirs is the geometry on caller-supplied variates
 * (u_angle in [0, 1) -> angle, r_variate: U(0,1) for sample_dist 0 / 1, N(0,1) for 2 / 3), bit-equal to the oracle's
 * restatement; rr_sample_cone_local draws the variates itself from a SEEDED MT19937 (the reference seeds from
 * std::random_device: unreproducible) in numpy.random.RandomState's streams, so that it returns the directions
 * radarays_ros_amd/beams.py: sample_cone_local(seed) returns.  width_rad = RadarModel.beam_width (radians);
 * out_dirs [n][3], local azimuth frame, ready for rr_set_beam_samples. */
int rr_cone_dirs(float width_rad, int sample_dist, float p_in_cone, const float* u_angle, const float* r_variate, size_t n,
                 float* out_dirs);
int rr_sample_cone_local(uint32_t seed, float width_rad, size_t n, int sample_dist, float p_in_cone, float* out_dirs);

/* The map file, as rm::import_embree_map(map_file) reads it for the node (src/radar_simulator.cpp:149): PLY (ascii,
 * binary little / big endian; MulRan maps, launch/mulran_sim.launch:7), Wavefront OBJ (objects `o` / `g` become
 * object ids, the index into object_materials) and COLLADA .dae (the reference's default map, launch/mro_husky.launch:4;
 * csrc/rr_collada.cpp: one object per instantiated geometry / primitive group, depth-first in scene order, node
 * transforms and <unit meter> applied, the up axis left as modelled) into the flat arrays rr_set_mesh takes; polygons
 * are fan-triangulated.  The arrays are malloc'ed: give them back with rr_free_mesh.  <0 + text in err on failure. */
typedef struct rr_mesh {
    float* verts;               /* [n_verts][3] */
    size_t n_verts;
    uint32_t* faces;            /* [n_faces][3] */
    size_t n_faces;
    uint32_t* face_object_id;   /* [n_faces] */
    size_t n_objects;           /* OBJ: number of o / g groups, DAE: instantiated primitive groups (>= 1) */
    char** object_names;        /* [n_objects] NUL-terminated names (OBJ group / DAE geometry names), or NULL (PLY):
                                   what to match a scene's material table against */
} rr_mesh;
int rr_load_mesh_file(const char* path, rr_mesh* out, char* err, size_t err_len);
/* Object numbering.  The reference indexes its material table by the object id rmagine's importer hands out
 * (m_object_materials[obj_id], RadarCPU.cpp:268; rm::import_embree_map, radar_simulator.cpp:149; the per-object lists of
 * config/oru4_test.yaml:37-56).  rr_load_mesh_file numbers objects in DEPTH-FIRST SCENE ORDER (OBJ: order of the o / g
 * groups; DAE: instantiated geometries as the visual scene is walked) -- this build's specification; whether rmagine / assimp
 * number a given file the same way cannot be checked without them.  A scene whose material list was written for another
 * numbering is put right here: object `order[k]` becomes id k, objects not listed keep their relative order behind the listed
 * ones; only face_object_id and the order of object_names change.  Unknown, duplicate or ambiguous names are refused. */
int rr_mesh_reorder_objects(rr_mesh* m, const char* const* order, size_t n_order, char* err, size_t err_len);
void rr_free_mesh(rr_mesh* m);

/* ---- environment switches (read at rr_create / at a build; none is needed in normal use) --------------------------
 * RR_LANES (4)            frame buffer sets = batches that can be in flight (1..8)
 * RR_STREAM_LANES (3)     lanes whose own stream rr_simulate_device rotates over
 * RR_STACKLESS (0)        1: k_trace walks the tree WITHOUT a stack (parent links, a node re-fetched each time the walk returns to it;
 *                         no LDS) -- the traversal north_star names, built and measured in round 6: same images, slower
 * RR_COPY_BLOCKS (8)      one-wave workgroups of a trace launch that trickle a deferred host copy; 0: never fold
 * RR_HOST_SDMA (1)         rr_simulate_batch_host_async hands a batch's images to the SDMA engines through ROCr (hsa_amd_memory_async_copy,
 *                         a worker thread per context; page-locked destinations) -- the same engine under every HIP runtime; 0: the
 *                         deferred copies below (trickled out by the next batch's trace launches / the copy kernel).  RR_HOST_SDMA_VERBOSE=1
 *                         says on stderr why the path was not available or was switched off
 * RR_SDMA_ACTIVE_US (0)    ... a worker waits this long actively for a copy's completion signal before it sleeps on it (measured: no
 *                         gain with two workers, the other one has the next copy queued already)
 * RR_HOST_COPY_STREAM (0) 1: one-pass frames (nothing later could carry their images) are copied out at once on one dedicated stream
 * RR_FLUSH_KERNEL (1)     a host copy that does not ride on a trace launch (one-pass frames, the end of a run, rr_copy_to_host_async)
 *                         is stored by the library's own kernel when the destination is page-locked; 0: hipMemcpyAsync
 * RR_FLUSH_BLOCKS (8)     ... its workgroups
 * RR_FLUSH_INFLIGHT (0)   ... 1-KB stores a wave keeps outstanding (0: no limit)
 * RR_FLUSH_THREADS (256)  ... threads per workgroup (64..1024)
 * RR_FLUSH_XCD (0)        ... the XCD all of them run on (PCIe-paced stores then fill the write queues of one XCD only); -1: all eight
 * RR_FOLD_MIN_BUSY (2)    other lanes that must be busy for a host copy to be folded into the next batch
 * RR_CULL_POP (1)         later passes drop stack entries at pop time by their distance bound; 0: off (same images)
 * RR_GRAPH_GUARD (1)      launch graphs: two execs per shape used alternately, the host waits for an exec's previous launch before it re-sets
 *                         its poses; 0 (probe): one exec, no wait -- relies on in-flight launches being unaffected by node updates
 * RR_GRAPHS (1)           launch chains of pose batches captured and replayed as hipGraphs (rr_get_graph_stats); 0: kernel by kernel
 * RR_TIGHT_GRID (1)       later-pass trace rows sized by the history of earlier batches (rr_get_trace_grid); 0: the doubling bound
 * RR_TRACE_CHUNK (16)     later-pass trace launches walk chunks of S neighbouring azimuths with the azimuth as the fast grid dimension;
 *                         0: one grid row per azimuth (same images)
 * RR_TIGHT_FORCE (0)      n > 0: rows of n workgroups whatever the history says (tests: nearly every ray goes through the repair launch)
 * RR_STACK_LDS (64)       traversal stack entries kept in LDS (lower: exercises the HBM spill path)
 * RR_PASS0_AZ (16)        neighbouring azimuths per pass-0 wave (1, 2, 4, 8, 16)
 * RR_ROCTX (0)            1: roctx ranges around the kernel enqueues (rocprofv3 --marker-trace)
 * RR_TRACE_STATS          set: rr_get_stats prints the wave-level loop shape of the statistics build
 * RR_BVH_THREADS, RR_BVH_VERBOSE, RR_BVH_ALPHA / _BETA / _BUDGET / _WZ   host builder: threads, phase times, and the
 *                         BvhOptions (csrc/rr_bvh.h) for experiments;  RR_LBVH_NO_SPLIT: GPU builder without split clipping
 * RR_BVH_CHOOSE (1)       host builder, meshes up to 2M triangles: build the candidates (SAH with spatial splits; plain SAH with
 *                         vertical weight 0.5 / 1.0), trace one sample of radar-like rays through each, keep the tree with
 *                         the fewest traversal steps; 0: the default tree only (images are the same whichever tree)
 * RR_MULTI_LOOPBACK (0)   1: rr_create_multi accepts one device several times (tests, see above)
 * RR_MULTI_SELF_RCCL (0)  1 with ONE device: its block travels to itself through the real RCCL calls (one-rank communicator, a
 *                         group of ncclSend / ncclRecv to self; 2: one pair per frame, the ragged plan) instead of the
 *                         single-device route (tests on one-GPU boxes)
 * RR_MULTI_THREADS (0)    1: rr_multi with several devices starts one enqueue thread per device, which issues that device's
 *                         launches of a call while the others issue theirs (off by default: on one physical device, the
 *                         only case measurable on a one-GPU box, the runtime serialises the threads and nothing is gained)
 * RR_MULTI_SLOTS (4)      batches rr_multi keeps in flight (streams + buffer sets per device, 1..8)
 * RR_HOST_PROFILE (0)     1: where the HOST time of the batch calls goes (named scopes in rr_ctx / rr_multi, csrc/rr_hostprof.h);
 *                         a table on stderr when the process ends (read it with RR_MULTI_THREADS=0) */

#ifdef __cplusplus
}
#endif
#endif
