/*
 * radarays_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's per-azimuth radar hot loop,
 * uos/radarays_ros  src/radarays_ros/RadarCPU.cpp:30-564  and the per-hit math
 * in include/radarays_ros/radar_algorithms.h, radar_types.h, radar_math.h,
 * image_algorithms.h.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (radarays_ros_amd/, libradarays_mi355.so)
 * never links, imports or calls it.
 *
 * PARITY STATUS
 *   per-hit math (fresnel, BRDF, smear kernels, Perlin, erfinvf):
 *       pinned to the known-answer values SURVEY.md §8c captured from the
 *       reference's compiled C++ (tests/golden/survey_kat.json), to the
 *       reference's importable python scripts (tests/golden/pyref_*.json,
 *       pyref_dense_*.npy: 11,000 Fresnel / Snell cases, pyref_brdf.npy: 3,624
 *       cases of the BRDF lobe) and,
 *       for erfinvf/quantile, to oracle/_ref (the reference's own
 *       radar_math.h compiled as-is).
 *   loop glue (RadarCPU.cpp:156-548) and the ray cast (rmagine/Embree, not in
 *       /root/reference): PARITY UNPINNED -- the reference holds no test,
 *       fixture or golden image for them and cannot be built here (needs ROS,
 *       OpenCV, rmagine >= 2.2.1, Embree).  They are restated line by line
 *       with the quirks of SURVEY.md Appendix A.
 */
#ifndef RADARAYS_ORACLE_H
#define RADARAYS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* msg/RadarMaterial.msg:1-4 */
typedef struct {
    float velocity;
    float ambient;
    float diffuse;
    float specular;
} orc_material;

/* the RadarModelConfig fields RadarCPU::simulate reads (cfg/RadarModel.cfg)
 * plus RadarModel {beam_width, n_samples, n_reflections} and the constants of
 * Radar.cpp:22-32 */
typedef struct {
    int32_t n_cells;                 /* RadarModel.cfg:16 */
    int32_t n_angles;                /* Radar.cpp:29 (400) */
    int32_t n_reflections;           /* number of ray-cast passes, RadarCPU.cpp:220 */
    int32_t signal_denoising;        /* 0 none, 1 triangular, 2 "gaussian", 3 maxwell-boltzmann */
    int32_t signal_denoising_triangular_width;
    int32_t signal_denoising_gaussian_width;
    int32_t signal_denoising_mb_width;
    int32_t ambient_noise;           /* 0 none, 1 uniform, 2 perlin */
    int32_t scroll_image;
    int32_t record_multi_reflection;
    int32_t record_multi_path;
    int32_t material_id_air;         /* Radar.cpp:23 */
    double  resolution;              /* dynamic_reconfigure double_t */
    double  energy_max;
    double  signal_max;
    double  signal_denoising_triangular_mode;
    double  signal_denoising_gaussian_mode;
    double  signal_denoising_mb_mode;
    double  ambient_noise_at_signal_0;
    double  ambient_noise_at_signal_1;
    double  ambient_noise_energy_max;
    double  ambient_noise_energy_min;
    double  ambient_noise_energy_loss;
    double  multipath_threshold;
    float   wave_energy_threshold;   /* Radar.cpp:24 (0.001) */
    float   theta_min;               /* Radar.cpp:28 */
    float   theta_inc;               /* Radar.cpp:27  -(2 pi)/400 */
    int32_t brdf_model;              /* NOT a reference field: 0 = radar_algorithms.h:168-187, 1 = orc_ct_lobe (see there) */
} orc_config;

typedef struct {
    uint64_t wave_passes;   /* waves ray-cast (all passes, all azimuths) */
    uint64_t hits;          /* of which hit something */
    uint64_t signals;       /* Signal records emitted */
    uint64_t nodes_visited; /* BVH nodes popped (0 for brute force) */
    uint64_t tris_tested;
    double   seconds;       /* stopwatch bracket of RadarCPU.cpp:147-148 -> :550 */
    uint64_t near_threshold;/* test bookkeeping: waves whose reflected / refracted energy lies within 1e-6 of the pruning threshold */
} orc_stats;

typedef struct orc_scene orc_scene;

/* scene = triangle soup + per-face object id (rmagine geometry/instance id).
 * use_bvh: 0 brute force, 1 BVH2, -1 auto (BVH when nf > 4096). */
orc_scene* orc_scene_create(const float* verts, size_t nv,
                            const uint32_t* faces, size_t nf,
                            const uint32_t* face_object_id, int use_bvh);
void orc_scene_destroy(orc_scene*);

/* nearest hit of ONE ray given in map coordinates (RadarCPU.cpp:236 -> rmagine
 * OnDnSimulatorEmbree): returns 1 on hit; t = range, tri = face index,
 * ng = un-normalised geometric normal cross(v1-v0, v2-v0). */
int orc_intersect(const orc_scene*, const float orig[3], const float dir[3],
                  float* t, uint32_t* tri, float ng[3]);

/* RadarCPU::simulate, azimuths [az_begin, az_end).
 *   pose_qxyzw_t : Tsm (sensor->map) as quaternion x,y,z,w + translation
 *   beam_dirs    : m_waves_start directions, local frame, [n_beam][3]
 *   noise_rnd    : per-azimuth `random_begin` (RadarCPU.cpp:472), [n_angles] or NULL
 *   out_u8       : [n_cells][n_angles] row-major (mono8, step n_angles)
 *   out_f32      : optional, same layout, the float slice before convertTo
 *   n_threads    : OpenMP threads over azimuths (RadarCPU.cpp:155); <=0 -> all */
int orc_simulate(const orc_scene*,
                 const orc_material* materials, size_t n_materials,
                 const int32_t* object_materials, size_t n_objects,
                 const orc_config* cfg,
                 const float* beam_dirs, size_t n_beam,
                 const float pose_qxyzw_t[7],
                 const float* noise_rnd,
                 int az_begin, int az_end,
                 uint8_t* out_u8, float* out_f32,
                 int n_threads, orc_stats* stats);

/* include_motion = true (RadarCPU.cpp:190-196): poses[n_angles][7], one Tsm per azimuth. */
int orc_simulate_motion(const orc_scene*,
                 const orc_material* materials, size_t n_materials,
                 const int32_t* object_materials, size_t n_objects,
                 const orc_config* cfg,
                 const float* beam_dirs, size_t n_beam,
                 const float* poses,
                 const float* noise_rnd,
                 int az_begin, int az_end,
                 uint8_t* out_u8, float* out_f32,
                 int n_threads, orc_stats* stats);

/* ---- per-hit math, exported one by one for the known-answer tests ---- */

/* radar_algorithms.h:55-139.  in: normal, incidence dir, energy, polarization,
 * v1 (incidence.velocity), v2.  out: reflection/refraction dir + energy. */
void orc_fresnel(const float normal[3], const float dir[3],
                 double energy, double polarization, double v1, double v2,
                 float refl_dir[3], double* refl_energy,
                 float refr_dir[3], double* refr_energy);
/* radar_algorithms.h:168-187 */
float orc_ct_lobe(float angle, float specular_exp);
float orc_back_reflection_shader_model(float incidence_angle, float energy,
                                       float diffuse, float specular_fac, float specular_exp, int model);
float orc_back_reflection_shader(float incidence_angle, float energy,
                                 float diffuse, float specular_fac, float specular_exp);
/* radar_algorithms.h:25-31 */
double orc_incidence_angle(const float normal[3], const float dir[3]);
/* radar_algorithms.h:283-351; kind 1 triangular, 2 gaussian, 3 maxwell-boltzmann.
 * rescale!=0 additionally applies RadarCPU.cpp:83-91 (w[mode] := 1). */
void orc_make_denoiser(int kind, int width, int mode, int rescale, float* out);
/* image_algorithms.h:69-106 */
double orc_perlin_noise(double x, double y, double z);
/* image_algorithms.h:108-128 */
double orc_perlin_noise_hilo(double off_x, double off_y, double x, double y,
                             double scale_low, double scale_high, double p_low);
/* radar_math.h:13-49 */
float orc_erfinvf(float a);
float orc_quantile(float p);
/* radar_types.h:108-113: orig += dir*d ; time += d/velocity */
void orc_wave_move(float orig[3], const float dir[3], double* time, double velocity, double distance);
/* radar_algorithms.cpp:248-294 with the uniform/normal variates supplied by the
 * caller (the reference draws them from std::random_device): u_angle[i] in
 * [0,1), r_variate[i] = U(0,1) for dist 0/1, N(0,1) for dist 2/3.
 * width in radians. out: [n][3]. */
float orc_noise_amplitude(float signal, float max_val, double at_signal_0, double at_signal_1);
float orc_cone_radius(float width, int sample_dist, float p_in_cone, float variate);
void orc_sample_cone_local(float width, int n_samples, int sample_dist, float p_in_cone,
                           const float* u_angle, const float* r_variate, float* out_dirs);
/* cv::Mat::convertTo(CV_8UC1) on one float: saturate_cast<uchar>(cvRound(x)) */
uint8_t orc_saturate_u8(float x);

#ifdef __cplusplus
}
#endif
#endif
