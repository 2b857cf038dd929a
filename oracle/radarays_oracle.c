/*
 * radarays_oracle.c -- TEST INFRASTRUCTURE ONLY (see radarays_oracle.h).
 *
 * Plain-C restatement of uos/radarays_ros' CPU path.  Every function cites the
 * reference lines it follows (paths relative to /root/reference).  Arithmetic
 * keeps the reference's types op by op: rmagine::Vector is 3 x f32, wave
 * energy/time are f64, the slice is f32.  The reference is built without
 * -march flags (CMakeLists.txt:4-5), i.e. no FMA contraction: this file must be
 * compiled with -ffp-contract=off.
 *
 * The ray cast itself (rmagine OnDnSimulatorEmbree -> Embree rtcIntersect1) is
 * NOT in the reference tree.  Assumed semantics (public rmagine 2.2.x API,
 * recalled, not verifiable offline -- "parity unpinned"):
 *   orig_m = Tam * o, dir_m = Tam.R * d, tnear = 0, tfar = model.range.max,
 *   range = t of the nearest hit, normal = geometric normal, normalised,
 *   rotated back by Tam^-1 and flipped to oppose the ray, object id = per-face
 *   geometry id, miss -> object id UINT_MAX.
 * Nearest hit is restated with Moeller-Trumbore in f32 (no FMA), ties on t
 * broken by the lower face index, so that the result does not depend on the
 * traversal order (brute force == BVH, bit for bit).
 */
#include "radarays_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* rmagine math types (Vector3_<float>, Quaternion_<float>, Transform)       */
/* ------------------------------------------------------------------------ */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } quat;

static inline v3 v3_add(v3 a, v3 b) { v3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static inline v3 v3_sub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline v3 v3_neg(v3 a) { v3 r = { -a.x, -a.y, -a.z }; return r; }
static inline v3 v3_scale(v3 a, float s) { v3 r = { a.x * s, a.y * s, a.z * s }; return r; }
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b)
{
    v3 r = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
    return r;
}
static inline float v3_l2norm(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline v3 v3_normalize(v3 a)
{
    const float d = v3_l2norm(a);
    v3 r = { a.x / d, a.y / d, a.z / d };
    return r;
}

/* rmagine Quaternion::mult(Quaternion) (Hamilton product) */
static inline quat q_mul(quat a, quat b)
{
    quat r;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    return r;
}
static inline quat q_inv(quat a) { quat r = { -a.x, -a.y, -a.z, a.w }; return r; }
/* rmagine Quaternion::mult(Vector): (q * (v,0) * q^-1).xyz */
static inline v3 q_rot(quat q, v3 v)
{
    const quat p = { v.x, v.y, v.z, 0.0f };
    const quat pt = q_mul(q_mul(q, p), q_inv(q));
    v3 r = { pt.x, pt.y, pt.z };
    return r;
}
/* rmagine Quaternion::set(EulerAngles) (ZYX) -- RadarCPU.cpp:202,
 * radar_algorithms.cpp:285-289 */
static inline quat q_from_euler(float roll, float pitch, float yaw)
{
    const float cr = cosf(roll / 2.0f), sr = sinf(roll / 2.0f);
    const float cp = cosf(pitch / 2.0f), sp = sinf(pitch / 2.0f);
    const float cy = cosf(yaw / 2.0f), sy = sinf(yaw / 2.0f);
    quat q;
    q.w = cr * cp * cy + sr * sp * sy;
    q.x = sr * cp * cy - cr * sp * sy;
    q.y = cr * sp * cy + sr * cp * sy;
    q.z = cr * cp * sy - sr * sp * cy;
    return q;
}

/* ------------------------------------------------------------------------ */
/* radar_math.h                                                              */
/* ------------------------------------------------------------------------ */
/* radar_math.h:13-44 */
float orc_erfinvf(float a)
{
    float p, r, t;
    t = fmaf(a, 0.0f - a, 1.0f);
    t = logf(t);
    if (fabsf(t) > 6.125f) {
        p = 3.03697567e-10f;
        p = fmaf(p, t, 2.93243101e-8f);
        p = fmaf(p, t, 1.22150334e-6f);
        p = fmaf(p, t, 2.84108955e-5f);
        p = fmaf(p, t, 3.93552968e-4f);
        p = fmaf(p, t, 3.02698812e-3f);
        p = fmaf(p, t, 4.83185798e-3f);
        p = fmaf(p, t, -2.64646143e-1f);
        p = fmaf(p, t, 8.40016484e-1f);
    } else {
        p = 5.43877832e-9f;
        p = fmaf(p, t, 1.43285448e-7f);
        p = fmaf(p, t, 1.22774793e-6f);
        p = fmaf(p, t, 1.12963626e-7f);
        p = fmaf(p, t, -5.61530760e-5f);
        p = fmaf(p, t, -1.47697632e-4f);
        p = fmaf(p, t, 2.31468678e-3f);
        p = fmaf(p, t, 1.15392581e-2f);
        p = fmaf(p, t, -2.32015476e-1f);
        p = fmaf(p, t, 8.86226892e-1f);
    }
    r = a * p;
    return r;
}

/* radar_math.h:46-49 */
float orc_quantile(float p)
{
    return (float)(M_SQRT2 * (double)orc_erfinvf((float)(2 * (double)p - 1.0)));
}

/* ------------------------------------------------------------------------ */
/* radar_types.h                                                             */
/* ------------------------------------------------------------------------ */
typedef struct {
    v3 orig, dir;          /* Ray, radar_types.h:9-21 */
    double energy;         /* radar_types.h:69 */
    double polarization;   /* :74 */
    double velocity;       /* :76 */
    double time;           /* :81 */
    uint32_t material_id;  /* :84 */
    uint32_t dbg_parent;   /* tooling only (ORC_RAYLOG): index of the parent wave in the previous pass << 1 | 1 for a refraction */
} wave_t;

/* radar_types.h:108-113 -- Vector * double narrows the factor to float */
static inline void wave_move_inplace(wave_t* w, double distance)
{
    w->orig = v3_add(w->orig, v3_scale(w->dir, (float)distance));
    w->time += distance / w->velocity;
}

void orc_wave_move(float orig[3], const float dir[3], double* time, double velocity, double distance)
{
    wave_t w;
    w.orig.x = orig[0]; w.orig.y = orig[1]; w.orig.z = orig[2];
    w.dir.x = dir[0]; w.dir.y = dir[1]; w.dir.z = dir[2];
    w.time = *time; w.velocity = velocity;
    wave_move_inplace(&w, distance);
    orig[0] = w.orig.x; orig[1] = w.orig.y; orig[2] = w.orig.z;
    *time = w.time;
}

/* ------------------------------------------------------------------------ */
/* radar_algorithms.h                                                        */
/* ------------------------------------------------------------------------ */
/* radar_algorithms.h:25-31 */
static inline double incidence_angle_of(v3 normal, v3 dir)
{
    /* acos(float) inside namespace radarays_ros resolves to the C++ float
     * overload (libstdc++ <math.h> does `using std::acos`), i.e. acosf; the
     * f32 result is then widened.  SURVEY.md §8c's R>1 known answers for v2=0
     * (theta=45deg: R=1.000000262) are only reproduced with acosf. */
    return (double)acosf(v3_dot(v3_neg(dir), normal));
}
double orc_incidence_angle(const float normal[3], const float dir[3])
{
    v3 n = { normal[0], normal[1], normal[2] }, d = { dir[0], dir[1], dir[2] };
    return incidence_angle_of(n, d);
}

/* radar_algorithms.h:55-139.  Only .ray.dir and .energy of the two returned
 * waves are consumed by RadarCPU.cpp:285-286,364-365. */
static void fresnel_split(v3 surface_normal, v3 dir, double energy, double polarization,
                          double v1, double v2,
                          v3* refl_dir, double* refl_energy, v3* refr_dir, double* refr_energy)
{
    const double n1 = v2;   /* :62 (sic) */
    const double n2 = v1;   /* :63 */

    double incidence_angle = (double)acosf(v3_dot(v3_neg(dir), surface_normal));  /* :69, float overload */

    /* :73  dir + normal * 2.0 * (-normal).dot(dir) */
    *refl_dir = v3_add(dir, v3_scale(v3_scale(surface_normal, 2.0f),
                                     v3_dot(v3_neg(surface_normal), dir)));

    v3 t = { 0.0f, 0.0f, 0.0f };  /* :77 */

    if (n1 > 0.0) {               /* :80 */
        double n21 = n2 / n1;
        double angle_limit = 100.0;
        if (fabs(n21) <= 1.0) {
            angle_limit = asin(n21);
        }
        if (incidence_angle <= angle_limit) {
            if (v3_dot(surface_normal, dir) > 0.0f) {   /* :92 */
                surface_normal = v3_neg(surface_normal);
            }
            if (n2 > 0.0) {
                double n12 = n1 / n2;
                double c = cos(incidence_angle);
                /* :100  dir * n12 + normal * (n12*c - sqrt(1 - n12*n12*(1 - c*c))) */
                t = v3_add(v3_scale(dir, (float)n12),
                           v3_scale(surface_normal,
                                    (float)(n12 * c - sqrt(1 - n12 * n12 * (1 - c * c)))));
            }
        }
    }
    *refr_dir = t;

    double refraction_angle = (double)acosf(v3_dot(t, v3_neg(surface_normal)));  /* :106, float overload */

    double rs = 0.0, rp = 0.0;
    const double eps = 0.0001;
    if (incidence_angle + refraction_angle < eps) {            /* :112 */
        rs = (n1 - n2) / (n1 + n2);
        rp = rs;
    } else if (incidence_angle + refraction_angle > M_PI - eps) {
        rs = 1.0;
        rp = 1.0;
    } else {
        rs = -sin(incidence_angle - refraction_angle) / sin(incidence_angle + refraction_angle);
        rp = tan(incidence_angle - refraction_angle) / tan(incidence_angle + refraction_angle);
    }
    double Rs = rs * rs;
    double Rp = rp * rp;
    double Reff = polarization * Rs + (1.0 - polarization) * Rp;   /* :129 */
    double Teff = 1.0 - Reff;
    *refl_energy = Reff * energy;     /* :135 */
    *refr_energy = Teff * energy;     /* :136 */
}

void orc_fresnel(const float normal[3], const float dir[3],
                 double energy, double polarization, double v1, double v2,
                 float refl_dir[3], double* refl_energy,
                 float refr_dir[3], double* refr_energy)
{
    v3 n = { normal[0], normal[1], normal[2] }, d = { dir[0], dir[1], dir[2] }, r, t;
    fresnel_split(n, d, energy, polarization, v1, v2, &r, refl_energy, &t, refr_energy);
    refl_dir[0] = r.x; refl_dir[1] = r.y; refl_dir[2] = r.z;
    refr_dir[0] = t.x; refr_dir[1] = t.y; refr_dir[2] = t.z;
}

/* radar_algorithms.h:168-187 (float overloads of cos/pow apply in C++) */
float orc_back_reflection_shader(float incidence_angle, float energy,
                                 float diffuse, float specular_fac, float specular_exp)
{
    float IdotR = cosf(incidence_angle);
    float I_diffuse = 1.0f;
    float I_specular = powf(IdotR, specular_exp);
    float I_total = diffuse * I_diffuse + specular_fac * I_specular;
    return I_total * energy;
}

/* NOT in the reference checkout.  BASELINE.json configs[4] names the Cook-Torrance reflection model of the
 * reference's dev/flex branch (README.md:83-85), which is not part of /root/reference: there is no code to
 * restate and no vector to pin, so this is the BUILD'S OWN specification (PARITY UNPINNED), checked by
 * self-consistency tests only.  The cos^C lobe of back_reflection_shader is replaced by the microfacet
 * backscatter lobe D_GGX * G_Smith, normalised to 1 at normal incidence; alpha^2 = 2 / (C + 2).  f32. */
float orc_ct_lobe(float angle, float specular_exp)
{
    float c = cosf(angle), sn = sinf(angle);
    if (!(c > 0.0f)) return 0.0f;
    float a2 = 2.0f / (specular_exp + 2.0f);
    a2 = fminf(fmaxf(a2, 1e-4f), 1.0f);
    float cc = c * c;
    float d = a2 * cc + sn * sn;              /* = cc (a2 - 1) + 1 without the cancellation near normal incidence */
    float D = a2 / (d * d);
    float g1 = (2.0f * c) / (c + sqrtf(a2 + (1.0f - a2) * cc));
    return a2 * D * (g1 * g1);
}

float orc_back_reflection_shader_model(float incidence_angle, float energy,
                                       float diffuse, float specular_fac, float specular_exp, int model)
{
    if (model != 1) return orc_back_reflection_shader(incidence_angle, energy, diffuse, specular_fac, specular_exp);
    float I_specular = orc_ct_lobe(incidence_angle, specular_exp);
    float I_total = diffuse * 1.0f + specular_fac * I_specular;
    return I_total * energy;
}

/* radar_algorithms.h:267-281 */
static void normalize_inplace(float* data, int n)
{
    float data_sum = 0.0f;
    for (int i = 0; i < n; i++) data_sum += data[i];
    for (int i = 0; i < n; i++) data[i] /= data_sum;
}

/* radar_algorithms.h:141-157 */
static float maxwell_boltzmann_pdf(float mode, float x)
{
    float a = (float)((double)mode / M_SQRT2);
    const float xx = x * x;
    const float aa = a * a;
    const float aaa = a * a * a;
    return (float)(sqrt(2.0 / M_PI) * (double)xx * (double)expf(-xx / (2 * aa)) / (double)aaa);
}

/* radar_algorithms.h:283-351 (+ RadarCPU.cpp:83-91 when rescale) */
void orc_make_denoiser(int kind, int width, int mode, int rescale, float* out)
{
    if (kind == 1 || kind == 2) {
        /* triangular; "gaussian" is the same code (radar_algorithms.h:310-335) */
        for (int i = 0; i < width; i++) {
            float p;
            if (i <= mode) {
                p = (float)i / (float)mode;
            } else {
                p = (float)(1.0 - (double)(((float)i - (float)mode) / ((float)width - (float)mode)));
            }
            /* p * 1.0f + (1.0 - p) * 0.0f, evaluated in double, stored as float */
            out[i] = (float)((double)(p * 1.0f) + (1.0 - (double)p) * (double)0.0f);
        }
    } else {
        for (int i = 0; i < width; i++) out[i] = maxwell_boltzmann_pdf((float)mode, (float)i);
    }
    normalize_inplace(out, width);
    if (rescale && width > 0) {
        double mode_val = out[mode];
        for (int i = 0; i < width; i++) out[i] = (float)((double)out[i] / mode_val);
    }
}

/* radar_algorithms.cpp:263-280: the radius law of one sample (D1..D4 of scripts/radaray_beams.py:23-25,78-92) */
float orc_cone_radius(float width, int sample_dist, float p_in_cone, float variate)
{
    float z = (float)(M_SQRT2 * (double)orc_erfinvf(p_in_cone));   /* :263 */
    float radius = (float)((double)width / 2.0);                    /* :265 */
    float random_radius = 0.0f;
    if (sample_dist == 0) {
        random_radius = variate * radius;
    } else if (sample_dist == 1) {
        random_radius = sqrtf(variate) * radius;
    } else if (sample_dist == 2) {
        random_radius = (variate / z) * radius;
    } else if (sample_dist == 3) {
        random_radius = sqrtf(fabsf(variate) / z) * radius;
    }
    return random_radius;
}

/* radar_algorithms.cpp:248-294 with externally supplied variates */
void orc_sample_cone_local(float width, int n_samples, int sample_dist, float p_in_cone,
                           const float* u_angle, const float* r_variate, float* out_dirs)
{
    for (int i = 0; i < n_samples; i++) {
        float random_angle = (float)((double)(u_angle[i] * 2.0f) * M_PI - M_PI);   /* :269 */
        float random_radius = orc_cone_radius(width, sample_dist, p_in_cone, r_variate[i]);
        float alpha = random_radius * cosf(random_angle);   /* :282 */
        float beta = random_radius * sinf(random_angle);
        quat q = q_from_euler(0.0f, alpha, beta);           /* :285 */
        v3 ex = { 1.0f, 0.0f, 0.0f };
        v3 d = q_rot(q, ex);                                /* :289 */
        out_dirs[3 * i + 0] = d.x; out_dirs[3 * i + 1] = d.y; out_dirs[3 * i + 2] = d.z;
    }
}

/* ------------------------------------------------------------------------ */
/* image_algorithms.h (Ken Perlin's improved noise)                          */
/* ------------------------------------------------------------------------ */
static const unsigned char PERM[256] = {
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30,
    69, 142, 8, 99, 37, 240, 21, 10, 23, 190, 6, 148, 247, 120, 234, 75, 0, 26, 197, 62,
    94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136,
    171, 168, 68, 175, 74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122,
    60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54, 65, 25, 63, 161,
    1, 216, 80, 73, 209, 76, 132, 187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86,
    164, 100, 109, 198, 173, 186, 3, 64, 52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126,
    255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213,
    119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253,
    19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112, 104, 218, 246, 97, 228, 251, 34, 242, 193,
    238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31,
    181, 199, 106, 157, 184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93,
    222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180
};
/* image_algorithms.h:14-50 stores the table twice (512 entries) */
static inline int perm(int i) { return PERM[i & 255]; }

static inline double perlin_fade(double t) { return t * t * t * (t * (t * 6 - 15) + 10); }   /* :53 */
static inline double perlin_lerp(double t, double a, double b) { return a + t * (b - a); }    /* :57 */
static inline double perlin_grad(int hash, double x, double y, double z)                       /* :61 */
{
    int h = hash & 15;
    double u = h < 8 ? x : y;
    double v = h < 4 ? y : (h == 12 || h == 14 ? x : z);
    return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
}

/* image_algorithms.h:69-106 */
double orc_perlin_noise(double src_x, double src_y, double src_z)
{
    int X = (int)floor(src_x) & 255;
    int Y = (int)floor(src_y) & 255;
    int Z = (int)floor(src_z) & 255;
    double x = src_x - floor(src_x);
    double y = src_y - floor(src_y);
    double z = src_z - floor(src_z);
    double u = perlin_fade(x), v = perlin_fade(y), w = perlin_fade(z);
    int A = perm(X) + Y;
    int AA = perm(A) + Z;
    int AB = perm(A + 1) + Z;
    int B = perm(X + 1) + Y;
    int BA = perm(B) + Z;
    int BB = perm(B + 1) + Z;
    return perlin_lerp(w,
        perlin_lerp(v,
            perlin_lerp(u, perlin_grad(perm(AA), x, y, z), perlin_grad(perm(BA), x - 1, y, z)),
            perlin_lerp(u, perlin_grad(perm(AB), x, y - 1, z), perlin_grad(perm(BB), x - 1, y - 1, z))),
        perlin_lerp(v,
            perlin_lerp(u, perlin_grad(perm(AA + 1), x, y, z - 1), perlin_grad(perm(BA + 1), x - 1, y, z - 1)),
            perlin_lerp(u, perlin_grad(perm(AB + 1), x, y - 1, z - 1), perlin_grad(perm(BB + 1), x - 1, y - 1, z - 1))));
}

/* image_algorithms.h:108-128 */
double orc_perlin_noise_hilo(double off_x, double off_y, double x, double y,
                             double scale_low, double scale_high, double p_low)
{
    double lo = orc_perlin_noise(off_x + x * scale_low, off_y + y * scale_low, 0.0);
    double hi = orc_perlin_noise(off_x + x * scale_high, off_y + y * scale_high, 0.0);
    return p_low * lo + (1.0 - p_low) * hi;
}

/* cv::saturate_cast<uchar>(float) = saturate(cvRound(x)); cvRound is
 * cvtss2si (round-half-even; NaN/out-of-range -> INT_MIN -> 0).  RadarCPU.cpp:542 */
uint8_t orc_saturate_u8(float x)
{
    if (!(x > -2147483648.0f && x < 2147483648.0f)) return 0;
    long iv = lrintf(x);
    return (uint8_t)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
}

/* ------------------------------------------------------------------------ */
/* Scene + nearest-hit query (stands in for rmagine/Embree, RadarCPU.cpp:236)*/
/* ------------------------------------------------------------------------ */
typedef struct {
    float bmin[3], bmax[3];
    uint32_t left;    /* inner: left child index (right = left+1); leaf: first prim */
    uint32_t count;   /* 0 inner, else #prims */
} bvh_node;

struct orc_scene {
    size_t nf;
    v3* v0; v3* e1; v3* e2;      /* per face, original order */
    uint32_t* obj;               /* per face object id */
    int use_bvh;
    bvh_node* nodes; size_t n_nodes;
    uint32_t* prim;              /* face indices in leaf order */
    float inflate;
    float* graze2;               /* per face: 2.5e-5 |e1 x e2|^2 (grazing guard of tri_hit) */
    float guard_pad;             /* 1e-5 x max(extent, largest |coordinate|) of the mesh */
};

#define ORC_TFAR 1000.0f   /* make_model range.max, radar_algorithms.cpp:157-158 */

/* Moeller-Trumbore, f32, no FMA.  Accepts 0 < t <= tfar.
 * Grazing guard (round 5; the BUILD'S definition -- rmagine / Embree are not in the checkout, their hit selection at
 * grazing incidence is unknown): for a ray within 0.3 degrees of the triangle's plane (det^2 < 2.5e-5 |e1 x e2|^2, i.e.
 * |d . n| < 5e-3) the test above is ill-conditioned and may accept a point centimetres outside the triangle -- a point
 * no bounding hierarchy over the triangle's box would ever visit, so the brute-force loop and a BVH (this file's own
 * intersect_bvh included) disagreed on such rays (1 in 39M fuzz rays, round 4).  Such a hit counts only if its point
 * o + t d lies inside the triangle's bounding box padded by guard_pad: the nearest hit is then a property of the mesh, not of
 * the structure that finds it.  Same arithmetic, un-fused, in the HIP kernel (traverse, rr_kernels.hip). */
static inline int tri_hit(const struct orc_scene* s, uint32_t f, v3 o, v3 d, float* t_out)
{
    const v3 e1 = s->e1[f], e2 = s->e2[f];
    const v3 pvec = v3_cross(d, e2);
    const float det = v3_dot(e1, pvec);
    if (det == 0.0f) return 0;
    const float inv = 1.0f / det;
    const v3 tvec = v3_sub(o, s->v0[f]);
    const float u = v3_dot(tvec, pvec) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return 0;
    const v3 qvec = v3_cross(tvec, e1);
    const float v = v3_dot(d, qvec) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return 0;
    const float t = v3_dot(e2, qvec) * inv;
    if (!(t > 0.0f && t <= ORC_TFAR)) return 0;
    if (det * det < s->graze2[f]) {
        const v3 v0 = s->v0[f];
        const v3 ph = v3_add(o, v3_scale(d, t)), v1 = v3_add(v0, e1), v2 = v3_add(v0, e2);
        const float pad = s->guard_pad;
        const int inside =
            ph.x >= fminf(v0.x, fminf(v1.x, v2.x)) - pad && ph.x <= fmaxf(v0.x, fmaxf(v1.x, v2.x)) + pad &&
            ph.y >= fminf(v0.y, fminf(v1.y, v2.y)) - pad && ph.y <= fmaxf(v0.y, fmaxf(v1.y, v2.y)) + pad &&
            ph.z >= fminf(v0.z, fminf(v1.z, v2.z)) - pad && ph.z <= fmaxf(v0.z, fmaxf(v1.z, v2.z)) + pad;
        if (!inside) return 0;
    }
    *t_out = t;
    return 1;
}

typedef struct { uint64_t nodes, tris; } trav_stats;

static int intersect_brute(const struct orc_scene* s, v3 o, v3 d, float* t, uint32_t* tri, trav_stats* st)
{
    float best = INFINITY; uint32_t bf = UINT32_MAX;
    for (uint32_t f = 0; f < (uint32_t)s->nf; f++) {
        float tt;
        if (tri_hit(s, f, o, d, &tt)) {
            if (tt < best || (tt == best && f < bf)) { best = tt; bf = f; }
        }
    }
    if (st) st->tris += s->nf;
    if (bf == UINT32_MAX) return 0;
    *t = best; *tri = bf;
    return 1;
}

static inline int slab(const bvh_node* n, v3 o, v3 inv, float tcull, float* tentry)
{
    float t1 = (n->bmin[0] - o.x) * inv.x, t2 = (n->bmax[0] - o.x) * inv.x;
    float tmin = fminf(t1, t2), tmax = fmaxf(t1, t2);
    t1 = (n->bmin[1] - o.y) * inv.y; t2 = (n->bmax[1] - o.y) * inv.y;
    tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
    t1 = (n->bmin[2] - o.z) * inv.z; t2 = (n->bmax[2] - o.z) * inv.z;
    tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
    *tentry = tmin;
    /* conservative: boxes are inflated at build time, exit padded by 2 ulp */
    return tmax * 1.0000004f >= fmaxf(tmin, 0.0f) && tmin <= tcull;
}

static int intersect_bvh(const struct orc_scene* s, v3 o, v3 d, float* t, uint32_t* tri, trav_stats* st)
{
    const v3 inv = { 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
    float best = INFINITY; uint32_t bf = UINT32_MAX;
    float tcull = ORC_TFAR * 1.0001f + 1e-3f;
    uint32_t stack[128]; int sp = 0;
    float te;
    if (!slab(&s->nodes[0], o, inv, tcull, &te)) return 0;
    stack[sp++] = 0;
    uint64_t nn = 0, nt = 0;
    while (sp) {
        const bvh_node* n = &s->nodes[stack[--sp]];
        nn++;
        if (n->count) {
            for (uint32_t i = 0; i < n->count; i++) {
                uint32_t f = s->prim[n->left + i];
                float tt; nt++;
                if (tri_hit(s, f, o, d, &tt)) {
                    if (tt < best || (tt == best && f < bf)) {
                        best = tt; bf = f;
                        tcull = best * 1.0001f + 1e-3f;
                    }
                }
            }
        } else {
            float ta, tb;
            int ha = slab(&s->nodes[n->left], o, inv, tcull, &ta);
            int hb = slab(&s->nodes[n->left + 1], o, inv, tcull, &tb);
            if (ha && hb) {
                if (ta <= tb) { stack[sp++] = n->left + 1; stack[sp++] = n->left; }
                else          { stack[sp++] = n->left; stack[sp++] = n->left + 1; }
            } else if (ha) stack[sp++] = n->left;
            else if (hb) stack[sp++] = n->left + 1;
        }
    }
    if (st) { st->nodes += nn; st->tris += nt; }
    if (bf == UINT32_MAX) return 0;
    *t = best; *tri = bf;
    return 1;
}

/* ---- binned-SAH BVH2 build ---- */
typedef struct { float mn[3], mx[3]; } aabb;
static inline void aabb_init(aabb* b) { for (int k = 0; k < 3; k++) { b->mn[k] = INFINITY; b->mx[k] = -INFINITY; } }
static inline void aabb_grow(aabb* b, const float* mn, const float* mx)
{
    for (int k = 0; k < 3; k++) { if (mn[k] < b->mn[k]) b->mn[k] = mn[k]; if (mx[k] > b->mx[k]) b->mx[k] = mx[k]; }
}
static inline float aabb_area(const aabb* b)
{
    float dx = b->mx[0] - b->mn[0], dy = b->mx[1] - b->mn[1], dz = b->mx[2] - b->mn[2];
    if (dx < 0) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
}

typedef struct {
    struct orc_scene* s;
    aabb* pb;        /* per-face bounds */
    float (*pc)[3];  /* per-face centroid */
} build_ctx;

#define NBINS 16
static void build_rec(build_ctx* c, uint32_t node_idx, uint32_t first, uint32_t count)
{
    struct orc_scene* s = c->s;
    aabb nb, cb; aabb_init(&nb); aabb_init(&cb);
    for (uint32_t i = first; i < first + count; i++) {
        uint32_t f = s->prim[i];
        aabb_grow(&nb, c->pb[f].mn, c->pb[f].mx);
        aabb_grow(&cb, c->pc[f], c->pc[f]);
    }
    bvh_node* n = &s->nodes[node_idx];
    for (int k = 0; k < 3; k++) { n->bmin[k] = nb.mn[k] - s->inflate; n->bmax[k] = nb.mx[k] + s->inflate; }
    if (count <= 4) { n->left = first; n->count = count; return; }

    int best_axis = -1, best_split = 0; float best_cost = INFINITY;
    for (int ax = 0; ax < 3; ax++) {
        float lo = cb.mn[ax], ext = cb.mx[ax] - cb.mn[ax];
        if (!(ext > 0.0f)) continue;
        aabb bb[NBINS]; uint32_t bc[NBINS];
        for (int b = 0; b < NBINS; b++) { aabb_init(&bb[b]); bc[b] = 0; }
        float scale = (float)NBINS / ext;
        for (uint32_t i = first; i < first + count; i++) {
            uint32_t f = s->prim[i];
            int b = (int)((c->pc[f][ax] - lo) * scale); if (b >= NBINS) b = NBINS - 1; if (b < 0) b = 0;
            bc[b]++; aabb_grow(&bb[b], c->pb[f].mn, c->pb[f].mx);
        }
        float la[NBINS], ra[NBINS]; uint32_t lc[NBINS], rc[NBINS];
        aabb acc; aabb_init(&acc); uint32_t cnt = 0;
        for (int b = 0; b < NBINS - 1; b++) { if (bc[b]) aabb_grow(&acc, bb[b].mn, bb[b].mx); cnt += bc[b]; la[b] = aabb_area(&acc); lc[b] = cnt; }
        aabb_init(&acc); cnt = 0;
        for (int b = NBINS - 1; b > 0; b--) { if (bc[b]) aabb_grow(&acc, bb[b].mn, bb[b].mx); cnt += bc[b]; ra[b - 1] = aabb_area(&acc); rc[b - 1] = cnt; }
        for (int b = 0; b < NBINS - 1; b++) {
            if (!lc[b] || !rc[b]) continue;
            float cost = la[b] * (float)lc[b] + ra[b] * (float)rc[b];
            if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = b; }
        }
    }
    uint32_t mid;
    if (best_axis < 0) {
        mid = first + count / 2;   /* all centroids coincide: split by index */
    } else {
        float lo = cb.mn[best_axis], ext = cb.mx[best_axis] - cb.mn[best_axis];
        float scale = (float)NBINS / ext;
        uint32_t i = first, j = first + count;
        while (i < j) {
            uint32_t f = s->prim[i];
            int b = (int)((c->pc[f][best_axis] - lo) * scale); if (b >= NBINS) b = NBINS - 1; if (b < 0) b = 0;
            if (b <= best_split) i++;
            else { j--; uint32_t tmp = s->prim[i]; s->prim[i] = s->prim[j]; s->prim[j] = tmp; }
        }
        mid = i;
        if (mid == first || mid == first + count) mid = first + count / 2;
    }
    uint32_t left = (uint32_t)s->n_nodes;
    s->n_nodes += 2;
    n->left = left; n->count = 0;
    build_rec(c, left, first, mid - first);
    build_rec(c, left + 1, mid, first + count - mid);
}

orc_scene* orc_scene_create(const float* verts, size_t nv, const uint32_t* faces, size_t nf,
                            const uint32_t* face_object_id, int use_bvh)
{
    struct orc_scene* s = (struct orc_scene*)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->nf = nf;
    s->v0 = (v3*)malloc(sizeof(v3) * (nf ? nf : 1));
    s->e1 = (v3*)malloc(sizeof(v3) * (nf ? nf : 1));
    s->e2 = (v3*)malloc(sizeof(v3) * (nf ? nf : 1));
    s->obj = (uint32_t*)malloc(sizeof(uint32_t) * (nf ? nf : 1));
    s->graze2 = (float*)malloc(sizeof(float) * (nf ? nf : 1));
    float smin[3] = { INFINITY, INFINITY, INFINITY }, smax[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t f = 0; f < nf; f++) {
        const float* a = verts + 3 * (size_t)faces[3 * f + 0];
        const float* b = verts + 3 * (size_t)faces[3 * f + 1];
        const float* c = verts + 3 * (size_t)faces[3 * f + 2];
        v3 va = { a[0], a[1], a[2] }, vb = { b[0], b[1], b[2] }, vc = { c[0], c[1], c[2] };
        s->v0[f] = va; s->e1[f] = v3_sub(vb, va); s->e2[f] = v3_sub(vc, va);
        { const v3 cr = v3_cross(s->e1[f], s->e2[f]); s->graze2[f] = 2.5e-5f * v3_dot(cr, cr); }
        s->obj[f] = face_object_id ? face_object_id[f] : 0u;
        for (int k = 0; k < 3; k++) {
            float lo = fminf(a[k], fminf(b[k], c[k])), hi = fmaxf(a[k], fmaxf(b[k], c[k]));
            if (lo < smin[k]) smin[k] = lo;
            if (hi > smax[k]) smax[k] = hi;
        }
    }
    (void)nv;
    s->use_bvh = use_bvh < 0 ? (nf > 4096) : use_bvh;
    s->guard_pad = 0.0f;
    if (nf > 0) {
        float ext = fmaxf(smax[0] - smin[0], fmaxf(smax[1] - smin[1], smax[2] - smin[2]));
        float mag = 0.0f;
        for (int k = 0; k < 3; k++) mag = fmaxf(mag, fmaxf(fabsf(smin[k]), fabsf(smax[k])));
        s->guard_pad = 1e-5f * fmaxf(ext, mag);
    }
    if (s->use_bvh && nf > 0) {
        float ext = fmaxf(smax[0] - smin[0], fmaxf(smax[1] - smin[1], smax[2] - smin[2]));
        float mag = 0.0f;
        for (int k = 0; k < 3; k++) mag = fmaxf(mag, fmaxf(fabsf(smin[k]), fabsf(smax[k])));
        s->inflate = 2e-5f * fmaxf(ext, mag) + 1e-6f;
        build_ctx c; c.s = s;
        c.pb = (aabb*)malloc(sizeof(aabb) * nf);
        c.pc = (float(*)[3])malloc(sizeof(float[3]) * nf);
        s->prim = (uint32_t*)malloc(sizeof(uint32_t) * nf);
        for (size_t f = 0; f < nf; f++) {
            const float* a = verts + 3 * (size_t)faces[3 * f + 0];
            const float* b = verts + 3 * (size_t)faces[3 * f + 1];
            const float* cc = verts + 3 * (size_t)faces[3 * f + 2];
            for (int k = 0; k < 3; k++) {
                c.pb[f].mn[k] = fminf(a[k], fminf(b[k], cc[k]));
                c.pb[f].mx[k] = fmaxf(a[k], fmaxf(b[k], cc[k]));
                c.pc[f][k] = 0.5f * (c.pb[f].mn[k] + c.pb[f].mx[k]);
            }
            s->prim[f] = (uint32_t)f;
        }
        s->nodes = (bvh_node*)malloc(sizeof(bvh_node) * (2 * nf + 1));
        s->n_nodes = 1;
        build_rec(&c, 0, 0, (uint32_t)nf);
        free(c.pb); free(c.pc);
    }
    return s;
}

void orc_scene_destroy(orc_scene* s)
{
    if (!s) return;
    free(s->v0); free(s->e1); free(s->e2); free(s->obj); free(s->graze2); free(s->nodes); free(s->prim);
    free(s);
}

static inline int scene_intersect(const struct orc_scene* s, v3 o, v3 d, float* t, uint32_t* tri, trav_stats* st)
{
    if (s->nf == 0) return 0;
    return s->use_bvh ? intersect_bvh(s, o, d, t, tri, st) : intersect_brute(s, o, d, t, tri, st);
}

int orc_intersect(const orc_scene* s, const float orig[3], const float dir[3],
                  float* t, uint32_t* tri, float ng[3])
{
    v3 o = { orig[0], orig[1], orig[2] }, d = { dir[0], dir[1], dir[2] };
    float tt; uint32_t f;
    if (!scene_intersect(s, o, d, &tt, &f, NULL)) return 0;
    *t = tt; *tri = f;
    if (ng) { v3 n = v3_cross(s->e1[f], s->e2[f]); ng[0] = n.x; ng[1] = n.y; ng[2] = n.z; }
    return 1;
}

/* ------------------------------------------------------------------------ */
/* RadarCPU::simulate                                                        */
/* ------------------------------------------------------------------------ */
typedef struct { double time, strength; } signal_t;   /* radar_types.h:23-27 */

typedef struct { wave_t* p; size_t n, cap; } wave_vec;
typedef struct { signal_t* p; size_t n, cap; } sig_vec;

static void wv_push(wave_vec* v, const wave_t* w)
{
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 256; v->p = (wave_t*)realloc(v->p, v->cap * sizeof(wave_t)); }
    v->p[v->n++] = *w;
}
static void sv_push(sig_vec* v, double time, double strength)
{
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 256; v->p = (signal_t*)realloc(v->p, v->cap * sizeof(signal_t)); }
    v->p[v->n].time = time; v->p[v->n].strength = strength; v->n++;
}

/* RadarCPU.cpp:497-512: amplitude of the ambient noise of one bin as a function of its signal (signal_min = 0,
 * signal_max = the column's max_val); the law of scripts/func_deformer.py:6-21 (noise_amplitude) with the labels the C++
 * uses: noise_at_0 scales the amplitude where the signal is 0, noise_at_1 where it is at its maximum */
float orc_noise_amplitude(float signal, float max_val, double at_signal_0, double at_signal_1)
{
    float signal_min = 0;
    float signal_max = max_val;
    float signal_amp = signal_max - signal_min;
    float signal_ = (float)(1.0 - (double)((signal - signal_min) / signal_amp));
    float noise_at_0 = (float)((double)signal_amp * at_signal_0);
    float noise_at_1 = (float)((double)signal_amp * at_signal_1);
    float signal__ = (float)pow((double)signal_, 4.0);
    return (float)((double)(signal__ * noise_at_0) + (1.0 - (double)signal__) * (double)noise_at_1);
}

/* counter-based uniform in [0,1) for ambient_noise==1; the reference draws
 * from std::random_device there (RadarCPU.cpp:461-482) -> unreproducible, so
 * the variate stream is DEFINED here (and identically in the product). */
static inline float uniform01(uint32_t seed, uint32_t col, uint32_t i)
{
    uint64_t z = ((uint64_t)seed << 40) ^ ((uint64_t)col << 20) ^ (uint64_t)i;
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

static double now_s(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int simulate_impl(const orc_scene* scene,
                 const orc_material* materials, size_t n_materials,
                 const int32_t* object_materials, size_t n_objects,
                 const orc_config* cfg,
                 const float* beam_dirs, size_t n_beam,
                 const float* pose, int pose_stride,
                 const float* noise_rnd,
                 int az_begin, int az_end,
                 uint8_t* out_u8, float* out_f32,
                 int n_threads, orc_stats* stats)
{
    if (!scene || !cfg || !pose || (!out_u8 && !out_f32)) return -1;
    const int n_cells = cfg->n_cells, n_angles = cfg->n_angles;
    if (az_begin < 0 || az_end > n_angles || az_begin > az_end) return -2;

    /* RadarCPU.cpp:48-93: smear kernel */
    float* w = NULL; int wn = 0, mode = 0;
    if (cfg->signal_denoising > 0) {
        int width = 0; double mfrac = 0.0;
        if (cfg->signal_denoising == 1) { width = cfg->signal_denoising_triangular_width; mfrac = cfg->signal_denoising_triangular_mode; }
        else if (cfg->signal_denoising == 2) { width = cfg->signal_denoising_gaussian_width; mfrac = cfg->signal_denoising_gaussian_mode; }
        else if (cfg->signal_denoising == 3) { width = cfg->signal_denoising_mb_width; mfrac = cfg->signal_denoising_mb_mode; }
        if (width > 0) {
            mode = (int)(mfrac * width);   /* :57 */
            w = (float*)malloc(sizeof(float) * (size_t)width);
            orc_make_denoiser(cfg->signal_denoising, width, mode, 1, w);
            wn = width;
        }
    }

    const float thr = cfg->wave_energy_threshold;

    uint64_t tot_wp = 0, tot_hits = 0, tot_sig = 0, tot_nodes = 0, tot_tris = 0, tot_near = 0;
    int err = 0;

    /* tooling only (tools/treeq: BVH-quality study on real ray sets): ORC_RAYLOG=<file> appends every cast ray
     * as { int32 azimuth, int32 pass, float o[3], float d[3], uint32 parent << 1 | refraction, uint32 material } in
     * map coordinates (parent = index of the parent wave among the rays of the previous pass of that azimuth) */
    FILE* raylog = getenv("ORC_RAYLOG") ? fopen(getenv("ORC_RAYLOG"), "ab") : NULL;
    /* tooling only (tools/xcd_rows.py): ORC_COUNTS=<file> appends "azimuth pass waves" for every pass of every azimuth */
    FILE* cntlog = getenv("ORC_COUNTS") ? fopen(getenv("ORC_COUNTS"), "a") : NULL;

    const double t_start = now_s();   /* RadarCPU.cpp:147-148 */

#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    (void)n_threads;
#endif

    /* The reference's loop is `#pragma omp parallel for` over the azimuths with everything allocated per azimuth inside
     * (RadarCPU.cpp:155-187,402) and every thread writing its column straight into the shared image (stride n_angles).
     * As a BASELINE that stops scaling at 32 threads: heap traffic per azimuth, and 64 neighbouring u8 columns share a
     * cache line that five threads write.  Same arithmetic, same order, kinder to the memory system: one arena per thread
     * (wave / signal vectors and the slice, reused over its azimuths), azimuths handed out dynamically (their cost varies
     * with the scene), columns kept column-major [azimuth][cell] and transposed into the image once at the end. */
    const size_t n_az = (size_t)(az_end > az_begin ? az_end - az_begin : 0);
    uint8_t* cols_u8 = out_u8 ? (uint8_t*)malloc(n_az * (size_t)n_cells + 1) : NULL;
    float* cols_f32 = out_f32 ? (float*)malloc((n_az * (size_t)n_cells + 1) * sizeof(float)) : NULL;
    #pragma omp parallel num_threads(n_threads) \
        reduction(+:tot_wp,tot_hits,tot_sig,tot_nodes,tot_tris,tot_near) reduction(|:err)
    {
    wave_vec waves = { 0 }, waves_new = { 0 };
    sig_vec signals = { 0 };
    float* slice = (float*)malloc((size_t)n_cells * sizeof(float) + 4);
    #pragma omp for schedule(dynamic, 1)
    for (int angle_id = az_begin; angle_id < az_end; angle_id++)    /* :155-156 */
    {
        waves.n = 0; waves_new.n = 0; signals.n = 0;
        trav_stats st = { 0, 0 };

        /* :106-114, :184  waves = m_waves_start */
        for (size_t i = 0; i < n_beam; i++) {
            wave_t wv;
            wv.orig.x = 0.0f; wv.orig.y = 0.0f; wv.orig.z = 0.0f;
            wv.dir.x = beam_dirs[3 * i]; wv.dir.y = beam_dirs[3 * i + 1]; wv.dir.z = beam_dirs[3 * i + 2];
            wv.energy = 1.0; wv.polarization = 0.5; wv.velocity = 0.3; wv.time = 0.0;
            wv.material_id = 0; wv.dbg_parent = 0;
            wv_push(&waves, &wv);
        }

        /* :190-196 include_motion: Tsm is looked up per azimuth (pose_stride = 7), else once (:127-134) */
        const float* ps = pose + (size_t)pose_stride * (size_t)angle_id;
        const quat q_sm = { ps[0], ps[1], ps[2], ps[3] };
        const v3 t_sm = { ps[4], ps[5], ps[6] };
        /* :201-206  Tas.R = Euler(0,0,theta(angle_id)), Tas.t = 0; Tam = Tsm * Tas */
        const float theta = cfg->theta_min + (float)angle_id * cfg->theta_inc;
        const quat q_as = q_from_euler(0.0f, 0.0f, theta);
        const v3 zero = { 0.0f, 0.0f, 0.0f };
        const quat q_am = q_mul(q_sm, q_as);
        const v3 t_am = v3_add(q_rot(q_sm, zero), t_sm);
        const quat q_ma = q_inv(q_am);

        for (int pass_id = 0; pass_id < cfg->n_reflections; pass_id++)   /* :220 */
        {
            waves_new.n = 0;
            if (cntlog) {
                #pragma omp critical(cntlog)
                fprintf(cntlog, "%d %d %zu\n", angle_id, pass_id, waves.n);
            }
            for (size_t i = 0; i < waves.n; i++)   /* :243 */
            {
                wave_t wave = waves.p[i];
                /* :236 ray cast in map frame */
                const v3 o_m = v3_add(q_rot(q_am, wave.orig), t_am);
                const v3 d_m = q_rot(q_am, wave.dir);
                float wave_range; uint32_t f;
                tot_wp++;
                if (raylog) {
                    const int32_t hd[2] = { angle_id, pass_id };
                    const float od[6] = { o_m.x, o_m.y, o_m.z, d_m.x, d_m.y, d_m.z };
                    #pragma omp critical(raylog)
                    { const uint32_t ex[2] = { wave.dbg_parent, wave.material_id };
                      fwrite(hd, sizeof hd, 1, raylog); fwrite(od, sizeof od, 1, raylog); fwrite(ex, sizeof ex, 1, raylog); }
                }
                if (!scene_intersect(scene, o_m, d_m, &wave_range, &f, &st)) continue;   /* :252 */
                tot_hits++;
                const uint32_t obj_id = scene->obj[f];
                v3 nint = v3_normalize(v3_cross(scene->e1[f], scene->e2[f]));
                nint = q_rot(q_ma, nint);
                if (v3_dot(wave.dir, nint) > 0.0f) nint = v3_neg(nint);
                const v3 surface_normal = v3_normalize(nint);   /* :248 */

                wave_t incidence = wave;                /* :258 */
                wave_move_inplace(&incidence, (double)wave_range);

                wave_t reflection = incidence, refraction = incidence;   /* :261-262 */

                if ((int32_t)incidence.material_id == cfg->material_id_air) {   /* :266 */
                    if (obj_id >= n_objects) { err |= 1; continue; }
                    refraction.material_id = (uint32_t)object_materials[obj_id];
                } else {
                    refraction.material_id = (uint32_t)cfg->material_id_air;
                }
                if (refraction.material_id >= n_materials) { err |= 2; continue; }

                float v_refraction = 1.0f;   /* :273 */
                if (incidence.material_id != refraction.material_id) {
                    v_refraction = materials[refraction.material_id].velocity;
                } else {
                    v_refraction = (float)incidence.velocity;
                }

                v3 rdir, tdir; double renergy, tenergy;   /* :283 */
                fresnel_split(surface_normal, incidence.dir, incidence.energy, incidence.polarization,
                              incidence.velocity, (double)v_refraction, &rdir, &renergy, &tdir, &tenergy);
                reflection.dir = rdir; reflection.energy = renergy;   /* :285-286 */

                /* test bookkeeping (not in the reference): energies within 1e-6 of the pruning threshold -- the last
                 * ulp of libm's acosf decides such a wave's fate, see tests/test_gpu_round3.py */
                if (fabs(reflection.energy - (double)thr) < 1e-6 || fabs(tenergy - (double)thr) < 1e-6) tot_near++;
                if (reflection.energy > (double)thr)   /* :288 */
                {
                    reflection.dbg_parent = (uint32_t)i << 1;
                    wv_push(&waves_new, &reflection);
                    if ((int32_t)reflection.material_id == cfg->material_id_air)   /* :302 */
                    {
                        const orc_material material = materials[refraction.material_id];
                        double incidence_angle = incidence_angle_of(surface_normal, incidence.dir);   /* :308 */
                        double return_energy_path = (double)orc_back_reflection_shader_model(   /* :310-316 */
                            (float)incidence_angle, (float)reflection.energy,
                            material.ambient, material.diffuse, material.specular, cfg->brdf_model);

                        if (pass_id == 0 || cfg->record_multi_reflection) {   /* :319 */
                            float time_back = (float)(incidence.time * 2.0);
                            sv_push(&signals, (double)time_back, return_energy_path);
                        }
                        if (pass_id > 0 && cfg->record_multi_path)   /* :325 */
                        {
                            v3 dir_sensor_to_hit = reflection.orig;
                            const double dist = (double)v3_l2norm(dir_sensor_to_hit);
                            dir_sensor_to_hit = v3_normalize(dir_sensor_to_hit);
                            const double time_to_sensor = dist / reflection.velocity;
                            double sensor_view_scalar = (double)v3_dot(wave.dir, dir_sensor_to_hit);
                            double ang = (double)acosf(v3_dot(v3_neg(reflection.dir), dir_sensor_to_hit));   /* angle_between, radar_algorithms.h:17-23 */
                            if (sensor_view_scalar > cfg->multipath_threshold) {   /* :344-345 */
                                double return_energy_air = (double)orc_back_reflection_shader_model(
                                    (float)ang, (float)reflection.energy,
                                    material.ambient, material.diffuse, material.specular, cfg->brdf_model);
                                sv_push(&signals, incidence.time + time_to_sensor, return_energy_air);
                            }
                        }
                    }
                }

                refraction.dir = tdir; refraction.energy = tenergy;   /* :364-365 */
                if (refraction.energy > (double)thr) {
                    refraction.dbg_parent = ((uint32_t)i << 1) | 1u;
                    wv_push(&waves_new, &refraction);
                }
            }

            const float skip_dist = 0.001f;   /* :374 */
            for (size_t i = 0; i < waves_new.n; i++) wave_move_inplace(&waves_new.p[i], (double)skip_dist);

            wave_vec tmp = waves; waves = waves_new; waves_new = tmp;   /* :380 */
        }

        /* :402-450 signals -> slice */
        memset(slice, 0, (size_t)n_cells * sizeof(float));
        float max_val = 0.0f;
        for (size_t i = 0; i < signals.n; i++)
        {
            const signal_t signal = signals.p[i];
            float half_time = (float)(signal.time / 2.0);       /* :410 */
            float signal_dist = (float)(0.3 * (double)half_time); /* :411 */
            int cell = (int)((double)signal_dist / cfg->resolution);   /* :413 */
            if (cell < n_cells)
            {
                if (cfg->signal_denoising > 0) {
                    for (int vid = 0; vid < wn; vid++) {
                        int glob_id = vid + cell - mode;
                        if (glob_id > 0 && glob_id < n_cells) {   /* :424 */
                            slice[glob_id] = (float)((double)slice[glob_id] + signal.strength * (double)w[vid]);
                            if (slice[glob_id] > max_val) max_val = slice[glob_id];
                        }
                    }
                } else if (cell >= 0) {
                    slice[cell] = fmaxf(slice[cell], (float)signal.strength);   /* :439 */
                    if (slice[cell] > max_val) max_val = slice[cell];
                }
            }
        }
        tot_sig += signals.n;

        /* :453  slice *= energy_max  (cv convertTo: x * (float)alpha) */
        {
            const float a = (float)cfg->energy_max;
            for (int i = 0; i < n_cells; i++) slice[i] = slice[i] * a;
        }

        const int col = (cfg->scroll_image + angle_id) % n_angles;   /* :457 */

        if (cfg->ambient_noise)   /* :459-528 */
        {
            const double scale = 0.05, scale2 = 0.2;
            const float rnd = noise_rnd ? noise_rnd[angle_id] : 0.0f;
            const double random_begin = (double)rnd;   /* :472 (dist_uni(gen) * 1000.0, injected) */
            for (int i = 0; i < n_cells; i++)
            {
                float signal = slice[i];
                double p = 0.0;
                if (cfg->ambient_noise == 1) {
                    p = (double)uniform01((uint32_t)(int32_t)rnd, (uint32_t)col, (uint32_t)i);
                } else if (cfg->ambient_noise == 2) {
                    double p1 = orc_perlin_noise(random_begin + (double)i * scale, (double)col * scale, 0.0);
                    double p2 = orc_perlin_noise(random_begin + (double)i * scale2, (double)col * scale2, 0.0);
                    p = 0.9 * p1 + 0.1 * p2;
                }
                float signal_max = max_val;
                float noise_amp = orc_noise_amplitude(signal, max_val, cfg->ambient_noise_at_signal_0, cfg->ambient_noise_at_signal_1);
                float noise_energy_max = (float)((double)signal_max * cfg->ambient_noise_energy_max);
                float noise_energy_min = (float)((double)signal_max * cfg->ambient_noise_energy_min);
                float energy_loss = (float)cfg->ambient_noise_energy_loss;
                float y_noise = (float)((double)noise_amp * p);
                float x = (float)(((double)(float)i + 0.5) * cfg->resolution);
                y_noise = y_noise + (noise_energy_max - noise_energy_min) * expf(-energy_loss * x) + noise_energy_min;
                y_noise = fabsf(y_noise);
                slice[i] = signal + y_noise;
            }
        }

        /* :533  slice *= signal_max / max_val */
        {
            const float a = (float)(cfg->signal_max / (double)max_val);
            for (int i = 0; i < n_cells; i++) slice[i] = slice[i] * a;
        }

        /* :542  convertTo(col, CV_8UC1) */
        {
            const size_t cb = (size_t)(angle_id - az_begin) * (size_t)n_cells;
            (void)col;
            if (cols_u8) for (int i = 0; i < n_cells; i++) cols_u8[cb + i] = orc_saturate_u8(slice[i]);
            if (cols_f32) memcpy(cols_f32 + cb, slice, (size_t)n_cells * sizeof(float));
        }

        tot_nodes += st.nodes; tot_tris += st.tris;
    }
    free(slice); free(waves.p); free(waves_new.p); free(signals.p);
    }
    /* :457,542  column `col` of the image, all azimuths at once (rows of the image in parallel) */
    #pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int i = 0; i < n_cells; i++)
        for (int angle_id = az_begin; angle_id < az_end; angle_id++) {
            const int col = (cfg->scroll_image + angle_id) % n_angles;
            const size_t k = (size_t)(angle_id - az_begin) * (size_t)n_cells + (size_t)i;
            if (out_u8) out_u8[(size_t)i * n_angles + col] = cols_u8[k];
            if (out_f32) out_f32[(size_t)i * n_angles + col] = cols_f32[k];
        }
    free(cols_u8); free(cols_f32);

    const double t_stop = now_s();   /* :550 */
    if (raylog) fclose(raylog);
    if (cntlog) fclose(cntlog);
    free(w);
    if (stats) {
        stats->wave_passes = tot_wp; stats->hits = tot_hits; stats->signals = tot_sig; stats->near_threshold = tot_near;
        stats->nodes_visited = tot_nodes; stats->tris_tested = tot_tris;
        stats->seconds = t_stop - t_start;
    }
    return err ? -10 - err : 0;
}

int orc_simulate(const orc_scene* scene,
                 const orc_material* materials, size_t n_materials,
                 const int32_t* object_materials, size_t n_objects,
                 const orc_config* cfg,
                 const float* beam_dirs, size_t n_beam,
                 const float pose[7],
                 const float* noise_rnd,
                 int az_begin, int az_end,
                 uint8_t* out_u8, float* out_f32,
                 int n_threads, orc_stats* stats)
{
    return simulate_impl(scene, materials, n_materials, object_materials, n_objects, cfg, beam_dirs, n_beam,
                         pose, 0, noise_rnd, az_begin, az_end, out_u8, out_f32, n_threads, stats);
}

/* include_motion = true (RadarCPU.cpp:190-196): one Tsm per azimuth, poses[n_angles][7].
 * (The reference serialises this mode only because of ros::spinOnce, :544-547.) */
int orc_simulate_motion(const orc_scene* scene,
                 const orc_material* materials, size_t n_materials,
                 const int32_t* object_materials, size_t n_objects,
                 const orc_config* cfg,
                 const float* beam_dirs, size_t n_beam,
                 const float* poses,
                 const float* noise_rnd,
                 int az_begin, int az_end,
                 uint8_t* out_u8, float* out_f32,
                 int n_threads, orc_stats* stats)
{
    return simulate_impl(scene, materials, n_materials, object_materials, n_objects, cfg, beam_dirs, n_beam,
                         poses, 7, noise_rnd, az_begin, az_end, out_u8, out_f32, n_threads, stats);
}
