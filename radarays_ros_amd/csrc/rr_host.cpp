// rr_host.cpp -- the host-only pieces of the drop-in that the reference does on the CPU around simulate():
//
//   * the beam sampler: sample_cone_local (src/radarays_ros/radar_algorithms.cpp:248-294) + erfinvf
//     (include/radarays_ros/radar_math.h:13-44).  RadarCPU::simulate re-draws m_waves_start whenever a
//     dynamic-reconfigure changed the beam (RadarCPU.cpp:136-145); a C/C++ host of this library needs the same
//     (include/radarays_ros_amd/RadarHIP.hpp::push).  The reference seeds std::mt19937 from std::random_device and
//     draws through libstdc++'s distributions (not portable, not reproducible); here the generator is SEEDED and the
//     variate streams are numpy's RandomState streams (MT19937, 53-bit doubles, polar Box-Muller), so that
//     rr_sample_cone_local(seed, ...) and radarays_ros_amd/beams.py give the same directions.  All angles of a call
//     are drawn first, then all radii (beams.variates).
//   * the map loader: what rm::import_embree_map(map_file) does for the node (src/radar_simulator.cpp:149): PLY
//     (ascii / binary, either byte order; MulRan maps are .ply, launch/mulran_sim.launch:7) and Wavefront OBJ
//     (objects `o` / `g` -> object ids) into the flat arrays rr_set_mesh takes; polygons are fan-triangulated.
//     Same results as radarays_ros_amd/meshio.py (tests/test_host_side.py).  COLLADA: rr_collada.cpp.
//
// No GPU, no HIP call in this file.
#include "../../include/radarays_mi355.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace rr_collada {
bool load_dae(const std::string& path, std::vector<float>& verts, std::vector<uint32_t>& faces, std::vector<uint32_t>& obj,
              std::vector<std::string>& names, std::string& err);
}

namespace {

// radar_math.h:13-44: single-precision polynomial in t = log(1 - a^2), two branches, Horner with fmaf
float erfinv_f32(float a)
{
    float t = std::fmaf(a, 0.0f - a, 1.0f);
    t = std::log(t);
    static const float far_c[9] = { 3.03697567e-10f, 2.93243101e-8f, 1.22150334e-6f, 2.84108955e-5f, 3.93552968e-4f,
                                    3.02698812e-3f, 4.83185798e-3f, -2.64646143e-1f, 8.40016484e-1f };
    static const float near_c[10] = { 5.43877832e-9f, 1.43285448e-7f, 1.22774793e-6f, 1.12963626e-7f, -5.61530760e-5f,
                                      -1.47697632e-4f, 2.31468678e-3f, 1.15392581e-2f, -2.32015476e-1f, 8.86226892e-1f };
    const bool far = std::fabs(t) > 6.125f;
    const float* c = far ? far_c : near_c;
    const int n = far ? 9 : 10;
    float p = c[0];
    for (int k = 1; k < n; k++) p = std::fmaf(p, t, c[k]);
    return a * p;
}

// MT19937 (Matsumoto & Nishimura) with numpy.random.RandomState's draws on top
struct NumpyRandomState {
    uint32_t mt[624]; int idx = 624;
    bool has_gauss = false; double gauss = 0.0;
    explicit NumpyRandomState(uint32_t seed)
    {
        mt[0] = seed;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    }
    uint32_t next32()
    {
        if (idx >= 624) {
            for (int k = 0; k < 624; k++) {
                const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7FFFFFFFu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680u; y ^= (y << 15) & 0xEFC60000u; y ^= y >> 18;
        return y;
    }
    double next_double() { const uint32_t a = next32() >> 5, b = next32() >> 6; return (a * 67108864.0 + b) / 9007199254740992.0; }
    double next_gauss()          // legacy polar Box-Muller, second value cached
    {
        if (has_gauss) { has_gauss = false; const double g = gauss; gauss = 0.0; return g; }
        double x1, x2, r2;
        do { x1 = 2.0 * next_double() - 1.0; x2 = 2.0 * next_double() - 1.0; r2 = x1 * x1 + x2 * x2; } while (r2 >= 1.0 || r2 == 0.0);
        const double f = std::sqrt(-2.0 * std::log(r2) / r2);
        gauss = f * x1; has_gauss = true;
        return f * x2;
    }
};

struct Q { float x, y, z, w; };
// rmagine EulerAngles{roll, pitch, yaw} -> Quaternion (ZYX), as the oracle states it
Q quat_of_euler(float roll, float pitch, float yaw)
{
    const float cr = std::cos(roll / 2.0f), sr = std::sin(roll / 2.0f);
    const float cp = std::cos(pitch / 2.0f), sp = std::sin(pitch / 2.0f);
    const float cy = std::cos(yaw / 2.0f), sy = std::sin(yaw / 2.0f);
    Q q;
    q.w = cr * cp * cy + sr * sp * sy;
    q.x = sr * cp * cy - cr * sp * sy;
    q.y = cr * sp * cy + sr * cp * sy;
    q.z = cr * cp * sy - sr * sp * cy;
    return q;
}
Q qmul(Q a, Q b)
{
    Q r;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    return r;
}

void set_err(char* err, size_t n, const std::string& msg)
{
    if (err && n) { std::snprintf(err, n, "%s", msg.c_str()); }
}

// ---- PLY ----------------------------------------------------------------------------------------------------
struct PlyProp { bool list = false; std::string type, count_type, name; };
struct PlyElem { std::string name; size_t count = 0; std::vector<PlyProp> props; };

int ply_size(const std::string& t)
{
    if (t == "char" || t == "int8" || t == "uchar" || t == "uint8") return 1;
    if (t == "short" || t == "int16" || t == "ushort" || t == "uint16") return 2;
    if (t == "int" || t == "int32" || t == "uint" || t == "uint32" || t == "float" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    return 0;
}
// one binary scalar of PLY type t, as a double (exact for every type a mesh uses; indices stay below 2^53)
bool ply_read(std::istream& f, const std::string& t, bool big, double& out)
{
    unsigned char b[8];
    const int n = ply_size(t);
    if (!n || !f.read((char*)b, n)) return false;
    if (big) for (int i = 0; i < n / 2; i++) std::swap(b[i], b[n - 1 - i]);
    if (t == "char" || t == "int8") { int8_t v; std::memcpy(&v, b, 1); out = v; }
    else if (t == "uchar" || t == "uint8") { out = b[0]; }
    else if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, b, 2); out = v; }
    else if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, b, 2); out = v; }
    else if (t == "int" || t == "int32") { int32_t v; std::memcpy(&v, b, 4); out = v; }
    else if (t == "uint" || t == "uint32") { uint32_t v; std::memcpy(&v, b, 4); out = v; }
    else if (t == "float" || t == "float32") { float v; std::memcpy(&v, b, 4); out = v; }
    else { double v; std::memcpy(&v, b, 8); out = v; }
    return true;
}

// a vertex index read as a double (PLY) -> long long: only a finite value inside [0, 2^32) is an index; anything else becomes
// -1, which fan() refuses (casting NaN / inf to an integer is undefined, and a value above 2^32 would wrap to a valid index)
inline long long index_of(double v) { return (std::isfinite(v) && v >= 0.0 && v < 4294967296.0) ? (long long)v : -1; }

// false: an index outside [0, 2^32) (advisor, round 4)
bool fan(const std::vector<long long>& idx, std::vector<uint32_t>& faces)
{
    for (long long i : idx) if (i < 0 || i > 0xFFFFFFFFll) return false;
    for (size_t k = 1; k + 1 < idx.size(); k++) { faces.push_back((uint32_t)idx[0]); faces.push_back((uint32_t)idx[k]); faces.push_back((uint32_t)idx[k + 1]); }
    return true;
}

bool load_ply(const std::string& path, std::vector<float>& verts, std::vector<uint32_t>& faces, std::string& err)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = path + ": cannot open"; return false; }
    f.seekg(0, std::ios::end);
    const size_t file_bytes = (size_t)std::max<std::streamoff>(0, f.tellg());     // no element count of the header is trusted beyond it
    f.seekg(0, std::ios::beg);
    std::string line;
    std::getline(f, line);
    while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
    if (line != "ply") { err = path + ": not a PLY file"; return false; }
    std::string fmt; std::vector<PlyElem> elems; bool ended = false;
    while (std::getline(f, line)) {
        std::istringstream ss(line); std::string t; ss >> t;
        if (t.empty() || t == "comment" || t == "obj_info") continue;
        if (t == "format") ss >> fmt;
        else if (t == "element") { PlyElem e; ss >> e.name >> e.count; elems.push_back(e); }
        else if (t == "property") {
            if (elems.empty()) { err = path + ": property before element"; return false; }
            PlyProp p; std::string a; ss >> a;
            if (a == "list") { p.list = true; ss >> p.count_type >> p.type >> p.name; } else { p.type = a; ss >> p.name; }
            elems.back().props.push_back(p);
        } else if (t == "end_header") { ended = true; break; }
    }
    if (!ended) { err = path + ": truncated PLY header"; return false; }
    const bool ascii = fmt == "ascii", big = fmt == "binary_big_endian";
    if (!ascii && !big && fmt != "binary_little_endian") { err = path + ": unsupported PLY format '" + fmt + "'"; return false; }
    bool have_verts = false;
    for (const PlyElem& e : elems) {
        const bool is_v = e.name == "vertex", is_f = e.name == "face";
        int ix = -1, iy = -1, iz = -1;
        for (size_t k = 0; k < e.props.size(); k++) {
            if (e.props[k].list) continue;
            if (e.props[k].name == "x") ix = (int)k; else if (e.props[k].name == "y") iy = (int)k; else if (e.props[k].name == "z") iz = (int)k;
        }
        // no element count of the header is trusted beyond the file: a row with properties takes at least a byte, and an
        // element WITHOUT properties reads nothing, so nothing would ever fail at EOF (`element foo 18446744073709551615`
        // spun for 2^64 rows: advisor, round 4)
        if (e.count > file_bytes) { err = path + ": element '" + e.name + "': count exceeds the file"; return false; }
        if (is_v) {
            if (ix < 0 || iy < 0 || iz < 0) { err = path + ": vertex element without x / y / z"; return false; }
            verts.reserve(3 * e.count); have_verts = true;
        }
        if (e.props.empty()) continue;
        std::vector<long long> idx;
        for (size_t r = 0; r < e.count; r++) {
            float xyz[3] = { 0, 0, 0 };
            for (size_t k = 0; k < e.props.size(); k++) {
                const PlyProp& p = e.props[k];
                if (!p.list) {
                    double v = 0.0;
                    if (ascii) { if (!(f >> v)) { err = path + ": truncated PLY body"; return false; } }
                    else if (!ply_read(f, p.type, big, v)) { err = path + ": truncated PLY body"; return false; }
                    if ((int)k == ix) xyz[0] = (float)v; else if ((int)k == iy) xyz[1] = (float)v; else if ((int)k == iz) xyz[2] = (float)v;
                } else {
                    double cnt = 0.0;
                    if (ascii) { if (!(f >> cnt)) { err = path + ": truncated PLY body"; return false; } }
                    else if (!ply_read(f, p.count_type, big, cnt)) { err = path + ": truncated PLY body"; return false; }
                    const bool want = is_f && (p.name == "vertex_indices" || p.name == "vertex_index");
                    if (want) idx.clear();
                    if (!(cnt >= 0.0 && cnt <= (double)file_bytes)) { err = path + ": list length exceeds the file"; return false; }
                    for (long long j = 0; j < (long long)cnt; j++) {
                        double v = 0.0;
                        if (ascii) { if (!(f >> v)) { err = path + ": truncated PLY body"; return false; } }
                        else if (!ply_read(f, p.type, big, v)) { err = path + ": truncated PLY body"; return false; }
                        if (want) idx.push_back(index_of(v));
                    }
                    if (want && !fan(idx, faces)) { err = path + ": face index outside [0, 2^32)"; return false; }
                }
            }
            if (is_v) { verts.push_back(xyz[0]); verts.push_back(xyz[1]); verts.push_back(xyz[2]); }
        }
    }
    if (!have_verts) { err = path + ": no vertex element"; return false; }
    return true;
}

// ---- OBJ ----------------------------------------------------------------------------------------------------
bool load_obj(const std::string& path, std::vector<float>& verts, std::vector<uint32_t>& faces, std::vector<uint32_t>& obj,
              std::vector<std::string>& names, std::string& err)
{
    std::ifstream f(path);
    if (!f) { err = path + ": cannot open"; return false; }
    std::string line; long long cur = -1;
    std::vector<long long> idx;
    while (std::getline(f, line)) {
        std::istringstream ss(line); std::string t; ss >> t;
        if (t.empty() || t[0] == '#') continue;
        if (t == "v") {
            double x, y, z;
            if (ss >> x >> y >> z) { verts.push_back((float)x); verts.push_back((float)y); verts.push_back((float)z); }
        } else if (t == "o" || t == "g") {
            std::string name, w;
            while (ss >> w) { if (!name.empty()) name += ' '; name += w; }
            cur = (long long)names.size();
            names.push_back(name);
        } else if (t == "f") {
            idx.clear();
            std::string tok;
            while (ss >> tok) {
                const long long i = std::strtoll(tok.c_str(), nullptr, 10);      // "v", "v/vt", "v/vt/vn", "v//vn": the part before the first '/'
                idx.push_back(i > 0 ? i - 1 : (long long)(verts.size() / 3) + i);
            }
            if (idx.size() < 3) continue;
            const size_t before = faces.size() / 3;
            if (!fan(idx, faces)) { err = path + ": face index before the first vertex"; return false; }
            for (size_t k = before; k < faces.size() / 3; k++) obj.push_back((uint32_t)(cur > 0 ? cur : 0));
        }
    }
    return true;
}

}  // namespace

extern "C" {

int rr_cone_dirs(float width_rad, int sample_dist, float p_in_cone, const float* u_angle, const float* r_variate, size_t n,
                 float* out_dirs)
{
    if (n && (!u_angle || !r_variate || !out_dirs)) return -3;
    if (sample_dist < 0 || sample_dist > 3) return -3;
    const float z = (float)(M_SQRT2 * (double)erfinv_f32(p_in_cone));     // radar_algorithms.cpp:263
    const float radius = (float)((double)width_rad / 2.0);                 // :265
    for (size_t i = 0; i < n; i++) {
        const float angle = (float)((double)(u_angle[i] * 2.0f) * M_PI - M_PI);     // :269
        float r = 0.0f;                                                              // :272-280
        if (sample_dist == 0) r = r_variate[i] * radius;
        else if (sample_dist == 1) r = std::sqrt(r_variate[i]) * radius;
        else if (sample_dist == 2) r = (r_variate[i] / z) * radius;
        else r = std::sqrt(std::fabs(r_variate[i]) / z) * radius;
        const float alpha = r * std::cos(angle), beta = r * std::sin(angle);        // :282-283
        const Q q = quat_of_euler(0.0f, alpha, beta);                                // :285
        const Q ex = { 1.0f, 0.0f, 0.0f, 0.0f }, qc = { -q.x, -q.y, -q.z, q.w };
        const Q d = qmul(qmul(q, ex), qc);                                           // :289  q * (1, 0, 0)
        out_dirs[3 * i + 0] = d.x; out_dirs[3 * i + 1] = d.y; out_dirs[3 * i + 2] = d.z;
    }
    return 0;
}

int rr_sample_cone_local(uint32_t seed, float width_rad, size_t n, int sample_dist, float p_in_cone, float* out_dirs)
{
    if (n && !out_dirs) return -3;
    if (sample_dist < 0 || sample_dist > 3) return -3;
    NumpyRandomState rs(seed);
    std::vector<float> u(n), r(n);
    for (size_t i = 0; i < n; i++) u[i] = (float)(0.0 + (1.0 - 0.0) * rs.next_double());
    if (sample_dist <= 1) for (size_t i = 0; i < n; i++) r[i] = (float)(0.0 + (1.0 - 0.0) * rs.next_double());
    else for (size_t i = 0; i < n; i++) r[i] = (float)rs.next_gauss();
    return rr_cone_dirs(width_rad, sample_dist, p_in_cone, u.data(), r.data(), n, out_dirs);
}

int rr_load_mesh_file(const char* path, rr_mesh* out, char* err, size_t err_len)
{
    if (!path || !out) { set_err(err, err_len, "rr_load_mesh_file: null argument"); return -3; }
    std::memset(out, 0, sizeof(*out));
    const std::string p(path);
    std::string ext = p.size() >= 4 ? p.substr(p.size() - 4) : "";
    for (char& ch : ext) ch = (char)std::tolower((unsigned char)ch);
    std::vector<float> verts; std::vector<uint32_t> faces, obj; std::vector<std::string> names; std::string e;
    try {
        if (ext == ".ply") { if (!load_ply(p, verts, faces, e)) { set_err(err, err_len, e); return -4; } obj.assign(faces.size() / 3, 0u); }
        else if (ext == ".obj") { if (!load_obj(p, verts, faces, obj, names, e)) { set_err(err, err_len, e); return -4; } }
        else if (ext == ".dae") { if (!rr_collada::load_dae(p, verts, faces, obj, names, e)) { set_err(err, err_len, e); return -4; } }
        else { set_err(err, err_len, p + ": unsupported mesh format (PLY, OBJ and COLLADA are read)"); return -4; }
    } catch (const std::exception& ex) { set_err(err, err_len, p + ": " + ex.what()); return -4; }
    const size_t nv = verts.size() / 3, nf = faces.size() / 3, n_obj = names.size();
    for (uint32_t i : faces) if ((size_t)i >= nv) { set_err(err, err_len, p + ": face index out of range"); return -4; }
    out->verts = (float*)std::malloc(std::max<size_t>(1, verts.size()) * sizeof(float));
    out->faces = (uint32_t*)std::malloc(std::max<size_t>(1, faces.size()) * sizeof(uint32_t));
    out->face_object_id = (uint32_t*)std::malloc(std::max<size_t>(1, nf) * sizeof(uint32_t));
    if (!out->verts || !out->faces || !out->face_object_id) { rr_free_mesh(out); set_err(err, err_len, "rr_load_mesh_file: out of memory"); return -4; }
    if (!verts.empty()) std::memcpy(out->verts, verts.data(), verts.size() * sizeof(float));
    if (!faces.empty()) std::memcpy(out->faces, faces.data(), faces.size() * sizeof(uint32_t));
    if (nf) std::memcpy(out->face_object_id, obj.data(), nf * sizeof(uint32_t));
    out->n_verts = nv; out->n_faces = nf; out->n_objects = std::max<size_t>(1, n_obj);
    if (n_obj) {
        out->object_names = (char**)std::calloc(n_obj, sizeof(char*));
        bool ok = out->object_names != nullptr;
        for (size_t k = 0; ok && k < n_obj; k++) {
            out->object_names[k] = (char*)std::malloc(names[k].size() + 1);
            if (!out->object_names[k]) { ok = false; break; }
            std::memcpy(out->object_names[k], names[k].c_str(), names[k].size() + 1);
        }
        if (!ok) { rr_free_mesh(out); set_err(err, err_len, "rr_load_mesh_file: out of memory"); return -4; }
    }
    return 0;
}

int rr_mesh_reorder_objects(rr_mesh* m, const char* const* order, size_t n_order, char* err, size_t err_len)
{
    if (!m || (n_order && !order)) { set_err(err, err_len, "rr_mesh_reorder_objects: null argument"); return -3; }
    if (n_order == 0) return 0;
    if (!m->object_names) { set_err(err, err_len, "rr_mesh_reorder_objects: the mesh carries no object names (PLY?)"); return -3; }
    const size_t n = m->n_objects;
    std::vector<uint32_t> new_id(n, 0xFFFFFFFFu);
    uint32_t next = 0;
    for (size_t k = 0; k < n_order; k++) {
        if (!order[k]) { set_err(err, err_len, "rr_mesh_reorder_objects: null name"); return -3; }
        size_t hit = n;
        for (size_t i = 0; i < n; i++) if (std::strcmp(m->object_names[i], order[k]) == 0) { if (hit != n) { set_err(err, err_len, std::string("rr_mesh_reorder_objects: the mesh has two objects named '") + order[k] + "'"); return -3; } hit = i; }
        if (hit == n) { set_err(err, err_len, std::string("rr_mesh_reorder_objects: no object named '") + order[k] + "' in the mesh"); return -3; }
        if (new_id[hit] != 0xFFFFFFFFu) { set_err(err, err_len, std::string("rr_mesh_reorder_objects: '") + order[k] + "' is listed twice"); return -3; }
        new_id[hit] = next++;
    }
    for (size_t i = 0; i < n; i++) if (new_id[i] == 0xFFFFFFFFu) new_id[i] = next++;      // the unlisted ones keep their relative order behind the listed
    for (size_t f = 0; f < m->n_faces; f++)          // (checked before anything is changed: a refusal leaves the mesh as it was)
        if ((size_t)m->face_object_id[f] >= n) { set_err(err, err_len, "rr_mesh_reorder_objects: face_object_id out of range"); return -3; }
    for (size_t f = 0; f < m->n_faces; f++) m->face_object_id[f] = new_id[m->face_object_id[f]];
    std::vector<char*> names(n);
    for (size_t i = 0; i < n; i++) names[new_id[i]] = m->object_names[i];
    for (size_t i = 0; i < n; i++) m->object_names[i] = names[i];
    return 0;
}

void rr_free_mesh(rr_mesh* m)
{
    if (!m) return;
    std::free(m->verts); std::free(m->faces); std::free(m->face_object_id);
    if (m->object_names) { for (size_t k = 0; k < m->n_objects; k++) std::free(m->object_names[k]); std::free(m->object_names); }
    std::memset(m, 0, sizeof(*m));
}

}  // extern "C"
