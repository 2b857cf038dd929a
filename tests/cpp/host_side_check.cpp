// rr_host.cpp + rr_collada.cpp (the host-only pieces of the C ABI: beam sampler, PLY / OBJ / COLLADA map loader) under AddressSanitizer + UBSan:
// well-formed files, then thousands of damaged ones (truncated, bytes flipped, counts inflated) -- the loader parses files
// from disk, it must fail with an error code, never with a crash or an out-of-bounds access.
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -I include tests/cpp/host_side_check.cpp radarays_ros_amd/csrc/rr_host.cpp radarays_ros_amd/csrc/rr_collada.cpp -o /tmp/hsc && /tmp/hsc /tmp
#include <radarays_mi355.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

static uint64_t g_s = 0x243F6A8885A308D3ull;
static uint32_t rnd() { g_s = g_s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(g_s >> 33); }
static void put(const std::string& path, const std::string& bytes) { std::ofstream f(path, std::ios::binary); f.write(bytes.data(), (std::streamsize)bytes.size()); }

// rr_mesh_reorder_objects (round 6) on a loaded mesh: the reversed order of its names must permute the ids and the names
// together (face f keeps its object's NAME) and nothing else; an unknown name, a name listed twice, a null name and a mesh
// without names are refused with a message and leave the mesh as it was.  Duplicate names in the mesh itself (possible in a
// damaged file) are refused too.
static int reorder_check(rr_mesh& m, const char* what)
{
    int bad = 0;
    char err[200] = "";
    if (!m.object_names) {
        const char* one[1] = { "x" };
        if (rr_mesh_reorder_objects(&m, one, 1, err, sizeof(err)) == 0 || !err[0]) { std::printf("%s: reorder accepted a mesh without names\n", what); bad++; }
        if (rr_mesh_reorder_objects(&m, nullptr, 0, err, sizeof(err)) != 0) { std::printf("%s: an empty order was refused\n", what); bad++; }
        return bad;
    }
    const size_t n = m.n_objects;
    std::vector<std::string> before_name(m.n_faces);
    for (size_t f = 0; f < m.n_faces; f++) before_name[f] = m.object_names[m.face_object_id[f]];
    const std::vector<uint32_t> before_id(m.face_object_id, m.face_object_id + m.n_faces);
    bool dup = false;
    for (size_t i = 0; i < n && !dup; i++) for (size_t j = i + 1; j < n; j++) if (std::strcmp(m.object_names[i], m.object_names[j]) == 0) { dup = true; break; }
    std::vector<std::string> keep(n);
    for (size_t i = 0; i < n; i++) keep[i] = m.object_names[n - 1 - i];
    std::vector<const char*> rev(n);
    for (size_t i = 0; i < n; i++) rev[i] = keep[i].c_str();
    const int rc = rr_mesh_reorder_objects(&m, rev.data(), n, err, sizeof(err));
    if (dup) {
        if (rc == 0) { std::printf("%s: duplicate object names were accepted\n", what); bad++; }
        else if (!std::equal(before_id.begin(), before_id.end(), m.face_object_id)) { std::printf("%s: a refused reorder changed the ids\n", what); bad++; }
        return bad;
    }
    if (rc != 0) { std::printf("%s: the reversed order was refused: %s\n", what, err); return 1; }
    for (size_t f = 0; f < m.n_faces; f++) {
        if (m.face_object_id[f] >= n || before_name[f] != m.object_names[m.face_object_id[f]]) { std::printf("%s: face %zu lost its object's name\n", what, f); bad++; break; }
        if (m.face_object_id[f] != (uint32_t)(n - 1 - before_id[f])) { std::printf("%s: face %zu: id not reversed\n", what, f); bad++; break; }
    }
    for (size_t i = 0; i < n; i++) if (keep[i] != m.object_names[i]) { std::printf("%s: names not in the requested order\n", what); bad++; break; }
    const std::vector<uint32_t> now(m.face_object_id, m.face_object_id + m.n_faces);
    const char* unknown[1] = { "\x01 no such object" };
    const char* twice[2] = { rev[0], rev[0] };
    const char* nul[1] = { nullptr };
    if (rr_mesh_reorder_objects(&m, unknown, 1, err, sizeof(err)) == 0) { std::printf("%s: an unknown name was accepted\n", what); bad++; }
    if (n && rr_mesh_reorder_objects(&m, twice, 2, err, sizeof(err)) == 0) { std::printf("%s: a name listed twice was accepted\n", what); bad++; }
    if (rr_mesh_reorder_objects(&m, nul, 1, err, sizeof(err)) == 0) { std::printf("%s: a null name was accepted\n", what); bad++; }
    if (rr_mesh_reorder_objects(&m, rev.data(), 1, nullptr, 0) != 0 && n) { /* a partial order is fine; no error buffer is fine */ std::printf("%s: a partial order was refused\n", what); bad++; }
    if (!std::equal(now.begin(), now.end(), m.face_object_id) && n <= 1) { std::printf("%s: ids changed by no-op orders\n", what); bad++; }
    return bad;
}

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    int fails = 0;
    // ---- beam sampler ------------------------------------------------------------------------------------------
    {
        std::vector<float> d(3 * 1000);
        for (int dist = 0; dist < 4; dist++) {
            if (rr_sample_cone_local(42, 0.17f, 1000, dist, 0.8f, d.data())) { std::printf("sampler dist %d failed\n", dist); fails++; }
            for (int i = 0; i < 1000; i++) {
                const float n = std::sqrt(d[3 * i] * d[3 * i] + d[3 * i + 1] * d[3 * i + 1] + d[3 * i + 2] * d[3 * i + 2]);
                if (!(std::fabs(n - 1.0f) < 1e-5f)) { std::printf("direction %d of dist %d is not a unit vector\n", i, dist); fails++; break; }
            }
        }
        if (rr_sample_cone_local(1, 0.1f, 4, 7, 0.8f, d.data()) == 0 || rr_cone_dirs(0.1f, -1, 0.8f, d.data(), d.data(), 1, d.data()) == 0) { std::printf("bad sample_dist accepted\n"); fails++; }
        if (rr_sample_cone_local(1, 0.1f, 0, 2, 0.8f, nullptr) != 0) { std::printf("n = 0 refused\n"); fails++; }
        std::printf("sampler: %s\n", fails ? "FAILED" : "ok");
    }
    // ---- well-formed files -------------------------------------------------------------------------------------
    const std::string ply_ascii = "ply\nformat ascii 1.0\ncomment x\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\n"
                                  "element face 2\nproperty list uchar int vertex_indices\nend_header\n0 0 0 1\n1 0 0 2\n1 1 0 3\n0 1 0 4\n4 0 1 2 3\n3 0 2 3\n";
    std::string ply_bin = "ply\nformat binary_little_endian 1.0\nelement vertex 3\nproperty double x\nproperty double y\nproperty double z\n"
                          "element face 1\nproperty list uchar uint vertex_index\nend_header\n";
    { const double v[9] = { 0, 0, 0, 1, 0, 0, 0, 1, 0 }; ply_bin.append((const char*)v, sizeof(v)); const unsigned char c = 3; ply_bin.push_back((char)c);
      const uint32_t idx[3] = { 0, 1, 2 }; ply_bin.append((const char*)idx, sizeof(idx)); }
    const std::string obj = "# c\no a\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\ng b\nf -4//1 -3//1 -2//1\nvn 0 0 1\n";
    // COLLADA: two geometries (polylist with two inputs per corner, triangles), nested nodes, a library node instantiated
    // by the scene, every transform element, a comment, a CDATA array
    const std::string dae =
        "<?xml version=\"1.0\"?>\n<!-- c --><COLLADA xmlns=\"http://www.collada.org/2005/11/COLLADASchema\" version=\"1.4.1\">"
        "<asset><unit name=\"cm\" meter=\"0.5\"/><up_axis>Z_UP</up_axis></asset><library_geometries>"
        "<geometry id=\"W\" name=\"Wall\"><mesh><source id=\"Wp\"><float_array id=\"Wa\" count=\"12\">0 0 0 1 0 0 1 1 0 0 1 0</float_array>"
        "<technique_common><accessor source=\"#Wa\" count=\"4\" stride=\"3\"/></technique_common></source>"
        "<vertices id=\"Wv\"><input semantic=\"POSITION\" source=\"#Wp\"/></vertices>"
        "<polylist count=\"1\" material=\"m0\"><input semantic=\"VERTEX\" source=\"#Wv\" offset=\"0\"/><input semantic=\"NORMAL\" source=\"#n\" offset=\"1\"/>"
        "<vcount>4</vcount><p>0 0 1 0 2 0 3 0</p></polylist><tristrips count=\"1\" material=\"m1\"><input semantic=\"VERTEX\" source=\"#Wv\" offset=\"0\"/><p>0 1 3 2</p></tristrips></mesh></geometry>"
        "<geometry id=\"D\"><mesh><source id=\"Dp\"><float_array id=\"Da\" count=\"9\"><![CDATA[0 0 0 2 0 0 0 2 0]]></float_array>"
        "<technique_common><accessor source=\"#Da\" count=\"3\" stride=\"3\"/></technique_common></source>"
        "<vertices id=\"Dv\"><input semantic=\"POSITION\" source=\"#Dp\"/></vertices>"
        "<triangles count=\"1\"><input semantic=\"VERTEX\" source=\"#Dv\" offset=\"0\"/><p>0 1 2</p></triangles></mesh></geometry></library_geometries>"
        "<library_nodes><node id=\"L\"><scale>2 2 2</scale><instance_geometry url=\"#D\"/></node></library_nodes>"
        "<library_visual_scenes><visual_scene id=\"S\"><node id=\"a\"><translate>10 0 0</translate><rotate>0 0 1 90</rotate><instance_geometry url=\"#D\"/></node>"
        "<node id=\"b\"><matrix>1 0 0 0 0 1 0 5 0 0 1 0 0 0 0 1</matrix><instance_geometry url=\"#W\"/><instance_node url=\"#L\"/>"
        "<node id=\"c\"><instance_geometry url=\"#D\"/></node></node></visual_scene></library_visual_scenes>"
        "<scene><instance_visual_scene url=\"#S\"/></scene></COLLADA>\n";
    struct Case { const char* name; std::string ext, bytes; size_t nv, nf; } good[4] = {
        { "ply ascii", ".ply", ply_ascii, 4, 3 }, { "ply binary", ".ply", ply_bin, 3, 1 }, { "obj", ".obj", obj, 4, 3 }, { "dae", ".dae", dae, 21, 7 } };
    for (const Case& c : good) {
        const std::string p = dir + "/hsc_good" + c.ext;
        put(p, c.bytes);
        rr_mesh m; char err[256] = "";
        const int rc = rr_load_mesh_file(p.c_str(), &m, err, sizeof(err));
        const bool ok = rc == 0 && m.n_verts == c.nv && m.n_faces == c.nf;
        std::printf("%s: %s %s\n", c.name, ok ? "ok" : "FAILED", err);
        fails += !ok;
        if (rc == 0) fails += reorder_check(m, c.name);
        rr_free_mesh(&m); rr_free_mesh(&m);          // twice: must be harmless
    }
    // ---- damaged files: error code or a mesh whose indices are in range, never a crash ----------------------------
    size_t n_ok = 0, n_err = 0;
    const int n_iter = argc > 2 ? std::atoi(argv[2]) : 8000;
    for (int it = 0; it < n_iter; it++) {
        const Case& c = good[it % 4];
        std::string b = c.bytes;
        const int kind = (int)(rnd() % 4);
        if (kind == 0) b.resize(rnd() % (b.size() + 1));                                        // truncated
        else if (kind == 1) for (int k = 0; k < 1 + (int)(rnd() % 4); k++) b[rnd() % b.size()] = (char)(rnd() & 0xFF);   // bytes flipped
        else if (kind == 2) { const size_t at = rnd() % b.size(); b.insert(at, std::to_string(rnd())); }   // digits inserted (inflated counts / indices)
        else { const size_t at = rnd() % b.size(), n = rnd() % 16; b.erase(at, std::min(n, b.size() - at)); }  // bytes removed
        const std::string p = dir + "/hsc_bad" + c.ext;
        put(p, b);
        rr_mesh m; char err[256] = "";
        const int rc = rr_load_mesh_file(p.c_str(), &m, err, sizeof(err));
        if (rc == 0) {
            n_ok++;
            for (size_t i = 0; i < 3 * m.n_faces; i++) if (m.faces[i] >= m.n_verts) { std::printf("iteration %d: index out of range in an accepted mesh\n", it); fails++; break; }
            for (size_t i = 0; i < m.n_faces; i++) if (m.face_object_id[i] >= m.n_objects) { std::printf("iteration %d: object id out of range\n", it); fails++; break; }
            if (m.object_names) for (size_t i = 0; i < m.n_objects; i++) if (!m.object_names[i] || std::strlen(m.object_names[i]) > b.size()) { std::printf("iteration %d: bad object name\n", it); fails++; break; }
            if (it % 8 == 0) fails += reorder_check(m, "damaged file");
            rr_free_mesh(&m);
        } else {
            n_err++;
            if (m.verts || m.faces || m.face_object_id || m.object_names || !err[0]) { std::printf("iteration %d: a failed load left pointers or no message\n", it); fails++; }
        }
    }
    std::printf("damaged files: %zu accepted (indices in range), %zu refused: %s\n", n_ok, n_err, fails ? "FAILED" : "ok");
    char e2[8]; rr_mesh m2;
    if (rr_load_mesh_file((dir + "/does_not_exist.ply").c_str(), &m2, e2, sizeof(e2)) == 0 || std::strlen(e2) >= sizeof(e2)) { std::printf("short error buffer mishandled\n"); fails++; }
    if (rr_load_mesh_file(nullptr, &m2, nullptr, 0) == 0) { std::printf("null path accepted\n"); fails++; }
    std::printf("%s\n", fails ? "FAILED" : "all: ok");
    return fails ? 1 : 0;
}
