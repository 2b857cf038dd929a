// rr_collada.cpp -- the COLLADA (.dae) arm of rr_load_mesh_file.
//
// The reference's DEFAULT map is a Blender COLLADA export (launch/mro_husky.launch:4 `oru4.dae`; also
// launch/tests/ray_tracing_test.launch:5, radar_sim_test.launch:6), read by rm::import_embree_map
// (src/radar_simulator.cpp:149) through assimp; its scene OBJECTS are what `object_materials` indexes
// (config/oru4_test.yaml:37-56).  assimp is not in this image, so this is an own reader of the subset such exports
// use: <library_geometries> (float sources with accessor stride / offset; <triangles>, <polylist>, <polygons>,
// <trifans>, <tristrips>, several inputs per corner), <library_visual_scenes> / <library_nodes> (<matrix>,
// <translate>, <rotate>, <scale> in document order, nested nodes, <instance_node>), <asset><unit meter>.
//
// Result = radarays_ros_amd/meshio.py::load_dae with its defaults (tests/test_host_side.py compares the two):
//   * one object per instantiated (geometry, primitive group), numbered depth-first in scene order
//     (a node's own geometries, then its instance_nodes, then its child nodes); names = the geometry's name,
//     with "[material]" appended when the geometry has several groups;
//   * corner positions transformed by the node chain and scaled by <unit meter> (assimp does both); the up axis is
//     NOT rotated: the radar works in the frame the file was modelled in (Blender writes Z_UP);
//   * vertices are not shared between triangles (3 per face).
// The object numbering is this build's specification: rmagine's own numbering of assimp's meshes could not be read
// offline -- check rr_mesh.object_names against the material table of your scene.
//
// No XML library either: a non-validating reader for well-formed documents (elements, attributes, text, comments,
// processing instructions, CDATA, DOCTYPE skipped), iterative, every count bounded by the file's size.
#include "../../include/radarays_mi355.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <string_view>
#include <vector>

namespace rr_collada {

namespace {

using sv = std::string_view;

struct XNode {
    sv name;                                              // local name (namespace prefix dropped)
    std::vector<std::pair<sv, std::string>> attrs;        // values with the five predefined entities decoded
    sv text;                                              // first character-data run (data elements have no children)
    std::vector<int> kids;
    const std::string* attr(sv k) const
    {
        for (const auto& a : attrs) if (a.first == k) return &a.second;
        return nullptr;
    }
};

struct Doc {
    std::string buf;
    std::vector<XNode> nodes;                             // nodes[0] = the document element
    std::vector<int> kids(int e, sv name) const
    {
        std::vector<int> r;
        for (int c : nodes[e].kids) if (nodes[c].name == name) r.push_back(c);
        return r;
    }
    int first(int e, sv name) const
    {
        for (int c : nodes[e].kids) if (nodes[c].name == name) return c;
        return -1;
    }
};

bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
bool is_name_char(char c) { return !(is_space(c) || c == '=' || c == '>' || c == '/' || c == '<' || c == '"' || c == '\''); }
sv local_name(sv n) { const size_t k = n.rfind(':'); return k == sv::npos ? n : n.substr(k + 1); }

std::string decode_entities(sv s)
{
    std::string o; o.reserve(s.size());
    for (size_t i = 0; i < s.size(); i++) {
        if (s[i] == '&') {
            static const struct { const char* e; char c; } tab[5] = { { "&lt;", '<' }, { "&gt;", '>' }, { "&amp;", '&' }, { "&quot;", '"' }, { "&apos;", '\'' } };
            bool hit = false;
            for (const auto& t : tab) {
                const size_t n = std::strlen(t.e);
                if (s.compare(i, n, t.e) == 0) { o.push_back(t.c); i += n - 1; hit = true; break; }
            }
            if (hit) continue;
        }
        o.push_back(s[i]);
    }
    return o;
}

bool parse_xml(Doc& d, std::string& err)
{
    const std::string& b = d.buf;
    const size_t n = b.size();
    size_t i = 0;
    if (n >= 3 && (unsigned char)b[0] == 0xEF && (unsigned char)b[1] == 0xBB && (unsigned char)b[2] == 0xBF) i = 3;
    std::vector<int> open;
    bool root_closed = false;
    auto skip_to = [&](const char* end_mark) -> bool {
        const size_t k = b.find(end_mark, i);
        if (k == std::string::npos) return false;
        i = k + std::strlen(end_mark);
        return true;
    };
    while (i < n) {
        if (b[i] != '<') {                                  // character data
            const size_t k = b.find('<', i);
            const size_t e = k == std::string::npos ? n : k;
            if (!open.empty()) {
                size_t a = i, z = e;
                while (a < z && is_space(b[a])) a++;
                while (z > a && is_space(b[z - 1])) z--;
                XNode& x = d.nodes[open.back()];
                if (z > a && x.text.empty()) x.text = sv(b.data() + a, z - a);
            } else {
                for (size_t k2 = i; k2 < e; k2++) if (!is_space(b[k2])) { err = "text outside the document element"; return false; }
            }
            i = e;
            continue;
        }
        if (b.compare(i, 4, "<!--") == 0) { i += 4; if (!skip_to("-->")) { err = "unterminated comment"; return false; } continue; }
        if (b.compare(i, 2, "<?") == 0) { i += 2; if (!skip_to("?>")) { err = "unterminated processing instruction"; return false; } continue; }
        if (b.compare(i, 9, "<![CDATA[") == 0) {
            const size_t a = i + 9;
            i = a;
            if (!skip_to("]]>")) { err = "unterminated CDATA section"; return false; }
            if (!open.empty() && d.nodes[open.back()].text.empty()) d.nodes[open.back()].text = sv(b.data() + a, i - 3 - a);
            continue;
        }
        if (b.compare(i, 2, "<!") == 0) {                   // DOCTYPE and friends: skip to the matching '>' ([...] subsets included)
            int depth = 0; size_t k = i + 2; bool done = false;
            for (; k < n; k++) {
                if (b[k] == '[') depth++;
                else if (b[k] == ']') depth--;
                else if (b[k] == '>' && depth <= 0) { done = true; break; }
            }
            if (!done) { err = "unterminated declaration"; return false; }
            i = k + 1;
            continue;
        }
        if (b.compare(i, 2, "</") == 0) {                   // end tag
            size_t k = i + 2;
            while (k < n && is_name_char(b[k])) k++;
            const sv nm = local_name(sv(b.data() + i + 2, k - (i + 2)));
            while (k < n && is_space(b[k])) k++;
            if (k >= n || b[k] != '>') { err = "malformed end tag"; return false; }
            if (open.empty() || d.nodes[open.back()].name != nm) { err = "mismatched end tag </" + std::string(nm) + ">"; return false; }
            open.pop_back();
            if (open.empty()) root_closed = true;
            i = k + 1;
            continue;
        }
        // start tag
        if (root_closed) { err = "more than one document element"; return false; }
        size_t k = i + 1;
        while (k < n && is_name_char(b[k])) k++;
        if (k == i + 1) { err = "malformed tag"; return false; }
        if (open.size() >= 256) { err = "elements nested deeper than 256"; return false; }
        const int id = (int)d.nodes.size();
        d.nodes.emplace_back();
        d.nodes[id].name = local_name(sv(b.data() + i + 1, k - (i + 1)));
        if (!open.empty()) d.nodes[open.back()].kids.push_back(id);
        bool self_closed = false, closed = false;
        while (k < n) {
            while (k < n && is_space(b[k])) k++;
            if (k >= n) break;
            if (b[k] == '>') { k++; closed = true; break; }
            if (b[k] == '/') {
                if (k + 1 < n && b[k + 1] == '>') { k += 2; closed = self_closed = true; break; }
                err = "malformed tag"; return false;
            }
            const size_t a0 = k;
            while (k < n && is_name_char(b[k])) k++;
            if (k == a0) { err = "malformed attribute"; return false; }
            const sv an(b.data() + a0, k - a0);
            while (k < n && is_space(b[k])) k++;
            if (k >= n || b[k] != '=') { err = "attribute without a value"; return false; }
            k++;
            while (k < n && is_space(b[k])) k++;
            if (k >= n || (b[k] != '"' && b[k] != '\'')) { err = "attribute value not quoted"; return false; }
            const char q = b[k++];
            const size_t v0 = k;
            while (k < n && b[k] != q) k++;
            if (k >= n) { err = "unterminated attribute value"; return false; }
            d.nodes[id].attrs.emplace_back(an, decode_entities(sv(b.data() + v0, k - v0)));   // (xmlns:* keep their prefix: never looked up)
            k++;
        }
        if (!closed) { err = "unterminated tag"; return false; }
        if (!self_closed) open.push_back(id);
        else if (open.empty()) root_closed = true;
        i = k;
    }
    if (!open.empty()) { err = "document ends inside <" + std::string(d.nodes[open.back()].name) + ">"; return false; }
    if (d.nodes.empty()) { err = "no document element"; return false; }
    return true;
}

// whitespace-separated numbers of a text run.  The run lies inside buf and is followed by '<' or the string's
// terminator, where strtod / strtoll stop by themselves.
bool parse_doubles(sv t, std::vector<double>& out)
{
    out.clear();
    out.reserve(t.size() / 2 + 1);
    const char* p = t.data(); const char* e = p + t.size();
    while (true) {
        while (p < e && is_space(*p)) p++;
        if (p >= e) return true;
        char* q = nullptr;
        const double v = std::strtod(p, &q);
        if (q == p || q > e) return false;
        out.push_back(v);
        p = q;
    }
}

bool parse_ints(sv t, std::vector<long long>& out)
{
    out.clear();
    out.reserve(t.size() / 2 + 1);
    const char* p = t.data(); const char* e = p + t.size();
    while (true) {
        while (p < e && is_space(*p)) p++;
        if (p >= e) return true;
        char* q = nullptr;
        const long long v = std::strtoll(p, &q, 10);
        if (q == p || q > e) return false;
        out.push_back(v);
        p = q;
    }
}

struct M4 { double a[4][4]; };
M4 eye() { M4 m; std::memset(&m, 0, sizeof(m)); for (int i = 0; i < 4; i++) m.a[i][i] = 1.0; return m; }
M4 mul(const M4& x, const M4& y)
{
    M4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        double s = 0.0;
        for (int k = 0; k < 4; k++) s += x.a[i][k] * y.a[k][j];
        r.a[i][j] = s;
    }
    return r;
}

// product of a node's transform elements in document order (COLLADA 1.4 ch. 5: column vectors, <matrix> row-major)
bool node_matrix(const Doc& d, int node, M4& m, std::string& err)
{
    m = eye();
    std::vector<double> v;
    for (int c : d.nodes[node].kids) {
        const sv k = d.nodes[c].name;
        if (k != "matrix" && k != "translate" && k != "rotate" && k != "scale") continue;
        if (!parse_doubles(d.nodes[c].text, v)) { err = "COLLADA: malformed <" + std::string(k) + ">"; return false; }
        M4 t = eye();
        if (k == "matrix" && v.size() == 16) { for (int i = 0; i < 16; i++) t.a[i / 4][i % 4] = v[i]; }
        else if (k == "translate" && v.size() == 3) { t.a[0][3] = v[0]; t.a[1][3] = v[1]; t.a[2][3] = v[2]; }
        else if (k == "scale" && v.size() == 3) { t.a[0][0] = v[0]; t.a[1][1] = v[1]; t.a[2][2] = v[2]; }
        else if (k == "rotate" && v.size() == 4) {
            const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            if (len > 0.0) {
                const double x = v[0] / len, y = v[1] / len, z = v[2] / len, ang = v[3] * (M_PI / 180.0);
                const double co = std::cos(ang), si = std::sin(ang), C = 1.0 - co;
                t.a[0][0] = co + x * x * C;     t.a[0][1] = x * y * C - z * si; t.a[0][2] = x * z * C + y * si;
                t.a[1][0] = y * x * C + z * si; t.a[1][1] = co + y * y * C;     t.a[1][2] = y * z * C - x * si;
                t.a[2][0] = z * x * C - y * si; t.a[2][1] = z * y * C + x * si; t.a[2][2] = co + z * z * C;
            }
        } else { err = "COLLADA: malformed <" + std::string(k) + ">"; return false; }
        m = mul(m, t);
    }
    return true;
}

struct Group { std::string material; bool has_material = false; std::vector<double> pos; };   // pos: 9 doubles per triangle
struct Geom { std::string name; std::vector<Group> groups; };

void fan(const long long* c, size_t n, std::vector<long long>& tri)
{
    for (size_t k = 1; k + 1 < n; k++) { tri.push_back(c[0]); tri.push_back(c[k]); tri.push_back(c[k + 1]); }
}

}  // namespace

bool load_dae(const std::string& path, std::vector<float>& verts, std::vector<uint32_t>& faces, std::vector<uint32_t>& obj,
              std::vector<std::string>& names, std::string& err)
{
    Doc d;
    {
        std::ifstream f(path, std::ios::binary);
        if (!f) { err = path + ": cannot open"; return false; }
        f.seekg(0, std::ios::end);
        const std::streamoff sz = f.tellg();
        f.seekg(0, std::ios::beg);
        d.buf.resize((size_t)std::max<std::streamoff>(0, sz));
        if (sz > 0 && !f.read(&d.buf[0], sz)) { err = path + ": cannot read"; return false; }
    }
    std::string xe;
    if (!parse_xml(d, xe)) { err = path + ": not well-formed XML (" + xe + ")"; return false; }
    if (d.nodes[0].name != "COLLADA") { err = path + ": not a COLLADA document"; return false; }

    double unit = 1.0;
    if (const int asset = d.first(0, "asset"); asset >= 0) {
        if (const int u = d.first(asset, "unit"); u >= 0)
            if (const std::string* m = d.nodes[u].attr("meter"); m && !m->empty()) {
                char* q = nullptr;
                const double v = std::strtod(m->c_str(), &q);
                if (q == m->c_str()) { err = path + ": malformed <unit meter>"; return false; }
                unit = v;
            }
    }

    // ---- geometries -------------------------------------------------------------------------------------------
    std::map<std::string, Geom> geoms;                     // "#id" -> geometry
    std::vector<double> data; std::vector<long long> ids, counts, corner, tri;
    for (int lib : d.kids(0, "library_geometries")) for (int g : d.kids(lib, "geometry")) {
        const int mesh = d.first(g, "mesh");
        if (mesh < 0) continue;
        const std::string* gid = d.nodes[g].attr("id");
        const std::string* gname = d.nodes[g].attr("name");
        struct Src { std::vector<double> data; long long stride = 3, offset = 0; };
        std::map<std::string, Src> sources;
        for (int s : d.kids(mesh, "source")) {
            const int fa = d.first(s, "float_array");
            if (fa < 0) continue;
            Src src;
            if (!parse_doubles(d.nodes[fa].text, src.data)) { err = path + ": malformed <float_array>"; return false; }
            const int tc = d.first(s, "technique_common");
            const int acc = tc >= 0 ? d.first(tc, "accessor") : -1;
            if (acc >= 0) {
                const std::string* st = d.nodes[acc].attr("stride"); const std::string* of = d.nodes[acc].attr("offset");
                src.stride = st ? std::strtoll(st->c_str(), nullptr, 10) : 1;
                src.offset = of ? std::strtoll(of->c_str(), nullptr, 10) : 0;
            }
            const std::string* sid = d.nodes[s].attr("id");
            sources["#" + (sid ? *sid : std::string())] = std::move(src);
        }
        std::map<std::string, std::string> vert_pos;       // "#vertices id" -> "#source id"
        for (int vs : d.kids(mesh, "vertices")) for (int in : d.kids(vs, "input")) {
            const std::string* sem = d.nodes[in].attr("semantic"); const std::string* so = d.nodes[in].attr("source");
            const std::string* vid = d.nodes[vs].attr("id");
            if (sem && *sem == "POSITION" && so) vert_pos["#" + (vid ? *vid : std::string())] = *so;
        }
        Geom geo;
        geo.name = gname && !gname->empty() ? *gname : (gid ? *gid : std::string("None"));
        for (int prim : d.nodes[mesh].kids) {
            const sv kind = d.nodes[prim].name;
            if (kind != "triangles" && kind != "polylist" && kind != "polygons" && kind != "trifans" && kind != "tristrips") continue;
            long long n_off = 1, v_off = -1; const std::string* v_src = nullptr;
            for (int in : d.kids(prim, "input")) {
                const std::string* of = d.nodes[in].attr("offset");
                const long long o = of ? std::strtoll(of->c_str(), nullptr, 10) : 0;
                if (o < 0 || o > 255) { err = path + ": input offset out of range"; return false; }
                n_off = std::max(n_off, o + 1);
                const std::string* sem = d.nodes[in].attr("semantic");
                if (sem && *sem == "VERTEX" && v_off < 0) { v_off = o; v_src = d.nodes[in].attr("source"); }
            }
            if (v_off < 0) continue;
            const auto vp = v_src ? vert_pos.find(*v_src) : vert_pos.end();
            const auto sit = vp != vert_pos.end() ? sources.find(vp->second) : sources.end();
            if (sit == sources.end()) { err = path + ": geometry " + (gid ? *gid : std::string("?")) + " has no POSITION source"; return false; }
            const Src& src = sit->second;
            if (src.stride < 3) { err = path + ": POSITION stride < 3 in " + (gid ? *gid : std::string("?")); return false; }
            if (src.offset < 0 || (size_t)src.offset > src.data.size() || (size_t)src.stride > src.data.size() + 3) { err = path + ": accessor outside its array in " + (gid ? *gid : std::string("?")); return false; }
            auto corner_ids = [&](sv text) -> bool {        // the VERTEX index of every corner of one <p>
                if (!parse_ints(text, ids)) return false;
                corner.clear();
                for (size_t r = 0; (r + 1) * (size_t)n_off <= ids.size(); r++) corner.push_back(ids[r * (size_t)n_off + (size_t)v_off]);
                return true;
            };
            tri.clear();
            const std::vector<int> ps = d.kids(prim, "p");
            const std::string bad_p = path + ": malformed <p>";
            if (kind == "triangles") {
                for (int p : ps) {
                    if (!corner_ids(d.nodes[p].text)) { err = bad_p; return false; }
                    tri.insert(tri.end(), corner.begin(), corner.end());
                }
                tri.resize(tri.size() / 3 * 3);
            } else if (kind == "polylist") {
                const int vc = d.first(prim, "vcount");
                counts.clear();
                if (vc >= 0 && !parse_ints(d.nodes[vc].text, counts)) { err = path + ": malformed <vcount>"; return false; }
                if (!ps.empty()) { if (!corner_ids(d.nodes[ps[0]].text)) { err = bad_p; return false; } } else corner.clear();
                size_t k = 0;
                for (long long c : counts) {
                    if (c < 0) { err = path + ": negative <vcount>"; return false; }
                    if (k >= corner.size()) break;
                    const size_t take = std::min((size_t)c, corner.size() - k);
                    fan(corner.data() + k, take, tri);
                    k += take;
                }
            } else if (kind == "polygons" || kind == "trifans") {
                for (int p : ps) { if (!corner_ids(d.nodes[p].text)) { err = bad_p; return false; } fan(corner.data(), corner.size(), tri); }
            } else {                                        // tristrips
                for (int p : ps) {
                    if (!corner_ids(d.nodes[p].text)) { err = bad_p; return false; }
                    for (size_t k = 0; k + 2 < corner.size(); k++) {
                        if (k % 2 == 0) { tri.push_back(corner[k]); tri.push_back(corner[k + 1]); tri.push_back(corner[k + 2]); }
                        else { tri.push_back(corner[k + 1]); tri.push_back(corner[k]); tri.push_back(corner[k + 2]); }
                    }
                }
            }
            if (tri.empty()) continue;
            Group grp;
            if (const std::string* m = d.nodes[prim].attr("material")) { grp.material = *m; grp.has_material = true; }
            grp.pos.reserve(3 * tri.size());
            for (long long ix : tri) {
                if (ix < 0 || (size_t)ix > src.data.size() / (size_t)src.stride) { err = path + ": vertex index out of range in " + (gid ? *gid : std::string("?")); return false; }
                const long long at = src.offset + src.stride * ix;      // <= 2 * data.size(): no overflow
                if ((size_t)at + 3 > src.data.size()) { err = path + ": vertex index out of range in " + (gid ? *gid : std::string("?")); return false; }
                grp.pos.push_back(src.data[(size_t)at]); grp.pos.push_back(src.data[(size_t)at + 1]); grp.pos.push_back(src.data[(size_t)at + 2]);
            }
            geo.groups.push_back(std::move(grp));
        }
        geoms["#" + (gid ? *gid : std::string())] = std::move(geo);
    }

    // ---- nodes of <library_nodes> that <instance_node> may name ------------------------------------------------
    std::map<std::string, int> lib_nodes;
    for (int lib : d.kids(0, "library_nodes")) {
        std::vector<int> todo(1, lib);
        while (!todo.empty()) {
            const int e = todo.back(); todo.pop_back();
            if (d.nodes[e].name == "node") if (const std::string* id = d.nodes[e].attr("id"); id && !id->empty()) lib_nodes["#" + *id] = e;
            for (int c : d.nodes[e].kids) todo.push_back(c);
        }
    }

    // ---- the visual scene ----------------------------------------------------------------------------------------
    std::map<std::string, int> scenes; int first_scene = -1;
    for (int lib : d.kids(0, "library_visual_scenes")) for (int vs : d.kids(lib, "visual_scene")) {
        const std::string* id = d.nodes[vs].attr("id");
        scenes["#" + (id ? *id : std::string())] = vs;
        if (first_scene < 0) first_scene = vs;
    }
    int chosen = -1;
    if (const int sc = d.first(0, "scene"); sc >= 0)
        if (const int ivs = d.first(sc, "instance_visual_scene"); ivs >= 0)
            if (const std::string* url = d.nodes[ivs].attr("url")) { const auto it = scenes.find(*url); if (it != scenes.end()) chosen = it->second; }
    if (chosen < 0) chosen = first_scene;
    if (chosen < 0) { err = path + ": no visual scene"; return false; }

    const size_t budget = 64 * d.buf.size() + 4096;         // output triangles: instancing may repeat a geometry, not without bound
    size_t n_tri = 0, n_visits = 0;
    struct Frame { int node; M4 m; int depth; };
    // depth-first in document order with an explicit stack: children are pushed in reverse
    std::vector<Frame> stack;
    M4 top = eye();
    for (int i = 0; i < 3; i++) top.a[i][i] *= unit;
    {
        const std::vector<int> roots = d.kids(chosen, "node");
        for (size_t k = roots.size(); k-- > 0;) stack.push_back({ roots[k], top, 0 });
    }
    while (!stack.empty()) {
        const Frame fr = stack.back(); stack.pop_back();
        if (++n_visits > budget) { err = path + ": node hierarchy too large (cyclic instance_node?)"; return false; }
        if (fr.depth > 64) { err = path + ": node hierarchy too deep (cyclic instance_node?)"; return false; }
        M4 local;
        if (!node_matrix(d, fr.node, local, err)) { err = path + ": " + err; return false; }
        const M4 m = mul(fr.m, local);
        for (int ig : d.kids(fr.node, "instance_geometry")) {
            const std::string* url = d.nodes[ig].attr("url");
            const auto it = url ? geoms.find(*url) : geoms.end();
            if (it == geoms.end()) continue;
            const Geom& geo = it->second;
            for (const Group& grp : geo.groups) {
                n_tri += grp.pos.size() / 9;
                if (n_tri > budget || n_tri > 0x55555555u) { err = path + ": the scene instantiates more triangles than a file of this size can hold"; return false; }
                const uint32_t id = (uint32_t)names.size();
                for (size_t k = 0; k + 2 < grp.pos.size(); k += 3) {
                    const double x = grp.pos[k], y = grp.pos[k + 1], z = grp.pos[k + 2];
                    for (int r = 0; r < 3; r++) verts.push_back((float)(x * m.a[r][0] + y * m.a[r][1] + z * m.a[r][2] + m.a[r][3]));
                }
                for (size_t k = 0; k < grp.pos.size() / 9; k++) {
                    const uint32_t base = (uint32_t)(faces.size());
                    faces.push_back(base); faces.push_back(base + 1); faces.push_back(base + 2);
                    obj.push_back(id);
                }
                names.push_back(geo.groups.size() == 1 ? geo.name : geo.name + "[" + (grp.has_material ? grp.material : std::string("None")) + "]");
            }
        }
        // what follows the node's own geometries: its instance_nodes, then its child nodes (pushed in reverse)
        std::vector<std::pair<int, int>> next;              // (element, kind) in the order they must be visited
        for (int inn : d.kids(fr.node, "instance_node")) {
            const std::string* url = d.nodes[inn].attr("url");
            const auto it = url ? lib_nodes.find(*url) : lib_nodes.end();
            if (it != lib_nodes.end()) next.push_back({ it->second, 0 });
        }
        for (int c : d.kids(fr.node, "node")) next.push_back({ c, 1 });
        if (stack.size() + next.size() > 1000000) { err = path + ": node hierarchy too large"; return false; }
        for (size_t k = next.size(); k-- > 0;) stack.push_back({ next[k].first, m, fr.depth + 1 });
    }
    if (verts.empty()) { err = path + ": the visual scene instantiates no triangle geometry"; return false; }
    return true;
}

}  // namespace rr_collada
