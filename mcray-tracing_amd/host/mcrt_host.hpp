// mcrt_host.hpp -- C++ host side above the C-ABI (include/mcrt.h), mirroring the reference's interface for the hot
// path: same class names, argument meaning and error behaviour, so a program written against the reference's
// `scene` / `transducer<N>` / `psf<...>` / `rf_image<...>` reads the same (namespace mcrt_host).
//
//   transducer<N>            transducer.h:24-137   (frequency MHz, radius cm, element separation mm, position, angles deg)
//   psf<ax,lat,elev,res>     psf.h:34-77
//   scene                    scene.h:19-76, scene.cpp:16-48,185-247 (JSON keys, "Error while loading scene: ..." wrapping)
//   rf_image<cols,us,um>     rfimage.h:20-219      (clear / convolve / envelope / postprocess; data lives on the GPU)
//   ray_physics::segment     ray.h:28-36           (vec3 stands in for btVector3; `media` is held BY VALUE: the reference's
//                                                   `const material &` dangles by the time main.cpp:126 reads it)
//   volume<size,res>         volume.h:19-61        (host copy of the texture the GPU samples)
//
// Units are plain doubles (the reference's units.h types are compile-time only); names say the unit.
#pragma once
#include <mcrt.h>

#include <array>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace mcrt_host {

inline void check(int rc, const char *what)
{
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + mcrt_last_error());
}

// ---------------------------------------------------------------- minimal JSON (objects, arrays, strings, numbers, bools)
struct json {
    enum kind_t { null_k, bool_k, num_k, str_k, arr_k, obj_k } kind = null_k;
    bool b = false; double num = 0; std::string str; std::vector<json> arr; std::vector<std::pair<std::string, json>> obj;

    const json &at(const std::string &key) const   // nlohmann::json::at semantics: throws when absent
    {
        for (auto &kv : obj) if (kv.first == key) return kv.second;
        throw std::out_of_range("key '" + key + "' not found");
    }
    bool contains(const std::string &key) const { for (auto &kv : obj) if (kv.first == key) return true; return false; }
    const json &operator[](size_t i) const { return arr.at(i); }
    bool is_array() const { return kind == arr_k; }
    operator double() const { if (kind != num_k) throw std::domain_error("type must be number"); return num; }
    operator float() const { return (float)(double)*this; }
    operator bool() const { if (kind != bool_k) throw std::domain_error("type must be boolean"); return b; }
    operator std::string() const { if (kind != str_k) throw std::domain_error("type must be string"); return str; }

    static json parse(const std::string &s) { size_t i = 0; json j = value(s, i, 0); ws(s, i); if (i != s.size()) fail("trailing characters", i); return j; }

    static constexpr int max_depth = 256;      // nesting the recursive descent accepts (a scene file nests 3 deep); deeper input is a parse error, not a stack overflow

private:
    static void fail(const char *m, size_t i) { throw std::invalid_argument("parse error at " + std::to_string(i) + ": " + m); }
    static void ws(const std::string &s, size_t &i) { while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) i++; }
    static unsigned hex4(const std::string &s, size_t &i)      // the four hex digits after \u; i is left on the last one
    {
        if (i + 4 >= s.size()) fail("truncated \\u escape", i);
        unsigned v = 0;
        for (int k = 1; k <= 4; k++) {
            const char h = s[i + (size_t)k];
            v = v * 16u + (h >= '0' && h <= '9' ? (unsigned)(h - '0') : h >= 'a' && h <= 'f' ? (unsigned)(h - 'a' + 10) : h >= 'A' && h <= 'F' ? (unsigned)(h - 'A' + 10) : (fail("bad hex digit in \\u escape", i + (size_t)k), 0u));
        }
        i += 4;
        return v;
    }
    static void utf8(std::string &o, unsigned cp)
    {
        if (cp < 0x80) o += (char)cp;
        else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
        else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    }
    static json value(const std::string &s, size_t &i, int depth)
    {
        if (depth > max_depth) fail("nested too deeply", i);
        ws(s, i);
        if (i >= s.size()) fail("unexpected end", i);
        json j;
        const char c = s[i];
        if (c == '{') {
            j.kind = obj_k; i++; ws(s, i);
            if (i < s.size() && s[i] == '}') { i++; return j; }
            for (;;) {
                ws(s, i);
                json k = value(s, i, depth + 1);
                if (k.kind != str_k) fail("object key must be a string", i);
                ws(s, i);
                if (i >= s.size() || s[i] != ':') fail("expected ':'", i);
                i++;
                j.obj.emplace_back(k.str, value(s, i, depth + 1));
                ws(s, i);
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == '}') { i++; return j; }
                fail("expected ',' or '}'", i);
            }
        }
        if (c == '[') {
            j.kind = arr_k; i++; ws(s, i);
            if (i < s.size() && s[i] == ']') { i++; return j; }
            for (;;) {
                j.arr.push_back(value(s, i, depth + 1));
                ws(s, i);
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == ']') { i++; return j; }
                fail("expected ',' or ']'", i);
            }
        }
        if (c == '"') {
            j.kind = str_k; i++;
            while (i < s.size() && s[i] != '"') {
                if ((unsigned char)s[i] < 0x20) fail("control character in string", i);
                if (s[i] == '\\') {
                    if (i + 1 >= s.size()) fail("unterminated string", i);
                    const char e = s[++i];
                    switch (e) {
                    case '"': case '\\': case '/': j.str += e; break;
                    case 'b': j.str += '\b'; break; case 'f': j.str += '\f'; break; case 'n': j.str += '\n'; break;
                    case 'r': j.str += '\r'; break; case 't': j.str += '\t'; break;
                    case 'u': {
                        unsigned cp = hex4(s, i);
                        if (cp >= 0xD800 && cp <= 0xDBFF) {                      // a surrogate pair: the low half must follow
                            if (i + 2 >= s.size() || s[i + 1] != '\\' || s[i + 2] != 'u') fail("lone surrogate in \\u escape", i);
                            i += 2;
                            const unsigned lo = hex4(s, i);
                            if (lo < 0xDC00 || lo > 0xDFFF) fail("bad low surrogate in \\u escape", i);
                            cp = 0x10000u + ((cp - 0xD800u) << 10) + (lo - 0xDC00u);
                        } else if (cp >= 0xDC00 && cp <= 0xDFFF) fail("lone surrogate in \\u escape", i);
                        utf8(j.str, cp);
                        break;
                    }
                    default: fail("bad escape", i);
                    }
                } else j.str += s[i];
                i++;
            }
            if (i >= s.size()) fail("unterminated string", i);
            i++;
            return j;
        }
        if (!s.compare(i, 4, "true")) { j.kind = bool_k; j.b = true; i += 4; return j; }
        if (!s.compare(i, 5, "false")) { j.kind = bool_k; j.b = false; i += 5; return j; }
        if (!s.compare(i, 4, "null")) { i += 4; return j; }
        {   // a JSON number: -? digits ...  (strtod alone would take "nan", "inf", hex floats and a leading '+')
            const size_t d = i + (c == '-' ? 1u : 0u);
            if (d >= s.size() || !(s[d] >= '0' && s[d] <= '9')) fail("unexpected character", i);
            if (s[d] == '0' && d + 1 < s.size() && ((s[d + 1] >= '0' && s[d + 1] <= '9') || s[d + 1] == 'x' || s[d + 1] == 'X')) fail("bad number", i);
        }
        char *end = nullptr;
        j.num = std::strtod(s.c_str() + i, &end);
        if (end == s.c_str() + i) fail("unexpected character", i);
        j.kind = num_k; i = (size_t)(end - s.c_str());
        return j;
    }
};

inline json load_json(const std::string &path)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::stringstream ss; ss << f.rdbuf();
    return json::parse(ss.str());
}

// ---------------------------------------------------------------- meshes (mesh.h:7-20) and OBJ triangles (objloader.h:28-139: positions, face order, fans)
struct material { float impedance, attenuation, mu0, mu1, sigma, specularity, shininess, thickness; };
struct mesh { std::string filename; bool is_rigid, is_vascular; std::array<float, 3> deltas; bool outside_normals; uint32_t material_inside, material_outside; };

// positions and faces the way the reference's loader reads them (tiny_obj_loader.cpp:97-187,504-717): a coordinate is (float)atof of its token
// (so "nan" and "inf" are values, a missing coordinate is 0), a face corner is atoi of the token up to its first '/', index > 0 counts from one,
// < 0 from the end, and 0 is the first vertex (fixIndex); polygons become fans.  The reference then indexes its arrays unchecked: a corner
// outside the vertices read so far is an error here, never a wild read.
inline void load_obj_triangles(std::istream &f, const std::string &name, std::vector<float> &tri9)
{
    std::vector<std::array<float, 3>> v;
    std::string line;
    while (std::getline(f, line)) {
        const char *t = line.c_str();
        t += std::strspn(t, " \t");
        if (t[0] == 'v' && (t[1] == ' ' || t[1] == '\t')) {
            t += 2;
            std::array<float, 3> p{};
            for (int k = 0; k < 3; k++) { t += std::strspn(t, " \t"); p[(size_t)k] = (float)std::atof(t); t += std::strcspn(t, " \t\r"); }
            v.push_back(p);
        } else if (t[0] == 'f' && (t[1] == ' ' || t[1] == '\t')) {
            t += 2;
            std::vector<long> idx;
            for (;;) {
                t += std::strspn(t, " \t");
                if (t[0] == '\0' || t[0] == '\r' || t[0] == '\n') break;
                const long i = std::strtol(t, nullptr, 10);            // stops at '/': the position index of v, v/vt, v//vn, v/vt/vn
                idx.push_back(i > 0 ? i - 1 : i == 0 ? 0 : (long)v.size() + i);
                t += std::strcspn(t, " \t\r");
            }
            for (size_t k = 1; k + 1 < idx.size(); k++)
                for (long i : { idx[0], idx[k], idx[k + 1] }) {
                    if (i < 0 || (size_t)i >= v.size()) throw std::runtime_error("face index out of range in '" + name + "'");
                    tri9.insert(tri9.end(), v[(size_t)i].begin(), v[(size_t)i].end());
                }
        }
    }
}
inline void load_obj_triangles(const std::string &path, std::vector<float> &tri9)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot read mesh '" + path + "'");
    load_obj_triangles(f, path, tri9);
}

// the subset of btVector3 the reference's host code uses (main.cpp:72,117,120,131; scene.cpp:342-346)
struct vec3 {
    float v[3] = { 0, 0, 0 };
    vec3() = default;
    vec3(float x, float y, float z) : v{ x, y, z } {}
    float x() const { return v[0]; } float y() const { return v[1]; } float z() const { return v[2]; }
    vec3 operator+(const vec3 &o) const { return { v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2] }; }
    vec3 operator-(const vec3 &o) const { return { v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2] }; }
    vec3 &operator+=(const vec3 &o) { v[0] += o.v[0]; v[1] += o.v[1]; v[2] += o.v[2]; return *this; }
    float dot(const vec3 &o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
    float length() const { return std::sqrt(dot(*this)); }
    float distance(const vec3 &o) const { return (o - *this).length(); }
};
inline vec3 operator*(float s, const vec3 &a) { return { s * a.v[0], s * a.v[1], s * a.v[2] }; }
inline vec3 operator*(const vec3 &a, float s) { return s * a; }

namespace ray_physics {
struct segment {   // ray.h:28-36
    vec3 from, to, direction;
    float reflected_intensity;   // reflected back to the transducer, at the end of the segment
    float initial_intensity, attenuation;
    double distance_traveled;    // [mm] traveled from the transducer to the beginning of the segment
    material media;              // by value (see the header comment)
    int32_t tri;                 // triangle hit at the end of the segment (-1 none): not in the reference, free with the GPU walk
};
}  // namespace ray_physics

// ---------------------------------------------------------------- transducer<N> (transducer.h)
template <size_t transducer_elements>
class transducer {
public:
    struct transducer_element { vec3 position, direction; };

    // transducer.h:24-62 (frequency MHz, radius cm, element separation mm, position in scene units, angles in degrees)
    transducer(float frequency_mhz, double radius_cm, double element_separation_mm, const vec3 &position, const std::array<float, 3> &angles_deg)
        : frequency(frequency_mhz), position(position), angles(angles_deg), radius_cm(radius_cm), separation_mm(element_separation_mm)
    {
        if (!(element_separation_mm * transducer_elements < 3.14159 * radius_cm * 10.0))      // the assert of transducer.h:35
            throw std::invalid_argument("transducer: elements do not fit on the arc");
        update();
    }
    void update()   // transducer.h:82-118: the elements from the current position and angles
    {
        pos.resize(3 * transducer_elements); dir.resize(3 * transducer_elements);
        check(mcrt_transducer_elements((uint32_t)transducer_elements, radius_cm, separation_mm, position.v, angles.data(), pos.data(), dir.data()), "transducer");
    }
    transducer_element element(size_t i) const   // transducer.h:64-67 (std::array::at: throws when out of range)
    {
        if (i >= transducer_elements) throw std::out_of_range("transducer::element");
        return { vec3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), vec3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]) };
    }
    void print(bool direction) const   // transducer.h:69-80: "x,z" per element
    {
        for (size_t i = 0; i < transducer_elements; i++) {
            const auto e = element(i);
            const vec3 &v = direction ? e.direction : e.position;
            std::cout << v.x() << "," << v.z() << std::endl;
        }
    }
    void setPosition(const vec3 &p) { position = p; }
    void setAngles(const std::array<float, 3> &a) { angles = a; }
    vec3 getPosition() const { return position; }
    static constexpr size_t size() { return transducer_elements; }

    const float frequency;
    vec3 position, direction;
    std::array<float, 3> angles;
    std::vector<float> pos, dir;          // [N][3] each, what mcrt_set_transducer takes
private:
    const double radius_cm, separation_mm;
};

// ---------------------------------------------------------------- psf (psf.h)
template <size_t axial_size, size_t lateral_size, size_t elevation_size, unsigned int resolution_micrometers>
class psf {
    static_assert(axial_size % 2 && lateral_size % 2 && elevation_size % 2, "kernel sizes must be odd");
public:
    psf(float freq, float var_x, float var_y, float /*var_z*/)
    {
        check(mcrt_psf_kernels(freq, var_x, var_y, resolution_micrometers, axial_kernel.data(), axial_size, lateral_kernel.data(), lateral_size), "psf");
    }
    constexpr size_t get_axial_size() const { return axial_size; }
    constexpr size_t get_lateral_size() const { return lateral_size; }
    constexpr size_t get_elevation_size() const { return elevation_size; }
    std::array<float, axial_size> axial_kernel;
    std::array<float, lateral_size> lateral_kernel;
    std::array<float, elevation_size> elevation_kernel{};   // declared and never filled in the reference (psf.h:77)
};

// ---------------------------------------------------------------- GPU context shared by scene and rf_image
// One GPU (mcrt_ctx), or several behind the same objects (mcrt_group: scan-line shards, blocks gathered on the first device; a
// device may be listed more than once).  `ctx` is the context the images live on -- the group's root -- in either case.
struct device {
    explicit device(int id = 0) { check(mcrt_create(id, &ctx), "mcrt_create"); }
    explicit device(const std::vector<int> &ids)
    {
        if (ids.size() == 1) { check(mcrt_create(ids[0], &ctx), "mcrt_create"); return; }
        check(mcrt_group_create(ids.data(), (uint32_t)ids.size(), &group), "mcrt_group_create");
        ctx = mcrt_group_root(group);
    }
    ~device() { if (group) mcrt_group_destroy(group); else mcrt_destroy(ctx); }
    device(const device &) = delete; device &operator=(const device &) = delete;
    mcrt_ctx *tracer() const { return group ? mcrt_group_member(group, 0) : ctx; }      // for calls on one shard (scene::cast_rays)
    // the set-up and trace calls, on the one context or on every rank of the group
    int set_params(const mcrt_params *p) { return group ? mcrt_group_set_params(group, p) : mcrt_set_params(ctx, p); }
    int upload_scene(const float *tri, const uint32_t *tri_mesh, uint32_t n_tri, const mcrt_mesh *meshes, uint32_t n_mesh, const float *mats, uint32_t n_mat, uint32_t start_mat, const float *spacing)
    {
        return group ? mcrt_group_upload_scene(group, tri, tri_mesh, n_tri, meshes, n_mesh, mats, n_mat, start_mat, spacing)
                     : mcrt_upload_scene(ctx, tri, tri_mesh, n_tri, meshes, n_mesh, mats, n_mat, start_mat, spacing);
    }
    int upload_texture(const float *vox, uint32_t n) { return group ? mcrt_group_upload_texture(group, vox, n) : mcrt_upload_texture(ctx, vox, n); }
    int set_transducer(const float *pos, const float *dir, uint32_t n) { return group ? mcrt_group_set_transducer(group, pos, dir, n) : mcrt_set_transducer(ctx, pos, dir, n); }
    int trace_frames(uint32_t frame, uint32_t n_frames, uint32_t columns, float *rf_dev)
    {
        return group ? mcrt_group_trace_frames(group, frame, n_frames, rf_dev) : mcrt_trace_frames(ctx, frame, n_frames, 0, columns, rf_dev);
    }
    int synchronize() { return group ? mcrt_group_synchronize(group) : mcrt_synchronize(ctx); }
    mcrt_ctx *ctx = nullptr;
    mcrt_group *group = nullptr;
};
// the reference's objects take no device argument: they share this process-wide one (GPU 0), created on first use
inline std::shared_ptr<device> default_device()
{
    static std::weak_ptr<device> weak;
    auto d = weak.lock();
    if (!d) { d = std::make_shared<device>(0); weak = d; }
    return d;
}

// ---------------------------------------------------------------- volume (volume.h): host copy of the tissue texture
template <unsigned int size, unsigned int resolution_micrometers>
class volume {
public:
    volume() : matrix((size_t)size * size * size * 2) { check(mcrt_generate_texture(matrix.data(), size), "mcrt_generate_texture"); }   // volume.h:19-35
    constexpr float get_resolution_in_millis() const { return static_cast<float>(resolution_micrometers) / 1000.0f; }
    // volume.h:46-61 (float -> unsigned of a negative coordinate wraps like x86-64: DESIGN.md, quirk 4)
    float get_scattering(const float scattering_density, const float scattering_mu, const float scattering_sigma,
                         const float x_millis, const float y_millis, const float z_millis) const
    {
        constexpr float resolution = resolution_micrometers / 1000.0f;
        const unsigned int x = index(x_millis / resolution), y = index(y_millis / resolution), z = index(z_millis / resolution);
        const float *voxel = &matrix[2 * (((size_t)x * size + y) * size + z)];     // { texture_noise, scattering_probability }
        return voxel[1] >= scattering_density ? voxel[0] * scattering_sigma + scattering_mu : 0.0f;
    }
    const float *data() const { return matrix.data(); }
private:
    static unsigned int index(float q)
    {
        long long i = std::fabs(q) < 9.2233720368547758e18f ? (long long)q : (long long)0x8000000000000000ull;
        return (unsigned int)i % size;
    }
    std::vector<float> matrix;
};

// ---------------------------------------------------------------- scene (scene.h / scene.cpp)
// what scene::parse_config (scene.cpp:185-247) reads of a scene file -- host data only, no GPU: every key but workingDirectory is mandatory
// (nlohmann::json::at throws), and load() wraps any failure the way the reference's constructor does (scene.cpp:19-26)
struct scene_config {
    std::vector<std::string> material_names;
    std::vector<material> materials;
    std::vector<mesh> meshes;
    std::string working_dir, starting_material;
    std::array<float, 3> spacing{}, origin{};
    float scaling = 1.f;

    static scene_config load(const json &config)
    {
        scene_config c;
        try { c.parse_config(config); }
        catch (const std::exception &ex) { throw std::runtime_error{ "Error while loading scene: " + std::string{ ex.what() } }; }
        return c;
    }
    uint32_t material_index(const std::string &name) const
    {
        for (size_t i = 0; i < material_names.size(); i++) if (material_names[i] == name) return (uint32_t)i;
        throw std::out_of_range("key '" + name + "' not found");
    }
    // scene.cpp:38-48 + 300-334: each mesh's OBJ, placed; the triangle soup mcrt_upload_scene takes
    void triangles(std::vector<float> &tri, std::vector<uint32_t> &tri_mesh, std::vector<mcrt_mesh> &recs) const
    {
        for (size_t mi = 0; mi < meshes.size(); mi++) {
            const mesh &m = meshes[mi];
            std::vector<float> t9;
            load_obj_triangles(working_dir + m.filename, t9);
            float pos[3];
            for (int i = 0; i < 3; i++) pos[i] = m.deltas[(size_t)i] * scaling * scaling + origin[(size_t)i];   // scene.cpp:322-324
            for (size_t k = 0; k < t9.size(); k++) t9[k] = t9[k] * scaling + pos[k % 3];
            tri.insert(tri.end(), t9.begin(), t9.end());
            tri_mesh.insert(tri_mesh.end(), t9.size() / 9, (uint32_t)mi);
            recs.push_back(mcrt_mesh{ m.material_inside, m.material_outside, m.is_vascular ? 1u : 0u, 0u });
        }
    }
private:
    static std::array<float, 3> float3(const json &a)
    {
        if (!a.is_array()) throw std::domain_error("type must be array");
        return { (float)a[0], (float)a[1], (float)a[2] };
    }
    void parse_config(const json &config)   // scene.cpp:185-247
    {
        working_dir = config.contains("workingDirectory") ? (std::string)config.at("workingDirectory") : "";
        (void)config.at("transducerPosition");
        origin = float3(config.at("origin"));
        spacing = float3(config.at("spacing"));
        starting_material = (std::string)config.at("startingMaterial");
        scaling = (float)config.at("scaling");
        const auto &mats = config.at("materials");
        if (!mats.is_array()) throw std::runtime_error("materials must be an array");
        for (const auto &m : mats.arr) {
            const std::string name = m.at("name");
            material v{ m.at("impedance"), m.at("attenuation"), m.at("mu0"), m.at("mu1"), m.at("sigma"), m.at("specularity"), m.at("shininess"), m.at("thickness") };
            bool found = false;
            for (size_t i = 0; i < material_names.size(); i++) if (material_names[i] == name) { materials[i] = v; found = true; }
            if (!found) { material_names.push_back(name); materials.push_back(v); }
        }
        const auto &ms = config.at("meshes");
        if (!ms.is_array()) throw std::runtime_error("meshes must be an array");
        for (const auto &m : ms.arr) {
            // (every field into a local first: an initializer that throws half way through a braced aggregate leaks the fields already built
            //  under GCC < 12 -- found by the sanitizer run, tests/test_host_sanitize.py)
            const std::string file = m.at("file");
            const bool rigid = m.at("rigid"), vascular = m.at("vascular"), outside_normals = m.at("outsideNormals");
            const std::array<float, 3> deltas = float3(m.at("deltas"));
            const uint32_t inside = material_index(m.at("material")), outside = material_index(m.at("outsideMaterial"));
            meshes.push_back(mesh{ file, rigid, vascular, deltas, outside_normals, inside, outside });
        }
        (void)material_index(starting_material);
    }
};

class scene : public scene_config {
public:
    // scene(json, transducer&): parse_config + upload (replaces create_empty_world/init/add_rigidbody_from_obj)
    template <size_t N>
    scene(const json &config, transducer<N> &t, std::shared_ptr<device> dev_ = nullptr, unsigned samples = 5, unsigned seed = 0x5EED)
        : scene_config(scene_config::load(config)), dev(dev_ ? std::move(dev_) : default_device())
    {
        mcrt_params p; mcrt_default_params(&p);
        p.n_elements = (uint32_t)N; p.n_samples = samples; p.frequency = t.frequency; p.seed = seed;
        check(this->dev->set_params(&p), "mcrt_set_params");
        params = p;
        init();
        check(this->dev->upload_texture(nullptr, p.tex_n), "mcrt_upload_texture");     // static volume_ texture_volume (main.cpp:52)
        set_transducer(t);
    }

    template <size_t N> void set_transducer(const transducer<N> &t) { check(dev->set_transducer(t.pos.data(), t.dir.data(), (uint32_t)N), "mcrt_set_transducer"); }

    // scene::cast_rays<sample_count, ray_count>(transducer) (scene.h:29-30, scene.cpp:50-183): the segments of every
    // (element, sample) path, traced on the GPU.  The reference draws fresh random numbers on every call (random_device);
    // here every call advances the frame id of the counter-based generator.
    template <unsigned int sample_count, unsigned int ray_count, size_t N>
    std::array<std::array<std::vector<ray_physics::segment>, sample_count>, ray_count> cast_rays(transducer<N> &t)
    {
        static_assert(ray_count == N, "one ray bundle per transducer element");
        if (params.n_samples != sample_count || params.n_elements != ray_count) {
            check(mcrt_get_params(dev->ctx, &params), "mcrt_get_params");
            params.n_samples = sample_count; params.n_elements = ray_count;
            check(dev->set_params(&params), "mcrt_set_params");
        }
        set_transducer(t);
        const size_t B = params.max_depth;
        std::vector<mcrt_segment> flat((size_t)ray_count * sample_count * B); std::vector<uint32_t> cnt((size_t)ray_count * sample_count);
        check(mcrt_cast_rays(dev->tracer(), frame_id++, 0, ray_count, flat.data(), cnt.data(), nullptr), "mcrt_cast_rays");
        std::array<std::array<std::vector<ray_physics::segment>, sample_count>, ray_count> out;
        for (size_t e = 0; e < ray_count; e++)
            for (size_t s = 0; s < sample_count; s++) {
                const size_t p = e * sample_count + s;
                auto &dst = out[e][s];
                dst.reserve(cnt[p]);
                for (uint32_t b = 0; b < cnt[p]; b++) {
                    const mcrt_segment &g = flat[p * B + b];
                    dst.push_back(ray_physics::segment{ vec3(g.from[0], g.from[1], g.from[2]), vec3(g.to[0], g.to[1], g.to[2]), vec3(g.dir[0], g.dir[1], g.dir[2]),
                                                        g.reflected_intensity, g.initial_intensity, g.attenuation, g.distance_traveled, materials[(size_t)g.media], g.tri });
                }
            }
        return out;
    }
    // the flat form of the same call, for a given frame id
    std::vector<std::vector<std::vector<mcrt_segment>>> cast_rays(uint32_t frame)
    {
        const size_t E = params.n_elements, S = params.n_samples, B = params.max_depth;
        std::vector<mcrt_segment> flat(E * S * B); std::vector<uint32_t> cnt(E * S);
        check(mcrt_cast_rays(dev->tracer(), frame, 0, (uint32_t)E, flat.data(), cnt.data(), nullptr), "mcrt_cast_rays");
        std::vector<std::vector<std::vector<mcrt_segment>>> out(E, std::vector<std::vector<mcrt_segment>>(S));
        for (size_t e = 0; e < E; e++)
            for (size_t s = 0; s < S; s++) {
                const size_t p = e * S + s;
                out[e][s].assign(flat.begin() + (long)(p * B), flat.begin() + (long)(p * B + cnt[p]));
            }
        return out;
    }
    void step(float) {}   // scene.cpp:336-339: all bodies are static, nothing to integrate
    double distance(const vec3 &from, const vec3 &to) const { return (double)(from.distance(to) * 10.0f); }   // [mm], scene.cpp:342-346

    std::shared_ptr<device> dev;
    mcrt_params params{};
    uint32_t frame_id = 0;

private:
    void init()   // scene.cpp:38-48 + 300-334: load each OBJ, place it, hand the triangle soup to the GPU
    {
        std::vector<float> tri; std::vector<uint32_t> tri_mesh; std::vector<mcrt_mesh> recs;
        triangles(tri, tri_mesh, recs);
        if (materials.empty()) throw std::runtime_error("Error while loading scene: no materials");
        check(dev->upload_scene(tri.data(), tri_mesh.data(), (uint32_t)(tri.size() / 9), recs.data(), (uint32_t)recs.size(),
                                &materials[0].impedance, (uint32_t)materials.size(), material_index(starting_material), spacing.data()), "mcrt_upload_scene");
    }
};

// ---------------------------------------------------------------- rf_image (rfimage.h)
// Two ways to fill it, both the reference's semantics:
//   trace(frame)            clear + cast_rays + the accumulation loop of main.cpp:102-144 fused on the GPU (the fast path)
//   clear() / add_echo()    the reference's own host-side accumulation (rfimage.h:33-40,161-164) into a host image [max_rows][columns]
// convolve / envelope / postprocess always run on the GPU: a host-accumulated image is uploaded first.
template <unsigned int columns, unsigned int max_travel_time_us, unsigned int axial_resolution_um, unsigned int speed_of_sound = 1500>
class rf_image {
public:
    static constexpr unsigned int max_rows = (speed_of_sound * max_travel_time_us) / axial_resolution_um;   // rfimage.h:180

    rf_image(double radius_mm, double angle_rad, std::shared_ptr<device> dev_ = nullptr)
        : dev(dev_ ? std::move(dev_) : default_device()), radius_mm(radius_mm), angle(angle_rad), host((size_t)columns * max_rows, 0.0f)
    {
        std::cout << "rf_image: " << max_rows << ", " << columns << std::endl;          // rfimage.h:30
        check(mcrt_alloc(this->dev->ctx, sizeof(float) * columns * max_rows, (void **)&rf_dev), "mcrt_alloc");
        check(mcrt_alloc(this->dev->ctx, sizeof(float) * 400 * 500, (void **)&scan_dev), "mcrt_alloc");
    }
    rf_image(std::shared_ptr<device> dev_, double radius_mm, double angle_rad) : rf_image(radius_mm, angle_rad, std::move(dev_)) {}
    ~rf_image() { mcrt_free(dev->ctx, rf_dev); mcrt_free(dev->ctx, scan_dev); }
    rf_image(const rf_image &) = delete; rf_image &operator=(const rf_image &) = delete;

    // rfimage.h:33-40: row = micros / (axial_resolution / speed_of_sound), integer micrometres over um/us
    void add_echo(const unsigned int column, const float echo, const double micros_from_source)
    {
        to_host();
        const double row = micros_from_source / ((double)axial_resolution_um / (double)speed_of_sound);
        if (row < (double)max_rows) host[(size_t)(int)row * columns + column] += echo;
    }
    constexpr double get_dt() const { return (double)axial_resolution_um / (double)speed_of_sound; }                      // [us] rfimage.h:43-46
    constexpr double micros_traveled(double microm_from_source) const { return microm_from_source / (double)speed_of_sound; }   // rfimage.h:48-51
    void clear() { std::fill(host.begin(), host.end(), 0.0f); where = on_host; }                                       // rfimage.h:161-164
    void print(size_t column) const                                                                                       // rfimage.h:166-173
    {
        const auto img = intensities();
        for (size_t i = 0; i < max_rows; i++) std::cout << img[i * columns + column] << ", ";
        std::cout << std::endl;
    }

    // clear() + cast_rays + the accumulation loop of main.cpp:102-144 in one call, on the GPU
    void trace(uint32_t frame_id)
    {
        mcrt_params p; check(mcrt_get_params(dev->ctx, &p), "mcrt_get_params");
        if (p.n_rows != max_rows || p.n_elements != columns) {       // the image's shape is the kernel's: rows from THIS image's template arguments
            p.n_rows = max_rows; p.n_elements = columns; p.speed_of_sound = speed_of_sound;
            check(dev->set_params(&p), "mcrt_set_params");
        }
        check(dev->trace_frames(frame_id, 1, columns, rf_dev), "mcrt_trace_frame");       // (every GPU of a group traces its scan-line shard)
        where = on_device;
    }
    template <typename psf_> void convolve(const psf_ &p)
    {
        to_device();
        check(mcrt_convolve(dev->ctx, rf_dev, columns, max_rows, p.axial_kernel.data(), (uint32_t)p.get_axial_size(), p.lateral_kernel.data(), (uint32_t)p.get_lateral_size()), "mcrt_convolve");
    }
    void envelope() { to_device(); check(mcrt_envelope(dev->ctx, rf_dev, columns, max_rows), "mcrt_envelope"); }
    void postprocess() { to_device(); check(mcrt_scan_convert(dev->ctx, rf_dev, columns, max_rows, radius_mm, angle, scan_dev, 400, 500), "mcrt_scan_convert"); }
    void show() const {}   // rfimage.h:150-158 opens an OpenCV window and blocks on a key: out of scope
    std::vector<float> intensities() const   // row-major [max_rows][columns], the cv::Mat of rfimage.h:217
    {
        if (where == on_host) return host;
        std::vector<float> h((size_t)columns * max_rows);
        check(mcrt_export_rf(dev->ctx, rf_dev, columns, max_rows, h.data()), "mcrt_export_rf");
        return h;
    }
    std::vector<float> scan_converted() const
    {
        std::vector<float> h(400 * 500);
        check(mcrt_memcpy_d2h(dev->ctx, h.data(), scan_dev, h.size() * 4), "mcrt_memcpy_d2h");
        return h;
    }
    void save(const std::string &filename) const   // rfimage.h:142-148 writes an 8-bit image; here: binary PGM
    {
        const auto img = scan_converted();
        std::ofstream f(filename, std::ios::binary);
        f << "P5\n500 400\n255\n";
        for (float v : img) { float x = v * 255.0f; unsigned char c = (unsigned char)(x != x || x < 0 ? 0 : x > 255 ? 255 : x); f.put((char)c); }
    }
    std::shared_ptr<device> dev;
private:
    void to_device()
    {
        if (where == on_host) { check(mcrt_import_rf(dev->ctx, host.data(), columns, max_rows, rf_dev), "mcrt_import_rf"); where = on_device; }
    }
    void to_host()
    {
        if (where == on_device) { check(mcrt_export_rf(dev->ctx, rf_dev, columns, max_rows, host.data()), "mcrt_export_rf"); where = on_host; }
    }
    double radius_mm, angle;
    std::vector<float> host;                     // [max_rows][columns]
    enum { on_host, on_device } where = on_host;
    float *rf_dev = nullptr, *scan_dev = nullptr;
};

}  // namespace mcrt_host
