// mcrt_host.hpp -- C++ host side above the C-ABI (include/mcrt.h), mirroring the reference's interface for the hot
// path: same class names, argument meaning and error behaviour, so a program written against the reference's
// `scene` / `transducer<N>` / `psf<...>` / `rf_image<...>` reads the same (namespace mcrt_host).
//
//   transducer<N>            transducer.h:24-137   (frequency MHz, radius cm, element separation mm, position, angles deg)
//   psf<ax,lat,elev,res>     psf.h:34-77
//   scene                    scene.h:19-76, scene.cpp:16-48,185-247 (JSON keys, "Error while loading scene: ..." wrapping)
//   rf_image<cols,us,um>     rfimage.h:20-219      (clear / convolve / envelope / postprocess; data lives on the GPU)
//   ray_physics::segment     ray.h:28-36
//
// Units are plain doubles (the reference's units.h types are compile-time only); names say the unit.
#pragma once
#include <mcrt.h>

#include <array>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
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

    static json parse(const std::string &s) { size_t i = 0; json j = value(s, i); ws(s, i); if (i != s.size()) fail("trailing characters", i); return j; }

private:
    static void fail(const char *m, size_t i) { throw std::invalid_argument("parse error at " + std::to_string(i) + ": " + m); }
    static void ws(const std::string &s, size_t &i) { while (i < s.size() && std::isspace((unsigned char)s[i])) i++; }
    static json value(const std::string &s, size_t &i)
    {
        ws(s, i);
        if (i >= s.size()) fail("unexpected end", i);
        json j;
        const char c = s[i];
        if (c == '{') {
            j.kind = obj_k; i++; ws(s, i);
            if (i < s.size() && s[i] == '}') { i++; return j; }
            for (;;) {
                ws(s, i);
                json k = value(s, i);
                if (k.kind != str_k) fail("object key must be a string", i);
                ws(s, i);
                if (i >= s.size() || s[i] != ':') fail("expected ':'", i);
                i++;
                j.obj.emplace_back(k.str, value(s, i));
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
                j.arr.push_back(value(s, i));
                ws(s, i);
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == ']') { i++; return j; }
                fail("expected ',' or ']'", i);
            }
        }
        if (c == '"') {
            j.kind = str_k; i++;
            while (i < s.size() && s[i] != '"') {
                if (s[i] == '\\' && i + 1 < s.size()) {
                    const char e = s[++i];
                    j.str += e == 'n' ? '\n' : e == 't' ? '\t' : e;
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

inline void load_obj_triangles(const std::string &path, std::vector<float> &tri9)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot read mesh '" + path + "'");
    std::vector<std::array<float, 3>> v;
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream is(line);
        std::string tag; is >> tag;
        if (tag == "v") {   // decimal -> double -> float, like the reference's (float)atof (tiny_obj_loader.cpp parseFloat)
            double d[3] = { 0, 0, 0 }; is >> d[0] >> d[1] >> d[2];
            v.push_back({ (float)d[0], (float)d[1], (float)d[2] });
        }
        else if (tag == "f") {
            std::vector<long> idx; std::string tok;
            while (is >> tok) { long i = std::strtol(tok.c_str(), nullptr, 10); idx.push_back(i > 0 ? i - 1 : (long)v.size() + i); }
            for (size_t k = 1; k + 1 < idx.size(); k++)
                for (long i : { idx[0], idx[k], idx[k + 1] }) {
                    if (i < 0 || (size_t)i >= v.size()) throw std::runtime_error("face index out of range in '" + path + "'");
                    tri9.insert(tri9.end(), v[(size_t)i].begin(), v[(size_t)i].end());
                }
        }
    }
}

namespace ray_physics { using segment = mcrt_segment; }   // ray.h:28-36; `media` is the material index

// ---------------------------------------------------------------- transducer<N> (transducer.h)
template <size_t transducer_elements>
class transducer {
public:
    struct transducer_element { std::array<float, 3> position, direction; };

    transducer(float frequency_mhz, double radius_cm, double element_separation_mm, const std::array<float, 3> &position, const std::array<float, 3> &angles_deg)
        : frequency(frequency_mhz), position(position), angles(angles_deg), radius_cm(radius_cm), separation_mm(element_separation_mm)
    {
        if (!(element_separation_mm * transducer_elements < 3.14159 * radius_cm * 10.0))      // the assert of transducer.h:35
            throw std::invalid_argument("transducer: elements do not fit on the arc");
        update();
    }
    void update()   // transducer.h:82-118
    {
        pos.resize(3 * transducer_elements); dir.resize(3 * transducer_elements);
        check(mcrt_transducer_elements((uint32_t)transducer_elements, radius_cm, separation_mm, position.data(), angles.data(), pos.data(), dir.data()), "transducer");
    }
    transducer_element element(size_t i) const
    {
        if (i >= transducer_elements) throw std::out_of_range("transducer::element");
        return { { pos[3 * i], pos[3 * i + 1], pos[3 * i + 2] }, { dir[3 * i], dir[3 * i + 1], dir[3 * i + 2] } };
    }
    void setPosition(const std::array<float, 3> &p) { position = p; }
    void setAngles(const std::array<float, 3> &a) { angles = a; }
    std::array<float, 3> getPosition() const { return position; }
    static constexpr size_t size() { return transducer_elements; }

    const float frequency;
    std::array<float, 3> position, angles;
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
struct device {
    explicit device(int id = 0) { check(mcrt_create(id, &ctx), "mcrt_create"); }
    ~device() { mcrt_destroy(ctx); }
    device(const device &) = delete; device &operator=(const device &) = delete;
    mcrt_ctx *ctx = nullptr;
};

// ---------------------------------------------------------------- scene (scene.h / scene.cpp)
class scene {
public:
    // scene(json, transducer&): parse_config + upload (replaces create_empty_world/init/add_rigidbody_from_obj)
    template <size_t N>
    scene(const json &config, transducer<N> &t, std::shared_ptr<device> dev, unsigned samples = 5, unsigned seed = 0x5EED) : dev(std::move(dev))
    {
        try { parse_config(config); }
        catch (const std::exception &ex) { throw std::runtime_error{ "Error while loading scene: " + std::string{ ex.what() } }; }
        mcrt_params p; mcrt_default_params(&p);
        p.n_elements = (uint32_t)N; p.n_samples = samples; p.frequency = t.frequency; p.seed = seed;
        check(mcrt_set_params(this->dev->ctx, &p), "mcrt_set_params");
        params = p;
        init();
        check(mcrt_upload_texture(this->dev->ctx, nullptr, p.tex_n), "mcrt_upload_texture");     // static volume_ texture_volume (main.cpp:52)
        set_transducer(t);
    }

    template <size_t N> void set_transducer(const transducer<N> &t) { check(mcrt_set_transducer(dev->ctx, t.pos.data(), t.dir.data(), (uint32_t)N), "mcrt_set_transducer"); }

    // cast_rays<sample_count, ray_count>(transducer): the segments of every (element, sample) path (scene.cpp:50-183)
    std::vector<std::vector<std::vector<ray_physics::segment>>> cast_rays(uint32_t frame_id = 0)
    {
        const size_t E = params.n_elements, S = params.n_samples, B = params.max_depth;
        std::vector<mcrt_segment> flat(E * S * B); std::vector<uint32_t> cnt(E * S);
        check(mcrt_cast_rays(dev->ctx, frame_id, 0, (uint32_t)E, flat.data(), cnt.data(), nullptr), "mcrt_cast_rays");
        std::vector<std::vector<std::vector<ray_physics::segment>>> out(E, std::vector<std::vector<ray_physics::segment>>(S));
        for (size_t e = 0; e < E; e++)
            for (size_t s = 0; s < S; s++) {
                const size_t p = e * S + s;
                out[e][s].assign(flat.begin() + (long)(p * B), flat.begin() + (long)(p * B + cnt[p]));
            }
        return out;
    }
    void step(float) {}   // scene.cpp:336-339: all bodies are static, nothing to integrate

    std::shared_ptr<device> dev;
    mcrt_params params{};
    std::vector<std::string> material_names;
    std::vector<material> materials;
    std::vector<mesh> meshes;
    std::string working_dir, starting_material;
    std::array<float, 3> spacing{}, origin{};
    float scaling = 1.f;

private:
    uint32_t material_index(const std::string &name) const
    {
        for (size_t i = 0; i < material_names.size(); i++) if (material_names[i] == name) return (uint32_t)i;
        throw std::out_of_range("key '" + name + "' not found");
    }
    void parse_config(const json &config)   // scene.cpp:185-247
    {
        working_dir = config.contains("workingDirectory") ? (std::string)config.at("workingDirectory") : "";
        (void)config.at("transducerPosition");
        const auto &orig = config.at("origin"); origin = { (float)orig[0], (float)orig[1], (float)orig[2] };
        const auto &spac = config.at("spacing"); spacing = { (float)spac[0], (float)spac[1], (float)spac[2] };
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
            const auto &d = m.at("deltas");
            meshes.push_back(mesh{ m.at("file"), m.at("rigid"), m.at("vascular"), { (float)d[0], (float)d[1], (float)d[2] }, m.at("outsideNormals"),
                                   material_index(m.at("material")), material_index(m.at("outsideMaterial")) });
        }
        (void)material_index(starting_material);
    }
    void init()   // scene.cpp:38-48 + 300-334: load each OBJ, place it, hand the triangle soup to the GPU
    {
        std::vector<float> tri; std::vector<uint32_t> tri_mesh; std::vector<mcrt_mesh> recs;
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
        check(mcrt_upload_scene(dev->ctx, tri.data(), tri_mesh.data(), (uint32_t)(tri.size() / 9), recs.data(), (uint32_t)recs.size(),
                                &materials[0].impedance, (uint32_t)materials.size(), material_index(starting_material), spacing.data()), "mcrt_upload_scene");
    }
};

// ---------------------------------------------------------------- rf_image (rfimage.h); the image lives on the GPU
template <unsigned int columns, unsigned int max_travel_time_us, unsigned int axial_resolution_um, unsigned int speed_of_sound = 1500>
class rf_image {
public:
    static constexpr unsigned int max_rows = (speed_of_sound * max_travel_time_us) / axial_resolution_um;   // rfimage.h:180

    rf_image(std::shared_ptr<device> dev, double radius_mm, double angle_rad) : dev(std::move(dev)), radius_mm(radius_mm), angle(angle_rad)
    {
        check(mcrt_alloc(this->dev->ctx, sizeof(float) * columns * max_rows, (void **)&rf_dev), "mcrt_alloc");
        check(mcrt_alloc(this->dev->ctx, sizeof(float) * 400 * 500, (void **)&scan_dev), "mcrt_alloc");
    }
    ~rf_image() { mcrt_free(dev->ctx, rf_dev); mcrt_free(dev->ctx, scan_dev); }

    // clear() + cast_rays + the accumulation loop of main.cpp:102-144 in one call
    void trace(uint32_t frame_id) { check(mcrt_trace_frame(dev->ctx, frame_id, 0, columns, rf_dev), "mcrt_trace_frame"); }
    template <typename psf_> void convolve(const psf_ &p)
    {
        check(mcrt_convolve(dev->ctx, rf_dev, columns, max_rows, p.axial_kernel.data(), (uint32_t)p.get_axial_size(), p.lateral_kernel.data(), (uint32_t)p.get_lateral_size()), "mcrt_convolve");
    }
    void envelope() { check(mcrt_envelope(dev->ctx, rf_dev, columns, max_rows), "mcrt_envelope"); }
    void postprocess() { check(mcrt_scan_convert(dev->ctx, rf_dev, columns, max_rows, radius_mm, angle, scan_dev, 400, 500), "mcrt_scan_convert"); }
    std::vector<float> intensities() const   // row-major [max_rows][columns], the cv::Mat of rfimage.h:217
    {
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
    double radius_mm, angle;
    float *rf_dev = nullptr, *scan_dev = nullptr;
};

}  // namespace mcrt_host
