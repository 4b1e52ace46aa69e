// host_sanitize_driver -- TEST code (not part of the product): the product's host-side C++ that faces user files and user sizes, run under
// AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on the pool).  Built by tests/test_host_sanitize.py from
//     csrc/mcrt_host.cpp   (SAH builder + BVH4 collapse, row thresholds, texture, PSF taps, transducer geometry, scan-conversion maps)
//     host/mcrt_host.hpp   (the JSON reader, the OBJ reader, scene_config::load = scene::parse_config + the reference's error wrapping)
// with `g++ -fsanitize=address,undefined`; nothing here needs a GPU or links the HIP runtime.
//     host_sanitize_driver <tricky.obj> <scene.json>
// Prints one `name: result` line per case; a sanitizer report aborts the run.  Behaviour to match: scene.cpp:19-26,185-247,
// tiny_obj_loader.cpp:97-187,504-717 (positions + faces).
#include "mcrt_host.hpp"

#include <cinttypes>
#include <cstdio>
#include <limits>
#include <random>
#include <sstream>

using namespace mcrt_host;

static uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

template <class Fn> static std::string outcome(Fn &&fn)
{
    try { return fn(); }
    catch (const std::exception &ex) { return std::string("error: ") + ex.what(); }
}

static std::string printable(const std::string &s)      // one line per case: control characters of a case's own text are shown escaped
{
    std::string o;
    for (unsigned char c : s) { if (c < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\x%02x", c); o += b; } else o += (char)c; }
    return o;
}
static void report(const std::string &name, const std::string &res) { std::printf("%s: %s\n", printable(name).c_str(), printable(res).c_str()); }

static std::string obj_case(const std::string &text)
{
    return outcome([&] {
        std::istringstream is(text);
        std::vector<float> t9;
        load_obj_triangles(is, "<case>", t9);
        char buf[96];
        int nan = 0; for (float x : t9) if (x != x) nan++;
        std::snprintf(buf, sizeof buf, "ok %zu triangles, %d NaN, fnv %016" PRIx64, t9.size() / 9, nan, fnv(t9.data(), t9.size() * 4));
        return std::string(buf);
    });
}

// the SAH builder + BVH4 collapse over a soup: sizes and a structural walk (every triangle id exactly once, references in range)
static std::string bvh_case(const std::vector<float> &tri9)
{
    const uint32_t n = (uint32_t)(tri9.size() / 9);
    std::vector<uint32_t> tm(n, 0u);
    mcrt_bvh b2{}; mcrt_bvh4 b4{};
    if (mcrt_build_bvh(tri9.data(), tm.data(), n, &b2) != 0) return std::string("error: ") + mcrt_last_error();
    std::string res;
    if (mcrt_build_bvh4(&b2, &b4) != 0) res = std::string("error: ") + mcrt_last_error();
    else {
        std::vector<int> seen(n, 0); bool ok = true; uint64_t leaves = 0;
        for (uint32_t k = 0; k < b4.n_nodes && ok; k++)
            for (int c = 0; c < 4; c++) {
                const int32_t ref = b4.nodes[k].c[c].ref;
                if (ref == MCRT_BVH4_EMPTY) continue;
                if (ref >= 0) { if ((uint32_t)ref >= b4.n_nodes) ok = false; continue; }
                const uint32_t v = (uint32_t)~ref, first = v >> 3, cnt = (v & 7u) + 1u;
                if (first + cnt > n) { ok = false; break; }
                for (uint32_t t = 0; t < cnt; t++) { uint32_t id; std::memcpy(&id, &b2.tri[(size_t)(first + t) * 12 + 3], 4); if (id >= n) ok = false; else seen[id]++; }
                leaves++;
            }
        for (uint32_t t = 0; t < n && ok; t++) if (seen[t] != 1) ok = false;
        char buf[160];
        std::snprintf(buf, sizeof buf, "%s bvh2 %u nodes depth %u, bvh4 %u nodes stack %u, %" PRIu64 " leaves", ok ? "ok" : "BROKEN", b2.n_nodes, b2.max_depth, b4.n_nodes, b4.max_stack, leaves);
        res = buf;
    }
    mcrt_free_bvh4(&b4); mcrt_free_bvh(&b2);
    return res;
}

int main(int argc, char **argv)
{
    if (argc != 3) { std::printf("usage: host_sanitize_driver <tricky.obj> <scene.json>\n"); return 99; }

    // ------------------------------------------------------------------------------------------ OBJ reader
    std::vector<float> tricky;
    report("obj.tricky", outcome([&] { load_obj_triangles(argv[1], tricky); char b[64]; std::snprintf(b, sizeof b, "ok %zu triangles, fnv %016" PRIx64, tricky.size() / 9, fnv(tricky.data(), tricky.size() * 4)); return std::string(b); }));
    report("obj.missing_file", outcome([&] { std::vector<float> t; load_obj_triangles("/nonexistent/mesh.obj", t); return std::string("ok"); }));
    report("obj.empty", obj_case(""));
    report("obj.forms", obj_case("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1\nf 1//1 2//1 4//1\nf 1/1 3/1 4/1\nf 2 3 4\n"));
    report("obj.negative", obj_case("v 0 0 0\nv 1 0 0\nv 0 1 0\nf -3 -2 -1\nv 0 0 1\nf -1 -2 -3\n"));
    report("obj.index_zero_is_first_vertex", obj_case("v 5 6 7\nv 1 0 0\nv 0 1 0\nf 0 2 3\n"));           // fixIndex, tiny_obj_loader.cpp:97-109
    report("obj.index_past_end", obj_case("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n"));
    report("obj.index_before_start", obj_case("v 0 0 0\nv 1 0 0\nv 0 1 0\nf -4 1 2\n"));
    report("obj.index_huge", obj_case("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 99999999999999999999 1 2\nf -99999999999999999999 1 2\n"));
    report("obj.nan_and_inf", obj_case("v nan 0 0\nv 1 inf 0\nv 0 1 -inf\nv 1 1 1\nf 1 2 3\nf 2 3 4\n"));
    report("obj.short_lines", obj_case("v\nv 1\nv 1 2\nf\nf 1\nf 1 2\nv 0 0 0\nf 1 2 3 4\n"));
    report("obj.crlf_tabs_polygon", obj_case("v\t0 0 0\r\nv 1 0 0\r\nv 1 1 0\r\nv 0 1 0\r\nv 0.5 2 0\r\nf\t1 2 3 4 5\r\n# comment\r\ng group\r\nusemtl m\r\n"));
    report("obj.garbage", obj_case(std::string("v 1e400 -1e400 1e-400\nf a b c\nf 1/ /2 //\nf 1 1 1\n") + std::string(70000, 'f') + "\nf " + std::string(70000, '7') + " 1 1\n"));
    {
        std::string bin; std::mt19937 g(7); for (int i = 0; i < 20000; i++) bin += (char)(g() & 0xff);
        report("obj.random_bytes", obj_case(bin));
    }

    // ------------------------------------------------------------------------------------------ JSON reader
    std::string scene_text;
    { std::ifstream f(argv[2]); std::stringstream ss; ss << f.rdbuf(); scene_text = ss.str(); }
    auto scene_case = [&](const std::string &text) {
        return outcome([&] {
            const scene_config c = scene_config::load(json::parse(text));
            char b[96]; std::snprintf(b, sizeof b, "ok %zu materials, %zu meshes, start %u", c.materials.size(), c.meshes.size(), c.material_index(c.starting_material));
            return std::string(b);
        });
    };
    report("json.scene", scene_case(scene_text));
    {   // every prefix of the file: a parse error or (never) a scene; no crash, no hang
        size_t errors = 0, parsed = 0;
        for (size_t n = 0; n < scene_text.size(); n++) { const std::string r = scene_case(scene_text.substr(0, n)); if (r.rfind("error:", 0) == 0) errors++; else parsed++; }
        char b[64]; std::snprintf(b, sizeof b, "%zu errors, %zu parsed of %zu prefixes", errors, parsed, scene_text.size());
        report("json.truncated_everywhere", b);
    }
    {   // one byte damaged at every position
        size_t errors = 0, parsed = 0; std::mt19937 g(11);
        for (size_t n = 0; n < scene_text.size(); n++) { std::string t = scene_text; t[n] = (char)(g() & 0xff); const std::string r = scene_case(t); if (r.rfind("error:", 0) == 0) errors++; else parsed++; }
        char b[64]; std::snprintf(b, sizeof b, "%zu errors, %zu parsed", errors, parsed);
        report("json.one_byte_damaged_everywhere", b);
    }
    report("json.nested_arrays_1e6", outcome([] { (void)json::parse(std::string(1000000, '[')); return std::string("ok"); }));
    report("json.nested_objects_1e5", outcome([] { std::string s; for (int i = 0; i < 100000; i++) s += "{\"a\":"; (void)json::parse(s); return std::string("ok"); }));
    report("json.nested_at_limit", outcome([] { const std::string s = std::string(256, '[') + "1" + std::string(256, ']'); (void)json::parse(s); return std::string("ok"); }));
    report("json.escapes", outcome([] { const json j = json::parse("\"a\\u00e9\\u20ac\\ud83d\\ude00\\n\\t\\\\\\/\\\"\\b\\f\\r\""); return "ok fnv " + std::to_string(fnv(j.str.data(), j.str.size())) + " bytes " + std::to_string(j.str.size()); }));
    for (const char *bad : { "\"\\ud83d\"", "\"\\ud83dx\"", "\"\\ude00\"", "\"\\u12\"", "\"\\u12g4\"", "\"\\q\"", "\"abc", "\"a\nb\"", "\"\\", "nan", "inf", "-inf", "+1", "0x10", "01", "-", ".5", "1 2", "{\"a\" 1}", "{1:2}", "[1,]", "{\"a\":1,}", "", "   ", "tru", "nul" })
        report(std::string("json.bad ") + bad, outcome([&] { (void)json::parse(bad); return std::string("ok"); }));
    for (const char *good : { "0", "-0", "1e5", "-1.5E-3", "0.25", "[]", "{}", "[[],{}]", " \t\r\n[1, 2,3 ]\n", "{\"a\":{\"b\":[true,false,null]}}" })
        report(std::string("json.good ") + good, outcome([&] { (void)json::parse(good); return std::string("ok"); }));
    // scene-level errors, wrapped the reference's way (scene.cpp:19-26)
    const json base = json::parse(scene_text);
    auto without = [&](const std::string &key) { json j = base; for (size_t i = 0; i < j.obj.size(); i++) if (j.obj[i].first == key) { j.obj.erase(j.obj.begin() + (long)i); break; } return j; };
    for (const char *key : { "transducerPosition", "origin", "spacing", "startingMaterial", "scaling", "materials", "meshes" })
        report(std::string("scene.missing ") + key, outcome([&] { (void)scene_config::load(without(key)); return std::string("ok"); }));
    report("scene.without_workingDirectory", outcome([&] { (void)scene_config::load(without("workingDirectory")); return std::string("ok"); }));
    auto with = [&](const std::string &key, const std::string &value) { json j = without(key); j.obj.emplace_back(key, json::parse(value)); return j; };
    report("scene.materials_not_array", outcome([&] { (void)scene_config::load(with("materials", "{}")); return std::string("ok"); }));
    report("scene.meshes_not_array", outcome([&] { (void)scene_config::load(with("meshes", "3")); return std::string("ok"); }));
    report("scene.scaling_string", outcome([&] { (void)scene_config::load(with("scaling", "\"big\"")); return std::string("ok"); }));
    report("scene.origin_short", outcome([&] { (void)scene_config::load(with("origin", "[1,2]")); return std::string("ok"); }));
    report("scene.origin_number", outcome([&] { (void)scene_config::load(with("origin", "7")); return std::string("ok"); }));
    report("scene.unknown_starting_material", outcome([&] { (void)scene_config::load(with("startingMaterial", "\"UNOBTAINIUM\"")); return std::string("ok"); }));
    report("scene.material_without_shininess", outcome([&] { (void)scene_config::load(with("materials", "[{\"name\":\"GEL\",\"impedance\":1,\"attenuation\":1,\"mu0\":0,\"mu1\":0,\"sigma\":0,\"specularity\":1,\"thickness\":0}]")); return std::string("ok"); }));
    report("scene.mesh_unknown_material", outcome([&] { (void)scene_config::load(with("meshes", "[{\"file\":\"a.obj\",\"rigid\":true,\"vascular\":false,\"deltas\":[0,0,0],\"outsideNormals\":true,\"material\":\"NOPE\",\"outsideMaterial\":\"GEL\"}]")); return std::string("ok"); }));
    report("scene.mesh_file_missing", outcome([&] {
        scene_config c = scene_config::load(with("meshes", "[{\"file\":\"/nonexistent/a.obj\",\"rigid\":true,\"vascular\":false,\"deltas\":[0,0,0],\"outsideNormals\":true,\"material\":\"GEL\",\"outsideMaterial\":\"GEL\"}]"));
        std::vector<float> tri; std::vector<uint32_t> tm; std::vector<mcrt_mesh> recs; c.triangles(tri, tm, recs); return std::string("ok"); }));

    // ------------------------------------------------------------------------------------------ host tables and builders (csrc/mcrt_host.cpp)
    report("bvh.tricky", bvh_case(tricky));
    report("bvh.zero_triangles", bvh_case({}));
    report("bvh.one_triangle", bvh_case({ 0, 0, 0, 1, 0, 0, 0, 1, 0 }));
    { std::vector<float> t; for (int i = 0; i < 5000; i++) for (float x : { 1.f, 2.f, 3.f, 2.f, 2.f, 3.f, 1.f, 3.f, 3.f }) t.push_back(x); report("bvh.5000_identical", bvh_case(t)); }
    { std::vector<float> t(9 * 300, 0.0f); report("bvh.300_points_at_origin", bvh_case(t)); }
    {
        std::mt19937 g(3); std::uniform_real_distribution<float> u(-10.f, 10.f);
        std::vector<float> t(9 * 20000); for (float &x : t) x = u(g);
        report("bvh.random_20000", bvh_case(t));
        std::vector<float> tn = t; const float nan = std::numeric_limits<float>::quiet_NaN(), inf = std::numeric_limits<float>::infinity();
        for (size_t i = 0; i < tn.size(); i += 97) tn[i] = (i / 97) % 3 == 0 ? nan : (i / 97) % 3 == 1 ? inf : -inf;
        report("bvh.random_20000_with_nan_inf", bvh_case(tn));
        std::vector<float> th = t; for (float &x : th) x *= 1e30f;
        report("bvh.random_20000_times_1e30", bvh_case(th));
        std::vector<float> tl = t; for (size_t i = 0; i < tl.size(); i += 9) for (int k = 3; k < 9; k++) tl[i + (size_t)k] = tl[i + (size_t)(k % 3)] + 1e-30f * (float)k;
        report("bvh.random_20000_degenerate", bvh_case(tl));
    }
    { mcrt_bvh b{}; report("bvh.null_arguments", mcrt_build_bvh(nullptr, nullptr, 5, &b) != 0 ? std::string("error: ") + mcrt_last_error() : "ok"); }
    { mcrt_bvh4 b4{}; report("bvh4.null_arguments", mcrt_build_bvh4(nullptr, &b4) != 0 ? std::string("error: ") + mcrt_last_error() : "ok"); }
    {
        std::vector<double> thr(466);
        const int rc = mcrt_row_thresholds(145.0 / 1500.0 * 0.001 * 1000.0, 465, thr.data());
        report("tables.row_thresholds", rc ? std::string("error: ") + mcrt_last_error() : "ok fnv " + std::to_string(fnv(thr.data(), thr.size() * 8)));
        report("tables.row_thresholds_bad", mcrt_row_thresholds(0.0, 465, thr.data()) ? std::string("error: ") + mcrt_last_error() : "ok");
    }
    {
        std::vector<float> vox(2 * 8 * 8 * 8);
        report("tables.texture_8", mcrt_generate_texture(vox.data(), 8) ? std::string("error: ") + mcrt_last_error() : "ok fnv " + std::to_string(fnv(vox.data(), vox.size() * 4)));
        report("tables.texture_null", mcrt_generate_texture(nullptr, 8) ? std::string("error: ") + mcrt_last_error() : "ok");
    }
    {
        float ax[7], lat[13];
        report("tables.psf", mcrt_psf_kernels(4.5f, 0.05f, 0.2f, 145, ax, 7, lat, 13) ? std::string("error: ") + mcrt_last_error() : "ok fnv " + std::to_string(fnv(ax, sizeof ax, fnv(lat, sizeof lat))));
    }
    {
        const float pos0[3] = { -13.5f, 0, 0 }, ang[3] = { 0, 0, -90 };
        std::vector<float> pos(3 * 512), dir(3 * 512);
        report("tables.transducer_512", mcrt_transducer_elements(512, 3.0, 60.0 * 3.14159265358979323846 / 180.0 * 3.0 / 512 * 10.0, pos0, ang, pos.data(), dir.data()) ? std::string("error: ") + mcrt_last_error()
                                                                                                     : "ok fnv " + std::to_string(fnv(pos.data(), pos.size() * 4, fnv(dir.data(), dir.size() * 4))));
        report("tables.transducer_zero", mcrt_transducer_elements(0, 3.0, 0.1, pos0, ang, pos.data(), dir.data()) ? std::string("error: ") + mcrt_last_error() : "ok");
    }
    {
        std::vector<float> mr(40 * 50), mc(40 * 50);
        report("tables.scan_maps", mcrt_scan_maps(128, 465, 30.0, 1.0471975511965976, 100, 1500, 40, 50, mr.data(), mc.data()) ? std::string("error: ") + mcrt_last_error()
                                                                                                            : "ok fnv " + std::to_string(fnv(mr.data(), mr.size() * 4, fnv(mc.data(), mc.size() * 4))));
        report("tables.scan_maps_bad", mcrt_scan_maps(0, 465, 30.0, 1.0, 100, 1500, 40, 50, mr.data(), mc.data()) ? std::string("error: ") + mcrt_last_error() : "ok");
    }
    std::printf("DONE\n");
    return 0;
}
