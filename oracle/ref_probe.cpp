// ref_probe.cpp -- ORACLE-SIDE TOOL (test infrastructure).
//
// Compiles the parts of the REFERENCE that build without Bullet/OpenCV, from the sources
// where they lie under /root/reference (nothing is copied into this repo), and prints the
// values the oracle is pinned against as JSON:
//   src/psf.h            -> PSF taps                         (main.cpp:54 parameters)
//   src/volume.h         -> tissue texture (hash, sums, samples, get_scattering probes)
//   include/units/units.h-> the unit arithmetic of main.cpp:23-37,114-139 and rfimage.h:33-51,178-180, and the scan-conversion maps
//                           of rfimage.h:183-215 (create_mapping) evaluated with the reference's own unit types
//   include/nlohmann/json.hpp + examples/**/*.scene -> the fields scene::parse_config (scene.cpp:185-247) and main.cpp:62-72 read,
//                           converted as the reference converts them (json number -> float), for every scene file that loads
// Built by oracle/Makefile into oracle/_ref/ref_probe (git-ignored); run by oracle/gen_golden.py,
// which writes tests/golden/ref_probe.json.  Only this container has /root/reference.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>

#define private public           // the probe reads volume::matrix directly
#include "volume.h"
#undef private
#include "psf.h"                 // note: redefines M_PI as 3.14159 (psf.h:9)
#include <units/units.h>
#include <nlohmann/json.hpp>
#include <fstream>
#include <string>

using namespace units::literals;
using namespace units::velocity;
using namespace units::length;
using namespace units::time;
using namespace units::angle;

// main.cpp:23-33 verbatim semantics (these are declarations of the reference's constants, re-typed
// here because main.cpp itself cannot be compiled: it includes Bullet and OpenCV headers)
constexpr meters_per_second_t speed_of_sound = 1500_m / 1_s;
constexpr float transducer_frequency = 4.5f;
constexpr millimeter_t axial_resolution = millimeter_t(1.45f / transducer_frequency);
constexpr size_t transducer_elements = 512;
constexpr radian_t transducer_amplitude = 60_deg;
constexpr centimeter_t transducer_radius = 3_cm;
constexpr centimeter_t ultrasound_depth = 15_cm;
constexpr microsecond_t max_travel_time = microsecond_t(ultrasound_depth / speed_of_sound);

static uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// rfimage.h:48-51 / :33-40 / :178-180 restated with the reference's unit types
template <unsigned int axial_resolution_um, unsigned int sos>
struct rf_axis {
    static constexpr meters_per_second_t speed_of_sound_ = meters_per_second_t(sos);
    static constexpr micrometer_t axial_resolution_ = micrometer_t(axial_resolution_um);
    static microsecond_t micros_traveled(micrometer_t um) { return um / speed_of_sound_; }
    static double row(microsecond_t t) { const units::dimensionless::dimensionless_t r = t / (axial_resolution_ / speed_of_sound_); return r; }
    static double dt() { microsecond_t d = axial_resolution_ / speed_of_sound_; return d(); }
};
template <unsigned int a, unsigned int s> constexpr meters_per_second_t rf_axis<a, s>::speed_of_sound_;
template <unsigned int a, unsigned int s> constexpr micrometer_t rf_axis<a, s>::axial_resolution_;

// rfimage.h:183-215 create_mapping with the reference's template parameters and unit types; cv::Mat (absent here) is replaced by
// plain float arrays, `scan_converted.rows / .cols` by ints as cv::Mat's are.  Every expression keeps the reference's operand types:
// max_travel_time, speed_of_sound and rf_height / rf_width are `unsigned int`, radius a millimeter_t, total_angle a radian_t.
template <unsigned int max_travel_time_, unsigned int speed_of_sound_>
struct scan_maps {
    int rows, cols;
    std::vector<float> map_x, map_y;
    float ratio_out; double shift_y_out;
    scan_maps(millimeter_t radius, radian_t total_angle, unsigned int rf_width, unsigned int rf_height, int out_rows, int out_cols)
        : rows(out_rows), cols(out_cols), map_x((size_t)out_rows * out_cols), map_y((size_t)out_rows * out_cols)
    {
        constexpr unsigned int max_travel_time = max_travel_time_, speed_of_sound = speed_of_sound_;
        float ratio = (max_travel_time * speed_of_sound * 0.001f + radius.to<float>() - radius.to<float>() * std::cos(total_angle.to<float>()/2.0)) / rows;
        millimeter_t shift_y = radius * std::cos(total_angle.to<float>() / 2.0f);
        float half_width = (float)cols / 2.0f;
        ratio_out = ratio; shift_y_out = shift_y();
        for (int j = 0; j < cols; j++)
            for (int i = 0; i < rows; i++) {
                float fi = static_cast<float>(i)+shift_y.to<float>()/ratio;
                float fj = static_cast<float>(j)-half_width;
                float r = std::sqrt(std::pow(fi,2.0f) + std::pow(fj,2.0f));
                radian_t angle = radian_t(std::atan2(fj, fi));
                map_x[(size_t)i * cols + j] = (r*ratio-radius.to<float>())/(max_travel_time*speed_of_sound*0.001f) * (float)rf_height;
                map_y[(size_t)i * cols + j] = ((angle - (-total_angle/2)) / (total_angle)) * (float)rf_width;
            }
    }
};
static uint64_t fnv_floats(const std::vector<float> &v)
{
    uint64_t h = 1469598103934665603ull;
    for (float f : v) { uint32_t b = fbits(f); for (int k = 0; k < 4; k++) { h ^= (b >> (8 * k)) & 0xff; h *= 1099511628211ull; } }
    return h;
}
template <unsigned int T, unsigned int S>
static void dump_maps(const char *name, millimeter_t radius, radian_t total_angle, unsigned int rf_width, unsigned int rf_height, int out_rows, int out_cols)
{
    const scan_maps<T, S> m(radius, total_angle, rf_width, rf_height, out_rows, out_cols);
    printf("\"%s\": {\"max_travel_time\":%u,\"speed_of_sound\":%u,\"radius_mm\":%.17g,\"total_angle\":%.17g,\"rf_width\":%u,\"rf_height\":%u,\"rows\":%d,\"cols\":%d,",
           name, T, S, radius(), total_angle(), rf_width, rf_height, out_rows, out_cols);
    printf("\"ratio_bits\":%u,\"shift_y\":%.17g,\"map_x_fnv1a64\":\"%016llx\",\"map_y_fnv1a64\":\"%016llx\",\"samples\":[",
           fbits(m.ratio_out), m.shift_y_out, (unsigned long long)fnv_floats(m.map_x), (unsigned long long)fnv_floats(m.map_y));
    uint64_t s = 0xD1B54A32D192ED03ull;
    for (int k = 0; k < 24; k++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const int i = k == 0 ? 0 : k == 1 ? out_rows - 1 : (int)(s % (uint64_t)out_rows), j = k == 0 ? 0 : k == 1 ? out_cols - 1 : (int)((s >> 32) % (uint64_t)out_cols);
        printf("%s[%d,%d,%u,%u]", k ? "," : "", i, j, fbits(m.map_x[(size_t)i * out_cols + j]), fbits(m.map_y[(size_t)i * out_cols + j]));
    }
    printf("]},\n");
}

// the fields scene::parse_config (scene.cpp:185-247) and main.cpp:62-72 read from a .scene file, in the reference's own conversions
// (float x = json number), as one JSON object; floats as bit patterns.  A file the reference could not load (json.at throws, e.g.
// ircad11.scene lacks shininess / thickness) is reported as {"error": ...}.
static void dump_scene(const char *path)
{
    try {
        std::ifstream in(path);
        if (!in) { printf("{\"error\":\"cannot open\"}"); return; }
        nlohmann::json j;
        in >> j;
        std::string out = "{";
        auto f3 = [&](const char *key) {
            const auto &v = j.at(key);
            char buf[128]; const float a = v[0], b = v[1], c = v[2];
            snprintf(buf, sizeof buf, "\"%s\":[%u,%u,%u],", key, fbits(a), fbits(b), fbits(c));
            out += buf;
        };
        f3("transducerPosition"); f3("transducerAngles"); f3("origin"); f3("spacing");
        { char buf[96]; const float sc = j.at("scaling"); snprintf(buf, sizeof buf, "\"scaling\":%u,", fbits(sc)); out += buf; }
        { const std::string sm = j.at("startingMaterial"); out += "\"startingMaterial\":\"" + sm + "\","; }
        out += "\"materials\":[";
        bool first = true;
        for (const auto &m : j.at("materials")) {
            const std::string name = m.at("name");
            const float v[8] = { m.at("impedance"), m.at("attenuation"), m.at("mu0"), m.at("mu1"), m.at("sigma"), m.at("specularity"), m.at("shininess"), m.at("thickness") };
            char buf[256];
            snprintf(buf, sizeof buf, "%s[\"%s\",%u,%u,%u,%u,%u,%u,%u,%u]", first ? "" : ",", name.c_str(), fbits(v[0]), fbits(v[1]), fbits(v[2]), fbits(v[3]), fbits(v[4]), fbits(v[5]), fbits(v[6]), fbits(v[7]));
            out += buf; first = false;
        }
        out += "],\"meshes\":[";
        first = true;
        for (const auto &m : j.at("meshes")) {
            const std::string file = m.at("file"), mat = m.at("material"), outm = m.at("outsideMaterial");
            const bool rigid = m.at("rigid"), vasc = m.at("vascular"), on = m.at("outsideNormals");
            const auto &d = m.at("deltas");
            const float d0 = d[0], d1 = d[1], d2 = d[2];
            char buf[512];
            snprintf(buf, sizeof buf, "%s[\"%s\",%d,%d,%u,%u,%u,%d,\"%s\",\"%s\"]", first ? "" : ",", file.c_str(), (int)rigid, (int)vasc, fbits(d0), fbits(d1), fbits(d2), (int)on, mat.c_str(), outm.c_str());
            out += buf; first = false;
        }
        out += "]}";
        fputs(out.c_str(), stdout);
    } catch (const std::exception &ex) {
        std::string msg = ex.what();
        for (auto &ch : msg) if (ch == '"' || ch == '\\') ch = '\'';
        printf("{\"error\":\"%s\"}", msg.c_str());
    }
}

int main(int argc, char **argv)
{
    printf("{\n");
    // ---- the reference's example scenes through its own JSON reader (argv: the .scene files, as name=path) ----
    printf("\"scenes\": {");
    for (int i = 1; i < argc; i++) {
        const char *eq = strchr(argv[i], '=');
        if (!eq) continue;
        printf("%s\"%.*s\": ", i > 1 ? "," : "", (int)(eq - argv[i]), argv[i]);
        dump_scene(eq + 1);
    }
    printf("},\n");
    // ---- psf (main.cpp:54: psf<7,13,7,145>{4.5f, 0.05f, 0.2f, 0.1f}) ----
    {
        const psf<7, 13, 7, 145> p{ transducer_frequency, 0.05f, 0.2f, 0.1f };
        printf("\"psf_axial_bits\": [");
        for (size_t i = 0; i < 7; i++) printf("%s%u", i ? "," : "", fbits(p.axial_kernel[i]));
        printf("],\n\"psf_lateral_bits\": [");
        for (size_t i = 0; i < 13; i++) printf("%s%u", i ? "," : "", fbits(p.lateral_kernel[i]));
        printf("],\n");
        const psf<5, 9, 7, 200> p2{ 3.0f, 0.1f, 0.3f, 0.1f };
        printf("\"psf2_axial_bits\": [");
        for (size_t i = 0; i < 5; i++) printf("%s%u", i ? "," : "", fbits(p2.axial_kernel[i]));
        printf("],\n\"psf2_lateral_bits\": [");
        for (size_t i = 0; i < 9; i++) printf("%s%u", i ? "," : "", fbits(p2.lateral_kernel[i]));
        printf("],\n");
    }
    // ---- constants ----
    {
        const unsigned int ar_um = static_cast<unsigned int>(axial_resolution.to<float>() * 1000.0f);
        using axis = rf_axis<322, 1500>;
        printf("\"axial_resolution_mm\": %.17g,\n", axial_resolution());
        printf("\"axial_resolution_um\": %u,\n", ar_um);
        printf("\"max_travel_time_us\": %.17g,\n", max_travel_time());
        printf("\"max_travel_time_uint\": %u,\n", max_travel_time.to<unsigned int>());
        printf("\"max_rows\": %u,\n", (1500u * max_travel_time.to<unsigned int>()) / ar_um);
        printf("\"time_step_us\": %.17g,\n", axis::micros_traveled(axial_resolution)());
        printf("\"row_dt_us\": %.17g,\n", axis::dt());
        printf("\"amplitude_rad\": %.17g,\n", transducer_amplitude());
        millimeter_t sep = transducer_amplitude.to<float>() * transducer_radius / transducer_elements;   // main.cpp:66
        printf("\"element_separation_mm\": %.17g,\n", sep());
        auto amp = sep / transducer_radius;                                                                // transducer.h:41
        printf("\"amp_float_bits\": %u,\n", fbits(amp.to<float>()));
        printf("\"radius_float_bits\": %u,\n", fbits(transducer_radius.to<float>()));
        // angle walk of transducer.h:42-59 (the float handed to sin/cos for each element)
        const radian_t amplitude{ amp.to<float>() };
        const radian_t angle_center{ amplitude / 2.0f };
        radian_t angle = -(amplitude * transducer_elements / 2) + angle_center;
        printf("\"element_angle_bits\": [");
        for (size_t t = 0; t < transducer_elements; t++) { printf("%s%u", t ? "," : "", fbits(angle.to<float>())); angle = angle + amplitude; }
        printf("],\n");
        // degree -> radian as transducer.h:37-39 does it
        const float degs[] = { 0.0f, -90.0f, 120.0f, 45.0f, 90.0f, 33.3f };
        printf("\"deg2rad\": [");
        for (int i = 0; i < 6; i++) { degree_t d(degs[i]); radian_t r{ d }; printf("%s[%u,%.17g,%u]", i ? "," : "", fbits(degs[i]), r(), fbits(r.to<float>())); }
        printf("],\n");
    }
    // ---- scan-conversion maps: rfimage.h:183-215 as main.cpp:56 instantiates it (rf_image<512, 100, 322>{3_cm, 60_deg}, 400 x 500),
    //      and two other shapes (the 128-column headline image; other template constants and a non-integral depth product) ----
    {
        dump_maps<100, 1500>("scan_maps_reference", transducer_radius, transducer_amplitude, 512, 465, 400, 500);
        dump_maps<100, 1500>("scan_maps_headline", transducer_radius, transducer_amplitude, 128, 465, 400, 500);
        dump_maps<133, 1540>("scan_maps_other", millimeter_t(41.5), radian_t(degree_t(75.0f)), 192, 777, 333, 257);
    }
    // ---- time axis: main.cpp:114-118,139 + rfimage.h:33-40 on a sweep of inputs ----
    {
        using axis = rf_axis<322, 1500>;
        printf("\"time_axis\": [");
        const microsecond_t time_step = axis::micros_traveled(axial_resolution);
        uint64_t s = 88172645463325252ull;
        for (int i = 0; i < 400; i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            double dist_mm = (double)(s >> 11) * (1.0 / 9007199254740992.0) * 160.0;            // distance_traveled
            float seg_len = (float)((s & 0xffff) * (1.0 / 65536.0) * 20.0);                       // |to-from| in scene units
            millimeter_t distance_traveled(dist_mm);
            const auto starting_micros = axis::micros_traveled(distance_traveled);
            millimeter_t distance(seg_len * 10.0f);
            unsigned int steps = (unsigned int)(distance / axial_resolution);
            microsecond_t t = starting_micros;
            for (unsigned int k = 0; k < (s >> 20) % 50; k++) t = t + time_step;
            double row = axis::row(t);
            microsecond_t t_end = starting_micros + time_step * (steps - 1);
            printf("%s[%.17g,%u,%.17g,%u,%u,%.17g,%.17g,%.17g]", i ? "," : "", dist_mm, fbits(seg_len), starting_micros(), steps,
                   (unsigned)((s >> 20) % 50), t(), row, t_end());
        }
        printf("],\n");
    }
    // ---- texture volume (main.cpp:52 volume<256,145>) ----
    {
        static const volume<256, 145> vol;
        const float *m = reinterpret_cast<const float *>(&vol.matrix[0][0][0]);
        const size_t n = (size_t)256 * 256 * 256 * 2;
        uint64_t h = 1469598103934665603ull;
        double sum_noise = 0, sum_prob = 0;
        for (size_t i = 0; i < n; i++) {
            uint32_t b = fbits(m[i]);
            for (int k = 0; k < 4; k++) { h ^= (b >> (8 * k)) & 0xff; h *= 1099511628211ull; }
            if (i & 1) sum_prob += m[i]; else sum_noise += m[i];
        }
        printf("\"texture_fnv1a64\": \"%016llx\",\n", (unsigned long long)h);
        printf("\"texture_sum_noise\": %.17g,\n\"texture_sum_prob\": %.17g,\n", sum_noise, sum_prob);
        printf("\"texture_first_bits\": [");
        for (int i = 0; i < 16; i++) printf("%s%u", i ? "," : "", fbits(m[i]));
        printf("],\n\"texture_samples\": [");
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < 64; i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            size_t idx = (size_t)(s % (n / 2));
            printf("%s[%zu,%u,%u]", i ? "," : "", idx, fbits(m[2 * idx]), fbits(m[2 * idx + 1]));
        }
        printf("],\n\"scattering_probes\": [");
        // get_scattering(density, mu, sigma, x, y, z) volume.h:46-61, including negative coordinates
        for (int i = 0; i < 64; i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            float x = (float)((double)(s & 0xfffff) / 1048576.0 * 60.0 - 30.0);
            float y = (float)((double)((s >> 20) & 0xfffff) / 1048576.0 * 60.0 - 30.0);
            float z = (float)((double)((s >> 40) & 0xfffff) / 1048576.0 * 60.0 - 30.0);
            float dens = (i & 1) ? 0.6f : 1.0f, mu = 0.4f, sg = 0.3f;
            float r = vol.get_scattering(dens, mu, sg, x, y, z);
            printf("%s[%u,%u,%u,%u,%u,%u,%u]", i ? "," : "", fbits(x), fbits(y), fbits(z), fbits(dens), fbits(mu), fbits(sg), fbits(r));
        }
        printf("]\n");
    }
    printf("}\n");
    return 0;
}
