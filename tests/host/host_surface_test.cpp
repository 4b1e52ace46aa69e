// host_surface_test -- TEST of the C++ host shim (mcray-tracing_amd/host/mcrt_host.hpp): every class the shim offers under the
// reference's names (transducer<N>, psf<>, volume<>, json, scene, rf_image<>, ray_physics::segment) is constructed and every public
// method is called once with a checked result.  Not a program of the product: built and run by tests/ only (GPU needed).
//     host_surface_test <scene.json> <rf_host_deposits.bin> <rf_gpu_frame.bin>
// Exit code = number of failed checks.  Two RF images [465][512] (after convolve) are written for the Python side, which holds them
// against the oracle: one deposited echo by echo on the host through rf_image::add_echo from the segments scene::cast_rays returned,
// one traced and accumulated on the GPU (rf_image::trace) for the same frame.
#include "mcrt_host.hpp"
#include <cstdio>

using namespace mcrt_host;

static int failed = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond); failed++; } } while (0)
template <class Ex, class Fn> static bool throws(Fn &&fn) { try { fn(); } catch (const Ex &) { return true; } catch (...) { return false; } return false; }

namespace cfg {                       // the reference's frame: 512 elements x 5 samples, 4.5 MHz, 15 cm, 145 um texture (main.cpp:23-37)
constexpr size_t E = 512, S = 5;
constexpr float mhz = 4.5f, step_mm = 1.45f / mhz;
constexpr double depth_us = 15.0 / 1500.0 * 10000.0, aperture = 60.0 * 3.14159265358979323846 / 180.0, radius_cm = 3.0;
using image = rf_image<E, (unsigned int)depth_us, (unsigned int)(step_mm * 1000.0f)>;
using probe = transducer<E>;
using kernel = psf<7, 13, 7, 145>;
using tissue = volume<256, 145>;
}

static void dump(const cfg::image &img, const char *path)
{
    const std::vector<float> px = img.intensities();
    std::ofstream(path, std::ios::binary).write((const char *)px.data(), (std::streamsize)(px.size() * sizeof(float)));
}

// one segment's echoes into column `col`, through the shim's host-side pieces only
static void deposit(cfg::image &img, const cfg::tissue &vol, const scene &sc, unsigned col, const ray_physics::segment &sg)
{
    const double t0 = img.micros_traveled(sg.distance_traveled * 1000.0), dt = img.micros_traveled((double)cfg::step_mm * 1000.0);
    const unsigned n = (unsigned)(sc.distance(sg.from, sg.to) / (double)cfg::step_mm);
    const vec3 hop = cfg::step_mm * sg.direction;
    vec3 p = sg.from; double t = t0; float level = sg.initial_intensity;
    for (unsigned k = 0; k < n && t < cfg::depth_us; k++, p += hop, t = t + dt, level *= std::exp(-sg.attenuation * cfg::step_mm * 0.01f * cfg::mhz * 1.0f))
        img.add_echo(col, level * vol.get_scattering(sg.media.mu1, sg.media.mu0, sg.media.sigma, p.x(), p.y(), p.z()), t);
    img.add_echo(col, sg.reflected_intensity / cfg::S, t0 + dt * (n - 1));
}

int main(int argc, char **argv)
{
    if (argc != 4) { std::printf("usage: host_surface_test <scene.json> <rf_host.bin> <rf_gpu.bin>\n"); return 99; }
    try {
        // ---- json + load_json
        const json js = load_json(argv[1]);
        CHECK(js.contains("materials") && js.at("materials").is_array() && !js.contains("no such key"));
        CHECK(throws<std::out_of_range>([&] { (void)js.at("no such key"); }));
        CHECK(throws<std::domain_error>([&] { (void)(double)js.at("startingMaterial"); }));
        CHECK(throws<std::runtime_error>([] { (void)load_json("/nonexistent/file.scene"); }));

        // ---- transducer<N>
        const auto &at = js.at("transducerPosition"), &ang = js.at("transducerAngles");
        const double sep_mm = (double)(float)cfg::aperture * cfg::radius_cm / cfg::E * 10.0;
        cfg::probe probe(cfg::mhz, cfg::radius_cm, sep_mm, vec3(at[0], at[1], at[2]), { (float)ang[0], (float)ang[1], (float)ang[2] });
        CHECK(cfg::probe::size() == cfg::E && probe.frequency == cfg::mhz);
        CHECK(std::fabs(probe.element(0).direction.length() - 1.0f) < 1e-5f && std::fabs(probe.element(cfg::E - 1).direction.length() - 1.0f) < 1e-5f);
        CHECK(std::fabs(probe.element(7).position.distance(probe.getPosition()) - (float)cfg::radius_cm) < 1e-4f);
        CHECK(throws<std::out_of_range>([&] { (void)probe.element(cfg::E); }));
        CHECK(throws<std::invalid_argument>([&] { transducer<64> wide(cfg::mhz, 1.0, 5.0, vec3(0, 0, 0), { 0, 0, 0 }); }));
        {   // setPosition / setAngles take effect at update() (transducer.h:82-118)
            transducer<8> small(cfg::mhz, cfg::radius_cm, 1.0, vec3(0, 0, 0), { 0, 0, 0 });
            const vec3 before = small.element(3).position;
            small.setPosition(vec3(1, 2, 3)); small.setAngles({ 10, 20, 30 });
            CHECK(small.element(3).position.distance(before) == 0.0f);
            small.update();
            CHECK(small.element(3).position.distance(before) > 1.0f && small.getPosition().distance(vec3(1, 2, 3)) == 0.0f);
            std::cout.setstate(std::ios::failbit); small.print(true); small.print(false); std::cout.clear();      // "x,z" lines (transducer.h:69-80)
        }

        // ---- psf<>
        const cfg::kernel taps{ cfg::mhz, 0.05f, 0.2f, 0.1f };
        CHECK(taps.get_axial_size() == 7 && taps.get_lateral_size() == 13 && taps.get_elevation_size() == 7);
        CHECK(taps.axial_kernel[2] > 0.6f && taps.axial_kernel[3] < -0.4f && taps.lateral_kernel[6] > 0.98f && taps.lateral_kernel[0] < 0.11f);

        // ---- volume<>
        static const cfg::tissue vol;
        CHECK(vol.get_resolution_in_millis() == 0.145f);
        {
            const float *m = vol.data();                        // voxel (1, 2, 3): noise, probability
            const size_t v = 2 * (((size_t)1 * 256 + 2) * 256 + 3);
            CHECK(vol.get_scattering(m[v + 1], 0.5f, 2.0f, 0.2f, 0.3f, 0.45f) == m[v] * 2.0f + 0.5f);     // probability >= density: noise * sigma + mu
            CHECK(vol.get_scattering(std::nextafter(m[v + 1], 10.0f), 0.5f, 2.0f, 0.2f, 0.3f, 0.45f) == 0.0f);
            CHECK(vol.get_scattering(-10.0f, 0.0f, 1.0f, -0.1f, 0.0f, 0.0f) == m[0]);                      // -0.1 / 0.145 truncates to voxel 0
            CHECK(vol.get_scattering(-10.0f, 0.0f, 1.0f, -0.2f, 0.0f, 0.0f) == m[2 * ((size_t)255 * 256 * 256)]);   // -1 as unsigned wraps to 4294967295 % 256 = 255
        }

        // ---- scene
        scene sc{ js, probe };
        sc.step(1000.0f);
        CHECK(sc.distance(vec3(0, 0, 0), vec3(3, 4, 0)) == 50.0);                                     // [mm] of scene units [cm]
        CHECK(throws<std::runtime_error>([&] { json broken = js; broken.obj.erase(broken.obj.begin()); scene bad{ broken, probe }; }));
        auto paths = sc.cast_rays<cfg::S, cfg::E>(probe);
        CHECK(paths.size() == cfg::E && paths[0].size() == cfg::S && !paths[cfg::E / 2][0].empty());
        {
            const ray_physics::segment &first = paths[100][2][0];
            CHECK(first.from.distance(probe.element(100).position) == 0.0f && first.initial_intensity == 1.0f / cfg::S && first.distance_traveled == 0.0);
            CHECK(std::fabs(first.direction.dot(probe.element(100).direction) - 1.0f) < 1e-6f && first.attenuation == 1e-8f);
            size_t total = 0; for (auto &line : paths) for (auto &smp : line) total += smp.size();
            CHECK(total > cfg::E * cfg::S && total <= cfg::E * cfg::S * 10);
        }

        // ---- rf_image<>
        cfg::image img{ cfg::radius_cm * 10.0, cfg::aperture };
        CHECK(cfg::image::max_rows == 465 && img.get_dt() == 322.0 / 1500.0 && img.micros_traveled(3000.0) == 2.0);
        img.add_echo(9, 0.25f, 3.0 * img.get_dt() + 1e-9); img.add_echo(9, 0.5f, 3.0 * img.get_dt() + 1e-9); img.add_echo(9, 1.0f, 1e9);
        CHECK(img.intensities()[3 * cfg::E + 9] == 0.75f);
        img.clear();
        { float sum = 0; for (float v : img.intensities()) sum += std::fabs(v); CHECK(sum == 0.0f); }
        std::cout.setstate(std::ios::failbit); img.print(9); std::cout.clear();
        for (unsigned col = 0; col < paths.size(); col++)
            for (auto &smp : paths[col])
                for (auto &sg : smp) deposit(img, vol, sc, col, sg);
        img.convolve(taps);
        dump(img, argv[2]);
        img.envelope(); img.postprocess(); img.show();
        { const auto sc_img = img.scan_converted(); size_t nz = 0; for (float v : sc_img) nz += v != 0.0f; CHECK(sc_img.size() == 400 * 500 && nz > 10000); }
        img.save(std::string(argv[2]) + ".pgm");
        { std::ifstream pgm(std::string(argv[2]) + ".pgm", std::ios::binary); std::string magic; pgm >> magic; CHECK(magic == "P5"); }
        img.trace(0);                                                                                  // the same frame on the GPU
        img.convolve(taps);
        dump(img, argv[3]);
        std::printf("host surface: %d check(s) failed\n", failed);
    } catch (const std::exception &ex) {
        std::printf("host surface: exception %s\n", ex.what());
        return 98;
    }
    return failed;
}
