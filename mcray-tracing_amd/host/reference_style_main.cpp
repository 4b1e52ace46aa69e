// reference_style_main -- a program written the way the reference's own main.cpp is (main.cpp:42-161): same constants, same
// objects (volume, psf, rf_image, transducer, scene), the frame body clear() -> cast_rays<S,E>(transducer) -> host-side
// accumulation loop over the returned segments (get_scattering + add_echo) -> convolve -> envelope -> postprocess.
// It exists to show (and test) that the host shim offers the reference's class surface: only the include, the namespace and
// the unit types (plain doubles instead of units.h quantities, vec3 instead of btVector3) differ.  The ray casting runs on
// the GPU behind scene::cast_rays; the loop below is host code exactly as in the reference.
//     reference_style_main <scene.json> <rf_host_loop.bin> <rf_fused.bin>
// writes the RF image [465][512] after convolve() twice: accumulated by the host loop, and by the fused GPU path
// (rf_image::trace) for the same frame -- the test compares the two.
#include "mcrt_host.hpp"

using namespace mcrt_host;

constexpr double speed_of_sound = 1500.0;                          // [um/us]
constexpr float transducer_frequency = 4.5f;                       // [MHz]
constexpr float axial_resolution = 1.45f / transducer_frequency;   // [mm]
constexpr size_t transducer_elements = 512;
constexpr size_t samples_te = 5;
constexpr double transducer_amplitude = 60.0 * 3.14159265358979323846 / 180.0;   // [rad]
constexpr double transducer_radius = 3.0;                          // [cm]
constexpr double ultrasound_depth = 15.0;                          // [cm]
constexpr double max_travel_time = ultrasound_depth / speed_of_sound * 10000.0;   // [us]

constexpr unsigned int resolution = 145;                           // [um]
using psf_ = psf<7, 13, 7, resolution>;
using volume_ = volume<256, resolution>;
using rf_image_ = rf_image<transducer_elements, (unsigned int)max_travel_time, static_cast<unsigned int>(axial_resolution * 1000.0f)>;
using transducer_ = transducer<transducer_elements>;

static void write_image(const rf_image_ &img, const char *path)
{
    const auto px = img.intensities();
    std::ofstream f(path, std::ios::binary);
    f.write((const char *)px.data(), (std::streamsize)(px.size() * sizeof(float)));
}

int main(int argc, char **argv)
{
    if (argc != 4) { std::cout << "Incorrect argument list." << std::endl; return 0; }
    static const volume_ texture_volume;
    const psf_ psf{ transducer_frequency, 0.05f, 0.2f, 0.1f };
    try {
        rf_image_ rf_image{ transducer_radius * 10.0, transducer_amplitude };
        const json json = load_json(argv[1]);
        const auto &t_pos = json.at("transducerPosition");
        const double transducer_element_separation = (double)(float)transducer_amplitude * transducer_radius / transducer_elements * 10.0;   // [mm]
        const auto &t_dir = json.at("transducerAngles");
        std::array<float, 3> transducer_angles = { (float)t_dir[0], (float)t_dir[1], (float)t_dir[2] };
        transducer_ transducer(transducer_frequency, transducer_radius, transducer_element_separation, vec3(t_pos[0], t_pos[1], t_pos[2]), transducer_angles);
        std::cout << max_travel_time << std::endl;

        scene scene{ json, transducer };
        scene.step(1000.0f);

        rf_image.clear();
        auto rays = scene.cast_rays<samples_te, transducer_elements>(transducer);
        for (unsigned int ray_i = 0; ray_i < rays.size(); ray_i++) {
            const auto &ray = rays[ray_i];
            for (unsigned int sample_i = 0; sample_i < samples_te; sample_i++) {
                const auto &sample = ray[sample_i];
                for (auto &segment : sample) {
                    const auto starting_micros = rf_image.micros_traveled(segment.distance_traveled * 1000.0 /*mm -> um*/);
                    const auto distance = scene.distance(segment.from, segment.to);   // [mm]
                    auto steps = (unsigned int)(distance / (double)axial_resolution);
                    const auto delta_step = axial_resolution * segment.direction;
                    const auto time_step = rf_image.micros_traveled((double)axial_resolution * 1000.0);   // [us]
                    auto point = segment.from;
                    auto time_elapsed = starting_micros;
                    auto intensity = segment.initial_intensity;
                    for (unsigned int step = 0; step < steps && time_elapsed < max_travel_time; step++) {
                        float scattering = texture_volume.get_scattering(segment.media.mu1, segment.media.mu0, segment.media.sigma, point.x(), point.y(), point.z());
                        rf_image.add_echo(ray_i, intensity * scattering, time_elapsed);
                        point += delta_step;
                        time_elapsed = time_elapsed + time_step;
                        constexpr auto k = 1.0f;
                        intensity *= std::exp(-segment.attenuation * axial_resolution * 0.01f * transducer_frequency * k);
                    }
                    rf_image.add_echo(ray_i, (segment.reflected_intensity) / samples_te, starting_micros + time_step * (steps - 1));
                }
            }
        }
        rf_image.convolve(psf);
        write_image(rf_image, argv[2]);          // (written before the envelope: peak picking is not a continuous function of the image)
        rf_image.envelope();
        rf_image.postprocess();
        rf_image.show();

        // the same frame through the fused GPU path
        rf_image.trace(0);
        rf_image.convolve(psf);
        write_image(rf_image, argv[3]);
        rf_image.envelope();
        rf_image.postprocess();
        std::cout << "rf_image rows " << rf_image_::max_rows << ", dt " << rf_image.get_dt() << " us" << std::endl;
    } catch (const std::exception &ex) {
        std::cout << "The program found an error and will terminate.\n" << "Reason:\n" << ex.what() << std::endl;
        return 1;
    }
    return 0;
}
