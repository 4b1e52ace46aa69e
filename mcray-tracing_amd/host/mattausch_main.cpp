// mattausch_hip -- the reference's program (main.cpp:42-161) on the MI355X library:
//     mattausch_hip <scene.json> [frames] [samples] [out.pgm] [rf.bin] [--gpus N | --devices 0,1,...]
// --gpus N: the first N GPUs of the node, the frame's scan-lines sharded over them (mcrt_group_*: one tracing context and host thread per
// GPU, the blocks gathered on GPU 0); --devices lists them explicitly, and may repeat one (two ranks sharing a GPU: the one-GPU test).
// Same constants (main.cpp:23-37), same frame loop body; instead of blocking on imshow/waitKey every frame it
// runs `frames` frames, prints rays/s and frames/s, and writes the last B-mode image as a PGM.
#include "mcrt_host.hpp"
#include <chrono>
#include <cstring>
#include <iostream>

using namespace mcrt_host;

constexpr float transducer_frequency = 4.5f;                  // [MHz]
constexpr size_t transducer_elements = 512;
constexpr double transducer_amplitude = 60.0 * 3.14159265358979323846264338327950288419716939937510 / 180.0;   // 60_deg -> rad
constexpr double transducer_radius_cm = 3.0;
constexpr unsigned int resolution = 145;                      // [um]
using psf_ = psf<7, 13, 7, resolution>;
using rf_image_ = rf_image<transducer_elements, 100, 322>;    // max_travel_time 100 us, axial resolution 322 um (main.cpp:31,36)
using transducer_ = transducer<transducer_elements>;

int main(int argc, char **argv)
{
    std::vector<int> devices{ 0 };
    {   // the options, taken out of the positional arguments
        int keep = 1;
        for (int i = 1; i < argc; i++) {
            if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) { devices.clear(); for (int d = 0; d < std::max(1, std::atoi(argv[i + 1])); d++) devices.push_back(d); i++; }
            else if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) {
                devices.clear();
                for (const char *q = argv[i + 1]; *q;) { devices.push_back(std::atoi(q)); while (*q && *q != ',') q++; if (*q == ',') q++; }
                if (devices.empty()) devices.push_back(0);
                i++;
            }
            else argv[keep++] = argv[i];
        }
        argc = keep;
    }
    if (argc < 2) { std::cout << "Incorrect argument list." << std::endl; return 0; }
    const int frames = argc > 2 ? std::atoi(argv[2]) : 10;
    const unsigned samples = argc > 3 ? (unsigned)std::atoi(argv[3]) : 5;   // samples_te (main.cpp:27)
    try {
        const json cfg = load_json(argv[1]);
        const psf_ psf{ transducer_frequency, 0.05f, 0.2f, 0.1f };
        const auto &t_pos = cfg.at("transducerPosition");
        const auto &t_dir = cfg.at("transducerAngles");
        // millimeter_t sep = amplitude.to<float>() * radius / elements (main.cpp:66): float * cm -> cm, then -> mm
        const double separation_mm = (((double)(float)transducer_amplitude * transducer_radius_cm) / (double)transducer_elements) * 10.0;
        transducer_ transducer(transducer_frequency, transducer_radius_cm, separation_mm, vec3((float)t_pos[0], (float)t_pos[1], (float)t_pos[2]),
                               std::array<float, 3>{ (float)t_dir[0], (float)t_dir[1], (float)t_dir[2] });
        auto dev = std::make_shared<device>(devices);
        scene scene{ cfg, transducer, dev, samples };
        scene.step(1000.0f);
        rf_image_ rf_image{ dev, transducer_radius_cm * 10.0, transducer_amplitude };

        const auto t0 = std::chrono::high_resolution_clock::now();
        for (int f = 0; f < frames; f++) {
            rf_image.trace((uint32_t)f);      // clear + cast_rays + accumulation (main.cpp:102-144)
            rf_image.convolve(psf);           // main.cpp:146
            rf_image.envelope();              // main.cpp:147
            rf_image.postprocess();           // main.cpp:148
        }
        check(dev->synchronize(), "mcrt_synchronize");
        const double dt = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        std::cout << frames / dt << " frames/s, " << (double)frames * transducer_elements * samples / dt << " rays/s on " << devices.size() << " GPU context(s)" << std::endl;
        if (argc > 4) rf_image.save(argv[4]);
        if (argc > 5) {   // the last frame's RF image after main.cpp:146-147, row-major [465][512] float32 (for the parity test)
            const auto img = rf_image.intensities();
            std::ofstream f(argv[5], std::ios::binary);
            f.write((const char *)img.data(), (std::streamsize)(img.size() * sizeof(float)));
        }
    } catch (const std::exception &ex) {
        std::cout << "The program found an error and will terminate.\n" << "Reason:\n" << ex.what() << std::endl;
        return 1;
    }
    return 0;
}
