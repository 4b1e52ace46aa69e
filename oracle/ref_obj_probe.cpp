// oracle/ref_obj_probe.cpp -- TEST INFRASTRUCTURE.  Prints the triangle soup the REFERENCE's OBJ path produces for a file:
// tinyobj::LoadObj (src/wavefront/tiny_obj_loader.cpp:504-717, compiled where it lies, never copied) followed by the
// shape-by-shape, three-indices-at-a-time walk of load_mesh_from_obj's conversion loop (src/objloader.h:23-37: positions
// of indices f, f+1, f+2 of every shape, in shape order).  Output: JSON {"shapes": n, "tri_bits": [uint32 float bits ...]}.
// Built by `make -C oracle ref` into oracle/_ref/; oracle/gen_golden.py turns its output into tests/golden/obj_soup.json.
#include "wavefront/tiny_obj_loader.h"
#include <cstdio>
#include <cstring>
#include <cstdint>

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ref_obj_probe file.obj\n"); return 2; }
    std::vector<tinyobj::shape_t> shapes;
    std::string err = tinyobj::LoadObj(shapes, argv[1], "");
    if (!err.empty()) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    printf("{\"shapes\":%zu,\"tri_bits\":[", shapes.size());
    bool first = true;
    for (size_t s = 0; s < shapes.size(); s++) {
        const tinyobj::shape_t &sh = shapes[s];
        const int face_count = (int)sh.mesh.indices.size();
        for (int f = 0; f + 2 < face_count; f += 3)
            for (int k = 0; k < 3; k++)
                for (int c = 0; c < 3; c++) {
                    float x = sh.mesh.positions[sh.mesh.indices[f + k] * 3 + c];
                    uint32_t b; memcpy(&b, &x, 4);
                    printf(first ? "%u" : ",%u", b); first = false;
                }
    }
    printf("]}\n");
    return 0;
}
