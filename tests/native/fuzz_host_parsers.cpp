// Mutation fuzzer for the host-only entry points of the C-ABI (scene JSON, static-object animation, OBJ reader); built with
// -fsanitize=address,undefined by tests/test_host_fuzz.py.  usage: fuzz <scene.json> <model.obj> <scratch.obj> [iterations]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "blubhip.h"
namespace blub { int set_error(int status, const char* msg) { (void)msg; return status; } }
static std::string slurp(const char* p) { FILE* f = fopen(p, "rb"); std::string s; char b[4096]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) s.append(b, n); fclose(f); return s; }
int main(int argc, char** argv) {
    std::string base = slurp(argv[1]);
    std::string obj = slurp(argv[2]);
    unsigned seed = 1;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    const char* junk[] = {"{", "}", "[", "]", "\"", ",", ":", "-", "1e999", "nan", "null", "true", "\\u00", "\\", "0x", "\n", " ", "9999999999999999999999", "f 1 2 3\n", "f -9 2 3\n", "v 1 2\n", "f 1/2/3 4//5 6\n"};
    long ok = 0, bad = 0;
    const int iterations = argc > 4 ? atoi(argv[4]) : 20000;
    for (int it = 0; it < iterations; ++it) {
        std::string s = (it & 1) ? obj : base;
        int muts = 1 + rnd() % 4;
        for (int m = 0; m < muts; ++m) {
            size_t pos = s.empty() ? 0 : rnd() % s.size();
            switch (rnd() % 4) {
            case 0: if (!s.empty()) s.erase(pos, 1 + rnd() % 8); break;
            case 1: s.insert(pos, junk[rnd() % (sizeof junk / sizeof *junk)]); break;
            case 2: if (!s.empty()) s[pos] = (char)(rnd() & 0xFF); break;
            case 3: if (!s.empty()) s.resize(pos); break;
            }
        }
        if (it & 1) {
            FILE* f = fopen(argv[3], "wb"); fwrite(s.data(), 1, s.size(), f); fclose(f);
            uint32_t nv = 0, ni = 0;
            int rc = blub_load_obj(argv[3], nullptr, 0, &nv, nullptr, 0, &ni);
            if (rc == 0) { std::vector<float> P(3 * (size_t)nv + 3); std::vector<uint32_t> I(ni + 3); rc = blub_load_obj(argv[3], P.data(), nv, &nv, I.data(), ni, &ni); for (uint32_t k = 0; k < ni; ++k) if (I[k] >= nv) { printf("BAD INDEX\n"); return 1; } }
            rc == 0 ? ++ok : ++bad;
        } else {
            blub_scene_config cfg;
            int rc = blub_scene_parse_json(s.data(), s.size(), &cfg);
            if (rc == 0) {
                if (cfg.num_fluid_cubes > BLUB_SCENE_MAX_CUBES || cfg.num_static_objects > BLUB_SCENE_MAX_STATIC_OBJECTS) { printf("BAD COUNT\n"); return 1; }
                for (uint32_t k = 0; k < cfg.num_static_objects; ++k) { blub_mesh_desc d; (void)blub_scene_mesh_desc_at_time(&cfg, k, 8333333ull * (it % 1000), 8333333ull, &d); if (memchr(cfg.static_objects[k].model, 0, BLUB_SCENE_MAX_PATH) == nullptr) { printf("UNTERMINATED\n"); return 1; } }
                ++ok;
            } else ++bad;
        }
    }
    printf("accepted %ld rejected %ld\n", ok, bad);
    return 0;
}
