// Internal declarations shared by the host-only and device translation units of libblubhip.so.
#pragma once
#include <cstddef>
#include <cstdint>

#include "blubhip.h"

namespace blub {
int set_error(int status, const char* msg);   // records msg for blub_last_error_string(); returns status
int scene_parse(const char* text, size_t len, blub_scene_config* out);
int seed_fluid_cube(const uint32_t dim[3], uint32_t max_particles, uint32_t before, const float mn_g[3], const float mx_g[3],
                    float* out, size_t capacity, uint32_t* count_out);
int scene_mesh_desc_at_time(const blub_scene_config* scene, uint32_t index, uint64_t total_ns, uint64_t delta_ns, blub_mesh_desc* out);
int load_obj(const char* path, float* pos, size_t vcap, uint32_t* nv_out, uint32_t* idx, size_t icap, uint32_t* ni_out);
}  // namespace blub
