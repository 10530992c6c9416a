// Host-only part of the drop-in boundary: scene JSON (src/scene/mod.rs:19-43, 56-66) and the particle generator of
// HybridFluid::add_fluid_cube (src/simulation/hybrid_fluid.rs:609-678).  No device code, no HIP calls.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "blub_internal.h"

namespace blub {

// ------------------------------------------------------------------------------------------------------------------
// Minimal JSON reader (serde_json::from_reader stand-in): objects, arrays, numbers, strings, true/false/null.
// ------------------------------------------------------------------------------------------------------------------
struct JValue {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    double num = 0;
    bool b = false;
    std::string str;
    std::vector<JValue> arr;
    std::vector<std::pair<std::string, JValue>> obj;
    const JValue* get(const char* key) const {
        if (kind != Obj) return nullptr;
        for (auto& kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    const char* p; const char* end; std::string err;
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    bool parse_string(std::string& out) {
        if (p >= end || *p != '"') return fail("expected string");
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("bad escape");
                switch (*p) {
                case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                case 'u': { if (end - p < 5) return fail("bad \\u"); unsigned v = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                            out += (v < 128) ? (char)v : '?'; p += 4; break; }
                default: out += *p;
                }
                ++p;
            } else out += *p++;
        }
        if (p >= end) return fail("unterminated string");
        ++p; return true;
    }
    bool parse(JValue& v, int depth = 0) {
        if (depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        char c = *p;
        if (c == '{') {
            v.kind = JValue::Obj; ++p; ws();
            if (p < end && *p == '}') { ++p; return true; }
            for (;;) {
                ws(); std::string k; if (!parse_string(k)) return false;
                ws(); if (p >= end || *p != ':') return fail("expected ':'"); ++p;
                JValue child; if (!parse(child, depth + 1)) return false;
                v.obj.emplace_back(std::move(k), std::move(child));
                ws(); if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = JValue::Arr; ++p; ws();
            if (p < end && *p == ']') { ++p; return true; }
            for (;;) {
                JValue child; if (!parse(child, depth + 1)) return false;
                v.arr.push_back(std::move(child));
                ws(); if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v.kind = JValue::Str; return parse_string(v.str); }
        if (!strncmp(p, "true", std::min<size_t>(4, end - p)) && end - p >= 4) { v.kind = JValue::Bool; v.b = true; p += 4; return true; }
        if (!strncmp(p, "false", std::min<size_t>(5, end - p)) && end - p >= 5) { v.kind = JValue::Bool; v.b = false; p += 5; return true; }
        if (!strncmp(p, "null", std::min<size_t>(4, end - p)) && end - p >= 4) { v.kind = JValue::Null; p += 4; return true; }
        {
            std::string tok; const char* q = p;
            while (q < end && (isdigit((unsigned char)*q) || *q == '-' || *q == '+' || *q == '.' || *q == 'e' || *q == 'E')) ++q;
            if (q == p) return fail("unexpected character");
            tok.assign(p, q); char* e = nullptr; v.num = strtod(tok.c_str(), &e);
            if (!e || *e) return fail("bad number");
            v.kind = JValue::Num; p = q; return true;
        }
    }
};

static bool get_f32(const JValue* o, const char* key, float* out) {
    const JValue* v = o ? o->get(key) : nullptr;
    if (!v || v->kind != JValue::Num) return false;
    *out = (float)v->num;   // serde: f64 text -> f32 (round to nearest)
    return true;
}
static bool get_u32(const JValue* o, const char* key, uint32_t* out) {
    const JValue* v = o ? o->get(key) : nullptr;
    if (!v || v->kind != JValue::Num || v->num < 0 || v->num > 4294967295.0 || v->num != std::floor(v->num)) return false;
    *out = (uint32_t)v->num;
    return true;
}
static bool get_vec3(const JValue* o, const char* key, float* out) {
    const JValue* v = o ? o->get(key) : nullptr;
    return v && get_f32(v, "x", out) && get_f32(v, "y", out + 1) && get_f32(v, "z", out + 2);
}

int scene_parse(const char* text, size_t len, blub_scene_config* out) {
    if (!text || !out) return set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    JParser ps{text, text + len, {}};
    JValue root;
    if (!ps.parse(root)) return set_error(BLUB_ERR_PARSE, ("scene JSON: " + ps.err).c_str());
    ps.ws();
    if (ps.p != ps.end) return set_error(BLUB_ERR_PARSE, "scene JSON: trailing characters");
    memset(out, 0, sizeof(*out));
    if (!get_vec3(&root, "gravity", out->gravity)) return set_error(BLUB_ERR_PARSE, "scene JSON: missing field `gravity`");
    const JValue* fluid = root.get("fluid");
    if (!fluid || fluid->kind != JValue::Obj) return set_error(BLUB_ERR_PARSE, "scene JSON: missing field `fluid`");
    if (!get_vec3(fluid, "world_position", out->world_position)) return set_error(BLUB_ERR_PARSE, "scene JSON: missing field `world_position`");
    if (!get_f32(fluid, "grid_to_world_scale", &out->grid_to_world_scale)) return set_error(BLUB_ERR_PARSE, "scene JSON: missing field `grid_to_world_scale`");
    const JValue* gd = fluid->get("grid_dimension");
    if (!gd || !get_u32(gd, "x", &out->grid_dimension[0]) || !get_u32(gd, "y", &out->grid_dimension[1]) || !get_u32(gd, "z", &out->grid_dimension[2]))
        return set_error(BLUB_ERR_PARSE, "scene JSON: missing field `grid_dimension`");
    if (!get_u32(fluid, "max_num_particles", &out->max_num_particles)) return set_error(BLUB_ERR_PARSE, "scene JSON: missing field `max_num_particles`");
    const JValue* cubes = fluid->get("fluid_cubes");
    if (!cubes || cubes->kind != JValue::Arr) return set_error(BLUB_ERR_PARSE, "scene JSON: missing field `fluid_cubes`");
    if (cubes->arr.size() > BLUB_SCENE_MAX_CUBES) return set_error(BLUB_ERR_UNSUPPORTED, "scene JSON: too many fluid cubes");
    out->num_fluid_cubes = (uint32_t)cubes->arr.size();
    for (size_t i = 0; i < cubes->arr.size(); ++i)
        if (!get_vec3(&cubes->arr[i], "min", out->cube_min[i]) || !get_vec3(&cubes->arr[i], "max", out->cube_max[i]))
            return set_error(BLUB_ERR_PARSE, "scene JSON: fluid cube needs `min` and `max`");
    const JValue* so = root.get("static_objects");   // #[serde(default)]
    out->num_static_objects = (so && so->kind == JValue::Arr) ? (uint32_t)so->arr.size() : 0;
    return BLUB_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// rand 0.8.5 `SmallRng` on 64-bit targets = xoshiro256++ (Cargo.lock:1741-1743); `SmallRng` does not forward
// seed_from_u64, so rand_core 0.6's default applies: the 32 seed bytes come from a PCG32 stream (LE u32 chunks).
// The crate sources are not vendored in /root/reference; this restates their published algorithms.
// ------------------------------------------------------------------------------------------------------------------
struct SmallRng {
    uint64_t s[4];
    static uint64_t rotl(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
    explicit SmallRng(uint64_t state) {
        const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
        uint32_t w[8];
        for (int c = 0; c < 8; ++c) {
            state = state * MUL + INC;
            uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
            w[c] = (xs >> rot) | (xs << ((32 - rot) & 31));
        }
        for (int i = 0; i < 4; ++i) s[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
        if (!(s[0] | s[1] | s[2] | s[3])) { s[0] = 0xe220a8397b1dcdafull; s[1] = 0x6e789e6aa1b965f4ull; s[2] = 0x06c45d188009454full; s[3] = 0xf88bb8a8724c81ecull; }
    }
    uint64_t next_u64() {
        uint64_t r = rotl(s[0] + s[3], 23) + s[0], t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    float gen_f32() { return (float)((uint32_t)(next_u64() >> 32) >> 8) * (1.0f / 16777216.0f); }
};

static uint32_t rust_f32_as_u32(float v) {   // `as u32`: saturating, NaN -> 0
    if (!(v > 0.0f)) return 0;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}

// hybrid_fluid.rs:609-617
static void clamp_to_grid(const uint32_t dim[3], const float g[3], uint32_t out[3]) {
    for (int k = 0; k < 3; ++k) { uint32_t v = rust_f32_as_u32(g[k]); v = v < dim[k] - 1 ? v : dim[k] - 1; out[k] = v > 1 ? v : 1; }
}

int seed_fluid_cube(const uint32_t dim[3], uint32_t max_particles, uint32_t before, const float mn_g[3], const float mx_g[3],
                    float* out, size_t capacity, uint32_t* count_out) {
    if (!dim || !mn_g || !mx_g || !count_out) return set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    if (dim[0] < 3 || dim[1] < 3 || dim[2] < 3 || before > max_particles) return set_error(BLUB_ERR_INVALID_ARGUMENT, "bad grid / particle count");
    uint32_t mn[3], mx[3], ext[3];
    clamp_to_grid(dim, mn_g, mn); clamp_to_grid(dim, mx_g, mx);
    for (int k = 0; k < 3; ++k) ext[k] = mx[k] - mn[k];   // wraps like the release-mode reference if max < min
    uint32_t num_new = ext[0] * ext[1] * ext[2] * BLUB_PARTICLES_PER_GRID_CELL;
    if (max_particles < num_new + before) num_new = max_particles - before;   // :627-633 logs error! and truncates
    *count_out = num_new;
    if (!out) return BLUB_OK;   // size query
    if (capacity < num_new) return set_error(BLUB_ERR_INVALID_ARGUMENT, "output buffer too small");
    SmallRng rng((uint64_t)(before + num_new));   // :637
    for (uint32_t i = 0; i < num_new; ++i) {
        float cx = (float)(mn[0] + i / 8 % ext[0]), cy = (float)(mn[1] + i / 8 / ext[0] % ext[1]), cz = (float)(mn[2] + i / 8 / ext[0] / ext[1]);
        uint32_t sidx = i % 8;
        float rx = rng.gen_f32(), ry = rng.gen_f32(), rz = rng.gen_f32();
        float* o = out + (size_t)i * 4;
        o[0] = cx + ((float)(sidx % 2) * 0.5f + rx * 0.5f);        // :664-667 stratified jitter
        o[1] = cy + ((float)(sidx / 2 % 2) * 0.5f + ry * 0.5f);
        o[2] = cz + ((float)(sidx / 4 % 2) * 0.5f + rz * 0.5f);
        uint32_t inv = 0xFFFFFFFFu; memcpy(o + 3, &inv, 4);
    }
    return BLUB_OK;
}

}  // namespace blub

extern "C" {
int blub_scene_parse_json(const char* text, size_t len, blub_scene_config* out) { return blub::scene_parse(text, len, out); }
int blub_scene_load_json(const char* path, blub_scene_config* out) {
    if (!path || !out) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return blub::set_error(BLUB_ERR_IO, (std::string("cannot open scene file ") + path).c_str());
    std::string text; char buf[4096]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    fclose(f);
    return blub::scene_parse(text.data(), text.size(), out);
}
int blub_seed_fluid_cube(const uint32_t grid_dim[3], uint32_t max_num_particles, uint32_t num_particles_before,
                         const float min_grid[3], const float max_grid[3], float* pos_ll_out, size_t capacity, uint32_t* count_out) {
    return blub::seed_fluid_cube(grid_dim, max_num_particles, num_particles_before, min_grid, max_grid, pos_ll_out, capacity, count_out);
}
}
