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
    out->num_static_objects = 0;
    if (so && so->kind == JValue::Arr) {
        if (so->arr.size() > BLUB_SCENE_MAX_STATIC_OBJECTS) return set_error(BLUB_ERR_UNSUPPORTED, "scene JSON: too many static objects");
        for (size_t i = 0; i < so->arr.size(); ++i) {   // StaticObjectConfig, scene/models.rs:11-19
            const JValue& o = so->arr[i];
            blub_static_object& d = out->static_objects[i];
            const JValue* model = o.get("model");
            if (!model || model->kind != JValue::Str || model->str.size() >= BLUB_SCENE_MAX_PATH) return set_error(BLUB_ERR_PARSE, "scene JSON: static object needs a `model` path");
            memcpy(d.model, model->str.c_str(), model->str.size() + 1);
            if (!get_vec3(&o, "world_position", d.world_position) || !get_f32(&o, "scale", &d.scale) || !get_vec3(&o, "rotation_angles", d.rotation_angles_deg))
                return set_error(BLUB_ERR_PARSE, "scene JSON: static object needs `world_position`, `scale`, `rotation_angles`");
            const JValue* anim = o.get("animation");
            if (anim && anim->kind == JValue::Obj) {
                const JValue* tr = anim->get("translation");
                if (tr && tr->kind == JValue::Obj) {   // TranslationAnimation, models.rs:27-32
                    const JValue* curve = tr->get("curve");
                    if (!get_vec3(tr, "target", d.translation_target) || !get_f32(tr, "duration", &d.translation_duration) || !curve || curve->kind != JValue::Str)
                        return set_error(BLUB_ERR_PARSE, "scene JSON: translation animation needs `target`, `curve`, `duration`");
                    if (curve->str == "Linear") d.translation_curve = BLUB_CURVE_LINEAR;
                    else if (curve->str == "SmoothStep") d.translation_curve = BLUB_CURVE_SMOOTHSTEP;
                    else return set_error(BLUB_ERR_PARSE, "scene JSON: unknown animation curve");
                    d.has_translation = 1;
                }
                const JValue* rot = anim->get("rotation");
                if (rot && rot->kind == JValue::Obj) {   // RotationAnimation, models.rs:34-38
                    if (!get_vec3(rot, "axis", d.rotation_axis) || !get_f32(rot, "deg_per_sec", &d.rotation_deg_per_sec))
                        return set_error(BLUB_ERR_PARSE, "scene JSON: rotation animation needs `axis`, `deg_per_sec`");
                    d.has_rotation = 1;
                }
            }
        }
        out->num_static_objects = (uint32_t)so->arr.size();
    }
    return BLUB_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// StaticMeshData::{world_position_at_time, rotation_at_time, to_gpu} (scene/models.rs:156-228) in f32, with the cgmath 0.18
// operations it calls restated (crate source not vendored in /root/reference: Euler -> Quaternion uses the XYZ formula of
// cgmath's quaternion.rs, matrix products sum their four terms left to right).
// ------------------------------------------------------------------------------------------------------------------
struct Quat { float s, x, y, z; };
struct Mat4 { float c[4][4]; };   // column major: c[column][row]
static float deg_to_rad(float d) { return d * (float)(3.14159265358979323846 / 180.0); }
static Quat quat_mul(const Quat& a, const Quat& b) {
    return {a.s * b.s - a.x * b.x - a.y * b.y - a.z * b.z, a.s * b.x + a.x * b.s + a.y * b.z - a.z * b.y,
            a.s * b.y + a.y * b.s + a.z * b.x - a.x * b.z, a.s * b.z + a.z * b.s + a.x * b.y - a.y * b.x};
}
static Quat quat_from_euler_deg(const float e[3]) {
    const float hx = deg_to_rad(e[0]) * 0.5f, hy = deg_to_rad(e[1]) * 0.5f, hz = deg_to_rad(e[2]) * 0.5f;
    const float sx = sinf(hx), cx = cosf(hx), sy = sinf(hy), cy = cosf(hy), sz = sinf(hz), cz = cosf(hz);
    return {-sx * sy * sz + cx * cy * cz, sx * cy * cz + sy * sz * cx, -sx * sz * cy + sy * cx * cz, sx * sy * cz + sz * cx * cy};
}
static void normalize3(const float v[3], float out[3]) {
    const float inv = 1.0f / sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    for (int k = 0; k < 3; ++k) out[k] = v[k] * inv;
}
static Mat4 mat_identity() { Mat4 m{}; for (int k = 0; k < 4; ++k) m.c[k][k] = 1.0f; return m; }
static Mat4 mat_mul(const Mat4& a, const Mat4& b) {
    Mat4 r{};
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row) r.c[col][row] = a.c[0][row] * b.c[col][0] + a.c[1][row] * b.c[col][1] + a.c[2][row] * b.c[col][2] + a.c[3][row] * b.c[col][3];
    return r;
}
static Mat4 mat_from_quat(const Quat& q) {
    const float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    const float xx2 = x2 * q.x, xy2 = x2 * q.y, xz2 = x2 * q.z, yy2 = y2 * q.y, yz2 = y2 * q.z, zz2 = z2 * q.z, sy2 = y2 * q.s, sz2 = z2 * q.s, sx2 = x2 * q.s;
    Mat4 m = mat_identity();
    m.c[0][0] = 1.0f - yy2 - zz2; m.c[0][1] = xy2 + sz2; m.c[0][2] = xz2 - sy2;
    m.c[1][0] = xy2 - sz2; m.c[1][1] = 1.0f - xx2 - zz2; m.c[1][2] = yz2 + sx2;
    m.c[2][0] = xz2 + sy2; m.c[2][1] = yz2 - sx2; m.c[2][2] = 1.0f - xx2 - yy2;
    return m;
}
static float duration_as_secs_f32(uint64_t ns) { return (float)(ns / 1000000000ull) + (float)(uint32_t)(ns % 1000000000ull) / 1000000000.0f; }

static void world_position_at_time(const blub_static_object& o, uint64_t total_ns, float out[3]) {   // models.rs:157-175
    if (!o.has_translation) { for (int k = 0; k < 3; ++k) out[k] = o.world_position[k]; return; }
    float progress = fmodf(duration_as_secs_f32(total_ns), o.translation_duration * 2.0f);
    if (progress > o.translation_duration) progress = o.translation_duration * 2.0f - progress;
    progress /= o.translation_duration;
    progress = progress < 0.0f ? 0.0f : (progress > 1.0f ? 1.0f : progress);   // f32::clamp (NaN stays NaN)
    if (o.translation_curve == BLUB_CURVE_SMOOTHSTEP) progress = progress * progress * (3.0f - 2.0f * progress);
    for (int k = 0; k < 3; ++k) out[k] = o.world_position[k] * (1.0f - progress) + o.translation_target[k] * progress;
}
static Quat rotation_at_time(const blub_static_object& o, uint64_t total_ns) {   // models.rs:177-188
    const Quat st = quat_from_euler_deg(o.rotation_angles_deg);
    if (!o.has_rotation) return st;
    float axis[3]; normalize3(o.rotation_axis, axis);
    const float half = deg_to_rad(o.rotation_deg_per_sec * duration_as_secs_f32(total_ns)) * 0.5f;
    const float s = sinf(half), c = cosf(half);
    return quat_mul(st, Quat{c, axis[0] * s, axis[1] * s, axis[2] * s});
}

int scene_mesh_desc_at_time(const blub_scene_config* scene, uint32_t index, uint64_t total_ns, uint64_t delta_ns, blub_mesh_desc* out) {   // to_gpu, models.rs:190-228
    if (!scene || !out || index >= scene->num_static_objects) return set_error(BLUB_ERR_INVALID_ARGUMENT, "bad static object index");
    const blub_static_object& o = scene->static_objects[index];
    float wp[3]; world_position_at_time(o, total_ns, wp);
    const Quat rot = rotation_at_time(o, total_ns);
    float vel[3] = {0.0f, 0.0f, 0.0f};
    if (total_ns > delta_ns) {   // "brute force" backward difference (:195-199)
        float prev[3]; world_position_at_time(o, total_ns - delta_ns, prev);
        const float dt = duration_as_secs_f32(delta_ns);
        for (int k = 0; k < 3; ++k) vel[k] = (wp[k] - prev[k]) / dt;
    }
    Mat4 T = mat_identity(); for (int k = 0; k < 3; ++k) T.c[3][k] = wp[k];
    Mat4 S = mat_identity(); for (int k = 0; k < 3; ++k) S.c[k][k] = o.scale;
    const Mat4 world = mat_mul(mat_mul(T, S), mat_from_quat(rot));
    const float inv = 1.0f / scene->grid_to_world_scale;
    Mat4 Sv = mat_identity(); for (int k = 0; k < 3; ++k) Sv.c[k][k] = inv;
    Mat4 Tv = mat_identity(); for (int k = 0; k < 3; ++k) Tv.c[3][k] = -scene->world_position[k];
    const Mat4 voxel = mat_mul(mat_mul(Sv, Tv), world);
    memset(out, 0, sizeof(*out));
    for (int row = 0; row < 3; ++row) for (int col = 0; col < 4; ++col) out->voxel_transform[row][col] = voxel.c[col][row];
    for (int k = 0; k < 3; ++k) out->fluid_space_velocity[k] = vel[k] / scene->grid_to_world_scale;
    if (o.has_rotation) {
        float axis[3]; normalize3(o.rotation_axis, axis);
        const float w = deg_to_rad(o.rotation_deg_per_sec);
        for (int k = 0; k < 3; ++k) out->fluid_space_rotation_axis_scaled[k] = axis[k] * w;
    }
    return BLUB_OK;
}

// tobj::load_obj stand-in (models.rs:267-276): positions and fan-triangulated faces; everything else is ignored.
int load_obj(const char* path, float* pos, size_t vcap, uint32_t* nv_out, uint32_t* idx, size_t icap, uint32_t* ni_out) {
    if (!path || !nv_out || !ni_out) return set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return set_error(BLUB_ERR_IO, (std::string("cannot open model file ") + path).c_str());
    std::vector<float> P; std::vector<uint32_t> I;
    char line[4096];
    bool bad = false;
    while (fgets(line, sizeof line, f)) {
        const char* c = line;
        while (*c == ' ' || *c == '\t') ++c;
        if (c[0] == 'v' && (c[1] == ' ' || c[1] == '\t')) {
            float x, y, z;
            if (sscanf(c + 1, "%f %f %f", &x, &y, &z) != 3) { bad = true; break; }
            P.push_back(x); P.push_back(y); P.push_back(z);
        } else if (c[0] == 'f' && (c[1] == ' ' || c[1] == '\t')) {
            std::vector<uint32_t> poly;
            const char* q = c + 1;
            for (;;) {
                while (*q == ' ' || *q == '\t') ++q;
                if (*q == 0 || *q == '\n' || *q == '\r') break;
                char* endp = nullptr;
                const long v = strtol(q, &endp, 10);
                if (endp == q) { bad = true; break; }
                const long nverts = (long)(P.size() / 3);
                const long k = v > 0 ? v - 1 : nverts + v;   // negative = relative to the vertices read so far
                if (k < 0 || k >= nverts) { bad = true; break; }
                poly.push_back((uint32_t)k);
                q = endp;
                while (*q && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') ++q;   // skip /vt/vn
            }
            if (bad) break;
            for (size_t k = 1; k + 1 < poly.size(); ++k) { I.push_back(poly[0]); I.push_back(poly[k]); I.push_back(poly[k + 1]); }
        }
    }
    fclose(f);
    if (bad) return set_error(BLUB_ERR_PARSE, (std::string("malformed OBJ record in ") + path).c_str());
    *nv_out = (uint32_t)(P.size() / 3); *ni_out = (uint32_t)I.size();
    if (pos) { if (vcap < P.size() / 3) return set_error(BLUB_ERR_INVALID_ARGUMENT, "vertex buffer too small"); if (!P.empty()) memcpy(pos, P.data(), P.size() * sizeof(float)); }
    if (idx) { if (icap < I.size()) return set_error(BLUB_ERR_INVALID_ARGUMENT, "index buffer too small"); if (!I.empty()) memcpy(idx, I.data(), I.size() * sizeof(uint32_t)); }
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
int blub_scene_mesh_desc_at_time(const blub_scene_config* scene, uint32_t object_index, uint64_t total_simulated_time_ns, uint64_t simulation_delta_ns, blub_mesh_desc* out) {
    return blub::scene_mesh_desc_at_time(scene, object_index, total_simulated_time_ns, simulation_delta_ns, out);
}
int blub_load_obj(const char* path, float* positions_xyz, size_t vertex_capacity, uint32_t* num_vertices, uint32_t* indices, size_t index_capacity, uint32_t* num_indices) {
    return blub::load_obj(path, positions_xyz, vertex_capacity, num_vertices, indices, index_capacity, num_indices);
}
int blub_seed_fluid_cube(const uint32_t grid_dim[3], uint32_t max_num_particles, uint32_t num_particles_before,
                         const float min_grid[3], const float max_grid[3], float* pos_ll_out, size_t capacity, uint32_t* count_out) {
    return blub::seed_fluid_cube(grid_dim, max_num_particles, num_particles_before, min_grid, max_grid, pos_ll_out, capacity, count_out);
}
}
