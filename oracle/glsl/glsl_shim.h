// =====================================================================================================
// glsl_shim.h -- enough of GLSL 4.50 (compute) as C++17 to compile the reference's UNMODIFIED
// shader/simulation/**/*.comp with g++ and run one dispatch at a time on the CPU.
//
// TEST INFRASTRUCTURE (oracle/): this is the recipe side of oracle/_ref/ -- the reference's own shader text,
// read from /root/reference where it lies, is turned into translation units by glsl2cpp.py (a LEXICAL
// preprocessor: #include flattening, `layout(...)` interface declarations -> C++ declarations, in/out/inout
// parameter qualifiers, float literals typed as f32) and compiled against this header.  Every arithmetic
// statement the library executes is the reference author's.
//
// Semantics chosen where Vulkan/GLSL leave room (documented because they ARE the pin):
//   * scalar and vector arithmetic is IEEE binary32, one rounding per operator, no contraction
//     (build with -ffp-contract=off); `/` and sqrt are correctly rounded
//   * built-ins follow the GLSL specification's defining expressions: mix(a,b,t) = a*(1-t) + b*t,
//     fract(x) = x - floor(x), clamp = min(max(x,lo),hi), dot = ((x*x + y*y) + z*z) + w*w, length = sqrt(dot)
//   * images / texel fetches: out-of-bounds loads return 0, out-of-bounds stores are dropped
//     (robustImageAccess); storage buffers: OOB reads 0, OOB writes dropped (robustBufferAccess)
//   * r8_snorm: store round(clamp(v,-1,1)*127), load max(v/127, -1)
//   * texelFetch with a LOD beyond the texture's single mip level: `oob_lod_mode` 0 = returns 0 (SURVEY Q1
//     reading "zero"), 1 = clamps to level 0 (reading "lod0")
//   * linear filtering (textureLod with the trilinear sampler): Vulkan's unnormalised-coordinate formula,
//     clamp-to-edge, weights in full f32 (filter_mode 0) or quantised to 8 fractional bits (filter_mode 1); filter_mode 2
//     evaluates the same interpolant separably (x, then y, then z) -- real samplers are free to do either
//   * invocations of a workgroup run in ascending gl_LocalInvocationIndex order, workgroups in ascending
//     (z, y, x) order; barrier() is a real rendez-vous (one fiber per invocation), so atomics resolve in
//     ascending invocation order
// =====================================================================================================
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace glsl {

typedef unsigned int uint;

struct vec2; struct vec3; struct vec4; struct ivec2; struct ivec3; struct uvec3; struct uvec4;

// ---- f32 scalar with GLSL member access (`x.xxxx`, `f().x`) ------------------------------------------
struct F32_xxxx { float v; inline operator vec4() const; };
struct F32 {
    union { float x; float r; F32_xxxx xxxx; };
    F32() = default;
    F32(float v) : x(v) {}
    F32(double v) : x((float)v) {}
    F32(int v) : x((float)v) {}
    F32(uint v) : x((float)v) {}
    F32(bool v) : x(v ? 1.0f : 0.0f) {}
    F32& operator+=(F32 b) { x = x + b.x; return *this; }
    F32& operator-=(F32 b) { x = x - b.x; return *this; }
    F32& operator*=(F32 b) { x = x * b.x; return *this; }
    F32& operator/=(F32 b) { x = x / b.x; return *this; }
};
inline F32 operator+(F32 a, F32 b) { return F32(a.x + b.x); }
inline F32 operator-(F32 a, F32 b) { return F32(a.x - b.x); }
inline F32 operator*(F32 a, F32 b) { return F32(a.x * b.x); }
inline F32 operator/(F32 a, F32 b) { return F32(a.x / b.x); }
inline F32 operator-(F32 a) { return F32(-a.x); }
inline bool operator==(F32 a, F32 b) { return a.x == b.x; }
inline bool operator!=(F32 a, F32 b) { return a.x != b.x; }
inline bool operator<(F32 a, F32 b) { return a.x < b.x; }
inline bool operator>(F32 a, F32 b) { return a.x > b.x; }
inline bool operator<=(F32 a, F32 b) { return a.x <= b.x; }
inline bool operator>=(F32 a, F32 b) { return a.x >= b.x; }

inline float raw(F32 a) { return a.x; }
inline float raw(float a) { return a; }
inline float raw(double a) { return (float)a; }
inline float raw(int a) { return (float)a; }
inline float raw(uint a) { return (float)a; }
inline int toi(F32 a) { return (int)a.x; }   // float -> int conversion truncates toward zero
inline int toi(float a) { return (int)a; }
inline int toi(int a) { return a; }
inline int toi(uint a) { return (int)a; }
inline uint tou(F32 a) { return (uint)a.x; }
inline uint tou(int a) { return (uint)a; }
inline uint tou(uint a) { return a; }

// ---- swizzle proxies: a view of the owning vector's first lanes (union member at offset 0) ----------
// (g++ refuses anonymous structs whose members have constructors, so every lane is its own anonymous union and a proxy
//  sits in the union of its FIRST lane and reaches the following lanes through the vector's contiguous storage)
template <class V2, class S, int A, int B> struct Swz2 { S first; operator V2() const { const S* d = &first; return V2(d[A], d[B]); } };
template <class V3, class S, int A, int B, int C> struct Swz3 { S first; operator V3() const { const S* d = &first; return V3(d[A], d[B], d[C]); } };

// ---- float vectors ----------------------------------------------------------------------------------
struct vec2 {
    union { F32 x; F32 r; };
    union { F32 y; F32 g; };
    vec2() = default;
    explicit vec2(F32 s) : x(s), y(s) {}
    template <class A, class B> vec2(A a, B b) : x(raw(a)), y(raw(b)) {}
    F32& operator[](uint i) { return (&x)[i]; }
    F32 operator[](uint i) const { return (&x)[i]; }
};
struct vec3 {
    union { F32 x; F32 r; Swz2<vec2, F32, 0, 1> xy; Swz3<vec3, F32, 0, 1, 2> xyz; };
    union { F32 y; F32 g; Swz2<vec2, F32, 0, 1> yz; };
    union { F32 z; F32 b; };
    vec3() = default;
    explicit vec3(F32 s) : x(s), y(s), z(s) {}
    template <class A, class B, class C> vec3(A a, B b, C c) : x(raw(a)), y(raw(b)), z(raw(c)) {}
    vec3(const vec2& a, F32 c) : x(a.x), y(a.y), z(c) {}
    vec3(F32 a, const vec2& b) : x(a), y(b.x), z(b.y) {}
    inline vec3(const ivec3& v);   // GLSL implicit conversions int -> float, uint -> float
    inline vec3(const uvec3& v);
    F32& operator[](uint i) { return (&x)[i]; }
    F32 operator[](uint i) const { return (&x)[i]; }
};
struct vec4 {
    union { F32 x; F32 r; Swz2<vec2, F32, 0, 1> xy; Swz3<vec3, F32, 0, 1, 2> xyz; };
    union { F32 y; F32 g; };
    union { F32 z; F32 b; };
    union { F32 w; F32 a; };
    vec4() = default;
    explicit vec4(F32 s) : x(s), y(s), z(s), w(s) {}
    template <class A, class B, class C, class D> vec4(A a, B b, C c, D d) : x(raw(a)), y(raw(b)), z(raw(c)), w(raw(d)) {}
    template <class D> vec4(const vec3& v, D d) : x(v.x), y(v.y), z(v.z), w(raw(d)) {}
    F32& operator[](uint i) { return (&x)[i]; }
    F32 operator[](uint i) const { return (&x)[i]; }
};
inline F32_xxxx::operator vec4() const { return vec4(F32(v)); }

// ---- integer vectors --------------------------------------------------------------------------------
struct uvec2 { uint x, y; };
struct ivec2 {
    int x, y;
    ivec2() = default;
    explicit ivec2(int s) : x(s), y(s) {}
    ivec2(int a, int b) : x(a), y(b) {}
};
struct ivec3 {
    union { int x; Swz2<ivec2, int, 0, 1> xy; Swz3<ivec3, int, 0, 1, 2> xyz; };
    union { int y; Swz2<ivec2, int, 0, 1> yz; };
    int z;
    ivec3() = default;
    explicit ivec3(int s) : x(s), y(s), z(s) {}
    template <class A, class B, class C> ivec3(A a, B b, C c) : x(toi(a)), y(toi(b)), z(toi(c)) {}
    ivec3(int a, const ivec2& b) : x(a), y(b.x), z(b.y) {}
    ivec3(const ivec2& a, int c) : x(a.x), y(a.y), z(c) {}
    explicit ivec3(const vec3& v) : x(toi(v.x)), y(toi(v.y)), z(toi(v.z)) {}
    explicit inline ivec3(const uvec3& v);
    int& operator[](uint i) { return (&x)[i]; }
    int operator[](uint i) const { return (&x)[i]; }
};
struct uvec3 {
    union { uint x; Swz3<uvec3, uint, 0, 1, 2> xyz; };
    uint y, z;
    uvec3() = default;
    explicit uvec3(uint s) : x(s), y(s), z(s) {}
    template <class A, class B, class C> uvec3(A a, B b, C c) : x(tou(a)), y(tou(b)), z(tou(c)) {}
    uvec3(const ivec3& v) : x((uint)v.x), y((uint)v.y), z((uint)v.z) {}   // GLSL implicit conversion int -> uint
    uint& operator[](uint i) { return (&x)[i]; }
    uint operator[](uint i) const { return (&x)[i]; }
};
struct uvec4 {
    union { uint x; uint r; };
    union { uint y; uint g; };
    union { uint z; uint b; };
    union { uint w; uint a; };
    uvec4() = default;
    explicit uvec4(uint s) : x(s), y(s), z(s), w(s) {}
    template <class A, class B, class C, class D> uvec4(A a, B b, C c, D d) : x(tou(a)), y(tou(b)), z(tou(c)), w(tou(d)) {}
    template <class D> uvec4(const uvec3& v, D d) : x(v.x), y(v.y), z(v.z), w(tou(d)) {}
    uint& operator[](uint i) { return (&x)[i]; }
};
struct bvec3 { bool x, y, z; };

inline vec3::vec3(const ivec3& v) : x((float)v.x), y((float)v.y), z((float)v.z) {}
inline vec3::vec3(const uvec3& v) : x((float)v.x), y((float)v.y), z((float)v.z) {}
inline ivec3::ivec3(const uvec3& v) : x((int)v.x), y((int)v.y), z((int)v.z) {}

// matrices only appear in interface blocks the simulation shaders never read
struct mat4 { vec4 c[4]; };
struct mat3x4 { vec4 c[3]; };

#define GLSL_VEC_OPS(V, S, LANES)                                                                                  \
    inline V operator+(const V& a, const V& b) { V r; for (uint i = 0; i < LANES; ++i) r[i] = a[i] + b[i]; return r; } \
    inline V operator-(const V& a, const V& b) { V r; for (uint i = 0; i < LANES; ++i) r[i] = a[i] - b[i]; return r; } \
    inline V operator*(const V& a, const V& b) { V r; for (uint i = 0; i < LANES; ++i) r[i] = a[i] * b[i]; return r; } \
    inline V operator/(const V& a, const V& b) { V r; for (uint i = 0; i < LANES; ++i) r[i] = a[i] / b[i]; return r; } \
    inline V operator+(const V& a, S b) { V r; for (uint i = 0; i < LANES; ++i) r[i] = a[i] + b; return r; }           \
    inline V operator-(const V& a, S b) { V r; for (uint i = 0; i < LANES; ++i) r[i] = a[i] - b; return r; }           \
    inline V operator*(const V& a, S b) { V r; for (uint i = 0; i < LANES; ++i) r[i] = a[i] * b; return r; }           \
    inline V operator/(const V& a, S b) { V r; for (uint i = 0; i < LANES; ++i) r[i] = a[i] / b; return r; }           \
    inline V operator+(S a, const V& b) { V r; for (uint i = 0; i < LANES; ++i) r[i] = a + b[i]; return r; }           \
    inline V operator-(S a, const V& b) { V r; for (uint i = 0; i < LANES; ++i) r[i] = a - b[i]; return r; }           \
    inline V operator*(S a, const V& b) { V r; for (uint i = 0; i < LANES; ++i) r[i] = a * b[i]; return r; }           \
    inline V operator/(S a, const V& b) { V r; for (uint i = 0; i < LANES; ++i) r[i] = a / b[i]; return r; }           \
    inline V operator-(const V& a) { V r; for (uint i = 0; i < LANES; ++i) r[i] = -a[i]; return r; }                   \
    inline V& operator+=(V& a, const V& b) { a = a + b; return a; }                                                    \
    inline V& operator-=(V& a, const V& b) { a = a - b; return a; }                                                    \
    inline V& operator*=(V& a, S b) { a = a * b; return a; }                                                           \
    inline bool operator==(const V& a, const V& b) { bool e = true; for (uint i = 0; i < LANES; ++i) e = e && (a[i] == b[i]); return e; } \
    inline bool operator!=(const V& a, const V& b) { return !(a == b); }
GLSL_VEC_OPS(vec2, F32, 2)
GLSL_VEC_OPS(vec3, F32, 3)
GLSL_VEC_OPS(vec4, F32, 4)
GLSL_VEC_OPS(ivec3, int, 3)
inline uvec3 operator+(const uvec3& a, const uvec3& b) { return uvec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline uvec3 operator-(const uvec3& a, const uvec3& b) { return uvec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline uvec3 operator*(const uvec3& a, const uvec3& b) { return uvec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline uvec3 operator/(const uvec3& a, const uvec3& b) { return uvec3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline bool operator==(const uvec3& a, const uvec3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool operator!=(const uvec3& a, const uvec3& b) { return !(a == b); }
inline ivec3 operator%(const ivec3& a, const ivec3& b) { return ivec3(a.x % b.x, a.y % b.y, a.z % b.z); }

// ---- built-in functions (GLSL 4.50 spec, chapter 8; the defining expression of each) ----------------
inline F32 abs(F32 a) { return F32(std::fabs(a.x)); }
inline int abs(int a) { return a < 0 ? -a : a; }
inline F32 sign(F32 a) { return F32(a.x > 0.0f ? 1.0f : (a.x < 0.0f ? -1.0f : 0.0f)); }
inline F32 floor(F32 a) { return F32(std::floor(a.x)); }
inline F32 fract(F32 a) { return F32(a.x - std::floor(a.x)); }
inline F32 sqrt(F32 a) { return F32(std::sqrt(a.x)); }
inline F32 min(F32 a, F32 b) { return b.x < a.x ? b : a; }   // min(x,y): y if y < x, otherwise x
inline F32 max(F32 a, F32 b) { return a.x < b.x ? b : a; }   // max(x,y): y if x < y, otherwise x
inline int min(int a, int b) { return b < a ? b : a; }
inline int max(int a, int b) { return a < b ? b : a; }
inline uint min(uint a, uint b) { return b < a ? b : a; }
inline uint max(uint a, uint b) { return a < b ? b : a; }
inline F32 clamp(F32 v, F32 lo, F32 hi) { return min(max(v, lo), hi); }
inline F32 mix(F32 a, F32 b, F32 t) { return a * (F32(1.0f) - t) + b * t; }
#define GLSL_VEC_FN1(V, LANES, fn) inline V fn(const V& a) { V r; for (uint i = 0; i < LANES; ++i) r[i] = fn(a[i]); return r; }
#define GLSL_VEC_FNS(V, LANES)                                                                                    \
    GLSL_VEC_FN1(V, LANES, abs) GLSL_VEC_FN1(V, LANES, sign) GLSL_VEC_FN1(V, LANES, floor) GLSL_VEC_FN1(V, LANES, fract) \
    inline V min(const V& a, const V& b) { V r; for (uint i = 0; i < LANES; ++i) r[i] = min(a[i], b[i]); return r; } \
    inline V max(const V& a, const V& b) { V r; for (uint i = 0; i < LANES; ++i) r[i] = max(a[i], b[i]); return r; } \
    inline V clamp(const V& v, const V& lo, const V& hi) { return min(max(v, lo), hi); }                            \
    inline V clamp(const V& v, F32 lo, F32 hi) { V r; for (uint i = 0; i < LANES; ++i) r[i] = clamp(v[i], lo, hi); return r; } \
    inline V mix(const V& a, const V& b, const V& t) { V r; for (uint i = 0; i < LANES; ++i) r[i] = mix(a[i], b[i], t[i]); return r; } \
    inline V mix(const V& a, const V& b, F32 t) { V r; for (uint i = 0; i < LANES; ++i) r[i] = mix(a[i], b[i], t); return r; }
GLSL_VEC_FNS(vec2, 2)
GLSL_VEC_FNS(vec3, 3)
GLSL_VEC_FNS(vec4, 4)
inline ivec3 min(const ivec3& a, const ivec3& b) { return ivec3(min(a.x, b.x), min(a.y, b.y), min(a.z, b.z)); }
inline ivec3 max(const ivec3& a, const ivec3& b) { return ivec3(max(a.x, b.x), max(a.y, b.y), max(a.z, b.z)); }
inline F32 dot(const vec2& a, const vec2& b) { return a.x * b.x + a.y * b.y; }
inline F32 dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline F32 dot(const vec4& a, const vec4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline F32 length(const vec3& a) { return sqrt(dot(a, a)); }
inline bvec3 equal(const uvec3& a, const uvec3& b) { return bvec3{a.x == b.x, a.y == b.y, a.z == b.z}; }
inline bvec3 equal(const ivec3& a, const ivec3& b) { return bvec3{a.x == b.x, a.y == b.y, a.z == b.z}; }
inline bool any(const bvec3& b) { return b.x || b.y || b.z; }
inline bool all(const bvec3& b) { return b.x && b.y && b.z; }
inline vec4 unpackSnorm4x8(uint p) {
    vec4 r;
    for (uint i = 0; i < 4; ++i) { float f = (float)(int8_t)((p >> (8 * i)) & 0xFF) / 127.0f; r[i] = F32(f < -1.0f ? -1.0f : f); }
    return r;
}
inline uint packSnorm4x8(const vec4& v) {
    uint p = 0;
    for (uint i = 0; i < 4; ++i) { float c = std::fmin(std::fmax(v[i].x, -1.0f), 1.0f); p |= ((uint)(uint8_t)(int8_t)std::nearbyint(c * 127.0f)) << (8 * i); }
    return p;
}

// ---- resources --------------------------------------------------------------------------------------
enum Format { FMT_R32F = 0, FMT_R8_SNORM = 1, FMT_R32UI = 2, FMT_RGBA32F = 3 };   // RGBA16F volumes are handed over as f32 quadruples holding f16 values
struct Volume { void* data = nullptr; int fmt = 0; int nx = 0, ny = 0, nz = 0; };
struct texture3D : Volume {}; struct utexture3D : Volume {}; struct image3D : Volume {}; struct uimage3D : Volume {};
struct texture2D : Volume {};
struct sampler { int linear = 0; };
struct sampler3D { Volume v; int linear; sampler3D(const Volume& t, const sampler& s) : v(t), linear(s.linear) {} };
extern int oob_lod_mode;   // 0: texelFetch beyond the last mip level returns 0; 1: clamps to level 0
extern int filter_mode;    // 0: f32 filter weights, weighted sum; 1: weights quantised to 8 fractional bits; 2: f32 weights, separable lerps

inline bool vol_in(const Volume& v, const ivec3& c) { return (uint)c.x < (uint)v.nx && (uint)c.y < (uint)v.ny && (uint)c.z < (uint)v.nz; }
inline size_t vol_idx(const Volume& v, const ivec3& c) { return ((size_t)c.z * v.ny + c.y) * v.nx + c.x; }
inline vec4 vol_load_f(const Volume& v, const ivec3& c) {
    if (!v.data || !vol_in(v, c)) return vec4(F32(0.0f));
    size_t i = vol_idx(v, c);
    switch (v.fmt) {
    case FMT_R32F: return vec4(((const float*)v.data)[i], 0.0f, 0.0f, 1.0f);
    case FMT_R8_SNORM: { float f = (float)((const int8_t*)v.data)[i] / 127.0f; return vec4(f < -1.0f ? -1.0f : f, 0.0f, 0.0f, 1.0f); }
    case FMT_RGBA32F: { const float* p = (const float*)v.data + 4 * i; return vec4(p[0], p[1], p[2], p[3]); }
    default: return vec4(F32(0.0f));
    }
}
inline uvec4 vol_load_u(const Volume& v, const ivec3& c) {
    if (!v.data || !vol_in(v, c) || v.fmt != FMT_R32UI) return uvec4(0u);
    return uvec4(((const uint*)v.data)[vol_idx(v, c)], 0u, 0u, 1u);
}
inline void vol_store_f(const Volume& v, const ivec3& c, const vec4& t) {
    if (!v.data || !vol_in(v, c)) return;
    size_t i = vol_idx(v, c);
    switch (v.fmt) {
    case FMT_R32F: ((float*)v.data)[i] = t.x.x; break;
    case FMT_R8_SNORM: { float f = std::fmin(std::fmax(t.x.x, -1.0f), 1.0f); ((int8_t*)v.data)[i] = (int8_t)std::nearbyint(f * 127.0f); break; }
    case FMT_RGBA32F: { float* p = (float*)v.data + 4 * i; p[0] = t.x.x; p[1] = t.y.x; p[2] = t.z.x; p[3] = t.w.x; break; }
    default: break;
    }
}
inline vec4 imageLoad(const image3D& v, const ivec3& c) { return vol_load_f(v, c); }
inline uvec4 imageLoad(const uimage3D& v, const ivec3& c) { return vol_load_u(v, c); }
inline void imageStore(const image3D& v, const ivec3& c, const vec4& t) { vol_store_f(v, c, t); }
inline void imageStore(const uimage3D& v, const ivec3& c, const uvec4& t) { if (v.data && vol_in(v, c)) ((uint*)v.data)[vol_idx(v, c)] = t.x; }
inline uint imageAtomicExchange(const uimage3D& v, const ivec3& c, uint val) {
    if (!v.data || !vol_in(v, c)) return 0u;
    uint* p = (uint*)v.data + vol_idx(v, c); uint old = *p; *p = val; return old;
}
inline uint imageAtomicAdd(const uimage3D& v, const ivec3& c, uint val) {
    if (!v.data || !vol_in(v, c)) return 0u;
    uint* p = (uint*)v.data + vol_idx(v, c); uint old = *p; *p = old + val; return old;
}
inline uint atomicAdd(uint& mem, uint val) { uint old = mem; mem = old + val; return old; }
inline ivec3 imageSize(const Volume& v) { return ivec3(v.nx, v.ny, v.nz); }
inline vec4 texelFetch(const texture3D& v, const ivec3& c, int lod) {
    if (lod != 0) { if (oob_lod_mode == 0) return vec4(F32(0.0f)); }
    return vol_load_f(v, c);
}
inline uvec4 texelFetch(const utexture3D& v, const ivec3& c, int lod) {
    if (lod != 0) { if (oob_lod_mode == 0) return uvec4(0u); }
    return vol_load_u(v, c);
}
// Vulkan 1.2 spec 16.6 / 16.8 (unnormalised coordinates, clamp to edge, nearest / linear)
vec4 sample_volume(const Volume& v, int linear, const vec3& uvw);
inline vec4 texture(const sampler3D& s, const vec3& uvw) { return sample_volume(s.v, s.linear, uvw); }
inline vec4 textureLod(const sampler3D& s, const vec3& uvw, int) { return sample_volume(s.v, s.linear, uvw); }

// storage buffer with robust access
template <class T> struct Buffer {
    T* p = nullptr; size_t n = 0; T dummy;
    T& operator[](uint i) { if (p && i < n) return p[i]; std::memset((void*)&dummy, 0, sizeof(T)); return dummy; }
};

// ---- execution state --------------------------------------------------------------------------------
extern uvec3 gl_NumWorkGroups, gl_WorkGroupID, gl_LocalInvocationID, gl_GlobalInvocationID;
extern uint gl_LocalInvocationIndex;
void barrier();

enum RegKind { REG_VOLUME = 0, REG_BUFFER = 1, REG_UNIFORM = 2, REG_BLOCKPTR = 3 };
struct RegEntry { void* addr; size_t size; int kind; size_t elem; };
struct Shader {
    std::string name; void (*entry)() = nullptr; uvec3 local_size = uvec3(1u); bool uses_barrier = false;
    std::map<std::string, RegEntry> regs;
    explicit Shader(const char* n);
};
struct Reg { Reg(Shader& s, const char* name, void* addr, size_t size, int kind, size_t elem = 0) { s.regs[name] = RegEntry{addr, size, kind, elem}; } };
struct Entry { Entry(Shader& s, void (*fn)(), const uvec3& ls, bool barrier_used) { s.entry = fn; s.local_size = ls; s.uses_barrier = barrier_used; } };

}  // namespace glsl

#define GLSL_LOCAL_SIZE(a, b, c) static const ::glsl::uvec3 gl_WorkGroupSize((::glsl::uint)(a), (::glsl::uint)(b), (::glsl::uint)(c));
#define GLSL_VOLUME(T, name) static ::glsl::T name; static ::glsl::Reg _reg_##name(_shader, #name, &name, sizeof(::glsl::Volume), ::glsl::REG_VOLUME);
#define GLSL_VOLUME_ARRAY(T, name, n) static ::glsl::T name[n];
#define GLSL_SAMPLER(name) static ::glsl::sampler name; static ::glsl::Reg _reg_##name(_shader, #name, &name, sizeof(int), ::glsl::REG_UNIFORM);
#define GLSL_BUFFER(T, name) static ::glsl::Buffer<T> name; static ::glsl::Reg _reg_##name(_shader, #name, &name, 0, ::glsl::REG_BUFFER, sizeof(T));
#define GLSL_UNIFORM_MEMBER(inst, name) static ::glsl::Reg _reg_##inst##_##name(_shader, #name, &inst.name, sizeof(inst.name), ::glsl::REG_UNIFORM);
#define GLSL_UNIFORM_ALIAS(inst, name) static auto& name = inst.name;
#define GLSL_BLOCKPTR(T, ptr, blockname) static T* ptr; static ::glsl::Reg _reg_##ptr(_shader, #blockname, &ptr, sizeof(T), ::glsl::REG_BLOCKPTR);
