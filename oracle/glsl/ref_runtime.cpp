// =====================================================================================================
// ref_runtime.cpp -- dispatcher for the reference's compute shaders compiled through glsl_shim.h
// (TEST INFRASTRUCTURE, oracle/_ref recipe; see glsl_shim.h).  One dispatch = the Vulkan execution model
// serialised: workgroups in ascending (z, y, x) order, invocations in ascending gl_LocalInvocationIndex;
// shaders that call barrier() run one fiber (ucontext) per invocation of the workgroup in lock step.
//
// C ABI (used by oracle/glsl/ref_fluid.py through ctypes):
//   ref_bind_volume(shader, name, ptr, fmt, nx, ny, nz)   texture3D / image3D / utexture3D / uimage3D
//   ref_bind_buffer(shader, name, ptr, count)              storage buffers (element count for robust access)
//   ref_set(shader, name, data, nbytes)                    uniform / push-constant members, samplers
//   ref_dispatch(shader, gx, gy, gz)
//   ref_set_mode(oob_lod_mode, filter_mode)
// =====================================================================================================
#include "glsl_shim.h"

#include <ucontext.h>

#include <cstdlib>

namespace glsl {

int oob_lod_mode = 0;
int filter_mode = 0;
uvec3 gl_NumWorkGroups, gl_WorkGroupID, gl_LocalInvocationID, gl_GlobalInvocationID;
uint gl_LocalInvocationIndex;

static std::map<std::string, Shader*>& registry() { static std::map<std::string, Shader*> r; return r; }
Shader::Shader(const char* n) : name(n) { registry()[name] = this; }

// ---- sampling (Vulkan 1.2: 16.5.4 "(u,v,w,a) Integer Coordinate Transformations", 16.8 "Texel Filtering") -------
static inline int clamp_i(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
vec4 sample_volume(const Volume& v, int linear, const vec3& s) {
    if (!v.data) return vec4(F32(0.0f));
    const float u = s.x.x * (float)v.nx, vv = s.y.x * (float)v.ny, w = s.z.x * (float)v.nz;   // unnormalised coordinates
    if (!linear) {   // nearest: i = floor(u), clamp to edge
        ivec3 c(clamp_i((int)std::floor(u), 0, v.nx - 1), clamp_i((int)std::floor(vv), 0, v.ny - 1), clamp_i((int)std::floor(w), 0, v.nz - 1));
        return vol_load_f(v, c);
    }
    const float um = u - 0.5f, vm = vv - 0.5f, wm = w - 0.5f;
    const float fu = std::floor(um), fv = std::floor(vm), fw = std::floor(wm);
    float a = um - fu, b = vm - fv, g = wm - fw;   // alpha, beta, gamma = frac(u - 1/2)
    if (filter_mode == 1) { a = std::floor(a * 256.0f) / 256.0f; b = std::floor(b * 256.0f) / 256.0f; g = std::floor(g * 256.0f) / 256.0f; }
    const int i0 = clamp_i((int)fu, 0, v.nx - 1), i1 = clamp_i((int)fu + 1, 0, v.nx - 1);
    const int j0 = clamp_i((int)fv, 0, v.ny - 1), j1 = clamp_i((int)fv + 1, 0, v.ny - 1);
    const int k0 = clamp_i((int)fw, 0, v.nz - 1), k1 = clamp_i((int)fw + 1, 0, v.nz - 1);
    vec4 r;
    for (uint ch = 0; ch < 4; ++ch) {
        auto t = [&](int i, int j, int k) { return vol_load_f(v, ivec3(i, j, k))[ch].x; };
        if (filter_mode == 2) {   // separable form: lerp along x, then y, then z, each lerp = a*(1-t) + b*t (what the CPU oracle does)
            auto lerp = [](float p, float q, float w) { return p * (1.0f - w) + q * w; };
            const float c00 = lerp(t(i0, j0, k0), t(i1, j0, k0), a), c10 = lerp(t(i0, j1, k0), t(i1, j1, k0), a);
            const float c01 = lerp(t(i0, j0, k1), t(i1, j0, k1), a), c11 = lerp(t(i0, j1, k1), t(i1, j1, k1), a);
            r[ch] = F32(lerp(lerp(c00, c10, b), lerp(c01, c11, b), g));
            continue;
        }
        // tau = (1-a)(1-b)(1-g) t000 + a(1-b)(1-g) t100 + (1-a) b (1-g) t010 + a b (1-g) t110 + ... (16.8.3), summed in that order
        const float na = 1.0f - a, nb = 1.0f - b, ng = 1.0f - g;
        float acc = na * nb * ng * t(i0, j0, k0);
        acc = acc + a * nb * ng * t(i1, j0, k0);
        acc = acc + na * b * ng * t(i0, j1, k0);
        acc = acc + a * b * ng * t(i1, j1, k0);
        acc = acc + na * nb * g * t(i0, j0, k1);
        acc = acc + a * nb * g * t(i1, j0, k1);
        acc = acc + na * b * g * t(i0, j1, k1);
        acc = acc + a * b * g * t(i1, j1, k1);
        r[ch] = F32(acc);
    }
    return r;
}

// ---- fibers -----------------------------------------------------------------------------------------
struct Fiber { ucontext_t ctx; bool done; uvec3 lid; uint lindex; };
static ucontext_t g_sched;
static Fiber* g_current = nullptr;
static void (*g_entry)() = nullptr;
static const size_t kStack = 96 * 1024;

void barrier() {
    if (!g_current) { std::fprintf(stderr, "glsl shim: barrier() outside a fiber\n"); std::abort(); }
    swapcontext(&g_current->ctx, &g_sched);
}
static void fiber_main() { g_entry(); g_current->done = true; swapcontext(&g_current->ctx, &g_sched); }

static void run_workgroup_fibers(Shader& sh) {
    const uvec3 ls = sh.local_size;
    const uint n = ls.x * ls.y * ls.z;
    static std::vector<Fiber> fibers;
    static std::vector<char> stacks;
    if (fibers.size() < n) { fibers.resize(n); stacks.resize((size_t)n * kStack); }
    g_entry = sh.entry;
    for (uint i = 0; i < n; ++i) {
        Fiber& f = fibers[i];
        f.done = false; f.lindex = i;
        f.lid = uvec3(i % ls.x, (i / ls.x) % ls.y, i / (ls.x * ls.y));
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = stacks.data() + (size_t)i * kStack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, fiber_main, 0);
    }
    uint live = n;
    while (live) {   // one pass = every live invocation runs up to its next barrier()
        live = 0;
        for (uint i = 0; i < n; ++i) {
            Fiber& f = fibers[i];
            if (f.done) continue;
            gl_LocalInvocationID = f.lid; gl_LocalInvocationIndex = f.lindex;
            gl_GlobalInvocationID = gl_WorkGroupID * ls + f.lid;
            g_current = &f;
            swapcontext(&g_sched, &f.ctx);
            g_current = nullptr;
            if (!f.done) ++live;
        }
    }
}

static void run_workgroup_plain(Shader& sh) {
    const uvec3 ls = sh.local_size;
    uint idx = 0;
    for (uint z = 0; z < ls.z; ++z)
        for (uint y = 0; y < ls.y; ++y)
            for (uint x = 0; x < ls.x; ++x, ++idx) {
                gl_LocalInvocationID = uvec3(x, y, z); gl_LocalInvocationIndex = idx;
                gl_GlobalInvocationID = gl_WorkGroupID * ls + gl_LocalInvocationID;
                sh.entry();
            }
}

}  // namespace glsl

using namespace glsl;

static Shader* find(const char* name) {
    auto it = glsl::registry().find(name);
    return it == glsl::registry().end() ? nullptr : it->second;
}
static RegEntry* find_reg(const char* shader, const char* name, int kind) {
    Shader* s = find(shader);
    if (!s) { std::fprintf(stderr, "ref: unknown shader %s\n", shader); return nullptr; }
    auto it = s->regs.find(name);
    if (it == s->regs.end() || it->second.kind != kind) { std::fprintf(stderr, "ref: shader %s has no binding %s of kind %d\n", shader, name, kind); return nullptr; }
    return &it->second;
}

extern "C" {

int ref_num_shaders() { return (int)glsl::registry().size(); }
const char* ref_shader_name(int i) { for (auto& kv : glsl::registry()) if (i-- == 0) return kv.first.c_str(); return nullptr; }
int ref_num_bindings(const char* shader) { Shader* s = find(shader); return s ? (int)s->regs.size() : -1; }
const char* ref_binding_name(const char* shader, int i, int* kind, int* size) {
    Shader* s = find(shader);
    if (!s) return nullptr;
    for (auto& kv : s->regs) if (i-- == 0) { *kind = kv.second.kind; *size = (int)kv.second.size; return kv.first.c_str(); }
    return nullptr;
}
int ref_local_size(const char* shader, unsigned* out) { Shader* s = find(shader); if (!s) return -1; out[0] = s->local_size.x; out[1] = s->local_size.y; out[2] = s->local_size.z; return 0; }

int ref_bind_volume(const char* shader, const char* name, void* ptr, int fmt, int nx, int ny, int nz) {
    RegEntry* r = find_reg(shader, name, REG_VOLUME);
    if (!r) return -1;
    Volume* v = (Volume*)r->addr;
    v->data = ptr; v->fmt = fmt; v->nx = nx; v->ny = ny; v->nz = nz;
    return 0;
}
int ref_bind_buffer(const char* shader, const char* name, void* ptr, unsigned long long count) {
    Shader* s = find(shader);
    if (!s) return -1;
    auto it = s->regs.find(name);
    if (it == s->regs.end()) { std::fprintf(stderr, "ref: shader %s has no buffer %s\n", shader, name); return -1; }
    if (it->second.kind == REG_BUFFER) {
        struct Raw { void* p; size_t n; };   // Buffer<T> begins with {T* p; size_t n}
        Raw* b = (Raw*)it->second.addr;
        b->p = ptr; b->n = (size_t)count;
        return 0;
    }
    if (it->second.kind == REG_BLOCKPTR) { *(void**)it->second.addr = ptr; return 0; }
    return -1;
}
int ref_set(const char* shader, const char* name, const void* data, int nbytes) {
    RegEntry* r = find_reg(shader, name, REG_UNIFORM);
    if (!r) return -1;
    if ((size_t)nbytes > r->size) { std::fprintf(stderr, "ref: %s.%s holds %zu bytes, %d given\n", shader, name, r->size, nbytes); return -2; }
    std::memcpy(r->addr, data, (size_t)nbytes);
    return 0;
}
void ref_set_mode(int oob_lod, int filter) { glsl::oob_lod_mode = oob_lod; glsl::filter_mode = filter; }

int ref_dispatch(const char* shader, unsigned gx, unsigned gy, unsigned gz) {
    Shader* s = find(shader);
    if (!s || !s->entry) return -1;
    gl_NumWorkGroups = uvec3(gx, gy, gz);
    for (unsigned z = 0; z < gz; ++z)
        for (unsigned y = 0; y < gy; ++y)
            for (unsigned x = 0; x < gx; ++x) {
                gl_WorkGroupID = uvec3(x, y, z);
                if (s->uses_barrier) run_workgroup_fibers(*s); else run_workgroup_plain(*s);
            }
    return 0;
}

}  // extern "C"
