// Solid voxelisation of the scene's static objects -- compute restatement of the reference's
// rasteriser pass: scene/voxelization.rs:116-157 draws every mesh with `conservative: true` through
// shader/voxelize/conservative_hull.vert (dominant-axis projection) and conservative_hull.frag (three image stores per
// fragment).  Here VOXELIZE_SPLIT waves share one triangle and walk the pixels of its bounding box; a pixel produces a fragment when
// its unit square overlaps the projected triangle (overestimating conservative rasterisation, exact separating-axis test).
// Choices the reference leaves to the Vulkan implementation, fixed here and in the oracle:
//   * window coordinates are the swizzled voxel coordinates themselves (no sub-pixel snapping),
//   * gl_FragCoord.z = the triangle's plane at the pixel centre, clamped to the triangle's depth range; fragments outside
//     0 <= z <= viewport are clipped; dFdxCoarse/dFdyCoarse(z) = the plane's slopes,
//   * degenerate (zero-area) projections produce no fragments,
//   * the RGBA16F store rounds to nearest even.
#pragma once
#include <hip/hip_fp16.h>

#include "blub_kernels.hip.h"

namespace blubk {

constexpr int VOXELIZE_SPLIT = 16;      // waves per triangle (blockIdx.y): a cube face is two triangles of thousands of pixels each
struct MeshDesc { float m[3][4]; float vel[3]; float axis[3]; uint32_t index_begin, index_end; };

__device__ __forceinline__ float f16_round(float v) { return __half2float(__float2half_rn(v)); }

// conservative_hull.frag:17-18
__device__ __forceinline__ void unswizzle_clamp(const Grid& g, int side, float sx, float sy, float sz, float out[3]) {
    float x, y, z;
    if (side == 0) { x = sz; y = sy; z = sx; } else if (side == 1) { x = sx; y = sz; z = sy; } else { x = sx; y = sy; z = sz; }
    out[0] = fminf(fmaxf(x, 0.0f), (float)g.nx - 1.0f);
    out[1] = fminf(fmaxf(y, 0.0f), (float)g.ny - 1.0f);
    out[2] = fminf(fmaxf(z, 0.0f), (float)g.nz - 1.0f);
}
// conservative_hull.frag:20-26 + the imageStore
__device__ __forceinline__ void store_voxel(const Grid& g, const MeshDesc& d, const float vp[3], float4* __restrict__ solid, int z_lo, int z_hi) {
    const float px = vp[0] - d.m[0][3], py = vp[1] - d.m[1][3], pz = vp[2] - d.m[2][3];
    const float dt = px * d.axis[0] + py * d.axis[1] + pz * d.axis[2];
    const float tx = px - dt * d.axis[0], ty = py - dt * d.axis[1], tz = pz - dt * d.axis[2];
    const float vx = (d.axis[1] * tz - d.axis[2] * ty) + d.vel[0];
    const float vy = (d.axis[2] * tx - d.axis[0] * tz) + d.vel[1];
    const float vz = (d.axis[0] * ty - d.axis[1] * tx) + d.vel[2];
    const int ix = (int)vp[0], iy = (int)vp[1], iz = (int)vp[2];
    if (iz < z_lo || iz >= z_hi) return;      // (a z-slab holds planes [z_lo, z_hi) only)
    solid[cidx(g, ix, iy, iz)] = make_float4(f16_round(vx), f16_round(vy), f16_round(vz), 1.0f);
}

__global__ __launch_bounds__(256) void k_voxelize_mesh(Grid g, MeshDesc d, const float* __restrict__ positions, const uint32_t* __restrict__ indices, float4* __restrict__ solid, int z_lo, int z_hi) {
    const uint32_t tri = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint32_t first = d.index_begin + tri * 3;
    if (first + 3 > d.index_end) return;
    // conservative_hull.vert:14-33
    float v[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t vi = indices[first + k];
        const float px = positions[3 * (size_t)vi], py = positions[3 * (size_t)vi + 1], pz = positions[3 * (size_t)vi + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r) v[k][r] = px * d.m[r][0] + py * d.m[r][1] + pz * d.m[r][2] + d.m[r][3];
    }
    const float e1x = v[1][0] - v[0][0], e1y = v[1][1] - v[0][1], e1z = v[1][2] - v[0][2];
    const float e2x = v[2][0] - v[0][0], e2y = v[2][1] - v[0][1], e2z = v[2][2] - v[0][2];
    const float nx = fabsf(e1y * e2z - e1z * e2y), ny = fabsf(e1z * e2x - e1x * e2z), nz = fabsf(e1x * e2y - e1y * e2x);
    int side = nx > ny ? 0 : 1;
    side = (side == 0 ? nx : ny) > nz ? side : 2;
    float ax, ay, az, bx, by, bz, cx, cy, cz;   // swizzled vertices
    if (side == 0) { ax = v[0][2]; ay = v[0][1]; az = v[0][0]; bx = v[1][2]; by = v[1][1]; bz = v[1][0]; cx = v[2][2]; cy = v[2][1]; cz = v[2][0]; }
    else if (side == 1) { ax = v[0][0]; ay = v[0][2]; az = v[0][1]; bx = v[1][0]; by = v[1][2]; bz = v[1][1]; cx = v[2][0]; cy = v[2][2]; cz = v[2][1]; }
    else { ax = v[0][0]; ay = v[0][1]; az = v[0][2]; bx = v[1][0]; by = v[1][1]; bz = v[1][2]; cx = v[2][0]; cy = v[2][1]; cz = v[2][2]; }
    float area2 = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
    if (!(area2 != 0.0f) || !(fabsf(area2) < 3.0e38f)) return;   // degenerate or non-finite
    if (area2 < 0.0f) { float t; t = bx; bx = cx; cx = t; t = by; by = cy; cy = t; t = bz; bz = cz; cz = t; area2 = -area2; }
    const float viewport = (float)max(g.nx, max(g.ny, g.nz));   // voxelization.rs:99
    const float dzdx = ((bz - az) * (cy - ay) - (cz - az) * (by - ay)) / area2;
    const float dzdy = ((cz - az) * (bx - ax) - (bz - az) * (cx - ax)) / area2;
    const float zmin = fminf(az, fminf(bz, cz)), zmax = fmaxf(az, fmaxf(bz, cz));
    const float max_change = fmaxf(fabsf(dzdx), fabsf(dzdy));   // conservative_hull.frag:39-44
    const float xlo = fmaxf(floorf(fminf(ax, fminf(bx, cx))), 0.0f), xhi = fminf(floorf(fmaxf(ax, fmaxf(bx, cx))), viewport - 1.0f);
    const float ylo = fmaxf(floorf(fminf(ay, fminf(by, cy))), 0.0f), yhi = fminf(floorf(fmaxf(ay, fmaxf(by, cy))), viewport - 1.0f);
    if (!(xhi >= xlo) || !(yhi >= ylo)) return;
    const int x0 = (int)xlo, y0 = (int)ylo, w = (int)xhi - x0 + 1, hgt = (int)yhi - y0 + 1;
    // edge functions of the counter-clockwise triangle, evaluated at the corner of the pixel square that maximises them
    const float e0x = bx - ax, e0y = by - ay, e1x2 = cx - bx, e1y2 = cy - by, e2x2 = ax - cx, e2y2 = ay - cy;
    // the pixels of the bounding box are dealt out to gridDim.y waves (a box is at most 4096^2 pixels: 32-bit arithmetic); every store of a voxel
    // writes a function of the voxel's own coordinates, so the order in which fragments land does not matter
    const uint32_t total = (uint32_t)w * (uint32_t)hgt, stride = 64u * gridDim.y;
    for (uint32_t k = (uint32_t)lane + 64u * blockIdx.y; k < total; k += stride) {
        const int i = x0 + (int)(k % (uint32_t)w), j = y0 + (int)(k / (uint32_t)w);
        const float fi = (float)i, fj = (float)j;
        const float c0x = e0y < 0.0f ? fi + 1.0f : fi, c0y = e0x > 0.0f ? fj + 1.0f : fj;
        const float c1x = e1y2 < 0.0f ? fi + 1.0f : fi, c1y = e1x2 > 0.0f ? fj + 1.0f : fj;
        const float c2x = e2y2 < 0.0f ? fi + 1.0f : fi, c2y = e2x2 > 0.0f ? fj + 1.0f : fj;
        const float E0 = e0x * (c0y - ay) - e0y * (c0x - ax);
        const float E1 = e1x2 * (c1y - by) - e1y2 * (c1x - bx);
        const float E2 = e2x2 * (c2y - cy) - e2y2 * (c2x - cx);
        if (!(E0 >= 0.0f && E1 >= 0.0f && E2 >= 0.0f)) continue;
        const float pcx = fi + 0.5f, pcy = fj + 0.5f;
        float z = az + dzdx * (pcx - ax) + dzdy * (pcy - ay);
        z = fminf(fmaxf(z, zmin), zmax);
        if (!(z >= 0.0f && z <= viewport)) continue;   // depth clipping
        float vp[3];
        unswizzle_clamp(g, side, truncf(pcx), truncf(pcy), truncf(z), vp);   // conservative_hull.frag:35-36
        store_voxel(g, d, vp, solid, z_lo, z_hi);
        if (floorf(z) != floorf(z - max_change)) { unswizzle_clamp(g, side, pcx, pcy, z - 1.0f, vp); store_voxel(g, d, vp, solid, z_lo, z_hi); }   // :46-49
        if (floorf(z) != floorf(z + max_change)) { unswizzle_clamp(g, side, pcx, pcy, z + 1.0f, vp); store_voxel(g, d, vp, solid, z_lo, z_hi); }   // :50-53
    }
}

}  // namespace blubk
