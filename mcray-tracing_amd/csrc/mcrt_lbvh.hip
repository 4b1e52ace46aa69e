// mcrt_lbvh.hip -- BVH construction ON the GPU (gfx950): Morton-order LBVH emitted directly as the 128-byte BVH4 nodes
// k_trace walks.  Replaces the per-mesh btBvhTriangleMeshShape construction of scene::add_rigidbody_from_obj
// (scene.cpp:306-309) for the interactive case -- moved vertices or a moved mesh are re-indexed in milliseconds, without
// the host's seconds-long SAH build (the reference's own hook for this is inputmanager.cpp:117-121).
//
// The tree's SHAPE is free: the closest-hit contract makes the result independent of the hierarchy as long
// as (1) leaves carry the triangles' own padded bounds, computed exactly as k_expand_tris / the host builder compute them,
// and (2) node boxes are exact unions of their children's.  Both hold here, so images traced through this tree are
// bit-identical to those traced through the host's SAH tree (tests/test_gpu_parity.py::test_device_lbvh_*).
//
// Pipeline (all on one stream, one small read-back):
//   k_scale      largest finite |coordinate|                     -> absolute pad (same formula as mcrt_build_bvh)
//   k_prims      per triangle: padded bounds, centroid; centroid bounds by wave-reduced atomics
//   k_morton     63-bit Morton code of the centroid (21 bits per axis)
//   rocprim      radix sort (code, triangle id)
//   k_karras     binary radix tree over the sorted codes (Karras 2012; ties broken by position) + node ranges
//   k_fit        bottom-up boxes (second arrival at a node computes the union)
//   k_depth      depth of every internal node; BVH4 nodes = internal nodes at EVEN depth holding more than MCRT_LBVH_LEAF triangles
//   rocprim      exclusive scan of the flags -> BVH4 node numbering (root = 0)
//   k_emit4      each BVH4 node adopts its grandchildren (a subtree of <= MCRT_LBVH_LEAF triangles becomes one leaf: sorted order makes
//                its triangles contiguous), k_emit_tris writes the leaf-order triangle array, k_stack bounds the stack
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <stdint.h>
#include <math.h>
#include "../../include/mcrt.h"
#include "mcrt_internal.h"
#include "mcrt_lbvh.h"

namespace mcrt {
namespace {

struct Scal {
    uint32_t scale_bits;          // max finite |coordinate|, float bits (non-negative floats order like their bits)
    uint32_t cb_lo[3], cb_hi[3];  // centroid bounds, order-preserving encoding
    uint32_t max_depth, max_stack;
    float pad_abs;
};

__device__ __forceinline__ uint32_t enc(float f) { const uint32_t b = __float_as_uint(f); return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float dec(uint32_t e) { return __uint_as_float(e ^ ((e >> 31) ? 0x80000000u : 0xffffffffu)); }

__global__ void k_scal_init(Scal *s)
{
    s->scale_bits = 0u; s->max_depth = 0u; s->max_stack = 0u; s->pad_abs = 0.0f;
    for (int a = 0; a < 3; a++) { s->cb_lo[a] = 0xffffffffu; s->cb_hi[a] = 0u; }
}

__global__ void k_scale(const float *tri, size_t n9, Scal *s)
{
    uint32_t m = 0u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n9; i += (size_t)gridDim.x * blockDim.x) {
        const float a = fabsf(tri[i]);
        if (a < INFINITY) m = max(m, __float_as_uint(a));     // finite only (NaN fails the compare)
    }
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(&s->scale_bits, m);
}

__global__ void k_pad(Scal *s) { s->pad_abs = 4e-6f * fmaxf(__uint_as_float(s->scale_bits), 1e-3f); }   // == mcrt_build_bvh

// padded bounds of triangle t -- bit for bit what k_expand_tris (mcrt_kernels.hip) and mcrt_build_bvh (mcrt_host.cpp) compute
__global__ void k_prims(const float *tri, uint32_t n, Scal *s, float4 *plo, float4 *phi, float4 *pc)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    float c[3] = { 0, 0, 0 };
    const bool live = t < n;
    if (live) {
        const float *v = tri + (size_t)t * 9;
        float lo[3], hi[3], ext = 0.0f;
        for (int a = 0; a < 3; a++) {
            const float l = fminf(v[a], fminf(v[3 + a], v[6 + a]));
            const float h = fmaxf(v[a], fmaxf(v[3 + a], v[6 + a]));
            lo[a] = l; hi[a] = h; c[a] = 0.5f * (l + h);
            ext = fmaxf(ext, h - l);
        }
        const float pad = 2e-4f * ext + s->pad_abs;
        plo[t] = make_float4(lo[0] - pad, lo[1] - pad, lo[2] - pad, 0.0f);
        phi[t] = make_float4(hi[0] + pad, hi[1] + pad, hi[2] + pad, 0.0f);
        pc[t] = make_float4(c[0], c[1], c[2], 0.0f);
    }
    for (int a = 0; a < 3; a++) {
        const bool ok = live && fabsf(c[a]) < INFINITY;
        uint32_t l = ok ? enc(c[a]) : 0xffffffffu, h = ok ? enc(c[a]) : 0u;
        for (int off = 32; off > 0; off >>= 1) { l = min(l, (uint32_t)__shfl_xor((int)l, off, 64)); h = max(h, (uint32_t)__shfl_xor((int)h, off, 64)); }
        if ((threadIdx.x & 63) == 0) { if (l != 0xffffffffu) atomicMin(&s->cb_lo[a], l); if (h != 0u) atomicMax(&s->cb_hi[a], h); }
    }
}

__device__ __forceinline__ uint64_t spread21(uint32_t x)   // 21 bits -> every third bit
{
    uint64_t v = x & 0x1fffffu;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

__global__ void k_morton(const float4 *pc, uint32_t n, const Scal *s, uint64_t *keys, uint32_t *vals)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float4 c = pc[t];
    const float cc[3] = { c.x, c.y, c.z };
    uint32_t q[3];
    for (int a = 0; a < 3; a++) {
        const float lo = dec(s->cb_lo[a]), hi = dec(s->cb_hi[a]);
        const float w = hi - lo;
        float u = (w > 0.0f) ? (cc[a] - lo) / w : 0.0f;
        u = u * 2097152.0f;
        q[a] = (u >= 0.0f) ? (u < 2097151.0f ? (uint32_t)u : 2097151u) : 0u;      // NaN -> 0
    }
    keys[t] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    vals[t] = t;
}

// common-prefix length of sorted positions i and j (-1 outside the array); equal codes are told apart by position
__device__ __forceinline__ int delta(const uint64_t *keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}

// Karras, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees" (2012), section 3: internal node i
// covers a range of sorted keys that starts or ends at i.  Children: >= 0 internal node, < 0 = ~(leaf position).
__global__ void k_karras(const uint64_t *keys, int n, int2 *child, int2 *range, int *parent_int, int *parent_leaf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2; ; t = (t + 1) / 2) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int left = (lo == gamma) ? ~gamma : gamma;
    const int right = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    child[i] = make_int2(left, right);
    range[i] = make_int2(lo, hi);
    if (left >= 0) parent_int[left] = i; else parent_leaf[~left] = i;
    if (right >= 0) parent_int[right] = i; else parent_leaf[~right] = i;
    if (i == 0) parent_int[0] = -1;
}

// boxes written by other workgroups are read past the vector L1 (it is not coherent between compute units)
__device__ __forceinline__ float4 load_coherent(const float4 *p)
{
    const volatile float *q = (const volatile float *)p;
    return make_float4(q[0], q[1], q[2], q[3]);
}

__global__ void k_fit(const int2 *child, const int *parent_int, const int *parent_leaf, const uint32_t *vals, const float4 *plo, const float4 *phi,
                      int n, float4 *ilo, float4 *ihi, uint32_t *arrivals)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int p = parent_leaf[j];
    while (p >= 0) {
        __threadfence();                                   // this thread's box (if any) is visible before it signs in
        if (atomicAdd(&arrivals[p], 1u) == 0u) return;     // first child to arrive: the sibling finishes the node
        __threadfence();                                   // acquire side: the sibling's box, published before ITS arrival, is read after ours
        const int2 c = child[p];
        float4 l0, h0, l1, h1;
        if (c.x >= 0) { l0 = load_coherent(&ilo[c.x]); h0 = load_coherent(&ihi[c.x]); } else { const uint32_t t = vals[~c.x]; l0 = plo[t]; h0 = phi[t]; }
        if (c.y >= 0) { l1 = load_coherent(&ilo[c.y]); h1 = load_coherent(&ihi[c.y]); } else { const uint32_t t = vals[~c.y]; l1 = plo[t]; h1 = phi[t]; }
        ilo[p] = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), 0.0f);
        ihi[p] = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), 0.0f);
        p = parent_int[p];
    }
}

__global__ void k_depth(const int *parent_int, const int2 *range, int n, int leaf_max, uint32_t *depth, uint32_t *is4, Scal *s)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    uint32_t d = 0;
    for (int a = i; a != 0; a = parent_int[a]) d++;
    depth[i] = d;
    const int2 r = range[i];
    is4[i] = (i == 0 || ((d & 1u) == 0u && r.y - r.x + 1 > leaf_max)) ? 1u : 0u;
    uint32_t m = d + 1u;
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&s->max_depth, m);
}

struct Slot { float4 lo, hi; int ref; };

__device__ __forceinline__ Slot make_slot(int w, int leaf_max, const int2 *range, const uint32_t *id4, const uint32_t *vals, const float4 *plo, const float4 *phi,
                                          const float4 *ilo, const float4 *ihi)
{
    Slot s;
    if (w < 0) { const int j = ~w; const uint32_t t = vals[j]; s.lo = plo[t]; s.hi = phi[t]; s.ref = ~(j << 3); return s; }
    const int2 r = range[w];
    const int cnt = r.y - r.x + 1;
    s.lo = ilo[w]; s.hi = ihi[w];
    s.ref = (cnt <= leaf_max) ? ~((r.x << 3) | (cnt - 1)) : (int)id4[w];
    return s;
}

__global__ void k_emit4(const int2 *child, const int2 *range, const uint32_t *is4, const uint32_t *id4, const uint32_t *vals,
                        const float4 *plo, const float4 *phi, const float4 *ilo, const float4 *ihi, int n, int leaf_max, float4 *nodes, uint8_t *k4)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || !is4[i]) return;
    Slot s[4]; int k = 0;
    const int2 c = child[i];
    const int two[2] = { c.x, c.y };
    for (int u = 0; u < 2; u++) {
        const int y = two[u];
        const bool leaflike = y < 0 || (range[y].y - range[y].x + 1 <= leaf_max);
        if (leaflike) s[k++] = make_slot(y, leaf_max, range, id4, vals, plo, phi, ilo, ihi);
        else { const int2 g = child[y]; s[k++] = make_slot(g.x, leaf_max, range, id4, vals, plo, phi, ilo, ihi); s[k++] = make_slot(g.y, leaf_max, range, id4, vals, plo, phi, ilo, ihi); }
    }
    float4 *N = nodes + 8 * (size_t)id4[i];
    for (int u = 0; u < 4; u++) {
        if (u < k) {
            N[2 * u] = make_float4(s[u].lo.x, s[u].lo.y, s[u].lo.z, s[u].hi.x);
            N[2 * u + 1] = make_float4(s[u].hi.y, s[u].hi.z, __int_as_float(s[u].ref), 0.0f);
        } else {
            N[2 * u] = make_float4(INFINITY, INFINITY, INFINITY, -INFINITY);
            N[2 * u + 1] = make_float4(-INFINITY, -INFINITY, __int_as_float(MCRT_BVH4_EMPTY), 0.0f);
        }
    }
    k4[i] = (uint8_t)k;
}

__global__ void k_emit_tris(const float *tri, const uint32_t *mesh, const uint32_t *vals, uint32_t n, float4 *tris, uint32_t *tri_slot)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t id = vals[p];
    const float *v = tri + (size_t)id * 9;
    tris[3 * (size_t)p] = make_float4(v[0], v[1], v[2], __uint_as_float(id));
    tris[3 * (size_t)p + 1] = make_float4(v[3], v[4], v[5], __uint_as_float(mesh ? mesh[id] : 0u));
    tris[3 * (size_t)p + 2] = make_float4(v[6], v[7], v[8], 0.0f);
    tri_slot[id] = p;
}

// worst-case traversal stack of k_trace: it stacks (hit children - 1) at every node on the way down
__global__ void k_stack(const int *parent_int, const uint32_t *is4, const uint8_t *k4, int n, Scal *s)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t need = 0;
    if (i < n - 1 && is4[i]) {
        for (int a = i; ; a = parent_int[parent_int[a]]) { need += (uint32_t)k4[a] - 1u; if (a == 0) break; }
    }
    for (int off = 32; off > 0; off >>= 1) need = max(need, (uint32_t)__shfl_xor((int)need, off, 64));
    if ((threadIdx.x & 63) == 0 && need) atomicMax(&s->max_stack, need);
}

// ------------------------------------------------------------------------------------------------------------------------
// REFIT: new vertex positions, same tree.  The topology (which triangles share a leaf, which nodes share a parent) is kept, every
// box is recomputed bottom-up from the triangles' new padded bounds -- exact unions again, so the closest-hit contract holds and
// frames equal those of a freshly built tree.  For deformations that keep the spatial order roughly intact this is ~10x cheaper
// than a rebuild and keeps the quality of a host SAH tree.
//   k_refit_records   the walk's triangle record of every leaf-order slot from the new vertices (as k_expand_tris)
//   k_refit_links     parent node and slot of every BVH4 node, number of inner children per node
//   k_refit_nodes     per node: boxes of its leaf children from the records; then the LAST arrival at a node (its own thread and
//                     the threads coming up from its inner children) computes the node's union and carries it to the parent
__global__ void k_refit_records(const float *tri, const float4 *old_rec, uint32_t n_tri, const Scal *s, float4 *rec)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_tri) return;
    const float4 id_rec = old_rec[MCRT_TRI_PIECES * (size_t)slot], mesh_rec = old_rec[MCRT_TRI_PIECES * (size_t)slot + 1];
    const uint32_t id = __float_as_uint(id_rec.w);
    const float *v = tri + (size_t)id * 9;
    const float v0x = v[0], v0y = v[1], v0z = v[2], v1x = v[3], v1y = v[4], v1z = v[5], v2x = v[6], v2y = v[7], v2z = v[8];
    // the contract's expressions, exactly as k_expand_tris (mcrt_kernels.hip) evaluates them
    const float ax = v1x - v0x, ay = v1y - v0y, az = v1z - v0z, bx = v2x - v0x, by = v2y - v0y, bz = v2z - v0z;
    const float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    const float edge_tol = (nx * nx + ny * ny + nz * nz) * -0.0001f;
    float4 *o = rec + MCRT_TRI_PIECES * (size_t)slot;
    o[0] = make_float4(v0x, v0y, v0z, id_rec.w);
    o[1] = make_float4(v1x, v1y, v1z, mesh_rec.w);
    o[2] = make_float4(v2x, v2y, v2z, edge_tol);
    (void)s;
}

__global__ void k_refit_links(const float4 *nodes, uint32_t n4, int *parent, uint32_t *inner, uint32_t *arrived)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    uint32_t cnt = 0;
    for (int k = 0; k < 4; k++) {
        const int ref = __float_as_int(nodes[8 * (size_t)i + 2 * k + 1].z);
        if (ref >= 0) { parent[ref] = (int)(i * 4u + (uint32_t)k); cnt++; }
    }
    inner[i] = cnt; arrived[i] = 0u;
    if (i == 0) parent[0] = -1;
}

__global__ void k_refit_nodes(float4 *nodes, uint32_t n4, const float4 *rec, const Scal *s, const int *parent, const uint32_t *inner, uint32_t *arrived, float4 *root_box)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    for (int k = 0; k < 4; k++) {                       // this node's leaf children: union of their triangles' padded bounds
        const int ref = __float_as_int(nodes[8 * (size_t)i + 2 * k + 1].z);
        if (ref >= 0 || ref == MCRT_BVH4_EMPTY) continue;
        const uint32_t v = (uint32_t)~ref, first = v >> 3, cnt = (v & 7u) + 1u;
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (uint32_t t = 0; t < cnt; t++) {
            // the triangle's own padded bounds from its record's vertices -- bit for bit what k_prims / mcrt_build_bvh / the walk compute
            const float4 *rp = rec + MCRT_TRI_PIECES * (size_t)(first + t);
            const float4 a0 = rp[0], a1 = rp[1], a2 = rp[2];
            const float v[9] = { a0.x, a0.y, a0.z, a1.x, a1.y, a1.z, a2.x, a2.y, a2.z };
            float l[3], h[3], ext = 0.0f;
            for (int a = 0; a < 3; a++) {
                l[a] = fminf(v[a], fminf(v[3 + a], v[6 + a])); h[a] = fmaxf(v[a], fmaxf(v[3 + a], v[6 + a]));
                ext = fmaxf(ext, h[a] - l[a]);
            }
            const float pad = 2e-4f * ext + s->pad_abs;
            for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], l[a] - pad); hi[a] = fmaxf(hi[a], h[a] + pad); }
        }
        nodes[8 * (size_t)i + 2 * k] = make_float4(lo[0], lo[1], lo[2], hi[0]);
        nodes[8 * (size_t)i + 2 * k + 1] = make_float4(hi[1], hi[2], __int_as_float(ref), 0.0f);
    }
    uint32_t n = i;
    for (;;) {
        __threadfence();                                       // what this thread wrote is visible before it signs in
        if (atomicAdd(&arrived[n], 1u) != inner[n]) return;     // not the last of (own thread + inner children) to arrive at n
        __threadfence();                                       // acquire side: the children's boxes, published before THEIR arrivals, are read after ours
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (int k = 0; k < 4; k++) {
            const float4 A = load_coherent(&nodes[8 * (size_t)n + 2 * k]), B = load_coherent(&nodes[8 * (size_t)n + 2 * k + 1]);
            if (__float_as_int(B.z) == MCRT_BVH4_EMPTY) continue;
            lo[0] = fminf(lo[0], A.x); lo[1] = fminf(lo[1], A.y); lo[2] = fminf(lo[2], A.z);
            hi[0] = fmaxf(hi[0], A.w); hi[1] = fmaxf(hi[1], B.x); hi[2] = fmaxf(hi[2], B.y);
        }
        const int p = parent[n];
        if (p < 0) { root_box[0] = make_float4(lo[0], lo[1], lo[2], 0.0f); root_box[1] = make_float4(hi[0], hi[1], hi[2], 0.0f); return; }
        const uint32_t pn = (uint32_t)p >> 2, pk = (uint32_t)p & 3u;
        const float ref = nodes[8 * (size_t)pn + 2 * pk + 1].z;      // (the reference itself never changes)
        nodes[8 * (size_t)pn + 2 * pk] = make_float4(lo[0], lo[1], lo[2], hi[0]);
        nodes[8 * (size_t)pn + 2 * pk + 1] = make_float4(hi[1], hi[2], ref, 0.0f);
        n = pn;
    }
}

struct Temp {
    std::vector<void *> p;
    ~Temp() { for (void *x : p) hipFree(x); }
    template <class T> hipError_t get(T **out, size_t count) { void *x = nullptr; hipError_t e = hipMalloc(&x, count * sizeof(T) + 16); if (e == hipSuccess) { p.push_back(x); *out = (T *)x; } return e; }
};

#define LB_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return set_error(MCRT_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

}  // namespace

int lbvh_build(const float *tri_dev, const uint32_t *mesh_dev, uint32_t n_tri, hipStream_t st, LbvhResult *out)
{
    if (n_tri < 8u) return set_error(MCRT_ERR_INVALID, "device LBVH needs at least 8 triangles");
    if (n_tri >= (1u << 28)) return set_error(MCRT_ERR_LIMIT, "more than 2^28 triangles");
    const int n = (int)n_tri;
    int leaf_max = MCRT_LBVH_LEAF;                    // a subtree of at most this many triangles becomes one leaf
    if (const char *e = mcrt::tuning_env("MCRT_LBVH_LEAF")) { int v = atoi(e); if (v >= 1 && v <= 4) leaf_max = v; }   // tuning knob
    const dim3 blk(256), grid_t((n_tri + 255u) / 256u);
    Temp tmp;
    Scal *s = nullptr; float4 *plo, *phi, *pc, *ilo, *ihi; uint64_t *k0, *k1; uint32_t *v0, *v1, *arrivals, *depth, *is4, *id4; int2 *child, *range; int *par_i, *par_l; uint8_t *k4;
    LB_TRY(tmp.get(&s, 1)); LB_TRY(tmp.get(&plo, n_tri)); LB_TRY(tmp.get(&phi, n_tri)); LB_TRY(tmp.get(&pc, n_tri));
    LB_TRY(tmp.get(&ilo, n_tri)); LB_TRY(tmp.get(&ihi, n_tri));
    LB_TRY(tmp.get(&k0, n_tri)); LB_TRY(tmp.get(&k1, n_tri)); LB_TRY(tmp.get(&v0, n_tri)); LB_TRY(tmp.get(&v1, n_tri));
    LB_TRY(tmp.get(&arrivals, n_tri)); LB_TRY(tmp.get(&depth, n_tri)); LB_TRY(tmp.get(&is4, n_tri)); LB_TRY(tmp.get(&id4, n_tri));
    LB_TRY(tmp.get(&child, n_tri)); LB_TRY(tmp.get(&range, n_tri)); LB_TRY(tmp.get(&par_i, n_tri)); LB_TRY(tmp.get(&par_l, n_tri)); LB_TRY(tmp.get(&k4, n_tri));

    hipLaunchKernelGGL(k_scal_init, dim3(1), dim3(1), 0, st, s);
    hipLaunchKernelGGL(k_scale, dim3(1024), blk, 0, st, tri_dev, (size_t)n_tri * 9, s);
    hipLaunchKernelGGL(k_pad, dim3(1), dim3(1), 0, st, s);
    hipLaunchKernelGGL(k_prims, grid_t, blk, 0, st, tri_dev, n_tri, s, plo, phi, pc);
    hipLaunchKernelGGL(k_morton, grid_t, blk, 0, st, pc, n_tri, s, k0, v0);
    {
        size_t bytes = 0;
        LB_TRY(rocprim::radix_sort_pairs(nullptr, bytes, k0, k1, v0, v1, (size_t)n_tri, 0, 64, st));
        void *scratch = nullptr;
        LB_TRY(tmp.get((uint8_t **)&scratch, bytes));
        LB_TRY(rocprim::radix_sort_pairs(scratch, bytes, k0, k1, v0, v1, (size_t)n_tri, 0, 64, st));
    }
    LB_TRY(hipMemsetAsync(arrivals, 0, 4 * (size_t)n_tri, st));
    hipLaunchKernelGGL(k_karras, grid_t, blk, 0, st, k1, n, child, range, par_i, par_l);
    hipLaunchKernelGGL(k_fit, grid_t, blk, 0, st, child, par_i, par_l, v1, plo, phi, n, ilo, ihi, arrivals);
    hipLaunchKernelGGL(k_depth, grid_t, blk, 0, st, par_i, range, n, leaf_max, depth, is4, s);
    {
        size_t bytes = 0;
        LB_TRY(rocprim::exclusive_scan(nullptr, bytes, is4, id4, 0u, (size_t)(n_tri - 1u), rocprim::plus<uint32_t>(), st));
        void *scratch = nullptr;
        LB_TRY(tmp.get((uint8_t **)&scratch, bytes));
        LB_TRY(rocprim::exclusive_scan(scratch, bytes, is4, id4, 0u, (size_t)(n_tri - 1u), rocprim::plus<uint32_t>(), st));
    }
    uint32_t last_id = 0, last_flag = 0;
    LB_TRY(hipMemcpyAsync(&last_id, id4 + (n_tri - 2u), 4, hipMemcpyDeviceToHost, st));
    LB_TRY(hipMemcpyAsync(&last_flag, is4 + (n_tri - 2u), 4, hipMemcpyDeviceToHost, st));
    LB_TRY(hipStreamSynchronize(st));
    const uint32_t n4 = last_id + last_flag;

    float4 *nodes = nullptr, *tris = nullptr; uint32_t *slot = nullptr;
    LB_TRY(hipMalloc(&nodes, 128 * (size_t)n4));
    if (hipMalloc(&tris, 48 * (size_t)n_tri) != hipSuccess || hipMalloc(&slot, 4 * (size_t)n_tri) != hipSuccess) {
        hipFree(nodes); hipFree(tris); return set_error(MCRT_ERR_NOMEM, "device LBVH: out of device memory");
    }
    hipLaunchKernelGGL(k_emit4, grid_t, blk, 0, st, child, range, is4, id4, v1, plo, phi, ilo, ihi, n, leaf_max, nodes, k4);
    hipLaunchKernelGGL(k_emit_tris, grid_t, blk, 0, st, tri_dev, mesh_dev, v1, n_tri, tris, slot);
    hipLaunchKernelGGL(k_stack, grid_t, blk, 0, st, par_i, is4, k4, n, s);
    Scal hs; float4 root_lo, root_hi;
    hipError_t e = hipMemcpyAsync(&hs, s, sizeof hs, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&root_lo, ilo, 16, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&root_hi, ihi, 16, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { hipFree(nodes); hipFree(tris); hipFree(slot); return set_error(MCRT_ERR_HIP, "device LBVH: %s", hipGetErrorString(e)); }
    out->d_nodes = nodes; out->d_tris = tris; out->d_tri_slot = slot;
    out->n_nodes4 = n4; out->max_stack = hs.max_stack; out->max_depth = hs.max_depth; out->pad_abs = hs.pad_abs;
    out->lo[0] = root_lo.x; out->lo[1] = root_lo.y; out->lo[2] = root_lo.z;
    out->hi[0] = root_hi.x; out->hi[1] = root_hi.y; out->hi[2] = root_hi.z;
    return MCRT_OK;
}

int bvh_refit(const float *tri_dev, uint32_t n_tri, float4 *d_nodes, uint32_t n_nodes4, float4 *d_recs, hipStream_t st, float *pad_abs, float lo[3], float hi[3])
{
    if (n_tri == 0 || n_nodes4 == 0) return set_error(MCRT_ERR_INVALID, "refit: no tree");
    Temp tmp;
    Scal *s = nullptr; float4 *new_rec = nullptr, *root = nullptr; int *parent = nullptr; uint32_t *inner = nullptr, *arrived = nullptr;
    LB_TRY(tmp.get(&s, 1)); LB_TRY(tmp.get(&new_rec, MCRT_TRI_PIECES * (size_t)n_tri)); LB_TRY(tmp.get(&root, 2));
    LB_TRY(tmp.get(&parent, n_nodes4)); LB_TRY(tmp.get(&inner, n_nodes4)); LB_TRY(tmp.get(&arrived, n_nodes4));
    const dim3 blk(256), grid_t((n_tri + 255u) / 256u), grid_n((n_nodes4 + 255u) / 256u);
    hipLaunchKernelGGL(k_scal_init, dim3(1), dim3(1), 0, st, s);
    hipLaunchKernelGGL(k_scale, dim3(1024), blk, 0, st, tri_dev, (size_t)n_tri * 9, s);
    hipLaunchKernelGGL(k_pad, dim3(1), dim3(1), 0, st, s);
    hipLaunchKernelGGL(k_refit_records, grid_t, blk, 0, st, tri_dev, (const float4 *)d_recs, n_tri, (const Scal *)s, new_rec);
    LB_TRY(hipMemcpyAsync(d_recs, new_rec, 16 * MCRT_TRI_PIECES * (size_t)n_tri, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_refit_links, grid_n, blk, 0, st, (const float4 *)d_nodes, n_nodes4, parent, inner, arrived);
    hipLaunchKernelGGL(k_refit_nodes, grid_n, blk, 0, st, d_nodes, n_nodes4, (const float4 *)d_recs, (const Scal *)s, (const int *)parent, (const uint32_t *)inner, arrived, root);
    Scal hs; float4 rb[2];
    LB_TRY(hipMemcpyAsync(&hs, s, sizeof hs, hipMemcpyDeviceToHost, st));
    LB_TRY(hipMemcpyAsync(rb, root, sizeof rb, hipMemcpyDeviceToHost, st));
    LB_TRY(hipStreamSynchronize(st));
    LB_TRY(hipGetLastError());
    *pad_abs = hs.pad_abs;
    lo[0] = rb[0].x; lo[1] = rb[0].y; lo[2] = rb[0].z; hi[0] = rb[1].x; hi[1] = rb[1].y; hi[2] = rb[1].z;
    return MCRT_OK;
}

}  // namespace mcrt
