// Marching cubes over the TSDF's active-voxel bitmap, gfx950.
//
// Replaces the reference's only native code (paths relative to /root/reference/src/doubletake/):
//   tools/marching_cubes/marching_cubes.cu:164-250  ClassifyVoxelKernel
//   tools/marching_cubes/marching_cubes.cu:263-278  CompactVoxelsKernel (+ two thrust scans)
//   tools/marching_cubes/marching_cubes.cu:294-424  GenerateFacesKernel
//   tools/marching_cubes/marching_cubes.cu:455-597  MarchingCubesCuda (host)
// keeping the CUDA path's semantics: only cells whose base voxel is in the active set, inside
// [min_bounds, max_bounds) and inside the volume; a cell with any corner < -0.99999 (unobserved)
// is skipped; case index bit vi set when value(vi) < isolevel; vertex interpolation snaps to an
// endpoint when |iso - v| < 1e-5 or |v1 - v2| < 1e-5; edge id = v1*(W + W*H + W*H*D) + v2.
//
// Differences by design: the active set is a bitmap in voxel-id order instead of
// an open3d HashSet list, so output order is deterministic (ascending voxel id); the cell count
// comes from the occupancy flags (the reference's voxelVerts[N-1]/voxelOccupied[N-1] mix-up is not
// reproduced); the half volume is read directly (the reference first materialises a float copy);
// one host read of two ints instead of four .item() syncs + two device synchronisations.
//
// Two phases over the same thread->voxel mapping (256 consecutive voxel ids per block):
//   count:    per-block vertex totals -> two-level exclusive scan (chunks of 4096 blocks) -> totals in counts_out
//   generate: reclassify, block-local scan of triangle counts in LDS, one thread per triangle writes vertices / faces / ids
#include "common.hpp"
#include "mc_tables.hpp"
#include "raster_device.hpp"

namespace dt {

typedef _Float16 half_t;

__device__ __forceinline__ float ldh(const uint16_t* p, size_t i) {
  half_t h;
  const uint16_t b = p[i];
  __builtin_memcpy(&h, &b, 2);
  return (float)h;
}

struct McArgs {
  const uint16_t* vol;
  const uint32_t* active;
  int X, Y, Z;
  float iso;
  int mn[3], mx[3];
};
DT_ARG_POINTERS(McArgs, offsetof(McArgs, vol), offsetof(McArgs, active));

// corner code c = dx + 2*dy + 4*dz (dx along Z/k, dy along Y/j, dz along X/i) -> Bourke corner vi
__device__ __constant__ const unsigned char kCodeToVi[8] = {0, 1, 4, 5, 3, 2, 7, 6};
// Bourke edge -> its two corner codes, always low -> high coordinate
__device__ __constant__ const unsigned char kEdgeCodes[12][2] = {{0, 1}, {1, 5}, {4, 5}, {0, 4}, {2, 3}, {3, 7},
                                                                {6, 7}, {2, 6}, {0, 2}, {1, 3}, {5, 7}, {4, 6}};

// returns the number of vertices the cell at voxel id emits (0 if inactive / out of bounds /
// unobserved); fills case index and corner values when non-zero
__device__ __forceinline__ int classify(const McArgs& a, size_t id, int& i, int& j, int& k, int& cubeindex,
                                        float val[8]) {
  const size_t total = (size_t)a.X * a.Y * a.Z;
  if (id >= total) return 0;
  if (!((a.active[id >> 5] >> (id & 31)) & 1u)) return 0;
  k = (int)(id % a.Z);
  j = (int)((id / a.Z) % a.Y);
  i = (int)(id / ((size_t)a.Z * a.Y));
  if (i >= a.X - 1 || j >= a.Y - 1 || k >= a.Z - 1) return 0;
  if (i < a.mn[0] || j < a.mn[1] || k < a.mn[2] || i >= a.mx[0] || j >= a.mx[1] || k >= a.mx[2]) return 0;
  cubeindex = 0;
  bool invalid = false;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int dx = c & 1, dy = (c >> 1) & 1, dz = (c >> 2) & 1;
    const float v = ldh(a.vol, ((size_t)(i + dz) * a.Y + (j + dy)) * a.Z + (k + dx));
    if (v < a.iso) cubeindex |= 1 << kCodeToVi[c];
    if (v < -0.99999f) invalid = true;
    val[c] = v;
  }
  if (invalid) return 0;
  return 3 * (int)kMcTris[cubeindex];
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* lds, int& block_total) {
  // 256 threads; simple Hillis-Steele in LDS
  const int t = threadIdx.x;
  lds[t] = v;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int add = (t >= off) ? lds[t - off] : 0;
    __syncthreads();
    lds[t] += add;
    __syncthreads();
  }
  block_total = lds[255];
  const int excl = lds[t] - v;
  __syncthreads();
  return excl;
}

__global__ __launch_bounds__(256) void mc_count_kernel(const McArgs a, int* __restrict__ block_sums,
                                                      int* __restrict__ block_cells) {
  __shared__ int wsum[4], wcells[4];
  const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
  // a workgroup covers eight words of the active bitmap; where all eight are zero (most of a room-sized volume) nothing
  // can be emitted: one scalar-cache read per wave instead of 256 classifications and two block scans
  const uint4* aw = reinterpret_cast<const uint4*>(a.active + (size_t)blockIdx.x * 8);
  const uint4 w0 = aw[0], w1 = aw[1];
  if ((w0.x | w0.y | w0.z | w0.w | w1.x | w1.y | w1.z | w1.w) == 0u) {
    if (threadIdx.x == 0) {
      block_sums[blockIdx.x] = 0;
      block_cells[blockIdx.x] = 0;
    }
    return;
  }
  int i, j, k, ci;
  float val[8];
  const int n = classify(a, id, i, j, k, ci, val);
  // totals only: wave reduction + four partials through LDS (the per-thread offsets are recomputed by mc_generate)
  int tot = n, cells = n > 0 ? 1 : 0;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    tot += __shfl_xor(tot, m, 64);
    cells += __shfl_xor(cells, m, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    wsum[threadIdx.x >> 6] = tot;
    wcells[threadIdx.x >> 6] = cells;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    block_cells[blockIdx.x] = wcells[0] + wcells[1] + wcells[2] + wcells[3];
  }
}

// Exclusive scan of block_sums in two launches (round 4: the single-workgroup version walked 98 strided elements per thread
// -- 258 us at 0.02 m, more than count and generate together).
//   mc_scan1: one 1024-thread workgroup per chunk of 4096 block sums: coalesced int4 load, exclusive scan inside the chunk
//             (written back in place), chunk total of vertices and of cells
//   mc_scan2: one workgroup scans the chunk totals (exclusive, in place) and writes the grand totals to counts_out
// mc_generate adds the chunk prefix to the in-chunk offset.
constexpr int kScanChunk = 4096;

__global__ __launch_bounds__(1024) void mc_scan1_kernel(int* __restrict__ block_sums, const int* __restrict__ block_cells,
                                                       int nblocks, long long* __restrict__ chunk_tot,
                                                       long long* __restrict__ chunk_cells) {
  __shared__ int wtot[16];
  __shared__ int ctot[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int e0 = blockIdx.x * kScanChunk + t * 4;
  int v[4], c = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[q] = (e0 + q < nblocks) ? block_sums[e0 + q] : 0;
    c += (e0 + q < nblocks) ? block_cells[e0 + q] : 0;
  }
  const int mine = v[0] + v[1] + v[2] + v[3];
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
  if (lane == 63) wtot[wave] = incl;
  if (lane == 0) ctot[wave] = c;
  __syncthreads();
  int base = 0, total = 0, cells = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int x = wtot[w];
    if (w < wave) base += x;
    total += x;
    cells += ctot[w];
  }
  int run = base + incl - mine;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (e0 + q < nblocks) block_sums[e0 + q] = run;
    run += v[q];
  }
  if (t == 0) {
    chunk_tot[blockIdx.x] = total;    // (a chunk holds at most 4096 x 256 x 15 vertices: fits an int; the prefix may not)
    chunk_cells[blockIdx.x] = cells;
  }
}

__global__ __launch_bounds__(1024) void mc_scan2_kernel(long long* __restrict__ chunk_tot, const long long* __restrict__ chunk_cells,
                                                       int nchunks, int* __restrict__ counts_out) {
  __shared__ long long wtot[16];
  __shared__ long long ctot[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int per = (nchunks + 1023) / 1024;
  const int b0 = min(t * per, nchunks), b1 = min(b0 + per, nchunks);
  long long s = 0, c = 0;
  for (int b = b0; b < b1; ++b) {
    s += chunk_tot[b];
    c += chunk_cells[b];
  }
  long long incl = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const long long up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
  if (lane == 63) wtot[wave] = incl;
  if (lane == 0) ctot[wave] = c;
  __syncthreads();
  long long base = 0, total = 0, cells = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const long long v = wtot[w];
    if (w < wave) base += v;
    total += v;
    cells += ctot[w];
  }
  if (t == 0) {
    counts_out[0] = (int)cells;
    counts_out[1] = (total > 2147483647LL) ? -1 : (int)total;
  }
  long long run = base + incl - s;
  for (int b = b0; b < b1; ++b) {
    const long long v = chunk_tot[b];
    chunk_tot[b] = run;
    run += v;
  }
}

__device__ __forceinline__ void vertex_interp(float iso, float p1x, float p1y, float p1z, float p2x, float p2y,
                                              float p2z, float v1, float v2, float& ox, float& oy, float& oz) {
  const float eps = 1e-5f;
  if (fabsf(iso - v1) < eps) {
    ox = p1x; oy = p1y; oz = p1z;
    return;
  }
  if (fabsf(iso - v2) < eps) {
    ox = p2x; oy = p2y; oz = p2z;
    return;
  }
  if (fabsf(v1 - v2) < eps) {
    ox = p1x; oy = p1y; oz = p1z;
    return;
  }
  const float ratio = (iso - v1) / (v2 - v1);
  ox = p1x * (1 - ratio) + p2x * ratio;
  oy = p1y * (1 - ratio) + p2y * ratio;
  oz = p1z * (1 - ratio) + p2z * ratio;
}

// One thread per TRIANGLE (round 4; was one thread per cell writing up to 15 vertices with 12-byte strided stores): the
// workgroup classifies its 256 cells, compacts (cell, triangle) records through LDS -- the scheme of mc_raster_kernel below -- and
// thread q then builds triangle q of the workgroup: 36 contiguous bytes of vertices, 24 of edge ids, 24 of face indices, so a wave
// writes whole cache lines.  Output order unchanged: ascending voxel id, the case table's triangle order within a cell.
__global__ __launch_bounds__(256) void mc_generate_kernel(const McArgs a, const int* __restrict__ block_offsets,
                                                         const long long* __restrict__ chunk_prefix, float* __restrict__ verts, int64_t* __restrict__ faces,
                                                         int64_t* __restrict__ ids, int num_verts) {
  __shared__ int lds[256];
  __shared__ float cell_val[256][9];           // 8 corner values (+1 pad)
  __shared__ int cell_ci[256];
  __shared__ unsigned short tri_rec[256 * 5];  // (owner thread << 3) | triangle index within the cell
  const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
  {  // (same early exit as mc_count_kernel: no active voxel in this workgroup's eight bitmap words)
    const uint4* aw = reinterpret_cast<const uint4*>(a.active + (size_t)blockIdx.x * 8);
    const uint4 w0 = aw[0], w1 = aw[1];
    if ((w0.x | w0.y | w0.z | w0.w | w1.x | w1.y | w1.z | w1.w) == 0u) return;
  }
  int i = 0, j = 0, k = 0, ci = 0;
  float val[8];
  const int ntri = classify(a, id, i, j, k, ci, val) / 3;
  if (__syncthreads_or(ntri) == 0) return;   // active voxels, but no surface cell
  int tot;
  const int first = block_exclusive_scan(ntri, lds, tot);
  if (ntri > 0) {
#pragma unroll
    for (int c = 0; c < 8; ++c) cell_val[threadIdx.x][c] = val[c];
    cell_ci[threadIdx.x] = ci;
    for (int t = 0; t < ntri; ++t) tri_rec[first + t] = (unsigned short)((threadIdx.x << 3) | t);
  }
  __syncthreads();
  const size_t base_id = (size_t)blockIdx.x * 256;
  const long long tri_base = ((long long)block_offsets[blockIdx.x] + chunk_prefix[blockIdx.x / kScanChunk]) / 3;
  // MC naming: x = k (Z axis, fastest), y = j, z = i;  W = Z, H = Y, D = X
  const long long W = a.Z, H = a.Y, D = a.X;
  const long long hash_mul = W + W * H + W * H * D;
  for (int q = threadIdx.x; q < tot; q += 256) {
    const int rec = tri_rec[q];
    const int owner = rec >> 3, t3 = (rec & 7) * 3;
    const size_t oid = base_id + owner;
    const int ok = (int)(oid % a.Z), oj = (int)((oid / a.Z) % a.Y), oi = (int)(oid / ((size_t)a.Z * a.Y));
    const int oci = cell_ci[owner];
    const long long tri = tri_base + q;
    if (tri * 3 + 2 >= num_verts) continue;
    float* vo = verts + tri * 9;
    int64_t* io = ids + tri * 3;
    int64_t* fo = faces + tri * 3;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int e = kMcEdges[oci][t3 + v];
      const int c1 = kEdgeCodes[e][0], c2 = kEdgeCodes[e][1];
      const int x1 = ok + (c1 & 1), y1 = oj + ((c1 >> 1) & 1), z1 = oi + ((c1 >> 2) & 1);
      const int x2 = ok + (c2 & 1), y2 = oj + ((c2 >> 1) & 1), z2 = oi + ((c2 >> 2) & 1);
      float ox, oy, oz;
      vertex_interp(a.iso, (float)x1, (float)y1, (float)z1, (float)x2, (float)y2, (float)z2, cell_val[owner][c1],
                    cell_val[owner][c2], ox, oy, oz);
      vo[v * 3 + 0] = ox;
      vo[v * 3 + 1] = oy;
      vo[v * 3 + 2] = oz;
      const long long v1 = x1 + y1 * W + z1 * W * H, v2 = x2 + y2 * W + z2 * W * H;
      io[v] = v1 * hash_mul + v2;
      fo[v] = tri * 3 + v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused marching cubes -> depth render for the per-frame hint of the incremental mode (test_incremental.py:204-258:
// tsdf.to_mesh_pytorch3d -> MeshRasterizer zbuf).  A depth render needs neither the merged vertex list nor any
// triangle order, only every triangle once: each workgroup classifies its 256 cells, compacts the triangles it found
// through LDS (so that 256 lanes rasterise 256 triangles, whichever cells they came from) and rasterises them into the
// z-buffer with the same raster_triangle() the stand-alone renderer uses.  No vertex buffer, no global scan, no vertex
// count on the host: the count -> scan -> host read -> generate -> raster sequence (5 launches + one synchronisation per
// frame) becomes one launch the host never waits for.  Same triangles, same vertex arithmetic, order-independent
// atomicMin: bit-identical to dt_mc_generate + dt_raster_soup_depth_f32.
// ------------------------------------------------------------------------------------------------------------------
struct McRasterArgs {
  float ox, oy, oz, vs;
  const float* cam_T_world;
  const float* K;
  int h, w;
  uint32_t* zb;
};
DT_ARG_POINTERS(McRasterArgs, offsetof(McRasterArgs, cam_T_world), offsetof(McRasterArgs, K), offsetof(McRasterArgs, zb));

__global__ __launch_bounds__(256) void mc_raster_kernel(const McArgs a, const McRasterArgs r) {
  __shared__ int lds[256];
  __shared__ float cell_val[256][9];       // 8 corner values (+1 pad: lanes of a wave on different banks)
  __shared__ int cell_ci[256];
  __shared__ unsigned short tri_rec[256 * 5];  // (owner thread << 3) | triangle index within the cell
  const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
  {  // no active voxel in this workgroup's eight bitmap words: leave before any classification or barrier
    const uint4* aw = reinterpret_cast<const uint4*>(a.active + (size_t)blockIdx.x * 8);
    const uint4 w0 = aw[0], w1 = aw[1];
    if ((w0.x | w0.y | w0.z | w0.w | w1.x | w1.y | w1.z | w1.w) == 0u) return;
  }
  int i = 0, j = 0, k = 0, ci = 0;
  float val[8];
  const int ntri = classify(a, id, i, j, k, ci, val) / 3;
  if (__syncthreads_or(ntri) == 0) return;   // (active voxels, but no surface in their 256 cells)
  int tot;
  const int first = block_exclusive_scan(ntri, lds, tot);
  if (ntri > 0) {
#pragma unroll
    for (int c = 0; c < 8; ++c) cell_val[threadIdx.x][c] = val[c];
    cell_ci[threadIdx.x] = ci;
    for (int t = 0; t < ntri; ++t) tri_rec[first + t] = (unsigned short)((threadIdx.x << 3) | t);
  }
  __syncthreads();
  const size_t base_id = (size_t)blockIdx.x * 256;
  for (int q = threadIdx.x; q < tot; q += 256) {
    const int rec = tri_rec[q];
    const int owner = rec >> 3, t3 = (rec & 7) * 3;
    const size_t oid = base_id + owner;
    const int ok = (int)(oid % a.Z), oj = (int)((oid / a.Z) % a.Y), oi = (int)(oid / ((size_t)a.Z * a.Y));
    const int oci = cell_ci[owner];
    float X[3], Y[3], Z[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int e = kMcEdges[oci][t3 + v];
      const int c1 = kEdgeCodes[e][0], c2 = kEdgeCodes[e][1];
      const int x1 = ok + (c1 & 1), y1 = oj + ((c1 >> 1) & 1), z1 = oi + ((c1 >> 2) & 1);
      const int x2 = ok + (c2 & 1), y2 = oj + ((c2 >> 1) & 1), z2 = oi + ((c2 >> 2) & 1);
      float px, py, pz;   // (k, j, i) voxel-index coordinates, as dt_mc_generate writes them
      vertex_interp(a.iso, (float)x1, (float)y1, (float)z1, (float)x2, (float)y2, (float)z2, cell_val[owner][c1],
                    cell_val[owner][c2], px, py, pz);
      X[v] = r.ox + pz * r.vs;   // world = origin + (i, j, k) * voxel_size (raster_soup_kernel)
      Y[v] = r.oy + py * r.vs;
      Z[v] = r.oz + px * r.vs;
    }
    raster_triangle(X, Y, Z, r.cam_T_world, r.K, r.h, r.w, r.zb);
  }
}

static int fill(McArgs& a, const uint16_t* vol, const uint32_t* active, int X, int Y, int Z, float iso, const int* mn,
                const int* mx, const char* who) {
  DT_REQUIRE(vol && active, "%s: null pointer", who);
  DT_REQUIRE(X > 1 && Y > 1 && Z > 1, "%s: bad volume extent", who);
  DT_REQUIRE(((size_t)X * Y * Z) % 256 == 0, "%s: voxel count must be a multiple of 256 (dims are multiples of 8)", who);
  // every workgroup reads its eight bitmap words as two uint4 (early exit on an empty block)
  DT_REQUIRE((reinterpret_cast<uintptr_t>(active) & 15) == 0, "%s: the active bitmap must be 16-byte aligned", who);
  a.vol = vol;
  a.active = active;
  a.X = X;
  a.Y = Y;
  a.Z = Z;
  a.iso = iso;
  for (int q = 0; q < 3; ++q) {
    a.mn[q] = mn ? mn[q] : -2147483647;
    a.mx[q] = mx ? mx[q] : 2147483647;
  }
  return 0;
}

}  // namespace dt

using namespace dt;

extern "C" {

int64_t dt_mc_workspace_bytes(int X, int Y, int Z) {
  const size_t nblocks = ((size_t)X * Y * Z + 255) / 256;
  const size_t nchunks = (nblocks + kScanChunk - 1) / kScanChunk;
  // [block sums | block cells] ints (padded to 8 bytes), then [chunk totals | chunk cells] 64-bit
  return (int64_t)(((2 * nblocks + 1) / 2 * 2) * sizeof(int) + 2 * nchunks * sizeof(long long));
}

int dt_mc_count(const uint16_t* values, const uint32_t* active, int X, int Y, int Z, float isolevel, const int* mn,
                const int* mx, void* workspace, int* counts_out, dt_stream_t s) {
  McArgs a;
  if (int rc = fill(a, values, active, X, Y, Z, isolevel, mn, mx, "dt_mc_count")) return rc;
  DT_REQUIRE(workspace && counts_out, "dt_mc_count: null workspace");
  const size_t nblocks = (size_t)X * Y * Z / 256;
  DT_REQUIRE(nblocks < 2147483647ull, "dt_mc_count: volume too large");
  int* sums = reinterpret_cast<int*>(workspace);
  int* cells = sums + nblocks;
  const size_t nchunks = (nblocks + kScanChunk - 1) / kScanChunk;
  long long* ctot = reinterpret_cast<long long*>(sums + (2 * nblocks + 1) / 2 * 2);
  long long* ccells = ctot + nchunks;
  DT_LAUNCH(mc_count_kernel, dim3((unsigned)nblocks), dim3(256), 0, to_stream(s), a, sums, cells);
  DT_LAUNCH(mc_scan1_kernel, dim3((unsigned)nchunks), dim3(1024), 0, to_stream(s), sums, cells, (int)nblocks, ctot, ccells);
  DT_LAUNCH(mc_scan2_kernel, dim3(1), dim3(1024), 0, to_stream(s), ctot, ccells, (int)nchunks, counts_out);
  return check_launch("dt_mc_count");
}

int dt_mc_generate(const uint16_t* values, const uint32_t* active, int X, int Y, int Z, float isolevel, const int* mn,
                   const int* mx, const void* workspace, float* verts, int64_t* faces, int64_t* ids, int num_verts,
                   dt_stream_t s) {
  McArgs a;
  if (int rc = fill(a, values, active, X, Y, Z, isolevel, mn, mx, "dt_mc_generate")) return rc;
  DT_REQUIRE(workspace, "dt_mc_generate: null workspace");
  DT_REQUIRE(num_verts >= 0 && num_verts % 3 == 0, "dt_mc_generate: num_verts=%d", num_verts);
  if (num_verts == 0) return 0;
  DT_REQUIRE(verts && faces && ids, "dt_mc_generate: null output");
  const size_t nblocks = (size_t)X * Y * Z / 256;
  const size_t nchunks = (nblocks + kScanChunk - 1) / kScanChunk;
  const int* sums = reinterpret_cast<const int*>(workspace);
  const long long* cprefix = reinterpret_cast<const long long*>(sums + (2 * nblocks + 1) / 2 * 2);
  (void)nchunks;
  DT_LAUNCH(mc_generate_kernel, dim3((unsigned)nblocks), dim3(256), 0, to_stream(s), a, sums, cprefix, verts, faces, ids,
                     num_verts);
  return check_launch("dt_mc_generate");
}

int dt_mc_raster_depth_f32(const uint16_t* values, const uint32_t* active, int X, int Y, int Z, float isolevel, const int* mn,
                           const int* mx, const float* origin3, float voxel_size, const float* cam_T_world_44,
                           const float* K_44, int h, int w, uint32_t* workspace_hw, float* depth_hw, dt_stream_t s) {
  McArgs a;
  if (int rc = fill(a, values, active, X, Y, Z, isolevel, mn, mx, "dt_mc_raster_depth_f32")) return rc;
  DT_REQUIRE(origin3 && cam_T_world_44 && K_44 && workspace_hw && depth_hw, "dt_mc_raster_depth_f32: null pointer");
  DT_REQUIRE(h > 0 && w > 0 && voxel_size > 0.f, "dt_mc_raster_depth_f32: bad extents");
  const size_t nblocks = (size_t)X * Y * Z / 256;
  DT_REQUIRE(nblocks < 2147483647ull, "dt_mc_raster_depth_f32: volume too large");
  McRasterArgs r;
  r.ox = origin3[0]; r.oy = origin3[1]; r.oz = origin3[2]; r.vs = voxel_size;
  r.cam_T_world = cam_T_world_44; r.K = K_44; r.h = h; r.w = w; r.zb = workspace_hw;
  const int64_t n = (int64_t)h * w;
  hipStream_t st = to_stream(s);
  DT_LAUNCH(raster_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace_hw, n);
  DT_LAUNCH(mc_raster_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, a, r);
  DT_LAUNCH(raster_resolve_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace_hw, depth_hw, n);
  return check_launch("dt_mc_raster_depth_f32");
}

}  // extern "C"
