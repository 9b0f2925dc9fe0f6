// Development microbenchmark (not part of the product): what the NEIGHBOUR TABLES of the DBSCAN batch cost.  k_db_union gives every
// active core cell a wave; lane l looks at neighbour cells l and l + 64 of the 5x5x5 block around the cell and loads their entries
// of four per-cell tables (minidx, active, parent, cellpos: 25 grid columns x 4 tables = 100 cache lines per wave), then a 48-byte
// box through cellpos.  Is the kernel's ~38 us a matter of how many lines a wave asks for?
//   mode 0: four u32 tables            mode 1: one table of 8-byte records       mode 2: one table of 16-byte records
//   mode 3: as 0 plus the dependent 48-byte box load      mode 4: as 1 plus the box      mode 5: 32-byte record holding a float box
// The tables are re-written by another kernel before every timed launch (as k_db_core does: the lines sit in another XCD's L2 or
// in memory, not in the reader's).     hipcc --offload-arch=gfx950 -O3 -o line_bench line_bench.hip && ./line_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Rec8 { unsigned a, b; };
struct Rec16 { unsigned a, b, c, d; };
struct Rec32 { unsigned a, b; float box[6]; };

__global__ void k_write(unsigned* t0, unsigned* t1, unsigned* t2, unsigned* t3, Rec8* r8, Rec16* r16, Rec32* r32, double* box, long long NC, unsigned salt) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC) return;
    const unsigned v = (unsigned)(i * 2654435761u) ^ salt;
    const unsigned occupied = (v >> 7) % 4u == 0u ? 1u : 0u;      // a quarter of the cells hold core points
    const unsigned pos = (unsigned)((v >> 3) % (unsigned)NC);
    t0[i] = occupied ? v : 0xffffffffu;
    t1[i] = v & 1u;
    t2[i] = (unsigned)i;
    t3[i] = pos;
    r8[i] = Rec8{(unsigned)i, occupied ? pos : 0xffffffffu};
    r16[i] = Rec16{occupied ? v : 0xffffffffu, v & 1u, (unsigned)i, pos};
    Rec32 r;
    r.a = (unsigned)i;
    r.b = occupied ? pos : 0xffffffffu;
    for (int a = 0; a < 6; ++a) {
        r.box[a] = (float)a;
        box[(size_t)i * 6 + a] = (double)a;
    }
    r32[i] = r;
}

__global__ void k_read(const int* cells, int ncells, int nx, int ny, int nz, const unsigned* t0, const unsigned* t1, const unsigned* t2,
                       const unsigned* t3, const Rec8* r8, const Rec16* r16, const Rec32* r32, const double* box, int mode, unsigned* out) {
    const int lane = threadIdx.x & 63;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
    double acc = 0.0;
    for (unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < (unsigned)ncells; w += nwaves) {
        const int c = cells[w];
        const int iz = c % nz, iy = (c / nz) % ny, ix = c / (nz * ny);
        for (int q = 0; q < 2; ++q) {
            const int o = lane + 64 * q;
            const int jx = ix + o / 25 - 2, jy = iy + (o / 5) % 5 - 2, jz = iz + o % 5 - 2;
            if (o >= 125 || jx < 0 || jy < 0 || jz < 0 || jx >= nx || jy >= ny || jz >= nz) continue;
            const long long c2 = ((long long)jx * ny + jy) * nz + jz;
            unsigned mi, pos, extra = 0;
            if (mode == 0 || mode == 3) {
                mi = t0[c2];
                extra = t1[c2] + t2[c2];
                pos = t3[c2];
            } else if (mode == 1 || mode == 4) {
                const Rec8 r = r8[c2];
                mi = r.b;
                extra = r.a;
                pos = r.b;
            } else if (mode == 2) {
                const Rec16 r = r16[c2];
                mi = r.a;
                extra = r.b + r.c;
                pos = r.d;
            } else {
                const Rec32 r = r32[c2];
                mi = r.b;
                extra = r.a;
                pos = r.b;
                for (int a = 0; a < 6; ++a) acc += (double)r.box[a];
            }
            acc += (double)extra;
            if ((mode == 3 || mode == 4) && mi != 0xffffffffu)
                for (int a = 0; a < 6; ++a) acc += box[(size_t)pos * 6 + a];
        }
    }
    if (acc == 1.2345) out[0] = 1u;
}

int main() {
    const int nx = 76, ny = 48, nz = 64;                       // 233 472 cells (configs[1]'s mean batch: 233 150)
    const long long NC = (long long)nx * ny * nz;
    const int ncells = 6864;                                   // active core cells of a mean batch
    unsigned *t0, *t1, *t2, *t3, *out;
    Rec8* r8;
    Rec16* r16;
    Rec32* r32;
    double* box;
    int* cells;
    CK(hipMalloc(&t0, NC * 4)); CK(hipMalloc(&t1, NC * 4)); CK(hipMalloc(&t2, NC * 4)); CK(hipMalloc(&t3, NC * 4));
    CK(hipMalloc(&r8, NC * 8)); CK(hipMalloc(&r16, NC * 16)); CK(hipMalloc(&r32, NC * 32)); CK(hipMalloc(&box, NC * 48));
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&cells, ncells * 4));
    std::vector<int> hc(ncells);
    unsigned seed = 12345u;
    for (int i = 0; i < ncells; ++i) {                         // a slab two cells thick (a surface), random cells of it
        seed = seed * 1664525u + 1013904223u;
        const int x = (int)((seed >> 8) % (unsigned)nx);
        seed = seed * 1664525u + 1013904223u;
        const int y = (int)((seed >> 8) % (unsigned)ny);
        seed = seed * 1664525u + 1013904223u;
        const int z = 30 + (int)((seed >> 8) % 2u);
        hc[i] = (x * ny + y) * nz + z;
    }
    CK(hipMemcpy(cells, hc.data(), ncells * 4, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const unsigned gW = (unsigned)prop.multiProcessorCount * 8u;
    const char* names[6] = {"four u32 tables", "one 8-byte record", "one 16-byte record", "four tables + f64 box", "8-byte record + f64 box", "32-byte record with f32 box"};
    for (int mode = 0; mode < 6; ++mode) {
        float best = 1e9f, sum = 0.f;
        const int reps = 20;
        for (int r = 0; r < reps + 2; ++r) {
            hipLaunchKernelGGL(k_write, dim3((unsigned)((NC + 255) / 256)), dim3(256), 0, s, t0, t1, t2, t3, r8, r16, r32, box, NC, (unsigned)r);
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(k_read, dim3(gW), dim3(256), 0, s, (const int*)cells, ncells, nx, ny, nz, (const unsigned*)t0, (const unsigned*)t1,
                               (const unsigned*)t2, (const unsigned*)t3, (const Rec8*)r8, (const Rec16*)r16, (const Rec32*)r32, (const double*)box, mode, out);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) {
                best = ms < best ? ms : best;
                sum += ms;
            }
        }
        printf("mode %d (%-28s): %d waves, mean %.1f us, best %.1f us\n", mode, names[mode], ncells, sum / reps * 1e3f, best * 1e3f);
    }
    return 0;
}
